"""Turn the rocprofv3 `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` counter_collection CSVs (two separate
passes over the same command) into profiles/pmc_traffic.json: average HBM bytes per launch of every kernel.

  python scripts/pmc_aggregate.py <fetch_counter_collection.csv> <write_counter_collection.csv> > profiles/pmc_traffic.json

Counters are in KB.  FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md;
confirmed on the box by scripts/pmc_calibrate.py), WRITE_SIZE is exact."""
import csv
import json
import re
import sys
from collections import defaultdict


def short_name(full):
    m = re.search(r"(?:dram::)?([A-Za-z_0-9]+)(<[^>(]*>)?\s*\(", full)
    if not m:
        return full.strip()
    return m.group(1) + (m.group(2) or "")


def collect(path, counter, scale):
    per = defaultdict(lambda: defaultdict(float))   # kernel -> dispatch -> value (summed over XCC instances)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            per[short_name(row["Kernel_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"]) * 1024.0 * scale
    return {k: (len(v), sum(v.values()) / len(v)) for k, v in per.items()}


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE", 2.0)
    write = collect(sys.argv[2], "WRITE_SIZE", 1.0)
    out = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 "
                      "--warmup 1` (the default workload: 64 chunks of 128^3 as one batch, two steps; scripts/profile_round.sh); "
                      "counters are in KB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B "
                      "requests at 64 B) -- confirmed by scripts/pmc_calibrate.py; aggregated by scripts/pmc_aggregate.py",
           "_unit": "HBM bytes per launch (average over the launches of that kernel in the run)"}
    for k in sorted(set(fetch) | set(write)):
        if "at::" in k or k.startswith("void ") or k.startswith("__amd") or "_kernel" not in k or "elementwise" in k \
                or k.startswith("direct_copy"):
            continue          # torch's own kernels (optimizer, fills): not ours to account for
        n = fetch.get(k, write.get(k))[0]
        r, w = fetch.get(k, (0, 0.0))[1], write.get(k, (0, 0.0))[1]
        out[k] = {"launches": n, "read_bytes": r, "write_bytes": w, "total_bytes": r + w}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()

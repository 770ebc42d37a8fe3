#!/bin/bash
# HBM read bytes (FETCH_SIZE, KB; x2 on gfx950) of the backward-weights kernel on one layer shape, for two builds of the
# library (diagnostics: $1 = alternative library).  DRAM_HIP_LIB is read by the Python process, not by the profiler.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_wgrad; mkdir -p $O
for v in cur alt; do
  if [ $v = alt ]; then export DRAM_HIP_LIB=$GRAFT_REPO_ROOT/$1; fi
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$v -o f -- python3 scripts/bench_conv.py --only wgrad --shapes "4,64,64,128;4,192,64,128" --iters 1 > $O/$v.out 2> $O/$v.err
  f=$(find $O/$v -name "*counter_collection.csv" | head -1)
  python3 - "$f" $v <<'PY'
import csv, sys, collections
per = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE" and "wgrad" in r["Kernel_Name"]:
        k = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][-40:])
        per[k] = per.get(k, 0) + float(r["Counter_Value"]) * 2048
print(sys.argv[2], [(k[1], round(v / 1e9, 2)) for k, v in per.items()])
PY
done

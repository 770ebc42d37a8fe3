"""Achieved HBM GB/s per kernel = average PMC bytes per launch (profiles/pmc_traffic.json) / average duration per launch
(profiles/<tag>_bench_kernel_stats.csv).  Both runs execute the same workload, so the two averages run over the same
mix of launches.
    python scripts/hbm_table.py profiles/r02_c_bench_kernel_stats.csv"""
import csv
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from pmc_aggregate import short_name  # noqa: E402

stats = sys.argv[1]
pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
rows = []
total = 0.0
for r in csv.DictReader(open(stats)):
    total += float(r["TotalDurationNs"])
for r in csv.DictReader(open(stats)):
    k = short_name(r["Name"])
    if k not in pmc:
        continue
    avg_s = float(r["AverageNs"]) * 1e-9
    rows.append((float(r["TotalDurationNs"]) / total, k, pmc[k]["total_bytes"] / avg_s / 1e9, pmc[k]["read_bytes"] / 1e9,
                 pmc[k]["write_bytes"] / 1e9, avg_s * 1e3))
rows.sort(reverse=True)
print("| kernel | share of kernel time | avg ms per launch | HBM GB read / written per launch | achieved GB/s | of 8 TB/s |")
print("|---|---|---|---|---|---|")
for share, k, gbs, r, w, ms in rows:
    print(f"| `{k}` | {share:.2%} | {ms:.2f} | {r:.2f} / {w:.2f} | {gbs:.0f} | {gbs / 8000:.1%} |")

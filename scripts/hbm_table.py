"""Achieved HBM GB/s per kernel = PMC bytes (profiles/pmc_traffic.json: one micro-batch) x micro-batches in the
kernel-stats run / total kernel time (profiles/<tag>_bench_kernel_stats.csv).
    python scripts/hbm_table.py profiles/r01_d_bench_kernel_stats.csv [micro_batches_in_stats_run=12]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from pmc_aggregate import short_name  # noqa: E402

stats = sys.argv[1]
nmicro = int(sys.argv[2]) if len(sys.argv) > 2 else 12
pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
rows = []
for r in csv.DictReader(open(stats)):
    k = short_name(r["Name"])
    if k not in pmc:
        continue
    t = float(r["TotalDurationNs"]) * 1e-9
    b = pmc[k]["total_bytes"] * pmc[k]["launches"] * nmicro
    rows.append((t, k, b / t / 1e9, pmc[k]["read_bytes"] / 1e9, pmc[k]["write_bytes"] / 1e9, t / nmicro * 1e3))
rows.sort(reverse=True)
print("| kernel | ms per micro-batch | HBM GB read / written per launch | achieved GB/s | of 8 TB/s |")
print("|---|---|---|---|---|")
for t, k, gbs, r, w, ms in rows:
    print(f"| `{k}` | {ms:.2f} | {r:.2f} / {w:.2f} | {gbs:.0f} | {gbs / 8000:.1%} |")

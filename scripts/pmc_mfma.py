"""Aggregate a rocprofv3 counter pass `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU
SQ_BUSY_CYCLES` (counters only, no trace flags) into per-kernel matrix-core figures:

    python scripts/pmc_mfma.py <counter_collection.csv> > profiles/rNN_pmc_mfma.json

per kernel (average over its dispatches):
  mfma_util        = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE / XCDs * SIMDs): the share of SIMD-cycles in which the matrix
                     pipe is busy -- rocprofv3's own derived `MfmaUtil`, with GRBM_GUI_ACTIVE (reported as the SUM over the 8 XCDs,
                     MI355X_MICROARCH.md "DVFS give-back") divided by 8;
  mfma_flops       = SQ_INSTS_VALU_MFMA_MOPS_F32 * 512 (rocprofv3's `MfmaFlopsF32`): the fp32 FLOPs the matrix cores executed,
                     to be compared with the count bench.py prices the kernel at (24 / 36 / 54 of the direct form's 54 multiply-adds);
  valu_per_mfma    = (SQ_INSTS_VALU - MFMA instructions) / MFMA instructions: vector instructions beside each matrix instruction;
  clock_ghz        = GRBM_GUI_ACTIVE / 8 / duration.
"""
import csv
import json
import re
import sys
from collections import defaultdict

XCDS, SIMDS = 8, 1024


def short_name(full):
    m = re.search(r"(?:dram::)?([A-Za-z_0-9]+)(<[^>(]*>)?\s*\(", full)
    return (m.group(1) + (m.group(2) or "")) if m else full.strip()


def main(path):
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))     # kernel -> dispatch -> counter -> value
    dur = defaultdict(dict)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = short_name(row["Kernel_Name"])
            per[k][row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
            dur[k][row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
    out = {"_method": __doc__.strip().splitlines()[0], "_unit": "averages per launch"}
    for k in sorted(per):
        if "_kernel" not in k or "at::" in k or k.startswith("void "):
            continue
        d = per[k]
        n = len(d)
        mf = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for v in d.values())
        if mf == 0:
            continue
        gui = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in d.values())
        mops = sum(v.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) for v in d.values())
        valu = sum(v.get("SQ_INSTS_VALU", 0.0) for v in d.values())
        t = sum(dur[k].values())
        flops = mops * 512.0
        # one v_mfma_f32_32x32x2_f32 = 4096 FLOP = 8 MOPS, one 16x16x4 = 2048 FLOP = 4 MOPS: instructions from busy cycles instead
        out[k] = {"launches": n, "avg_ms_under_pmc": 1e3 * t / n, "mfma_util": mf / (gui / XCDS * SIMDS),
                  "mfma_flops_per_launch": flops / n, "mfma_tflops_under_pmc": flops / t / 1e12,
                  "valu_insts_per_launch": valu / n, "clock_ghz": gui / XCDS / t / 1e9}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])

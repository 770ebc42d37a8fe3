#!/bin/bash
# SQ counters of the Winograd-(z,y) forward kernel variants on one layer shape (diagnostics)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_wzy; rm -rf $O; mkdir -p $O
SH="${1:-4,192,64,128}"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/a -o a -- python3 scripts/check_wzy.py --no-check --iters 2 --shapes "$SH" > $O/a.out 2> $O/a.err
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $O/b -o b -- python3 scripts/check_wzy.py --no-check --iters 2 --shapes "$SH" > $O/b.out 2> $O/b.err
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $O/c -o c -- python3 scripts/check_wzy.py --no-check --iters 2 --shapes "$SH" > $O/c.out 2> $O/c.err
python3 - <<'P'
import csv, glob, collections
for tag in "abc":
    fs = glob.glob(f"gpurun_out/pmc_wzy/{tag}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv3d_k3_fwd" not in k: continue
            k = k.split("(")[0][-60:] + "|grid" + r.get("Grid_Size", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        
    for k, d in agg.items():
        print(tag, k, {c: f"{v:.3g}" for c, v in d.items()})
P

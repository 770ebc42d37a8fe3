#!/bin/bash
# SQ counters of the conv kernels on one layer shape (diagnostics): where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_conv; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/a -o a -- python3 scripts/bench_conv.py --shapes "4,64,64,128" --iters 2 > $O/a.out 2> $O/a.err
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $O/b -o b -- python3 scripts/bench_conv.py --shapes "4,64,64,128" --iters 2 > $O/b.out 2> $O/b.err
ls $O/a $O/b

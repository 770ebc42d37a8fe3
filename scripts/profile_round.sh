#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats of the default bench, then the PMC passes
# (FETCH_SIZE, WRITE_SIZE, the matrix-core counters -- separate runs, counters only) over two steps of the same workload
# (default bench: the 64 chunks as one batch).
#   gpurun -- 'bash scripts/profile_round.sh r01_d'
# Results land in gpurun_out/<tag>/ ; copy the summaries into profiles/ (see scripts/pmc_aggregate.py).
set -eo pipefail
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG
mkdir -p "$O"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -o bench -- \
    python3 bench.py --no-cpu-baseline --no-att > "$O/bench_under_rocprof.json" 2> "$O/kt.err"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -o f -- \
    python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-att --no-kernel-timer > "$O/f.out" 2> "$O/f.err"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -o w -- \
    python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-att --no-kernel-timer > "$O/w.out" 2> "$O/w.err"
# matrix-core counters (their own pass, counters only): MFMA busy cycles, fp32 MFMA MOPS, the clock (scripts/pmc_mfma.py)
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_BUSY_CYCLES \
    --output-format csv -d "$O/mfma" -o m -- \
    python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-att --no-kernel-timer > "$O/m.out" 2> "$O/m.err"
find "$O" -name "*kernel_trace.csv" -delete     # per-dispatch trace: large, the stats CSV is what is kept
find "$O" -name "*.csv" | xargs ls -la

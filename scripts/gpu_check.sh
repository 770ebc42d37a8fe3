#!/bin/bash
# One GPU-box call: GPU test suite, smoke(), default bench, 2-rank rehearsal of `bench.py --gpus 2` over gloo, 1-rank
# rehearsal of the bench's RCCL path.
#   gpurun --timeout 1200 -- 'bash scripts/gpu_check.sh <tag>'
# A step that times out / is killed (rc >= 124) ends the call: no further GPU step is started after it.
TAG=${1:-chk}
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$TAG
mkdir -p "$O"
step() {   # step <seconds> <logfile> <cmd...>
    local secs=$1 log=$2; shift 2
    echo "== $* (limit ${secs}s)"
    timeout -k 10 "$secs" "$@" > "$log" 2>&1
    local rc=$?
    echo "   rc=$rc"; tail -n 4 "$log"
    if [ $rc -ge 124 ]; then echo "step killed: stopping"; exit $rc; fi
    return $rc
}
step 900 "$O/tests.log" python -m pytest tests -m gpu -q -x --durations=15
step 200 "$O/smoke.log" python __graft_entry__.py smoke
step 400 "$O/bench.json" python bench.py
step 300 "$O/bench_dp2_gloo.json" python bench.py --gpus 2 --backend gloo --chunks 4 --size 64 --micro 4 --no-cpu-baseline
# the bench's RCCL code path with the one rank a 1-GPU box allows (process group over backend nccl, warm-up all-reduce, barriers)
DRAM_BENCH_FORCE_DIST=1 step 300 "$O/bench_rccl1.json" python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29871 bench.py --gpus 1 --chunks 4 --size 64 --micro 4 --no-cpu-baseline
exit 0

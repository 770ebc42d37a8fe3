"""Data-parallel train step on the device: W ranks with one shard each must
reproduce one rank that runs the same shards as gradient-accumulated micro-batches -- the rank-count
independence check of SURVEY section 4 -- through forward, fused loss, backward, the bucketed gradient
all-reduce and the optimiser step of `DataParallelTrainer`.

The GPU box has ONE MI355X, so both ranks share cuda:0.  gloo (which stages device tensors through the host)
always works that way; RCCL (backend "nccl") refuses two ranks on one device, so that variant skips cleanly
here and is what the driver's multi-GPU run exercises.  `GroupNorm(1, C)` ("ln") keeps the statistics per
sample, so sharding the batch does not change the math."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
N_PER_RANK, SIZE, LR = 3, 16, 1e-2


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "bodyct-dram_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _model(norm):
    import models
    from dram_amd.configs import SLIM
    torch.manual_seed(11)
    m = models.DC3D(**SLIM, norm_method=norm)
    m.init(models.HeNorm(mode="fan_in"))
    return m


def _shard(rank):
    from dram_amd.train_step import synthetic_batch
    return synthetic_batch(N_PER_RANK, SIZE, 300 + rank, torch.device("cuda", 0))


def _worker(rank, world, port, backend, norm, out):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    try:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
            probe = torch.ones(4, device="cuda")
            dist.all_reduce(probe)                # RCCL builds its communicator lazily: force it now
            torch.cuda.synchronize()
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    except Exception as e:                        # two ranks on one device: RCCL says "Duplicate GPU detected"
        out[rank] = f"refused: {type(e).__name__}: {str(e)[:200]}"
        return
    from dram_amd.train_step import DataParallelTrainer
    m = _model(norm).cuda().train()
    tr = DataParallelTrainer(m, torch.optim.SGD(m.parameters(), lr=LR), bucket_mb=0.05)   # several buckets
    assert tr.world == world and len(tr.buckets) > 3
    reg, seg = tr.step(_shard(rank))
    torch.cuda.synchronize()
    out[rank] = {"params": {k: v.detach().cpu() for k, v in m.named_parameters()},
                 "reg": float(reg), "seg": float(seg), "overlapped": tr.overlapped_buckets, "buckets": len(tr.buckets)}
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_rank_step_equals_one_rank_micro_batched_step(backend):
    _setup_paths()
    from dram_amd.train_step import Batch, DataParallelTrainer
    world, norm = 2, "ln"
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = 29700 + (os.getpid() % 200) + (0 if backend == "gloo" else 250)
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, norm, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    hung = [p for p in procs if p.is_alive()]
    for p in hung:
        p.kill()
    res = dict(out)
    refused = [v for v in res.values() if isinstance(v, str)]
    if backend == "nccl" and (refused or hung or any(p.exitcode != 0 for p in procs)):
        pytest.skip("RCCL does not run two ranks on one device here "
                    f"({refused[0] if refused else 'init failed'}); the multi-GPU path is the driver's SCALE run")
    assert not hung and all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert set(res) == {0, 1}
    # the all-reduce of (nearly) every bucket started INSIDE backward, from the engine's gradient sink: only the bucket that
    # holds the first layer's parameters can complete with the last gradient of backward
    for r in range(world):
        assert res[r]["overlapped"] >= res[r]["buckets"] - 1 >= 3, (res[r]["overlapped"], res[r]["buckets"])

    # one rank, the two shards as two micro-batches of the same global batch
    m = _model(norm).cuda().train()
    sd0 = {k: v.detach().clone() for k, v in m.named_parameters()}
    tr = DataParallelTrainer(m, torch.optim.SGD(m.parameters(), lr=LR))
    b0, b1 = _shard(0), _shard(1)
    cat = lambda a, b: torch.cat([a, b], 0)
    both = Batch(cat(b0.images, b1.images), cat(b0.lobes, b1.lobes), cat(b0.lesions, b1.lesions), b0.ctss + b1.ctss,
                 {k: 1.0 / 6 for k in range(6)})
    reg, seg = tr.step(both, micro_batch=N_PER_RANK)
    ref = {k: v.detach().cpu() for k, v in m.named_parameters()}

    assert abs(res[0]["reg"] + res[1]["reg"] - float(reg)) <= 1e-5 * max(1.0, abs(float(reg)))
    assert abs(res[0]["seg"] + res[1]["seg"] - float(seg)) <= 1e-5 * max(1.0, abs(float(seg)))
    moved = 0
    for k, e in ref.items():
        step = (e - sd0[k].cpu()).abs().max().item()
        moved += step > 0.0
        for r in range(world):
            err = (res[r]["params"][k] - e).abs().max().item()
            # same kernels, same shards: only the order of the two-term gradient sum differs
            assert err <= 1e-4 * step + 1e-9, (backend, r, k, err, step)
        assert torch.equal(res[0]["params"][k], res[1]["params"][k]), k   # replicas stay bit-identical
    assert moved >= 0.9 * len(ref)                                # the step really moved the parameters


def _rccl_single_worker(port, out):
    """One rank over backend "nccl" (= RCCL): the calls of the multi-GPU path -- communicator bound to the device,
    asynchronous SUM all-reduce of the flat gradient buckets, barrier -- on the one GPU of the test box."""
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from dram_amd.train_step import DataParallelTrainer
    m = _model("ln").cuda().train()
    tr = DataParallelTrainer(m, torch.optim.SGD(m.parameters(), lr=LR), bucket_mb=0.05)
    tr.world = 2                                  # take the all-reduce branch (a sum over the one rank: the identity)
    reg, seg = tr.step(_shard(0), global_batch=N_PER_RANK)
    dist.barrier()
    torch.cuda.synchronize()
    out["params"] = {k: v.detach().cpu() for k, v in m.named_parameters()}
    out["buckets"] = len(tr.buckets)
    out["backend"] = dist.get_backend()
    dist.destroy_process_group()


def test_rccl_all_reduce_path_single_rank():
    """The RCCL leg of DataParallelTrainer on hardware: the step through `dist.all_reduce` over backend nccl must leave
    exactly the weights of the same step without a process group."""
    _setup_paths()
    from dram_amd.train_step import DataParallelTrainer
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    p = ctx.Process(target=_rccl_single_worker, args=(29950 + os.getpid() % 40, out))
    p.start()
    p.join(240)
    if p.is_alive():
        p.kill()
        pytest.fail("RCCL single-rank step hung")
    assert p.exitcode == 0, p.exitcode
    res = dict(out)
    assert res["backend"] == "nccl" and res["buckets"] > 3
    m = _model("ln").cuda().train()
    tr = DataParallelTrainer(m, torch.optim.SGD(m.parameters(), lr=LR))
    tr.step(_shard(0))
    for k, v in m.named_parameters():
        assert torch.equal(v.detach().cpu(), res["params"][k]), k


def test_bench_two_ranks_over_gloo_as_a_child_process():
    """`python bench.py --gpus 2` as the driver starts it (it launches its two ranks itself: torch.distributed.run children), over
    gloo because the test box has one GPU: the multi-rank code path of the benchmark -- process group, per-rank data, gradient
    all-reduce overlapped with backward, MAX over ranks of the time, the JSON line with `roofline` and a `cpu_baseline` field
    (null + reason for N > 1) -- must run and report finite losses for both ranks' shards."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--chunks", "2",
                        "--size", "32", "--micro", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-att"],
                       env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["scaling"] == "weak"
    assert line["dist"]["ranks_seen"] == 2 and line["dist"]["backend"] == "gloo"
    assert line["loss"]["finite"] and len(line["loss"]["reg"]) == 1
    assert line["value"] > 0 and abs(line["value"] - 2 * 2 * 32 ** 3 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    assert line["roofline"] is not None and line["roofline"]["bound"] == "mfma" and 0 < line["roofline"]["frac"] <= 1.0
    assert line["cpu_baseline"] is None and "N = 1" in line["cpu_baseline_note"]
    ar = line["allreduce"]
    assert ar["overlapped_with_backward"] and ar["buckets_started_inside_backward"] == ar["buckets"] >= 1, ar


def test_bench_ranks_agree_on_the_micro_batch_when_one_runs_out_of_memory():
    """One rank's "does it fit" probe fails (test hook DRAM_BENCH_FAKE_OOM_RANK), the other's does not: both must halve the
    micro-batch together and finish -- the probe issues no collective of its own, the outcome is agreed by one all-reduce per
    attempt (an out-of-memory error inside a real step would leave the ranks with different numbers of collectives issued)."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DRAM_BENCH_FAKE_OOM_RANK="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--chunks", "32",
                        "--size", "16", "--micro", "32", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-att"],
                       env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "retrying with micro-batch 16" in r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["config"]["micro_batch"] == 16 and line["dist"]["ranks_seen"] == 2 and line["loss"]["finite"]

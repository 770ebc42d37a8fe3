#!/usr/bin/env python
"""Benchmark of the DRAM DC3D hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With N > 1 and no WORLD_SIZE in the environment, `python bench.py --gpus N` starts the N ranks itself
(fresh `torch.distributed.run` children, before this process touches the GPU) and exits with their status.

A "step" is one optimisation step (forward + loss + backward + gradient all-reduce + Adam) of the
reference's benchmark model `st_dram_ref.MODEL` (DC3D, BatchNorm, fp32; `checkpoint_layers` honoured in
'stats' mode by default: no recomputation, the flagged blocks' double BatchNorm running-stat update is
reproduced) on this rank's batch of synthetic lobe chunks: 64 chunks of 1x128^3 per GPU (the shape
BASELINE.json's metric names), as ONE batch by default (BatchNorm statistics over all 64 chunks, like the
reference): un-fused, 64x128^3 of saved activations does not fit 288 GB; the fused engine
(dram_amd/engine.py) keeps the raw conv outputs only and peaks at 251 GB.  `--micro M` runs gradient-accumulated
micro-batches of M chunks instead (M = 16: 123 GB, about +1 % throughput, statistics per 16 chunks).  Inputs are
resident in HBM before the timed region.  Weak scaling: the per-GPU batch is fixed.

Rank 0 prints one JSON line: metric/value (whole-job voxels/s), roofline of the dominant kernel
(3x3x3 conv forward / backward-data as Winograd F(2x2,3x3)-over-(z,y) implicit GEMMs on fp32 MFMA, timed live with HIP
events: `frac` prices the MFMA FLOPs the kernel really issues, 4/9 of the direct form's), cpu_baseline (the oracle's
torch-CPU DC3D timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "bodyct-dram_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4 dense peak
PEAK_HBM_GBS = 8000.0
PMC_MICRO = 64        # micro-batch of the rocprofv3 --pmc passes behind profiles/pmc_traffic.json (per-launch bytes scale with it)


def wz_factor(kernel_name):
    """Executed / algorithmic MFMA FLOPs of a conv kernel: 2/3 for the Winograd F(2,3)-along-z kernels (36 of 54
    multiply-adds), 4/9 for the F(2x2,3x3)-over-(z,y) kernels (16 products per channel pair and x tap for 4 outputs
    instead of 36: 24 of 54)."""
    if "_wzy" in kernel_name:             # conv3d_k3_fwd_wzy_kernel, conv3d_k3_fwd_wzy16_kernel, conv3d_k3_wgrad_wzy_kernel<..>
        return 4.0 / 9.0
    return 2.0 / 3.0 if "_wz_" in kernel_name else 1.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chunks", type=int, default=64, help="chunks per GPU per step (metric: 64)")
    ap.add_argument("--size", type=int, default=128, help="chunk edge (metric: 128)")
    ap.add_argument("--micro", type=int, default=64, help="micro-batch (chunks): 64 = the whole per-GPU batch in one pass "
                    "(BatchNorm statistics over all 64 chunks, like the reference); smaller = gradient accumulation")
    ap.add_argument("--norm", default="bn", help="norm_method of the model (reference default: bn)")
    ap.add_argument("--checkpoint-mode", default="stats", choices=["stats", "recompute"],
                    help="how checkpoint_layers flags are honoured (models.DC3D.checkpoint_mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=90.0, help="budget for the CPU baseline sample (1 warm-up + 3 reps "
                    "at 2x64^3 and at 1x128^3: ~60 s on 16 threads)")
    ap.add_argument("--no-kernel-timer", action="store_true", help="skip the per-kernel HIP-event pass")
    ap.add_argument("--no-att", action="store_true", help="skip the side lines (reference-shape steps, inference step, "
                    "measured ceilings) reported beside the headline")
    ap.add_argument("--no-overlap", action="store_true", help="reduce the gradient buckets after backward instead of inside it "
                    "(A/B of the overlap; N > 1 only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; "
                    "gloo only to rehearse the multi-rank path with several ranks on one GPU)")
    return ap.parse_args()


def ensure_built():
    """`make` every time (a no-op when libdram_hip.so is newer than csrc/): a stale library must not be what is timed.
    Without hipcc (a box that only received the prebuilt library) the existing library is used as it is."""
    import shutil
    import subprocess
    lib = os.path.join(PKG, "libdram_hip.so")
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if os.path.exists(hipcc) and shutil.which("make"):
        rc = subprocess.call(["make", "-C", os.path.join(PKG, "csrc"), "-j", str(min(8, os.cpu_count() or 1)), "ARCH=gfx950",
                              f"HIPCC={hipcc}"], stdout=subprocess.DEVNULL if os.path.exists(lib) else None)
        if rc != 0 and not os.path.exists(lib):
            sys.exit("bench.py: building libdram_hip.so failed")
    elif not os.path.exists(lib):
        sys.exit(f"bench.py: {lib} is missing and there is no hipcc to build it (no CPU fallback)")


def host_cores():
    """Cores this process may really use: affinity mask, cgroup CPU quota, and the GPU box's
    per-GPU CPU share (16) -- os.cpu_count() alone reports the whole host (256 threads) and
    oversubscribing it makes the CPU baseline ~10x slower."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("DRAM_CPU_THREADS", "16"))))


def cpu_baseline(budget_s):
    """The oracle's torch-CPU DC3D (same config, same kind of step: forward + backward, BatchNorm in training mode)
    on this box's host cores, at the two reduced batches BASELINE.md section 3 names: 2 x 64^3 and 1 x 128^3, each as
    1 warm-up + 3 timed repetitions, median reported.  `value` is the 1 x 128^3 figure (the
    metric's chunk shape); the 2 x 64^3 one rides along.  Bounded: should a size overrun its share of `budget_s` it stops
    after the repetition in flight and `sample` says how many were timed."""
    import torch
    from oracle import dram_oracle as O
    threads = host_cores()
    torch.set_num_threads(threads)
    cfg = O.ST_DRAM_REF_MODEL
    params, buffers = O.init_params(cfg, "bn", seed=0)
    for p in params.values():
        p.requires_grad_(True)

    def measure(n, s, max_reps, budget):
        x = torch.rand(n, 1, s, s, s, generator=torch.Generator().manual_seed(1))
        gout = torch.randn(n, 1, s, s, s, generator=torch.Generator().manual_seed(2))

        def step():
            for p in params.values():
                p.grad = None
            out = O.dc3d_forward(cfg, params, buffers, x, training=True, norm_method="bn")
            (out * gout).sum().backward()

        t_start = time.perf_counter()
        step()                                                  # warm-up (threads, primitives, allocator)
        times = []
        while len(times) < max_reps:
            t0 = time.perf_counter()
            step()
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget:
                break
        times.sort()
        return n * s ** 3 / times[len(times) // 2], len(times)

    t0 = time.perf_counter()
    v64, r64 = measure(2, 64, 3, 0.25 * budget_s)
    left = max(1.0, budget_s - (time.perf_counter() - t0))
    v128, r128 = measure(1, 128, 3, left)
    return {"value": v128, "unit": "voxels/s", "cores": threads, "kind": "port",
            "value_2x64": v64, "reps": {"1x128": r128, "2x64": r64}, "warmups": 1,
            "sample": f"oracle torch-CPU DC3D(st_dram_ref, bn) fwd+bwd on {threads} threads: 1x1x128^3, median of {r128} "
                      f"rep(s) after 1 warm-up (value); 2x1x64^3, median of {r64} rep(s) after 1 warm-up (value_2x64)"}


def measured_ceilings(dev):
    """both peaks calibrated on THIS box, in this process, ~1 s: a 16-byte-per-lane copy of 1 GiB
    (read + written bytes / time) and a register-only v_mfma_f32_32x32x2_f32 loop on every CU (csrc/calibrate.hip), each
    bracketed by HIP events on the launch stream; best of 5 after a warm-up launch."""
    import ctypes
    import torch
    from dram_amd import _lib
    st = torch.cuda.current_stream().cuda_stream
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=dev).zero_()
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    sink = torch.empty(cus * 512, dtype=torch.float32, device=dev)
    flops = ctypes.c_double(0.0)

    def best(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        out = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            out.append(e0.elapsed_time(e1))
        return min(out)
    ms_copy = best(lambda: _lib.call("dram_calibrate_hbm_copy", src.data_ptr(), dst.data_ptr(), n, st))
    ms_mfma = best(lambda: _lib.call("dram_calibrate_mfma_f32", sink.data_ptr(), cus, 20000, ctypes.addressof(flops), st))
    del src, dst, sink
    torch.cuda.empty_cache()
    return {"hbm_copy_tbs": 2.0 * n / (ms_copy * 1e-3) / 1e12, "mfma_f32_tflops": flops.value / (ms_mfma * 1e-3) / 1e12,
            "how": f"1 GiB 16-byte-per-lane copy (read + write bytes / time) and a register-only v_mfma_f32_32x32x2_f32 loop "
                   f"(8 accumulator tiles, 2 waves per SIMD, {cus} CUs): best of 5 launches each, HIP events on the launch stream"}


def model_train_step(dev, kind, n=10, size=80, steps=3):
    """One train step of the reference's own training shape -- TRAIN_BATCH_SIZE 10 chunks of RESAMPLE_SIZE 80^3
    (st_dram_ref.py:37,41 / st_dram_ref_att.py:40-45) -- through the same engine and trainer as the headline:
    kind 'dc3d' = DC3D(st_dram_ref), 'att' = DC3DATGeneric(st_dram_ref_att) (models.py:415-597, what process_pipeline.py loads)."""
    import torch
    import models
    from dram_amd.configs import ST_DRAM_REF_ATT_MODEL, ST_DRAM_REF_MODEL
    from dram_amd.train_step import DataParallelTrainer, synthetic_batch
    torch.manual_seed(0)
    m = models.DC3DATGeneric(**ST_DRAM_REF_ATT_MODEL) if kind == "att" else models.DC3D(**ST_DRAM_REF_MODEL)
    m.init(models.HeNorm(mode="fan_in"))
    m = m.to(dev).train()
    tr = DataParallelTrainer(m, torch.optim.Adam(m.parameters(), lr=1e-4))
    b = synthetic_batch(n, size, 100, dev)
    tr.step(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        reg, seg = tr.step(b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    name = "DC3DATGeneric st_dram_ref_att (attention grid 64^3, 18 neighbours)" if kind == "att" else "DC3D st_dram_ref"
    return {"model": name, "chunks": n, "chunk": [size] * 3,
            "ms_per_step": 1e3 * dt, "voxels_per_s": n * size ** 3 / dt, "fused_engine": bool(m.fused),
            "reg": float(reg), "seg": float(seg)}


def inference_step(dev, shape=(300, 512, 512), reps=3):
    """BASELINE config 5: whole-scan per-lobe inference (dram_amd/inference.py: crop -> 80^3 -> model -> paste -> Otsu -> brightness
    gate) of a synthetic 300 x 512 x 512 CT with five lobes, FULL-WIDTH DC3DATGeneric(st_dram_ref_att) in eval mode -- the model
    process_pipeline.py:11 loads; scan, label map and vessel mask resident in HBM before the timed region."""
    import numpy as np
    import torch
    import models
    from dram_amd.configs import ST_DRAM_REF_ATT_MODEL
    from dram_amd.inference import LobeInference, synthetic_ct
    torch.manual_seed(0)
    m = models.DC3DATGeneric(**ST_DRAM_REF_ATT_MODEL)
    m.init(models.HeNorm(mode="fan_in"))
    m = m.to(dev).eval()
    scan, lobe, spacing = synthetic_ct(shape, (1.0, 0.7, 0.7), seed=7, n_lesions=20)
    scan_d, lobe_d = torch.as_tensor(scan).to(dev), torch.as_tensor(lobe).to(dev)
    vessel_d = torch.zeros_like(lobe_d)
    inf = LobeInference(m, resample_size=80)
    res = inf.run(scan_d, lobe_d, spacing, vessel=vessel_d)            # warm-up (first launches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        res = inf.run(scan_d, lobe_d, spacing, vessel=vessel_d)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return {"model": "DC3DATGeneric st_dram_ref_att, eval mode, full width", "scan": list(shape), "lobes": len(res["chunks"]),
            "resample": [80] * 3, "ms_per_scan": 1e3 * dt, "scan_voxels_per_s": float(np.prod(shape)) / dt,
            "lesion_ratio": res["lesion_ratio"], "mask_voxels": int(res["mask"].sum()), "mask_post_voxels": int(res["mask_post"].sum())}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (one per GPU,
    `python -m torch.distributed.run`), before this process has imported torch or touched the GPU, and return
    their exit status.  Rank 0 of the children prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ or int(os.environ.get("LOCAL_RANK", "0")) == 0:
        ensure_built()          # (under a launcher only local rank 0 runs make; the others meet it at the first barrier)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks "
                 f"(or run `python bench.py --gpus {args.gpus}`, which starts them itself)")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # DRAM_BENCH_FORCE_DIST=1: create the process group even for one rank (rehearsal of the RCCL code path on a 1-GPU box:
    # `python -m torch.distributed.run --nproc-per-node 1 ... bench.py`)
    use_dist = world > 1 or (os.environ.get("DRAM_BENCH_FORCE_DIST") and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:                                  # rehearsal: ranks may share a GPU
            torch.cuda.set_device(local % torch.cuda.device_count())
            dist.init_process_group(backend=args.backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    if use_dist:
        # RCCL sets up its channels and staging buffers (its own hipMalloc, outside torch's allocator) at the first
        # collective of a given size class: do that now, while the device is empty -- the step below fills 251 of 288 GB
        warm = torch.zeros(16 << 20, dtype=torch.float32, device=dev)          # 64 MB, the size of the gradient buckets
        dist.all_reduce(warm)
        dist.barrier()
        torch.cuda.synchronize()
        del warm
    dist_info = None
    if use_dist:
        # What the process group really spans, gathered over the ranks (the line would otherwise prove n_gpus only as
        # WORLD_SIZE): one record per rank = (device uuid / PCI bus id, device name, free bytes beside RCCL's buffers).
        prop = torch.cuda.get_device_properties(dev)
        ident = str(getattr(prop, "uuid", "")) or f"{getattr(prop, 'pci_bus_id', '?')}:{getattr(prop, 'pci_device_id', '?')}"
        if hasattr(prop, "pci_bus_id"):
            ident += f"/bus{prop.pci_bus_id:02x}.{getattr(prop, 'pci_device_id', 0):02x}.{getattr(prop, 'pci_domain_id', 0)}"
        free_b, _tot = torch.cuda.mem_get_info(dev)
        mine = {"rank": rank, "local_rank": local, "device": ident, "name": prop.name, "free_gb_after_rccl_warmup": free_b / 2 ** 30}
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, mine)
        distinct = len({g["device"] for g in gathered})
        dist_info = {"backend": dist.get_backend(), "ranks_seen": dist.get_world_size(), "distinct_devices": distinct,
                     "devices": [g["device"] for g in gathered],
                     "rccl_free_gb": min(g["free_gb_after_rccl_warmup"] for g in gathered)}
        if dist.get_backend() == "nccl" and distinct != dist.get_world_size():
            if rank == 0:
                print(f"bench.py: {dist.get_world_size()} RCCL ranks on {distinct} distinct device(s): {dist_info['devices']}",
                      file=sys.stderr, flush=True)
            dist.destroy_process_group()
            sys.exit(3)

    import models
    from dram_amd import functional as HF
    from dram_amd.train_step import DataParallelTrainer, synthetic_batch
    from dram_amd.configs import ST_DRAM_REF_MODEL

    torch.manual_seed(0)                                   # same initial replica on every rank
    model = models.DC3D(**ST_DRAM_REF_MODEL, norm_method=args.norm)
    model.init(models.HeNorm(mode="fan_in"))
    model.checkpoint_mode = args.checkpoint_mode
    model = model.to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    trainer = DataParallelTrainer(model, opt, overlap=not args.no_overlap)
    batch = synthetic_batch(args.chunks, args.size, 100 + rank, dev)   # resident in HBM before timing
    vox_per_rank = args.chunks * args.size ** 3

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up.  The 64-chunk batch goes through as ONE batch (the fused engine keeps raw conv outputs only and runs the
    # widest stage in slices: 251 of 288 GB at its peak); should that not fit on this device (other tenants, a
    # smaller part), fall back to gradient-accumulated micro-batches of half the size and say so in the output.
    micro = min(args.micro, args.chunks)
    # The "does it fit" probe: forward + loss + backward of one micro-batch with NO collective and no optimiser step
    # (DataParallelTrainer.probe), then ONE all-reduce of the outcome on every rank -- so that a rank that runs out of memory
    # and its peers that do not still issue the same collectives -- and every rank halves together.
    while micro > 16 or use_dist:
        fits = True
        try:
            if os.environ.get("DRAM_BENCH_FAKE_OOM_RANK") == str(rank) and micro == min(args.micro, args.chunks):
                raise torch.OutOfMemoryError("DRAM_BENCH_FAKE_OOM_RANK (test hook: this rank's first probe fails)")
            trainer.probe(batch, micro)
        except torch.OutOfMemoryError:
            fits = False
            torch.cuda.empty_cache()
        if use_dist:
            ft = torch.tensor([1 if fits else 0], device=dev)
            dist.all_reduce(ft, op=dist.ReduceOp.MIN)
            fits = bool(int(ft.item()))
        if fits:
            break
        if micro <= 8:
            raise RuntimeError(f"bench.py: a micro-batch of {micro} chunks does not fit on every rank's device")
        micro //= 2
        if rank == 0:
            print(f"bench.py: out of memory, retrying with micro-batch {micro}", file=sys.stderr, flush=True)
    for i in range(args.warmup):
        trainer.step(batch, micro)
    args.micro = micro
    # per-kernel HIP events (one pair per conv launch, on the launch stream) ride along in the timed region
    use_timer = not args.no_kernel_timer
    sync()
    if use_timer:
        HF.TIMER = HF.KernelTimer()
    t0 = time.perf_counter()
    losses = []
    for _ in range(args.steps):
        losses.append(trainer.step(batch, args.micro))        # (device scalars: no host synchronisation here)
    sync()
    elapsed = time.perf_counter() - t0
    timer, HF.TIMER = HF.TIMER, None
    trainer_overlap, n_buckets, overlapped = trainer.overlap, len(trainer.buckets), trainer.overlapped_buckets
    # ---- outside the timed region: what the timed steps computed (a wrong-but-fast step must not look like a result)
    loss_hist = [(float(r), float(sg)) for r, sg in losses]
    with torch.no_grad():
        psum = sum(float(p.double().abs().sum()) for p in model.parameters())
        gfinite = all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
        pfinite = all(bool(torch.isfinite(p).all()) for p in model.parameters())
    finite = bool(gfinite and pfinite and all(x == x and abs(x) != float("inf") for pair in loss_hist for x in pair))
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)
    value = world * vox_per_rank * args.steps / elapsed

    # ---- roofline of the dominant kernel, from the HIP events recorded inside the timed region
    roofline = None
    kernels = {}
    summ = {}
    if rank == 0 and use_timer:
        summ = timer.summary()
        for k, d in summ.items():
            ex = d["flops"] * wz_factor(k)
            kernels[k] = {"launches": d["launches"], "avg_ms": d["ms"] / d["launches"], "total_ms": d["ms"],
                          "executed_tflops": ex / (d["ms"] * 1e-3) / 1e12,
                          "algorithmic_equiv_tflops": d["flops"] / (d["ms"] * 1e-3) / 1e12}
        if summ:
            dom = max(summ, key=lambda k: summ[k]["ms"])
            d = summ[dom]
            # SURVEY section 8(d) counts the ALGORITHMIC work of a 3x3x3 conv as the direct 27-tap form,
            # 54*Cin*Cout FLOP per voxel.  The *_wz_* kernels run Winograd F(2,3) along z: they EXECUTE 2/3 of those
            # multiply-adds on the fp32 matrix cores (exact fp32); the *_wzy_* kernels F(2x2,3x3) over (z,y): 4/9.  The roofline fraction prices what was really
            # issued against the MFMA peak (always <= 1); the direct-equivalent rate is reported beside it.
            alg = d["flops"] / (d["ms"] * 1e-3) / 1e12
            ex = alg * wz_factor(dom)
            roofline = {"bound": "mfma", "kernel": dom, "achieved": ex, "peak": PEAK_FP32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": ex / PEAK_FP32_MFMA_TFLOPS,
                        "executed_tflops": ex, "algorithmic_equiv_tflops": alg,
                        "executed_flops_per_launch": d["flops"] * wz_factor(dom) / d["launches"],
                        "algorithmic_flops_per_launch": d["flops"] / d["launches"],
                        "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                        "launches": d["launches"], "avg_launch_ms": d["ms"] / d["launches"],
                        "traffic": None, "traffic_bytes_per_launch": None,
                        "executed_over_algorithmic": wz_factor(dom),
                        "note": "achieved/frac = MFMA FLOPs actually issued (Winograd: F(2,3) along z executes 36, "
                                "F(2x2,3x3) over (z,y) 24 of the direct form's 54 multiply-adds per channel pair and "
                                "voxel) / fp32-MFMA peak; algorithmic_equiv_tflops = direct-conv FLOPs of SURVEY 8(d) / time"}
            mf = os.path.join(ROOT, "profiles", "pmc_mfma.json")       # per-kernel matrix-core counters (scripts/pmc_mfma.py)
            if os.path.exists(mf) and (args.size, args.chunks, args.micro) == (128, 64, PMC_MICRO):
                try:
                    t = json.load(open(mf)).get(dom)
                    if t:
                        roofline["pmc_mfma_util"] = t["mfma_util"]
                        roofline["pmc_mfma_flops_per_launch"] = t["mfma_flops_per_launch"]
                        roofline["pmc_note"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 over "
                                                "the same workload (profiles/): share of SIMD-cycles with the matrix pipe busy at the "
                                                "clock the chip held, and the fp32 MFMA FLOPs it counted per launch")
                except Exception:
                    pass
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # per-launch HBM bytes from rocprofv3 --pmc passes
            # (collected on the default workload, 64 x 128^3 as one batch: only that workload has the same launches)
            if os.path.exists(pmc) and (args.size, args.chunks, args.micro) == (128, 64, PMC_MICRO):
                try:
                    t = json.load(open(pmc)).get(dom)
                    if t:
                        roofline["traffic"] = roofline["traffic_bytes_per_launch"] = t["total_bytes"]
                        roofline["traffic_read_bytes_per_launch"] = t["read_bytes"]
                        roofline["traffic_write_bytes_per_launch"] = t["write_bytes"]
                except Exception:
                    pass
    if use_dist:
        dist.barrier()

    cpu = None
    cpu_note = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_seconds)
    elif world > 1:
        cpu_note = "the CPU baseline is timed on rank 0 at N = 1 only (bench contract): see the n_gpus = 1 line"
    elif args.no_cpu_baseline:
        cpu_note = "skipped (--no-cpu-baseline)"

    # ---- outside the timed region and not part of `value` (N = 1 only): the ceilings measured on this box, the reference's own
    # training shape (10 x 80^3) with DC3D(st_dram_ref) and with the attention model process_pipeline.py loads, and config 5
    att = ref_shape = infer = ceilings = None
    ck_mode = model.checkpoint_mode
    peak_alloc = torch.cuda.max_memory_allocated() / 2 ** 30
    peak_res = torch.cuda.max_memory_reserved() / 2 ** 30
    if rank == 0 and world == 1 and not args.no_att:
        del trainer, opt, model, batch, losses
        torch.cuda.empty_cache()
        ceilings = measured_ceilings(dev)
        ref_shape = model_train_step(dev, "dc3d")
        att = model_train_step(dev, "att")
        torch.cuda.empty_cache()
        infer = inference_step(dev)
    if roofline is not None and ceilings is not None:
        roofline["frac_of_measured"] = roofline["achieved"] / ceilings["mfma_f32_tflops"]
        roofline["peak_measured"] = ceilings["mfma_f32_tflops"]

    if rank == 0:
        flops_per_voxel = 5415936.0     # fwd+bwd algorithmic FLOPs per input voxel
        tot_alg = sum(d["flops"] for d in summ.values())
        tot_ex = sum(d["flops"] * wz_factor(k) for k, d in summ.items())
        exec_ratio = tot_ex / tot_alg if tot_alg else (2.0 / 3.0)
        line = {
            "metric": "voxels/sec fwd+bwd (DC3D train step) on 64x128^3 CT chunks per GPU",
            "batchnorm_statistics_over_chunks": args.micro,
            "value": value, "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"DC3D st_dram_ref (norm={args.norm}, checkpoint_layers as shipped, honoured as "
                                   f"'{ck_mode}') "
                                   f"fwd+loss+bwd+Adam, {args.chunks}x1x{args.size}^3 chunks per GPU, "
                                   f"micro-batch {args.micro}, fp32",
                       "chunks_per_gpu": args.chunks, "chunk": [args.size] * 3, "micro_batch": args.micro,
                       "parallelism": f"dp{world}"},
            "per_gpu_voxels_per_s": value / world,
            "peak_hbm_allocated_gb": peak_alloc,
            "peak_hbm_reserved_gb": peak_res,      # what torch's allocator holds at the peak of the timed steps
            "hbm_total_gb": torch.cuda.get_device_properties(dev).total_memory / 2 ** 30,
            # whole step against the fp32 peak: executed = what the kernels issue (conv FLOPs x 2/3 resp. 4/9 where the
            # Winograd kernels run: every layer but the first), algorithmic = the direct-conv count of SURVEY 8(d)
            "network_executed_frac_of_fp32_peak": value / world * flops_per_voxel * exec_ratio / (PEAK_FP32_MFMA_TFLOPS * 1e12),
            "network_algorithmic_equiv_tflops": value / world * flops_per_voxel / 1e12,
            # the losses of the timed steps (rank 0's shard; reg summed over its chunks, seg weighted by its share of the
            # global batch), a checksum of the parameters after the last step, and whether everything is finite
            "loss": {"reg": [l[0] for l in loss_hist], "seg": [l[1] for l in loss_hist],
                     "param_abs_sum_after": psum, "finite": finite},
            "dist": dist_info,
            "allreduce": None if world == 1 else {"overlapped_with_backward": bool(trainer_overlap), "buckets": n_buckets,
                                                  "buckets_started_inside_backward": overlapped},
            "ceilings_measured": ceilings,
            "reference_shape_step": ref_shape, "attention_model_step": att, "inference_step": infer,
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_note": cpu_note, "kernels": kernels,
            "kernels_note": "a kernel's time is that of its C-ABI call on the launch stream (HIP events around the call): a "
                            "backward-weights entry includes its fixed-order slab reduction (0.07 % of it)",
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()
    if not finite:
        sys.exit("bench.py: non-finite loss, gradient or parameter in the timed steps")


if __name__ == "__main__":
    main()

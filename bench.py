#!/usr/bin/env python3
"""Benchmark of the un-projection hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--train-step]

N > 1 without a launcher: this process starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child
BEFORE anything touches the GPU and relays rank 0's JSON line (the driver's own torchrun launch is used as is).

A step = one pass of the hot path (layout pass + fused un-projection kernel, through the C ABI) over one batch
of synthetic input that is already resident in HBM.  Workload at N = 1: BASELINE.json's metric configuration --
64^3 voxel grid, 256 channels, 4 views (H36M-like ring), 96x96 feature maps, batch 32, fp32, softmax aggregate,
forward.  N > 1 is weak scaling: every rank owns its own batch of 32 (samples are independent, no data-path
collective: SURVEY.md 8e); value = all ranks' voxel*views / max-over-ranks time.

Rank 0 prints ONE JSON line.  Besides the driver's fields it carries
  roofline      algorithmic HBM bytes of the path (features + coords + proj + output, once each)
                / the dominant kernel's average duration, measured with HIP events on the launch stream inside
                the timed region; peak = 8 TB/s (MI355X_MICROARCH.md)
  cpu_baseline  the reference's CPU algorithm (per-(b,v) F.grid_sample loop, oracle/reference_loop_torch.py)
                timed on this box's host cores on a bounded sample (rank 0, N = 1 only)
  auto_call     the same step through the drop-in Python function unprojection(..., variant='auto') (one C-ABI call: gate
                kernel + layout pass + fused kernel, workspace from the caching allocator), for comparison with `value`
  rccl          N > 1: the path's only collective, the flat fp32 all-reduce of process_feature's gradient, in microseconds
--train-step replaces the step by one training step of the aggregator (BASELINE configs[4]): VolumeGenerator forward, scalar
loss, backward (1x1 conv + un-projection backward) and that all-reduce; shapes come from --grid/--channels/--batch.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from multiviewhmr_amd import _capi, multiview  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU")
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--feat", type=int, default=96)
    ap.add_argument("--dtype", choices=("f32", "f16"), default="f32")
    ap.add_argument("--method", default="softmax")
    ap.add_argument("--variant", default="auto", choices=("auto", "gather", "brick"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-backward", action="store_true", help="skip the (untimed-for-`value`) backward measurement")
    ap.add_argument("--cpu-sample-batch", type=int, default=8, help="samples of the workload timed on the host (about 15 s)")
    ap.add_argument("--train-step", action="store_true", help="time one aggregator training step (fwd + bwd + grad all-reduce)")
    ap.add_argument("--in-channels", type=int, default=None, help="--train-step: input channels of the 1x1 conv (default: --channels)")
    ap.add_argument("--soak-ms", type=float, default=1000.0,
                    help="untimed steps run for about this long before the warm-up: the chip reaches its sustained clock (this path is power-limited, "
                         "DESIGN.md 5.1) and an activity sampler beside the run sees it busy")
    ap.add_argument("--rccl-selftest", action="store_true",
                    help="one rank, one GPU: initialise the 'nccl' (= RCCL) backend with world size 1 and run the path's collective on the device "
                         "(proves init_process_group, the IPC-mode pin and the flat-buffer all-reduce load and run on this pool; no scaling number)")
    return ap.parse_args()


def rccl_selftest(a):
    """The RCCL leg of this file on ONE GPU: the 8-GPU driver run must not be that code's first execution.
    World size 1 measures no scaling; it proves that backend 'nccl' initialises on this pool (HSA_ENABLE_IPC_MODE_LEGACY=0), that the
    collectives of the N > 1 path (barrier, the MAX reduce of the elapsed time, the flat fp32 gradient all-reduce) execute on the
    device, and that sharding.allreduce_aggregator_grads leaves a one-rank module's gradients as they were."""
    import socket
    import torch.distributed as dist
    from multiviewhmr_amd import sharding
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    t0 = time.perf_counter()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    t_init = time.perf_counter() - t0
    dist.barrier()
    t = torch.tensor([1.25], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                       # the reduce bench.py applies to the elapsed time
    assert float(t.item()) == 1.25
    C = a.channels
    rccl = time_grad_allreduce(C * C + C, dev, True, 1)
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(C, C, 1).to(dev)                        # process_feature (models/aggregation.py:108-110)
    conv(torch.randn(2, C, 8, 8, device=dev)).square().mean().backward()
    before = [p.grad.clone() for p in conv.parameters()]
    n = sharding.allreduce_aggregator_grads(conv)
    torch.cuda.synchronize()
    same = all(torch.equal(b, p.grad) for b, p in zip(before, conv.parameters()))
    assert n == C * C + C and same, "one-rank all-reduce changed the gradients"
    print(json.dumps({"rccl_selftest": {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "init_s": round(t_init, 2),
                                        "allreduce": rccl, "flat_grad_elements": n, "grads_unchanged": same,
                                        "note": "one rank: no scaling number; proves the RCCL leg loads and runs on this pool"}}), flush=True)
    dist.destroy_process_group()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run the N ranks as a child torchrun job (this process has not
    touched the GPU and never will) and relay what rank 0 prints."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # ROCm's legacy IPC handles need a driver feature this pool's hosts lack (dmabuf IPC only): with the legacy mode on, RCCL's
    # intra-node transport setup fails in hipIpcGetMemHandle ("invalid argument").  The image exports this already; it is pinned
    # here so that a child started from a scrubbed environment still initialises RCCL.  Nothing on the data path depends on it.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    for l in (lines[-1:] if lines else res.stdout.splitlines()):
        print(l, flush=True)
    return res.returncode


# ------------------------------------------------------------------------------------------- synthetic inputs
def ring_projections(batch, views, feat_hw, image_hw=(384, 384), seed=0):
    """H36M-like ring: V cameras at azimuth 2*pi*k/V + 0.3, radius 4.5-5.5 m, height 1.5 m, looking at the origin,
    z-up, 1000x1000 sensor, f = 1145 px; the reference's own bookkeeping (crop -> resize -> resize, Q3) gives P."""
    rng = np.random.default_rng(seed)
    P = np.empty((batch, views, 3, 4), np.float32)
    for b in range(batch):
        for v in range(views):
            az = 2 * np.pi * v / views + 0.3
            eye = np.array([np.cos(az), np.sin(az), 0.0]) * rng.uniform(4500.0, 5500.0) + np.array([0, 0, 1500.0])
            fwd = -eye / np.linalg.norm(eye)
            right = np.cross(fwd, [0.0, 0.0, 1.0]); right /= np.linalg.norm(right)
            R = np.stack([right, np.cross(fwd, right), fwd])
            cam = multiview.Camera(R, -R @ eye, [[1145.0, 0, 512.0], [0, 1145.0, 512.0], [0, 0, 1.0]])
            cam.update_after_crop((200, 200, 824, 824))
            cam.update_after_resize((624, 624), (image_hw[1], image_hw[0]))
            cam.update_after_resize(image_hw, feat_hw)
            P[b, v] = cam.projection
    return P


def cuboid_volume(batch, S, side=2500.0, seed=0):
    """aggregation.py:140-187 with theta = 0 (eval): origin-centred cuboid; the pivot is irrelevant at theta = 0."""
    g = np.stack(np.meshgrid(np.arange(S), np.arange(S), np.arange(S), indexing="ij"), -1).astype(np.float32)
    coords = (np.float32(-side / 2) + np.float32(side / (S - 1)) * g).astype(np.float32)
    return np.broadcast_to(coords, (batch,) + coords.shape).copy()


def frustum_stats(P, coords, H, W):
    pts = coords.reshape(-1, 3).astype(np.float64)
    hom = np.concatenate([pts, np.ones((len(pts), 1))], 1)
    inside, invalid = [], []
    for Pv in P:
        r = hom @ Pv.astype(np.float64).T
        z = r[:, 2]
        with np.errstate(all="ignore"):
            ix = (r[:, 0] / z) / H * (W - 1)
            iy = (r[:, 1] / z) / W * (H - 1)
        inv = z <= 0
        invalid.append(inv.mean())
        inside.append(((ix > -1) & (ix < W) & (iy > -1) & (iy < H) & ~inv).mean())
    return float(np.mean(inside)), float(np.mean(invalid))


def time_grad_allreduce(n_floats, dev, use_rccl, world, iters=20):
    """The path's only collective (sharding.allreduce_aggregator_grads' flat fp32 buffer): microseconds per all-reduce."""
    import torch.distributed as dist
    flat = torch.ones(n_floats, dtype=torch.float32, device=dev if use_rccl else "cpu")
    for _ in range(5):
        dist.all_reduce(flat)
    if use_rccl:
        torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        dist.all_reduce(flat)
    if use_rccl:
        torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / iters * 1e6
    return {"ranks": world, "allreduce_us": round(us, 1), "bytes": n_floats * 4, "backend": "rccl" if use_rccl else "gloo (ranks share a GPU)"}


def train_step_bench(a, rank, world, dev, use_rccl):
    """BASELINE configs[4]: one training step of the aggregator on this rank's batch shard -- VolumeGenerator forward, a
    scalar loss, backward (1x1 conv + un-projection backward) and the flat gradient all-reduce."""
    from multiviewhmr_amd import aggregation, sharding
    if world > 1:
        import torch.distributed as dist
    B, S, C, V, HW, IMG = a.batch, a.grid, a.channels, a.views, a.feat, 384
    Cin = a.in_channels or C
    rng = np.random.default_rng(rank)
    cams = [[None] * B for _ in range(V)]
    for v in range(V):
        az = 2 * np.pi * v / V + 0.3
        for b in range(B):
            eye = np.array([np.cos(az), np.sin(az), 0.0]) * rng.uniform(4500.0, 5500.0) + np.array([0, 0, 1500.0])
            fwd = -eye / np.linalg.norm(eye)
            right = np.cross(fwd, [0.0, 0.0, 1.0]); right /= np.linalg.norm(right)
            R = np.stack([right, np.cross(fwd, right), fwd])
            cam = multiview.Camera(R, -R @ eye, [[1145.0, 0, 512.0], [0, 1145.0, 512.0], [0, 0, 1.0]])
            cam.update_after_crop((200, 200, 824, 824))
            cam.update_after_resize((624, 624), (IMG, IMG))
            cams[v][b] = cam
    batch = {"images": np.zeros((B, V, IMG, IMG, 3), np.uint8), "cameras": cams,
             "keypoints_3d": [rng.normal(0, 100, (17, 3)).astype(np.float32) for _ in range(B)]}
    torch.manual_seed(0)                                            # same initial weights on every rank
    gen = aggregation.VolumeGenerator(volume_size=S, input_channels=Cin, output_channels=C, device=dev).train()
    np.random.seed(rank)
    torch.manual_seed(100 + rank)
    feats = torch.randn(B, V, Cin, HW, HW, device=dev, requires_grad=True)
    proj_org = torch.from_numpy(np.stack([[cams[v][b].projection for v in range(V)] for b in range(B)]).astype(np.float32)).to(dev)
    n_grad = sum(p.numel() for p in gen.parameters())
    grad_vol = torch.randn(B, C, S, S, S, device=dev) * (1.0 / (B * C * S ** 3))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    parts = np.zeros(3)

    def step(record=False):
        gen.zero_grad(set_to_none=True)
        feats.grad = None
        t = [time.perf_counter()]
        vol = gen(feats, proj_org, batch)
        if record: torch.cuda.synchronize(); t.append(time.perf_counter())
        vol.backward(grad_vol)    # the consumer (the 3-D regressor) hands back a dense gradient; a scalar loss here would time torch's
                                  # 8.6 GB reduce + broadcast copy (6.7 ms at this size) instead of the aggregator
        if record: torch.cuda.synchronize(); t.append(time.perf_counter())
        if world > 1:
            sharding.allreduce_aggregator_grads(gen)
        if record:
            torch.cuda.synchronize(); t.append(time.perf_counter())
            parts[:] += np.diff(t)

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if use_rccl else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    for _ in range(2):
        step(record=True)                                          # untimed: where the step goes (synchronising)
    ms = elapsed / a.steps * 1e3
    value = B * S ** 3 * V * world / (elapsed / a.steps) / 1e6
    result = {
        "metric": "Mvoxel*views/s, one aggregator train step (fwd + bwd + grad all-reduce)",
        "value": round(value, 1), "unit": "Mvoxel*views/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "VolumeGenerator train step: %d^3 grid, %d views, 1x1 conv %d->%d ch, %dx%d maps, batch %d per GPU, softmax"
                               % (S, V, Cin, C, HW, HW, B), "global_batch": B * world,
                   "parallelism": "batch-sharded x%d; flat all-reduce of %d fp32 conv gradients" % (world, n_grad)},
        "phases_ms": {"forward": round(parts[0] / 2 * 1e3, 2), "backward": round(parts[1] / 2 * 1e3, 2),
                      "grad_allreduce": round(parts[2] / 2 * 1e3, 3)},
    }
    if world > 1:
        result["rccl"] = time_grad_allreduce(n_grad, dev, use_rccl, world)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------- main
def main():
    a = parse()
    if a.rccl_selftest:
        return rccl_selftest(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a.gpus))          # before any GPU call: the parent only waits for its child
    if world != a.gpus and world > 1:
        a.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    n_dev = torch.cuda.device_count()
    # one rank per GPU; on a box with fewer GPUs than ranks (rehearsals on a 1-GPU box) ranks share devices and the
    # control-plane collectives fall back to gloo -- the data path has no collective either way
    dev = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(dev)
    use_rccl = n_dev >= world
    if world > 1:
        import torch.distributed as dist
        if use_rccl:
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    if a.train_step:
        return train_step_bench(a, rank, world, dev, use_rccl)

    B, S, C, V, HW = a.batch, a.grid, a.channels, a.views, a.feat
    N = S ** 3
    tdt = torch.float32 if a.dtype == "f32" else torch.float16
    esz = 4 if a.dtype == "f32" else 2

    torch.manual_seed(rank)
    feats = torch.randn(B, V, C, HW, HW, device=dev, dtype=torch.float32).to(tdt)
    P_np = ring_projections(B, V, (HW, HW), seed=rank)
    coords_np = cuboid_volume(1, S)
    proj = torch.from_numpy(P_np).to(dev)
    coords = torch.from_numpy(coords_np).to(dev).expand(B, S, S, S, 3).contiguous()
    out = torch.empty(B, C, S, S, S, device=dev, dtype=tdt)

    L = _capi.lib()
    desc = _capi.Desc()
    desc.abi_version = _capi.ABI_VERSION
    desc.batch, desc.views, desc.channels, desc.feat_h, desc.feat_w = B, V, C, HW, HW
    desc.vol_x = desc.vol_y = desc.vol_z = S
    desc.method = _capi.AGG[a.method]
    desc.feat_dtype = desc.out_dtype = _capi.F32 if a.dtype == "f32" else _capi.F16
    desc.variant = _capi.VARIANT[a.variant]
    # planar (reference-contract) input: the step runs the layout pass, then the fused kernel on its result
    desc.feat_layout = _capi.LAYOUT_BVCHW
    vp = ctypes.c_void_p
    stream = vp(torch.cuda.current_stream(dev).cuda_stream)
    # Set-up (outside the timed region): ask which variant the geometry gate selects for these cameras and this voxel
    # pitch.  The step below runs the layout pass and the kernel as two calls so that HIP events can bracket each, and
    # explicitly converted layouts are not gated -- so the variant is fixed here, as a drop-in caller's first step would.
    variant = L.mvhmr_unproject_query_variant(ctypes.byref(desc), vp(proj.data_ptr()), vp(coords.data_ptr()), stream)
    assert variant > 0, L.mvhmr_last_error().decode()
    desc.variant = variant
    # the step = layout pass (planar reference-contract input -> the layout the kernel reads) + fused kernel;
    # they are launched through separate C-ABI calls so that HIP events can bracket each of them
    lay = L.mvhmr_preferred_layout(ctypes.byref(desc))
    conv = torch.empty(L.mvhmr_feature_layout_bytes(ctypes.byref(desc), lay), dtype=torch.uint8, device=dev)
    d_k = _capi.Desc.from_buffer_copy(desc)
    d_k.feat_layout = lay

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * max(a.steps, 1))]

    def step(i=None):
        """one pass of the hot path; with i given, HIP events bracket the two kernels (same stream as the launches)"""
        if i is not None: ev[3 * i].record()
        _capi.check(L.mvhmr_convert_features(ctypes.byref(desc), vp(feats.data_ptr()), lay, vp(conv.data_ptr()), stream))
        if i is not None: ev[3 * i + 1].record()
        _capi.check(L.mvhmr_unproject_forward(ctypes.byref(d_k), vp(conv.data_ptr()), vp(proj.data_ptr()), vp(coords.data_ptr()),
                                              vp(out.data_ptr()), vp(0), 0, stream))
        if i is not None: ev[3 * i + 2].record()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    t_soak = time.perf_counter()
    while (time.perf_counter() - t_soak) * 1e3 < a.soak_ms:          # untimed; the queue is drained every 16 steps so the loop tracks wall time
        for _ in range(16):
            step()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if use_rccl else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_step = elapsed / a.steps * 1e3
    k_layout = float(np.mean([ev[3 * i].elapsed_time(ev[3 * i + 1]) for i in range(a.steps)]))
    k_all = [ev[3 * i + 1].elapsed_time(ev[3 * i + 2]) for i in range(a.steps)]
    k_main = float(np.mean(k_all))

    voxel_views = B * N * V * world
    value = voxel_views / (elapsed / a.steps) / 1e6
    # algorithmic bytes: every tensor of the path touched once; intermediates count zero
    alg_bytes = B * V * C * HW * HW * esz + B * N * 3 * 4 + B * V * 12 * 4 + B * C * N * esz
    achieved = alg_bytes / (k_main * 1e-3) / 1e9

    result = {
        "metric": "Mvoxel*views/s (64^3 grid, 256ch, 4 views) fwd; max-abs vs ref",
        "value": round(value, 1), "unit": "Mvoxel*views/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": "unprojection fwd %d^3 grid x %d ch x %d views, %dx%d maps, batch %d per GPU, %s, %s aggregate"
                               % (S, C, V, HW, HW, B, a.dtype, a.method),
                   "global_batch": B * world, "parallelism": "batch-sharded x%d, no data-path collective" % world,
                   "kernel_variant": "brick" if variant == 2 else "gather", "input_layout": "BVCHW (reference contract)"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernel": (L.mvhmr_unproject_forward_kernel_name(ctypes.byref(d_k)) or b"?").decode(), "kernel_ms": round(k_main, 4),
                     "kernel_ms_min": round(float(np.min(k_all)), 4), "layout_pass_ms": round(k_layout, 4), "algorithmic_bytes": alg_bytes,
                     "step_frac": round(alg_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
    }

    # The path's only collective, timed on every rank (N > 1): process_feature's flat gradient buffer (C*C + C fp32)
    if world > 1:
        result["rccl"] = time_grad_allreduce(C * C + C, dev, use_rccl, world)

    # The same step as a reference user calls it: unprojection(features, proj, coords) with AUTO -- one C-ABI call that
    # runs the device-side geometry gate, the layout pass and the fused kernel, workspace from the caching allocator.
    if rank == 0 and world == 1:
        from multiviewhmr_amd import aggregation
        ea = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        with torch.no_grad():
            for it in range(13):
                if it == 3: ea[0].record()
                o2 = aggregation.unprojection(feats, proj, coords, aggregation_method=a.method, variant="auto")
            ea[1].record()
        torch.cuda.synchronize()
        ms_auto = ea[0].elapsed_time(ea[1]) / 10
        result["auto_call"] = {"ms": round(ms_auto, 4), "value": round(B * N * V / (ms_auto * 1e-3) / 1e6, 1), "unit": "Mvoxel*views/s",
                               "what": "aggregation.unprojection(features, proj, coords, variant='auto'): gate + layout pass + kernel in one call"}
        del o2

    # Backward (gradient w.r.t. the features), reported beside the headline: SURVEY.md 8(d) quotes fwd and fwd+bwd
    # separately.  Outside the timed region of `value`; grad_out = the forward output (same shape, realistic values).
    if rank == 0 and world == 1 and not a.no_backward:
        wsb = L.mvhmr_unproject_backward_workspace_bytes(ctypes.byref(desc))
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
        gfeat = torch.empty_like(feats)
        eb = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        n_b = 5
        for it in range(n_b + 1):
            if it == 1: eb[0].record()
            _capi.check(L.mvhmr_unproject_backward(ctypes.byref(desc), vp(out.data_ptr()), vp(feats.data_ptr()), vp(proj.data_ptr()),
                                                   vp(coords.data_ptr()), vp(gfeat.data_ptr()), vp(ws.data_ptr()), wsb, stream))
        eb[1].record()
        torch.cuda.synchronize()
        ms_b = eb[0].elapsed_time(eb[1]) / n_b
        # grad_out + features (re-read) + coords + proj + grad_features
        bwd_bytes = B * C * N * esz + B * V * C * HW * HW * esz + B * N * 12 + B * V * 48 + B * V * C * HW * HW * 4
        result["backward"] = {"ms": round(ms_b, 3), "value": round(B * N * V / (ms_b * 1e-3) / 1e6, 1), "unit": "Mvoxel*views/s",
                              "algorithmic_bytes": bwd_bytes, "frac": round(bwd_bytes / (ms_b * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "includes": "layout pass of the features + gradient clear + kernel (one C-ABI call)",
                              "fwd_plus_bwd_ms": round(ms_step + ms_b, 3)}
        del ws, gfeat

    if rank == 0:
        traffic = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(traffic):
            try:
                tj = json.load(open(traffic))
                md5_path = os.path.join(ROOT, "multiviewhmr_amd", "lib", "sources.md5")
                md5_now = open(md5_path).read().strip() if os.path.exists(md5_path) else None
                if tj.get("workload_key") != "%d-%d-%d-%d-%d-%s" % (S, C, V, HW, B, a.dtype):
                    pass
                elif tj.get("sources_md5") != md5_now or tj.get("kernel_name") != result["roofline"]["kernel"]:
                    # counters taken from other kernel sources than the library this run loaded: not replayed
                    result["roofline"]["traffic_source"] = "not replayed: %s was taken at sources %s (commit %s), this library is %s" % (
                        os.path.relpath(traffic, ROOT), tj.get("sources_md5"), tj.get("commit"), md5_now)
                else:
                    result["roofline"]["traffic"] = tj.get("hbm_bytes_per_launch")
                    result["roofline"]["traffic_source"] = "replayed from %s, taken at commit %s = these kernel sources (separate rocprofv3 --pmc passes, not this run)" % (
                        os.path.relpath(traffic, ROOT), tj.get("commit"))
                if result["roofline"].get("traffic") is not None and "backward" in result and "backward" in tj:
                        result["backward"]["traffic"] = tj["backward"].get("hbm_bytes_per_launch")
                        result["backward"]["traffic_source"] = result["roofline"]["traffic_source"]
            except Exception:
                pass
        inside, invalid = frustum_stats(P_np[0], coords_np[0], HW, HW)
        result["config"]["in_frustum_fraction"] = round(inside, 4)
        result["config"]["invalid_fraction"] = round(invalid, 4)
        if not a.no_check:
            from oracle import cport     # checker only: one sample against the CPU oracle
            ref = cport.forward(feats[:1].float().cpu().numpy(), P_np[:1], np.ascontiguousarray(coords_np[:1]), a.method)
            err = float(np.abs(out[:1].float().cpu().numpy() - ref).max())
            result["max_abs_vs_ref"] = err
        if world == 1 and not a.no_cpu_baseline:
            from oracle.reference_loop_torch import unprojection_cpu_loop
            nb = min(a.cpu_sample_batch, B)
            fc, pc = feats[:nb].float().cpu(), proj[:nb].cpu()
            cc = torch.from_numpy(np.ascontiguousarray(coords_np)).expand(nb, S, S, S, 3).contiguous()
            t1 = time.perf_counter()
            unprojection_cpu_loop(fc, pc, cc, a.method)
            dt = time.perf_counter() - t1
            result["cpu_baseline"] = {"value": round(nb * N * V / dt / 1e6, 4), "unit": "Mvoxel*views/s",
                                      "cores": torch.get_num_threads(), "kind": "port",
                                      "sample": "%d of %d samples of the same workload (samples are independent: the full batch is this rate, "
                                                "not re-timed), per-(b,v) F.grid_sample loop (oracle/reference_loop_torch.py), %.1f s, host has %d cpus"
                                                % (nb, B, dt, os.cpu_count())}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

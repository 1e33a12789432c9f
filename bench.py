#!/usr/bin/env python3
"""bench.py -- Mrays/s of the fused trace+plot hot path on the built-in scene.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one fused TraceUnit::render + PlotUnit::plot launch over `--batches-per-step` (default 256) batches
of 524,288 camera paths (trace_unit.rs:67) of the built-in demo scene (app.rs:166-363), followed,
every `--gather-every` steps, by the GatherUnit step (Kahan accumulate + clear; with N > 1 the XYZ
plot buffers are first sum-reduced to rank 0 over RCCL).  Every rank renders the full frame with its
own RNG stream (stream = rank) -- samples shard, nothing else is exchanged -- so per-GPU work is
fixed as N grows ("weak").  A ray = one Scene::intersect call (one path segment, scene.rs:39),
counted on the device.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 1024 * 512  # trace_unit.rs:67
# Algorithmic flops per ray = the reference's own reject-path arithmetic per primitive test
# (SURVEY 8d): sphere 19, paraboloid 38, plane / circle / half-space 14.
FLOPS_SPHERE, FLOPS_PARABOLOID, FLOPS_PLANE = 19, 38, 14
PEAK_FP32_VECTOR_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (vector)"
# HBM-side bytes per traced path of rl_trace_kernel from the committed PMC passes of this same command
# (profiles/r01m_pmc_summary.txt: (2*FETCH_SIZE + WRITE_SIZE) KB per 134,217,728-path launch); the
# counters cannot be read from inside this process, so `roofline.traffic` scales that measurement.
PROFILED_TRAFFIC_BYTES_PER_PATH = (2 * 482.4 + 5.308e6) * 1024 / 134217728

CONFIGS = {
    # name: (scene, param, width, height)
    "demo-1080p": ("demo", 0, 1920, 1080),      # BASELINE.json metric config (configs[3] per GPU)
    "demo-720p": ("demo", 0, 1280, 720),        # configs[1]
    "glass-720p": ("glass", 0, 1280, 720),      # configs[2]
    "replicated-1080p": ("demo", 158, 1920, 1080),  # configs[4]
    # ablation scenes (not BASELINE configs): prefixes of the demo scene's object list
    "ablate-noprisms": ("demo[:317]", 0, 1920, 1080),
    "ablate-fixed7": ("demo[:7]", 0, 1920, 1080),
    "ablate-seeds": ("demo[:207]", 0, 1920, 1080),
    "ablate-allgrey": ("demo[grey]", 0, 1920, 1080),  # every reflective material -> DiffuseGrey(0.8)
}


def flops_per_ray(objs):
    import numpy as np
    kinds = np.bincount(objs["surface_kind"], minlength=5)
    return int(FLOPS_SPHERE * kinds[0] + FLOPS_PARABOLOID * kinds[3] + FLOPS_PLANE * (kinds[1] + kinds[2] + 8 * kinds[4]))


def usable_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(objs, cam, width, height, seconds_target=15.0):
    """Times the CPU oracle (C++ restatement of the Rust reference; the Rust crate cannot be built
    here: no rustc/cargo) on this box's host cores over a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    threads = usable_cores()
    scene = O.Scene(objs.view(O.OBJECT_DTYPE), O.RlCameraDesc.from_buffer_copy(bytes(cam)))
    import ctypes as C
    segs = C.c_uint64(0)
    # calibrate on a small slice, then size the sample for ~seconds_target
    n0 = 20000 * threads
    dt0 = O.lib().oracle_render_mt(scene.h, width, height, 1, 0, 0, n0, None, C.byref(segs), threads)
    n = int(max(n0, min(n0 * seconds_target / max(dt0, 1e-3), 64 * BATCH)))
    dt = O.lib().oracle_render_mt(scene.h, width, height, 1, 0, n0, n, None, C.byref(segs), threads)
    segs1 = C.c_uint64(0)                     # the 1-thread number beside it (~3 s)
    n1 = max(20000, int(n / threads * 3.0 / max(dt, 1e-3)))
    dt1 = O.lib().oracle_render_mt(scene.h, width, height, 1, 0, n0 + n, n1, None, C.byref(segs1), 1)
    return {"value": segs.value / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "one_thread": {"value": segs1.value / dt1 / 1e6, "unit": "Mrays/s", "sample": "%d paths, %.1f s" % (n1, dt1)},
            "sample": "%d camera paths (%d rays) of the same scene/resolution, seed 1, %d threads, %.1f s"
                      % (n, segs.value, threads, dt),
            "mpaths_per_s": n / dt / 1e6, "batches_per_s": n / dt / BATCH}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="demo-1080p", choices=sorted(CONFIGS))
    ap.add_argument("--batches-per-step", type=int, default=256)
    ap.add_argument("--gather-every", type=int, default=2)
    ap.add_argument("--fetch", default="lds", choices=["lds", "global"])
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo stages the XYZ reduce through host memory and lets several "
                         "ranks share one GPU -- only for exercising the N > 1 code path on a 1-GPU box")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))

    import numpy as np
    import torch  # first: our library must share torch's HIP runtime (same SONAME) to share device pointers
    import torch.distributed as dist
    import robigo_luculenta_amd as R

    if R.device_count() < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU implementation")
    device = local_rank % R.device_count() if args.dist_backend == "gloo" else local_rank
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    on_device = args.dist_backend == "nccl"

    scene_name, param, W, H = CONFIGS[args.config]
    which = R.SCENE_DEMO if scene_name.startswith("demo") else R.SCENE_GLASS_STRESS
    objs, cam = R.builtin_scene_desc(which, param)
    if "[grey]" in scene_name:
        objs = objs.copy()
        refl = objs["material_kind"] != 0
        objs["material_kind"][refl] = 1
        objs["m"][refl] = (0.8, 0, 0)
    if "[:" in scene_name:
        objs = objs[: int(scene_name.split("[:")[1].rstrip("]"))].copy()
    scene = R.Scene(objs, cam, device=device)
    trace = R.TraceUnit(rank, W, H, n_photons=64, device=device)  # fused mode does not use mapped_photons
    trace.set_fetch(R.FETCH_LDS if args.fetch == "lds" else R.FETCH_GLOBAL)
    xyz = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")  # PlotUnit.tristimulus_buffer, reducible by RCCL
    plot = R.PlotUnit(rank, W, H, device=device, external_xyz=xyz.data_ptr())
    gather = R.GatherUnit(W, H, device=device)
    paths_per_step = args.batches_per_step * BATCH

    next_path = [0]

    def step(i):
        trace.render_fused(scene, plot, paths_per_step, seed=args.seed, stream=rank, first_path_index=next_path[0])
        next_path[0] += paths_per_step
        if (i + 1) % args.gather_every == 0:
            trace.sync()
            if world > 1 and on_device:
                dist.reduce(xyz, dst=0, op=dist.ReduceOp.SUM)  # GatherUnit-time exchange over xGMI
            elif world > 1:
                host = xyz.cpu()
                dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                if rank == 0:
                    xyz.copy_(host)
            if rank == 0:
                gather.accumulate(plot)   # Kahan + clear (gather_unit.rs:49-64, app.rs:147)
            else:
                plot.clear()

    def fence():
        trace.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    if world > 1:
        # Create the communicator and run the reduce once outside the timed region even when the warm-up
        # steps did not reach a gather (RCCL builds its rings lazily on the first collective).
        if on_device:
            dist.reduce(torch.zeros_like(xyz), dst=0, op=dist.ReduceOp.SUM)
        else:
            dist.reduce(torch.zeros(xyz.numel(), dtype=torch.float32), dst=0, op=dist.ReduceOp.SUM)
    fence()
    p0, s0, ms0 = trace.stats()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    t1 = time.perf_counter()
    p1, s1, ms1 = trace.stats()
    elapsed = t1 - t0
    rays, paths, kernel_ms = s1 - s0, p1 - p0, ms1 - ms0

    if world > 1:
        t = torch.tensor([elapsed, float(rays), float(paths), kernel_ms], dtype=torch.float64,
                         device="cuda" if on_device else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        total_rays, total_paths = float(t[1]), float(t[2])
    else:
        total_rays, total_paths = float(rays), float(paths)

    if rank == 0:
        f_seg = flops_per_ray(objs)
        # dominant kernel = rl_trace_kernel; its launches are timed with HIP events on the unit's own stream
        launch_ms = kernel_ms / args.steps
        achieved = (rays / args.steps) * f_seg / (launch_ms * 1e-3) / 1e12
        out = {
            "metric": "Mrays/sec on built-in scene at %dx%d" % (W, H),
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "built-in %s scene (%d objects), %dx%d, fused trace+plot, %d batches of 524288 paths per step "
                                   "per GPU, gather every %d steps, RNG stream = rank, primitives in %s"
                                   % (scene_name if param == 0 else "%s(seeds=%d)" % (scene_name, param), len(objs), W, H,
                                      args.batches_per_step, args.gather_every, "LDS" if args.fetch == "lds" else "global/scalar cache"),
                       "config": args.config, "paths_per_step_per_gpu": paths_per_step, "seed": args.seed},
            "mpaths_per_s": total_paths / elapsed / 1e6,
            "batches_per_s": total_paths / elapsed / BATCH,
            "segments_per_path": total_rays / max(total_paths, 1.0),
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VECTOR_TFLOPS,
                         "frac_of_unpacked_fp32_peak": achieved / (PEAK_FP32_VECTOR_TFLOPS / 2),  # 78.6 TFLOP/s, SURVEY 8(d)
                         "traffic": PROFILED_TRAFFIC_BYTES_PER_PATH * paths_per_step,
                         "hbm": {"achieved": PROFILED_TRAFFIC_BYTES_PER_PATH * paths_per_step / (launch_ms * 1e-3) / 1e9,
                                 "peak": 8000.0, "unit": "GB/s",
                                 "frac": PROFILED_TRAFFIC_BYTES_PER_PATH * paths_per_step / (launch_ms * 1e-3) / 8e12},
                         "traffic_note": "bytes per launch, scaled from the rocprofv3 FETCH_SIZE/WRITE_SIZE passes in profiles/ "
                                         "(f32 atomics count as 32-byte writes); algorithmic: 48 B per contributing path",
                         "kernel": "rl_trace_kernel", "kernel_ms_per_launch": launch_ms,
                         "algorithmic_flops_per_ray": f_seg,
                         "valu_busy_profiled": 0.95, "active_lanes_profiled": 0.71,
                         "note": "VALU-issue bound (no dense contraction -> no MFMA); HBM traffic is the XYZ splat only; "
                                 "valu_busy / active_lanes from the PMC passes in profiles/r01m_pmc_summary.txt"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(objs, cam, W, H)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

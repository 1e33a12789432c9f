#!/usr/bin/env python3
"""bench.py -- Mrays/s of the fused trace+plot hot path on the built-in scene.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: bench.py starts its own ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = `--launches-per-step` (8) fused TraceUnit::render + PlotUnit::plot launches of `--batches-per-launch`
(256) batches of 524,288 camera paths each (trace_unit.rs:67) on the built-in demo scene (app.rs:166-363), then the
GatherUnit step: with N > 1 the ranks' XYZ plot buffers are summed onto rank 0 by the library's own RCCL exchange
(rl_plot_unit_reduce: one ncclReduce over xGMI), rank 0 Kahan-accumulates and every rank clears
(gather_unit.rs:49-64, app.rs:147).  Every rank renders the full frame with its own RNG stream (stream = rank) --
samples shard, nothing else is exchanged -- so per-GPU work is fixed as N grows ("weak").  A ray = one
Scene::intersect call (one path segment, scene.rs:39), counted on the device.  Rank 0 prints ONE JSON line.

torch is not used for device work at all: N = 1 never imports it, and N > 1 uses torch.distributed (gloo) only for
the control plane (communicator id, barriers, max/sum of timings; robigo_luculenta_amd/distributed.py).
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 1024 * 512  # trace_unit.rs:67
# Algorithmic flops per ray = the reference's own reject-path arithmetic per primitive test
# (SURVEY 8d): sphere 19, paraboloid 38, plane / circle / half-space 14.
FLOPS_SPHERE, FLOPS_PARABOLOID, FLOPS_PLANE = 19, 38, 14
PEAK_FP32_VECTOR_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (vector)"
# Wave64 VALU issue interval of a plain mul/add/sub stream (the op mix the reference arithmetic allows) at the trace
# kernel's occupancy of 4 waves per SIMD, measured in shader cycles: profiles/r02_valu_microbench.txt ("mul/add/sub
# mix", w/SIMD = 4, wall c/i).  MI355X_MICROARCH.md's nominal figure is 2.
MEASURED_STREAM_CYCLES_4_WAVES = 2.7

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
# The other BASELINE configs, timed briefly in the default N = 1 run and reported under config.others.
OTHERS = (("demo-720p", "lds"), ("glass-720p", "lds"), ("replicated-1080p", "lds"), ("replicated-1080p", "global"))


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


def cpu_baseline(objs, cam, width, height, seconds_target=30.0):
    """Times the CPU oracle (C++ restatement of the Rust reference, built -O3 without fast-math; the Rust crate
    cannot be built here: no rustc/cargo) on this box's host cores over a bounded sample of the same workload:
    >= 30 s on every usable core, then ~3 s on one thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    threads = usable_cores()
    scene = O.Scene(objs.view(O.OBJECT_DTYPE), O.RlCameraDesc.from_buffer_copy(bytes(cam)))
    import ctypes as C
    segs = C.c_uint64(0)
    # calibrate on a small slice, then size the sample for ~seconds_target
    n0 = 20000 * threads
    dt0 = O.lib().oracle_render_mt(scene.h, width, height, 1, 0, 0, n0, None, C.byref(segs), threads)
    n = int(max(n0, min(n0 * seconds_target * 1.05 / max(dt0, 1e-3), 256 * BATCH)))
    dt = O.lib().oracle_render_mt(scene.h, width, height, 1, 0, n0, n, None, C.byref(segs), threads)
    segs1 = C.c_uint64(0)
    n1 = max(20000, int(n / threads * 3.0 / max(dt, 1e-3)))
    dt1 = O.lib().oracle_render_mt(scene.h, width, height, 1, 0, n0 + n, n1, None, C.byref(segs1), 1)
    return {"value": segs.value / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "one_thread": {"value": segs1.value / dt1 / 1e6, "unit": "Mrays/s", "sample": "%d paths, %.1f s" % (n1, dt1)},
            "sample": "%d camera paths (%d rays) of the same scene/resolution, seed 1, %d threads, %.1f s; oracle built -O3"
                      % (n, segs.value, threads, dt),
            "mpaths_per_s": n / dt / 1e6, "batches_per_s": n / dt / BATCH}


def scene_of(R, config):
    scene_name, param, W, H = CONFIGS[config]
    which = R.SCENE_DEMO if scene_name.startswith("demo") else R.SCENE_GLASS_STRESS
    objs, cam = R.builtin_scene_desc(which, param)
    if "[grey]" in scene_name:
        objs = objs.copy()
        refl = objs["material_kind"] != 0
        objs["material_kind"][refl] = 1
        objs["m"][refl] = (0.8, 0, 0)
    if "[:" in scene_name:
        objs = objs[: int(scene_name.split("[:")[1].rstrip("]"))].copy()
    label = scene_name if param == 0 else "%s(seeds=%d)" % (scene_name, param)
    return objs, cam, W, H, label


def executed_from_profile(R, config, fetch, rays_per_launch, launch_ms):
    """The counter-derived half of the roofline, from the newest committed profiles/*_pmc.json that was measured
    on THIS build of the library and this workload (rl_build_id: a hash of the device code's sources).  The
    counters cannot be read from inside the process, so they come from the rocprofv3 --pmc passes of this same
    command (tools/profile_round.sh); a profile of another build is refused rather than quoted."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("config") != config or d.get("fetch") != fetch:
            continue
        if d.get("build_id") == R.build_id() or best is None or best[1].get("build_id") != R.build_id():
            best = (path, d)   # the newest profile of this build, else the newest of any build (reported as stale)
    if best is None:
        return {"stale": True, "reason": "no profiles/*_pmc.json for config %s / fetch %s" % (config, fetch)}
    path, d = best
    if d.get("build_id") != R.build_id():
        return {"stale": True, "reason": "%s was measured on build %s, this library is build %s: counters not quoted"
                                         % (os.path.relpath(path, ROOT), d.get("build_id"), R.build_id())}
    c = d["counters"]
    segs64 = d["rays_per_launch"] / 64.0
    simds = 1024.0
    cyc = (c["GRBM_GUI_ACTIVE"] / 8.0) / (c["SQ_INSTS_VALU"] / simds)
    lanes = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
    out = {
        "profile": os.path.relpath(path, ROOT), "build_id": d["build_id"],
        "valu_insts_per_64ray_segment": c["SQ_INSTS_VALU"] / segs64,
        "cycles_per_valu_inst_per_simd": cyc,
        "issue_frac_vs_2cyc": 2.0 / cyc,
        "issue_frac_vs_measured_stream": MEASURED_STREAM_CYCLES_4_WAVES / cyc,
        "active_lanes": lanes,
        "useful_lane_slots_vs_2cyc": 2.0 / cyc * lanes,
        "salu_insts_per_64ray_segment": c.get("SQ_INSTS_SALU", 0.0) / segs64,
        "lds_bank_conflict_share_of_lds_cycles": (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_ACTIVE_INST_LDS"]) if c.get("SQ_ACTIVE_INST_LDS") else None,
        "wave_time": ({"issuing": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "issue_stalled": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                       "waiting": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]} if c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAIT_ANY") else None),
        "profiled_launch_ms": d["kernel_ns"] / 1e6, "profiled_rays_per_launch": d["rays_per_launch"],
        # the same workload: launches of the same size over other path ranges differ by a few ppm in ray count
        "rays_per_launch_match": abs(d["rays_per_launch"] - rays_per_launch) <= 1e-3 * rays_per_launch,
    }
    traffic = None
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        traffic = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0  # MI355X_MICROARCH.md: FETCH_SIZE halves wide reads on gfx950
    return out, traffic


def measure(R, config, fetch, rank, launches, batches_per_launch, warm_launches, seed, device):
    """Times `launches` fused launches of one config; returns the per-launch figures."""
    objs, cam, W, H, label = scene_of(R, config)
    scene = R.Scene(objs, cam, device=device)
    trace = R.TraceUnit(rank, W, H, n_photons=64, device=device)
    trace.set_fetch(R.FETCH_LDS if fetch == "lds" else R.FETCH_GLOBAL)
    plot = R.PlotUnit(rank, W, H, device=device)
    n = batches_per_launch * BATCH
    nxt = 0
    for _ in range(warm_launches):
        trace.render_fused(scene, plot, n, seed=seed, stream=rank, first_path_index=nxt)
        nxt += n
    trace.sync()
    p0, s0, ms0 = trace.stats()
    t0 = time.perf_counter()
    for _ in range(launches):
        trace.render_fused(scene, plot, n, seed=seed, stream=rank, first_path_index=nxt)
        nxt += n
    trace.sync()
    t1 = time.perf_counter()
    p1, s1, ms1 = trace.stats()
    f_seg = flops_per_ray(objs)
    launch_ms = (ms1 - ms0) / launches
    achieved = (s1 - s0) / launches * f_seg / (launch_ms * 1e-3) / 1e12
    return {"config": config, "workload": "built-in %s scene (%d objects), %dx%d, primitives in %s, %d launches of %d batches"
            % (label, len(objs), W, H, "LDS" if fetch == "lds" else "global/scalar cache", launches, batches_per_launch),
            "value": (s1 - s0) / (t1 - t0) / 1e6, "unit": "Mrays/s", "mpaths_per_s": (p1 - p0) / (t1 - t0) / 1e6,
            "kernel_ms_per_launch": launch_ms, "algorithmic_flops_per_ray": f_seg,
            "roofline_algorithmic": {"achieved": achieved, "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                                     "frac": achieved / PEAK_FP32_VECTOR_TFLOPS}}


def measure_app(R, fused, workers, device, batches=8192):
    """The drop-in at the reference's own task size: rl_app_run (TaskScheduler + worker pool, csrc/rl_app.cpp) with Trace
    tasks of 524,288 paths (trace_unit.rs:67) and as many workers as the host has cores (app.rs:55), at 1280x720."""
    rgb, st = R.app_run(1280, 720, batches, concurrency=workers, photons_per_batch=BATCH, fused=fused, verbose=False, device=device)
    return {"config": "app-720p-%s" % ("fused" if fused else "unfused"),
            "workload": "rl_app_run: built-in demo scene, 1280x720, %d batches of %d paths through the scheduler's Trace / Plot / Gather "
                        "tasks, %d workers, %s; includes the final tonemap"
                        % (batches, BATCH, workers, "Trace + Plot fused at plot time" if fused else "separate Trace and Plot tasks as in the reference"),
            "value": st["segments"] / st["seconds"] / 1e6, "unit": "Mrays/s", "mpaths_per_s": st["paths"] / st["seconds"] / 1e6,
            "batches_per_s": st["paths"] / BATCH / st["seconds"], "workers": workers, "seconds": st["seconds"]}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (the same environment contract
    torch.distributed.run sets up), relay rank 0's JSON line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RL_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # A rank that dies leaves the others waiting at the next barrier: watch all of them, stop the rest when one fails.
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=5)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit("bench.py: rank exit codes %r" % (rcs,))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="demo-1080p", choices=sorted(CONFIGS))
    ap.add_argument("--launches-per-step", type=int, default=8)
    ap.add_argument("--batches-per-launch", type=int, default=256)
    ap.add_argument("--fetch", default="lds", choices=["lds", "global"])
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the short runs of the other BASELINE configs (config.others)")
    ap.add_argument("--dist-backend", default="rccl", choices=["rccl", "gloo"],
                    help="rccl = the library's ncclReduce over xGMI, one rank per GPU (default); gloo = the same sum staged "
                         "through host memory, ranks may share a GPU -- for exercising the N > 1 path on a 1-GPU box")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before the HSA runtime starts: dmabuf IPC for RCCL
    from robigo_luculenta_amd import distributed as D
    rank, local_rank, world = D.env_rank()
    if world == 1 and args.gpus > 1:
        return spawn_ranks(args)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import numpy as np
    if world > 1:
        import torch  # noqa: F401  control plane only; imported before the library so that one HIP/RCCL copy is shared
    import robigo_luculenta_amd as R

    if R.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU implementation")
    D.init_control_plane(rank, world)
    comm, exchange_note = None, None
    if world > 1 and args.dist_backend == "rccl":
        # The library's RCCL communicator.  Should it fail to come up on ANY rank (a box without usable xGMI / IPC),
        # every rank falls back to the host-staged exchange together and the line says so -- a measured line with
        # a slower exchange is worth more than no line.  ncclCommInitRank is a collective: a rank that cannot even
        # get there (no GPU of its own, RCCL not loadable) would leave the others waiting inside it for ever, so the
        # ranks first agree that all of them can.
        failed = 0.0
        if local_rank >= R.device_count():
            exchange_note, failed = "rank %d has no GPU of its own (%d visible): RCCL admits one rank per device" % (rank, R.device_count()), 1.0
        elif rank == 0:   # (the other ranks load the same library on the same node)
            try:
                R.Comm.unique_id()   # loads RCCL; the id itself is not used
            except Exception as e:  # noqa: BLE001
                exchange_note, failed = "RCCL is not usable: %s" % e, 1.0
        _, (n_failed,) = D.aggregate(0.0, [failed])
        if not n_failed:
            try:
                comm = D.make_comm(R, rank, world, local_rank)
            except Exception as e:  # noqa: BLE001
                exchange_note, failed = "rl_comm_init_rank failed on rank %d: %s" % (rank, e), 1.0
            _, (n_failed,) = D.aggregate(0.0, [failed])
        if n_failed:
            comm = None
            exchange_note = exchange_note or "the communicator did not come up on another rank"
            args.dist_backend = "gloo (fallback: %s)" % exchange_note
    device = local_rank if comm is not None or world == 1 else local_rank % R.device_count()

    objs, cam, W, H, label = scene_of(R, args.config)
    scene = R.Scene(objs, cam, device=device)
    trace = R.TraceUnit(rank, W, H, n_photons=64, device=device)  # fused mode does not use mapped_photons
    trace.set_fetch(R.FETCH_LDS if args.fetch == "lds" else R.FETCH_GLOBAL)
    plot = R.PlotUnit(rank, W, H, device=device)
    gather = R.GatherUnit(W, H, device=device) if rank == 0 else None
    paths_per_launch = args.batches_per_launch * BATCH
    next_path = [0]

    def gather_step():
        """Task::Gather on `world` ranks (app.rs:143-148)."""
        if comm is not None:
            R.gather_allreduce(gather, plot, comm)   # rl_gather_unit_allreduce: ncclReduce onto rank 0, Kahan there, clear elsewhere
        elif world > 1:   # host-staged: download, gloo sum, upload on the root
            host = plot.tristimulus_buffer
            if D.host_staged_reduce(host, root=0):
                plot.upload(host)
                gather.accumulate(plot)
            else:
                plot.clear()
        else:
            gather.accumulate(plot)   # Kahan + clear (gather_unit.rs:49-64, app.rs:147)

    def step():
        for _ in range(args.launches_per_step):
            trace.render_fused(scene, plot, paths_per_launch, seed=args.seed, stream=rank, first_path_index=next_path[0])
            next_path[0] += paths_per_launch
        gather_step()

    def fence():
        trace.sync()
        plot.sync()
        if gather is not None:
            gather.sync()
        D.barrier()
        trace.sync()
        plot.sync()

    for _ in range(args.warmup):
        step()
    if args.warmup == 0 and world > 1:
        gather_step()  # build the communicator's rings outside the timed region (RCCL connects lazily)
    fence()
    p0, s0, ms0 = trace.stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    t1 = time.perf_counter()
    p1, s1, ms1 = trace.stats()
    rays, paths, kernel_ms = s1 - s0, p1 - p0, ms1 - ms0
    elapsed, (total_rays, total_paths) = D.aggregate(t1 - t0, [rays, paths])

    if rank == 0:
        f_seg = flops_per_ray(objs)
        n_launches = args.steps * args.launches_per_step
        launch_ms = kernel_ms / n_launches            # HIP events on the trace unit's own stream (rl_api.hip)
        rays_per_launch = rays / n_launches
        achieved = rays_per_launch * f_seg / (launch_ms * 1e-3) / 1e12
        ex = executed_from_profile(R, args.config, args.fetch, rays_per_launch, launch_ms)
        executed, traffic = ex if isinstance(ex, tuple) else (ex, None)
        out = {
            "metric": "Mrays/sec on built-in scene at %dx%d" % (W, H),
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "built-in %s scene (%d objects), %dx%d, fused trace+plot; one step = %d launches of %d batches of "
                                   "524288 paths per GPU, then the gather (%s); RNG stream = rank, primitives in %s"
                                   % (label, len(objs), W, H, args.launches_per_step, args.batches_per_launch,
                                      "Kahan accumulate + clear" if world == 1 else
                                      ("RCCL reduce of the XYZ buffers onto rank 0, Kahan accumulate, clear" if comm is not None
                                       else "host-staged gloo sum onto rank 0, Kahan accumulate, clear"),
                                      "LDS" if args.fetch == "lds" else "global/scalar cache"),
                       "config": args.config, "paths_per_step_per_gpu": paths_per_launch * args.launches_per_step,
                       "paths_per_launch": paths_per_launch, "seed": args.seed, "build_id": R.build_id(),
                       "dist_backend": None if world == 1 else args.dist_backend},
            "mpaths_per_s": total_paths / elapsed / 1e6,
            "batches_per_s": total_paths / elapsed / BATCH,
            "segments_per_path": total_rays / max(total_paths, 1.0),
            "timed_region_s": elapsed,
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VECTOR_TFLOPS,
                         "frac_is": "ALGORITHMIC: the reference's linear-scan flops per ray (SURVEY 8d) x rays / kernel time, over the "
                                    "FP32-vector peak.  The kernel culls most of that scan, so this is a speed-up-over-linear-scan "
                                    "figure, not a utilisation; `executed` below is what the hardware did",
                         "frac_of_unpacked_fp32_peak": achieved / (PEAK_FP32_VECTOR_TFLOPS / 2),  # 78.6 TFLOP/s, SURVEY 8(d)
                         "kernel": "rl_trace_kernel", "kernel_ms_per_launch": launch_ms, "rays_per_launch": rays_per_launch,
                         "algorithmic_flops_per_ray": f_seg,
                         "executed": executed,
                         "traffic": traffic,
                         "hbm": ({"achieved": traffic / (launch_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                  "frac": traffic / (launch_ms * 1e-3) / 8e12} if traffic else None),
                         "traffic_note": "HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB of the profile named in `executed` "
                                         "(each f32 atomic is billed as one 32-byte write); algorithmic: 48 B per contributing path; "
                                         "null when the profile is stale",
                         "note": "VALU-issue bound (no dense contraction -> no MFMA); HBM traffic is the XYZ splat only"},
        }
        if world == 1 and not args.no_others:
            out["config"]["others"] = [measure(R, c, f, rank, 6, 64, 1, args.seed, device) for c, f in OTHERS]
            workers = max(1, min(usable_cores(), 85))
            out["config"]["others"] += [measure_app(R, fused, workers, device) for fused in (False, True)]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(objs, cam, W, H)
        print(json.dumps(out), flush=True)
    if world > 1:
        D.shutdown()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- Mrays/s of the fused trace+plot hot path on the built-in scene.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: bench.py starts its own ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = `--launches-per-step` (2) fused TraceUnit::render + PlotUnit::plot launches of `--batches-per-launch`
(1024) batches of 524,288 camera paths each (trace_unit.rs:67) on the built-in demo scene (app.rs:166-363), then the
GatherUnit step: with N > 1 the ranks' XYZ plot buffers are summed onto rank 0 by the library's own RCCL exchange
(rl_plot_unit_reduce: one ncclReduce over xGMI), rank 0 Kahan-accumulates and every rank clears
(gather_unit.rs:49-64, app.rs:147).  Every rank renders the full frame with its own RNG stream (stream = rank) --
samples shard, nothing else is exchanged -- so per-GPU work is fixed as N grows ("weak"); `--total-paths T` fixes the
paths of a step for the whole job instead and splits them over the ranks ("strong").  A ray = one
Scene::intersect call (one path segment, scene.rs:39), counted on the device.  Rank 0 prints ONE JSON line; with
N > 1 it says what the exchange ran on (`config.rccl`: the communicator size and version RCCL itself reports) and
what it cost (`exchange`: device time of the ncclReduce per step, max over ranks).

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
# 256 CUs x 4 SIMDs x 64 lanes per wave64 instruction / 2 cycles x 2.4 GHz: VALU lane-operations per second (an FMA counts once)
PEAK_VALU_TLANEOPS = 256 * 4 * 64 / 2 * 2.4e9 / 1e12   # 78.6
PLAIN_STREAM_CYCLES_AT_4_WAVES = 2.6

CONFIGS = {
    # name: (scene, param, width, height)
    "demo-1080p": ("demo", 0, 1920, 1080),      # BASELINE.json metric config (configs[3] per GPU)
    "demo-720p": ("demo", 0, 1280, 720),        # configs[1]
    "glass-720p": ("glass", 0, 1280, 720),      # configs[2]
    "replicated-1080p": ("demo", 158, 1920, 1080),  # configs[4]
    "spill-1080p": ("demo", 1500, 1920, 1080),      # configs[4]'s "LDS-spill" half taken literally: 4,539 objects, does not fit LDS
    # ablation scenes (not BASELINE configs): prefixes of the demo scene's object list
    "ablate-noprisms": ("demo[:317]", 0, 1920, 1080),
    "ablate-fixed7": ("demo[:7]", 0, 1920, 1080),
    "ablate-empty": ("demo[:0]", 0, 1920, 1080),      # no object: every path is one segment into the Void (refill + loop overhead)
    "ablate-sun": ("demo[:1]", 0, 1920, 1080),        # the sun alone
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


def cpu_app(R, O, scene, width, height, workers, max_batches, seed=1):
    """The reference's App on the host: `workers` threads (app.rs:66-70) take tasks from ONE TaskScheduler
    (task_scheduler.rs:91-182 -- csrc/rl_scheduler.cpp through the C ABI, host code) under a mutex (app.rs:57,107) and
    execute them with the CPU oracle's units outside it: Trace = 524,288 paths into the unit's mapped_photons
    (trace_unit.rs:151-168), Plot, Gather (Kahan + clear), Tonemap, Sleep = 100 ms (app.rs:113-164).  Stops handing out
    Trace tasks after `max_batches`; returns (rays of the completed Trace tasks, paths, seconds, tasks by kind)."""
    import ctypes as C
    import threading
    import numpy as np
    sched = R.TaskScheduler(workers, tonemap_interval_ms=30000)
    n_trace, n_plot = 3 * workers, max(1, workers // 2)       # task_scheduler.rs:95-96
    photons = [np.zeros(BATCH, dtype=O.PHOTON_DTYPE) for _ in range(n_trace)]
    plots = [np.zeros((width * height, 3), np.float32) for _ in range(n_plot)]
    acc, comp = np.zeros((width * height, 3), np.float32), np.zeros((width * height, 3), np.float32)
    lock = threading.Lock()
    state = {"issued": 0, "done": 0, "rays": 0, "tasks": [0] * 5}
    t0 = time.perf_counter()

    def worker():
        task = R.Task()
        while True:
            with lock:
                task = sched.get_new_task(task, int((time.perf_counter() - t0) * 1000))
                batch = None
                if task.kind == R.TASK_TRACE:
                    if state["issued"] >= max_batches:
                        return                      # the budget is handed out: this worker is done (its unit stays taken)
                    batch = state["issued"]
                    state["issued"] += 1
                state["tasks"][task.kind] += 1
            if task.kind == R.TASK_TRACE:
                segs = C.c_uint64(0)
                O.lib().oracle_render(scene.h, width, height, seed, 0, batch * BATCH, BATCH, O.ptr(photons[task.unit]), C.byref(segs))
                with lock:
                    state["done"] += 1
                    state["rays"] += segs.value
            elif task.kind == R.TASK_PLOT:
                for u in task.units:
                    O.plot(width, height, photons[u], plots[task.unit])
            elif task.kind == R.TASK_GATHER:
                for u in task.units:
                    O.accumulate(acc, comp, plots[u])
                    plots[u][:] = 0.0
            elif task.kind == R.TASK_TONEMAP:
                O.tonemap(acc, width, height)
            else:
                time.sleep(0.1)                     # app.rs:129

    threads = [threading.Thread(target=worker) for _ in range(workers)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    return state["rays"], state["done"] * BATCH, time.perf_counter() - t0, state["tasks"]


def cpu_baseline(R, objs, cam, width, height, seconds_target=30.0):
    """SURVEY 8(d)'s CPU baseline: the CPU oracle (C++ restatement of the Rust reference, built -O3 without fast-math;
    the Rust crate cannot be built here: no rustc / cargo) driven like the reference drives its own units -- the
    TaskScheduler with concurrency = the host's usable cores and 524,288-path Trace tasks -- for >= ~30 s at the bench's
    resolution, plus BASELINE config 1 (256x256, ONE worker, seed 1) time-boxed to two of its 128 batches."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import _oracle as O
    workers = usable_cores()
    scene = O.Scene(objs.view(O.OBJECT_DTYPE), O.RlCameraDesc.from_buffer_copy(bytes(cam)))
    segs = C.c_uint64(0)
    t = time.perf_counter()
    import numpy as np
    scratch = np.zeros(20000, dtype=O.PHOTON_DTYPE)
    O.lib().oracle_render(scene.h, width, height, 1, 0, 1 << 40, 20000, O.ptr(scratch), C.byref(segs))   # calibrate one thread
    per_path = (time.perf_counter() - t) / 20000.0
    batches = max(workers, int(seconds_target * workers / (per_path * BATCH) + 0.5))
    rays, paths, dt, tasks = cpu_app(R, O, scene, width, height, workers, batches)
    rays1, paths1, dt1, tasks1 = cpu_app(R, O, scene, 256, 256, 1, 2)
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": workers, "kind": "port", "scheduler": True,
            "sample": "%d Trace tasks of %d paths (%d rays) of the same scene / resolution through the TaskScheduler with %d workers "
                      "(tasks Sleep/Trace/Plot/Gather/Tonemap = %s), seed 1, %.1f s; oracle built -O3" % (paths // BATCH, BATCH, rays, workers, tasks, dt),
            "mpaths_per_s": paths / dt / 1e6, "batches_per_s": paths / dt / BATCH,
            "config1": {"value": rays1 / dt1 / 1e6, "unit": "Mrays/s", "cores": 1, "batches_per_s": paths1 / dt1 / BATCH,
                        "sample": "BASELINE config 1 (built-in scene, 256x256, scheduler concurrency 1, seed 1) time-boxed: %d of its 128 "
                                  "batches (%d rays), %.1f s" % (paths1 // BATCH, rays1, dt1)},
            "one_thread": {"value": rays1 / dt1 / 1e6, "unit": "Mrays/s", "sample": "the config-1 run"}}


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


def chip_shape():
    """(CUs, SIMDs, XCDs) of the device the counters were collected on: the CU count from the device's properties (four SIMDs
    per CU on CDNA), the XCD count from the KFD topology (num_xcc; eight on MI300-class parts when that cannot be read) unless the counter files say
    otherwise (executed_live counts the GRBM_GUI_ACTIVE instances).  ADVICE r04 / r05: not hard-coded."""
    cus, xcds = 256, 8
    try:
        import torch
        if torch.cuda.is_available():
            cus = int(torch.cuda.get_device_properties(0).multi_processor_count)
    except Exception:   # noqa: BLE001 -- no torch, no device: the MI355X figures
        pass
    try:   # (the XCD count from the driver's topology -- num_xcc of the first GPU node -- not a literal)
        for node in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")):
            props = dict(l.split()[:2] for l in open(node).read().splitlines() if len(l.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0 and int(props.get("num_xcc", "0")) > 0:
                xcds = int(props["num_xcc"])
                break
    except (OSError, ValueError):
        pass
    return cus, 4 * cus, xcds


def issue_ceiling(executed):
    """The kernel's cycles per vector instruction per SIMD beside the wall-clock rate of a plain stream of independent f32 vector
    instructions at the same occupancy (tools/valu_microbench.hip, `wall c/i` at 4 waves per SIMD: 2.51-2.73 over the instruction
    kinds and rounds measured, profiles/r0*_valu_microbench.txt): what this chip issues in practice, against the nominal 2."""
    cyc = (executed or {}).get("cycles_per_valu_inst_per_simd")
    plain = {"cycles_per_valu_inst_plain_stream": PLAIN_STREAM_CYCLES_AT_4_WAVES, "valu_busy_ceiling": 2.0 / PLAIN_STREAM_CYCLES_AT_4_WAVES,
             "source": "tools/valu_microbench.hip wall-clock column at 4 waves per SIMD (profiles/r05_valu_microbench.txt: 2.51-2.73)"}
    if cyc:
        plain["kernel_cycles_per_valu_inst"] = cyc
        plain["kernel_vs_plain_stream"] = PLAIN_STREAM_CYCLES_AT_4_WAVES / cyc   # 1.0 = the kernel's vector instructions issue as fast as a plain stream's
    return plain


def executed_from_profile(R, config, fetch, rays_per_launch, launch_ms, paths_per_launch=None):
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
    cus, simds, xcds = chip_shape()
    cyc = (c["GRBM_GUI_ACTIVE"] / xcds) / (c["SQ_INSTS_VALU"] / simds)
    lanes = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
    out = {
        "profile": os.path.relpath(path, ROOT), "build_id": d["build_id"],
        "valu_insts_per_64ray_segment": c["SQ_INSTS_VALU"] / segs64,
        "cycles_per_valu_inst_per_simd": cyc,
        "issue_frac_vs_2cyc": 2.0 / cyc,
        "active_lanes": lanes,
        "useful_lane_slots_vs_2cyc": 2.0 / cyc * lanes,
        "salu_insts_per_64ray_segment": c.get("SQ_INSTS_SALU", 0.0) / segs64,
        # occupancy actually achieved: wave-cycles (a quad-cycle counter) per elapsed cycle and SIMD.  The compiler reports 118-120
        # VGPRs for this kernel, rocprofv3 "VGPR_Count 60" (the unified file counted in halves): 4 waves per SIMD either way.
        "resident_waves_per_simd": (4.0 * c["SQ_WAVE_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / xcds) / simds) if c.get("SQ_WAVE_CYCLES") else None,
        # round 5: a wave's time goes into issuing instructions of EVERY kind (and into LDS round trips) -- DESIGN.md 4.2
        "insts_per_64ray_segment": (c["SQ_INSTS"] / segs64) if c.get("SQ_INSTS") else None,
        "wave_cycles_per_inst": (4.0 * c["SQ_WAVE_CYCLES"] / c["SQ_INSTS"]) if c.get("SQ_INSTS") and c.get("SQ_WAVE_CYCLES") else None,
        # SQ_LDS_BANK_CONFLICT counts LDS-array cycles (one per extra address on a busy bank), summed over the CUs; GRBM_GUI_ACTIVE is
        # summed over the 8 XCDs: conflict cycles per CU-cycle = the share of time a CU's LDS spends on conflicts (round 3 divided
        # by SQ_ACTIVE_INST_LDS, a quad-cycle counter of something else -- VERDICT r03).  Attribution: profiles/r04_lds_conflicts.txt.
        "lds_bank_conflict_cycles_per_cu_cycle": (c["SQ_LDS_BANK_CONFLICT"] / (c["GRBM_GUI_ACTIVE"] / xcds * cus)) if c.get("SQ_LDS_BANK_CONFLICT") else None,
        "lds_bank_conflict_share_of_lds_array_cycles": (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None,
        "wave_time": ({"issuing": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "issue_stalled": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                       "waiting": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]} if c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAIT_ANY") else None),
        "profiled_launch_ms": d["kernel_ns"] / 1e6, "profiled_rays_per_launch": d["rays_per_launch"],
        # the same workload: rays per path of the profiled launch and of this run's launches (other path ranges, possibly
        # another launch size) agree to a few ppm
        "rays_per_path_match": abs(d["rays_per_launch"] / d["paths_per_launch"] - rays_per_launch / paths_per_launch) <= 2e-3 * rays_per_launch / paths_per_launch
                               if paths_per_launch else None,
    }
    traffic = None
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        traffic = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0  # MI355X_MICROARCH.md: FETCH_SIZE halves wide reads on gfx950
        if paths_per_launch and d.get("paths_per_launch"):  # per launch of THIS run: the splat's bytes are proportional to the paths
            traffic *= paths_per_launch / d["paths_per_launch"]
    return out, traffic


LIVE_PASSES = (
    "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS SQ_INSTS_SALU GRBM_GUI_ACTIVE",
    "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES GRBM_GUI_ACTIVE",
)


def executed_live(args, committed, timeout_s=100.0):
    """the counter-derived figures measured in THIS run instead of quoted from a committed file.  After the
    timed region, bench.py runs itself twice under `rocprofv3 --pmc` (the VALU pass and the wait pass of
    tools/profile_round.sh) for one 256-batch launch of the same config and reads the trace kernel's last dispatch from the
    counter CSVs.  Returns the same keys as `executed` plus `agrees` (with the committed profile, at 2 %); {"skipped": why}
    when rocprofv3 is not on the box or a pass fails -- never an error: the bench line does not depend on it."""
    import collections
    import csv
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return {"skipped": "rocprofv3 is not on this box"}
    t_begin = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="rl_live_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", RL_BENCH_LIVE_CHILD="1")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--launches-per-step", "1", "--batches-per-launch", "256",
             "--config", args.config, "--fetch", args.fetch, "--seed", str(args.seed), "--no-cpu-baseline", "--no-others", "--no-live-counters"]
    c, line, kernel_ns, instances = {}, None, [], {}
    try:
        for i, counters in enumerate(LIVE_PASSES):
            left = timeout_s - (time.perf_counter() - t_begin)
            if left < 15.0:
                return {"skipped": "no time left for counter pass %d" % i}
            out_dir = os.path.join(tmp, "p%d" % i)
            run = subprocess.run([prof, "--pmc"] + counters.split() + ["-f", "csv", "-d", out_dir, "-o", "p", "--"] + child, env=env, cwd="/tmp",
                                 capture_output=True, timeout=left)
            lines = [l for l in run.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
            files = glob.glob(os.path.join(out_dir, "**", "p_counter_collection.csv"), recursive=True)
            if run.returncode != 0 or not lines or not files:
                return {"skipped": "counter pass %d failed (rc %d): %s" % (i, run.returncode, run.stderr.decode(errors="replace")[-300:])}
            line = json.loads(lines[-1])
            d = collections.defaultdict(lambda: collections.defaultdict(float))
            rows = collections.defaultdict(lambda: collections.defaultdict(int))
            ns = {}
            for r in csv.DictReader(open(files[0])):
                if "rl_trace" in r["Kernel_Name"]:
                    d[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
                    rows[r["Dispatch_Id"]][r["Counter_Name"]] += 1
                    ns[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if not d:
                return {"skipped": "counter pass %d saw no trace kernel" % i}
            k = sorted(d, key=int)[-1]
            for name, v in d[k].items():
                c.setdefault(name, v)     # (SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE of the first pass that has them)
                instances.setdefault(name, rows[k][name])
            kernel_ns.append(ns[k])
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as e:
        return {"skipped": "counter pass failed: %r" % (e,)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    try:   # (a counter the profiler dropped, or one that reads zero, must not cost the bench its line)
        segs64 = line["roofline"]["rays_per_launch"] / 64.0
        cus, simds, xcds = chip_shape()
        xcds = instances.get("GRBM_GUI_ACTIVE") or xcds   # (one GRBM instance per XCD; rocprofv3 may also report the sum as one row)
        if instances.get("GRBM_GUI_ACTIVE") == 1:
            xcds = chip_shape()[2]
        cyc = (c["GRBM_GUI_ACTIVE"] / xcds) / (c["SQ_INSTS_VALU"] / simds)
        lanes = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
        live = {
            "how": "rocprofv3 --pmc around `bench.py --steps 1 --launches-per-step 1 --batches-per-launch 256` of this config, run by this "
                   "process after its timed region: two passes, last rl_trace_kernel dispatch of each",
            "build_id": line["config"]["build_id"],
            "valu_insts_per_64ray_segment": c["SQ_INSTS_VALU"] / segs64,
            "cycles_per_valu_inst_per_simd": cyc,
            "issue_frac_vs_2cyc": 2.0 / cyc,
            "active_lanes": lanes,
            "useful_lane_slots_vs_2cyc": 2.0 / cyc * lanes,
            "salu_insts_per_64ray_segment": c.get("SQ_INSTS_SALU", 0.0) / segs64,
            "resident_waves_per_simd": 4.0 * c["SQ_WAVE_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / xcds) / simds,
            "insts_per_64ray_segment": (c["SQ_INSTS"] / segs64) if c.get("SQ_INSTS") else None,
            "wave_cycles_per_inst": (4.0 * c["SQ_WAVE_CYCLES"] / c["SQ_INSTS"]) if c.get("SQ_INSTS") else None,
            "wave_time": {"issuing": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "issue_stalled": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                          "waiting": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]},
            "lds_bank_conflict_cycles_per_cu_cycle": c["SQ_LDS_BANK_CONFLICT"] / (c["GRBM_GUI_ACTIVE"] / xcds * cus),
            "lds_bank_conflict_share_of_lds_array_cycles": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else None,
            "profiled_launch_ms": sum(kernel_ns) / len(kernel_ns) / 1e6, "profiled_rays_per_launch": line["roofline"]["rays_per_launch"],
            "seconds": time.perf_counter() - t_begin,
        }
    except (KeyError, ZeroDivisionError, TypeError) as e:
        return {"skipped": "counter derivation failed: %r" % (e,)}
    if committed and not committed.get("stale"):
        keys = ("valu_insts_per_64ray_segment", "cycles_per_valu_inst_per_simd", "active_lanes", "useful_lane_slots_vs_2cyc")
        live["vs_committed"] = {k: live[k] / committed[k] for k in keys}
        live["agrees"] = all(abs(v - 1.0) <= 0.02 for v in live["vs_committed"].values())
    else:
        live["agrees"] = None
    return live


def staged_where(ran):
    """What the launches behind `ran` (a difference of rl_debug_variant_launches) kept in LDS, as the workload text says it."""
    if sum(ran[16:24]):
        return "tables (planes, prisms, cull table) in LDS, spheres and objects from L2 / HBM"
    return "primitives in LDS" if sum(ran[8:16]) else "primitives in global/scalar cache"


def measure(R, config, fetch, rank, launches, batches_per_launch, warm_launches, seed, device):
    """Times `launches` fused launches of one config; returns the per-launch figures."""
    objs, cam, W, H, label = scene_of(R, config)
    scene = R.Scene(objs, cam, device=device)
    trace = R.TraceUnit(rank, W, H, n_photons=64, device=device)
    trace.set_fetch(R.FETCH_LDS if fetch == "lds" else R.FETCH_GLOBAL)
    plot = R.PlotUnit(rank, W, H, device=device)
    n = batches_per_launch * BATCH
    nxt = 0
    variants0 = R.variant_launches()
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
    ex = executed_from_profile(R, config, fetch, (s1 - s0) / launches, launch_ms, n)
    executed = ex[0] if isinstance(ex, tuple) else ex
    if not executed.get("stale"):   # the short form: what SURVEY 8(d) asks for per config (VALUUtilization = active lanes)
        executed = {k: executed[k] for k in ("profile", "build_id", "valu_insts_per_64ray_segment", "cycles_per_valu_inst_per_simd",
                                              "issue_frac_vs_2cyc", "active_lanes", "useful_lane_slots_vs_2cyc")}
    return {"config": config, "executed": executed, "workload": "built-in %s scene (%d objects), %dx%d, %s, %d launches of %d batches"
            % (label, len(objs), W, H, staged_where([a - b for a, b in zip(R.variant_launches(), variants0)]), launches, batches_per_launch),
            "value": (s1 - s0) / (t1 - t0) / 1e6, "unit": "Mrays/s", "mpaths_per_s": (p1 - p0) / (t1 - t0) / 1e6,
            "kernel_ms_per_launch": launch_ms, "algorithmic_flops_per_ray": f_seg,
            "roofline_algorithmic": {"achieved": achieved, "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                                     "frac": achieved / PEAK_FP32_VECTOR_TFLOPS}}


def measure_app(R, fused, threads, device, batches=8192, depth=64):
    """The drop-in at the reference's own task size: rl_app_run (TaskScheduler + worker pool, csrc/rl_app.cpp) with Trace
    tasks of 524,288 paths (trace_unit.rs:67) at 1280x720.  Round 6: the scheduler's depth (RlAppConfig::concurrency: 3 x depth
    trace units in circulation, task_scheduler.rs:95-96) is no longer the host thread count (RlAppConfig::threads) -- a few host
    threads keep a deep pool of begun renders in flight (profiles/r06_app.txt: depth x threads)."""
    rgb, st = R.app_run(1280, 720, batches, concurrency=depth, threads=threads, photons_per_batch=BATCH, fused=fused, verbose=False, device=device)
    return {"config": "app-720p-%s" % ("fused" if fused else "unfused"),
            "workload": "rl_app_run: built-in demo scene, 1280x720, %d batches of %d paths through the scheduler's Trace / Plot / Gather "
                        "tasks, scheduler depth %d, %d host threads, %s; includes the final tonemap"
                        % (batches, BATCH, depth, threads, "Trace + Plot fused at plot time" if fused else "separate Trace and Plot tasks as in the reference"),
            "value": st["segments"] / st["seconds"] / 1e6, "unit": "Mrays/s", "mpaths_per_s": st["paths"] / st["seconds"] / 1e6,
            "batches_per_s": st["paths"] / BATCH / st["seconds"], "workers": threads, "depth": depth, "seconds": st["seconds"]}


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


def rccl_report(world, rccl_used, info, sum_worlds, backend):
    """`config.rccl` of an N > 1 line: did RCCL see N ranks?  `world` is what every rank's communicator reports
    (ncclCommCount), summed over the ranks by the control plane and divided by N -- N iff all of them say N."""
    return {"world": sum_worlds / world, "ranks_agree": sum_worlds == float(world) * world, "version": info.get("rccl_version"),
            "library": info.get("library"), "backend_used": "rccl" if rccl_used else backend}


def metric_name(W, H, world, scaling):
    """BASELINE.json's metric; at N > 1 the name also says which scaling mode `value` is (SURVEY 8(e) defines the
    metric on a fixed total -- strong --, the driver's contract lets per-GPU work stay fixed -- weak, the default; every N > 1 line
    carries BOTH in `scaling_detail`, and `value` is the one named here)."""
    name = "Mrays/sec on built-in scene at %dx%d" % (W, H)
    if world > 1:
        name += ("; %d GPUs, WEAK scaling (per-GPU paths fixed; strong in scaling_detail)" if scaling == "weak"
                 else "; %d GPUs, STRONG scaling (total paths fixed; weak in scaling_detail)") % world
    return name


def scaling_detail(world, scaling, value, n1, other):
    """`scaling_detail` of an N > 1 line: the one-GPU rate of the same run and both scaling modes against it.
    efficiency = N-rank rays/s / (N x the one-GPU rays/s); SURVEY 8(e)'s target is >= 0.9 at N = 8."""
    if n1 is None:
        return None
    out = {"n1_same_run": n1, scaling: {"value": value, "efficiency": value / (world * n1["value"]), "headline": True}}
    if other is not None:
        out[other["scaling"]] = dict(other, efficiency=other["value"] / (world * n1["value"]), headline=False)
    out["efficiency_is"] = "whole-job Mrays/s / (n_gpus x n1_same_run.value); >= 0.9 is SURVEY 8(e)'s target"
    return out


def dry_run(args, D, R, rank, world, paths_per_launch, scaling):
    """The control plane and the line's N-rank fields without a GPU: every rank contributes made-up counters (rank r:
    1000 (r + 1) rays per path-thousand, 2 ms of exchange per step) through D.aggregate exactly like a real run."""
    D.init_control_plane(rank, world)
    paths = float(paths_per_launch * args.launches_per_step * args.steps)
    rays = paths * (3.0 + rank)
    elapsed, (total_rays, total_paths) = D.aggregate(1.0 + 0.25 * rank, [rays, paths])
    exchange_ms, (sum_worlds, sum_exchanges, _) = D.aggregate(2.0 + rank, [world, args.steps, 0.0])
    # the same-run one-GPU rate and the other scaling mode, made up the same way (rank 0 alone: 3 rays per path in 1 s per step)
    n1 = {"value": paths / args.steps * 3.0 / 1.0 / 1e6, "unit": "Mrays/s", "steps": 1, "paths_per_step": paths / args.steps, "what": "dry run"}
    o_dt, (o_rays, _) = D.aggregate((1.0 + 0.25 * rank) / world, [rays / world, paths / world])
    other = {"scaling": "strong" if scaling == "weak" else "weak", "value": o_rays / o_dt / 1e6, "unit": "Mrays/s", "steps": args.steps,
             "ms_per_step": o_dt / args.steps * 1e3, "paths_per_step_per_gpu": paths / args.steps / world, "what": "dry run"}
    if rank == 0:
        print(json.dumps({
            "metric": metric_name(1920, 1080, world, scaling), "value": total_rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "none (dry run of the control plane: no GPU work, made-up counters)",
            "config": {"workload": "dry run", "paths_per_launch": paths_per_launch, "total_paths_per_step": paths_per_launch * args.launches_per_step * world,
                       "dist_backend": "gloo (dry run)",
                       "rccl": rccl_report(world, False, {"rccl_version": 0, "library": None}, sum_worlds, "gloo (dry run)")},
            "mpaths_per_s": total_paths / elapsed / 1e6,
            "scaling_detail": None if world == 1 else scaling_detail(world, scaling, total_rays / elapsed / 1e6, n1, other),
            "exchange": {"ms_per_step": exchange_ms, "per_step": sum_exchanges / world / args.steps, "share_of_step": exchange_ms / (elapsed / args.steps * 1e3)}}), flush=True)
    if world > 1:
        D.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="demo-1080p", choices=sorted(CONFIGS))
    ap.add_argument("--launches-per-step", type=int, default=2)
    ap.add_argument("--batches-per-launch", type=int, default=1024)
    ap.add_argument("--fetch", default="lds", choices=["lds", "global"])
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the short runs of the other BASELINE configs (config.others)")
    ap.add_argument("--no-live-counters", action="store_true",
                    help="skip roofline.executed_live (bench.py re-running itself under rocprofv3 --pmc after the timed region, N = 1 only)")
    ap.add_argument("--total-paths", type=int, default=0,
                    help="strong scaling (SURVEY 8e): the paths of ONE STEP for the whole job, split evenly over the ranks "
                         "(rounded down to launches-per-step x 64 per rank); default 0 = weak scaling, the per-GPU work is fixed")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: every rank reports made-up counters through the same control plane and line assembly "
                         "(tests/test_distributed_gloo.py runs this with two CPU ranks)")
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

    paths_per_launch = args.batches_per_launch * BATCH
    scaling = "weak"
    if args.total_paths:
        scaling = "strong"
        paths_per_launch = args.total_paths // world // args.launches_per_step // 64 * 64
        if paths_per_launch == 0:
            raise SystemExit("bench.py: --total-paths %d is less than 64 paths per launch and rank" % args.total_paths)
    if args.dry_run:
        return dry_run(args, D, R, rank, world, paths_per_launch, scaling)
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
        # one GPU per rank?  Decided by what the ranks' devices ARE (PCI bus ids), not by their numbers: a launcher may leave
        # all GPUs visible to every rank (device = local rank) or hand each rank a mask of its own (device 0 everywhere)
        mine = D.pick_device(local_rank, R.device_count())
        ids = D.all_gather_strings(socket.gethostname()[:24] + "/" + R.device_pci_bus_id(mine))
        if not D.one_gpu_per_rank(ids):
            exchange_note, failed = "ranks share a GPU (%s): RCCL admits one rank per device" % ", ".join(ids), 1.0
        elif rank == 0:   # (the other ranks load the same library on the same node)
            try:
                R.Comm.unique_id()   # loads RCCL; the id itself is not used
            except Exception as e:  # noqa: BLE001
                exchange_note, failed = "RCCL is not usable: %s" % e, 1.0
        _, (n_failed,) = D.aggregate(0.0, [failed])
        if not n_failed:
            try:
                comm = D.make_comm(R, rank, world, D.pick_device(local_rank, R.device_count()))
            except Exception as e:  # noqa: BLE001
                exchange_note, failed = "rl_comm_init_rank failed on rank %d: %s" % (rank, e), 1.0
            _, (n_failed,) = D.aggregate(0.0, [failed])
        if n_failed:
            comm = None
            exchange_note = exchange_note or "the communicator did not come up on another rank"
            args.dist_backend = "gloo (fallback: %s)" % exchange_note
    device = D.pick_device(local_rank, R.device_count())

    objs, cam, W, H, label = scene_of(R, args.config)
    scene = R.Scene(objs, cam, device=device)
    trace = R.TraceUnit(rank, W, H, n_photons=64, device=device)  # fused mode does not use mapped_photons
    trace.set_fetch(R.FETCH_LDS if args.fetch == "lds" else R.FETCH_GLOBAL)
    variants0 = R.variant_launches()
    plot = R.PlotUnit(rank, W, H, device=device)
    gather = R.GatherUnit(W, H, device=device) if rank == 0 else None
    next_path = [0]
    host_exchange = [0.0, 0]   # the host-staged fallback: seconds and count of the gloo sums

    def gather_step():
        """Task::Gather on `world` ranks (app.rs:143-148)."""
        if comm is not None:
            R.gather_allreduce(gather, plot, comm)   # rl_gather_unit_allreduce: ncclReduce onto rank 0, Kahan there, clear elsewhere
        elif world > 1:   # host-staged: download, gloo sum, upload on the root
            host = plot.tristimulus_buffer
            t_ex = time.perf_counter()
            holds_sum = D.host_staged_reduce(host, root=0)
            host_exchange[0] += time.perf_counter() - t_ex
            host_exchange[1] += 1
            if holds_sum:
                plot.upload(host)
                gather.accumulate(plot)
            else:
                plot.clear()
        else:
            gather.accumulate(plot)   # Kahan + clear (gather_unit.rs:49-64, app.rs:147)

    def step(ppl=None, alone=False):
        ppl = ppl or paths_per_launch
        for _ in range(args.launches_per_step):
            trace.render_fused(scene, plot, ppl, seed=args.seed, stream=rank, first_path_index=next_path[0])
            next_path[0] += ppl
        if alone:
            gather.accumulate(plot)   # rank 0 on its own: no exchange
        else:
            gather_step()

    def timed(steps, ppl=None):
        """`steps` steps between fences; returns (max-over-ranks seconds, total rays, total paths)."""
        fence()
        pa, sa, _ = trace.stats()
        ta = time.perf_counter()
        for _ in range(steps):
            step(ppl)
        fence()
        tb = time.perf_counter()
        pb, sb, _ = trace.stats()
        dt, (r, q) = D.aggregate(tb - ta, [sb - sa, pb - pa])
        return dt, r, q

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
    # an N-rank line that answers ">= 90 % of linear?" by itself.  Before the timed region rank 0 ALONE runs a
    # few steps of the one-GPU workload (the others wait at the barrier): the N = 1 rate of this very run, on this very box.
    n1 = None
    if world > 1:
        n1_steps = max(1, min(3, args.steps))
        n1_ppl = args.batches_per_launch * BATCH if not args.total_paths else args.total_paths // args.launches_per_step // 64 * 64
        if rank == 0:
            trace.sync()
            pa, sa, _ = trace.stats()
            ta = time.perf_counter()
            for _ in range(n1_steps):
                step(n1_ppl, alone=True)
            trace.sync(); plot.sync(); gather.sync()
            tb = time.perf_counter()
            pb, sb, _ = trace.stats()
            n1 = {"value": (sb - sa) / (tb - ta) / 1e6, "unit": "Mrays/s", "steps": n1_steps, "paths_per_step": n1_ppl * args.launches_per_step,
                  "what": "rank 0 alone (the other ranks idle at a barrier), same scene / resolution / launch structure, no exchange, before the timed region"}
        fence()
    p0, s0, ms0 = trace.stats()
    ex_n0, ex_ms0 = plot.exchange_stats() if comm is not None else (0, 0.0)
    hx0 = list(host_exchange)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    t1 = time.perf_counter()
    p1, s1, ms1 = trace.stats()
    rays, paths, kernel_ms = s1 - s0, p1 - p0, ms1 - ms0
    elapsed, (total_rays, total_paths) = D.aggregate(t1 - t0, [rays, paths])
    # the exchange, per step: device time of the ncclReduce calls on this rank's plot stream (rl_plot_unit_exchange_stats),
    # or host time of the staged sums; MAX over ranks (a collective ends when its slowest rank does)
    if comm is not None:
        ex_n1, ex_ms1 = plot.exchange_stats()
        my_exchange_ms, my_exchanges = (ex_ms1 - ex_ms0) / args.steps, ex_n1 - ex_n0
        info = comm.info()
    else:
        my_exchange_ms, my_exchanges = (host_exchange[0] - hx0[0]) * 1e3 / args.steps, host_exchange[1] - hx0[1]
        info = {"world": world, "rccl_version": 0, "library": None}
    exchange_ms, (sum_worlds, sum_exchanges, sum_kernel_ms) = D.aggregate(my_exchange_ms, [info["world"], my_exchanges, kernel_ms])
    # ... and the OTHER scaling mode in the same line: the headline is weak scaling (per-GPU work fixed: what the driver's --gpus N
    # runs mean) unless --total-paths asks for strong; SURVEY 8(e) defines the metric on a fixed total, so the line carries both.
    other = None
    if world > 1:
        if scaling == "weak":   # strong: ONE GPU's step split over the ranks
            o_ppl = paths_per_launch // world // 64 * 64
            o_name, o_what = "strong", "the paths of one GPU's step (%d) split evenly over the ranks" % (paths_per_launch * args.launches_per_step)
        else:                   # weak: every rank renders the whole of --total-paths
            o_ppl = args.total_paths // args.launches_per_step // 64 * 64
            o_name, o_what = "weak", "every rank renders %d paths per step" % (o_ppl * args.launches_per_step)
        if o_ppl >= 64:
            o_steps = max(1, min(args.steps, 10))
            o_dt, o_rays, o_paths = timed(o_steps, o_ppl)
            other = {"scaling": o_name, "value": o_rays / o_dt / 1e6, "unit": "Mrays/s", "steps": o_steps, "ms_per_step": o_dt / o_steps * 1e3,
                     "paths_per_step_per_gpu": o_ppl * args.launches_per_step, "what": o_what}

    if rank == 0:
        f_seg = flops_per_ray(objs)
        n_launches = args.steps * args.launches_per_step
        launch_ms = kernel_ms / n_launches            # HIP events on the trace unit's own stream (rl_api.hip)
        rays_per_launch = rays / n_launches
        achieved = rays_per_launch * f_seg / (launch_ms * 1e-3) / 1e12
        ex = executed_from_profile(R, args.config, args.fetch, rays_per_launch, launch_ms, paths_per_launch)
        executed, traffic = ex if isinstance(ex, tuple) else (ex, None)
        out = {
            "metric": metric_name(W, H, world, scaling),
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "built-in %s scene (%d objects), %dx%d, fused trace+plot; one step = %d launches of %d batches of "
                                   "524288 paths per GPU, then the gather (%s); RNG stream = rank, %s"
                                   % (label, len(objs), W, H, args.launches_per_step, args.batches_per_launch,
                                      "Kahan accumulate + clear" if world == 1 else
                                      ("RCCL reduce of the XYZ buffers onto rank 0, Kahan accumulate, clear" if comm is not None
                                       else "host-staged gloo sum onto rank 0, Kahan accumulate, clear"),
                                      staged_where([a - b for a, b in zip(R.variant_launches(), variants0)])),
                       "config": args.config, "paths_per_step_per_gpu": paths_per_launch * args.launches_per_step,
                       "paths_per_launch": paths_per_launch, "seed": args.seed, "build_id": R.build_id(),
                       "total_paths_per_step": paths_per_launch * args.launches_per_step * world,
                       "dist_backend": None if world == 1 else args.dist_backend,
                       "rccl": None if world == 1 else rccl_report(world, comm is not None, info, sum_worlds, args.dist_backend)},
            "mpaths_per_s": total_paths / elapsed / 1e6,
            "batches_per_s": total_paths / elapsed / BATCH,
            "segments_per_path": total_rays / max(total_paths, 1.0),
            "timed_region_s": elapsed,
            "exchange": None if world == 1 else {
                "ms_per_step": exchange_ms, "per_step": sum_exchanges / world / args.steps, "bytes": 3 * W * H * 4,
                "what": ("device time of ncclReduce (f32 sum of the %d-float XYZ buffer onto rank 0) between events on the plot unit's "
                         "stream, max over ranks" % (3 * W * H)) if comm is not None else "host time of the gloo sum of the downloaded buffers, max over ranks",
                "share_of_step": exchange_ms / (elapsed / args.steps * 1e3),
                "kernel_share_of_step": (sum_kernel_ms / world) / (elapsed * 1e3)},
            "scaling_detail": None if world == 1 else scaling_detail(world, scaling, total_rays / elapsed / 1e6, n1, other),
            # `frac` is a UTILISATION -- the share of the vector ALU's lane-slots the kernel executed, from the
            # counters (this run's own rocprofv3 passes when they ran, else the committed profile of this build) -- and
            # achieved / peak are that same quantity in lane-operations per second.  The reference-flops figure of rounds 1-5
            # (> 1 because the kernel culls most of the reference's linear scan) is kept as `algorithmic`, labelled as a speed-up.
            "roofline": {"bound": "valu", "unit": "Tlane-op/s", "peak": PEAK_VALU_TLANEOPS,
                         "achieved": (executed.get("useful_lane_slots_vs_2cyc") or 0.0) * PEAK_VALU_TLANEOPS if executed.get("useful_lane_slots_vs_2cyc") else None,
                         "frac": executed.get("useful_lane_slots_vs_2cyc"),
                         "frac_source": None if executed.get("stale") else executed.get("profile"),
                         "frac_is": "UTILISATION: VALU lane-slots executed (SQ_THREAD_CYCLES_VALU) / lane-slots the chip offers at one wave64 "
                                    "instruction per 2 cycles per SIMD over the kernel's cycles = valu_busy x active lanes; from the counters",
                         "frac_executed": executed.get("useful_lane_slots_vs_2cyc"),   # (the name rounds 3-5 reported it under)
                         "valu_busy": executed.get("issue_frac_vs_2cyc"),              # VALU instructions issued / issue slots at that rate
                         # how far the kernel's vector issue rate is from what this chip issues IN PRACTICE at the kernel's occupancy:
                         # tools/valu_microbench.hip, wall-clock cycles per wave64 instruction per SIMD of a plain independent stream
                         "issue_ceiling": issue_ceiling(executed),
                         "algorithmic": {"achieved": achieved, "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                                         "frac": achieved / PEAK_FP32_VECTOR_TFLOPS,
                                         "frac_of_unpacked_fp32_peak": achieved / (PEAK_FP32_VECTOR_TFLOPS / 2),  # 78.6 TFLOP/s, SURVEY 8(d)
                                         "is": "SPEED-UP over the reference's linear scan, not a utilisation: the reference's reject-path flops per "
                                               "ray (SURVEY 8d) x rays / kernel time over the FP32-vector peak; above 1 because the kernel culls most of that scan "
                                               "(bit-identical photons and segment counts)"},
                         "frac_algorithmic": achieved / PEAK_FP32_VECTOR_TFLOPS,
                         "kernel": "rl_trace_kernel", "kernel_ms_per_launch": launch_ms, "rays_per_launch": rays_per_launch,
                         "algorithmic_flops_per_ray": f_seg,
                         "executed": executed,
                         "executed_live": None,   # filled in below (N = 1, rocprofv3 on the box): the same figures counted in this run
                         "traffic": traffic,
                         "hbm": ({"achieved": traffic / (launch_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                  "frac": traffic / (launch_ms * 1e-3) / 8e12} if traffic else None),
                         "traffic_note": "HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB of the profile named in `executed`, scaled to this run's paths per launch "
                                         "(each f32 atomic is billed as one 32-byte write); algorithmic: 48 B per contributing path; "
                                         "null when the profile is stale",
                         "note": "no dense contraction -> no MFMA; HBM traffic is the XYZ splat only (3 % of peak).  Bound: the vector ALU's issue slots at four waves per SIMD -- "
                                 "the kernel issues one vector instruction per ~2.75 cycles per SIMD against ~2.6 for a plain stream of independent FMAs at this "
                                 "occupancy (issue_ceiling.kernel_vs_plain_stream), a fifth wave per SIMD gains 0.4-1.7 % (profiles/r06_fifth_wave.txt): what is left "
                                 "is fewer vector instructions per ray.  DESIGN.md 4.1"},
        }
        if world == 1 and not args.no_others:
            out["config"]["others"] = [measure(R, c, f, rank, 6, 64, 1, args.seed, device) for c, f in OTHERS]
            workers = max(1, min(usable_cores(), 4))
            out["config"]["others"] += [measure_app(R, fused, workers, device) for fused in (False, True)]
        if world == 1 and not args.no_live_counters and not os.environ.get("RL_BENCH_LIVE_CHILD"):
            live = executed_live(args, executed)
            out["roofline"]["executed_live"] = live
            if not live.get("skipped"):   # counted in THIS run: the live counters lead, the committed profile is the cross-check
                rf = out["roofline"]
                rf["frac"] = rf["frac_executed"] = live["useful_lane_slots_vs_2cyc"]
                rf["valu_busy"] = live["issue_frac_vs_2cyc"]
                rf["achieved"] = rf["frac"] * PEAK_VALU_TLANEOPS
                rf["frac_source"] = "executed_live (rocprofv3 --pmc passes of this run)"
                rf["issue_ceiling"] = issue_ceiling(live)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(R, objs, cam, W, H)
        print(json.dumps(out), flush=True)
    if world > 1:
        D.shutdown()


if __name__ == "__main__":
    main()

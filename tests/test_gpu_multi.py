"""GPU tests of what round 2 added around the trace kernel: explicit stream ordering, the GatherUnit-time exchange
(RCCL through the C ABI, device-local add, host-staged), the multi-rank App, resume without repeated samples, the
degenerate-image tonemap, and bench.py's N > 1 branch on one GPU."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def R():
    import robigo_luculenta_amd as R
    assert R.device_count() > 0
    return R


def _ocam(cam):
    return O.RlCameraDesc.from_buffer_copy(bytes(cam))


def test_black_image_tonemaps_to_black_like_the_reference(R):
    """max_intensity = 0 makes every pixel 0/0 = NaN; the reference carries NaN through ln, the matrix, the gamma curve
    and the clamp to `NaN as u8` = 0 (tonemap_unit.rs:73-100).  ADVICE r01: the kernel used to write 255."""
    W, H = 64, 36
    g, tm = R.GatherUnit(W, H), R.TonemapUnit(W, H)
    tm.tonemap(g)
    want_rgb, want_srgb, want_mx = O.tonemap(np.zeros((W * H, 3), np.float32), W, H)
    assert not want_rgb.any() and want_mx == 0.0
    assert tm.rgb_buffer.tobytes() == want_rgb.tobytes()
    srgb, mx = tm.srgb_float()
    assert mx == 0.0 and np.isnan(srgb).all() and np.isnan(want_srgb).all()
    # an image with a single lit pixel: finite exposure, everything else exactly as the oracle
    xyz = np.zeros((W * H, 3), np.float32)
    xyz[5] = (0.3, 0.5, 0.2)
    p = R.PlotUnit(0, W, H)
    p.upload(xyz)
    g.accumulate(p)
    tm.tonemap(g)
    want_rgb, want_srgb, _ = O.tonemap(xyz, W, H)
    assert tm.rgb_buffer.tobytes() == want_rgb.tobytes() and tm.srgb_float()[0].tobytes() == want_srgb.tobytes()


def test_plot_unit_add_upload_and_async_ordering(R):
    W, H, N = 96, 54, 1 << 14
    objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
    scene = R.Scene(objs, cam)
    oscene = O.Scene(objs.view(O.OBJECT_DTYPE), _ocam(cam))
    t0, t1 = R.TraceUnit(0, W, H, n_photons=N), R.TraceUnit(1, W, H, n_photons=N)
    a, b = R.PlotUnit(0, W, H), R.PlotUnit(1, W, H)
    # asynchronous renders on two units, plotted without a host-side wait in between: the device-side events order it
    t0.render_async(scene, seed=4, stream=0, first_path_index=0)
    t1.render_async(scene, seed=4, stream=1, first_path_index=0)
    a.plot([t0])
    b.plot([t1])
    want = [O.plot(W, H, oscene.render(W, H, 4, s, 0, N, threads=4)[0]) for s in (0, 1)]
    xa, xb = a.tristimulus_buffer, b.tristimulus_buffer
    assert np.allclose(xa, want[0], rtol=2e-5, atol=1e-7) and np.allclose(xb, want[1], rtol=2e-5, atol=1e-7)
    a.add(b)                                            # rl_plot_unit_add: dst += src on one device
    assert a.tristimulus_buffer.tobytes() == (xa + xb).tobytes() and b.tristimulus_buffer.tobytes() == xb.tobytes()
    # unit re-use right after an asynchronous plot: the next render must wait for the plot that reads mapped_photons
    t0.render(scene, seed=4, stream=0, first_path_index=N)
    assert t0.mapped_photons.tobytes() == oscene.render(W, H, 4, 0, N, N, threads=4)[0].tobytes()
    with pytest.raises(R.RlError):
        a.add(a)


def test_rccl_exchange_through_the_c_abi_with_one_rank(R):
    """The library's own RCCL binding (dlopen, ncclCommInitAll / ncclCommInitRank, ncclReduce on the plot stream) on
    the one GPU this box has: with a single rank the reduce leaves the buffer as it is, and rl_gather_unit_allreduce
    equals accumulate.  (Two ranks need two GPUs; ranks sharing a GPU use rl_plot_unit_add, tested above.)"""
    W, H, N = 64, 36, 1 << 13
    objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
    scene = R.Scene(objs, cam)
    (comm,) = R.Comm.init_all([0])
    assert (comm.rank, comm.world) == (0, 1)
    t, p, g, g2 = R.TraceUnit(0, W, H, n_photons=N), R.PlotUnit(0, W, H), R.GatherUnit(W, H), R.GatherUnit(W, H)
    t.render_fused(scene, p, N, seed=2, stream=0, first_path_index=0)
    t.sync()
    before = p.tristimulus_buffer
    assert before.any()
    p.reduce(comm, root=0)
    assert p.tristimulus_buffer.tobytes() == before.tobytes()
    R.gather_allreduce(g, p, comm)
    assert g.tristimulus_buffer.tobytes() == before.tobytes() and not p.tristimulus_buffer.any()
    # multi-process style: id from rank 0, rl_comm_init_rank
    comm2 = R.Comm(R.Comm.unique_id(), 1, 0, 0)
    p.upload(before)
    R.gather_allreduce(g2, p, comm2)
    assert g2.tristimulus_buffer.tobytes() == before.tobytes()
    with pytest.raises(R.RlError):
        R.Comm.init_all([0, 0])       # RCCL admits one rank per device
    # what the exchange ran on and what it cost (bench.py's config.rccl / exchange): the communicator's own count, the
    # RCCL version, the library that was loaded; every reduce timed on the plot stream, unread timings bounded
    info = comm.info()
    assert info["rank"] == 0 and info["world"] == 1 and info["rccl_version"] > 20000 and "rccl" in info["library"]
    n0, ms0 = p.exchange_stats()
    assert n0 == 3 and ms0 > 0.0      # reduce + two allreduces so far
    for _ in range(70):               # more than the library keeps unread: the oldest are folded into the totals
        p.reduce(comm, root=0)
    n1, ms1 = p.exchange_stats()
    assert n1 == 73 and ms1 > ms0 and (ms1 - ms0) / 70 < 5.0
    assert len(R.device_pci_bus_id(0).split(":")) == 3


@pytest.mark.parametrize("fused", [False, True])
def test_app_with_two_ranks_on_one_gpu_renders_the_sum_of_two_streams(R, fused):
    """rl_app_run with devices = [0, 0]: every scheduler unit is one unit per rank, rank r renders RNG stream r, and
    Task::Gather sums the ranks' plot buffers onto rank 0 before the Kahan accumulation -- the single-process form of
    the 8-GPU layout (distinct devices take the ncclReduce branch of the same code)."""
    W, H, n, batches = 96, 54, 1 << 14, 10
    rgb, st = R.app_run(W, H, batches, concurrency=3, photons_per_batch=n, seed=5, fused=fused, devices=[0, 0])
    assert st["batches"] == batches and st["paths"] == 2 * batches * n
    objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
    oscene = O.Scene(objs.view(O.OBJECT_DTYPE), _ocam(cam))
    xyz = np.zeros((W * H, 3), np.float32)
    segs = 0
    for stream in (0, 1):
        photons, s = oscene.render(W, H, 5, stream, 0, batches * n, threads=8)
        O.plot(W, H, photons, xyz)
        segs += s
    assert st["segments"] == segs
    want_rgb, _, _ = O.tonemap(xyz, W, H)
    assert np.abs(rgb.reshape(-1, 3).astype(int) - want_rgb.astype(int)).max() <= 1


@pytest.mark.parametrize("fused", [False, True])
def test_resumed_run_adds_new_samples(R, fused, tmp_path):
    """resume used to start at path 0 again and add the same samples twice.  Two runs of N batches, the
    second resumed from the first's checkpoint, must equal one run of 2N batches."""
    W, H, n, N = 80, 45, 1 << 13, 6
    raw = str(tmp_path / "buffer.raw")
    kw = dict(concurrency=2, photons_per_batch=n, seed=9, fused=fused)
    _, st1 = R.app_run(W, H, N, checkpoint=raw, **kw)
    assert st1["next_batch"] == N and open(raw + ".next").read().startswith("next_batch %d\n" % N)
    rgb2, st2 = R.app_run(W, H, N, checkpoint=raw, resume=True, **kw)
    assert st2["batches"] == N and st2["next_batch"] == 2 * N
    rgb_once, st_once = R.app_run(W, H, 2 * N, **kw)
    assert st1["segments"] + st2["segments"] == st_once["segments"]      # the same 2N batches, none twice
    assert np.abs(rgb2.astype(int) - rgb_once.astype(int)).max() <= 1
    # the old behaviour, for contrast: N batches rendered twice is a different (brighter-noise) image
    rgb_twice, _ = R.app_run(W, H, N, first_batch=0, **kw)
    assert st_once["segments"] != 2 * st1["segments"]
    # first_batch by hand continues a run without a sidecar
    _, st3 = R.app_run(W, H, 2, first_batch=st2["next_batch"], **kw)
    assert st3["next_batch"] == 2 * N + 2
    assert rgb_twice.shape == rgb_once.shape


def test_scheduler_limit_is_reported_with_a_message(R):
    with pytest.raises(R.RlError) as e:
        R.app_run(64, 36, 1, concurrency=86)
    assert "RL_TASK_MAX_UNITS" in str(e.value)
    rgb, st = R.app_run(64, 36, 4, concurrency=24, photons_per_batch=4096)   # > 21 workers used to be refused
    assert st["batches"] == 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _two_rank_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import torch  # noqa: F401
    import robigo_luculenta_amd as R
    from robigo_luculenta_amd import distributed as D
    D.init_control_plane(rank, world)
    W, H, n, launches = 96, 54, 1 << 15, 3
    objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
    scene = R.Scene(objs, cam, device=0)
    trace, plot = R.TraceUnit(rank, W, H, n_photons=64, device=0), R.PlotUnit(rank, W, H, device=0)
    gather = R.GatherUnit(W, H, device=0) if rank == 0 else None
    for k in range(launches):                       # bench.py's step with the host-staged exchange
        trace.render_fused(scene, plot, n, seed=1, stream=rank, first_path_index=k * n)
        host = plot.tristimulus_buffer
        if D.host_staged_reduce(host, root=0):
            plot.upload(host)
            gather.accumulate(plot)
        else:
            plot.clear()
    if rank == 0:
        tm = R.TonemapUnit(W, H)
        tm.tonemap(gather)
        np.save(os.path.join(out_dir, "srgb.npy"), tm.srgb_float()[0])
    D.shutdown()


def test_two_ranks_on_one_gpu_reproduce_the_two_stream_image(R, tmp_path):
    """Two processes share GPU 0, render RNG streams 0 and 1 and exchange at every gather exactly as
    `bench.py --gpus 2 --dist-backend gloo` does; the image must be the oracle's two-stream image within 1e-3 sRGB."""
    import torch.multiprocessing as mp
    mp.spawn(_two_rank_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "srgb.npy")
    W, H, n, launches = 96, 54, 1 << 15, 3
    objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
    oscene = O.Scene(objs.view(O.OBJECT_DTYPE), _ocam(cam))
    acc, comp = np.zeros((W * H, 3), np.float32), np.zeros((W * H, 3), np.float32)
    for k in range(launches):
        xyz = np.zeros((W * H, 3), np.float32)
        for stream in (0, 1):
            O.plot(W, H, oscene.render(W, H, 1, stream, k * n, n, threads=8)[0], xyz)
        O.accumulate(acc, comp, xyz)
    _, want, _ = O.tonemap(acc, W, H)
    assert np.abs(got - want).max() <= 1e-3


def test_bench_runs_its_own_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` without a launcher spawns its ranks; with --dist-backend gloo they share GPU 0."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1",
           "--launches-per-step", "1", "--batches-per-launch", "4", "--config", "demo-720p"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run(cmd, env=env, capture_output=True, timeout=600, check=True).stdout.decode()
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["config"]["dist_backend"] == "gloo"
    assert line["mpaths_per_s"] > 0 and abs(line["segments_per_path"] - 3.54) < 0.1
    assert line["value"] > 0 and "cpu_baseline" not in line
    # the line carries the one-GPU rate of the same run and both scaling modes (two ranks SHARING a GPU: efficiency ~ 0.5 or less)
    sd = line["scaling_detail"]
    assert sd["n1_same_run"]["value"] > 0 and sd["weak"]["headline"] and not sd["strong"]["headline"]
    assert 0.05 < sd["weak"]["efficiency"] < 0.8 and 0.02 < sd["strong"]["efficiency"] < 0.8
    assert sd["strong"]["paths_per_step_per_gpu"] == 2 * 524288


def test_bench_line_counts_its_own_counters_when_rocprofv3_is_on_the_box():
    """roofline.executed_live -- lane use and cycles per instruction counted in the bench's own run (bench.py
    re-runs itself under rocprofv3 --pmc after the timed region), no committed file involved."""
    import shutil
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("no rocprofv3")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--launches-per-step", "1", "--batches-per-launch", "64",
           "--no-cpu-baseline", "--no-others"]
    out = subprocess.run(cmd, capture_output=True, timeout=600, check=True).stdout.decode()
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    live = line["roofline"]["executed_live"]
    assert live and not live.get("skipped"), live
    assert live["build_id"] == line["config"]["build_id"]
    assert 1500 < live["valu_insts_per_64ray_segment"] < 4000 and 2.0 < live["cycles_per_valu_inst_per_simd"] < 5.0
    assert 0.6 < live["active_lanes"] < 1.0 and 0.3 < live["useful_lane_slots_vs_2cyc"] < 0.8
    assert abs(sum(live["wave_time"].values()) - 1.0) < 0.15
    assert live["seconds"] < 100
    # the occupancy the design leans on, from the counters: a resident trace kernel holds 4 waves per SIMD (the
    # compiler reports 118-120 VGPRs, rocprofv3 "VGPR_Count 60" -- the unified register file counted in halves)
    assert 3.5 < live["resident_waves_per_simd"] <= 5.02
    # the driver-parsed roofline.frac IS the utilisation counted in this run (<= 1), achieved / peak say the same in
    # lane-operations per second, and the reference-flops figure (> 1: the kernel culls the reference's scan) is labelled a speed-up
    rf = line["roofline"]
    assert rf["frac"] == live["useful_lane_slots_vs_2cyc"] and 0.3 < rf["frac"] <= 1.0 and rf["frac_source"].startswith("executed_live")
    assert abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-9 and rf["unit"] == "Tlane-op/s"
    assert rf["algorithmic"]["frac"] == rf["frac_algorithmic"] and "SPEED-UP" in rf["algorithmic"]["is"]
    assert 0.5 < rf["issue_ceiling"]["kernel_vs_plain_stream"] < 1.3


def test_bench_with_more_ranks_than_gpus_falls_back_instead_of_hanging(R):
    """The default exchange is RCCL, one rank per GPU.  On a box with fewer GPUs than ranks a rank has no device of its own
    and never reaches ncclCommInitRank -- a collective the others would wait in for ever; the ranks must agree on the
    host-staged exchange BEFORE anybody enters it, and the line must say so."""
    if R.device_count() >= 2:
        pytest.skip("needs fewer GPUs than ranks")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--launches-per-step", "1", "--batches-per-launch", "4", "--config", "demo-720p"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run(cmd, env=env, capture_output=True, timeout=600, check=True).stdout.decode()
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["dist_backend"].startswith("gloo (fallback")
    assert line["value"] > 0


def test_concurrent_renders_share_open_launches_and_stay_bit_exact(R):
    """The reference runs one TraceUnit::render per worker thread (app.rs:92-134).  The library appends such calls to
    a trace kernel that is already running (open launches, rl_api.hip); every unit must still receive exactly its own
    paths: photons bit-equal to the oracle's for that unit's path range, whatever shared a launch with whatever."""
    import threading
    W, H = 160, 90
    objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
    scene = R.Scene(objs, cam)
    oscene = O.Scene(objs.view(O.OBJECT_DTYPE), _ocam(cam))
    for n, workers in ((1 << 12, 8), (1000, 5)):        # 1000 is not a multiple of the refill size: launched alone
        units = [R.TraceUnit(i, W, H, n_photons=n) for i in range(workers)]
        start = threading.Barrier(workers)
        results, errors = {}, []

        def work(i):
            try:
                for rnd in range(3):
                    first = (rnd * workers + (workers - 1 - i)) * n + 7 * rnd      # not in unit order, not contiguous
                    start.wait()
                    units[i].render(scene, seed=6, stream=2, first_path_index=first)
                    results[(i, rnd)] = (first, units[i].mapped_photons)
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=work, args=(i,)) for i in range(workers)]
        [t.start() for t in threads]
        [t.join() for t in threads]
        assert not errors, errors
        segments = [0] * workers
        for (i, rnd), (first, got) in results.items():
            want, segs = oscene.render(W, H, 6, 2, first, n, threads=4)
            assert got.tobytes() == want.tobytes(), (n, i, rnd)
            segments[i] += segs
        stats = [u.stats() for u in units]       # paths and segments are counted per call, so per unit
        assert [s[0] for s in stats] == [3 * n] * workers and [s[1] for s in stats] == segments
        assert sum(s[2] for s in stats) > 0      # kernel time: credited to whichever unit asked first


def test_concurrent_fused_renders_share_open_launches_and_hit_their_own_plot_units(R):
    """rl_trace_unit_render_fused_sync from several threads: the kernel they share splats into every caller's own plot
    unit (job table in the kernel); each buffer must equal the oracle's plot of that caller's path range."""
    import threading
    W, H, workers = 96, 54, 6
    objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
    scene = R.Scene(objs, cam)
    oscene = O.Scene(objs.view(O.OBJECT_DTYPE), _ocam(cam))
    sizes = [1 << 13, 3 << 12, 1 << 13, 5 << 11, 1 << 12, 1000]       # the last is not a multiple of 64: launched alone
    units = [R.TraceUnit(i, W, H, n_photons=64) for i in range(workers)]
    plots = [R.PlotUnit(i, W, H) for i in range(workers)]
    firsts = [100000 * (workers - i) + 17 for i in range(workers)]
    start = threading.Barrier(workers)
    before = R.batch_histogram()
    errors = []

    def work(i):
        try:
            start.wait()
            units[i].render_fused_sync(scene, plots[i], sizes[i], seed=8, stream=1, first_path_index=firsts[i])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(workers)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    segs = 0
    for i in range(workers):
        photons, s = oscene.render(W, H, 8, 1, firsts[i], sizes[i], threads=4)
        segs += s
        want = O.plot(W, H, photons)
        assert np.allclose(plots[i].tristimulus_buffer, want, rtol=2e-5, atol=1e-7), i
    assert sum(u.stats()[1] for u in units) == segs and [u.stats()[0] for u in units] == sizes
    after = R.batch_histogram()
    assert sum(k * (after.get(k, 0) - before.get(k, 0)) for k in after) == workers - 1   # all but the ragged one were appended


def test_an_open_launch_rolls_over_when_its_job_table_is_full(R):
    """More blocking calls than one kernel's job table holds (256), from four threads without a pause: the launch that
    is full closes, the next call starts another one, nothing is lost or traced twice."""
    import threading
    W, H, n, workers, rounds = 64, 36, 256, 4, 120
    objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
    scene = R.Scene(objs, cam)
    oscene = O.Scene(objs.view(O.OBJECT_DTYPE), _ocam(cam))
    units = [R.TraceUnit(i, W, H, n_photons=n) for i in range(workers)]
    kept, errors = {}, []
    before = R.batch_histogram()

    def work(i):
        try:
            for rnd in range(rounds):
                first = (rnd * workers + i) * n
                units[i].render(scene, seed=11, stream=0, first_path_index=first)
                if rnd % 17 == 0 or rnd == rounds - 1:
                    kept[(i, rnd)] = (first, units[i].mapped_photons)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(workers)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    for (i, rnd), (first, got) in kept.items():
        want, _ = oscene.render(W, H, 11, 0, first, n, threads=2)
        assert got.tobytes() == want.tobytes(), (i, rnd)
    _, segs = oscene.render(W, H, 11, 0, 0, workers * rounds * n, threads=8)     # the calls tile [0, workers * rounds * n)
    stats = [u.stats() for u in units]
    assert [s[0] for s in stats] == [rounds * n] * workers and sum(s[1] for s in stats) == segs
    after = R.batch_histogram()
    carried = {k: after.get(k, 0) - before.get(k, 0) for k in after if after.get(k, 0) != before.get(k, 0)}
    assert sum(k * v for k, v in carried.items()) == workers * rounds and max(carried) <= 256


def test_calls_with_different_parameters_get_launches_of_their_own(R):
    """Two scenes, two seeds, two image sizes at the same time: only calls that agree in all of them may share a kernel
    (it holds one scene and one RlTraceJob); each result is still the oracle's."""
    import threading
    n = 1 << 12
    cases = []
    for k, (which, seed, (W, H)) in enumerate([(R.SCENE_DEMO, 3, (64, 36)), (R.SCENE_GLASS_STRESS, 3, (64, 36)), (R.SCENE_DEMO, 4, (64, 36)),
                                               (R.SCENE_DEMO, 3, (80, 45)), (R.SCENE_GLASS_STRESS, 9, (48, 27)), (R.SCENE_DEMO, 3, (64, 36))]):
        objs, cam = R.builtin_scene_desc(which)
        cases.append((objs, cam, R.Scene(objs, cam), seed, W, H, R.TraceUnit(k, W, H, n_photons=n)))
    got, errors = {}, []

    def work(k):
        try:
            for rnd in range(4):
                _, _, scene, seed, W, H, unit = cases[k]
                unit.render(scene, seed=seed, stream=1, first_path_index=rnd * n + k)
                got[(k, rnd)] = unit.mapped_photons
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(cases))]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    for (k, rnd), photons in got.items():
        objs, cam, _, seed, W, H, _ = cases[k]
        want, _ = O.Scene(objs.view(O.OBJECT_DTYPE), _ocam(cam)).render(W, H, seed, 1, rnd * n + k, n, threads=2)
        assert photons.tobytes() == want.tobytes(), (k, rnd)


def test_small_kernels_run_beside_a_resident_trace_kernel(R):
    """An open launch keeps every CU occupied for as long as calls keep coming; PlotUnit::plot, GatherUnit::accumulate
    and the clears must not queue up behind it.  They run beside it because the trace kernel leaves them registers
    (<= 120 of the 128 VGPRs a wave may have at four waves per SIMD) and has a hardware queue of its own."""
    import threading
    import time
    W, H = 1280, 720
    scene = R.Scene.builtin()
    big, other = R.TraceUnit(0, W, H), R.TraceUnit(1, W, H)
    pa, pb, g = R.PlotUnit(0, W, H), R.PlotUnit(1, W, H), R.GatherUnit(W, H)
    n = 64 * 524288                                          # ~8.5 ms of tracing
    big.render_fused_sync(scene, pa, n)
    other.render(scene)
    pb.plot([other]); pb.sync(); g.accumulate(pb); g.sync()  # warm-up
    best = None
    for _ in range(3):
        th = threading.Thread(target=lambda: big.render_fused_sync(scene, pa, n))
        t0 = time.perf_counter()
        th.start()
        time.sleep(0.002)
        pb.plot([other]); pb.sync(); g.accumulate(pb); g.sync()
        small_done = time.perf_counter() - t0
        th.join()
        trace_done = time.perf_counter() - t0
        best = (small_done, trace_done) if best is None or small_done / trace_done < best[0] / best[1] else best
    assert best[0] < 0.6 * best[1], best     # measured: done at 3.3 ms of 8.6 (2 ms of it the sleep above)


def test_render_in_two_halves_begin_on_many_units_then_end(R):
    """rl_trace_unit_render_begin / _end (what one host thread feeding several GPUs uses): calls begun back to back on
    several units are all in flight in one open launch; ended in any order they hold exactly what render() gives."""
    W, H, n = 96, 54, 1 << 12
    objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
    scene = R.Scene(objs, cam)
    oscene = O.Scene(objs.view(O.OBJECT_DTYPE), _ocam(cam))
    units = [R.TraceUnit(i, W, H, n_photons=n) for i in range(5)]
    plot = R.PlotUnit(0, W, H)
    fused = R.TraceUnit(9, W, H, n_photons=64)
    for i, u in enumerate(units):
        u.render_begin(scene, seed=4, stream=3, first_path_index=1000 * i)
    fused.render_fused_begin(scene, plot, 3 * n, seed=4, stream=3, first_path_index=77)
    plot2 = R.PlotUnit(1, W, H)
    fused.render_fused_begin(scene, plot2, n, seed=4, stream=3, first_path_index=9000)   # the ticket is the plot unit's: the trace unit is free
    with pytest.raises(R.RlError):
        units[0].render_begin(scene, seed=4, stream=3, first_path_index=0)       # one begun render per unit
    for u in reversed(units):
        u.render_end()
    units[2].render_end()                                                       # nothing begun: a no-op
    g = R.GatherUnit(W, H)
    plot3 = R.PlotUnit(2, W, H)
    fused.render_fused_begin(scene, plot3, n, seed=4, stream=3, first_path_index=9000)
    g.accumulate(plot3)                                                         # a consumer of the buffer ends the begun render
    assert np.allclose(g.tristimulus_buffer, plot2.tristimulus_buffer, rtol=2e-5, atol=1e-7)
    fused.sync()                                                                # the trace unit's sync ends what it began
    for i, u in enumerate(units):
        want, segs = oscene.render(W, H, 4, 3, 1000 * i, n, threads=2)
        assert u.mapped_photons.tobytes() == want.tobytes() and u.stats()[:2] == (n, segs)
    photons, segs = oscene.render(W, H, 4, 3, 77, 3 * n, threads=2)
    photons2, segs2 = oscene.render(W, H, 4, 3, 9000, n, threads=2)
    assert np.allclose(plot2.tristimulus_buffer, O.plot(W, H, photons2), rtol=2e-5, atol=1e-7)   # the download ends the begun render
    assert np.allclose(plot.tristimulus_buffer, O.plot(W, H, photons), rtol=2e-5, atol=1e-7)
    assert fused.stats()[:2] == (5 * n, segs + 2 * segs2)                       # plot, plot2 and plot3 (the same range as plot2)
    ragged = R.TraceUnit(10, W, H, n_photons=1000)                               # not a multiple of 64: a launch of its own
    ragged.render_begin(scene, seed=4, stream=3, first_path_index=5)
    ragged.render_end()
    want, _ = oscene.render(W, H, 4, 3, 5, 1000, threads=2)
    assert ragged.mapped_photons.tobytes() == want.tobytes()


@pytest.mark.parametrize("fused", [False, True])
def test_results_of_a_call_are_visible_the_moment_it_returns(R, fused):
    """The hazard of an open launch: a call returns while the kernel that produced its results is still running (the
    other threads keep it busy), and the very next thing the caller does is launch a kernel that reads them -- the plot
    of the photons, or the download of the splatted buffer.  Nothing may be missing: every (thread, round) is compared
    with the oracle's plot of exactly that path range."""
    import threading
    W, H, n, workers, rounds = 48, 27, 1 << 11, 4, 40
    objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
    scene = R.Scene(objs, cam)
    oscene = O.Scene(objs.view(O.OBJECT_DTYPE), _ocam(cam))
    units = [R.TraceUnit(i, W, H, n_photons=n) for i in range(workers)]
    plots = [R.PlotUnit(i, W, H) for i in range(workers)]
    got, errors = {}, []

    def work(i):
        try:
            for rnd in range(rounds):
                first = (rnd * workers + i) * n
                if fused:
                    units[i].render_fused_sync(scene, plots[i], n, seed=13, stream=0, first_path_index=first)
                else:
                    units[i].render(scene, seed=13, stream=0, first_path_index=first)
                    plots[i].plot([units[i]])
                got[(i, rnd)] = plots[i].tristimulus_buffer
                plots[i].clear()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(workers)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    for (i, rnd), xyz in sorted(got.items()):
        photons, _ = oscene.render(W, H, 13, 0, (rnd * workers + i) * n, n, threads=2)
        want = O.plot(W, H, photons)
        assert np.allclose(xyz, want, rtol=2e-5, atol=1e-7), (i, rnd, float(np.abs(xyz - want).max()))


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("blocking", [False, True])
def test_app_with_many_workers_traces_every_path_exactly_once(R, fused, blocking):
    """16 workers over 48 trace units and 8 plot units: renders are begun on one thread and ended on another (whoever
    plots / gathers next), trace units change hands while fused renders begun on them are still in flight.  Whatever
    the interleaving, the paths [0, batches * n) are traced exactly once: the segment total is the oracle's."""
    W, H, n, batches = 64, 36, 1 << 10, 600
    rgb, st = R.app_run(W, H, batches, concurrency=16, photons_per_batch=n, seed=9, fused=fused, blocking_trace=blocking)
    objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
    _, segs = O.Scene(objs.view(O.OBJECT_DTYPE), _ocam(cam)).render(W, H, 9, 0, 0, batches * n, threads=8)
    assert st["batches"] == batches and st["paths"] == batches * n and st["segments"] == segs
    assert rgb.any()


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("threads", [1, 3])
def test_app_pool_depth_is_not_the_host_thread_count(R, fused, threads):
    """VERDICT r05 #3a: RlAppConfig::threads.  The scheduler's pools are sized by `concurrency` (48 trace units, 8 plot units here:
    task_scheduler.rs:95-96) -- how many batches the DEVICE may have in flight -- while one or three host threads issue the tasks
    (app.rs:66 starts one thread per unit of depth; a GPU host has few cores per device).  Same budget, every path traced exactly
    once, and the image is the one the 16-thread pool renders up to the order of the float adds."""
    W, H, n, batches = 64, 36, 1 << 10, 600
    rgb, st = R.app_run(W, H, batches, concurrency=16, threads=threads, photons_per_batch=n, seed=9, fused=fused)
    objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
    _, segs = O.Scene(objs.view(O.OBJECT_DTYPE), _ocam(cam)).render(W, H, 9, 0, 0, batches * n, threads=8)
    assert st["batches"] == batches and st["paths"] == batches * n and st["segments"] == segs
    ref, _ = R.app_run(W, H, batches, concurrency=16, photons_per_batch=n, seed=9, fused=fused)
    assert np.abs(rgb.astype(int) - ref.astype(int)).max() <= 1


def test_a_fifth_combination_begun_without_ending_the_others_gets_a_launch_of_its_own(R):
    """a device has four open-launch slots, one per (scene, seed, stream, size, fetch, fused) combination.
    Renders begun for more combinations than that -- none ended yet, all on ONE thread -- used to spin for ever in the
    fifth begin; now the fifth and sixth get plain launches.  Every result is still the oracle's."""
    W, H, n = 64, 36, 1 << 12
    objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
    scene = R.Scene(objs, cam)
    oscene = O.Scene(objs.view(O.OBJECT_DTYPE), _ocam(cam))
    units = [R.TraceUnit(i, W, H, n_photons=n) for i in range(6)]
    done = []

    def begin_all():
        for i, u in enumerate(units):
            u.render_begin(scene, seed=5, stream=i, first_path_index=100 * i)   # six streams = six combinations
        done.append(True)

    import threading
    t = threading.Thread(target=begin_all, daemon=True)
    t.start()
    t.join(timeout=60)
    assert done, "render_begin for a fifth combination did not return"
    for i, u in reversed(list(enumerate(units))):
        u.render_end()
        want, segs = oscene.render(W, H, 5, i, 100 * i, n, threads=2)
        assert u.mapped_photons.tobytes() == want.tobytes() and u.stats()[:2] == (n, segs), i


def test_a_fused_render_waits_for_what_is_queued_on_its_plot_units_stream(R):
    """rl_plot_unit_add leaves work on the destination's plot stream; a fused render begun into that buffer
    right behind it (no clear, no accumulate in between) must splat AFTER the add, for plain and open launches."""
    W, H, n = 96, 54, 1 << 14
    objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
    scene = R.Scene(objs, cam)
    oscene = O.Scene(objs.view(O.OBJECT_DTYPE), _ocam(cam))
    a, b = O.plot(W, H, oscene.render(W, H, 2, 0, 0, n, threads=2)[0]), O.plot(W, H, oscene.render(W, H, 2, 0, n, n, threads=2)[0])
    tr = R.TraceUnit(0, W, H, n_photons=64)
    for sync_call in (False, True):
        src, dst = R.PlotUnit(0, W, H), R.PlotUnit(1, W, H)
        tr.render_fused(scene, src, n, seed=2, stream=0, first_path_index=0)
        for _ in range(20):
            dst.add(src)                                   # 20 x a on the plot stream
            if sync_call:
                tr.render_fused_sync(scene, dst, n, seed=2, stream=0, first_path_index=n)   # open launch
            else:
                tr.render_fused(scene, dst, n, seed=2, stream=0, first_path_index=n)        # plain launch
        assert np.allclose(dst.tristimulus_buffer, 20.0 * (a + b), rtol=1e-4, atol=1e-6)


def test_resume_converts_the_index_when_the_batch_size_changed_and_refuses_overlapping_streams(R, tmp_path):
    """the sidecar's index counts batches of the size it was written with.  A resume with half the batch size
    starts behind the same PATH index (not at the same batch number, which would repeat half the samples); a resume
    whose RNG streams only partly overlap the checkpoint's is refused; another seed has nothing to continue from."""
    W, H, n, N = 80, 45, 1 << 13, 6
    raw = str(tmp_path / "buffer.raw")
    kw = dict(concurrency=2, seed=9, fused=True)
    _, st1 = R.app_run(W, H, N, checkpoint=raw, photons_per_batch=n, **kw)
    assert not os.path.exists(raw + ".tmp") and not os.path.exists(raw + ".next.tmp")      # both files were renamed into place
    assert open(raw + ".next").read().split() == ["next_batch", str(N), "photons_per_batch", str(n), "seed", "9", "stream", "0", "ranks", "1"]
    rgb2, st2 = R.app_run(W, H, 2 * N, checkpoint=raw, resume=True, photons_per_batch=n // 2, **kw)
    assert st2["next_batch"] == 2 * N + 2 * N                                              # started at batch 2N of the smaller size
    rgb_once, st_once = R.app_run(W, H, 2 * N, photons_per_batch=n, **kw)
    assert st1["segments"] + st2["segments"] == st_once["segments"]                        # the same paths, none twice
    assert np.abs(rgb2.astype(int) - rgb_once.astype(int)).max() <= 1
    with pytest.raises(R.RlError) as e:
        R.app_run(W, H, 2, checkpoint=raw, resume=True, photons_per_batch=n, devices=[0, 0], concurrency=4, seed=9, fused=True)
    assert "streams" in str(e.value)
    _, st4 = R.app_run(W, H, 2, checkpoint=raw, resume=True, photons_per_batch=n, concurrency=2, seed=10, fused=True)
    assert st4["next_batch"] == 2                                                          # a new seed: every sample is new
    # ... and the index still lists seed 9's samples (the buffer still holds them), so a later resume with seed 9
    # continues behind them instead of adding them a second time
    words = open(raw + ".next").read().split()
    assert words[:10] == ["next_batch", "2", "photons_per_batch", str(n), "seed", "10", "stream", "0", "ranks", "1"]
    assert words[10:] == ["next_batch", str(4 * N), "photons_per_batch", str(n // 2), "seed", "9", "stream", "0", "ranks", "1"]
    _, st5 = R.app_run(W, H, 1, checkpoint=raw, resume=True, photons_per_batch=n, concurrency=2, seed=9, fused=True)
    assert st5["next_batch"] == 2 * N + 1                                                  # behind seed 9's 2N batches of n paths
    words = open(raw + ".next").read().split()
    assert words[:6] == ["next_batch", str(2 * N + 1), "photons_per_batch", str(n), "seed", "9"] and words[10:16] == ["next_batch", "2", "photons_per_batch", str(n), "seed", "10"]

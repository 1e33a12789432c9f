"""GPU parity tests proper: the HIP path through the C ABI against the CPU oracle.  Integer-like
outputs (per-photon records, Kahan buffers, exposure, tonemapped bytes) are compared BIT-EXACTLY;
only the atomically accumulated XYZ splat, whose summation order is not deterministic, gets a
float tolerance (stated at each assert)."""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu

import robigo_luculenta_amd as R  # a missing HIP library is a failure, never a skip


def _ocam(cam):
    return O.RlCameraDesc.from_buffer_copy(bytes(cam))


@pytest.fixture(scope="module")
def demo():
    objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
    return objs, cam, R.Scene(objs, cam), O.Scene(objs, _ocam(cam))


def test_device_present():
    assert R.device_count() >= 1


@pytest.mark.parametrize("fn,lo,hi", [("sin", -20, 20), ("cos", -20, 20), ("tan", 0.1, 1.4), ("exp", -100, 10),
                                      ("log", 1e-6, 100), ("acos", -0.999, 0.999), ("sin", -300, 300), ("cos", 25, 40), ("exp", -120, 95),
                                      ("acos", -1, 1), ("sin_d", -20, 20), ("cos_d", -300, 300), ("exp_d", -100, 10), ("acos_d", -1, 1)])
def test_math_header_bit_exact_on_device(fn, lo, hi):
    rng = np.random.default_rng(7)
    x = rng.uniform(lo, hi, 1 << 16).astype(np.float32)
    got = R.math_probe(fn, x)
    want = O.math_f32(fn, x)
    assert got.tobytes() == want.tobytes()


def test_roulette_fast_path_decides_like_the_exact_form():
    """trace_unit.rs:122-125.  The kernel decides `rand * 0.85 > continue_chance * (1 - exp(-20 intensity))`
    from the hardware exp2 when the two sides are more than 2e-5 apart (rl_roulette_ends) and from the exact
    f64 exp otherwise; the decision must always be the exact form's.  Whole waves of 64 share one distance
    from the threshold, so waves just outside the band take the fast path; out-of-range intensities
    (> 1, negative, NaN, inf) must fall back to the exact form."""
    rng = np.random.default_rng(11)
    m = 1 << 18
    intensity = rng.uniform(0, 1, m).astype(np.float32) ** 2
    intensity[::7] = rng.uniform(0.0, 0.05, len(intensity[::7])).astype(np.float32)   # where 1 - e is small
    cc = (0.96 ** rng.integers(1, 40, m)).astype(np.float32)
    e = O.math_f32("exp", intensity * np.float32(-20.0))
    threshold = cc * (np.float32(1.0) - e)
    deltas = np.array([0, 1e-7, 1e-6, 5e-6, 1.5e-5, 1.9e-5, 2.1e-5, 2.5e-5, 3e-5, 1e-4, 1e-3, 1e-1], np.float32)
    delta = np.repeat(deltas[rng.integers(0, len(deltas), m // 64)], 64) * rng.choice([-1, 1], m).astype(np.float32)
    unit = np.clip((threshold + delta) / np.float32(0.85), 0, 1).astype(np.float32)
    odd = rng.integers(0, m, 512)                       # out-of-range intensities in some waves
    intensity[odd] = np.resize(np.array([1.5, -0.25, np.nan, np.inf, -np.inf, 1.0000001], np.float32), len(odd))
    e = O.math_f32("exp", intensity * np.float32(-20.0))
    with np.errstate(invalid="ignore", over="ignore"):
        want = (unit * np.float32(0.85) > cc * (np.float32(1.0) - e)).astype(np.float32)
    got = R.math_probe("roulette", np.concatenate([unit, cc, intensity]))[:m]
    assert np.array_equal(got, want)
    assert 0.2 < want.mean() < 0.8                     # both outcomes occur


def test_ieee_sqrt_div_and_f64_islands_on_device():
    rng = np.random.default_rng(8)
    x = np.exp(rng.uniform(-30, 30, 1 << 16)).astype(np.float32)
    assert R.math_probe("sqrt", x).tobytes() == np.sqrt(x).tobytes()
    assert R.math_probe("div", x).tobytes() == (x / np.roll(x, -1)).tobytes()
    lam = rng.uniform(380, 780, 1 << 16).astype(np.float32)
    want = np.array([O.lib().oracle_sf10_ior(float(v)) for v in lam[:4096]], dtype=np.float32)
    assert R.math_probe("sf10", lam[:4096]).tobytes() == want.tobytes()
    g = rng.uniform(0.0031308, 4.0, 1 << 14).astype(np.float32)
    wantg = np.zeros_like(g)
    O.lib().oracle_powf(O.ptr(g), C.c_float(np.float32(1.0) / np.float32(2.4)), O.ptr(wantg), g.size)
    assert R.math_probe("gamma", g).tobytes() == wantg.tobytes()


def test_sf10_index_decided_from_the_cheap_evaluation_is_the_reference_f64_expression_for_every_wavelength():
    """material.rs:203-213 in f64, rounded to f32.  The kernel decides that f32 from a cheaper evaluation (one division, Newton
    steps on the hardware's reciprocal / inverse square root) whenever everything within 2^-40 of it rounds to the same float,
    and evaluates the reference's expression otherwise (rl_sf10_ior, rl_core.h): the result must be the oracle's for EVERY f32
    wavelength of the visible range [380, 780] (8.7 M values) and for a sweep of others, zero, negatives, infinities and NaN included."""
    lo, hi = np.float32(380.0).view(np.uint32), np.float32(780.0).view(np.uint32)
    lam = np.arange(int(lo), int(hi) + 1, dtype=np.uint32).view(np.float32)
    assert lam.size > 8_000_000 and lam[0] == 380.0 and lam[-1] == 780.0
    odd = np.concatenate([np.exp(np.linspace(-20, 20, 200001)).astype(np.float32), -np.linspace(0, 2000, 4001).astype(np.float32),
                          np.array([0.0, 114.84, 114.85, 249.6, 249.62, 12459.38, 12459.4, np.inf, -np.inf, np.nan], np.float32)])
    for x in (lam, odd):
        got = R.math_probe("sf10", x)
        with np.errstate(all="ignore"):
            want = O.math_f32("sf10", x)
        assert got.view(np.uint32).tobytes() == want.view(np.uint32).tobytes()


def _exhaustive_slices(lo, hi, step):
    """The slices of [lo, hi) that go through the HOST comparison (the library's probe against numpy).  The exhaustive part of these
    tests runs on the device (R.math_sweep: every argument, every run, seconds); the host comparison pins what the device compares
    against -- the compiler's IEEE expansion -- on a FIXED sample: every eighth slice from phase RL_SLICE_PHASE (default 0) plus the
    first and the last, so the same commit tests the same inputs on every day (rounds 4-5 rotated the phase with the
    date).  RL_EXHAUSTIVE=1: all slices through the host as well (minutes, many GB of host traffic)."""
    firsts = list(range(lo, hi, step))
    if os.environ.get("RL_EXHAUSTIVE"):
        return firsts
    phase = int(os.environ.get("RL_SLICE_PHASE", "0")) % 8
    return sorted(set(firsts[phase::8]) | {firsts[0], firsts[-1]})


def _device_sweep(fn, lo, hi, both_signs):
    """Every float with bits in [lo, hi) (and its negative): the library's short form against the compiler's correctly rounded
    expansion, on the device.  Returns the number of arguments compared."""
    bad, seen, example = R.math_sweep(fn, lo, hi, both_signs)
    assert bad == 0, "%s: %d of %d arguments differ from the IEEE result, e.g. bits 0x%08x" % (fn, bad, seen, example)
    assert seen == (hi - lo) * (2 if both_signs else 1)
    return seen


def test_short_square_root_is_the_ieee_one_for_every_normal_float():
    """rl_sqrtf (rl_core.h): y = v_rsq_f32(x), s = x y, s + (x - s s) y / 2 -- four operations after the hardware's inverse
    square root -- in place of the compiler's 16-instruction correctly rounded expansion, for waves whose arguments are all
    normal floats >= 2^-96.  Proven by exhaustion, not argued: every one of the 1,879,048,192 such floats through the library's
    probe against numpy's IEEE square root, in slices; waves that hold anything else (zero, denormals, tiny normals, infinity,
    NaN, negatives) must take the compiler's form and give the IEEE result too."""
    lo, hi = 0x0f800000, 0x7f800000
    step = 1 << 25
    assert _device_sweep("sqrt_short", lo, hi, False) == 1_879_048_192   # all of them, on the device, against the compiler's sqrtf
    assert _device_sweep("sqrt_short", 0x00800000, lo, False) > 0        # the tiny normals below the short form's range: its fallback
    for first in _exhaustive_slices(lo, hi, step):                        # ... and a fixed sample against numpy, through the host
        x = np.arange(first, min(first + step, hi), dtype=np.uint32).view(np.float32)
        got = R.math_probe("sqrt_short", x)
        assert got.view(np.uint32).tobytes() == np.sqrt(x).view(np.uint32).tobytes(), "slice 0x%08x (RL_SLICE_PHASE=%s)" % (first, os.environ.get("RL_SLICE_PHASE", "0"))
    rng = np.random.default_rng(5)
    odd = rng.integers(0, 1 << 32, 1 << 20, dtype=np.uint64).astype(np.uint32).view(np.float32)   # every class of float, mixed within waves
    odd[::3] = np.array([0.0, -0.0, np.inf, 1e-40, 3e-30, 1.0], np.float32)[rng.integers(0, 6, len(odd[::3]))]
    with np.errstate(invalid="ignore"):
        want = np.sqrt(odd)
    got = R.math_probe("sqrt_short", odd)
    both_nan = np.isnan(got) & np.isnan(want)
    assert np.array_equal(got.view(np.uint32)[~both_nan], want.view(np.uint32)[~both_nan])


@pytest.mark.parametrize("fn", ["recip_short", "div200_short"])
def test_short_one_operand_divisions_are_the_ieee_ones(fn):
    """rl_recipf / rl_div200f (rl_core.h): 1 / x and x / 200 in three operations for waves whose arguments are normal floats with
    2^-100 <= |x| < 2^100.  tools/sqrt_exhaustive.hip compares every such float of either sign on the device
    (profiles/r04_sqrt_exhaustive.txt: 0 of 3.36 G differ); here: every positive one through the library's probe against
    numpy's IEEE division, negative ones and mixed waves (zeros, denormals, huge, infinities, NaN: the compiler's form) sampled."""
    ref = (lambda x: np.float32(1.0) / x) if fn == "recip_short" else (lambda x: x / np.float32(200.0))
    lo, hi = 0x0d800000, 0x71800000
    step = 1 << 25
    assert _device_sweep(fn, lo, hi, True) == 3_355_443_200               # every such float of either sign, on the device
    assert _device_sweep(fn, 0x00800000, lo, True) > 0 and _device_sweep(fn, hi, 0x7f800000, True) > 0   # outside the range: the fallback
    for first in _exhaustive_slices(lo, hi, step):                        # ... and a fixed sample against numpy, through the host
        x = np.arange(first, min(first + step, hi), dtype=np.uint32).view(np.float32)
        assert R.math_probe(fn, x).view(np.uint32).tobytes() == ref(x).view(np.uint32).tobytes(), "slice 0x%08x (RL_SLICE_PHASE=%s)" % (first, os.environ.get("RL_SLICE_PHASE", "0"))
    rng = np.random.default_rng(6)
    neg = (rng.integers(lo, hi, 1 << 22, dtype=np.uint64).astype(np.uint32) | np.uint32(0x80000000)).view(np.float32)
    assert R.math_probe(fn, neg).view(np.uint32).tobytes() == ref(neg).view(np.uint32).tobytes()
    odd = rng.integers(0, 1 << 32, 1 << 20, dtype=np.uint64).astype(np.uint32).view(np.float32)
    odd[::3] = np.array([0.0, -0.0, np.inf, -np.inf, 1e-40, 3e37], np.float32)[rng.integers(0, 6, len(odd[::3]))]
    with np.errstate(all="ignore"):
        want = ref(odd)
    got = R.math_probe(fn, odd)
    both_nan = np.isnan(got) & np.isnan(want)
    assert np.array_equal(got.view(np.uint32)[~both_nan], want.view(np.uint32)[~both_nan])


def test_normalise_with_the_shared_reciprocal_is_the_ieee_division_bit_for_bit():
    """rl_normalise (vector3.rs:56-67) divides three components by one length; on the device the reciprocal's refinement is
    shared and the scaling / fix-up steps of the compiler's division are skipped where they pass their operands through
    (rl_core.h).  Against numpy's correctly rounded sqrt and division: ordinary vectors (the shortcut), vectors with zero,
    negative-zero, tiny and huge components, zero and non-finite vectors (waves that fall back to the plain divisions)."""
    rng = np.random.default_rng(12)

    def ref(v):
        v = v.astype(np.float32)
        m = np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]).astype(np.float32)   # rl_dot's order
        with np.errstate(all="ignore"):
            u = (v / m).astype(np.float32)
        return np.where(m == 0, v, u)

    def check(v):
        got = R.math_probe("normalise", v.reshape(-1)).reshape(3, -1)
        want = ref(v)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (v[:, ~same.all(0)][:, :4], got[:, ~same.all(0)][:, :4], want[:, ~same.all(0)][:, :4])

    n = 1 << 20
    check(rng.normal(0, 1, (3, n)).astype(np.float32))                                          # directions
    check((rng.normal(0, 1, (3, n)) * np.exp(rng.uniform(-40, 40, n))).astype(np.float32))      # any length the shortcut takes
    v = rng.normal(0, 1, (3, n)).astype(np.float32)
    v[rng.integers(0, 3, n), np.arange(n)] = 0.0                                                # axis-aligned normals: exact zeros
    v[rng.integers(0, 3, n), np.arange(n)] *= np.float32(-1.0)                                  # ... of either sign
    check(v)
    edge = (rng.normal(0, 1, (3, n)) * np.exp(rng.uniform(-100, 88, (3, n)))).astype(np.float32)  # components tiny against the length,
    edge[:, ::7] = 0.0                                                                            # lengths beyond 2^60, zero vectors
    edge[0, ::11] = np.float32(1e-42)                                                             # denormal components
    edge[1, ::13] = np.inf
    check(edge)


@pytest.mark.parametrize("fetch", [R.FETCH_LDS, R.FETCH_GLOBAL])
def test_trace_photons_bit_exact_demo(demo, fetch):
    objs, cam, scene, oscene = demo
    W, H, N = 1280, 720, 1 << 16
    t = R.TraceUnit(0, W, H, n_photons=N)
    t.set_fetch(fetch)
    for seed, stream, first in [(1, 0, 0), (2, 3, 5_000_000_000)]:
        t.render(scene, seed=seed, stream=stream, first_path_index=first)
        got = t.mapped_photons
        want, segs = oscene.render(W, H, seed, stream, first, N, threads=8)
        assert got.tobytes() == want.tobytes()
    paths, segments, ms = t.stats()
    assert paths == 2 * N


def test_trace_photons_bit_exact_glass_and_replicated():
    W, H, N = 640, 360, 1 << 15
    for which, param in [(R.SCENE_GLASS_STRESS, 0), (R.SCENE_DEMO, 158)]:
        objs, cam = R.builtin_scene_desc(which, param)
        scene, oscene = R.Scene(objs, cam), O.Scene(objs, _ocam(cam))
        for fetch in (R.FETCH_LDS, R.FETCH_GLOBAL):
            t = R.TraceUnit(0, W, H, n_photons=N)
            t.set_fetch(fetch)
            t.render(scene, seed=9, stream=1, first_path_index=123)
            want, segs = oscene.render(W, H, 9, 1, 123, N, threads=8)
            assert t.mapped_photons.tobytes() == want.tobytes()
            assert t.stats()[1] == segs


_MATRIX_SCENES = {}


@pytest.mark.parametrize("scene_name", ["demo", "glass"])           # demo: prisms without a second bound; glass stress: with (CYL)
@pytest.mark.parametrize("open_launch", [False, True], ids=["plain", "open"])
@pytest.mark.parametrize("fused", [False, True], ids=["unfused", "fused"])
@pytest.mark.parametrize("fetch", [R.FETCH_LDS, R.FETCH_GLOBAL], ids=["lds", "global"])
def test_parity_matrix_over_all_sixteen_instantiations(fetch, fused, open_launch, scene_name):
    """every instantiation rl_trace_kernel<fetch, fused, open, cyl> against the oracle, and
    rl_debug_variant_launches says that the instantiation meant is the one that ran.  Un-fused: MappedPhoton records byte
    for byte.  Fused: the unit's (paths, segments) counters equal the oracle's and the splatted XYZ buffer equals
    PlotUnit::plot of the oracle's photons up to the order of the float atomics (rtol 2e-5, as in
    test_plot_fused_and_unfused_match_oracle)."""
    if scene_name not in _MATRIX_SCENES:
        objs, cam = R.builtin_scene_desc(R.SCENE_DEMO if scene_name == "demo" else R.SCENE_GLASS_STRESS)
        _MATRIX_SCENES[scene_name] = (R.Scene(objs, cam), O.Scene(objs, _ocam(cam)), {})
    scene, oscene, cache = _MATRIX_SCENES[scene_name]
    W, H, N = 640, 360, 1 << 15                                   # a multiple of 64: blocking calls go through open launches
    seed, stream, first = 5, 2, 777
    if "want" not in cache:
        cache["want"] = oscene.render(W, H, seed, stream, first, N, threads=8)
    want, segs = cache["want"]
    t = R.TraceUnit(0, W, H, n_photons=N)
    t.set_fetch(fetch)
    before = R.variant_launches()
    if not fused:
        if open_launch:
            t.render(scene, seed=seed, stream=stream, first_path_index=first)          # blocking: appended to an open launch
        else:
            t.render_async(scene, seed=seed, stream=stream, first_path_index=first)    # one plain launch
            t.sync()
        assert t.mapped_photons.tobytes() == want.tobytes()
    else:
        p = R.PlotUnit(0, W, H)
        if open_launch:
            t.render_fused_sync(scene, p, N, seed=seed, stream=stream, first_path_index=first)
        else:
            t.render_fused(scene, p, N, seed=seed, stream=stream, first_path_index=first)
            t.sync()
        got = p.tristimulus_buffer
        ref = O.plot(W, H, want)
        assert np.allclose(got, ref, rtol=2e-5, atol=1e-6 * np.abs(ref).max())
        assert got.any()
    paths, segments, _ = t.stats()
    assert (paths, segments) == (N, segs)
    ran = [a - b for a, b in zip(R.variant_launches(), before)]
    meant = (8 if fetch == R.FETCH_LDS else 0) | (4 if fused else 0) | (2 if open_launch else 0) | (1 if scene_name == "glass" else 0)
    assert ran[meant] >= 1 and sum(ran) == ran[meant], (meant, ran)


def test_ragged_batch_sizes(demo):
    objs, cam, scene, oscene = demo
    for n in (1, 63, 65, 257, 1000):
        t = R.TraceUnit(0, 64, 36, n_photons=n)
        t.render(scene, seed=4, stream=0, first_path_index=77)
        want, _ = oscene.render(64, 36, 4, 0, 77, n)
        assert t.mapped_photons.tobytes() == want.tobytes()


def test_plot_fused_and_unfused_match_oracle(demo):
    objs, cam, scene, oscene = demo
    W, H, N = 320, 180, 1 << 18
    t = R.TraceUnit(0, W, H, n_photons=N)
    t.render(scene, seed=1, stream=0, first_path_index=0)
    photons = t.mapped_photons
    want = O.plot(W, H, photons)
    p = R.PlotUnit(0, W, H)
    p.plot([t])
    got = p.tristimulus_buffer
    # f32 atomics commute but do not associate: tolerance = a few ulps of the per-pixel sum.
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-6 * scale + 1e-6 * np.abs(want).max()
    assert np.allclose(got, want, rtol=2e-5, atol=1e-6 * scale)
    # fused: same photons splatted straight from the trace kernel
    p2 = R.PlotUnit(1, W, H)
    t.render_fused(scene, p2, N, seed=1, stream=0, first_path_index=0)
    t.sync()
    got2 = p2.tristimulus_buffer
    assert np.allclose(got2, want, rtol=2e-5, atol=1e-6 * scale)
    # clear
    p2.clear()
    assert not p2.tristimulus_buffer.any()


def test_plot_reads_photons_before_the_next_render_overwrites_them(demo):
    """PlotUnit::plot is asynchronous on the plot unit's stream while the trace unit is handed out again
    at once (task_scheduler.rs:262-271): the next render into the same mapped_photons must wait for the plot
    kernel (rl_api.hip records an event the trace stream waits on).  Twenty render -> plot rounds back to back
    must add up to the oracle's twenty plots."""
    objs, cam, scene, oscene = demo
    W, H, N, rounds = 96, 54, 1 << 16, 20
    t = R.TraceUnit(0, W, H, n_photons=N)
    p = R.PlotUnit(0, W, H)
    want = np.zeros((W * H, 3), np.float32)
    for k in range(rounds):
        t.render(scene, seed=21, stream=0, first_path_index=k * N)
        p.plot([t])                                       # no host synchronisation before the next render
    for k in range(rounds):
        photons, _ = oscene.render(W, H, 21, 0, k * N, N, threads=8)
        O.plot(W, H, photons, want)
    got = p.tristimulus_buffer
    assert np.allclose(got, want, rtol=2e-5, atol=1e-6 * np.abs(want).max())
    assert abs(float(got.sum(dtype=np.float64)) / float(want.sum(dtype=np.float64)) - 1) < 1e-6


def test_gather_kahan_bit_exact_and_clears_plot(demo):
    objs, cam, scene, oscene = demo
    W, H, N = 160, 90, 1 << 16
    t = R.TraceUnit(0, W, H, n_photons=N)
    p = R.PlotUnit(0, W, H)
    g = R.GatherUnit(W, H)
    acc = np.zeros((W * H, 3), np.float32)
    comp = np.zeros_like(acc)
    for k in range(3):
        t.render(scene, seed=1, stream=0, first_path_index=k * N)
        p.plot([t])
        px = p.tristimulus_buffer
        g.accumulate(p)
        O.accumulate(acc, comp, px)
        assert not p.tristimulus_buffer.any()  # unit.clear(), app.rs:147
    assert g.tristimulus_buffer.tobytes() == acc.tobytes()
    assert g.compensation_buffer.tobytes() == comp.tobytes()


def test_checkpoint_round_trip(demo, tmp_path):
    objs, cam, scene, oscene = demo
    W, H, N = 64, 36, 1 << 14
    t = R.TraceUnit(0, W, H, n_photons=N)
    p = R.PlotUnit(0, W, H)
    g = R.GatherUnit(W, H)
    t.render(scene, seed=3)
    p.plot([t])
    g.accumulate(p)
    path = str(tmp_path / "buffer.raw")
    g.save(path)
    raw = np.fromfile(path, dtype=np.float32)
    assert raw.size == 2 * W * H * 3  # gather_unit.rs:68-78: headerless, tristimulus then compensation
    assert raw[: W * H * 3].tobytes() == g.tristimulus_buffer.tobytes()
    g2 = R.GatherUnit(W, H)
    g2.load(path)
    assert g2.tristimulus_buffer.tobytes() == g.tristimulus_buffer.tobytes()
    assert g2.compensation_buffer.tobytes() == g.compensation_buffer.tobytes()
    # short file leaves the tail untouched (read.rs:20-32)
    raw[: W * H * 3 // 2].tofile(path)
    g3 = R.GatherUnit(W, H)
    g3.load(path)
    got = g3.tristimulus_buffer.reshape(-1)
    assert got[: W * H * 3 // 2].tobytes() == raw[: W * H * 3 // 2].tobytes() and not got[W * H * 3 // 2:].any()


def test_tonemap_bit_exact_given_same_xyz(demo):
    objs, cam, scene, oscene = demo
    W, H, N = 320, 180, 1 << 19
    t = R.TraceUnit(0, W, H, n_photons=N)
    p = R.PlotUnit(0, W, H)
    g = R.GatherUnit(W, H)
    t.render(scene, seed=5)
    p.plot([t])
    g.accumulate(p)
    tm = R.TonemapUnit(W, H)
    tm.tonemap(g)
    xyz = g.tristimulus_buffer
    rgb, srgb, mx = O.tonemap(xyz, W, H)
    got_srgb, got_mx = tm.srgb_float()
    assert np.float32(got_mx).tobytes() == np.float32(mx).tobytes()  # sequential f32 sums, tonemap_unit.rs:55-69
    assert got_srgb.tobytes() == srgb.tobytes()
    assert tm.rgb_buffer.tobytes() == rgb.tobytes()


def test_end_to_end_image_within_1e3(demo):
    """north_star: sRGB within 1e-3 per channel of the CPU path at matched seed (float sRGB before
    the *255 quantisation)."""
    objs, cam, scene, oscene = demo
    W, H = 160, 90
    N = 1 << 18
    batches = 8
    t = R.TraceUnit(0, W, H, n_photons=N)
    p = R.PlotUnit(0, W, H)
    g = R.GatherUnit(W, H)
    acc = np.zeros((W * H, 3), np.float32)
    comp = np.zeros_like(acc)
    for k in range(batches):
        t.render_fused(scene, p, N, seed=1, stream=0, first_path_index=k * N)
        g.accumulate(p)
        photons, _ = oscene.render(W, H, 1, 0, k * N, N, threads=8)
        O.accumulate(acc, comp, O.plot(W, H, photons))
    tm = R.TonemapUnit(W, H)
    tm.tonemap(g)
    got, _ = tm.srgb_float()
    _, want, _ = O.tonemap(acc, W, H)
    assert np.abs(got - want).max() <= 1e-3
    assert np.abs(tm.rgb_buffer.astype(int) - (want * 255).astype(np.uint8).astype(int)).max() <= 1


@pytest.mark.parametrize("fused,concurrency", [(False, 1), (False, 3), (True, 2)])
def test_app_worker_pool_renders_the_same_image_as_the_oracle(demo, fused, concurrency, tmp_path):
    """App (csrc/rl_app.cpp = app.rs:48-164): scheduler + worker threads + units end to end."""
    objs, cam, scene, oscene = demo
    W, H, n, batches = 96, 54, 1 << 14, 14
    ppm, raw = str(tmp_path / "output.ppm"), str(tmp_path / "buffer.raw")
    rgb, st = R.app_run(W, H, batches, concurrency=concurrency, photons_per_batch=n, seed=3, fused=fused, output_ppm=ppm,
                        checkpoint=raw)
    assert st["batches"] == batches and st["paths"] == batches * n and st["tonemaps"] >= 1
    assert st["tasks"]["trace"] == batches and st["tasks"]["plot"] >= 1 and st["tasks"]["gather"] >= 1
    photons, segs = oscene.render(W, H, 3, 0, 0, batches * n, threads=8)
    assert st["segments"] == segs
    xyz = O.plot(W, H, photons)
    want_rgb, want_srgb, _ = O.tonemap(xyz, W, H)
    # batches are plotted/gathered in a thread-dependent order: float sums differ in the last bits only
    assert np.abs(rgb.reshape(-1, 3).astype(int) - want_rgb.astype(int)).max() <= 1
    data = open(ppm, "rb").read()
    assert data.startswith(b"P6\n%d %d\n255\n" % (W, H)) and data[-W * H * 3:] == rgb.tobytes()
    assert os.path.getsize(raw) == 2 * W * H * 12      # gather_unit.rs:68-78
    # PNG output (the reference's output.png): decode the stored-deflate stream and compare
    png = str(tmp_path / "output.png")
    rgb3, _ = R.app_run(W, H, 0, concurrency=1, photons_per_batch=n, seed=3, checkpoint=raw, resume=True, output_ppm=png)
    import zlib
    blob = open(png, "rb").read()
    assert blob[:8] == b"\x89PNG\r\n\x1a\n" and blob[12:16] == b"IHDR"
    idat = blob[blob.index(b"IDAT") + 4: blob.index(b"IEND") - 8]
    rows = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(H, 1 + 3 * W)
    assert (rows[:, 0] == 0).all() and rows[:, 1:].tobytes() == rgb3.tobytes() == rgb.tobytes()
    # resume from the checkpoint: rendering 0 more batches reproduces the image from buffer.raw alone
    rgb2, st2 = R.app_run(W, H, 0, concurrency=1, photons_per_batch=n, seed=3, checkpoint=raw, resume=True)
    assert st2["batches"] == 0 and rgb2.tobytes() == rgb.tobytes()


def test_scene_too_large_for_lds_spills_to_global_fetch():
    """BASELINE config 5's "LDS-spill / global-HBM primitive path": with 1500 seeds per spiral the scene
    blob plus the per-wave scratch exceed 160 KB of LDS, so RL_FETCH_LDS stages the scene's tables (planes, prisms,
    the cull table) and reads the spheres and objects from HBM/L2 instead -- results must not change."""
    objs, cam = R.builtin_scene_desc(R.SCENE_DEMO, 1500)
    assert len(objs) > 4500
    scene, oscene = R.Scene(objs, cam), O.Scene(objs, _ocam(cam))
    W, H, N = 320, 180, 1 << 13
    t = R.TraceUnit(0, W, H, n_photons=N)
    before = R.variant_launches()
    t.render(scene, seed=2, stream=0, first_path_index=0)      # default fetch = LDS, falls back
    want, segs = oscene.render(W, H, 2, 0, 0, N, threads=8)
    assert t.mapped_photons.tobytes() == want.tobytes() and t.stats()[1] == segs
    ran = [a - b for a, b in zip(R.variant_launches(), before)]
    assert sum(ran[16:]) >= 1 and sum(ran[:16]) == 0, ran       # an instantiation that stages the tables only, although the unit asked for LDS


def _poison_lds(pattern):
    """tests/lds_poison: every CU's LDS filled with `pattern` (test infrastructure, built by __graft_entry__.build())."""
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lds_poison", "_build", "liblds_poison.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.run(["make", "-s", "-C", os.path.dirname(os.path.dirname(so))], check=True)
    lib = C.CDLL(so)
    lib.lds_poison.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
    rc = lib.lds_poison(0, pattern, 2048)
    assert rc == 0, rc


@pytest.mark.parametrize("pattern", [0xFFFFFFFF, 0x7FC00000, 0x00ABCDEF])
def test_stale_ring_slots_are_never_loaded_through(pattern):
    """A round of fewer than 64 pairs reads 64 ring slots; the slots beyond the round hold whatever was there -- at the start of a
    kernel, whatever the previous kernel left in LDS.  Their lanes must be pointed at record 0 before anything is loaded through
    them: from LDS a wild index reads zero, from global memory it faults (round 5: the sphere tails of the global-fetch
    variants loaded `spheres[stale >> 6]` unconditionally for a few builds -- found only because the LDS happened to hold
    scene floats).  Here the LDS is filled with a pattern first, then every fetch mode renders few paths (few waves, partial
    rounds everywhere) and must still equal the oracle: built-in scene from LDS and from global memory, and a scene too large
    for LDS (tables staged, spheres from global memory), un-fused plain and open launches."""
    W, H, N = 320, 180, 1 << 12
    for which, param, fetches in [(R.SCENE_DEMO, 0, (R.FETCH_GLOBAL, R.FETCH_LDS)), (R.SCENE_GLASS_STRESS, 0, (R.FETCH_GLOBAL,)),
                                  (R.SCENE_DEMO, 1500, (R.FETCH_LDS,))]:
        objs, cam = R.builtin_scene_desc(which, param)
        scene, oscene = R.Scene(objs, cam), O.Scene(objs, _ocam(cam))
        want, segs = oscene.render(W, H, 4, 2, 77, N, threads=8)
        for fetch in fetches:
            for blocking in (True, False):
                t = R.TraceUnit(0, W, H, n_photons=N)
                t.set_fetch(fetch)
                _poison_lds(pattern)
                if blocking:
                    t.render(scene, seed=4, stream=2, first_path_index=77)          # an open launch
                else:
                    t.render_async(scene, seed=4, stream=2, first_path_index=77)    # a plain launch
                    t.sync()
                assert t.mapped_photons.tobytes() == want.tobytes(), (which, param, fetch, blocking)
                assert t.stats()[1] == segs


_TABLES_SCENES = {}


@pytest.mark.parametrize("scene_name", ["seeds", "prisms"])         # 4,539 objects, 22 prisms; 3,000 random spheres, 48 prisms with a second bound (CYL)
@pytest.mark.parametrize("open_launch", [False, True], ids=["plain", "open"])
@pytest.mark.parametrize("fused", [False, True], ids=["unfused", "fused"])
def test_parity_matrix_over_the_eight_instantiations_that_stage_the_tables(fused, open_launch, scene_name):
    """The parity matrix above for rl_trace_kernel<RL_STAGE_TABLES, fused, open, cyl>: scenes too large for LDS whose planes,
    prisms and cull table are staged and whose spheres and objects come from global memory."""
    if scene_name not in _TABLES_SCENES:
        if scene_name == "seeds":
            objs, cam = R.builtin_scene_desc(R.SCENE_DEMO, 1500)
        else:
            import _random_scene as RS
            objs, cam = RS.random_scene(77, n_spheres=3000, n_prisms=48, n_planes=2, n_circles=3, n_parabs=1)
        _TABLES_SCENES[scene_name] = (R.Scene(objs, cam), O.Scene(objs, _ocam(cam)), {})
    scene, oscene, cache = _TABLES_SCENES[scene_name]
    W, H, N = 320, 180, 1 << 13
    seed, stream, first = 6, 3, 4096
    if "want" not in cache:
        cache["want"] = oscene.render(W, H, seed, stream, first, N, threads=8)
    want, segs = cache["want"]
    t = R.TraceUnit(0, W, H, n_photons=N)
    before = R.variant_launches()
    if not fused:
        if open_launch:
            t.render(scene, seed=seed, stream=stream, first_path_index=first)
        else:
            t.render_async(scene, seed=seed, stream=stream, first_path_index=first)
            t.sync()
        assert t.mapped_photons.tobytes() == want.tobytes()
    else:
        p = R.PlotUnit(0, W, H)
        if open_launch:
            t.render_fused_sync(scene, p, N, seed=seed, stream=stream, first_path_index=first)
        else:
            t.render_fused(scene, p, N, seed=seed, stream=stream, first_path_index=first)
            t.sync()
        got = p.tristimulus_buffer
        ref = O.plot(W, H, want)
        assert np.allclose(got, ref, rtol=2e-5, atol=1e-6 * np.abs(ref).max())
        assert got.any()
    paths, segments, _ = t.stats()
    assert (paths, segments) == (N, segs)
    ran = [a - b for a, b in zip(R.variant_launches(), before)]
    meant = 16 | (4 if fused else 0) | (2 if open_launch else 0) | (1 if scene_name == "prisms" else 0)
    assert ran[meant] >= 1 and sum(ran) == ran[meant], (meant, ran)


@pytest.mark.parametrize("seed", [31, 32, 33, 34])
def test_random_scenes_of_thousands_of_objects_spill_and_stay_bit_exact(seed):
    """2,000 - 8,000 objects: random spheres over a wide range of sizes (a tenth of them far larger than the
    median: the direct list is capped, the rest are clustered whatever their size), prisms, every material; too large for
    LDS, so every record comes from global memory.  Photons bit for bit, segments counted exactly."""
    import _random_scene as RS
    n_spheres = {31: 2000, 32: 3500, 33: 5200, 34: 8000}[seed]
    objs, cam = RS.random_scene(seed, n_spheres=n_spheres, n_prisms=12, n_planes=2, n_circles=3, n_parabs=1)
    scene, oscene = R.Scene(objs, cam), O.Scene(objs, _ocam(cam))
    W, H, N = 160, 90, 1 << 12
    want, segs = oscene.render(W, H, seed, 1, 0, N, threads=8)
    t = R.TraceUnit(0, W, H, n_photons=N)
    t.render(scene, seed=seed, stream=1, first_path_index=0)
    assert t.mapped_photons.tobytes() == want.tobytes() and t.stats()[1] == segs


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["random-6000", "built-in-2500-seeds", "random-20000"])
def test_third_table_level_scenes_bit_exact_in_every_launch_kind(which):
    """Round 6: scenes whose cull table has a third level (>= 112 cluster groups: SUPER bounds, ring T) and whose rounds use the
    progressive far bound -- the variants that do not stage the whole scene.  Tables staged (where they fit) and nothing staged,
    plain and open launches, un-fused byte for byte against the oracle, fused by its segment count and its XYZ image within the
    atomic-order tolerance; the host-side planner (same code on the box, tests/host_mirror) confirms that the scene HAS the level."""
    import _mirror as M
    import _random_scene as RS
    if which == "random-6000":
        objs, cam = RS.random_scene(41, n_spheres=6000, n_prisms=12, n_planes=2, n_circles=3, n_parabs=1)
    elif which == "random-20000":
        objs, cam = RS.random_scene(35, n_spheres=20000, n_prisms=12, n_planes=2, n_circles=3, n_parabs=1)
    else:
        objs, cam = R.builtin_scene_desc(R.SCENE_DEMO, 2500)
    L = M.lib()
    L.mirror_super_bounds.restype = C.c_uint32
    L.mirror_super_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    sup, sg = np.zeros((1024, 4), np.float32), C.c_uint32(0)
    assert L.mirror_super_bounds(M.Scene(objs, _ocam(cam)).h, O.ptr(sup), len(sup), C.byref(sg)) >= 14 and sg.value == 8
    scene, oscene = R.Scene(objs, cam), O.Scene(objs, _ocam(cam))
    W, H, N = 160, 90, 1 << 12
    want, segs = oscene.render(W, H, 9, 3, 128, N, threads=8)
    want_xyz = O.plot(W, H, want)
    scale = np.abs(want_xyz).max()
    for fetch in (R.FETCH_LDS, R.FETCH_GLOBAL):
        t = R.TraceUnit(0, W, H, n_photons=N)
        t.set_fetch(fetch)
        before = R.variant_launches()
        t.render(scene, seed=9, stream=3, first_path_index=128)                       # an open launch
        assert t.mapped_photons.tobytes() == want.tobytes() and t.stats()[1] == segs, (which, fetch, "open")
        t.render_async(scene, seed=9, stream=3, first_path_index=128)                 # a plain launch
        t.sync()
        assert t.mapped_photons.tobytes() == want.tobytes() and t.stats()[1] == 2 * segs, (which, fetch, "plain")
        p = R.PlotUnit(0, W, H)
        t.render_fused(scene, p, N, seed=9, stream=3, first_path_index=128)
        t.sync()
        assert t.stats()[1] == 3 * segs and np.allclose(p.tristimulus_buffer, want_xyz, rtol=2e-5, atol=1e-6 * scale), (which, fetch, "fused")
        ran = [a - b for a, b in zip(R.variant_launches(), before)]
        assert sum(ran[8:16]) == 0                                                    # never the whole-scene variants
        if fetch == R.FETCH_GLOBAL or which == "random-20000":
            assert sum(ran[16:]) == 0 and sum(ran[:8]) == 3                           # nothing staged
        else:                                                                         # the tables staged -- where they fit beside ring T and,
            assert sum(ran[16:]) >= 2 and sum(ran[16:]) + sum(ran[:8]) == 3           # in the open launch, the workgroup's job counters


def test_degenerate_scenes_and_empty_launches(demo):
    """Edge cases of the scan: the empty scene (every path ends in The Void, scene.rs:43-60 returns None),
    scenes that hold a single surface kind (each of the kernel's per-kind loops runs with the others
    empty), a zero-path fused launch, and zero-sized units."""
    from _random_scene import random_scene
    full, cam = random_scene(3, n_spheres=40, n_prisms=3, n_planes=2, n_circles=2, n_parabs=2)
    N = 1 << 12
    subsets = {"empty": full[:0]}
    for kind, name in ((0, "spheres"), (1, "planes"), (2, "circles"), (3, "paraboloids"), (4, "prisms")):
        subsets[name] = full[full["surface_kind"] == kind].copy()
    subsets["one emissive sphere"] = full[:1].copy()
    for name, objs in subsets.items():
        scene, oscene = R.Scene(objs, cam), O.Scene(objs, _ocam(cam))
        for fetch in (R.FETCH_LDS, R.FETCH_GLOBAL):
            t = R.TraceUnit(0, 160, 90, n_photons=N)
            t.set_fetch(fetch)
            t.render(scene, seed=5, stream=2, first_path_index=1000)
            want, segs = oscene.render(160, 90, 5, 2, 1000, N, threads=4)
            assert t.mapped_photons.tobytes() == want.tobytes() and t.stats()[1] == segs, name
        if name == "empty":
            assert not t.mapped_photons["probability"].any() and segs == N
    objs, cam, scene, oscene = demo
    t = R.TraceUnit(0, 64, 36, n_photons=64)
    plot = R.PlotUnit(0, 64, 36)
    t.render_fused(scene, plot, 0, seed=1, stream=0, first_path_index=0)   # nothing to do, nothing written
    assert t.stats()[0] == 0 and not plot.tristimulus_buffer.any()
    for bad in (dict(width=0, height=36, n_photons=64), dict(width=64, height=0, n_photons=64),
                dict(width=64, height=36, n_photons=0)):
        with pytest.raises(R.RlError):
            R.TraceUnit(0, bad["width"], bad["height"], n_photons=bad["n_photons"])


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_random_scenes_bit_exact_on_device(seed):
    from _random_scene import random_scene
    # from a handful of spheres (direct list only) to several hundred (60+ clusters), 0..20 prisms
    objs, cam = random_scene(seed, n_spheres=[5, 20, 33, 64, 100, 150, 220, 300, 400, 520, 31, 32][seed - 1],
                             n_prisms=[0, 1, 3, 6, 10, 20, 2, 4, 8, 12, 0, 5][seed - 1])
    scene, oscene = R.Scene(objs, cam), O.Scene(objs, _ocam(cam))
    N = 1 << 15
    for fetch in (R.FETCH_LDS, R.FETCH_GLOBAL):
        t = R.TraceUnit(0, 320, 180, n_photons=N)
        t.set_fetch(fetch)
        t.render(scene, seed=seed, stream=0, first_path_index=0)
        want, segs = oscene.render(320, 180, seed, 0, 0, N, threads=8)
        assert t.mapped_photons.tobytes() == want.tobytes() and t.stats()[1] == segs


@pytest.mark.parametrize("layout", ["same", "line", "zero_radius", "huge_spread", "infinite"])
def test_degenerate_sphere_layouts_through_the_planned_table_on_device(layout):
    """The cull table is planned per scene (rl_scene.cpp: cluster size 10 or 14, 3 or 4 clusters per group, local search, cost
    estimate over sample paths); here it gets sphere sets it cannot do anything sensible with, at the sizes where clusters
    start and where groups fill unevenly.  The photons must equal the oracle's linear scan bit for bit in both fetch modes."""
    objs0, cam = R.builtin_scene_desc(R.SCENE_DEMO)
    proto = objs0[objs0["surface_kind"] == 0][:1]
    rest = objs0[objs0["surface_kind"] != 0]
    rng = np.random.default_rng(7)
    N = 1 << 14
    for n in (40, 41, 57, 141):
        o = np.repeat(proto, n)
        o["v0"] = rng.normal(0, 8, (n, 3)).astype(np.float32)
        o["v0"][:, 1] = np.abs(o["v0"][:, 1])
        o["f"][:, 0] = rng.uniform(0.1, 1.0, n).astype(np.float32)
        if layout == "same":
            o["v0"] = np.array([1.0, 2.0, 3.0], np.float32)
        elif layout == "line":
            o["v0"] = np.stack([np.linspace(-20, 20, n), np.ones(n), np.ones(n)], 1).astype(np.float32)
        elif layout == "zero_radius":
            o["f"][:, 0] = 0.0
        elif layout == "huge_spread":
            o["v0"] = (rng.normal(0, 1, (n, 3)) * np.exp(rng.uniform(-5, 12, (n, 1)))).astype(np.float32)
            o["f"][:, 0] = np.exp(rng.uniform(-8, 3, n)).astype(np.float32)
        elif layout == "infinite":
            o["f"][0, 0] = np.inf
        objs = np.concatenate([rest, o])
        want, segs = O.Scene(objs, _ocam(cam)).render(320, 180, 3, 0, 0, N, threads=8)
        scene = R.Scene(objs, cam)
        for fetch in (R.FETCH_LDS, R.FETCH_GLOBAL):
            t = R.TraceUnit(0, 320, 180, n_photons=N)
            t.set_fetch(fetch)
            t.render(scene, seed=3, stream=0, first_path_index=0)
            assert t.mapped_photons.tobytes() == want.tobytes() and t.stats()[1] == segs, (layout, n, fetch)


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_random_scenes_with_many_prisms_use_the_second_bound_and_stay_bit_exact(seed):
    """From 40 prisms on the kernel tests a cylinder around every prism's axis besides its bounding sphere (the CYL
    instantiations of rl_trace_kernel): randomly placed and oriented prisms, both fetch modes, against the oracle."""
    import _random_scene as RS
    objs, cam = RS.random_scene(seed, n_spheres=60, n_prisms=48 + seed)
    scene, oscene = R.Scene(objs, cam), O.Scene(objs.view(O.OBJECT_DTYPE), _ocam(cam))
    n = 1 << 15
    want, segs = oscene.render(320, 180, seed, 0, 0, n, threads=4)
    for fetch in (R.FETCH_LDS, R.FETCH_GLOBAL):
        t = R.TraceUnit(0, 320, 180, n_photons=n)
        t.set_fetch(fetch)
        t.render(scene, seed=seed, stream=0, first_path_index=0)            # an open launch
        assert t.mapped_photons.tobytes() == want.tobytes() and t.stats()[:2] == (n, segs)
        t.render_async(scene, seed=seed, stream=0, first_path_index=0)      # a plain launch
        t.sync()
        assert t.mapped_photons.tobytes() == want.tobytes()


def test_cpp_client_of_the_c_abi_renders_a_png(tmp_path):
    """examples/render.cpp: main.rs as a compiled client that links only against the C ABI."""
    import subprocess
    import zlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "render")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(root, "examples")], check=True)
    out = str(tmp_path / "output.png")
    r = subprocess.run([exe, "2", "128", "72", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "wrote image" in r.stdout
    blob = open(out, "rb").read()
    idat = blob[blob.index(b"IDAT") + 4: blob.index(b"IEND") - 8]
    rows = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(72, 1 + 3 * 128)
    assert rows[:, 1:].mean() > 5   # an image, not a black frame


def test_full_size_properties_1080p(demo):
    """BASELINE full size (1920x1080, built-in scene) through size-independent properties, since the oracle
    cannot trace tens of millions of paths in a test: path/segment counters, additivity over path ranges,
    fused == un-fused + plot, energy conservation of the splat, clear/gather identities."""
    objs, cam, scene, oscene = demo
    W, H = 1920, 1080
    n = 32 * R.NUMBER_OF_PHOTONS                     # 16.8 M paths
    t = R.TraceUnit(0, W, H, n_photons=1 << 22)
    whole, a, b = R.PlotUnit(0, W, H), R.PlotUnit(1, W, H), R.PlotUnit(2, W, H)
    t.render_fused(scene, whole, n, seed=1, stream=0, first_path_index=0)
    p0, s0, _ = t.stats()
    assert p0 == n
    # additivity: the same paths in two launches give the same counters and the same image up to float order
    t.render_fused(scene, a, n // 2, seed=1, stream=0, first_path_index=0)
    t.render_fused(scene, b, n - n // 2, seed=1, stream=0, first_path_index=n // 2)
    p1, s1, _ = t.stats()
    assert p1 - p0 == n and s1 - s0 == s0
    xw, xa, xb = whole.tristimulus_buffer, a.tristimulus_buffer, b.tristimulus_buffer
    scale = np.abs(xw).max()
    assert np.allclose(xa + xb, xw, rtol=1e-4, atol=2e-6 * scale)
    assert abs(float(xw.sum(dtype=np.float64)) / float((xa.astype(np.float64) + xb).sum()) - 1) < 1e-6
    # a different stream is a different sample of the same image: equal in the mean, not in the pixels
    t.render_fused(scene, a, n // 2, seed=1, stream=7, first_path_index=0)  # a now holds first half + stream 7
    xa2 = a.tristimulus_buffer - xa
    assert not np.array_equal(xa2, xa) and abs(xa2[:, 1].sum(dtype=np.float64) / xa[:, 1].sum(dtype=np.float64) - 1) < 0.01
    # fused == un-fused + plot, and the splat conserves energy: sum(Y) == sum(prob * ybar(lambda))
    t.render(scene, seed=1, stream=0, first_path_index=0)
    ph = t.mapped_photons
    b.clear()
    b.plot([t])
    whole.clear()
    t.render_fused(scene, whole, t.n_photons, seed=1, stream=0, first_path_index=0)
    xu, xf = b.tristimulus_buffer, whole.tristimulus_buffer
    assert np.allclose(xu, xf, rtol=1e-4, atol=2e-6 * np.abs(xu).max())
    import json
    tab = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cie1931_xyz.json")))
    Y = np.array(tab["Y"], dtype=np.float64)
    idxf = (ph["wavelength"].astype(np.float64) - 380.0) / 5.0
    i0 = np.clip(np.floor(idxf).astype(int), 0, 79)
    ybar = Y[i0] * (1 - (idxf - i0)) + Y[i0 + 1] * (idxf - i0)          # cie1931.rs:41-47
    want_y = float((ph["probability"].astype(np.float64) * ybar).sum())
    got_y = float(xu[:, 1].sum(dtype=np.float64))
    assert abs(got_y / want_y - 1) < 1e-5       # bilinear weights sum to one (plot_unit.rs:74-77)
    # gather of a cleared buffer is the identity; gather clears its input
    g = R.GatherUnit(W, H)
    g.accumulate(whole)
    before = g.tristimulus_buffer
    assert not whole.tristimulus_buffer.any()
    g.accumulate(whole)
    assert g.tristimulus_buffer.tobytes() == before.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("which", [R.SCENE_DEMO, R.SCENE_GLASS_STRESS])
def test_prism_shortcut_on_the_device_never_contradicts_the_compound_tree(which):
    """rl_hex_prism_fast with the hardware's v_rcp_f32 against the Compound tree (geometry.rs:380-407), both on the GPU
    (rl_debug_prism_probe): whatever the shortcut decides is the tree's distance and half-space bit for bit, and the
    device's tree is the g++ build's.  Adversarial pairs from tests/host_mirror (rays starting on faces, aimed at edges and
    vertices, nearly parallel to faces)."""
    import _mirror as M
    objs, cam = R.builtin_scene_desc(which)
    scene = R.Scene(objs, cam)
    ms = M.Scene(objs.view(O.OBJECT_DTYPE), O.RlCameraDesc.from_buffer_copy(bytes(cam)))
    prisms, rays, tree = M.prism_pairs(ms, 1_500_000, 77 + which)
    assert len(prisms) > 1_000_000 and R.prism_count(scene) > prisms.max()
    decided = undecided = 0
    for p in np.unique(prisms):
        sel = prisms == p
        out = R.prism_probe(scene, p, rays[sel])
        assert (out[:, 3:5][out[:, 3] != 0xffffffff] == tree[sel][tree[sel][:, 0] != 0xffffffff]).all()   # same tree on both sides
        assert ((out[:, 3] == 0xffffffff) == (tree[sel][:, 0] == 0xffffffff)).all()
        hit, miss = out[:, 0] == 1, out[:, 0] == 0
        assert (out[hit, 1] == out[hit, 3]).all() and (out[hit, 2] == out[hit, 4]).all()   # a decided hit: the tree's t and half-space
        assert (out[miss, 3] == 0xffffffff).all()                                          # a decided miss: the tree misses
        decided += int(hit.sum() + miss.sum())
        undecided += int((out[:, 0] == 2).sum())
    assert decided > 0.75 * len(prisms) and undecided > 0   # it decides most pairs, and the adversarial ones reach the tree

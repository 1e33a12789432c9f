"""Known-answer tests that pin the CPU oracle.  The reference's own tests assert no numeric value
("parity unpinned"), so every expectation here is derived independently of the oracle's
C++: either from a numpy float32/float64 restatement of the Rust formula written in this file, from
published vectors (Random123's Philox KATs), or from the data tables the reference tabulates."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import _oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

f32 = np.float32
PI = f32(math.pi)
HERE = os.path.dirname(os.path.abspath(__file__))


def ulp_diff(a, b):
    a = np.asarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.asarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7fffffff), a)
    b = np.where(b < 0, -(b & 0x7fffffff), b)
    return np.abs(a - b)


# ---- random numbers ------------------------------------------------------------------------------

def py_philox4x32_10(ctr, key, rounds=10):
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c, k = list(ctr), list(key)
    for r in range(rounds):
        if r:
            k = [(k[0] + W0) & 0xffffffff, (k[1] + W1) & 0xffffffff]
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xffffffff, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xffffffff]
    return c


@pytest.mark.parametrize("ctr,key,want", [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
])
def test_philox_random123_known_answers(ctr, key, want):
    c = np.array(ctr, dtype=np.uint32)
    k = np.array(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    O.lib().oracle_philox(O.ptr(c), O.ptr(k), O.ptr(out))
    assert list(out) == want
    assert py_philox4x32_10(ctr, key) == want


def test_rng_block_keying():
    out = np.zeros(4, dtype=np.uint32)
    seed, stream, path, block = 0x1122334455667788, 7, 0x0000000512345678, 9
    O.lib().oracle_rng_block(seed, stream, path, block, O.ptr(out))
    want = py_philox4x32_10([path & 0xffffffff, path >> 32, block, stream], [seed & 0xffffffff, seed >> 32], rounds=7)
    assert list(out) == want                            # RL_PHILOX_ROUNDS = 7 (csrc/rl_rng.h)


def test_uniform_conversions_follow_rand_0_3():
    u = np.array([0, 1 << 8, 0x7fffffff, 0xffffff00, 0xffffffff, 0x12345678], dtype=np.uint32)
    half = O.math_f32("halfopen01", u.view(np.float32))
    closed = O.math_f32("closed01", u.view(np.float32))
    want_half = (u >> 8).astype(np.float32) * f32(2.0 ** -24)
    assert half.tobytes() == want_half.tobytes()
    assert half.max() < 1.0 and half[0] == 0.0
    want_closed = want_half * (f32(16777216.0) / f32(16777215.0))
    assert closed.tobytes() == want_closed.tobytes()
    assert closed[4] == 1.0 and closed[0] == 0.0  # Closed01 reaches both ends (monte_carlo.rs:25-28)


# ---- rl_math.h against numpy's libm ----------------------------------------------------------------

@pytest.mark.parametrize("fn,ref,lo,hi,correctly_rounded", [
    # the per-bounce functions, f32 arithmetic (round 4): never farther than the neighbouring float of the correctly rounded
    # value; of uniformly distributed arguments 97 % (sin, cos), 91 % (exp), 86 % (acos) ARE the correctly rounded value
    ("sin", np.sin, -25.0, 25.0, 0.96), ("cos", np.cos, -25.0, 25.0, 0.96), ("exp", np.exp, -80.0, 20.0, 0.89), ("acos", np.arccos, -0.999, 0.999, 0.85),
    ("sin", np.sin, -300.0, 300.0, 0.96), ("exp", np.exp, -110.0, 95.0, 0.89),   # (beyond |x| = 32 / below -86: the f64-evaluated forms)
    # f64 evaluation, single rounding: almost always correctly rounded
    ("sin_d", np.sin, -25.0, 25.0, 0.999), ("cos_d", np.cos, -25.0, 25.0, 0.999), ("exp_d", np.exp, -80.0, 20.0, 0.999),
    ("acos_d", np.arccos, -0.999, 0.999, 0.999), ("tan", np.tan, 0.05, 1.5, 0.999), ("log", np.log, 1e-30, 1e30, 0.999)])
def test_math_header_within_one_ulp_of_libm(fn, ref, lo, hi, correctly_rounded):
    rng = np.random.default_rng(1)
    if fn == "log":
        x = np.exp(rng.uniform(math.log(lo), math.log(hi), 200000)).astype(np.float32)
    else:
        x = rng.uniform(lo, hi, 200000).astype(np.float32)
    got = O.math_f32(fn, x)
    with np.errstate(over="ignore", under="ignore"):
        want = ref(x.astype(np.float64)).astype(np.float32)
    d = ulp_diff(got, want)
    assert d.max() <= 1
    assert (d == 0).mean() > correctly_rounded


def test_math_ulp_tool_sampled():
    """tools/math_ulp_check.cpp over every 4096th f32 argument of each domain (the exhaustive run is profiles/r04_math_ulp.txt):
    no result of either family is farther from the correctly rounded float than its neighbour."""
    import subprocess, tempfile
    exe = os.path.join(tempfile.mkdtemp(), "math_ulp_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-pthread", "-I", os.path.join(ROOT, "robigo_luculenta_amd", "csrc"),
                    "-o", exe, os.path.join(ROOT, "tools", "math_ulp_check.cpp")], check=True)
    out = subprocess.run([exe, "4096"], check=True, capture_output=True, text=True).stdout
    rows = [l for l in out.splitlines() if "max error" in l]
    assert len(rows) == 9
    for l in rows:
        assert "farther than the neighbour 0 " in l, l
        assert float(l.split("max error")[1].split()[0]) < (1.1 if not "_d" in l.split()[0] else 0.5001), l


def test_math_header_special_points():
    x = np.array([0.0, 1.0, -1.0, math.pi, 0.5 * math.pi], dtype=np.float32)
    assert O.math_f32("sin", x[:1])[0] == 0.0 and O.math_f32("cos", x[:1])[0] == 1.0
    assert O.math_f32("exp", x[:1])[0] == 1.0 and O.math_f32("log", x[1:2])[0] == 0.0
    assert O.math_f32("log", np.array([4.0], np.float32))[0] == f32(math.log(4.0))
    xs = np.array([-745.5, -104.0, -87.4, 88.0], dtype=np.float64)
    ys = np.zeros_like(xs)
    O.lib().oracle_exp_f64(O.ptr(xs), O.ptr(ys), xs.size)
    assert ys[0] == 0.0 and np.allclose(ys[1:], np.exp(xs[1:]), rtol=1e-14)


def test_exp_f64_accuracy():
    rng = np.random.default_rng(2)
    xs = rng.uniform(-40, 40, 100000)
    ys = np.zeros_like(xs)
    O.lib().oracle_exp_f64(O.ptr(xs), O.ptr(ys), xs.size)
    assert np.max(np.abs(ys / np.exp(xs) - 1.0)) < 1e-15 * 4


# ---- data tables: cie1931.rs -----------------------------------------------------------------------

def test_cie_table_and_interpolation():
    tab = json.load(open(os.path.join(HERE, "golden", "cie1931_xyz.json")))
    X, Y, Z = (np.array(tab[k], dtype=np.float32) for k in "XYZ")
    assert len(X) == len(Y) == len(Z) == 81
    assert (X[35], Y[35], Z[35]) == (f32(0.512050), f32(1.0), f32(0.005750))  # 555 nm, cie1931.rs:89,174,259
    out = np.zeros(3, dtype=np.float32)

    def tri(w):
        O.lib().oracle_tristimulus(f32(w), O.ptr(out))
        return out.copy()

    for i in range(81):  # at the nodes the lerp returns the table entry
        assert tri(380 + 5 * i).tobytes() == np.array([X[i], Y[i], Z[i]]).tobytes()
    rng = np.random.default_rng(3)
    for w in rng.uniform(380, 780, 2000).astype(np.float32):
        indexf = (w - f32(380.0)) / f32(5.0)
        i = int(np.floor(indexf))
        r = indexf - f32(i)
        if i == 80:
            want = np.array([X[80] * (f32(1) - r), Y[80] * (f32(1) - r), Z[80] * (f32(1) - r)])
        else:
            want = np.array([X[i] * (f32(1) - r) + X[i + 1] * r, Y[i] * (f32(1) - r) + Y[i + 1] * r,
                             Z[i] * (f32(1) - r) + Z[i + 1] * r], dtype=np.float32)
        assert tri(w).tobytes() == want.astype(np.float32).tobytes()
    # below 380: index -1 scales the first entry; outside [375, 785): black (cie1931.rs:25-33)
    w = f32(377.0)
    r = (w - f32(380)) / f32(5) - f32(-1)
    assert tri(w).tobytes() == np.array([X[0] * r, Y[0] * r, Z[0] * r], dtype=np.float32).tobytes()
    assert not tri(374.0).any() and not tri(786.0).any()
    assert tri(782.5).tobytes() == (np.array([X[80], Y[80], Z[80]]) * (f32(1) - f32(0.5))).astype(np.float32).tobytes()


# ---- materials --------------------------------------------------------------------------------------

def np_sf10(w):
    w2 = np.float64(f32(w) * f32(w) * f32(1.0e-6))
    return f32(np.sqrt(1.0 + 1.737596950 * w2 / (w2 - 0.0131887070) + 0.313747346 * w2 / (w2 - 0.0623068142)
                       + 1.898781010 * w2 / (w2 - 155.23629000)))


def test_sf10_index_of_refraction():
    spots = {380: 1.8607413, 480: 1.8084033, 580: 1.7859017, 680: 1.773569, 780: 1.7658347}
    for w, n in spots.items():
        got = O.lib().oracle_sf10_ior(f32(w))
        assert f32(got) == f32(n)
        assert f32(got) == np_sf10(w)
    rng = np.random.default_rng(4)
    for w in rng.uniform(380, 780, 500).astype(np.float32):
        assert f32(O.lib().oracle_sf10_ior(w)) == np_sf10(w)


def np_boltzmann(wavelength, temperature):
    h, k, c = 6.62606957e-34, 1.3806488e-23, 299792458.0
    f = c / (wavelength * 1.0e-9)
    return (2.0 * h * f * f * f) / (c * c * (np.exp(h * f / (k * temperature)) - 1.0))


def test_black_body():
    norm = C.c_float(0)
    # SURVEY 8a a13 spot values: T = 6504, intensity 1
    for w, want in [(380, 0.68180025), (580, 1.455191), (780, 1.6527456)]:
        got = O.lib().oracle_black_body(f32(6504), f32(1.0), f32(w), C.byref(norm))
        assert abs(got / want - 1) < 2e-7
    assert abs(norm.value / 31682932.0 - 1) < 2e-7
    for T, I in [(6504.0, 1.0), (7600.0, 0.6), (5000.0, 0.6)]:
        n = f32(I) / f32(np_boltzmann((2.897772126e-3 / np.float64(f32(T))) * 1.0e9, np.float64(f32(T))))
        for w in (380.0, 431.7, 555.0, 780.0):
            got = O.lib().oracle_black_body(f32(T), f32(I), f32(w), C.byref(norm))
            want = f32(np_boltzmann(np.float64(f32(w)), np.float64(f32(T)))) * n
            assert ulp_diff(got, want) <= 1  # rl_exp_d vs libm exp differ by < 1 ulp of f64
            assert ulp_diff(norm.value, n) <= 1


def test_srgb_transform():
    xyz = np.array([0.9505, 1.0, 1.089], dtype=np.float32)  # D65 white -> (1, 1, 1) within matrix rounding
    rgb = np.zeros(3, dtype=np.float32)
    O.lib().oracle_srgb(O.ptr(xyz), O.ptr(rgb))
    assert np.allclose(rgb, 1.0, atol=2e-3)
    xyz = np.array([0.001, 0.001, 0.001], dtype=np.float32)  # linear segment: 12.92 f (srgb.rs:21-22)
    O.lib().oracle_srgb(O.ptr(xyz), O.ptr(rgb))
    r = f32(3.2406) * xyz[0] - f32(1.5372) * xyz[1] - f32(0.4986) * xyz[2]
    assert rgb[0] == f32(12.92) * r
    xyz = np.array([0.3, 0.4, 0.2], dtype=np.float32)
    O.lib().oracle_srgb(O.ptr(xyz), O.ptr(rgb))
    g = f32(-0.9689) * xyz[0] + f32(1.8758) * xyz[1] + f32(0.0415) * xyz[2]
    want = f32(1.055) * f32(np.float64(g) ** np.float64(f32(1.0) / f32(2.4))) - f32(0.055)
    assert ulp_diff(rgb[1], want) <= 1


# ---- camera: app.rs:327-357 -------------------------------------------------------------------------

def test_camera_spot_values():
    objs, cam = O.demo_scene_desc()
    out = np.zeros(10, dtype=np.float32)
    O.lib().oracle_camera(C.byref(cam), f32(0.0), O.ptr(out))
    # t = 0 -> position (-2.57e-6, -29.38926, 40.45085), focal 45.0, screen distance 1.6318517
    assert abs(out[0] - -2.5692905e-06) < 1e-9 and abs(out[1] - -29.38926) < 1e-5 and abs(out[2] - 40.45085) < 1e-5
    assert out[8] == f32(45.0)
    assert abs(out[9] - 1.6318517) < 2e-7
    assert out[7] == PI * f32(0.35)
    # numpy restatement at t = 0.37
    t = f32(0.37)
    phi = PI * (f32(1.0) + f32(0.01) * t)
    alpha = PI * (f32(0.3) - f32(0.01) * t)
    dist = f32(50.0) - f32(0.5) * t
    O.lib().oracle_camera(C.byref(cam), t, O.ptr(out))
    s, c = lambda v: f32(np.sin(np.float64(v))), lambda v: f32(np.cos(np.float64(v)))
    want = np.array([c(alpha) * s(phi) * dist, c(alpha) * c(phi) * dist, s(alpha) * dist], dtype=np.float32)
    assert ulp_diff(out[:3], want).max() <= 2
    # orientation is a unit quaternion
    assert abs(np.sum(out[3:7].astype(np.float64) ** 2) - 1.0) < 1e-6


# ---- scene: app.rs:166-325 --------------------------------------------------------------------------

def test_demo_scene_inventory():
    objs, cam = O.demo_scene_desc()
    assert len(objs) == 339
    assert list(np.bincount(objs["surface_kind"], minlength=5)) == [311, 1, 2, 3, 22]  # SURVEY 2.1 totals
    assert list(np.bincount(objs["material_kind"], minlength=6)) == [3, 1, 103, 100, 22, 110]
    gamma = PI * f32(2.0) * (f32(1.0) - f32(1.0) / f32(1.6180339887498948))
    assert abs(gamma - 2.3999631) < 1e-7
    fs = f32(5.0) / f32(1.5) + f32(1.0)
    assert int(fs * fs + f32(0.5)) == 19
    # first sunflower seed (object 7): i = 19
    i = 19
    phi = f32(i) * gamma
    r = np.sqrt(f32(i)) * f32(1.5)
    want = np.array([f32(np.cos(np.float64(phi))) * r, f32(np.sin(np.float64(phi))) * r, (r - f32(5.0)) * f32(-0.5)])
    assert ulp_diff(objs[7]["v0"], want).max() <= 1
    assert objs[7]["f"][0] == f32(0.8) and objs[7]["m"][1] == f32(600.0)
    assert objs[106]["m"][1] == f32(99) / f32(100) * f32(130.0) + f32(600.0)
    # bubbles start at i = 9 with negative angle (app.rs:271-272)
    assert objs[207]["material_kind"] == 5 and objs[316]["material_kind"] == 5 and objs[317]["surface_kind"] == 4
    r9 = np.sqrt(f32(9)) * f32(1.5) * f32(1.5)
    assert objs[207]["v0"][2] == (r9 - f32(5.0)) * f32(1.5) + f32(10.0)
    # prisms: edge 3, bevel 1, heights 8 and 12; axis is a unit vector leaning with the floor
    pr = objs[317:]
    assert set(pr["f"][:, 3]) == {f32(8.0), f32(12.0)} and (pr["f"][:, 0] == 3).all() and (pr["f"][:, 1] == 1).all()
    assert np.allclose(np.linalg.norm(pr["v0"].astype(np.float64), axis=1), 1.0, atol=1e-6)
    assert (pr["v0"][:, 2] > 0.9).all()  # -intersection.normal: standing up from the floor
    # replicated-primitive scene (BASELINE config 5): seeds = 158 -> 513 objects
    assert len(O.demo_scene_desc(158)[0]) == 513


# ---- geometry known answers incl. the quirk list ------------------------------------------

def one_object_scene(surface_kind, v0=(0, 0, 0), v1=(0, 0, 0), f=(0, 0, 0, 0)):
    objs = np.zeros(1, dtype=O.OBJECT_DTYPE)
    objs[0]["surface_kind"] = surface_kind
    objs[0]["material_kind"] = 1
    objs[0]["v0"], objs[0]["v1"], objs[0]["f"] = v0, v1, f
    objs[0]["m"] = (0.5, 0, 0)
    return O.Scene(objs, O.demo_scene_desc()[1])


def test_sphere_hits_and_quirks():
    s = one_object_scene(0, v0=(0, 0, 10), f=(2, 0, 0, 0))
    h = s.intersect_object(0, (0, 0, 0), (0, 0, 1))
    assert h is not None and h[9] == 8.0 and tuple(h[:3]) == (0, 0, 8) and tuple(h[3:6]) == (0, 0, -1)
    assert tuple(h[6:9]) == (-1, 0, 0)  # tangent = normalise(cross((0,1,0), n)) (geometry.rs:250-251)
    assert s.intersect_object(0, (0, 0, 0), (0, 0, -1)) is None      # behind the ray
    assert s.intersect_object(0, (0, 0, 10), (0, 0, 1)) is None      # origin inside: t1 < 0 < t2 -> no hit (:236-240)
    assert s.intersect_object(0, (2, 0, 0), (0, 0, 1)) is None       # tangent graze: disc == 0 -> t1 == t2 -> no hit
    assert s.intersect_object(0, (2.5, 0, 0), (0, 0, 1)) is None     # clean miss
    # generic ray against the numpy restatement of geometry.rs:204-221
    o = np.array([0.3, -0.2, 1.0], np.float32)
    d = np.array([0.1, 0.05, 1.0], np.float32)
    d = d / np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
    co = np.array([0, 0, 10], np.float32) - o
    b = f32(2.0) * (d[0] * co[0] + d[1] * co[1] + d[2] * co[2])
    c = (co[0] * co[0] + co[1] * co[1] + co[2] * co[2]) - f32(4.0)
    disc = b * b - f32(4.0) * c
    t1 = f32(-0.5) * (-b + np.sqrt(disc))
    h = s.intersect_object(0, o, d)
    assert h[9] == t1


def test_plane_circle_and_asymmetric_rejection():
    p = one_object_scene(1, v0=(0, 0, -1), v1=(0, 0, 60))
    h = p.intersect_object(0, (1, 2, 0), (0, 0, 1))
    assert h[9] == 60 and tuple(h[3:6]) == (0, 0, -1)                # d < 0 -> +normal
    h = p.intersect_object(0, (1, 2, 70), (0, 0, -1))
    assert h[9] == 10 and tuple(h[3:6]) == (0, 0, 1)                 # two-sided (geometry.rs:79)
    assert p.intersect_object(0, (0, 0, 0), (1, 0, 0)) is None       # d == 0
    assert p.intersect_object(0, (0, 0, 60), (0, 0, 1)) is None      # t == 0 rejected (t <= 0, geometry.rs:66)
    c = one_object_scene(2, v0=(0, 0, -1), v1=(-5, 0, 30), f=(5, 0, 0, 0))
    assert c.intersect_object(0, (-5, 5, 0), (0, 0, 1)) is not None  # on the rim: <= (geometry.rs:170)
    assert c.intersect_object(0, (-5, 5.001, 0), (0, 0, 1)) is None


def test_paraboloid_known_answers():
    # floor of the demo scene: normal (0,0,-1), offset (0,0,-5), focal distance 25
    s = one_object_scene(3, v0=(0, 0, -1), v1=(0, 0, -5), f=(25, 0, 0, 0))
    # straight down the axis direction: a == 0 -> linear branch (geometry.rs:314-319)
    h = s.intersect_object(0, (17, 0, 0), (0, 0, -1))
    # z = -5 + r^2/(4 f) is the paraboloid through the vertex (0,0,-5) opening upward... check position on surface:
    r2 = 17.0 * 17.0
    assert abs(h[2] - (-5 + r2 / 100.0)) < 1e-4 or abs(h[2] - (-5 - r2 / 100.0)) < 1e-4
    n = h[3:6].astype(np.float64)
    assert abs(np.linalg.norm(n) - 1) < 1e-6
    # generic oblique ray: the hit satisfies |p - focus| == distance to directrix plane
    o = np.array([3.0, -4.0, 20.0], np.float32)
    d = np.array([0.2, 0.1, -1.0], np.float32)
    d = (d / np.linalg.norm(d)).astype(np.float32)
    h = s.intersect_object(0, o, d)
    assert h is not None and h[9] > 0
    offset = np.array([0, 0, -5.0]) - np.array([0, 0, -1.0]) * 25.0  # plane point
    focus = offset + np.array([0, 0, -1.0]) * 50.0
    p = h[:3].astype(np.float64)
    assert abs(np.linalg.norm(p - focus) - abs(np.dot(p - offset, [0, 0, -1.0]))) < 1e-3


def np_hex_planes(axis, offset, edge, bevel, angle, height):
    """8 half-spaces (normal, point) of new_hexagonal_prism for an axis with |axis.z| > 0.9999 or not."""
    def rot(v, n):
        if n[2] > 0.9999:
            return v
        if n[2] < -0.9999:
            return np.array([v[0], v[1], -v[2]])
        up = np.array([0, 0, 1.0])
        a1 = np.cross(up, n); a1 /= np.linalg.norm(a1)
        a2 = np.cross(a1, n); a2 /= np.linalg.norm(a2)
        return a1 * v[0] + a2 * v[1] + n * v[2]

    def inf_prism(e, ang):
        radius = math.sqrt(3.0) / 6.0 * e
        out = []
        for k in range(3):
            a = ang + k * 2 * math.pi / 3
            p = rot(np.array([math.cos(a), math.sin(a), 0.0]), axis)
            out.append((p, p * radius + offset))
        return out

    axis, offset = np.asarray(axis, float), np.asarray(offset, float)
    planes = inf_prism(edge * 2 - bevel * 3, angle + math.pi) + inf_prism(edge, angle)
    planes += [(-axis, offset), (axis, offset + axis * height)]
    return planes


def test_hex_prism_against_convex_polytope():
    """For a convex intersection of half-spaces the recursive Compound (geometry.rs:380-399) returns the
    entry point when the origin is outside and the exit point when inside (f64 interval arithmetic)."""
    axis = np.array([0.1, -0.2, -1.0]); axis /= np.linalg.norm(axis)
    offset = np.array([17.0, 3.0, -2.0])
    s = one_object_scene(4, v0=axis, v1=offset, f=(3.0, 1.0, 0.7, 8.0))
    planes = np_hex_planes(axis, offset, 3.0, 1.0, 0.7, 8.0)
    rng = np.random.default_rng(5)
    centre = offset + axis * 4.0
    hits = 0
    for k in range(4000):
        if k % 2:
            o = centre + rng.normal(size=3) * 0.4      # inside
        else:
            o = centre + rng.normal(size=3) * 12.0
        d = centre + rng.normal(size=3) * 2.5 - o
        d /= np.linalg.norm(d)
        o32, d32 = o.astype(np.float32), d.astype(np.float32)
        d32 = d32 / np.sqrt((d32 * d32).sum(dtype=np.float32))
        lo, hi, lo_n, hi_n = -np.inf, np.inf, None, None
        for n, p in planes:
            dn = float(np.dot(n, d32))
            num = float(np.dot(n, o32.astype(np.float64) - p))
            t = -num / dn
            if dn < 0 and t > lo:
                lo, lo_n = t, n
            if dn > 0 and t < hi:
                hi, hi_n = t, n
        h = s.intersect_object(0, o32, d32)
        if lo < hi - 1e-3 and hi > 1e-3 and abs(lo) > 1e-3:
            want_t, want_n = (lo, lo_n) if lo > 0 else (hi, hi_n)
            assert h is not None
            assert abs(h[9] - want_t) < 1e-3 * max(1.0, want_t)
            assert np.allclose(h[3:6], want_n, atol=1e-5)          # SpacePartitioning returns +normal (:110-121)
            hits += 1
        elif lo > hi + 1e-3 or hi < -1e-3:
            assert h is None
    assert hits > 1500


# ---- materials: one bounce against numpy restatements --------------------------------------------------

def bounce(kind, m, in7, isect9, seed=5, stream=1, path=99, block=4):
    out = np.zeros(7, dtype=np.float32)
    rc = O.lib().oracle_material_bounce(kind, f32(m[0]), f32(m[1]), f32(m[2]), O.ptr(np.asarray(in7, np.float32)),
                                        O.ptr(np.asarray(isect9, np.float32)), seed, stream, path, block, O.ptr(out))
    assert rc == 0
    return out


def test_glass_refraction_tir_and_unnormalised_direction():
    n_in = np.array([0, 0, 1.0], np.float32)
    d = np.array([0.6, 0.0, -0.8], np.float32)
    isect = [1, 2, 3, *n_in, 0, 0, 0]
    out = bounce(4, (0, 0, 0), [0, 0, 10, *d, 580.0], isect)
    ior = f32(1.0) / np_sf10(580.0)
    cos_i = f32(0.8)
    sin2 = ior * ior * (f32(1) - cos_i * cos_i)
    cos_t = np.sqrt(f32(1) - sin2)
    want = d * ior + n_in * (ior * cos_i - cos_t)
    assert out[3:6].tobytes() == want.astype(np.float32).tobytes() and out[6] == 1.0
    assert tuple(out[:3]) == (1, 2, 3)
    # leaving at a shallow angle: total internal reflection about the flipped normal (material.rs:249-252)
    d = np.array([0.8, 0.0, 0.6], np.float32)
    out = bounce(4, (0, 0, 0), [0, 0, 0, *d, 400.0], isect)
    assert np.allclose(out[3:6], [0.8, 0, -0.6], atol=1e-6)
    # leaving steeply: refracts with ior = n and the direction is NOT re-normalised (material.rs:246)
    d = np.array([0.1, 0.0, 0.99498744], np.float32)
    out = bounce(4, (0, 0, 0), [0, 0, 0, *d, 700.0], isect)
    assert abs(np.linalg.norm(out[3:6].astype(np.float64)) - 1.0) < 1e-5  # Snell keeps unit length only up to rounding


def test_diffuse_family_and_soap_bubble():
    blk = np.zeros(4, dtype=np.uint32)
    O.lib().oracle_rng_block(5, 1, 99, 4, O.ptr(blk))
    u = lambda w: f32(w >> 8) * f32(2.0 ** -24)
    phi = u(blk[0]) * PI * f32(2.0)
    rq = u(blk[1]) * (f32(16777216.0) / f32(16777215.0))
    r = np.sqrt(rq)
    hemi = np.array([f32(np.cos(np.float64(phi))) * r, f32(np.sin(np.float64(phi))) * r, np.sqrt(f32(1) - rq)], np.float32)
    # normal (0,0,1), ray coming down: rotate_towards short-circuits (vector3.rs:73)
    isect = [0, 0, 0, 0, 0, 1, 1, 0, 0]
    d = np.array([0.0, 0.6, -0.8], np.float32)
    out = bounce(1, (0.8, 0, 0), [0, 0, 5, *d, 500.0], isect)
    assert out[3:6].tobytes() == hemi.tobytes() and out[6] == f32(0.8)
    # ray coming up from below: face-forward flips the normal -> mirror z (vector3.rs:76)
    out = bounce(1, (0.8, 0, 0), [0, 0, -5, 0.0, 0.6, 0.8, 500.0], isect)
    assert out[3:6].tobytes() == (hemi * np.array([1, 1, -1], np.float32)).tobytes()
    # coloured: gaussian falloff (material.rs:157-158)
    out = bounce(2, (0.9, 550.0, 40.0), [0, 0, 5, *d, 500.0], isect)
    p = (f32(550.0) - f32(500.0)) / f32(40.0)
    want = f32(0.9) * f32(np.exp(np.float64(f32(-0.5) * p * p)))
    assert ulp_diff(out[6], want) <= 1
    # glossy: blend with the mirror direction, probability 1 (material.rs:185-196)
    out = bounce(3, (0.1, 0, 0), [0, 0, 5, *d, 500.0], isect)
    refl = d - np.array([0, 0, 1], np.float32) * f32(2.0) * d[2]
    v = hemi * f32(0.1) + refl * (f32(1.0) - f32(0.1))
    v = v / np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
    assert ulp_diff(out[3:6], v).max() <= 1 and out[6] == 1.0
    # soap bubble: pass straight through unless grazing; probability in [0.8, 1.0]
    out = bounce(5, (0, 0, 0), [0, 0, 5, *d, 500.0], isect)
    unit = u(blk[0]) * (f32(16777216.0) / f32(16777215.0))
    reflects = unit - f32(0.3) > abs(d[2])
    want_dir = refl if reflects else d
    assert np.allclose(out[3:6], want_dir, atol=1e-7)
    cp = np.clip(np.dot(want_dir, [0, 0, 1]), -0.999, 0.999)
    ct = np.clip(np.dot(want_dir, [1, 0, 0]), -0.999, 0.999)
    phase = (500.0 - 380.0) / 200.0 * math.pi
    want_p = math.cos(phase - math.acos(cp) * 3 - math.acos(ct) * 2 + math.pi * 0.5) * 0.1 + 0.9
    assert abs(out[6] - want_p) < 1e-5 and 0.8 <= out[6] <= 1.0


# ---- plot / gather / tonemap against numpy restatements -------------------------------------------------

def test_plot_pixel_weights_and_clamping():
    W, H = 8, 4
    ph = np.zeros(3, dtype=O.PHOTON_DTYPE)
    ph[0] = (0.0, 0.0, 2.0, 555.0)                 # centre
    ph[1] = (1.0, 0.5, 1.0, 555.0)                 # bottom-right corner: y * aspect = 1
    ph[2] = (-1.0, -0.5, 0.0, 555.0)               # zero probability still "plotted" (adds 0)
    buf = O.plot(W, H, ph).reshape(H, W, 3)
    cie = np.array([0.512050, 1.0, 0.005750], np.float32)
    px, py = f32(0.5) * f32(7), f32(0.5) * f32(3)   # 3.5, 1.5
    assert np.allclose(buf[1, 3], cie * 2 * 0.25) and np.allclose(buf[2, 4], cie * 2 * 0.25)
    assert buf[3, 7].tobytes() == (cie * f32(1.0) * f32(1.0)).tobytes()  # c11 = 1 on the clamped corner
    assert not buf[0, 0].any()
    assert np.isclose(buf[..., 1].sum(), 3.0)


def test_kahan_accumulate_numpy():
    rng = np.random.default_rng(6)
    acc = np.zeros((50, 3), np.float32)
    comp = np.zeros_like(acc)
    a2, c2 = acc.copy(), comp.copy()
    exact = np.zeros((50, 3), np.float64)
    for _ in range(200):
        px = (rng.random((50, 3)) * 1e-3 + 1.0).astype(np.float32)
        O.accumulate(acc, comp, px)
        extra = px - c2
        s = a2 + extra
        c2 = (s - a2) - extra
        a2 = s
        exact += px
    assert acc.tobytes() == a2.tobytes() and comp.tobytes() == c2.tobytes()
    assert np.abs(acc - exact).max() < 2e-5  # compensated: far below the 200 * eps * 200 of a naive sum


def test_tonemap_numpy():
    rng = np.random.default_rng(7)
    W, H = 16, 9
    xyz = (rng.random((W * H, 3)) * 3).astype(np.float32)
    rgb, srgb, mx = O.tonemap(xyz, W, H)
    n = f32(W * H)
    s = f32(0)
    for y in xyz[:, 1]:
        s = s + y
    q = f32(0)
    for y in xyz[:, 1]:
        q = q + y * y
    mean = s / n
    want_mx = mean + np.sqrt(q / n - mean * mean)
    assert f32(mx) == want_mx                       # sequential f32 sums (tonemap_unit.rs:61,64)
    v = np.log(xyz.astype(np.float64) / np.float64(want_mx) + 1.0) / math.log(4.0)
    M = np.array([[3.2406, -1.5372, -0.4986], [-0.9689, 1.8758, 0.0415], [0.0557, -0.2040, 1.0570]])
    lin = v @ M.T
    g = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * np.maximum(lin, 1e-9) ** (1 / 2.4) - 0.055)
    want = np.clip(g, 0, 1)
    assert np.abs(srgb - want).max() < 1e-5
    assert (rgb == (srgb * f32(255.0)).astype(np.uint8)).all()  # truncation (tonemap_unit.rs:96-98)

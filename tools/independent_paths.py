#!/usr/bin/env python3
"""An independent, vectorised numpy-f32 restatement of the reference's path tracer, end to end:
App::set_up_scene (app.rs:166-363), TraceUnit::render / render_camera_ray / render_ray
(trace_unit.rs:81-168), Camera::get_ray (camera.rs:47-108), Scene::intersect (scene.rs:39-60), every
Surface / Volume of geometry.rs (full Intersection records, the recursive Compound<T1, T2>), every
material of material.rs and monte_carlo.rs.

Purpose: oracle/rl_oracle.cpp and the GPU kernel are both compared with the
fixture this script writes (tests/golden/independent_paths.npz), bit for bit.  The script was written from
the Rust sources only -- it shares no code with oracle/rl_oracle.cpp or csrc/rl_core.h.  What it does
share, by necessity, is the build's own definition of the two things the reference leaves undefined:

  * the random numbers: Philox4x32-7 words addressed by (seed, stream, path, block, slot) with the slot
    assignment of csrc/rl_rng.h and rand 0.3.11's u32 -> f32 conversions.  Philox itself is
    re-implemented here in numpy (and checked against the Random123 known-answer vectors below);
  * libm: the per-path sin, cos, exp, acos (f32) and the f64 exp of Planck's law are evaluated by csrc/rl_math.h
    through the oracle library, element-wise on arrays -- "the rl_math.h outputs fed in as arrays" -- when the
    fixture is written (LIBM = "build").  The script also carries its OWN evaluation (LIBM = "platform": numpy's f64
    functions rounded once to f32), used for everything computed once per scene and by `--libm-distance`, which counts
    how many of the fixture's photons change when the build's libm is replaced by it.
    + - * / sqrt are numpy's (IEEE-754, correctly rounded, never fused: every ufunc call is one rounding).

This is still NOT a pin by the reference (the Rust crate cannot be built here and is unseedable); it
removes the single-author risk on the glue of render_ray.

Usage:  python tools/independent_paths.py            # writes tests/golden/independent_paths.npz
        python tools/independent_paths.py --check    # recomputes and compares with the committed fixture
        python tools/independent_paths.py --libm-distance   # photons that differ under a platform libm
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402  (only math_f32 / exp_f64 are used: the shared libm)

F = np.float32
PI = F(np.pi)  # std::f32::consts::PI


# ---- libm ------------------------------------------------------------------------------------------
# LIBM = "build": the per-path transcendentals are the build's own definition (csrc/rl_math.h, evaluated element-wise
# through the oracle's library) -- the fixture tests/golden/independent_paths.npz is computed this way and must be
# reproduced bit for bit by the oracle and the GPU.
# LIBM = "platform": this script's OWN evaluation -- numpy's f64 sin / cos / exp / arccos rounded once to f32, i.e. the
# correctly rounded values that a good platform libm (what Rust's f32::sin etc. call) returns in all but rare cases.
# `--libm-distance` counts the photons that differ between the two: the measured distance between "the build's libm"
# and "a platform libm" (profiles/r04_libm_distance.txt).
# Scene construction (set_up_scene, the prism constructors, the camera's tan) always uses the platform evaluation: the
# build computes its scenes with the f64-evaluated forms (rl_sinf_d / rl_cosf_d / rl_tanf), which are the correctly
# rounded values for every argument a scene uses.
LIBM = "build"


def _m(fn, x):
    x = np.asarray(x, dtype=F)
    return O.math_f32(fn, np.ascontiguousarray(x.reshape(-1))).reshape(x.shape)


def _own(fn, x):
    with np.errstate(all="ignore"):
        return fn(np.asarray(x, dtype=F).astype(np.float64)).astype(F)


def sin(x): return _m("sin", x) if LIBM == "build" else _own(np.sin, x)
def cos(x): return _m("cos", x) if LIBM == "build" else _own(np.cos, x)
def exp(x): return _m("exp", x) if LIBM == "build" else _own(np.exp, x)
def acos(x): return _m("acos", x) if LIBM == "build" else _own(np.arccos, x)
def sin_scene(x): return _own(np.sin, x)
def cos_scene(x): return _own(np.cos, x)
def tan(x): return _own(np.tan, x)


def exp64(x):
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    y = np.zeros_like(x)
    O.lib().oracle_exp_f64(O.ptr(x.reshape(-1)), O.ptr(y.reshape(-1)), x.size)
    return y


def sqrt(x):
    with np.errstate(invalid="ignore"):
        return np.sqrt(np.asarray(x, dtype=F))


# ---- Philox4x32 (Salmon et al., SC'11), numpy: 10 rounds for the published known answers, PHILOX_ROUNDS for the draws --

PHILOX_ROUNDS = 7  # csrc/rl_rng.h: RL_PHILOX_ROUNDS


def philox4x32_10(c0, c1, c2, c3, k0, k1, rounds=10):
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & np.uint64(0xffffffff) for c in (c0, c1, c2, c3))
    mask = np.uint64(0xffffffff)
    sh = np.uint64(32)
    for _ in range(rounds):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> sh, p0 & mask
        hi1, lo1 = p1 >> sh, p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0 = (k0 + W0) & 0xffffffff
        k1 = (k1 + W1) & 0xffffffff
    return [c.astype(np.uint32) for c in (c0, c1, c2, c3)]


def _philox_kat():
    # Random123 kat_vectors: philox4x32 10 rounds
    z = philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(v) for v in z] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    z = philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)
    assert [int(v) for v in z] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    z = philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)
    assert [int(v) for v in z] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def rng_block(seed, stream, path, block):
    """The four 32-bit words of (seed, stream, path, block): csrc/rl_rng.h's addressing."""
    path = np.asarray(path, dtype=np.uint64)
    return philox4x32_10(path & np.uint64(0xffffffff), path >> np.uint64(32), np.full(path.shape, block, dtype=np.uint64),
                         np.full(path.shape, stream, dtype=np.uint64), seed & 0xffffffff, (seed >> 32) & 0xffffffff,
                         rounds=PHILOX_ROUNDS)


def halfopen01(u):  # rand 0.3.11 random::<f32>(): top 24 bits * 2^-24
    return (u >> np.uint32(8)).astype(F) * F(2.0 ** -24)


def closed01(u):    # rand 0.3.11 Closed01<f32>: the above rescaled so that 1.0 is reachable
    return halfopen01(u) * (F(16777216.0) / F(16777215.0))


# monte_carlo.rs:25-43
def get_unit(u): return closed01(u)
def get_bi_unit(u): return closed01(u) * F(2.0) - F(1.0)
def get_longitude(u): return halfopen01(u) * PI * F(2.0)
def get_wavelength(u): return closed01(u) * F(400.0) + F(380.0)


# ---- vector3.rs / quaternion.rs over arrays ---------------------------------------------------------

class V:
    __slots__ = ("x", "y", "z")

    def __init__(self, x, y, z):
        self.x, self.y, self.z = (np.asarray(v, dtype=F) for v in (x, y, z))

    def __add__(self, o): return V(self.x + o.x, self.y + o.y, self.z + o.z)
    def __sub__(self, o): return V(self.x - o.x, self.y - o.y, self.z - o.z)
    def __neg__(self): return V(-self.x, -self.y, -self.z)
    def __mul__(self, f): return V(self.x * f, self.y * f, self.z * f)

    def magnitude_squared(self): return dot(self, self)
    def magnitude(self): return sqrt(self.magnitude_squared())

    def normalise(self):
        m = self.magnitude()
        with np.errstate(divide="ignore", invalid="ignore"):
            return vwhere(m == 0.0, self, V(self.x / m, self.y / m, self.z / m))

    def rotate_towards(self, normal):
        d = normal.z
        up = V(F(0.0), F(0.0), F(1.0))
        a1 = cross(up, normal).normalise()
        a2 = cross(a1, normal).normalise()
        general = a1 * self.x + a2 * self.y + normal * self.z
        mirrored = V(self.x, self.y, -self.z)
        return vwhere(d > F(0.9999), self, vwhere(d < F(-0.9999), mirrored, general))

    def rotate(self, q):
        p = Q(self.x, self.y, self.z, F(0.0))
        r = q * p * q.conjugate()
        return V(r.x, r.y, r.z)

    def reflect(self, normal):
        return self - normal * F(2.0) * dot(normal, self)

    def take(self, idx): return V(self.x[idx], self.y[idx], self.z[idx])

    def broadcast(self, n):
        return V(*(np.broadcast_to(c, (n,)).copy() for c in (self.x, self.y, self.z)))


def vwhere(c, a, b):
    return V(np.where(c, a.x, b.x), np.where(c, a.y, b.y), np.where(c, a.z, b.z))


def cross(a, b):
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x)


def dot(a, b):
    return a.x * b.x + a.y * b.y + a.z * b.z


class Q:
    __slots__ = ("x", "y", "z", "w")

    def __init__(self, x, y, z, w):
        self.x, self.y, self.z, self.w = (np.asarray(v, dtype=F) for v in (x, y, z, w))

    @staticmethod
    def rotation(x, y, z, angle):
        s, c = sin(angle * F(0.5)), cos(angle * F(0.5))
        return Q(s * F(x), s * F(y), s * F(z), c)

    def conjugate(self): return Q(-self.x, -self.y, -self.z, self.w)

    def __mul__(a, b):
        return Q(a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                 a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
                 a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w,
                 a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z)


# ---- geometry.rs ---------------------------------------------------------------------------------------
# An Option<Intersection> for n rays: (some: bool[n], position, normal, tangent: V, distance: f32[n]).

class Isect:
    __slots__ = ("some", "position", "normal", "tangent", "distance")

    def __init__(self, some, position, normal, tangent, distance):
        self.some, self.position, self.normal, self.tangent, self.distance = some, position, normal, tangent, distance

    def filter(self, cond):
        return Isect(self.some & cond, self.position, self.normal, self.tangent, self.distance)


def iwhere(c, a, b):
    return Isect(np.where(c, a.some, b.some), vwhere(c, a.position, b.position), vwhere(c, a.normal, b.normal),
                 vwhere(c, a.tangent, b.tangent), np.where(c, a.distance, b.distance))


ZERO = V(F(0.0), F(0.0), F(0.0))


def intersect_plane(normal, offset, ro, rd):
    origin = ro - offset
    d = dot(normal, rd)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = -dot(normal, origin) / d
    with np.errstate(invalid="ignore"):
        some = (d != 0.0) & ~(t <= 0.0)
    return some, ro + rd * t, t, d


class Plane:
    def __init__(self, normal, offset): self.normal, self.offset = normal, offset

    def intersect(self, ro, rd):
        some, pos, t, d = intersect_plane(self.normal, self.offset, ro, rd)
        return Isect(some, pos, vwhere(d < 0.0, self.normal, -self.normal), ZERO.broadcast(t.shape[0]), t)


class SpacePartitioning:
    def __init__(self, normal, offset): self.normal, self.offset = normal, offset

    def intersect(self, ro, rd):
        some, pos, t, _ = intersect_plane(self.normal, self.offset, ro, rd)
        n = t.shape[0]
        return Isect(some, pos, self.normal.broadcast(n), ZERO.broadcast(n), t)

    def lies_inside(self, p):
        with np.errstate(invalid="ignore"):
            return dot(p - self.offset, self.normal) < 0.0


class Circle:
    def __init__(self, normal, position, radius):
        self.normal, self.position, self.radius_squared = normal, position, F(radius) * F(radius)

    def intersect(self, ro, rd):
        some, pos, t, d = intersect_plane(self.normal, self.position, ro, rd)
        with np.errstate(invalid="ignore"):
            some = some & ((pos - self.position).magnitude_squared() <= self.radius_squared)
        return Isect(some, pos, vwhere(d < 0.0, self.normal, -self.normal), ZERO.broadcast(t.shape[0]), t)


class Sphere:
    def __init__(self, position, radius):
        self.position, self.radius_squared = position, F(radius) * F(radius)

    def intersect(self, ro, rd):
        a = F(1.0)
        centre_offset = self.position - ro
        b = F(2.0) * dot(rd, centre_offset)
        c = centre_offset.magnitude_squared() - self.radius_squared
        discriminant = b * b - F(4.0) * a * c
        with np.errstate(invalid="ignore"):
            has = ~(discriminant < 0.0)
            d = np.sqrt(discriminant)
            t1 = F(-0.5) * (-b + d) / a
            t2 = F(-0.5) * (-b - d) / a
            first = (t1 > 0.0) & (t1 < t2)
            second = (t2 > 0.0) & (t2 < t1)
        t = np.where(first, t1, t2)
        some = has & (first | second)
        position = ro + rd * t
        normal = (position - self.position).normalise()
        up = V(F(0.0), F(1.0), F(0.0))
        tangent = cross(up, normal).normalise()
        return Isect(some, position, normal, tangent, t)


class Paraboloid:
    def __init__(self, normal, offset, focal_distance):
        self.normal = normal
        self.offset = offset - normal * F(focal_distance)
        self.focal_point = normal * (F(focal_distance) * F(2.0))

    def intersect(self, ro, rd):
        origin = ro - self.offset
        focal_offset = origin - self.focal_point
        n_dot_d = dot(self.normal, rd)
        n_dot_o = dot(self.normal, origin)
        d_dot_f = dot(rd, focal_offset)
        a = n_dot_d * n_dot_d - F(1.0)
        b = F(2.0) * n_dot_d * n_dot_o - F(2.0) * d_dot_f
        c = n_dot_o * n_dot_o - focal_offset.magnitude_squared()
        with np.errstate(divide="ignore", invalid="ignore"):
            lin_t = -c / b
            lin_some = ~(lin_t < 0.0)
            d = b * b - F(4.0) * a * c
            quad_has = ~(d < 0.0)
            sqrt_d = np.sqrt(d)
            t1 = F(0.5) * (-b + sqrt_d) / a
            t2 = F(0.5) * (-b - sqrt_d) / a
            pick1 = (t1 > 0.0) & ((t1 < t2) | (t2 < 0.0))
            pick2 = ~pick1 & (t2 > 0.0)
        quad_t = np.where(pick1, t1, t2)
        quad_some = quad_has & (pick1 | pick2)
        linear = a == 0.0
        t = np.where(linear, lin_t, quad_t)
        some = np.where(linear, lin_some, quad_some)
        pos = ro + rd * t
        local_pos = pos - self.offset
        plane_pr = local_pos - self.normal * dot(local_pos, self.normal)
        normal = (self.focal_point - plane_pr).normalise()
        return Isect(some, pos, normal, ZERO.broadcast(t.shape[0]), t)


class Compound:
    def __init__(self, s1, s2): self.surface1, self.surface2 = s1, s2

    def intersect(self, ro, rd):
        i1 = self.surface1.intersect(ro, rd)
        i2 = self.surface2.intersect(ro, rd)
        i1 = i1.filter(self.surface2.lies_inside(i1.position))
        i2 = i2.filter(self.surface1.lies_inside(i2.position))
        both = i1.some & i2.some
        with np.errstate(invalid="ignore"):
            first_nearer = i1.distance < i2.distance
        take1 = np.where(both, first_nearer, i1.some)  # i1.or(i2)
        return iwhere(take1, i1, i2)

    def lies_inside(self, p):
        return self.surface1.lies_inside(p) & self.surface2.lies_inside(p)


def new_infinite_prism(axis, offset, edge_length, angle):
    radius = sqrt(F(3.0)) / F(6.0) * F(edge_length)
    a1 = F(angle)
    a2 = F(angle) + PI * F(2.0) / F(3.0)
    a3 = F(angle) + PI * F(4.0) / F(3.0)
    ps = [V(cos_scene(a), sin_scene(a), F(0.0)).rotate_towards(axis) for a in (a1, a2, a3)]
    sp1, sp2, sp3 = (SpacePartitioning(p, p * radius + offset) for p in ps)
    return Compound(Compound(sp1, sp2), sp3)


def new_thick_plane(normal, offset, thickness):
    return Compound(SpacePartitioning(-normal, offset), SpacePartitioning(normal, offset + normal * F(thickness)))


def new_prism(axis, offset, edge_length, angle, height):
    return Compound(new_infinite_prism(axis, offset, edge_length, angle), new_thick_plane(axis, offset, height))


def new_hexagonal_prism(axis, offset, edge_length, bevel_size, angle, height):
    iprism = new_infinite_prism(axis, offset, F(edge_length) * F(2.0) - F(bevel_size) * F(3.0), F(angle) + PI)
    return Compound(iprism, new_prism(axis, offset, edge_length, angle, height))


# ---- material.rs ---------------------------------------------------------------------------------------

PLANCKS_CONSTANT, BOLTZMANNS_CONSTANT, SPEED_OF_LIGHT, WIENS_CONSTANT = 6.62606957e-34, 1.3806488e-23, 299792458.0, 2.897772126e-3
GOLDEN_RATIO = 1.6180339887498948482045868343656381177203091798057628


def boltzmann(wavelength, temperature):
    wavelength, temperature = np.asarray(wavelength, dtype=np.float64), np.float64(temperature)
    h, k, c = np.float64(PLANCKS_CONSTANT), np.float64(BOLTZMANNS_CONSTANT), np.float64(SPEED_OF_LIGHT)
    f = c / (wavelength * 1.0e-9)
    return (2.0 * h * f * f * f) / (c * c * (exp64(h * f / (k * temperature)) - 1.0))


class BlackBody:
    emissive = True

    def __init__(self, kelvins, intensity):
        self.temperature = F(kelvins)
        peak = boltzmann((np.float64(WIENS_CONSTANT) / np.float64(F(kelvins))) * 1.0e9, np.float64(F(kelvins)))
        self.normalisation_factor = F(intensity) / F(peak)

    def get_intensity(self, wavelength):
        return boltzmann(wavelength.astype(np.float64), np.float64(self.temperature)).astype(F) * self.normalisation_factor


def get_hemisphere_vector(rng):
    phi = get_longitude(rng[0])
    rq = get_unit(rng[1])
    r = sqrt(rq)
    return V(cos(phi) * r, sin(phi) * r, sqrt(F(1.0) - rq))


def get_diffuse_ray(rd, isect, rng):
    hemi_vec = get_hemisphere_vector(rng)
    normal = vwhere(dot(rd, isect.normal) < 0.0, isect.normal, -isect.normal)
    return hemi_vec.rotate_towards(normal)


class DiffuseGrey:
    emissive = False

    def __init__(self, refl): self.reflectance = F(refl)

    def get_new_ray(self, rd, wavelength, isect, rng):
        return get_diffuse_ray(rd, isect, rng), np.full(wavelength.shape, self.reflectance, dtype=F)


class DiffuseColoured:
    emissive = False

    def __init__(self, refl, wavel, dev): self.reflectance, self.wavelength, self.deviation = F(refl), F(wavel), F(dev)

    def get_new_ray(self, rd, wavelength, isect, rng):
        p = (self.wavelength - wavelength) / self.deviation
        q = exp(F(-0.5) * p * p)
        return get_diffuse_ray(rd, isect, rng), self.reflectance * q


class GlossyMirror:
    emissive = False

    def __init__(self, gloss): self.glossiness = F(gloss)

    def get_new_ray(self, rd, wavelength, isect, rng):
        direction = get_diffuse_ray(rd, isect, rng)
        reflection = rd.reflect(isect.normal)
        direction = (direction * self.glossiness + reflection * (F(1.0) - self.glossiness)).normalise()
        return direction, np.full(wavelength.shape, F(1.0), dtype=F)


class Sf10Glass:
    emissive = False

    @staticmethod
    def get_index_of_refraction(wavelength):
        w2 = (wavelength * wavelength * F(1.0e-6)).astype(np.float64)
        return np.sqrt(1.0 + 1.737596950 * w2 / (w2 - 0.0131887070) + 0.313747346 * w2 / (w2 - 0.0623068142)
                       + 1.898781010 * w2 / (w2 - 155.23629000)).astype(F)

    def get_new_ray(self, rd, wavelength, isect, rng):
        cos_i = -dot(rd, isect.normal)
        ior = Sf10Glass.get_index_of_refraction(wavelength)
        entering = cos_i > 0.0
        ior = np.where(entering, F(1.0) / ior, ior)
        normal = vwhere(entering, isect.normal, -isect.normal)
        cos_i = np.where(entering, cos_i, -cos_i)
        sin_t_sqr = ior * ior * (F(1.0) - cos_i * cos_i)
        with np.errstate(invalid="ignore"):
            tir = sin_t_sqr > 1.0
            cos_t = np.sqrt(F(1.0) - sin_t_sqr)
        refracted = rd * ior + normal * (ior * cos_i - cos_t)
        return vwhere(tir, rd.reflect(normal), refracted), np.full(wavelength.shape, F(1.0), dtype=F)


class SoapBubble:
    emissive = False

    def get_new_ray(self, rd, wavelength, isect, rng):
        cos_alpha = dot(rd, isect.normal)
        reflect = get_unit(rng[0]) - F(0.3) > np.abs(cos_alpha)
        direction = vwhere(reflect, rd.reflect(isect.normal), rd)
        phase_shift = (wavelength - F(380.0)) / F(200.0) * PI

        def clamp(x):
            return np.where(x < F(-0.999), F(-0.999), np.where(x > F(0.999), F(0.999), x)).astype(F)
        cos_phi = clamp(dot(direction, isect.normal))
        cos_theta = clamp(dot(direction, isect.tangent))
        p = cos(phase_shift - acos(cos_phi) * F(3.0) - acos(cos_theta) * F(2.0) + PI * F(0.5))
        return direction, p * F(0.1) + F(0.9)


# ---- app.rs:166-363 ------------------------------------------------------------------------------------

def powi2(x): return F(x) * F(x)


def set_up_scene(seeds=100, prism_rings=(17.0,), fixed_only=False):
    """app.rs:166-325.  seeds / prism_rings / fixed_only parametrise the two derived BASELINE scenes exactly as
    SURVEY 8(d) defines them: config 5 = the same generator with seeds = 158; config 3 ("dispersive-glass stress") =
    the seven fixed objects without the spheres plus three rings of prisms built by the recipe of app.rs:287-325 at
    prism_radius 10, 17, 24."""
    objects = []
    sun_radius = F(5.0)
    sun_position = V(F(0), F(0), F(0))
    objects.append((Sphere(sun_position, sun_radius), BlackBody(6504.0, 1.0)))

    floor_normal = V(F(0.0), F(0.0), F(-1.0))
    floor_position = V(F(0.0), F(0.0), -sun_radius)
    floor_paraboloid = Paraboloid(floor_normal, floor_position, powi2(sun_radius))
    objects.append((floor_paraboloid, DiffuseGrey(0.8)))

    up = V(F(0.0), F(0.0), F(1.0))
    objects.append((Paraboloid(up, V(F(1.0), F(0.0), -powi2(sun_radius)), powi2(sun_radius)), DiffuseColoured(0.9, 550.0, 40.0)))
    objects.append((Paraboloid(up, V(F(-1.0), F(0.0), -powi2(sun_radius)), powi2(sun_radius)), DiffuseColoured(0.9, 660.0, 60.0)))

    sky_height = F(30.0)
    sky1_radius = F(5.0)
    objects.append((Circle(floor_normal, V(-sun_radius, F(0.0), sky_height), sky1_radius), BlackBody(7600.0, 0.6)))
    sky2_radius = F(15.0)
    sky2_position = V(-sun_radius * F(0.5), sun_radius * F(2.0) + sky2_radius, sky_height)
    objects.append((Circle(floor_normal, sky2_position, sky2_radius), BlackBody(5000.0, 0.6)))
    objects.append((Plane(floor_normal, V(F(0.0), F(0.0), sky_height * F(2.0))), DiffuseColoured(0.5, 470.0, 25.0)))

    gamma = PI * F(2.0) * (F(1.0) - F(1.0) / F(GOLDEN_RATIO))
    seed_size = F(0.8)
    seed_scale = F(1.5)
    first_seed = int(powi2(sun_radius / seed_scale + F(1.0)) + F(0.5))
    if fixed_only:
        seeds, first_seed_bubbles = 0, None
    for i in range(first_seed, first_seed + seeds):
        phi = F(i) * gamma
        r = sqrt(F(i)) * seed_scale
        position = V(cos_scene(phi) * r, sin_scene(phi) * r, (r - sun_radius) * F(-0.5)) + sun_position
        mat = DiffuseColoured(0.9, F(i - first_seed) / F(seeds) * F(130.0) + F(600.0), 60.0)
        objects.append((Sphere(position, seed_size), mat))
    for i in range(first_seed, first_seed + seeds):
        phi = (F(i) + F(0.5)) * gamma
        r = sqrt(F(i) + F(0.5)) * seed_scale
        position = V(cos_scene(phi) * r, sin_scene(phi) * r, (r - sun_radius) * F(-0.25)) + sun_position
        objects.append((Sphere(position, seed_size * F(0.5)), GlossyMirror(0.1)))
    for i in (() if fixed_only else range(first_seed // 2, first_seed + seeds)):
        phi = F(-i) * gamma
        r = sqrt(F(i)) * seed_scale * F(1.5)
        position = V(cos_scene(phi) * r, sin_scene(phi) * r, (r - sun_radius) * F(1.5) + sun_radius * F(2.0)) + sun_position
        objects.append((Sphere(position, seed_size * (F(0.5) + sqrt(F(i)) * F(0.2))), SoapBubble()))

    prisms = 11
    prism_angle = PI * F(2.0) / F(prisms)
    prism_height = F(8.0)
    for prism_radius, i in ((F(pr), i) for pr in prism_rings for i in range(prisms)):
        for ofs, radius, phi_ofs, h in ((F(0.0), F(1.0), F(0.0), F(1.0)), (F(0.5) * prism_angle, F(1.2), PI * F(0.5), F(1.5))):
            phi = F(i) * prism_angle + ofs
            position = V(cos_scene(phi) * prism_radius * radius, sin_scene(phi) * prism_radius * radius, F(0.0))
            normal = V(F(0.0), F(0.0), F(-1.0))
            hit = floor_paraboloid.intersect(position.broadcast(1), normal.broadcast(1))
            if bool(hit.some[0]):
                normal = -hit.normal.take(0)
                position = hit.position.take(0) + normal * F(2.0) * h
            prism = new_hexagonal_prism(normal, position, 3.0, 1.0, phi + phi_ofs, prism_height * h)
            objects.append((prism, Sf10Glass()))
    return objects


def make_camera(t):
    phi = PI * (F(1.0) + F(0.01) * t)
    alpha = PI * (F(0.3) - F(0.01) * t)
    distance = F(50.0) - F(0.5) * t
    position = V(cos(alpha) * sin(phi) * distance, cos(alpha) * cos(phi) * distance, sin(alpha) * distance)
    orientation = Q.rotation(0.0, 0.0, -1.0, phi + PI) * Q.rotation(1.0, 0.0, 0.0, -alpha)
    return dict(position=position, field_of_view=PI * F(0.35), focal_distance=distance * F(0.9), depth_of_field=F(2.0),
                chromatic_abberation=F(0.012), orientation=orientation)


def get_ray(camera, x, y, wavelength, rng):  # camera.rs:47-108
    dof_angle = get_longitude(rng[0])
    dof_radius = get_unit(rng[1]) / camera["depth_of_field"]
    d = (wavelength - F(580.0)) / F(200.0)
    chromatic_zoom = F(1.0) + d * camera["chromatic_abberation"]
    screen_distance = F(1.0) / tan(camera["field_of_view"] * F(0.5))
    xs = x * chromatic_zoom
    ys = y * chromatic_zoom
    direction = V(xs, np.broadcast_to(screen_distance, xs.shape), -ys).normalise()
    focus_point = direction * (camera["focal_distance"] / direction.y)
    lens_point = V(cos(dof_angle) * dof_radius, np.zeros_like(dof_radius), sin(dof_angle) * dof_radius)
    origin = camera["position"] + lens_point.rotate(camera["orientation"])
    direction = (focus_point - lens_point).rotate(camera["orientation"]).normalise()
    return origin, direction


# ---- trace_unit.rs ---------------------------------------------------------------------------------------

def scene_intersect(objects, ro, rd):  # scene.rs:39-60
    n = ro.x.shape[0]
    distance = np.full(n, F(1.0e12), dtype=F)
    which = np.full(n, -1, dtype=np.int64)
    best = Isect(np.zeros(n, dtype=bool), ZERO.broadcast(n), ZERO.broadcast(n), ZERO.broadcast(n), distance.copy())
    for k, (surface, _) in enumerate(objects):
        isect = surface.intersect(ro, rd)
        with np.errstate(invalid="ignore"):
            nearer = isect.some & (isect.distance < distance)
        best = iwhere(nearer, isect, best)
        distance = np.where(nearer, isect.distance, distance)
        which = np.where(nearer, k, which)
    return which, best


def render(objects, aspect_ratio, seed, stream, first_path, n_paths):
    """TraceUnit::render for paths first_path .. first_path + n_paths.  Returns the MappedPhoton fields
    and the number of Scene::intersect calls."""
    path = np.arange(n_paths, dtype=np.uint64) + np.uint64(first_path)
    b0 = rng_block(seed, stream, path, 0)
    wavelength = get_wavelength(b0[0])
    x = get_bi_unit(b0[1])
    y = get_bi_unit(b0[2]) / F(aspect_ratio)
    t = get_unit(b0[3])                                   # render_camera_ray
    camera = make_camera(t)
    ro, rd = get_ray(camera, x, y, wavelength, rng_block(seed, stream, path, 1))

    probability = np.zeros(n_paths, dtype=F)
    intensity = np.ones(n_paths, dtype=F)
    continue_chance = np.ones(n_paths, dtype=F)
    alive = np.arange(n_paths)
    segments = 0
    bounce = 0
    while alive.size:
        o, d, wl = ro.take(alive), rd.take(alive), wavelength[alive]
        which, isect = scene_intersect(objects, o, d)
        segments += alive.size
        rng = rng_block(seed, stream, path[alive], 2 + bounce)
        n = alive.size
        new_d = V(np.zeros(n, F), np.zeros(n, F), np.zeros(n, F))
        prob = np.ones(n, dtype=F)
        ended = which < 0                                  # The Void: contribution 0
        for k in np.unique(which[which >= 0]):
            material = objects[k][1]
            sel = np.nonzero(which == k)[0]
            if material.emissive:
                probability[alive[sel]] = intensity[alive[sel]] * material.get_intensity(wl[sel])
                ended[sel] = True
                continue
            sub = Isect(isect.some[sel], isect.position.take(sel), isect.normal.take(sel), isect.tangent.take(sel),
                        isect.distance[sel])
            nd, p = material.get_new_ray(d.take(sel), wl[sel], sub, [w[sel] for w in rng])
            new_d.x[sel], new_d.y[sel], new_d.z[sel] = nd.x, nd.y, nd.z
            prob[sel] = p
        cont = ~ended
        a = alive[cont]
        intensity[a] = intensity[a] * prob[cont]
        nd = new_d.take(cont)
        no = isect.position.take(cont) + nd * F(0.00001)
        ro.x[a], ro.y[a], ro.z[a] = no.x, no.y, no.z
        rd.x[a], rd.y[a], rd.z[a] = nd.x, nd.y, nd.z
        continue_chance[a] = continue_chance[a] * F(0.96)
        roulette = get_unit(rng[2][cont]) * F(0.85) > continue_chance[a] * (F(1.0) - exp(intensity[a] * F(-20.0)))
        alive = a[~roulette]                               # killed paths keep probability 0
        bounce += 1
    return x, y, probability, wavelength, segments


CASES = (  # (scene, width, height, seed, stream, first_path, n_paths); scene: 0 demo, 1 glass stress, 2 demo(seeds=158)
    (0, 1280, 720, 1, 0, 0, 16384),
    (0, 1920, 1080, 7, 3, (1 << 33) + 12345, 4096),   # another stream / seed, path indices beyond 32 bits
    (1, 1280, 720, 1, 0, 0, 4096),                    # BASELINE config 3
    (2, 1920, 1080, 1, 2, 1000, 4096),                # BASELINE config 5 (513 objects)
)


def compute():
    _philox_kat()
    scenes = {0: set_up_scene(), 1: set_up_scene(fixed_only=True, prism_rings=(10.0, 17.0, 24.0)), 2: set_up_scene(seeds=158)}
    assert [len(scenes[k]) for k in (0, 1, 2)] == [339, 73, 513]
    out = {}
    for i, (which, w, h, seed, stream, first, n) in enumerate(CASES):
        x, y, p, wl, segs = render(scenes[which], F(w) / F(h), seed, stream, first, n)
        out["case%d_params" % i] = np.array([w, h, seed, stream, first, n, which], dtype=np.uint64)
        out["case%d_x" % i], out["case%d_y" % i], out["case%d_probability" % i], out["case%d_wavelength" % i] = x, y, p, wl
        out["case%d_segments" % i] = np.array([segs], dtype=np.uint64)
    return out


def libm_distance():
    """Photons of the fixture's cases that change when the per-path transcendentals are this script's own evaluation
    (numpy f64, rounded once) instead of the build's rl_math.h."""
    global LIBM
    LIBM = "build"
    a = compute()
    LIBM = "platform"
    b = compute()
    LIBM = "build"
    lines = []
    total = changed = 0
    for i in range(len(CASES)):
        n = a["case%d_x" % i].size
        diff = np.zeros(n, dtype=bool)
        for f in ("x", "y", "probability", "wavelength"):
            diff |= a["case%d_%s" % (i, f)].view(np.uint32) != b["case%d_%s" % (i, f)].view(np.uint32)
        lines.append("case %d (scene %d, %d photons): %d photons differ, segments %d vs %d" % (
            i, CASES[i][0], n, int(diff.sum()), int(a["case%d_segments" % i][0]), int(b["case%d_segments" % i][0])))
        total += n
        changed += int(diff.sum())
    lines.append("total: %d of %d photons differ (%.3f %%) between the build's libm (csrc/rl_math.h) and numpy's f64 "
                 "functions rounded once to f32" % (changed, total, 100.0 * changed / total))
    return lines


if __name__ == "__main__":
    target = os.path.join(ROOT, "tests", "golden", "independent_paths.npz")
    if "--libm-distance" in sys.argv:
        print("\n".join(libm_distance()))
        sys.exit(0)
    got = compute()
    if "--check" in sys.argv:
        want = np.load(target)
        for k in want.files:
            assert want[k].tobytes() == got[k].tobytes(), k
        print("independent_paths.npz reproduced bit for bit")
    else:
        np.savez_compressed(target, **got)
        print("wrote", target, {k: v.shape for k, v in got.items()})

// rl_core.h -- the per-path arithmetic of the trace kernel: camera ray, primitive scan, hit
// completion, material bounce, Russian roulette, CIE lookup and splat weights.
//
// This is the device code.  It is written against RlSceneView (flat 16-byte records) instead of the
// reference's trait objects and is restructured for a 64-wide wave:
//   * the scan carries only (distance, id) and completes position/normal/tangent for the single
//     winning primitive afterwards (the reference builds a full Intersection per candidate,
//     geometry.rs:242-259);
//   * the sphere test drops the reference's factors of two (b = 2 d.co, disc = b^2 - 4c, t = -(−b±√disc)/2,
//     geometry.rs:204-221) -- scaling by powers of two is exact in binary floating point, so
//     q = (d.co)^2 - c, t = d.co -/+ sqrt(q) gives bit-identical distances with 5 fewer operations;
//   * a hexagonal prism (geometry.rs:493-515: Compound<InfinitePrism, Compound<InfinitePrism,
//     ThickPlane>>) is evaluated as its fixed tree over 8 half-space records.
// Every value that feeds a branch or the result is computed with the reference's operation order
// and no fused multiply-add (build with -ffp-contract=off), so a path takes the same branches here
// as in any IEEE-754 restatement of the Rust source.
//
// The functions are RL_HD only so that tests/host_mirror can compile this same file with g++ and
// compare it with the oracle without a GPU.  The product never runs them on the CPU.
#pragma once
#include "rl_math.h"
#include "rl_rng.h"
#include "rl_scene.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "rl_core.h's device fast paths (64-bit ballots, two scratch slots per lane, ds_bpermute batches) are written for wave64 gfx950"
#endif

struct RlF3 {
    float x, y, z;
};
RL_HD RlF3 rl_f3(float x, float y, float z) {
    RlF3 r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
RL_HD RlF3 rl_add(RlF3 a, RlF3 b) { return rl_f3(a.x + b.x, a.y + b.y, a.z + b.z); }
RL_HD RlF3 rl_sub(RlF3 a, RlF3 b) { return rl_f3(a.x - b.x, a.y - b.y, a.z - b.z); }
RL_HD RlF3 rl_neg(RlF3 a) { return rl_f3(-a.x, -a.y, -a.z); }
RL_HD RlF3 rl_mul(RlF3 a, float f) { return rl_f3(a.x * f, a.y * f, a.z * f); }
RL_HD float rl_dot(RlF3 a, RlF3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                    // vector3.rs:35-37
RL_HD RlF3 rl_cross(RlF3 a, RlF3 b) {                                                                // vector3.rs:27-33
    return rl_f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// Two one-operand IEEE divisions in three operations each, where the wave's arguments are normal floats with 2^-100 <= |x| < 2^100
// (else the compiler's 11-instruction division): 1 / x = y + (1 - x y) y with y = v_rcp_f32(x) (material.rs:224), and
// x / 200 = q + (x - 200 q) c with c = fl(1 / 200), q = x c (material.rs:283, camera.rs:100).  Like rl_sqrtf's short form these ARE
// the correctly rounded quotients for every such float of either sign on gfx950 -- tools/sqrt_exhaustive.hip compares all
// 3,355,443,200 of them with the compiler's divisions (0 differ), and the GPU tests repeat it through the library.
RL_HD float rl_recipf(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const bool in_range = (rl_f2u(x) & 0x7fffffffu) - 0x0d800000u < 0x71800000u - 0x0d800000u;
    if (RL_LIKELY(__builtin_amdgcn_ballot_w64(!in_range) == 0)) {
        const float y = __builtin_amdgcn_rcpf(x);
        return __builtin_fmaf(__builtin_fmaf(-x, y, 1.0f), y, y);
    }
#endif
    return 1.0f / x;
}
RL_HD float rl_div200f(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const bool in_range = (rl_f2u(x) & 0x7fffffffu) - 0x0d800000u < 0x71800000u - 0x0d800000u;
    if (RL_LIKELY(__builtin_amdgcn_ballot_w64(!in_range) == 0)) {
        const float c = 0.005f, q = x * c;
        return __builtin_fmaf(__builtin_fmaf(-200.0f, q, x), c, q);
    }
#endif
    return x / 200.0f;
}

RL_HD RlF3 rl_normalise(RlF3 v) {                                                                    // vector3.rs:56-67
    const float d2 = rl_dot(v, v);
#if defined(__HIP_DEVICE_COMPILE__)
    // The square root in its short form (rl_sqrtf) and three IEEE divisions by the same m.  The compiler's division is
    // v_div_scale x 2, v_rcp, a refinement of the reciprocal, the quotient with two residual corrections, v_div_fmas,
    // v_div_fixup (11 instructions, 33 for a vector).
    // Everything that depends on the divisor alone is shared here, and the scaling / fix-up steps are left out where
    // they do nothing: v_div_scale passes its operands through and v_div_fixup its quotient when the divisor is a normal
    // number far from the ends of the range and the numerator is zero or not tiny against it (the conditions below,
    // from the instruction's definition).  What remains is the very same sequence of correctly rounded operations --
    // rcp, e = fma(-m, y, 1), y = fma(e, y, y), q = x y, r = fma(-m, q, x), q = fma(r, y, q), r = fma(-m, q, x),
    // q = fma(r, y, q) -- so the quotients are the compiler's bit for bit (the parity tests compare them with the g++
    // build's divisions); a zero keeps its sign.  Any other operand in the wave: the plain forms below.
    // (round 6: 2|x| - 1 instead of |x| - 1 -- the shift drops the sign, so each component costs one v_lshl_add_u32 instead of a
    // v_and and a v_add; for 1 <= |x| bits: 2|x| - 1 >= 2K - 1 <=> |x| >= K, and a zero still wraps to the largest value)
    const uint32_t bx = (rl_f2u(v.x) << 1) - 1u, by = (rl_f2u(v.y) << 1) - 1u, bz = (rl_f2u(v.z) << 1) - 1u;
    uint32_t least = bx < by ? bx : by;
    least = bz < least ? bz : least;
    const bool plain = (least >= 2u * 0x1f800000u - 1u)                  // every component is 0 or at least 2^-64
                       & (rl_f2u(d2) - 0x0f800000u < 0x7a800000u - 0x0f800000u); // 2^-96 <= |v|^2 < 2^118: 2^-48 <= m < 2^59 (NaN and 0 fail)
    if (RL_LIKELY(__builtin_amdgcn_ballot_w64(!plain) == 0)) {
        const float ys = __builtin_amdgcn_rsqf(d2);
        const float s = d2 * ys, h = 0.5f * ys;
        const float m = __builtin_fmaf(__builtin_fmaf(-s, s, d2), h, s); // = sqrtf(d2), see rl_sqrtf
        float y = __builtin_amdgcn_rcpf(m);
        y = __builtin_fmaf(__builtin_fmaf(-m, y, 1.0f), y, y);
        auto quotient = [&](float x) {
            float q = x * y;
            q = __builtin_fmaf(__builtin_fmaf(-m, q, x), y, q);
            q = __builtin_fmaf(__builtin_fmaf(-m, q, x), y, q);
            return rl_u2f((rl_f2u(q) & 0x7fffffffu) | (rl_f2u(x) & 0x80000000u)); // m > 0: the quotient has x's sign, a zero too
        };
        return rl_f3(quotient(v.x), quotient(v.y), quotient(v.z));
    }
#endif
    const float m = sqrtf(d2);
    const RlF3 u = rl_f3(v.x / m, v.y / m, v.z / m); // inf/NaN for m == 0, discarded below
    return (m == 0.0f) ? v : u;
}
RL_HD RlF3 rl_reflect(RlF3 v, RlF3 n) { return rl_sub(v, rl_mul(rl_mul(n, 2.0f), rl_dot(n, v))); }   // vector3.rs:91-93
RL_HD RlF3 rl_rotate_towards(RlF3 v, RlF3 n, RlF3 a1);
RL_HD RlF3 rl_rotate_towards(RlF3 v, RlF3 n) {                                                       // vector3.rs:69-83
    return rl_rotate_towards(v, n, rl_normalise(rl_cross(rl_f3(0.0f, 0.0f, 1.0f), n)));
}
// vector3.rs:69-83 with the first axis, a1 = normalise(cross((0, 0, 1), n)), supplied by the caller (rl_bounce
// shares that normalisation with the soap bubble's tangent).
RL_HD RlF3 rl_rotate_towards(RlF3 v, RlF3 n, RlF3 a1) {
    if (n.z > 0.9999f) return v;
    if (n.z < -0.9999f) return rl_f3(v.x, v.y, -v.z);
    const RlF3 a2 = rl_normalise(rl_cross(a1, n));
    return rl_add(rl_add(rl_mul(a1, v.x), rl_mul(a2, v.y)), rl_mul(n, v.z));
}
RL_HD RlF3 rl_xyz(RlF4 v) { return rl_f3(v.x, v.y, v.z); }

struct RlQuat {
    float x, y, z, w;
};
RL_HD RlQuat rl_qmul(RlQuat a, RlQuat b) {                                                           // quaternion.rs:100-111
    RlQuat r;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    return r;
}
RL_HD RlF3 rl_qrotate(RlF3 v, RlQuat q) {                                                            // vector3.rs:85-89
    RlQuat p;
    p.x = v.x; p.y = v.y; p.z = v.z; p.w = 0.0f;
    RlQuat c;
    c.x = -q.x; c.y = -q.y; c.z = -q.z; c.w = q.w;
    const RlQuat r = rl_qmul(rl_qmul(q, p), c);
    return rl_f3(r.x, r.y, r.z);
}

// ---- state of one path ------------------------------------------------------------------------

struct RlPath {
    RlF3 origin, direction;  // ray.rs:19-33 (probability folded into `intensity`)
    float wavelength;
    float intensity;         // trace_unit.rs:88
    float continue_chance;   // trace_unit.rs:84
    float sx, sy;            // screen position, trace_unit.rs:157-158
    float ior;               // SF10 index of refraction at `wavelength` (material.rs:203-213): a function of
                             // the path's wavelength only, so it is evaluated once per path, not per bounce
    uint32_t bounce;         // RNG block = 2 + bounce
};

enum { RL_HIT_NONE = 0xffffffffu };

struct RlHit {
    float t;
    uint32_t obj; // object index or RL_HIT_NONE
    uint32_t sub; // prism: which of the 8 half-spaces
};

// material.rs:203-213.  The reference evaluates the Sellmeier sum in f64 -- three divisions and a square root, ~100 f64
// operations on this chip, once per path -- and rounds to f32.  On the GPU the same f32 is DECIDED from a cheaper evaluation
// whenever it can be: the three fractions over one denominator, reciprocal and inverse square root from the hardware's
// approximations refined by two Newton steps each (~40 operations, relative error far below 2^-40; the reference's own
// rounding errors are ~2^-51), accepted only if every value within 2^-40 of it rounds to the same float -- then the reference's
// double does too.  Otherwise (one wavelength in ~10^4 is that close to a rounding boundary; NaN, infinities and a zero
// denominator never compare equal) the wave evaluates the reference's expression.  Checked against it for EVERY f32 wavelength of
// [380, 780] and a range of others on the device (tests/test_gpu_parity.py).
RL_HD float rl_sf10_ior(float wavelength) {
    const double w2 = (double)(wavelength * wavelength * 1.0e-6f);
#if defined(__HIP_DEVICE_COMPILE__)
    {
        const double t1 = w2 - 0.0131887070, t2 = w2 - 0.0623068142, t3 = w2 - 155.23629000;
        const double t23 = t2 * t3, t13 = t1 * t3, t12 = t1 * t2;
        const double num = __builtin_fma(1.737596950 * w2, t23, __builtin_fma(0.313747346 * w2, t13, (1.898781010 * w2) * t12));
        const double den = t1 * t23;
        double y = __builtin_amdgcn_rcp(den);
        y = __builtin_fma(y, __builtin_fma(-den, y, 1.0), y);
        y = __builtin_fma(y, __builtin_fma(-den, y, 1.0), y);
        const double sum = 1.0 + num * y;
        double r = __builtin_amdgcn_rsq(sum);
        r = __builtin_fma(0.5 * r, __builtin_fma(-(sum * r), r, 1.0), r);
        r = __builtin_fma(0.5 * r, __builtin_fma(-(sum * r), r, 1.0), r);
        const double n = sum * r;
        const float lo = (float)(n * (1.0 - 9.094947017729282e-13)), hi = (float)(n * (1.0 + 9.094947017729282e-13)); // 2^-40
        if (RL_LIKELY(__builtin_amdgcn_ballot_w64(!(lo == hi)) == 0)) return lo;
    }
#endif
    return (float)sqrt(1.0 + 1.737596950 * w2 / (w2 - 0.0131887070) + 0.313747346 * w2 / (w2 - 0.0623068142) +
                       1.898781010 * w2 / (w2 - 155.23629000));
}

// ---- camera: trace_unit.rs:151-148, app.rs:327-357, camera.rs:47-108 --------------------------

RL_HD void rl_begin_path(const RlSceneView& sv, float aspect_ratio, uint64_t seed, uint32_t stream, uint64_t path,
                         RlPath* p) {
    const RlRngBlock b0 = rl_rng_block(seed, stream, path, 0);
    const float wavelength = rl_get_wavelength(b0.w[0]);
    const float x = rl_get_bi_unit(b0.w[1]);
    const float y = rl_get_bi_unit(b0.w[2]) / aspect_ratio;
    const float t = rl_get_unit(b0.w[3]);

    // make_camera(t)
    const float* crec = (const float*)sv.camera_rec;
    RlCameraDesc cd;
    cd.phi0 = crec[0]; cd.phi1 = crec[1]; cd.alpha0 = crec[2]; cd.alpha1 = crec[3]; cd.dist0 = crec[4]; cd.dist1 = crec[5];
    cd.fov_over_pi = crec[6]; cd.focal_factor = crec[7]; cd.depth_of_field = crec[8]; cd.chromatic_abberation = crec[9];
    const float screen_distance = crec[10];
    const float phi = RL_PI_F * (cd.phi0 + cd.phi1 * t);
    const float alpha = RL_PI_F * (cd.alpha0 + cd.alpha1 * t);
    const float distance = cd.dist0 + cd.dist1 * t;
    float sin_a, cos_a, sin_p, cos_p;
    rl_sincosf(alpha, &sin_a, &cos_a);
    rl_sincosf(phi, &sin_p, &cos_p);
    const RlF3 position = rl_f3(cos_a * sin_p * distance, cos_a * cos_p * distance, sin_a * distance);
    float s1, c1, s2, c2;
    rl_sincosf((phi + RL_PI_F) * 0.5f, &s1, &c1);
    rl_sincosf((-alpha) * 0.5f, &s2, &c2);
    RlQuat q1, q2;
    q1.x = s1 * 0.0f; q1.y = s1 * 0.0f; q1.z = s1 * -1.0f; q1.w = c1;   // rotation(0, 0, -1, phi + PI)
    q2.x = s2 * 1.0f; q2.y = s2 * 0.0f; q2.z = s2 * 0.0f;  q2.w = c2;   // rotation(1, 0, 0, -alpha)
    const RlQuat orientation = rl_qmul(q1, q2);
    const float focal_distance = distance * cd.focal_factor;

    // Camera::get_ray
    const RlRngBlock b1 = rl_rng_block(seed, stream, path, 1);
    const float dof_angle = rl_get_longitude(b1.w[0]);
    const float dof_radius = rl_get_unit(b1.w[1]) / cd.depth_of_field;
    const float d = rl_div200f(wavelength - 580.0f);
    const float zoom = 1.0f + d * cd.chromatic_abberation;

    // Camera::get_screen_ray
    const float xs = x * zoom;
    const float ys = y * zoom;
    const RlF3 direction = rl_normalise(rl_f3(xs, screen_distance, -ys));
    const RlF3 focus_point = rl_mul(direction, focal_distance / direction.y);
    float sin_d, cos_d;
    rl_sincosf(dof_angle, &sin_d, &cos_d);
    const RlF3 lens_point = rl_f3(cos_d * dof_radius, 0.0f, sin_d * dof_radius);

    p->origin = rl_add(position, rl_qrotate(lens_point, orientation));
    p->direction = rl_normalise(rl_qrotate(rl_sub(focus_point, lens_point), orientation));
    p->wavelength = wavelength;
    p->intensity = 1.0f;
    p->continue_chance = 1.0f;
    p->sx = x;
    p->sy = y;
    p->ior = rl_sf10_ior(wavelength);
    p->bounce = 0;
}

// ---- the scan: scene.rs:39-60 over the flat records --------------------------------------------

// geometry.rs:55-71 without the position.  Returns t > 0 or a negative number for "no hit".
template <bool AXIS_Z = false> // (AXIS_Z: see rl_paraboloid_t)
RL_HD float rl_plane_t(RlF3 n, RlF3 off, RlF3 o, RlF3 dir, float* dn_out) {
    const RlF3 lo = rl_sub(o, off);
    const float dn = AXIS_Z ? n.z * dir.z : rl_dot(n, dir);
    *dn_out = dn;
    // Branch-free: for dn == 0 the quotient is inf/NaN and is discarded; `t > 0` is false for NaN.
    const float t = -(AXIS_Z ? n.z * lo.z : rl_dot(n, lo)) / dn;
    return (dn != 0.0f && t > 0.0f) ? t : -1.0f;
}
// The same as a predicate (round 6, the kernel's straight-line block): "t > 0 and the ray is not parallel" without the detour through
// a -1 that the caller compares with zero again; *t_out is the quotient whatever the answer.
template <bool AXIS_Z = false>
RL_HD bool rl_plane_hit(RlF3 n, RlF3 off, RlF3 o, RlF3 dir, float* t_out) {
    const RlF3 lo = rl_sub(o, off);
    const float dn = AXIS_Z ? n.z * dir.z : rl_dot(n, dir);
    const float t = -(AXIS_Z ? n.z * lo.z : rl_dot(n, lo)) / dn;
    *t_out = t;
    return (dn != 0.0f) & (t > 0.0f);
}
// geometry.rs:123-127
RL_HD bool rl_inside(RlF3 n, RlF3 off, RlF3 p) { return rl_dot(rl_sub(p, off), n) < 0.0f; }

// geometry.rs:298-341: the distance only.
RL_HD float rl_paraboloid_roots(float a, float b, float c) { // geometry.rs:316-341 given the quadratic's coefficients
    if (a == 0.0f) {
        const float t1 = -c / b;
        if (t1 < 0.0f) return -1.0f;
        return t1; // may be +0 or NaN exactly as in the reference
    }
    const float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return -1.0f;
    const float sq = sqrtf(disc);
    const float p = 0.5f * (-b + sq) / a;
    const float q = 0.5f * (-b - sq) / a;
    if (p > 0.0f && (p < q || q < 0.0f)) return p;
    if (q > 0.0f) return q;
    return -1.0f;
}
// AXIS_Z (device, round 6): the caller knows that normal.x and normal.y are zeros -- every paraboloid, plane and circle of the built-in
// room has its normal along z (app.rs:179-231), RlFlatScene::small_axis_z -- and the two dot products with the normal are one
// product each.  rl_dot(normal, v) = (0 v.x + 0 v.y) + normal.z v.z IS normal.z v.z for finite v, except for the SIGN of a zero
// result (the sum of the two zero products may be +0 where normal.z v.z is -0).  A zero n.d or n.o cannot change what is returned:
// n.d enters squared (a) and through 2 n.d n.o - 2 d.f, whose sign matters only where it and the discriminant's root are both zero --
// then np = -b + 0 is +0 for either sign of b, never < 0, and nq = +-0 is not < 0 either: no hit both ways; a plane with
// n.d = +-0 is rejected by `dn != 0`, and with n.lo = +-0 its t = -+0 fails `t > 0` both ways.
// hit_out (device): where given, *hit_out says whether the returned distance is a hit, and a miss may return any number -- the
// caller then needs no `!(t < 0)` of its own (the straight-line block of rl_scan_wave).
template <bool AXIS_Z = false>
RL_HD float rl_paraboloid_t(RlF3 offset, RlF3 normal, RlF3 focal_point, RlF3 o, RlF3 dir, bool* hit_out = nullptr) {
    const RlF3 origin = rl_sub(o, offset);
    const RlF3 focal_offset = rl_sub(origin, focal_point);
    const float n_dot_d = AXIS_Z ? normal.z * dir.z : rl_dot(normal, dir);
    const float n_dot_o = AXIS_Z ? normal.z * origin.z : rl_dot(normal, origin);
    const float d_dot_f = rl_dot(dir, focal_offset);
    const float a = n_dot_d * n_dot_d - 1.0f;
    const float b = 2.0f * n_dot_d * n_dot_o - 2.0f * d_dot_f;
    const float c = n_dot_o * n_dot_o - rl_dot(focal_offset, focal_offset);
#if defined(__HIP_DEVICE_COMPILE__)
    // One division instead of two.  For a < 0 (|normal.direction| < 1: every ray but one along the axis) dividing by a
    // reverses the order and the signs of the two numerators np >= nq: p <= q, p > 0 iff np < 0, q > 0 iff nq < 0, so the
    // reference's selection (geometry.rs:327-341) returns p when np < 0 (if p == q it returns q, the same number), else q
    // when nq < 0, else nothing -- the quotient of ONE numerator, chosen by sign.  Signs survive the division unless a
    // numerator is so small that half of it is no float (|n| < 2^-125, 0 excepted); those waves, and the ones holding a
    // ray with a >= 0 or a NaN, take the literal form below (wave-uniform).
    const float disc = b * b - 4.0f * a * c;
    const float sq = rl_sqrtf(disc, true); // (a negative discriminant is a miss whatever `sq` is)
    const float np = -b + sq, nq = -b - sq;
    const float pick = np < 0.0f ? np : nq;
    // (one unsigned compare per numerator: |n| >= 2^-123 or n == 0 -- the zero wraps round; an exact zero is common, a ray that
    // leaves the paraboloid has c ~ 1e-5 and its discriminant rounds to b^2.  A NaN passes as well and gives the literal form's
    // "no hit" here too: every comparison with it is false in both.  The conditions are combined with | and &, not || and &&:
    // as written with those the compiler built a tree of a dozen exec-mask branches around eight compares, ~45 scalar
    // instructions per paraboloid, and a wave's time goes into issuing instructions whatever their kind)
    // Round 6: the same guarantee from ONE compare.  If max(|b|, sq) >= 2^-90, each numerator is zero or at least 2^-114: a sum of
    // two floats is a multiple of the smaller one's ulp, so a non-zero -b +- sq is either dominated by the larger operand (>= 2^-91)
    // or a cancellation between two operands that are both >= 2^-91, i.e. a multiple of 2^-114.  (A NaN among them fails the fast
    // path's `pick < 0` as it fails every compare of the literal form: no hit both ways.)
#ifndef RL_PARAB_CHECK_OLD
    float largest; // max(|b|, sq) -- spelled out: fmaxf comes with a canonicalising v_max_f32 x, x in front (see rl_hex_prism_fast)
    asm("v_max_f32 %0, |%1|, %2" : "=v"(largest) : "v"(b), "v"(sq));
    const bool plain = (a < 0.0f) & ((disc < 0.0f) | (largest >= 8.0779356694631609e-28f)); // 2^-90
#else
    const uint32_t up = (rl_f2u(np) & 0x7fffffffu) - 1u, uq = (rl_f2u(nq) & 0x7fffffffu) - 1u;
    const bool plain = (a < 0.0f) & ((disc < 0.0f) | ((up >= 0x02000000u - 1u) & (uq >= 0x02000000u - 1u)));
#endif
    if (RL_LIKELY(__builtin_amdgcn_ballot_w64(!plain) == 0)) {
        const float t = 0.5f * pick / a; // (for every lane: the quotient of a miss is discarded -- no branch around the division)
        if (hit_out) {
            *hit_out = !(disc < 0.0f) & (pick < 0.0f); // (then t = (pick / 2) / a with both negative: not below zero)
            return t;
        }
        return ((disc < 0.0f) | !(pick < 0.0f)) ? -1.0f : t;
    }
#endif
    const float t_literal = rl_paraboloid_roots(a, b, c);
    if (hit_out) *hit_out = !(t_literal < 0.0f);
    return t_literal;
}

// One candidate of a Compound: distance and which half-space.  Inside rl_hex_prism "None" is
// t = +infinity, which makes Compound::intersect's selection a single compare; the function returns
// t < 0 for None.
struct RlCand {
    float t;
    uint32_t k;
};

// Compound::intersect's selection (geometry.rs:386-399) given both children's (already filtered)
// candidates: nearest wins, tie -> second child; None = +inf loses against anything.
RL_HD RlCand rl_compound_pick(RlCand a, RlCand b) {
    const bool take_a = a.t < b.t;
    RlCand r;
    r.t = take_a ? a.t : b.t;
    r.k = take_a ? a.k : b.k;
    return r;
}

// HexagonalPrism = Compound<InfinitePrism[0,1,2], Compound<InfinitePrism[3,4,5], ThickPlane[6,7]>>
// with InfinitePrism[a,b,c] = Compound<Compound<a,b>,c> (geometry.rs:409-416).  pr points at the
// prism's 16 half-space records.
#if defined(__HIP_DEVICE_COMPILE__)
// The device's form of the tree below: the same operations on the same values, but nothing of a half-space is kept in registers
// between its uses -- every plane test and every inside test loads its normal and offset again (through an index the optimiser
// cannot see through, or it would merge the loads and keep all 48 values live as the literal form does: the tree then sets the
// register count of the whole trace kernel, 56 registers for something 3.6 % of the prism rounds reach).  Twice the loads, a fifth
// of the registers; what a round that evaluates the tree costs is immaterial next to what a register costs every wave of the
// kernel (occupancy).
__device__ __forceinline__ RlCand rl_hex_prism(const RlF4* pr, RlF3 o, RlF3 dir) {
    const float NONE = __builtin_inff();
    float t[8];
    auto plane = [&](int k, RlF3* n, RlF3* off) {
        uint32_t kk = 2u * (uint32_t)k;
        asm volatile("" : "+v"(kk));
        *n = rl_xyz(pr[kk]);
        *off = rl_xyz(pr[kk + 1u]);
    };
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        RlF3 n, off;
        plane(k, &n, &off);
        float dn;
        const float tk = rl_plane_t(n, off, o, dir, &dn);
        t[k] = tk > 0.0f ? tk : NONE;
    }
    auto filt = [&](RlCand c, uint32_t mask) -> RlCand {
        const RlF3 pos = rl_add(o, rl_mul(dir, c.t));
        bool in = true;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (mask & (1u << j)) {
                RlF3 n, off;
                plane(j, &n, &off);
                in = in & rl_inside(n, off, pos);
            }
        if (!in) c.t = NONE;
        return c;
    };
    auto leaf = [&](int k) -> RlCand {
        RlCand c;
        c.t = t[k];
        c.k = (uint32_t)k;
        return c;
    };
    auto inf_prism = [&](int a, int b, int c) -> RlCand {
        const RlCand ab = rl_compound_pick(filt(leaf(a), 1u << b), filt(leaf(b), 1u << a));
        return rl_compound_pick(filt(ab, 1u << c), filt(leaf(c), (1u << a) | (1u << b)));
    };
    const RlCand ip_bevel = inf_prism(0, 1, 2);
    const RlCand ip_main = inf_prism(3, 4, 5);
    const RlCand thick = rl_compound_pick(filt(leaf(6), 1u << 7), filt(leaf(7), 1u << 6));
    const RlCand prism = rl_compound_pick(filt(ip_main, 0xC0u), filt(thick, 0x38u));
    RlCand hit = rl_compound_pick(filt(ip_bevel, 0xF8u), filt(prism, 0x07u));
    if (!(hit.t < NONE)) hit.t = -1.0f;
    return hit;
}
#else
RL_HD RlCand rl_hex_prism(const RlF4* pr, RlF3 o, RlF3 dir) {
    const float NONE = __builtin_inff();
    RlF3 n[8], off[8];
    float t[8];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 0; k < 8; ++k) {
        n[k] = rl_xyz(pr[2 * k]);
        off[k] = rl_xyz(pr[2 * k + 1]);
        float dn;
        const float tk = rl_plane_t(n[k], off[k], o, dir, &dn);
        t[k] = tk > 0.0f ? tk : NONE;
    }
    // A candidate of half-space k survives a filter against the set `mask` when its position lies
    // inside every half-space of the set (Compound::intersect's lies_inside filters, geometry.rs:383-384).
    auto filt = [&](RlCand c, uint32_t mask) -> RlCand {
        // No early-out for None: with t = +inf the position is inf/NaN and whatever the tests say, the
        // candidate stays +inf -- on a 64-wide wave a branch would run both sides anyway.
        const RlF3 pos = rl_add(o, rl_mul(dir, c.t));
        bool in = true;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int j = 0; j < 8; ++j)
            if (mask & (1u << j)) in = in && rl_inside(n[j], off[j], pos);
        if (!in) c.t = NONE;
        return c;
    };
    auto leaf = [&](int k) -> RlCand {
        RlCand c;
        c.t = t[k];
        c.k = (uint32_t)k;
        return c;
    };
    auto inf_prism = [&](int a, int b, int c) -> RlCand {
        const RlCand ab = rl_compound_pick(filt(leaf(a), 1u << b), filt(leaf(b), 1u << a));
        return rl_compound_pick(filt(ab, 1u << c), filt(leaf(c), (1u << a) | (1u << b)));
    };
    const RlCand ip_bevel = inf_prism(0, 1, 2);
    const RlCand ip_main = inf_prism(3, 4, 5);
    const RlCand thick = rl_compound_pick(filt(leaf(6), 1u << 7), filt(leaf(7), 1u << 6));
    const RlCand prism = rl_compound_pick(filt(ip_main, 0xC0u), filt(thick, 0x38u));
    RlCand hit = rl_compound_pick(filt(ip_bevel, 0xF8u), filt(prism, 0x07u));
    if (!(hit.t < NONE)) hit.t = -1.0f;
    return hit;
}
#endif

// ---- the same result without walking the tree ----------------------------------------------------------------------
// rl_hex_prism above costs ~600 instructions: 8 plane tests with an IEEE division each and 25 inside tests.  The prism
// is a convex polytope, so the tree's answer is almost always the obvious one -- the last half-space the ray enters (if
// it then still is inside all the others) or, from inside, the first one it leaves -- and whenever every decision the
// tree would take is FAR from its threshold, that answer can be PROVED to be the tree's without evaluating the tree:
//
//   s_i(t) = n_i.(o + t d - off_i) = nm_i + t dn_i is what the tree's inside tests evaluate (rl_inside, < 0 = inside);
//   t_k = -nm_k / dn_k (the reference's expression, geometry.rs:62) is where the ray crosses plane k: entering if
//   dn_k < 0, leaving if dn_k > 0.  A candidate (a crossing with t_k > 0) reaches the root of the tree iff it passes the
//   inside tests of all seven other half-spaces on its way up, and loses a pick only to a candidate that is not farther
//   (rl_compound_pick).  With t_in = max entering t_k and t_out = min leaving t_k:
//     hit:  t_out > max(t_in, 0).  k* = the entering plane of t_in if t_in > 0, else the leaving plane of t_out.  If
//           (a) the point at t* is inside every other half-space j by a margin, |dn_j| |t_j - t*| > delta, and
//           (b) every other candidate that is not farther than t* (an entering plane with 0 < t_j < t_in) is outside
//               k*'s half-space by a margin when it gets there, |dn_k*| (t* - t_j) > delta,
//           then k* passes all its tests, every candidate it meets in a pick is either None by then or strictly
//           farther, and the tree returns (t_k*, k*).  Both conditions are min_j max(dn_j, dn_in) (t_j - t*) > delta.
//     miss: t_out <= max(t_in, 0): every candidate has to pass the inside test of the entering plane e of t_in AND of
//           the leaving plane x of t_out, and each fails one of them by a margin if (t_in - t_out) min(|dn_e|, dn_x) >
//           2 delta (origin outside: candidates before the middle are outside e, the others outside x), or, from a
//           point beyond plane x (t_in <= 0: no entering plane is a candidate), if the nearest candidate is:
//           dn_x (t_min+ - t_out) > delta -- the margins only grow with t because |dn| >> the error per unit of t.
//   delta bounds everything float arithmetic can do to such a test: the tree evaluates fl(n_i.(fl(o + fl(d t_j)) -
//   off_i)) with t_j rounded; against nm_i + t_j dn_i (the float dot products taken as exact) that is off by at most
//   27 u (|o| + |off| + |d| t) |n|, u = 2^-24, including the 17 u by which the t_k used HERE may be off: they come from
//   v_rcp_f32 (1 ulp) instead of the division and carry the plane number in their low 3 bits (so that max / min return
//   the plane with the value).  delta = 64 u (...) in 1-norms.  Signs are exact: t_k > 0 iff nm_k and dn_k differ in
//   sign, in the reference's division and here alike; rays closer than 2^-16 |d| to parallel to a face, or with a
//   crossing closer than 1e-30 to the origin, are not decided here.
// Whatever is not decided by a margin -- near an edge, grazing, within delta of a face -- returns RL_PRISM_UNSURE and
// the caller evaluates the tree (wave-uniformly on the GPU: a few per cent of the rounds).  A decided answer IS the
// tree's answer, bit for bit: t is the reference's own division for plane k*.
// pr[0].w holds max_k |off_k|_1 and pr[2].w max(1, max_k |n_k|_1) (rl_scene.cpp).
enum { RL_PRISM_MISS = 0, RL_PRISM_HIT = 1, RL_PRISM_UNSURE = 2 };

RL_HD float rl_rcp_approx(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x); // 1 ulp
#elif defined(RL_TEST_RCP_NOISE)
    // host mirror under test: a reciprocal that is off by up to one ulp in either direction, like the hardware's may be
    const float r = 1.0f / x;
    const uint32_t h = (rl_f2u(x) * 2654435761u) >> 30;
    return h == 0 ? r : rl_u2f(rl_f2u(r) + (h == 1 ? 1u : h == 2 ? 0xffffffffu : 0u));
#else
    return 1.0f / x;
#endif
}

// PIPELINED (device): the next plane's records are requested behind this plane's arithmetic (eight more live registers: the
// plain launches, which have them; the open ones load plane by plane).
// PRELOADED (device, with PIPELINED): the caller requested the first plane's two records (pre_n = pr[0], pre_off = pr[1]) itself,
// ahead of fetching the ray -- the round then starts with one LDS round trip instead of two.
template <bool PIPELINED = true, bool PRELOADED = false>
RL_HD int rl_hex_prism_fast(const RlF4* pr, RlF3 o, RlF3 d, RlCand* out, RlF4 pre_n = RlF4(), RlF4 pre_off = RlF4()) {
    const float INF = __builtin_inff();
    float dn[8], ta[8];
    float t_in = -INF, t_out = INF;  // carry the plane number in their low 3 bits
    float min_dn = INF, min_ta = INF;
    uint32_t min_pos = 0xffffffffu;  // the smallest positive ta: positive floats order like their bits, negative ones are larger
#ifndef RL_W_P
#define RL_W_P 1 // rl_hex_prism_fast on the device: the next plane's records in flight behind this plane's arithmetic (A/B builds set 0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    float pair_in = 0.0f, pair_out = 0.0f, pair_dn = 0.0f, pair_ta = 0.0f; // the even plane of the current pair (below)
    RlF4 rec_n, rec_off;
    if (PIPELINED && RL_W_P) {
        if (PRELOADED) rec_n = pre_n, rec_off = pre_off;
        else rec_n = pr[0], rec_off = pr[1];
    }
#endif
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 0; k < 8; ++k) {
#if defined(__HIP_DEVICE_COMPILE__)
        // The next plane's two records are requested before this plane's arithmetic and waited for behind it: one plane
        // in flight, not sixteen loads at once (the round would spill) and not eight LDS round trips in a row either.
        RlF4 cur_n, cur_off;
        if (PIPELINED && RL_W_P) {
            cur_n = rec_n, cur_off = rec_off;
            if (k < 7) {
                rec_n = pr[2 * k + 2];
                rec_off = pr[2 * k + 3];
            }
        } else {
            cur_n = pr[2 * k], cur_off = pr[2 * k + 1];
        }
        // (one plane at a time: without this the scheduler issues the sixteen record loads first.  The records' unused fourth
        // components are operands of the empty statement so that the loads stay 16 bytes wide: a 12-byte LDS read is served in
        // eight groups of eight lanes, a 16-byte one in four of sixteen -- half the LDS time, MI355X_MICROARCH "LDS")
        if (PIPELINED) asm volatile("" : : "v"(cur_n.w), "v"(cur_off.w) : "memory"); // (PIPELINED: records in LDS, registers to spare)
        else asm volatile("" ::: "memory");
        const RlF3 n = rl_xyz(cur_n);
        const RlF3 lo = rl_sub(o, rl_xyz(cur_off));
#else
        const RlF3 n = rl_xyz(pr[2 * k]);
        const RlF3 lo = rl_sub(o, rl_xyz(pr[2 * k + 1]));
#endif
        const float dnk = rl_dot(n, d);   // exactly the reference's two dot products (geometry.rs:59-62)
        const float nm = rl_dot(n, lo);
        const float tk = rl_u2f((rl_f2u(-nm * rl_rcp_approx(dnk)) & 0xfffffff8u) | (uint32_t)k);
        dn[k] = dnk;
        ta[k] = tk;
        const bool entering = dnk < 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
        // The four running extrema two planes at a time, with v_max3 / v_min3 spelled out (round 6): tk is made by bit operations, so
        // the compiler cannot rule out a signalling NaN and put a canonicalising v_max_f32 x, x in front of every fmaxf / fminf of
        // it -- fourteen instructions per round.  The instructions themselves treat a (quiet) NaN as fmaxf / fminf do: the other operand.
        const float sel_in = entering ? tk : -INF, sel_out = entering ? INF : tk;
        if (k & 1) {
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(t_in) : "v"(pair_in), "v"(sel_in));
            asm("v_min3_f32 %0, %0, %1, %2" : "+v"(t_out) : "v"(pair_out), "v"(sel_out));
            asm("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(min_dn) : "v"(pair_dn), "v"(dnk));
            asm("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(min_ta) : "v"(pair_ta), "v"(tk));
        } else {
            pair_in = sel_in, pair_out = sel_out, pair_dn = dnk, pair_ta = tk;
        }
#else
        t_in = fmaxf(t_in, entering ? tk : -INF);
        t_out = fminf(t_out, entering ? INF : tk);
        min_dn = fminf(min_dn, fabsf(dnk));
        min_ta = fminf(min_ta, fabsf(tk));
#endif
        const uint32_t tb = rl_f2u(tk);
        min_pos = tb < min_pos ? tb : min_pos;
    }
    const float U64 = 3.814697265625e-06f; // 64 * 2^-24
    const float d1 = fabsf(d.x) + fabsf(d.y) + fabsf(d.z);
    const float scale = U64 * pr[2].w;
    const float s0 = fabsf(o.x) + fabsf(o.y) + fabsf(o.z) + pr[0].w;
    // not decided here: a ray nearly parallel to a face, a crossing at the origin (NaNs fail both compares)
    bool sure = (min_dn >= 1.52587890625e-05f * pr[2].w * d1) & (min_ta >= 1.0e-30f); // (& and | below: no branches, see rl_paraboloid_t)
    const uint32_t k_e = rl_f2u(t_in) & 7u, k_x = rl_f2u(t_out) & 7u;
#if defined(__HIP_DEVICE_COMPILE__)
    // The entry and the exit plane's records, all four requested at once (round 6): plane k* below is one of the two, so its normal,
    // its offset and its n.d are selected from these instead of loaded -- and waited for -- again once k* is known; 16-byte loads
    // (the fourth components are operands of the empty statement: a 12-byte LDS read takes twice the LDS time).
    const RlF4 rec_e = pr[2 * k_e], rec_x = pr[2 * k_x], off_e = pr[2 * k_e + 1], off_x = pr[2 * k_x + 1];
    if (PIPELINED) asm volatile("" : : "v"(rec_e.w), "v"(rec_x.w), "v"(off_e.w), "v"(off_x.w));
    const float dn_e = rl_dot(rl_xyz(rec_e), d), dn_x = rl_dot(rl_xyz(rec_x), d); // = dn[k_e], dn[k_x]
#else
    const float dn_e = rl_dot(rl_xyz(pr[2 * k_e]), d), dn_x = rl_dot(rl_xyz(pr[2 * k_x]), d); // = dn[k_e], dn[k_x]
#endif
    const bool from_outside = t_in > 0.0f;
    const float t0 = from_outside ? t_in : 0.0f;
    const bool reaches = t_out > t0;
    // hit: k* and the margins (a) and (b)
    const uint32_t k_star = from_outside ? k_e : k_x;
    const float t_star = from_outside ? t_in : t_out;
    const float dn_in = from_outside ? dn_e : -INF;
    float margin = INF;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < 8; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
        float larger; // fmaxf(dn[j], dn_in) without the canonicalising v_max_f32 x, x the compiler puts in front (above)
        asm("v_max_f32 %0, %1, %2" : "=v"(larger) : "v"(dn[j]), "v"(dn_in));
        const float a = larger * (ta[j] - t_star);
#else
        const float a = fmaxf(dn[j], dn_in) * (ta[j] - t_star);
#endif
        margin = fminf(margin, ta[j] == t_star ? INF : a);
    }
    const bool hit_sure = margin > scale * (s0 + d1 * t_star);
    // miss
    const float gap = t0 - t_out;
    const float t_first = (min_pos & 0x80000000u) ? INF : rl_u2f(min_pos);
#if defined(__HIP_DEVICE_COMPILE__)
    float least_dn, reach; // fminf(|dn_e|, dn_x), fmaxf(t0, |t_out|): spelled out for the same reason as the extrema above
    asm("v_min_f32 %0, |%1|, %2" : "=v"(least_dn) : "v"(dn_e), "v"(dn_x));
    asm("v_max_f32 %0, %1, |%2|" : "=v"(reach) : "v"(t0), "v"(t_out));
    const float miss_a = gap * least_dn, delta_a = 2.0f * scale * (s0 + d1 * reach);
#else
    const float miss_a = gap * fminf(fabsf(dn_e), dn_x), delta_a = 2.0f * scale * (s0 + d1 * fmaxf(t0, fabsf(t_out)));
#endif
    const float miss_b = dn_x * (t_first - t_out), delta_b = scale * (s0 + d1 * t_first);
    const bool miss_sure = from_outside ? miss_a > delta_a : (!(t_first < INF) | (miss_b > delta_b));
    sure = sure & (reaches ? hit_sure : miss_sure);
    // the reference's own t for plane k* (geometry.rs:62)
#if defined(__HIP_DEVICE_COMPILE__)
    const RlF3 n = rl_f3(from_outside ? rec_e.x : rec_x.x, from_outside ? rec_e.y : rec_x.y, from_outside ? rec_e.z : rec_x.z);
    const RlF3 lo = rl_sub(o, rl_f3(from_outside ? off_e.x : off_x.x, from_outside ? off_e.y : off_x.y, from_outside ? off_e.z : off_x.z));
    out->t = -rl_dot(n, lo) / (from_outside ? dn_e : dn_x); // (rl_dot(n, d) of the same operands: the same float)
#else
    const RlF3 n = rl_xyz(pr[2 * k_star]);
    const RlF3 lo = rl_sub(o, rl_xyz(pr[2 * k_star + 1]));
    out->t = -rl_dot(n, lo) / rl_dot(n, d);
#endif
    out->k = k_star;
    return !sure ? RL_PRISM_UNSURE : (reaches ? RL_PRISM_HIT : RL_PRISM_MISS);
}

// rl_hex_prism's result through the shortcut, the tree only where the shortcut does not decide.
RL_HD RlCand rl_hex_prism_decided(const RlF4* pr, RlF3 o, RlF3 dir) {
    RlCand c;
    const int status = rl_hex_prism_fast(pr, o, dir, &c);
    if (status == RL_PRISM_UNSURE) return rl_hex_prism(pr, o, dir);
    if (status == RL_PRISM_MISS) c.t = -1.0f;
    return c;
}

// Conservative cull (not in the reference) for a hexagonal prism or a sphere cluster: a valid hit
// lies inside the bounding sphere `b` = {centre, radius^2} (radius inflated by >= 5 %, see
// rl_scene.cpp).  Returns false only when the ray certainly misses the sphere in front of its origin.
RL_HD bool rl_bound_pass(RlF4 b, RlF3 o, RlF3 dir) {
    const float cox = b.x - o.x, coy = b.y - o.y, coz = b.z - o.z;
    const float dd = dir.x * cox + dir.y * coy + dir.z * coz;
    const float c = (cox * cox + coy * coy + coz * coz) - b.w;
    const float dlen2 = dir.x * dir.x + dir.y * dir.y + dir.z * dir.z; // glass leaves directions un-normalised
    // inside the sphere, or the line reaches it ahead of the origin (1e-3 relative slack on the discriminant)
    return !(c > 0.0f) || (dd > 0.0f && dd * dd >= c * dlen2 * 0.999f);
}

RL_HD bool rl_nearer(float t, uint32_t obj, const RlHit& best) {
    return t < best.t || (t == best.t && obj < best.obj);
}

RL_HD RlHit rl_scan(const RlSceneView& sv, RlF3 o, RlF3 dir) {
    RlHit best;
    best.t = 1.0e12f; // scene.rs:43
    best.obj = RL_HIT_NONE;
    best.sub = 0;

    // Spheres (geometry.rs:204-240): the direct list, then the clusters whose bound the ray reaches.
    auto sphere_test = [&](uint32_t pos) {
        const RlF4 s = sv.spheres[pos];
        const float cox = s.x - o.x, coy = s.y - o.y, coz = s.z - o.z;
        const float dd = dir.x * cox + dir.y * coy + dir.z * coz;
        const float c = (cox * cox + coy * coy + coz * coz) - s.w;
        const float q = dd * dd - c;
        if (q >= 0.0f && dd > 0.0f) {
            const float sq = sqrtf(q);
            const float t1 = dd - sq;
            const float t2 = dd + sq;
            const uint32_t obj = sv.sphere_obj[pos];
            if (t1 > 0.0f && t1 < t2 && rl_nearer(t1, obj, best)) {
                best.t = t1;
                best.obj = obj;
            }
        }
    };
    for (uint32_t i = 0; i < sv.n_direct; ++i) sphere_test(i);
    for (uint32_t k = 0; k < sv.n_clusters; ++k) {
        const uint32_t base = sv.cluster_base + (sv.cluster_k + 1u) * k;
        if (!rl_bound_pass(sv.spheres[base], o, dir)) continue;
        for (uint32_t j = 1; j <= sv.cluster_k; ++j) sphere_test(base + j);
    }

    // Paraboloids.
    for (uint32_t i = 0; i < sv.n_parabs; ++i) {
        const RlF4 r0 = sv.parabs[3 * i], r1 = sv.parabs[3 * i + 1], r2 = sv.parabs[3 * i + 2];
        const float t = rl_paraboloid_t(rl_xyz(r0), rl_xyz(r1), rl_xyz(r2), o, dir);
        const uint32_t obj = rl_f2u(r0.w);
        if (!(t < 0.0f) && rl_nearer(t, obj, best)) {
            best.t = t;
            best.obj = obj;
        }
    }
    // Planes and circles.
    for (uint32_t i = 0; i < sv.n_planes; ++i) {
        const RlF4 r0 = sv.planes[2 * i], r1 = sv.planes[2 * i + 1];
        float dn;
        const float t = rl_plane_t(rl_xyz(r0), rl_xyz(r1), o, dir, &dn);
        bool hit = t > 0.0f;
        if (hit && r0.w >= 0.0f) { // circle: geometry.rs:168-171
            const RlF3 dp = rl_sub(rl_add(o, rl_mul(dir, t)), rl_xyz(r1));
            hit = rl_dot(dp, dp) <= r0.w;
        }
        const uint32_t obj = rl_f2u(r1.w);
        if (hit && rl_nearer(t, obj, best)) {
            best.t = t;
            best.obj = obj;
        }
    }
    // Hexagonal prisms.
    for (uint32_t i = 0; i < sv.n_prisms; ++i) {
        const RlF4* pr = sv.prisms + RL_PRISM_STRIDE * i;
        if (!rl_bound_pass(pr[16], o, dir)) continue;
        const RlCand c = rl_hex_prism_decided(pr, o, dir);
        const uint32_t obj = rl_f2u(pr[1].w);
        if (c.t >= 0.0f && rl_nearer(c.t, obj, best)) {
            best.t = c.t;
            best.obj = obj;
            best.sub = c.k;
        }
    }
    return best;
}

// ---- hit completion (geometry.rs:73-86,110-121,166-184,242-259,343-357) -------------------------

struct RlIsect {
    RlF3 position, normal;
};

// Intersection.tangent (geometry.rs:250-251) is not built here: only the soap bubble reads it
// (material.rs:294), and rl_bounce derives it from the normal where it is needed.
// The sphere's and the paraboloid's normals both end in a normalisation (sqrt + three IEEE divisions);
// the un-normalised vector is selected per lane first so that a wave holding both kinds pays for one.
RL_HD RlIsect rl_finish_hit(const RlSceneView& sv, RlF3 o, RlF3 dir, const RlHit& hit, uint32_t surface_kind,
                            uint32_t group_index) {
    RlIsect is;
    is.position = rl_add(o, rl_mul(dir, hit.t));
#if defined(__HIP_DEVICE_COMPILE__)
    // The device's form: ONE record load whatever the surface is (`group_index` is a blob index here, RlSceneView::records; hit.sub
    // is zero unless the hit is a prism's), selects instead of a branch per kind -- as an if / else-if chain the compiler built a
    // tree of exec-mask branches with copies at every join, ~45 scalar instructions and a dozen moves per bounce, and a wave's
    // time goes into issuing instructions whatever their kind.  Same operations on the same values as below.
    {
        const RlF4 rec4 = sv.records[group_index + 2u * hit.sub];
        asm volatile("" : : "v"(rec4.w)); // (a 16-byte load: a 12-byte LDS read takes twice the LDS time)
        const RlF3 rec = rl_xyz(rec4);
        RlF3 curved = rl_sub(is.position, rec); // sphere: position - centre; paraboloid: local_pos
        if (surface_kind == RL_SURFACE_PARABOLOID) {
            const RlF4 rec_n = sv.records[group_index + 1u], rec_f = sv.records[group_index + 2u];
            asm volatile("" : : "v"(rec_n.w), "v"(rec_f.w)); // (16-byte loads, as above)
            const RlF3 normal = rl_xyz(rec_n);
            const RlF3 focal_point = rl_xyz(rec_f);
            const RlF3 plane_pr = rl_sub(curved, rl_mul(normal, rl_dot(curved, normal)));
            curved = rl_sub(focal_point, plane_pr);
        }
        // plane, circle: two-sided (the normal that faces the ray); prism: one-sided, as stored
        const bool flip = ((surface_kind == RL_SURFACE_PLANE) | (surface_kind == RL_SURFACE_CIRCLE)) & !(rl_dot(rec, dir) < 0.0f);
        const uint32_t sign = flip ? 0x80000000u : 0u;
        const RlF3 flat = rl_f3(rl_u2f(rl_f2u(rec.x) ^ sign), rl_u2f(rl_f2u(rec.y) ^ sign), rl_u2f(rl_f2u(rec.z) ^ sign));
        // (every lane normalises: the lanes of a flat surface a vector nobody reads)
        const RlF3 unit = rl_normalise(curved);
        const bool is_curved = (surface_kind == RL_SURFACE_SPHERE) | (surface_kind == RL_SURFACE_PARABOLOID);
        is.normal = rl_f3(is_curved ? unit.x : flat.x, is_curved ? unit.y : flat.y, is_curved ? unit.z : flat.z);
        return is;
    }
#endif
    is.normal = rl_f3(0.0f, 0.0f, 0.0f);
    RlF3 curved = rl_f3(0.0f, 0.0f, 0.0f);
    if (surface_kind == RL_SURFACE_SPHERE) {
        curved = rl_sub(is.position, rl_xyz(sv.spheres[group_index]));
    } else if (surface_kind == RL_SURFACE_PARABOLOID) {
        const RlF3 offset = rl_xyz(sv.parabs[3 * group_index]);
        const RlF3 normal = rl_xyz(sv.parabs[3 * group_index + 1]);
        const RlF3 focal_point = rl_xyz(sv.parabs[3 * group_index + 2]);
        const RlF3 local_pos = rl_sub(is.position, offset);
        const RlF3 plane_pr = rl_sub(local_pos, rl_mul(normal, rl_dot(local_pos, normal)));
        curved = rl_sub(focal_point, plane_pr);
    } else if (surface_kind == RL_SURFACE_HEX_PRISM) {
        is.normal = rl_xyz(sv.prisms[RL_PRISM_STRIDE * group_index + 2 * hit.sub]); // SpacePartitioning: one-sided
    } else { // plane, circle: two-sided
        const RlF3 n = rl_xyz(sv.planes[2 * group_index]);
        is.normal = (rl_dot(n, dir) < 0.0f) ? n : rl_neg(n);
    }
    if (surface_kind == RL_SURFACE_SPHERE || surface_kind == RL_SURFACE_PARABOLOID) is.normal = rl_normalise(curved);
    return is;
}

// ---- materials (material.rs) -------------------------------------------------------------------

// material.rs:61-74
RL_HD double rl_boltzmann(double wavelength, double temperature) {
    const double h = 6.62606957e-34, k = 1.3806488e-23, c = 299792458.0; // constants.rs:19-23
    const double f = c / (wavelength * 1.0e-9);
    return (2.0 * h * f * f * f) / (c * c * (rl_exp_d(h * f / (k * temperature)) - 1.0));
}
// material.rs:93-99
RL_HD float rl_black_body_normalisation(float kelvins, float intensity) {
    const double wien = 2.897772126e-3; // constants.rs:25
    return intensity / (float)rl_boltzmann((wien / (double)kelvins) * 1.0e9, (double)kelvins);
}
RL_HD float rl_clamp999(float x) { // material.rs:288-292
    if (x < -0.999f) return -0.999f;
    if (x > 0.999f) return 0.999f;
    return x;
}

// EmissiveMaterial::get_intensity scaled by the path's intensity (trace_unit.rs:99-101, material.rs:101-105)
// for a path that ended on emitter `obj`.
RL_HD float rl_emission(const RlSceneView& sv, float intensity, float wavelength, uint32_t obj) {
    const RlF4 ob = sv.objects[obj];
    return intensity * ((float)rl_boltzmann((double)wavelength, (double)ob.x) * ob.y);
}

enum { RL_PATH_CONTINUES = 0, RL_PATH_ENDED = 1, RL_PATH_ENDED_ON_EMITTER = 2 };

// Russian roulette, trace_unit.rs:122-125: `rand * 0.85 > continue_chance * (1 - exp(intensity * -20))`.
// Only the outcome of the comparison is used.  On the GPU the f64 exp (~50 f64 operations per bounce) is
// skipped when the hardware exp2 already decides it: e_fast is within 4e-6 of the exactly rounded
// exp(fl(intensity * -20)) for intensity in [0, 1] (1.8e-6 from the rounded exponent, 1.2e-6 from the
// product the exact form rounds differently, 1 ulp of v_exp_f32, the roundings of 1 - e and of the
// product with continue_chance <= 1), so outside a 2e-5 band both forms agree; inside it (or for any
// other lane of the wave, the branch is wave-uniform) the exact form decides.
RL_HD bool rl_roulette_ends(float unit, float continue_chance, float intensity) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float e_fast = __builtin_amdgcn_exp2f(intensity * -28.853901f); // -20 log2(e)
    const float gap = unit * 0.85f - continue_chance * (1.0f - e_fast);
    const bool undecided = !(fabsf(gap) >= 2.0e-5f) | !((intensity >= 0.0f) & (intensity <= 1.0f));
    if (RL_LIKELY(__builtin_amdgcn_ballot_w64(undecided) == 0)) return gap > 0.0f;
#endif
    return unit * 0.85f > continue_chance * (1.0f - rl_expf(intensity * -20.0f));
}

// One step of TraceUnit::render_ray's loop body after the scan (trace_unit.rs:92-126).
// RL_PATH_ENDED: *value is the path's contribution (0: The Void or roulette, trace_unit.rs:94,131).
// RL_PATH_ENDED_ON_EMITTER: the path hit emitter *emitter; its contribution is
// rl_emission(sv, p->intensity, p->wavelength, *emitter), left to the caller so that the kernel can
// evaluate the f64 Planck term for 64 ended paths at once instead of under divergence.
RL_HD int rl_bounce(const RlSceneView& sv, uint64_t seed, uint32_t stream, uint64_t path_index, RlPath* p,
                    const RlHit& hit, float* value, uint32_t* emitter) {
    *value = 0.0f;
    if (hit.obj == RL_HIT_NONE) return RL_PATH_ENDED; // The Void
    const RlF4 ob = sv.objects[hit.obj];
    const uint32_t kinds = rl_f2u(ob.w);
    const uint32_t surface_kind = rl_object_surface(kinds);
    const uint32_t material_kind = rl_object_material(kinds);
    if (material_kind == RL_MATERIAL_BLACK_BODY) {
        *emitter = hit.obj;
        return RL_PATH_ENDED_ON_EMITTER;
    }
    const RlIsect is = rl_finish_hit(sv, p->origin, p->direction, hit, surface_kind, rl_object_group(kinds));
    const RlRngBlock rb = rl_rng_block(seed, stream, path_index, 2u + p->bounce);
    const RlF3 in_dir = p->direction;
    // Mirror direction about the surface (vector3.rs:91-93): total internal reflection, the bubble's reflection
    // and the glossy blend all use it, and flipping the normal (glass leaving the medium) changes no bit of it
    // ((-a)(-b) == ab), so it is computed once for the wave instead of once per branch.
    const RlF3 mirrored = rl_reflect(in_dir, is.normal);
    RlF3 new_dir;
    float probability;

    if (material_kind == RL_MATERIAL_SF10_GLASS) { // material.rs:216-260
        float cos_i = -rl_dot(in_dir, is.normal);
        float ior = p->ior;
        RlF3 normal = is.normal;
        if (cos_i > 0.0f) {
            ior = rl_recipf(ior);
        } else {
            normal = rl_neg(normal);
            cos_i = -cos_i;
        }
        const float sin_t_sqr = ior * ior * (1.0f - cos_i * cos_i);
        if (sin_t_sqr > 1.0f) {
            new_dir = mirrored;
        } else {
            const float cos_t = rl_sqrtf(1.0f - sin_t_sqr);
            new_dir = rl_add(rl_mul(in_dir, ior), rl_mul(normal, ior * cos_i - cos_t));
        }
        probability = 1.0f;
    } else {
        // Soap bubbles and the diffuse family each need one f64 sin/cos evaluation: the cosine of the film's
        // phase (material.rs:293-294) and the hemisphere longitude (monte_carlo.rs:47-58).  A 64-wide wave
        // holds both kinds in nearly every iteration (profiles/: 99 %), so the argument is selected per lane
        // and rl_sincosf runs once for both; rl_cosf(x) is the cosine half of the same evaluation.
        // The same goes for a normalisation: the bubble's tangent normalise(cross((0, 1, 0), n)) -- spheres only,
        // every other surface leaves it the zero vector (geometry.rs:250-251), which normalise returns
        // unchanged -- and the first axis of rotate_towards, normalise(cross((0, 0, 1), facing)).
        const bool soap = material_kind == RL_MATERIAL_SOAP_BUBBLE;
        const RlF3 facing = (rl_dot(in_dir, is.normal) < 0.0f) ? is.normal : rl_neg(is.normal);
        RlF3 axis = rl_cross(rl_f3(0.0f, 0.0f, 1.0f), facing);
        // (rotate_towards returns before it looks at its first axis when |n.z| > 0.9999, vector3.rs:73-76: give those lanes -- every
        // hit on a horizontal plane, whose cross product is the zero vector -- a unit vector, so that the normalisation below
        // keeps its short form for the wave instead of falling back for a result nobody reads)
        if (fabsf(facing.z) > 0.9999f) axis = rl_f3(1.0f, 0.0f, 0.0f);
        if (soap) axis = surface_kind == RL_SURFACE_SPHERE ? rl_cross(rl_f3(0.0f, 1.0f, 0.0f), is.normal) : rl_f3(0.0f, 0.0f, 0.0f);
        const RlF3 unit_axis = rl_normalise(axis);
        float angle;
        float cos_phi = 0.0f, cos_theta = 0.0f;
        if (soap) { // material.rs:267-305
            const float cos_alpha = rl_dot(in_dir, is.normal);
            if (rl_get_unit(rb.w[0]) - 0.3f > fabsf(cos_alpha)) new_dir = mirrored;
            else new_dir = in_dir;
            cos_phi = rl_clamp999(rl_dot(new_dir, is.normal));
            cos_theta = rl_clamp999(rl_dot(new_dir, unit_axis)); // Intersection.tangent
        }
        // The film's two arc cosines (material.rs:293-294), under the film lanes' mask.  (Rounds 3-4 packed the 2n arguments of
        // the n film lanes into the first 2n lanes of the branch through LDS and evaluated them in one pass: better lane use,
        // but ~25 instructions and three LDS round trips more than the second evaluation costs -- and a wave's time goes into
        // issuing instructions, DESIGN.md 4.2: +0.3 % without it.)
        float acos_phi = 0.0f, acos_theta = 0.0f;
        if (soap) {
            acos_phi = rl_acosf(cos_phi);
            acos_theta = rl_acosf(cos_theta);
        }
        if (soap) {
            const float phase_shift = rl_div200f(p->wavelength - 380.0f) * RL_PI_F;
            angle = phase_shift - acos_phi * 3.0f - acos_theta * 2.0f + RL_PI_F * 0.5f;
        } else {
            angle = rl_get_longitude(rb.w[0]); // monte_carlo.rs:47-58
        }
        float sin_a, cos_a;
        rl_sincosf(angle, &sin_a, &cos_a);
        if (soap) {
            probability = cos_a * 0.1f + 0.9f;
        } else { // the diffuse family: material.rs:38-58 then :122-130 / :155-168 / :185-196
            const float rq = rl_get_unit(rb.w[1]);
            const float r = rl_sqrtf(rq);
            const RlF3 hemi = rl_f3(cos_a * r, sin_a * r, rl_sqrtf(1.0f - rq));
            new_dir = rl_rotate_towards(hemi, facing, unit_axis);
            probability = 1.0f;
            if (material_kind == RL_MATERIAL_DIFFUSE_GREY) {
                probability = ob.x;
            } else if (material_kind == RL_MATERIAL_DIFFUSE_COLOURED) {
                const float pw = (ob.y - p->wavelength) / ob.z;
                probability = ob.x * rl_expf(-0.5f * pw * pw);
            } else { // glossy mirror: blends with the mirror direction about the un-flipped normal
                new_dir = rl_normalise(rl_add(rl_mul(new_dir, ob.x), rl_mul(mirrored, 1.0f - ob.x)));
            }
        }
    }

    p->intensity = p->intensity * probability;                         // trace_unit.rs:106
    p->direction = new_dir;
    p->origin = rl_add(is.position, rl_mul(new_dir, 0.00001f));        // trace_unit.rs:114
    p->continue_chance = p->continue_chance * 0.96f;                   // trace_unit.rs:117
    p->bounce += 1;
    return rl_roulette_ends(rl_get_unit(rb.w[2]), p->continue_chance, p->intensity) ? RL_PATH_ENDED : RL_PATH_CONTINUES;
}

// ---- cie1931.rs:20-48 and plot_unit.rs:56-84 ----------------------------------------------------

RL_HD RlF3 rl_tristimulus(const RlF4* cie, float wavelength) {
    const float indexf = (wavelength - 380.0f) / 5.0f;
    const float fl = floorf(indexf);
    const int index = (int)fl;
    const float remainder = indexf - (float)index;
    if (index < -1 || index > 80) return rl_f3(0.0f, 0.0f, 0.0f);
    if (index == -1) {
        const RlF4 a = cie[0];
        return rl_f3(a.x * remainder, a.y * remainder, a.z * remainder);
    }
    if (index == 80) {
        const RlF4 a = cie[80];
        return rl_f3(a.x * (1.0f - remainder), a.y * (1.0f - remainder), a.z * (1.0f - remainder));
    }
    const RlF4 a = cie[index], b = cie[index + 1];
    return rl_f3(a.x * (1.0f - remainder) + b.x * remainder, a.y * (1.0f - remainder) + b.y * remainder,
                 a.z * (1.0f - remainder) + b.z * remainder);
}

struct RlSplat {
    uint32_t idx[4]; // pixel indices in the reference's order: (py1,px1) (py1,px2) (py2,px1) (py2,px2)
    float w[4];      // c11 c21 c12 c22
};

// wm1 = (float)width - 1.0f and hm1 = (float)height - 1.0f are passed in: plot_unit.rs:60-61 computes them per photon, the
// trace kernel gets them with its launch parameters (scalar registers) -- evaluated per call they were hoisted out of
// the persistent loop into two vector registers that then spilled.
RL_HD RlSplat rl_splat_weights(uint32_t width, uint32_t height, float wm1, float hm1, float aspect_ratio, float x, float y) {
    const int w = (int)width, h = (int)height;
    const float px = (x * 0.5f + 0.5f) * wm1;
    const float py = (y * aspect_ratio * 0.5f + 0.5f) * hm1;
    int px1 = (int)floorf(px), px2 = (int)ceilf(px), py1 = (int)floorf(py), py2 = (int)ceilf(py);
    px1 = px1 < 0 ? 0 : (px1 > w - 1 ? w - 1 : px1);
    px2 = px2 < 0 ? 0 : (px2 > w - 1 ? w - 1 : px2);
    py1 = py1 < 0 ? 0 : (py1 > h - 1 ? h - 1 : py1);
    py2 = py2 < 0 ? 0 : (py2 > h - 1 ? h - 1 : py2);
    const float cx = px - (float)px1;
    const float cy = py - (float)py1;
    RlSplat s;
    s.w[0] = (1.0f - cx) * (1.0f - cy);
    s.w[1] = cx * (1.0f - cy);
    s.w[2] = (1.0f - cx) * cy;
    s.w[3] = cx * cy;
    s.idx[0] = (uint32_t)(py1 * w + px1);
    s.idx[1] = (uint32_t)(py1 * w + px2);
    s.idx[2] = (uint32_t)(py2 * w + px1);
    s.idx[3] = (uint32_t)(py2 * w + px2);
    return s;
}
RL_HD RlSplat rl_splat_weights(uint32_t width, uint32_t height, float aspect_ratio, float x, float y) {
    return rl_splat_weights(width, height, (float)(int)width - 1.0f, (float)(int)height - 1.0f, aspect_ratio, x, y);
}

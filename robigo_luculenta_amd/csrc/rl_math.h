// rl_math.h -- scalar numerics shared by the gfx950 kernels (hipcc) and the host code (g++).
//
// Why this exists: a spectral path is a chain of branch decisions (discriminant < 0, t > 0,
// roulette compare, ...), so a one-ulp difference between a CPU and a GPU transcendental sends the
// path somewhere else entirely.  Every transcendental the reference's hot path uses
// (sin/cos: monte_carlo.rs:47-58, quaternion.rs:34-41, app.rs:327-357; exp: trace_unit.rs:122-123,
// material.rs:61-74,155-158; acos/cos: material.rs:293-294; ln/powf: tonemap_unit.rs:76-84,
// srgb.rs:20-26; tan: camera.rs:56) is therefore implemented here from IEEE-754 exactly-rounded
// primitives only (+ - * / sqrt floor, int<->float casts, bit casts), evaluated in f64 and rounded
// once to f32.  Built with -ffp-contract=off on both compilers, g++ and hipcc produce bit-identical
// results; the f64 evaluation keeps every function within 1 ulp (f32) of the exact value, which is
// as close to the reference's libm as the reference's own platforms are to each other.
//
// Nothing here is copied from a libm: reductions are Cody-Waite with a two-part constant,
// polynomials are plain Taylor series carried far enough for f64 (coefficients = 1/n!).
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RL_HD __host__ __device__ __forceinline__
#else
#define RL_HD inline
#endif

// f32 constants exactly as the reference's std::f32::consts::PI.
#define RL_PI_F 3.14159274101257324f

RL_HD uint64_t rl_bits_d(double x) { return __builtin_bit_cast(uint64_t, x); }
RL_HD double rl_from_bits_d(uint64_t b) { return __builtin_bit_cast(double, b); }
RL_HD uint32_t rl_bits_f(float x) { return __builtin_bit_cast(uint32_t, x); }
RL_HD float rl_from_bits_f(uint32_t b) { return __builtin_bit_cast(float, b); }

// ---------------------------------------------------------------------------------------------
// sin / cos, f64 core.  Valid for |x| < ~1e5 (the hot path stays below 25).
// ---------------------------------------------------------------------------------------------
RL_HD void rl_sincos_d(double x, double* s_out, double* c_out) {
    const double TWO_OVER_PI = 0.6366197723675814;
    const double PIO2_1 = 1.5707963267341256;     // 0x3ff921fb54400000: pi/2 with 20 trailing zero bits
    const double PIO2_1T = 6.077100506506192e-11; // pi/2 - PIO2_1
    const double kd = floor(x * TWO_OVER_PI + 0.5);
    const double r = (x - kd * PIO2_1) - kd * PIO2_1T;
    const int k = (int)kd;
    const double z = r * r;
    // sin(r) = r + r z (S1 + z (S2 + ...)), |r| <= pi/4
    double ps = 2.8114572543455206e-15;
    ps = ps * z + -7.647163731819816e-13;
    ps = ps * z + 1.6059043836821613e-10;
    ps = ps * z + -2.505210838544172e-08;
    ps = ps * z + 2.7557319223985893e-06;
    ps = ps * z + -0.0001984126984126984;
    ps = ps * z + 0.008333333333333333;
    ps = ps * z + -0.16666666666666666;
    const double sr = r + r * (z * ps);
    // cos(r) = 1 + z (C1 + z (C2 + ...))
    double pc = -1.5619206968586225e-16;
    pc = pc * z + 4.779477332387385e-14;
    pc = pc * z + -1.1470745597729725e-11;
    pc = pc * z + 2.08767569878681e-09;
    pc = pc * z + -2.755731922398589e-07;
    pc = pc * z + 2.48015873015873e-05;
    pc = pc * z + -0.001388888888888889;
    pc = pc * z + 0.041666666666666664;
    pc = pc * z + -0.5;
    const double cr = 1.0 + z * pc;
    const int q = k & 3;
    const double s = (q & 1) ? cr : sr;
    const double c = (q & 1) ? sr : cr;
    *s_out = (q & 2) ? -s : s;
    *c_out = ((q + 1) & 2) ? -c : c;
}

RL_HD float rl_sinf(float x) {
    double s, c;
    rl_sincos_d((double)x, &s, &c);
    return (float)s;
}
RL_HD float rl_cosf(float x) {
    double s, c;
    rl_sincos_d((double)x, &s, &c);
    return (float)c;
}
RL_HD void rl_sincosf(float x, float* s_out, float* c_out) {
    double s, c;
    rl_sincos_d((double)x, &s, &c);
    *s_out = (float)s;
    *c_out = (float)c;
}
RL_HD float rl_tanf(float x) {
    double s, c;
    rl_sincos_d((double)x, &s, &c);
    return (float)(s / c);
}

// ---------------------------------------------------------------------------------------------
// exp, f64.
// ---------------------------------------------------------------------------------------------
RL_HD double rl_exp_d(double x) {
    if (x < -745.0) return 0.0;
    if (x > 709.0) return rl_from_bits_d(0x7ff0000000000000ull);
    const double LOG2E = 1.4426950408889634;
    const double LN2_HI = 0.6931471803691238;     // 0x3fe62e42fee00000
    const double LN2_LO = 1.9082149292705877e-10; // ln2 - LN2_HI
    const double kd = floor(x * LOG2E + 0.5);
    const double r = (x - kd * LN2_HI) - kd * LN2_LO; // |r| <= ln2/2
    double p = 1.1470745597729725e-11; // 1/14!
    p = p * r + 1.6059043836821613e-10;
    p = p * r + 2.08767569878681e-09;
    p = p * r + 2.505210838544172e-08;
    p = p * r + 2.755731922398589e-07;
    p = p * r + 2.7557319223985893e-06;
    p = p * r + 2.48015873015873e-05;
    p = p * r + 0.0001984126984126984;
    p = p * r + 0.001388888888888889;
    p = p * r + 0.008333333333333333;
    p = p * r + 0.041666666666666664;
    p = p * r + 0.16666666666666666;
    p = p * r + 0.5;
    const double e = 1.0 + (r + (r * r) * p);
    const int k = (int)kd;
    const int k1 = k / 2;
    const int k2 = k - k1;
    const double s1 = rl_from_bits_d((uint64_t)(k1 + 1023) << 52);
    const double s2 = rl_from_bits_d((uint64_t)(k2 + 1023) << 52);
    return (e * s1) * s2;
}
RL_HD float rl_expf(float x) { return (float)rl_exp_d((double)x); }

// ---------------------------------------------------------------------------------------------
// natural log, f64; x must be positive, finite and normal (all call sites cast from positive f32).
// ---------------------------------------------------------------------------------------------
RL_HD double rl_log_d(double x) {
    const double LN2_HI = 0.6931471803691238;
    const double LN2_LO = 1.9082149292705877e-10;
    const uint64_t b = rl_bits_d(x);
    int e = (int)(b >> 52) - 1023;
    double m = rl_from_bits_d((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e = e + 1;
    }
    const double s = (m - 1.0) / (m + 1.0); // |s| <= 0.1716
    const double z = s * s;
    double p = 0.047619047619047616; // 1/21
    p = p * z + 0.05263157894736842;
    p = p * z + 0.058823529411764705;
    p = p * z + 0.06666666666666667;
    p = p * z + 0.07692307692307693;
    p = p * z + 0.09090909090909091;
    p = p * z + 0.1111111111111111;
    p = p * z + 0.14285714285714285;
    p = p * z + 0.2;
    p = p * z + 0.3333333333333333;
    const double lm = 2.0 * s + 2.0 * s * (z * p);
    const double ed = (double)e;
    return ed * LN2_HI + (lm + ed * LN2_LO);
}
RL_HD float rl_logf(float x) { return (float)rl_log_d((double)x); }

// x^y for x > 0 (srgb.rs:24 gamma).
RL_HD float rl_powf(float x, float y) { return (float)rl_exp_d((double)y * rl_log_d((double)x)); }

// ---------------------------------------------------------------------------------------------
// acos, f64 core, |x| <= 1 (the only call sites clamp to +-0.999, material.rs:288-294).
// ---------------------------------------------------------------------------------------------
RL_HD double rl_asin_core_d(double y) { // |y| <= 0.5 (+ a hair), Taylor: sum c_n y^(2n+1)
    const double z = y * y;
    double p = 0.003297059503473485;
    p = p * z + 0.0035692053938259347;
    p = p * z + 0.003880964558837669;
    p = p * z + 0.004240907093679363;
    p = p * z + 0.004660143486915096;
    p = p * z + 0.005153309682319905;
    p = p * z + 0.005740037670841924;
    p = p * z + 0.006447210311889649;
    p = p * z + 0.0073125258735988454;
    p = p * z + 0.008390335809616815;
    p = p * z + 0.009761609529194078;
    p = p * z + 0.011551800896139705;
    p = p * z + 0.01396484375;
    p = p * z + 0.017352764423076924;
    p = p * z + 0.022372159090909092;
    p = p * z + 0.030381944444444444;
    p = p * z + 0.044642857142857144;
    p = p * z + 0.075;
    p = p * z + 0.16666666666666666;
    return y + y * (z * p);
}
RL_HD double rl_acos_d(double x) {
    const double PI = 3.141592653589793;
    const double PIO2 = 1.5707963267948966;
    const double ax = x < 0.0 ? -x : x;
    if (ax <= 0.5) return PIO2 - rl_asin_core_d(x);
    const double a = 2.0 * rl_asin_core_d(sqrt((1.0 - ax) * 0.5));
    return x < 0.0 ? PI - a : a;
}
RL_HD float rl_acosf(float x) { return (float)rl_acos_d((double)x); }

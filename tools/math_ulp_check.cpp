// math_ulp_check.cpp -- TEST TOOL: the f32 transcendentals of csrc/rl_math.h (rl_sincosf, rl_expf, rl_acosf) against the
// platform's f64 libm over EVERY f32 argument of their domain (step 1) or every step-th one.
//   g++ -O2 -std=c++17 -ffp-contract=off -mfma -pthread -I robigo_luculenta_amd/csrc -o /tmp/math_ulp_check tools/math_ulp_check.cpp
//   /tmp/math_ulp_check [step] > profiles/r04_math_ulp.txt
// Per function: the largest error in ulps of the correctly rounded result (measured against the f64 value, itself within
// 1e-9 f32 ulps of the truth), how many results are not the correctly rounded float, and how many are farther than its
// neighbour (must be 0: "at most 1 ulp from the correctly rounded value"); the same for the f64-evaluated forms
// (rl_*_d) the build used before round 4, and how many results differ between the two families.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <thread>
#include <vector>

#include "rl_math.h"

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
struct Acc {
    double max_err = 0;
    uint32_t at = 0;
    uint64_t n = 0, not_cr = 0, far = 0, differ = 0;
};
static inline void rec(Acc& a, uint32_t bits, float y, double t, float other) {
    const float cr = (float)t;
    int e;
    frexpf(cr == 0.0f ? 1.0e-30f : fabsf(cr), &e);
    const double err = fabs((double)y - t) / ldexp(1.0, e - 24);
    a.n++;
    if (err > a.max_err) a.max_err = err, a.at = bits;
    if (y != cr) {
        a.not_cr++;
        const int32_t d = (int32_t)(f2u(y) - f2u(cr));
        if (d < -1 || d > 1) a.far++;
    }
    if (f2u(y) != f2u(other)) a.differ++;
}
template <class F> static void sweep(const char* name, float lo, float hi, uint32_t step, F f) {
    const int nt = (int)std::thread::hardware_concurrency() > 0 ? (int)std::thread::hardware_concurrency() : 4;
    std::vector<Acc> acc(nt);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            Acc a;
            for (uint64_t b = (uint64_t)f2u(lo) + (uint64_t)t * step; b <= f2u(hi); b += (uint64_t)nt * step)
                for (int sign = 0; sign < 2; ++sign) f((uint32_t)b | (sign ? 0x80000000u : 0u), a);
            acc[t] = a;
        });
    for (auto& x : th) x.join();
    Acc tot;
    for (auto& a : acc) {
        if (a.max_err > tot.max_err) tot.max_err = a.max_err, tot.at = a.at;
        tot.n += a.n, tot.not_cr += a.not_cr, tot.far += a.far, tot.differ += a.differ;
    }
    printf("%-34s arguments %11llu  max error %.4f ulp at %-14.9g not correctly rounded %9llu (%.4f %%)  farther than the neighbour %llu  differ from the other family %llu\n",
           name, (unsigned long long)tot.n, tot.max_err, u2f(tot.at), (unsigned long long)tot.not_cr, 100.0 * tot.not_cr / tot.n,
           (unsigned long long)tot.far, (unsigned long long)tot.differ);
    fflush(stdout);
}
int main(int argc, char** argv) {
    const uint32_t step = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u;
    printf("rl_math.h against the platform's f64 libm, every %u-th f32 argument of each domain, both signs\n", step);
    sweep("rl_sincosf sin  |x| <= 32", 0.0f, 32.0f, step, [](uint32_t b, Acc& a) { float x = u2f(b), s, c, s2, c2; rl_sincosf(x, &s, &c); rl_sincosf_d(x, &s2, &c2); rec(a, b, s, sin((double)x), s2); });
    sweep("rl_sincosf cos  |x| <= 32", 0.0f, 32.0f, step, [](uint32_t b, Acc& a) { float x = u2f(b), s, c, s2, c2; rl_sincosf(x, &s, &c); rl_sincosf_d(x, &s2, &c2); rec(a, b, c, cos((double)x), c2); });
    sweep("rl_sincosf sin  32 < |x| <= 1e5", 32.0f, 1.0e5f, step, [](uint32_t b, Acc& a) { float x = u2f(b), s, c, s2, c2; if (fabsf(x) <= 32.0f) return; rl_sincosf(x, &s, &c); rl_sincosf_d(x, &s2, &c2); rec(a, b, s, sin((double)x), s2); });
    sweep("rl_expf  -86 <= x <= 88", 0.0f, 88.0f, step, [](uint32_t b, Acc& a) { float x = u2f(b); if (x < -86.0f) return; rec(a, b, rl_expf(x), exp((double)x), rl_expf_d(x)); });
    sweep("rl_acosf |x| <= 1", 0.0f, 1.0f, step, [](uint32_t b, Acc& a) { float x = u2f(b); rec(a, b, rl_acosf(x), acos((double)x), rl_acosf_d(x)); });
    sweep("rl_sincosf_d sin  |x| <= 32", 0.0f, 32.0f, step, [](uint32_t b, Acc& a) { float x = u2f(b), s, c; rl_sincosf_d(x, &s, &c); rec(a, b, s, sin((double)x), s); });
    sweep("rl_sincosf_d cos  |x| <= 32", 0.0f, 32.0f, step, [](uint32_t b, Acc& a) { float x = u2f(b), s, c; rl_sincosf_d(x, &s, &c); rec(a, b, c, cos((double)x), c); });
    sweep("rl_expf_d  -86 <= x <= 88", 0.0f, 88.0f, step, [](uint32_t b, Acc& a) { float x = u2f(b); if (x < -86.0f) return; float y = rl_expf_d(x); rec(a, b, y, exp((double)x), y); });
    sweep("rl_acosf_d |x| <= 1", 0.0f, 1.0f, step, [](uint32_t b, Acc& a) { float x = u2f(b); float y = rl_acosf_d(x); rec(a, b, y, acos((double)x), y); });
    return 0;
}

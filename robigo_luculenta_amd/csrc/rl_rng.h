// rl_rng.h -- counter-based random numbers shared by the gfx950 kernels and the host code.
//
// The reference draws from rand 0.3.11's OS-seeded thread-local generator (monte_carlo.rs:22-38),
// which cannot be seeded, so "matched seed" has to be defined by the build: every draw is a pure
// function of (seed, stream, path_index, block, slot).  Philox4x32 (Salmon et al., SC'11) is the
// generator; one call yields the four 32-bit slots of one block.  Rounds: RL_PHILOX_ROUNDS = 7, the fewest for which the
// authors report the generator Crush-resistant (their Table 2; 10 is their default with a safety margin, and what rounds
// 1-3 of this build used: the three rounds less are 1.2 % of the trace kernel's time).  The round function is pinned by
// Random123's published 10-round vectors (rl_philox4x32_10, tests/test_oracle_kat.py), the 7-round words by an independent
// numpy implementation over a million tuples (tests/test_independent.py).
//
//   block 0        : slot 0 wavelength, slot 1 screen x, slot 2 screen y, slot 3 camera time
//                    (trace_unit.rs:154-158,138)
//   block 1        : slot 0 depth-of-field angle (half-open), slot 1 depth-of-field radius
//                    (camera.rs:96-97)
//   block 2 + b    : bounce b: slot 0 hemisphere longitude (half-open) or the soap-bubble
//                    reflect/pass draw, slot 1 hemisphere radius^2, slot 2 Russian roulette
//                    (monte_carlo.rs:48-49, material.rs:273, trace_unit.rs:122)
//
// Every lane of a wave therefore makes exactly one Philox call per bounce regardless of which
// material it hit -- no divergence and no cached words live across the intersection scan.
//
// u32 -> f32 follows rand 0.3.11's conversions: [0,1) keeps the top 24 bits and scales by 2^-24;
// Closed01 rescales that by 2^24/(2^24-1) so that 1.0 is reachable (monte_carlo.rs:25-28).
#pragma once
#include "rl_math.h"

struct RlRngBlock {
    uint32_t w[4];
};

RL_HD uint32_t rl_mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

#define RL_PHILOX_ROUNDS 7
// UNIFORM_KEY (device): the caller guarantees that the key is the same in every lane of the wave (rl_rng_block: the launch's
// seed).  Only then may the key be pinned to scalar registers -- a per-lane key forced through an "s" constraint would silently
// become lane 0's -- so the public rl_philox4x32_10 and every other caller leave it false.
template <int ROUNDS, bool UNIFORM_KEY = false>
RL_HD RlRngBlock rl_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if defined(__HIP_DEVICE_COMPILE__)
    // The key (the launch's seed) is wave-uniform and loop-invariant over the trace kernel's persistent loop: left alone, the
    // optimiser hoists the whole key schedule -- 18 sums key + round * W -- out of that loop into scalar registers that
    // then do not fit (spilled to lanes of a vector register and read back with v_readlane).  Opaque here, the schedule
    // is 18 s_add_i32 per block on the scalar unit.
    if (UNIFORM_KEY) asm volatile("" : "+s"(k0), "+s"(k1));
#endif
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int round = 0; round < ROUNDS; ++round) {
#if defined(__HIP_DEVICE_COMPILE__)
        // One v_mad_u64_u32 (measured 5.0 cycles per wave) yields both halves of the 32x32 product; the
        // compiler otherwise emits v_mul_hi_u32 + v_mul_lo_u32 (4.4 + 4.7 cycles) for the constant multiplier.
        unsigned long long p0, p1, carry;
        asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p0), "=s"(carry) : "v"(c0), "s"(M0));
        asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p1), "=s"(carry) : "v"(c2), "s"(M1));
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#else
        const uint32_t hi0 = rl_mulhi32(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = rl_mulhi32(M1, c2), lo1 = M1 * c2;
#endif
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += W0;
        k1 += W1;
    }
    RlRngBlock r;
    r.w[0] = c0;
    r.w[1] = c1;
    r.w[2] = c2;
    r.w[3] = c3;
    return r;
}

RL_HD RlRngBlock rl_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    return rl_philox4x32<10>(c0, c1, c2, c3, k0, k1); // the published variant: known-answer tests
}

// One block of draws for (seed, stream, path, block).  `seed` is the launch's seed: the same in every lane (UNIFORM_KEY).
RL_HD RlRngBlock rl_rng_block(uint64_t seed, uint32_t stream, uint64_t path, uint32_t block) {
    return rl_philox4x32<RL_PHILOX_ROUNDS, true>((uint32_t)path, (uint32_t)(path >> 32), block, stream, (uint32_t)seed,
                                           (uint32_t)(seed >> 32));
}

// rand 0.3.11 `random::<f32>()`: 24 random bits in [0, 1).
RL_HD float rl_halfopen01(uint32_t u) { return (float)(u >> 8) * 5.9604644775390625e-8f; }
// rand 0.3.11 `random::<Closed01<f32>>()`: [0, 1].
RL_HD float rl_closed01(uint32_t u) { return rl_halfopen01(u) * (16777216.0f / 16777215.0f); }

// monte_carlo.rs:25-43 on top of the draws.
RL_HD float rl_get_unit(uint32_t u) { return rl_closed01(u); }
RL_HD float rl_get_bi_unit(uint32_t u) { return rl_closed01(u) * 2.0f - 1.0f; }
RL_HD float rl_get_longitude(uint32_t u) { return rl_halfopen01(u) * RL_PI_F * 2.0f; }
RL_HD float rl_get_wavelength(uint32_t u) { return rl_closed01(u) * 400.0f + 380.0f; }

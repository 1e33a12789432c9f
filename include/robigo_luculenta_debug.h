/* robigo_luculenta_debug.h -- diagnostics of librobigo_luculenta.so.  NOT part of the drop-in boundary
 * (include/robigo_luculenta.h): nothing here replaces a reference interface, a host never needs it, and the
 * Rust binding (bindings/rust/ffi.rs) does not declare it.  Used by tests/, tools/ and the profiling scripts. */
#ifndef ROBIGO_LUCULENTA_DEBUG_H
#define ROBIGO_LUCULENTA_DEBUG_H
#include "robigo_luculenta.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Evaluates the shared numerics header (csrc/rl_math.h) on the GPU so a test can check that the hipcc and
 * g++ builds agree bit-for-bit.  fn: 0 sin, 1 cos, 2 tan, 3 exp, 4 ln, 5 acos, 6 SF10 index of refraction
 * (material.rs:203-213), 7 sqrt, 8 x[i] / x[i+1 mod n], 9 x^(1/2.4) (srgb.rs:24), 10 the Russian-roulette
 * decision (trace_unit.rs:122-125) for the triples (x[i], x[m+i], x[2m+i]) = (rand, continue_chance,
 * intensity), i < m = n / 3, result 1 or 0 in y[i], 11 vector3.rs:56-67 normalise of the triples (x[i], x[m+i], x[2m+i])
 * into (y[i], y[m+i], y[2m+i]). */
int rl_debug_math_probe(int device, int fn, const float* x, float* y, uint32_t n);
/* The short square root / reciprocal / division by 200 of csrc/rl_math.h and rl_core.h (fn 16, 17, 18 of rl_debug_math_probe)
 * against the compiler's correctly rounded expansions, ON THE DEVICE, for every float whose bits lie in [lo_bits, hi_bits) -- and
 * its negative when both_signs != 0.  counts[0] = arguments whose results differ, counts[1] = arguments compared, *example = the
 * bits of one argument that differs.  Seconds for the whole normal range: the exhaustive checks of the -m gpu suite. */
int rl_debug_math_sweep(int device, int fn, uint32_t lo_bits, uint32_t hi_bits, int both_signs, uint64_t* counts, uint32_t* example);
/* The rank bookkeeping of rl_app_run for RlAppConfig{device, devices, n_devices} -- which device each rank renders on (its RNG
 * stream is config.stream + rank), the first rank on the same device (ranks sharing a GPU are added onto it), the rank's place in
 * the RCCL communicator (one per DISTINCT device in order of first appearance, so rank 0 is the root; -1: none) and the
 * communicator's size (0: a run on one device needs none).  Host arithmetic only: callable without a GPU.  Arrays of
 * max(n_devices, 1) entries. */
int rl_debug_app_rank_plan(int device, const int* devices, uint32_t n_devices, int* rank_device, int* leader, int* comm_rank,
                           uint32_t* n_communicator_ranks);
/* Blocking render calls share launches (see rl_trace_unit_render).  out[k], k = 1..256: launches on `device` that
 * carried k calls since the library was loaded (257 counters, out[0] unused).  Waits for running ones to end. */
int rl_debug_batch_histogram(int device, uint64_t* out);
/* out[v], v = 0..23: launches of instantiation v of the trace kernel since the library was loaded.  v < 16: v = 8 * (the whole
 * scene staged in LDS) + 4 * (trace fused with the splat) + 2 * (open launch: a blocking render call) + 1 * (prisms carry a
 * second bound); v = 16 + the low three bits: the variants that stage the scene's tables only (a scene too large for LDS).
 * The parity matrix (tests/test_gpu_parity.py) asserts with it that the variant it means to test is the one that ran. */
int rl_debug_variant_launches(uint64_t* out);
/* The prism shortcut (csrc/rl_core.h: rl_hex_prism_fast) against the Compound tree it stands in for
 * (geometry.rs:380-407), both evaluated ON THE GPU -- with the hardware's v_rcp_f32 -- for n rays against prism
 * `prism` (0-based, in the scene's flattened order) of `scene`.  rays: n x {origin.xyz, direction.xyz}.
 * out: n x {status (0 miss, 1 hit, 2 undecided), t bits and half-space of the shortcut, t bits (or 0xffffffff for
 * "no hit") and half-space of the tree} as 5 uint32 each. */
int rl_debug_prism_probe(const RlScene* scene, uint32_t prism, const float* rays, uint32_t n, uint32_t* out);
/* Number of prism records of the scene's flattened form (padding prisms of the cull groups included). */
int rl_debug_prism_count(const RlScene* scene, uint32_t* n_prisms);

#ifdef __cplusplus
}
#endif
#endif /* ROBIGO_LUCULENTA_DEBUG_H */

// rl_kernels.hip.h -- the gfx950 kernels: persistent-wavefront trace (+ fused CIE splat), plot,
// Kahan gather, exposure, tonemap.  Included once by rl_api.hip.
//
// Trace kernel design (wave64, CDNA4):
//   * one workgroup of 16 waves per CU stages ONE copy of the scene (16-byte records, rl_scene.h) in
//     LDS (RL_FETCH_LDS; wave-uniform ds_read_b128 broadcasts) or reads it from global memory with
//     wave-uniform addresses (RL_FETCH_GLOBAL; automatic when the scene does not fit beside the
//     per-wave scratch in 160 KB);
//   * persistent waves, one live path per lane.  Camera rays are generated 64 at a time with a full
//     exec mask into a per-wave LDS stash; a lane whose path ended pops the next ray from the stash,
//     so the scan always runs with all lanes busy until the global work queue (one atomic per 256
//     paths per wave) drains;
//   * the scan (rl_scan_wave) runs only cheap reject / cull tests lane-per-ray; every expensive tail
//     (sphere roots, cluster members, prisms) is compacted with ballot/mbcnt into LDS rings and
//     evaluated 64 (item, ray) pairs at a time -- a pair's lane gathers its ray from the owner lane's slot of the
//     wave's LDS scratch -- results min-merged per ray with ds_min_u64;
//   * results leave either as MappedPhoton records (un-fused, bit-comparable with the CPU) or as
//     12 hardware f32 atomics per contributing path into the XYZ buffer (fused TraceUnit+PlotUnit);
//     in fused mode the paths that ended on a light wait in a per-wave LDS queue until 64 of them can
//     be evaluated (f64 Planck term) and splatted with a full exec mask;
//   * no MFMA: there is no dense contraction in this workload.  What bounds it (round 5; probes in this file under
//     RL_EXP_EXTRA): not the vector ALU's lanes but what ONE WAVE can issue -- at four waves per SIMD a wave spends half of its
//     time issuing instructions, vector, scalar and LDS alike (a scalar instruction costs what a vector one does, a packed FMA
//     two, a taken branch two), a third waiting for LDS round trips (an exposed one costs ten instructions, a ds_bpermute_b32
//     five) and a fifth issue-stalled.  Hence: record loads issued ahead of the data they are used with, the rays' cull terms
//     gathered from LDS slots instead of permuted across lanes, conditions combined without branches, ring pushes in four
//     vector instructions -- and no attention to lane occupancy for its own sake.  The second half of round 5 read the compiled
//     kernel region by region (-DRL_MARK, tools/region_census.py) for what the source does not say: the pushes' exec dance (now
//     eight instructions of inline assembly, RL_RING_PUSH / RL_LE_PUSH), an exec-mask tree for an if / else-if chain
//     (rl_finish_hit: one load and selects), sixteen copies at a loop's back edge (the stash hand-out: straight-line now);
//   * OPEN variant: the kernel stays resident and takes the paths of blocking render calls from a job table the host
//     appends to while it runs (RlOpenDev / RlOpenCtl below): rl_trace_kernel_open, at most 120 VGPRs so that the small
//     kernels of the other units run beside it (the plain launches, rl_trace_kernel, may use all 128).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <type_traits>

#include "rl_cie1931.h"
#include "rl_core.h"

#define RL_BLOCK 256        // streaming kernels (plot, gather, tonemap)
#ifndef RL_TRACE_BLOCK
#define RL_TRACE_BLOCK 1024 // trace kernel: one workgroup of 16 waves per CU shares one LDS copy of the scene
#endif
#ifndef RL_TRACE_WPS
#define RL_TRACE_WPS 4      // waves per SIMD the trace kernel is compiled for (launch bounds; the register caps below follow from it)
#endif
#ifndef RL_TRACE_VGPRS
#define RL_TRACE_VGPRS (RL_TRACE_WPS == 4 ? 128 : 96)      // plain launches: all a wave can have at that occupancy (512 / waves, in eights)
#endif
#ifndef RL_TRACE_VGPRS_OPEN
#define RL_TRACE_VGPRS_OPEN (RL_TRACE_WPS == 4 ? 120 : 96) // open launches leave the small kernels of the other units their registers
#endif
#ifndef RL_W_S
#define RL_W_S 1 // ring-S rounds of the plain launches: the children's bounds requested ahead of the cross-lane fetch (A/B builds set 0)
#endif
#ifndef RL_W_B
#define RL_W_B 1 // ring-B rounds: the sphere record, its radius^2 and its object likewise
#endif
#ifndef RL_W_M
#define RL_W_M 1 // cluster-member rounds: the first member likewise
#endif
#ifndef RL_W_PR
#define RL_W_PR 1 // prism rounds: the first plane's two records likewise
#endif
#ifndef RL_EMIT_MAX_AGE
#define RL_EMIT_MAX_AGE 8u
#endif
#ifndef RL_OPEN_FULL
#define RL_OPEN_FULL 1
#endif
#ifndef RL_SMALL_UNROLL
#define RL_SMALL_UNROLL 1
#endif
#ifndef RL_DIRECT_ONE
#define RL_DIRECT_ONE 1
#endif
#ifndef RL_S_AHEAD2
#define RL_S_AHEAD2 1 // ring-S rounds of a scene whose cull table is in global memory: two children's bounds requested ahead
#endif
#ifndef RL_MEMBER_AHEAD2
#define RL_MEMBER_AHEAD2 0 // cluster-member rounds of a scene whose spheres are in global memory: two records requested ahead instead of one (measured: 4,539 objects -4.5 %, 5,019 random spheres +-0 -- the gathers are bound by the lines the L1 serves per cycle, not by latency)
#endif
#ifndef RL_MEMBER_FENCE
#define RL_MEMBER_FENCE (RL_TRACE_WPS > 4)
#endif
// Five waves per SIMD leave a wave 96 registers: the options that buy latency with registers are off there unless a build asks
#ifndef RL_LEAN_SPLIT
#define RL_LEAN_SPLIT (RL_TRACE_WPS <= 4)
#endif
#ifndef RL_LEAN_HOIST
#define RL_LEAN_HOIST (RL_TRACE_WPS <= 4)
#endif
#ifndef RL_PROGRESSIVE_FAR
#define RL_PROGRESSIVE_FAR 1 // in the variants for scenes that are not staged whole (rl_scan_wave: SUPER): the rounds' far bound follows the merge keys -- the nearest hit found SO FAR in this scan -- not just the small primitives' hit
#endif
#define RL_CHUNK 256ull     // paths a wave takes from the global queue at a time (4 stash refills) in large launches

struct RlSceneLayout {
    // Offsets into the scene blob, in RlF4 units.  Spheres start at 0.
    uint32_t off_planes, off_parabs, off_prisms, off_objects, off_cull, off_camera, off_cie, off_sphere_obj, total_f4;
    float cull_cmax2; // RlFlatScene::cull_cmax2
    uint32_t n_planes, n_parabs, n_prisms, n_objects, n_direct, n_direct_padded, cluster_base, n_clusters, cluster_k;
    uint32_t n_cluster_groups, n_prism_groups; // RlFlatScene: second level of the cull table
    uint32_t off_sphere_r2;                    // blob offset of RlSceneView::sphere_r2
    uint32_t off_prism_cyl, prism_cylinders;   // RlFlatScene::prism_cyl (2 records per prism) and whether to test them
    uint32_t group_gc;                         // RlFlatScene::group_gc: clusters per group of the cull table
    uint32_t small_ordered;                    // the paraboloids' objects all precede the planes' and circles' (rl_scan_wave: one compare per candidate)
    uint32_t n_cluster_supers, super_g;        // RlFlatScene: third level of the cull table (0: none); a scene that has one is never staged whole
    // Records [off_planes, off_objects) of the blob are the TABLES -- planes, paraboloids, prisms, the cull table, the camera:
    // everything a scan reads with wave-uniform addresses or once per (group, ray) / (prism, ray) pair, 10-40 KB whatever the
    // scene's size.  The spheres in front of them and the per-object arrays behind them grow with the scene: one too large for
    // LDS stages its tables only (RL_STAGE_TABLES).
};
enum { RL_STAGE_NONE = 0, RL_STAGE_TABLES = 1, RL_STAGE_ALL = 2 };

struct RlTraceJob {
    uint32_t width, height;
    float aspect_ratio;
    uint32_t stream;
    uint64_t seed;
    uint64_t first_path;     // plain launch: path index of offset 0
    uint64_t n_paths;        // plain launch: number of paths
    uint32_t grace_ticks;    // open launch: how long (10 ns ticks) it waits for another call once every call is complete
    uint32_t reserved;
    float wm1, hm1;          // (float)width - 1, (float)height - 1 (plot_unit.rs:60-61): see rl_splat_weights
};

// One call of an open launch: its path offsets [0, end) are the path indices first_path .. of the call's RNG stream and
// go to `target` (un-fused: the unit's mapped_photons; fused: the plot unit's tristimulus buffer).  end is a multiple
// of 64, so the 64 paths of a stash refill always belong to one call.
struct RlJobEntry {
    void* target;
    uint64_t first_path;
    uint64_t start, end; // start = 0
};

// ---- open launches ("sessions", rl_api.hip) -------------------------------------------------------
// A trace kernel that stays resident while the host keeps appending render calls to it: the reference's workers
// call TraceUnit::render for 524,288 paths at a time (trace_unit.rs:67) -- 0.15 ms of this chip's work followed by
// ~0.2 ms of waiting for the batch's longest paths -- so instead of one launch per call (or per group of calls
// that happen to wait at the same moment) the waves of ONE launch take the calls' paths from a job table that
// grows while they run, and every call is signalled complete on its own as soon as its last path has ended.
// Host <-> kernel protocol (RlOpenCtl lives in pinned, coherent host memory; RlOpenDev in device memory):
//   host:    writes ctl->jobs[k], then published = k + 1 (seq_cst), then reads closed_at.
//   kernel:  a wave that finds every known job handed out takes the `closing` lock, copies newly published
//            entries into dev->jobs and raises dev->known; if there are none it writes closed_at = known, fences,
//            re-reads published: unchanged -> final_at = known, dev->closed = 1 (the launch drains and ends);
//            changed -> closed_at = NONE (re-opened).
//   host:    job k is accepted iff it reads closed_at == NONE or closed_at > k; rejected (start a new launch) iff
//            final_at <= k; anything else is the kernel between its two steps: read again.
//   Both sides write-then-read with sequentially consistent fences, so at least one sees the other (Dekker).
//   Completion: waves count the paths of job j they finished in dev->fin[j] (a release at agent scope first, so the
//   photons / splats are visible to later kernels); whoever brings it to the job's size sets ctl->done[j].
#define RL_OPEN_CAP 256u
#define RL_OPEN_NONE 0xffffffffu
struct RlOpenDev {
    uint32_t known;   // entries [0, known) of jobs[] are valid
    uint32_t closed;  // no further jobs will be accepted
    uint32_t pad0[30]; // (polled by waves that are out of work: a cache line of their own)
    uint32_t closing; // lock: one wave at a time talks to the host
    uint32_t completed; // jobs whose last path has finished
    unsigned long long idle_since;  // wall_clock64() when a wave first found every job complete and nothing new (0: not idle)
    unsigned long long stuck_since; // ... when a wave first found nothing to hand out while calls were still finishing
    uint32_t pad1[26];
    uint32_t next[RL_OPEN_CAP]; // per job: next path offset to hand out
    uint32_t fin[RL_OPEN_CAP];  // per job: paths finished
    uint32_t seg[RL_OPEN_CAP];  // per job: segments (Scene::intersect calls) of its finished paths
    RlJobEntry jobs[RL_OPEN_CAP]; // start = 0, end = number of paths
};
struct RlOpenCtl {
    uint32_t published, closed_at, final_at, pad;
    uint32_t done[RL_OPEN_CAP];
    uint32_t segs[RL_OPEN_CAP]; // valid once done[j] is set
    RlJobEntry jobs[RL_OPEN_CAP];
};

// Diagnostic build only (make stats): wave-level event counters of the trace kernel, read back by
// tools/kernel_stats.py.  RL_STAT(k, v) adds v to counter k once per wave (lane 0).
#ifdef RL_STATS
enum { RL_ST_ITER, RL_ST_SCAN_LANES, RL_ST_A_ROUNDS, RL_ST_A_LANES, RL_ST_B_ROUNDS, RL_ST_B_LANES, RL_ST_P_ROUNDS, RL_ST_P_LANES,
       RL_ST_SHADE_DIFFUSE, RL_ST_SHADE_GLASS, RL_ST_SHADE_SOAP, RL_ST_END_EMITTER, RL_ST_END_VOID, RL_ST_ANY_GLASS, RL_ST_ANY_SOAP,
       RL_ST_ANY_COLOURED, RL_ST_ANY_GLOSSY, RL_ST_REFILLS, RL_ST_EMIT_BATCHES, RL_ST_EMIT_LANES, RL_ST_A_ITEMS, RL_ST_P_ITEMS,
       RL_ST_ANY_DIFFUSE, RL_ST_S_ROUNDS, RL_ST_S_LANES, RL_ST_S_ITEMS, RL_ST_P_SLOW,
       // shader cycles (s_memtime) a wave spent in each region of the main loop, summed over waves
       RL_ST_T_TOTAL, RL_ST_T_REFILL, RL_ST_T_SMALL, RL_ST_T_DIRECT, RL_ST_T_CLUSTER, RL_ST_T_TAIL, RL_ST_T_PRISM, RL_ST_T_SHADE,
       RL_ST_T_EMIT, RL_ST_T_A_ROUNDS, RL_ST_T_B_ROUNDS, RL_ST_T_P_ROUNDS, RL_ST_T_CAMERA, RL_ST_T_S_ROUNDS, RL_ST_COUNT };
__device__ unsigned long long rl_stat_counters[48];
// asm volatile + "memory": ordered against every LDS access, barrier and other timer read (the builtin may be
// hoisted or sunk by the optimiser); pure ALU work may still drift across a read by a few instructions.
__device__ __forceinline__ unsigned long long rl_cycles() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
#define RL_T0(V) const unsigned long long V = rl_cycles()
#define RL_T1(K, V) st[K] += rl_cycles() - (V)
#define RL_TACC_PARAM , unsigned long long* st
#define RL_TACC_ARG , st
// Counters live in (wave-uniform) registers of the wave and reach memory once, when the wave ends: one atomic
// per event from 4096 waves onto two dozen addresses made the diagnostic build 20x slower than the product.
#define RL_STAT(K, V) st[K] += (unsigned long long)(V)
#elif defined(RL_MARK) // region boundaries as comments in the ISA (tools/region_census.py counts the instructions between them; never shipped)
#define RL_STAT(K, V) do { } while (0)
#define RL_T0(V) asm volatile("; RL_MARK begin " #V)
#define RL_T1(K, V) asm volatile("; RL_MARK end " #V)
#define RL_TACC_PARAM
#define RL_TACC_ARG
#else
#define RL_STAT(K, V) do { } while (0)
#define RL_T0(V) do { } while (0)
#define RL_T1(K, V) do { } while (0)
#define RL_TACC_PARAM
#define RL_TACC_ARG
#endif

typedef __attribute__((address_space(3))) unsigned long long RlLdsU64;
typedef __attribute__((address_space(3))) uint32_t RlLdsU32;

// All LDS traffic of the scan stays inside one wave, whose DS instructions execute in order; this
// only has to stop the compiler from moving or caching LDS accesses across the hand-over points.
__device__ __forceinline__ void rl_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Number of set bits of `mask` below this lane (v_mbcnt_lo/hi).
__device__ __forceinline__ uint32_t rl_mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// base + that count: the instruction pair adds onto its third operand, so a ring's tail goes in there instead of a v_add behind it
__device__ __forceinline__ uint32_t rl_mbcnt_from(uint64_t mask, uint32_t base) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, base));
}

// Cross-lane fetches of a round: every lane reads the values of its pair's owner lane.  The bpermutes are issued back to
// back and waited for once -- the build's register-minimising scheduler otherwise issues them one at a time, each
// followed by its own s_waitcnt and its first use (six to nine LDS round trips in a row at the head of every round; the
// values are all live in the round's loop anyway, so nothing is saved by serialising them).
__device__ __forceinline__ void rl_fetch6(uint32_t owner, float a0, float a1, float a2, float a3, float a4, float a5, float& r0, float& r1,
                                          float& r2, float& r3, float& r4, float& r5) {
    const uint32_t addr = owner << 2;
    asm volatile("ds_bpermute_b32 %0, %6, %7\n\tds_bpermute_b32 %1, %6, %8\n\tds_bpermute_b32 %2, %6, %9\n\t"
                 "ds_bpermute_b32 %3, %6, %10\n\tds_bpermute_b32 %4, %6, %11\n\tds_bpermute_b32 %5, %6, %12\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5)
                 : "v"(addr), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5)
                 : "memory");
}
__device__ __forceinline__ void rl_fetch3(uint32_t owner, float a0, float a1, float a2, float& r0, float& r1, float& r2) {
    const uint32_t addr = owner << 2;
    asm volatile("ds_bpermute_b32 %0, %3, %4\n\tds_bpermute_b32 %1, %3, %5\n\tds_bpermute_b32 %2, %3, %6\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2)
                 : "v"(addr), "v"(a0), "v"(a1), "v"(a2)
                 : "memory");
}

// The conservative cull of rl_bound_pass() for the wave-uniform loops, in expanded form so that
// everything ray-dependent is hoisted out of the loop (8 float ops + one compare per bound):
//   with s = 0.999 |d|^2 and D = d / sqrt(s):   D.(c - o) = D.c + P,        P = -D.o
//   |c - o|^2 - R^2 = w - 2 o.c + |o|^2,                                    w = |c|^2 - R^2 (table)
//   pass <=> origin inside the bound, or the ray reaches it ahead of the origin
//        <=> max(d.co, 0)^2 - s (|co|^2 - R^2) >= 0  <=>  (w - 2 o.c) - max(D.co, 0)^2 <= -|o|^2
// -- a single float compare, so the ballot reads the compare mask directly (a compound condition
// would be materialised lane by lane first).  The expansion cancels, so the |o|^2 term carries a slack
// of 2e-5 (|o|^2 + max|c|^2) -- several times the worst rounding error of the product sums and of the far-bound
// terms below (each a few eps (|o|^2 + |c|^2) at any scene scale) -- which only ever lets MORE pairs through; the 0.999 in s covers the
// roundings of D (an approximate rsqrt, three products).  This is the build's own test (not reference
// arithmetic), so FMA is fine.  A lane without a path gets q = -inf and fails (its D is NaN and so is the left-hand side).
struct RlCullRay {
    RlF3 d;       // direction / sqrt(0.999 |direction|^2)
    float p;      // -d.o
    RlF3 m;       // -2 o
    float q;      // slack - |o|^2
    float len;    // length of `direction` in the units of d.(c - o), rounded up: a hit at ray parameter t is t * len away
};
__device__ __forceinline__ RlCullRay rl_cull_ray(RlF3 o, RlF3 dir, float cmax2, bool idle) {
    RlCullRay r;
    const float d2 = dir.x * dir.x + dir.y * dir.y + dir.z * dir.z;
    const float inv = __builtin_amdgcn_rsqf(d2 * 0.999f);
    r.d = rl_f3(dir.x * inv, dir.y * inv, dir.z * inv);
    r.p = -(r.d.x * o.x + r.d.y * o.y + r.d.z * o.z);
    const float o2 = o.x * o.x + o.y * o.y + o.z * o.z;
    r.m = rl_f3(-2.0f * o.x, -2.0f * o.y, -2.0f * o.z);
    r.q = idle ? -__builtin_inff() : 2.0e-5f * (o2 + cmax2) - o2;
    r.len = d2 * inv * 1.0001f;
    return r;
}
// `far` is the distance (in the units of RlCullRay::len) of the nearest hit the ray already has, rounded up: a bound
// the ray enters farther away than that cannot hold the nearest hit, and a hit at exactly that distance (scene.rs:51
// keeps the lower object index on a tie) lies inside its bound, i.e. not farther than where the ray enters it.
//   f(x) = x^2 - 2 x D.co + (|co|^2 - R^2) is <= 0 exactly between the two points where the ray's line crosses the bound;
//   with x = clamp(D.co, 0, far) -- the point of the segment [0, far] nearest to the bound's centre -- f(x) <= 0 says that
//   the segment reaches the bound: for x = D.co it is the discriminant test above, for x = 0 "the origin is inside",
//   for x = far "the entry point is not beyond the hit".  One v_med3 and one FMA more than the test without `far`.
// rl_cull_margin() >= 0 <=> the bound passes (rl_cull_pass); for a ray with a path its sign bit is set exactly where the test
// fails (x - x is +0; a never-reached record has w = +inf and a margin of -inf).  A ray WITHOUT a path has NaN terms and a
// margin whose sign says nothing: callers mask such lanes out themselves.
__device__ __forceinline__ float rl_cull_margin(const RlCullRay& r, RlF4 b, float far) {
    const float dd = __builtin_fmaf(r.d.z, b.z, __builtin_fmaf(r.d.y, b.y, __builtin_fmaf(r.d.x, b.x, r.p)));
    const float cs = __builtin_fmaf(r.m.z, b.z, __builtin_fmaf(r.m.y, b.y, __builtin_fmaf(r.m.x, b.x, b.w)));
    const float x = __builtin_amdgcn_fmed3f(dd, 0.0f, far);
    return r.q - __builtin_fmaf(x, __builtin_fmaf(-2.0f, dd, x), cs);
}
// the left-hand side of rl_cull_pass's compare (the bound passes iff it is <= r.q)
__device__ __forceinline__ float rl_cull_lhs(const RlCullRay& r, RlF4 b, float far) {
    const float dd = __builtin_fmaf(r.d.z, b.z, __builtin_fmaf(r.d.y, b.y, __builtin_fmaf(r.d.x, b.x, r.p)));
    const float cs = __builtin_fmaf(r.m.z, b.z, __builtin_fmaf(r.m.y, b.y, __builtin_fmaf(r.m.x, b.x, b.w)));
    const float x = __builtin_amdgcn_fmed3f(dd, 0.0f, far);
    return __builtin_fmaf(x, __builtin_fmaf(-2.0f, dd, x), cs);
}
// ... with its last FMA spelled out, for the loops that re-load `b` in place between the test and the push: the compiler's
// two-address form accumulates into b.w and then COPIES the result out of the way of the load (v_mov + v_fmac); the three-address
// instruction writes it where it can stay
__device__ __forceinline__ float rl_cull_lhs_apart(const RlCullRay& r, RlF4 b, float far) {
    const float dd = __builtin_fmaf(r.d.z, b.z, __builtin_fmaf(r.d.y, b.y, __builtin_fmaf(r.d.x, b.x, r.p)));
    const float cs = __builtin_fmaf(r.m.z, b.z, __builtin_fmaf(r.m.y, b.y, __builtin_fmaf(r.m.x, b.x, b.w)));
    const float x = __builtin_amdgcn_fmed3f(dd, 0.0f, far);
    const float y = __builtin_fmaf(-2.0f, dd, x);
    float lhs;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(lhs) : "v"(x), "v"(y), "v"(cs));
    return lhs;
}
__device__ __forceinline__ bool rl_cull_pass(const RlCullRay& r, RlF4 b, float far) {
    const float dd = __builtin_fmaf(r.d.z, b.z, __builtin_fmaf(r.d.y, b.y, __builtin_fmaf(r.d.x, b.x, r.p)));
    const float cs = __builtin_fmaf(r.m.z, b.z, __builtin_fmaf(r.m.y, b.y, __builtin_fmaf(r.m.x, b.x, b.w)));
    const float x = __builtin_amdgcn_fmed3f(dd, 0.0f, far);
    return __builtin_fmaf(x, __builtin_fmaf(-2.0f, dd, x), cs) <= r.q;
}

// Second bound of a prism (RlFlatScene::prism_cyl): does the ray's LINE pass within `radius` of the prism's axis?
//   |co . (D x a)| <= radius |D x a|   with co = c - o = c + m / 2 (m = -2 o), D the cull ray's (scaled) direction.
// The build's own conservative test (FMA allowed); the radius carries 5 % + 1e-3 and, here, 4e-6 |co|_1 for the rounding
// of the triple product far from the prism.  A ray along the axis, an unbounded prism (radius = inf) and NaNs pass.
__device__ __forceinline__ bool rl_cyl_pass(const RlCullRay& r, RlF4 c, RlF3 a) {
    const float cox = __builtin_fmaf(0.5f, r.m.x, c.x), coy = __builtin_fmaf(0.5f, r.m.y, c.y), coz = __builtin_fmaf(0.5f, r.m.z, c.z);
    const float crx = __builtin_fmaf(r.d.y, a.z, -(r.d.z * a.y)), cry = __builtin_fmaf(r.d.z, a.x, -(r.d.x * a.z)),
                crz = __builtin_fmaf(r.d.x, a.y, -(r.d.y * a.x));
    const float w = __builtin_fmaf(coz, crz, __builtin_fmaf(coy, cry, cox * crx));
    const float radius = __builtin_fmaf(4.0e-6f, fabsf(cox) + fabsf(coy) + fabsf(coz), c.w);
    const float cr2 = __builtin_fmaf(crz, crz, __builtin_fmaf(cry, cry, crx * crx));
    return !(w * w > radius * radius * cr2);
}

// Per-wave LDS scratch: the merge keys, three rings of deferred work, the rays' cull terms, the camera-ray stash and the
// emitter queue.  8 KB, a multiple of 512 bytes (RL_RING_SLOT).
struct RlWaveScratch {
    unsigned long long key[64]; // (bits(distance) << 32) | (object << 3 | half-space), min-merged
    uint32_t ring_a[128];       // (cluster or prism index << 6) | owner lane
    uint32_t ring_b[128];       // (sphere record position << 6) | owner lane
    uint32_t ring_s[128];       // (group index << 6) | owner lane: second level of the cull table
    // The cull terms of the wave's 64 rays, two 16-byte slots per lane: {d.xyz, p}, {m.xyz, q} (RlCullRay), and the far bounds.
    // Written by every lane once per scan; a round's lane reads the slot of its pair's OWNER with two 16-byte loads and one
    // 4-byte load.  Rounds 1-4 fetched the nine values across lanes with nine ds_bpermute_b32 -- ~430 cycles of a wave's
    // time per round measured in the kernel (tools/ab3.sh, the RL_EXP_EXTRA probes), against ~90 for the gathers; five such
    // rounds per iteration.  The room came from the emitter queue (128 -> 64 slots), the stash's path indices (derived
    // now) and the object table (one record per object).
    float terms_d[64][4]; // {d.xyz, p}: two arrays of 16-byte slots rather than one of 32-byte ones -- a 16-byte gather is served
    float terms_m[64][4]; // {m.xyz, q}  in groups of 16 lanes, and 16-byte strides spread 16 owners over all 16 bank quads
    // The far bound of the lane's ray: the distance of its nearest plane / circle / paraboloid hit (before the prisms: of its nearest
    // hit) in the cull's units -- or, in the variants for scenes that are not staged whole (RL_PROGRESSIVE_FAR), RlCullRay::len alone:
    // a round's far bound is then the distance of the nearest hit its owner has SO FAR -- the high word of the owner's merge key,
    // which every sphere-tail round lowers -- times this, so that bounds behind a sphere an earlier round of the SAME scan found are
    // culled like those behind the walls.  (Dense scenes: 5,000 random spheres 1.7 -> 3.6 Grays/s, 20,000: 0.41 -> 5.3; one LDS
    // read and one product per round, -0.3 % on the built-in scene, which keeps the plain form.)
    float far[64];
    // Stash of 64 freshly generated camera rays (SoA): ox oy oz dx dy dz wavelength sx sy ior.  Refilled with all 64 lanes
    // busy; slot s holds path (stash_path0 + s) of the RNG stream, the first stash_valid slots hold a path at all
    // (wave-uniform registers of rl_trace_body).
    float stash[10][64];
    // Fused mode: paths that ended on an emitter wait here (sx, sy, wavelength, intensity, emitter object)
    // until (about) 64 of them can be evaluated and splatted with a full exec mask.
    float emit[5][64];
};
static_assert(sizeof(RlWaveScratch) == 8192, "rl_scan_wave's ring addressing wants the wave's scratch 512-byte aligned");

typedef float RlV4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) RlV4 RlLdsV4;
typedef __attribute__((address_space(3))) float RlLdsF;
// A lane's cull terms (and far bound) into its slot of the wave's scratch, once per scan ...
template <bool PF = false>
__device__ __forceinline__ void rl_store_cull_ray(RlWaveScratch* ws, uint32_t lane, const RlCullRay& cr, float far) {
    ((RlLdsV4*)&ws->terms_d[0][0])[lane] = (RlV4){cr.d.x, cr.d.y, cr.d.z, cr.p};
    ((RlLdsV4*)&ws->terms_m[0][0])[lane] = (RlV4){cr.m.x, cr.m.y, cr.m.z, cr.q};
    ((RlLdsF*)&ws->far[0])[lane] = PF ? cr.len : far;
}
// ... and the OWNER lane's, for a round: two 16-byte gathers and a 4-byte one (the caller has passed a wave sync since the store).
template <bool PF = false>
__device__ __forceinline__ void rl_fetch_cull_ray(RlWaveScratch* ws, uint32_t owner, RlCullRay& r, float& r_far) {
    const RlV4 a = ((const RlLdsV4*)&ws->terms_d[0][0])[owner], b = ((const RlLdsV4*)&ws->terms_m[0][0])[owner];
    r_far = ((const RlLdsF*)&ws->far[0])[owner];
    if (PF) r_far *= rl_u2f(((const RlLdsU32*)&ws->key[0])[2u * owner + 1u]); // (the caller has passed a wave sync since the key's last merge)
    r.d = rl_f3(a.x, a.y, a.z);
    r.p = a.w;
    r.m = rl_f3(b.x, b.y, b.z);
    r.q = b.w;
    r.len = 0.0f;
}
// The OWNER lane's ray for an exact round (sphere tails, prisms): its origin out of the cull terms -- m = -2 o, and halving is exact
// (a denormal component of o would not come back, but scene coordinates are not denormals: every parity test runs through this) --
// and its direction across lanes: one 16-byte gather and three ds_bpermute_b32 instead of six.
__device__ __forceinline__ void rl_fetch_ray(RlWaveScratch* ws, uint32_t owner, RlF3 dir, RlF3& ro, RlF3& rd) {
    const RlV4 b = ((const RlLdsV4*)&ws->terms_m[0][0])[owner];
    rl_fetch3(owner, dir.x, dir.y, dir.z, rd.x, rd.y, rd.z);
    // (the slot's fourth word counts as used until the fetch's wait is over: the compiler otherwise hands its register to the next
    // LDS load of the round and puts an s_waitcnt -- one more exposed round trip -- in front of that load)
    asm volatile("" : : "v"(b.w));
    ro = rl_f3(b.x * -0.5f, b.y * -0.5f, b.z * -0.5f);
}

// Open launches: per-workgroup LDS area behind the waves' scratch.
struct RlOpenWg {
    uint32_t fin[RL_OPEN_CAP]; // per job: paths this workgroup finished (results acknowledged) and has not yet reported
    uint32_t seg[RL_OPEN_CAP]; // per job: segments of paths this workgroup ended and has not yet reported
    uint32_t flushed_at;       // wall clock (10 ns ticks, low 32 bits) of the last report to RlOpenDev
    uint32_t poll[2];          // RlOpenDev::known as last read by a wave of this workgroup, and when
    uint32_t pad;
};

// Scene::intersect (scene.rs:39-60) for the 64 rays of a wave.  Must be called by all 64 lanes in
// uniform control flow.  `idle_bit` is 0x80000000 on lanes without a path (0 otherwise); it is OR-ed into
// the integer reject/cull compares so such lanes never enqueue work (they also carry dir = 0).
//
// Every ray is tested against wave-uniform records (LDS broadcast or scalar loads) only up to a cheap
// reject test; expensive tails never run under divergence.  The (item, ray) pairs that survive are
// compacted with ballot/mbcnt into per-wave LDS rings and evaluated 64 pairs at a time, one pair per
// lane, with the ray fetched across lanes (ds_bpermute):
//   * direct spheres: reject = sign bits of the discriminant q and of d.co (16 flops + 3 int ops,
//     geometry.rs:204-216 in the scaled form of rl_core.h) -> ring B;
//   * sphere clusters (rl_scene.h): group bounds (wave-uniform) -> ring S; a ring-S round tests the group's cluster
//     bounds -> ring A; a ring-A round tests the cluster's members (RlSceneView::cluster_k of them), all with the conservative cull test
//     (rl_cull_pass: far bound included) -> ring B;
//   * ring B rounds: exact IEEE sqrt / root selection (geometry.rs:217-240), then min-merge;
//   * hexagonal prisms: group bounds -> ring S -> bounding spheres (-> ring B -> the prisms' cylinders, in scenes whose prisms
//     carry that second bound: process_cylinders) -> ring A; a round decides the Compound tree's
//     answer by margins (rl_hex_prism_fast, ~330 instructions) and evaluates the tree itself (rl_hex_prism, ~600) only
//     when one of its pairs is undecided, then min-merges.
// Results are min-merged per owning ray as 64-bit (distance bits, object index) keys in LDS: exactly
// scene.rs:51's strict `<` over objects in scan order, in any evaluation order.
template <bool CYL, bool SPLIT, bool UNROLL_S, bool HOIST_S, bool SPHERES_IN_LDS, bool TABLES_IN_LDS, bool SUPER>
__device__ __forceinline__ RlHit rl_scan_wave(const RlSceneView& sv, const RlF4* cull, const RlF4* prism_cyl, uint32_t group_gc, uint32_t small_ordered, float sv_cull_cmax2,
                                              uint32_t n_cluster_groups, uint32_t n_prism_groups, uint32_t n_cluster_supers, uint32_t super_g, RlLdsU32* ring_t, RlF3 o, RlF3 dir,
                                              uint32_t idle_bit, RlWaveScratch* ws, uint32_t lane RL_TACC_PARAM) {
    // Explicit LDS address space: generic pointers here would become flat_* accesses.
    constexpr bool PF = SUPER && RL_PROGRESSIVE_FAR;
    RlLdsU64* keys = (RlLdsU64*)ws->key;
    RlLdsU32* ring_a = (RlLdsU32*)ws->ring_a;
    RlLdsU32* ring_b = (RlLdsU32*)ws->ring_b;
    RlLdsU32* ring_s = (RlLdsU32*)ws->ring_s;
    // Wave-uniform ring indices: entries [lim - 64, tail) are waiting; a round runs when tail reaches lim (one compare per push).
    uint32_t a_lim = 64u, a_tail = 0, b_lim = 64u, b_tail = 0, s_lim = 64u, s_tail = 0;
    // A push writes slot ((tail + rank) & 127) of a ring: the wave's scratch is 512-byte aligned (rl_trace_body) and the rings are
    // 512 bytes each, so the slot's LDS address is (scratch address | ((tail + rank) << 2 & 0x1fc)) + the ring's offset -- one
    // v_add_lshl and one v_and_or behind the two v_mbcnt, the ring's offset in the store's immediate field.
    const uint32_t ws_addr = (uint32_t)(size_t)(RlLdsU32*)ws;
#define RL_RING_SLOT(RING, M, TAIL) \
    ((RlLdsU32*)(size_t)((ws_addr | (((rl_mbcnt(M) + (TAIL)) << 2) & 0x1fcu)) + (uint32_t)offsetof(RlWaveScratch, RING)))
    // RL_RING_PUSH: the lanes of mask M (the ballot of a test every lane of the wave ran: exec is all ones here, rl_scan_wave's
    // contract) write ENTRY to consecutive slots of a ring from TAIL on.  Spelled out because the compiler's form of
    // `if (pass) *slot = entry` is s_and_saveexec / s_cbranch_execz / ... / s_or exec around the store and re-materialises the
    // slot mask in a scalar register at every push -- four instructions of a wave's issue for each of the ~35
    // pushes of an iteration: here the mask becomes exec for the one store, without a branch.
#ifndef RL_PUSH_ASM
#define RL_PUSH_ASM 1
#endif
#if RL_PUSH_ASM
#define RL_RING_PUSH(RING, COND, M, TAIL, ENTRY)                                                                        \
    {                                                                                                                   \
        uint32_t slot_;                                                                                                 \
        /* (gfx950: a scalar register written by a vector instruction -- the ballot -- may be read by a vector instruction   \
           only two wait states later; the compiler does not look into this block, so the block starts with them) */        \
        asm volatile("s_mov_b64 exec, %6\n\ts_nop 0\n\tv_mbcnt_lo_u32_b32 %0, %1, 0\n\tv_mbcnt_hi_u32_b32 %0, %2, %0\n\t"         \
                     "v_add_lshl_u32 %0, %0, %3, 2\n\tv_and_or_b32 %0, %0, %4, %5\n\tds_write_b32 %0, %7 offset:%8\n\ts_mov_b64 exec, -1" \
                     : "=&v"(slot_)                                                                                     \
                     : "s"((uint32_t)(M)), "s"((uint32_t)((M) >> 32)), "s"(TAIL), "s"(ring_mask), "v"(ws_addr), "s"(M),  \
                       "v"(ENTRY), "n"(offsetof(RlWaveScratch, RING))                                                   \
                     : "memory");                                                                                       \
    }
#else
#define RL_RING_PUSH(RING, COND, M, TAIL, ENTRY) if (COND) *RL_RING_SLOT(RING, M, TAIL) = (ENTRY);
#endif
    // RL_LE_PUSH: test (LHS <= RHS, the cull's compare) AND push in one: v_cmpx writes the compare's mask to vcc and to exec at
    // once, the store runs under it, and the ring's tail advances by the mask's population -- one instruction less than a
    // compare into a scalar pair followed by RL_RING_PUSH and the count.
#ifndef RL_PUSH_CMPX
#define RL_PUSH_CMPX 1
#endif
#if RL_PUSH_ASM && RL_PUSH_CMPX
#define RL_LE_PUSH(RING, LHS, RHS, TAIL, ENTRY)                                                                        \
    {                                                                                                                   \
        uint32_t slot_, n_;                                                                                             \
        /* (the count and an s_nop are the two wait states between the compare's write of vcc and the first vector read of it) */ \
        asm volatile("v_cmpx_le_f32_e32 vcc, %2, %3\n\ts_bcnt1_i32_b64 %1, vcc\n\ts_nop 0\n\t"                              \
                     "v_mbcnt_lo_u32_b32 %0, vcc_lo, 0\n\tv_mbcnt_hi_u32_b32 %0, vcc_hi, %0\n\t"                            \
                     "v_add_lshl_u32 %0, %0, %4, 2\n\tv_and_or_b32 %0, %0, %5, %6\n\tds_write_b32 %0, %7 offset:%8\n\t"     \
                     "s_mov_b64 exec, -1"                                                                               \
                     : "=&v"(slot_), "=&s"(n_)                                                                           \
                     : "v"(LHS), "v"(RHS), "s"(TAIL), "s"(ring_mask), "v"(ws_addr), "v"(ENTRY),                          \
                       "n"(offsetof(RlWaveScratch, RING))                                                               \
                     : "vcc", "scc", "memory");                                                                         \
        TAIL += n_;                                                                                                     \
    }
#else
#define RL_LE_PUSH(RING, LHS, RHS, TAIL, ENTRY)                                                                        \
    {                                                                                                                   \
        const bool pass_ = (LHS) <= (RHS);                                                                              \
        const uint64_t m_ = __builtin_amdgcn_ballot_w64(pass_);                                                         \
        RL_RING_PUSH(RING, pass_, m_, TAIL, ENTRY)                                                                      \
        TAIL += (uint32_t)__popcll(m_);                                                                                 \
    }
#endif
    // ... and the same push into ring T (SUPER: the third level's (super, ray) pairs), which lives outside the wave's scratch -- 512 bytes
    // per wave between the staged tables and the scratch blocks, 512-byte aligned like the rings inside
    const uint32_t wt_addr = (uint32_t)(size_t)ring_t;
    uint32_t rt_lim = 64u, rt_tail = 0;
#if RL_PUSH_ASM && RL_PUSH_CMPX
#define RL_LE_PUSH_T(LHS, RHS, TAIL, ENTRY)                                                                             \
    {                                                                                                                   \
        uint32_t slot_, n_;                                                                                             \
        asm volatile("v_cmpx_le_f32_e32 vcc, %2, %3\n\ts_bcnt1_i32_b64 %1, vcc\n\ts_nop 0\n\t"                              \
                     "v_mbcnt_lo_u32_b32 %0, vcc_lo, 0\n\tv_mbcnt_hi_u32_b32 %0, vcc_hi, %0\n\t"                            \
                     "v_add_lshl_u32 %0, %0, %4, 2\n\tv_and_or_b32 %0, %0, %5, %6\n\tds_write_b32 %0, %7\n\t"               \
                     "s_mov_b64 exec, -1"                                                                               \
                     : "=&v"(slot_), "=&s"(n_)                                                                           \
                     : "v"(LHS), "v"(RHS), "s"(TAIL), "s"(ring_mask), "v"(wt_addr), "v"(ENTRY)                           \
                     : "vcc", "scc", "memory");                                                                         \
        TAIL += n_;                                                                                                     \
    }
#else
#define RL_LE_PUSH_T(LHS, RHS, TAIL, ENTRY)                                                                             \
    {                                                                                                                   \
        const bool pass_ = (LHS) <= (RHS);                                                                              \
        const uint64_t m_ = __builtin_amdgcn_ballot_w64(pass_);                                                         \
        if (pass_) ring_t[(rl_mbcnt(m_) + (TAIL)) & 127u] = (ENTRY);                                                    \
        TAIL += (uint32_t)__popcll(m_);                                                                                 \
    }
#endif
    uint32_t ring_mask = 0x1fcu;
    asm volatile("" : "+s"(ring_mask)); // (opaque: one scalar register for the scan instead of an s_movk at every push)
    const RlF4* sph = sv.spheres;
    // The scene's counts are launch constants.  Whatever is derived from them -- "is there any", "how many full groups of
    // four", the member loop's choice -- is loop-invariant over the kernel's persistent loop too, and the optimiser keeps
    // every such flag in a scalar register PAIR (a lane mask) across it: more than the register file holds, the rest is
    // spilled to lanes of a vector register and read back with v_readlane (half-rate VALU).  Opaque copies: the flags are
    // one s_cmp each where they are used.
    uint32_t n_parabs = sv.n_parabs, n_planes = sv.n_planes, n_direct = sv.n_direct, cluster_k = sv.cluster_k;
    asm volatile("" : "+s"(n_parabs), "+s"(n_planes), "+s"(n_direct), "+s"(cluster_k), "+s"(n_cluster_groups), "+s"(n_prism_groups), "+s"(group_gc), "+s"(small_ordered));

    // Paraboloids, planes and circles: a handful of records, evaluated in registers.
    // (round 6: a scene with ONE direct sphere -- the built-in scenes' sun -- has its record requested here, ahead of the small
    // primitives' arithmetic, and tested straight-line below instead of in the direct list's loop with its four-record prefetch)
    RlF4 first_direct = RlF4();
    if (SPHERES_IN_LDS && RL_DIRECT_ONE) {
        first_direct = sph[0]; // (the list is padded: record 0 always exists)
        asm volatile("" ::: "memory");
    }
    RlHit best;
    best.t = 1.0e12f; // scene.rs:43
    best.obj = RL_HIT_NONE;
    best.sub = 0;
    RL_T0(t_small);
    // scene.rs:51 keeps the FIRST object among equal distances: `t < best.t || (t == best.t && obj < best.obj)` in any evaluation
    // order.  Both record lists are in object order (rl_scene.cpp), so a plain `t < best.t` decides within each; it also decides
    // between them when every paraboloid's object precedes every plane's (RlSceneLayout::small_ordered, the built-in scenes) --
    // two compares, two mask operations and a branch less per candidate than the general form, which other scenes take.
#define RL_SMALL_PRIMITIVES(NEARER, NP, NL)                                                           \
    for (uint32_t i = 0; i < (NP); ++i) {                                                             \
        const RlF4 r0 = sv.parabs[3 * i], r1 = sv.parabs[3 * i + 1], r2 = sv.parabs[3 * i + 2];       \
        if (HOIST_S && SPLIT) asm volatile("" : : "v"(r1.w), "v"(r2.w)); /* records in LDS: 16-byte loads (rl_hex_prism_fast), a 12-byte LDS read takes twice the LDS time */ \
        const float t = rl_paraboloid_t(rl_xyz(r0), rl_xyz(r1), rl_xyz(r2), o, dir);                  \
        const uint32_t obj = rl_f2u(r0.w);                                                            \
        if (!(t < 0.0f) && NEARER(t, obj, best)) {                                                    \
            best.t = t;                                                                               \
            best.obj = obj;                                                                           \
        }                                                                                             \
    }                                                                                                 \
    for (uint32_t i = 0; i < (NL); ++i) {                                                             \
        const RlF4 r0 = sv.planes[2 * i], r1 = sv.planes[2 * i + 1];                                  \
        float dn;                                                                                     \
        const float t = rl_plane_t(rl_xyz(r0), rl_xyz(r1), o, dir, &dn);                              \
        bool hit = t > 0.0f;                                                                          \
        if (hit && r0.w >= 0.0f) {                                                                    \
            const RlF3 dp = rl_sub(rl_add(o, rl_mul(dir, t)), rl_xyz(r1));                            \
            hit = rl_dot(dp, dp) <= r0.w;                                                             \
        }                                                                                             \
        const uint32_t obj = rl_f2u(r1.w);                                                            \
        if (hit && NEARER(t, obj, best)) {                                                            \
            best.t = t;                                                                               \
            best.obj = obj;                                                                           \
        }                                                                                             \
    }
#define RL_NEARER_ORDERED(T, OBJ, BEST) ((T) < (BEST).t)
    // (three paraboloids and three planes / circles -- the room of every built-in scene, app.rs:166-236 -- get straight-line code: the
    // records' addresses are immediates, their loads can be requested ahead of the arithmetic that is in the way, and the two loops'
    // counters and branches go; any other count takes the loops)
    if (RL_SMALL_UNROLL && TABLES_IN_LDS && (small_ordered & 1u) != 0u && n_parabs == 3u && n_planes == 3u) { // (from global memory the records are scalar loads: two objects in flight cost 24 scalar registers the global-fetch variants do not have)
        // (bit 1 of the flag word: every normal of the six is along z -- the built-in room's are -- and the dot products with it are
        // one product each, rl_paraboloid_t<AXIS_Z>: ~50 instructions of the ~390 this block costs a wave per iteration)
        auto six = [&](auto axz) {
            constexpr bool AXZ = decltype(axz)::value;
        // One object's records in flight behind the previous object's arithmetic: six exposed LDS round trips become one.  (The
            // arithmetic has wave-uniform fallback branches -- short square root, one-division form -- so the scheduler, which works on
            // basic blocks, never moves a load up by itself; the fences keep the loads where they are written.)
            auto parab = [&](const RlF4& r0, const RlF4& r1, const RlF4& r2) {
                if (HOIST_S && SPLIT) asm volatile("" : : "v"(r1.w), "v"(r2.w)); /* 16-byte LDS loads */
                bool hit;
                const float t = rl_paraboloid_t<AXZ>(rl_xyz(r0), rl_xyz(r1), rl_xyz(r2), o, dir, &hit);
                const bool nearer = hit & (t < best.t);
                best.t = nearer ? t : best.t;
                best.obj = nearer ? rl_f2u(r0.w) : best.obj;
            };
            auto plane = [&](const RlF4& r0, const RlF4& r1) {
                float t;
                bool hit = rl_plane_hit<AXZ>(rl_xyz(r0), rl_xyz(r1), o, dir, &t);
                if (hit && r0.w >= 0.0f) {
                    const RlF3 dp = rl_sub(rl_add(o, rl_mul(dir, t)), rl_xyz(r1));
                    hit = rl_dot(dp, dp) <= r0.w;
                }
                const bool nearer = hit & (t < best.t);
                best.t = nearer ? t : best.t;
                best.obj = nearer ? rl_f2u(r1.w) : best.obj;
            };
#define RL_FENCE asm volatile("" ::: "memory")
            const RlF4 a0 = sv.parabs[0], a1 = sv.parabs[1], a2 = sv.parabs[2];
            RL_FENCE;
            const RlF4 b0 = sv.parabs[3], b1 = sv.parabs[4], b2 = sv.parabs[5];
            RL_FENCE;
            parab(a0, a1, a2);
            const RlF4 c0 = sv.parabs[6], c1 = sv.parabs[7], c2 = sv.parabs[8];
            RL_FENCE;
            parab(b0, b1, b2);
            const RlF4 d0 = sv.planes[0], d1 = sv.planes[1];
            RL_FENCE;
            parab(c0, c1, c2);
            const RlF4 e0 = sv.planes[2], e1 = sv.planes[3];
            RL_FENCE;
            plane(d0, d1);
            const RlF4 f0 = sv.planes[4], f1 = sv.planes[5];
            RL_FENCE;
            plane(e0, e1);
            plane(f0, f1);
#undef RL_FENCE
        };
        if ((small_ordered & 2u) != 0u) six(std::integral_constant<bool, true>());
        else six(std::integral_constant<bool, false>());
    } else if ((small_ordered & 1u) != 0u) {
        RL_SMALL_PRIMITIVES(RL_NEARER_ORDERED, n_parabs, n_planes)
    } else {
        RL_SMALL_PRIMITIVES(rl_nearer, n_parabs, n_planes)
    }
#undef RL_NEARER_ORDERED
#undef RL_SMALL_PRIMITIVES
    keys[lane] = ((unsigned long long)rl_f2u(best.t) << 32) |
                 (unsigned long long)(best.obj == RL_HIT_NONE ? 0xffffffffu : (best.obj << 3));
    RL_T1(RL_ST_T_SMALL, t_small);

    // The ray's cull terms (RlCullRay) and its far bound -- the nearest hit so far: here the planes, circles and paraboloids (in
    // the built-in scene the floor, the walls and the ceiling: every ray has one), before the prisms also the spheres -- go to this
    // lane's slots of the wave's scratch, where the rounds' lanes find them (behind their wave sync): every round below, the
    // sphere tails of the direct list included, reads its pairs' rays from there.
    RlCullRay cr = rl_cull_ray(o, dir, sv_cull_cmax2, idle_bit != 0u);
    float far = best.t * cr.len;
    rl_store_cull_ray<PF>(ws, lane, cr, far);
    // (RL_PROGRESSIVE_FAR) this lane's own far bound for the wave-uniform loops, refreshed behind a nested round: its key may have come nearer
#define RL_REFRESH_FAR() \
    if (PF) far = rl_u2f(((const RlLdsU32*)keys)[2u * lane + 1u]) * cr.len;

    // ---- ring B round: exact sphere tail for (record position, owner) pairs ----
    auto process_spheres = [&](uint32_t count) {
        RL_STAT(RL_ST_B_ROUNDS, 1);
        RL_STAT(RL_ST_B_LANES, count);
        RL_T0(t_b);
        rl_wave_sync();
        const uint32_t e = ring_b[(b_lim - 64u + lane) & 127u];
        const uint32_t owner = e & 63u;
        // (the record depends on the ring entry alone: its loads are issued ahead of the cross-lane fetch, whose wait then
        // covers both -- a wave's time is a third waiting for LDS round trips, DESIGN.md 4.2.  Entries beyond the round are
        // stale -- an earlier round's, a cylinder round's prism, an emitter batch's tag, or whatever the LDS held when the
        // kernel started: an LDS read beyond the allocation returns zero, a global one faults, so where the spheres are in
        // global memory such lanes read record 0)
        const uint32_t pos = (SPHERES_IN_LDS || lane < count) ? e >> 6 : 0u;
        RlF4 s;
        float s_r2;     // (a clustered sphere's s.w is its cull term)
        uint32_t s_obj;
        if (RL_W_B) s = sph[pos], s_r2 = sv.sphere_r2[pos], s_obj = sv.sphere_obj[pos];
        RlF3 fo, fd;
        rl_fetch_ray(ws, owner, dir, fo, fd);
        if (RL_W_B) asm volatile("" : : "v"(s.w)); // (a 16-byte LDS load: a 12-byte one takes twice the LDS time)
        const float ox = fo.x, oy = fo.y, oz = fo.z, dx = fd.x, dy = fd.y, dz = fd.z;
        if (lane < count) {
            if (!RL_W_B) s = sph[pos], s_r2 = sv.sphere_r2[pos];
            const float cox = s.x - ox, coy = s.y - oy, coz = s.z - oz;
            const float dd = dx * cox + dy * coy + dz * coz;
            const float c = (cox * cox + coy * coy + coz * coz) - s_r2;
            const float q = dd * dd - c;
            const float sq = rl_sqrtf(q, true); // |q| is rooted: a miss of the pre-tested pair (q < 0, geometry.rs:213-215) is tested for itself
            const float t1 = dd - sq;
            const float t2 = dd + sq;
            if ((q >= 0.0f) & (t1 > 0.0f) & (t1 < t2)) {
                const uint32_t obj = RL_W_B ? s_obj : sv.sphere_obj[pos];
                __hip_atomic_fetch_min(keys + owner, ((unsigned long long)rl_f2u(t1) << 32) | (unsigned long long)(obj << 3),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        rl_wave_sync();
        RL_T1(RL_ST_T_B_ROUNDS, t_b);
    };

    // Reject test of one sphere record S for the ray (OX.., DX..); survivors go to ring B as
    // (POS << 6) | OWNER.  The integer compare is a superset of (q >= 0 && d.co > 0) on the sign
    // bits; the ring-B round re-evaluates the exact float conditions.
#define RL_SPHERE_REJECT(S, POS, OWNER, DISABLE_BIT, OX, OY, OZ, DX, DY, DZ)                        \
    {                                                                                               \
        const float cox = (S).x - (OX), coy = (S).y - (OY), coz = (S).z - (OZ);                     \
        const float dd = (DX) * cox + (DY) * coy + (DZ) * coz;                                      \
        const float c = (cox * cox + coy * coy + coz * coz) - (S).w;                                \
        const float q = dd * dd - c;                                                                \
        const bool cand = (int)(rl_f2u(q) | (rl_f2u(dd) - 1u) | (DISABLE_BIT)) >= 0;               \
        const uint64_t m = __builtin_amdgcn_ballot_w64(cand);                                       \
        if (m != 0) {                                                                               \
            RL_RING_PUSH(ring_b, cand, m, b_tail, ((POS) << 6) | (OWNER))               \
            b_tail += (uint32_t)__popcll(m);                                                        \
            if (RL_UNLIKELY(b_tail >= b_lim)) {                                                           \
                process_spheres(64u);                                                               \
                b_lim += 64u;                                                                      \
            }                                                                                       \
        }                                                                                           \
    }

    // ---- direct spheres: every ray against every record; groups of 4 with one group of prefetch, then
    // the remainder one by one (the padding of rl_scene.h keeps every prefetch in bounds) ----
    RL_T0(t_direct);
    if (SPHERES_IN_LDS && RL_DIRECT_ONE && n_direct == 1u) {
        RL_SPHERE_REJECT(first_direct, 0u, lane, idle_bit, o.x, o.y, o.z, dir.x, dir.y, dir.z)
    } else if (n_direct != 0) {
        const uint32_t full = n_direct & ~3u;
        RlF4 c0 = sph[0], c1 = sph[1], c2 = sph[2], c3 = sph[3];
        for (uint32_t i = 0; i < full; i += 4) { // (each record is re-loaded in place once it has been tested: no copies, see RL_GROUP_CULLS)
            RL_SPHERE_REJECT(c0, i, lane, idle_bit, o.x, o.y, o.z, dir.x, dir.y, dir.z)
            c0 = sph[i + 4];
            RL_SPHERE_REJECT(c1, i + 1, lane, idle_bit, o.x, o.y, o.z, dir.x, dir.y, dir.z)
            c1 = sph[i + 5];
            RL_SPHERE_REJECT(c2, i + 2, lane, idle_bit, o.x, o.y, o.z, dir.x, dir.y, dir.z)
            c2 = sph[i + 6];
            RL_SPHERE_REJECT(c3, i + 3, lane, idle_bit, o.x, o.y, o.z, dir.x, dir.y, dir.z)
            c3 = sph[i + 7];
        }
        for (uint32_t i = full; i < n_direct; ++i) {
            uint32_t pos = i; // opaque: (full << 6) | lane is loop-invariant over the PERSISTENT loop too, gets hoisted out of
            asm volatile("" : "+s"(pos)); // it into a vector register that lives through the whole kernel -- and spills
            RL_SPHERE_REJECT(c0, pos, lane, idle_bit, o.x, o.y, o.z, dir.x, dir.y, dir.z)
            c0 = sph[i + 1];
        }
    }
    RL_T1(RL_ST_T_DIRECT, t_direct);

    RL_T0(t_cluster);
    // ---- ring A round for clusters: each lane runs one (cluster, ray) pair over the members.  The members are
    // tested with the cull's own arithmetic -- on the device a clustered sphere's record is {centre, |c|^2 - R^2}, see
    // RlSceneView::sphere_r2 -- (8 FMAs, a v_med3 and a compare per member, far bound included) -- a conservative pre-test: ring B re-evaluates the pairs that pass with the
    // reference's exact operations, so only "never drops a pair the reference would hit" matters here ----
    auto process_clusters = [&](uint32_t count) {
        RL_STAT(RL_ST_A_ROUNDS, 1);
        RL_STAT(RL_ST_A_LANES, count);
        RL_T0(t_a);
        rl_wave_sync();
        // SPLIT (the plain launches; the open ones have no registers to spare for it): a round of at most 32 pairs -- 18 % of
        // them, the flush at the end of the sphere pass -- gives every pair two lanes, each with half of the members: half the
        // loop for the same round.
        const uint32_t n_members = cluster_k; // wave-uniform, <= RL_CLUSTER_K_MAX
#ifdef RL_CLUSTER_K
        const bool split = SPLIT && count <= 32u && (n_members == 10u || n_members == 14u);
#else
        const bool split = SPLIT && count <= 32u;
#endif
        const uint32_t slot = split ? (lane & 31u) : lane;
        const uint32_t e = ring_a[(a_lim - 64u + slot) & 127u];
        const uint32_t owner = e & 63u;
        // lanes beyond the round hold stale ring entries: point them at cluster 0 so their (ignored) loads stay in bounds
        uint32_t first = sv.cluster_base + __umul24(cluster_k + 1u, slot < count ? (e >> 6) : 0u) + 1u; // (24-bit multiply: full rate)
        if (split) first += (lane >> 5) * (n_members >> 1);
        RlF4 mb;
        if (RL_W_M) mb = sph[first]; // (ahead of the cross-lane fetch: one wait for both)
        RlCullRay r;
        float r_far;
        rl_fetch_cull_ray<PF>(ws, owner, r, r_far);
        if (!RL_W_M) mb = sph[first];
        // The members that pass are collected as one bit per member in a lane-private mask (one v_alignbit per member: shift
        // left, take in the sign of the test's margin) and pushed afterwards, lowest set bit of every lane per step -- ~0.5 members pass
        // per pair, so two or three steps replace ten ballot / count / write sequences.
        uint32_t passed = 0;
        uint32_t n_mine = n_members; // members this lane tests (wave-uniform)
        // MEMBERS(N): the loop over N members from `first`, N a constant where the cluster size is one of those rl_scene.cpp chooses
        // from -- unrolled, the members' addresses are immediates -- and n_members otherwise (a build that forces another size: rolled, ~2 % slower).
#define RL_MEMBERS(N)                                                                                                    \
        {                                                                                                                \
            uint32_t failed = 0; /* one bit per member: the sign of the test's margin, shifted in with one v_alignbit */  \
            /* (spheres in global memory: TWO records of prefetch -- a member is a per-lane 16-byte gather from L2 / HBM, the loop is    \
               bound by that latency, and 57 % of the time of a 5,000-sphere scene went into these rounds with one) */          \
            RlF4 mb_ahead = RlF4();                                                                                      \
            if (!SPHERES_IN_LDS && RL_MEMBER_AHEAD2) mb_ahead = sph[first + 1];                                           \
            for (uint32_t j = 0; j < (N); ++j) {                                                                         \
                RlF4 mb_next; /* one record of prefetch (behind the last member: the next cluster's bound, or the blob's next array) */ \
                if (!SPHERES_IN_LDS && RL_MEMBER_AHEAD2) mb_next = mb_ahead, mb_ahead = sph[first + j + 2];               \
                else mb_next = sph[first + j + 1];                                                                       \
                if (RL_MEMBER_FENCE) asm volatile("" ::: "memory"); /* ONE record in flight: without it a tight register budget makes the scheduler request all N up front and spill them */ \
                failed = __builtin_amdgcn_alignbit(failed, rl_f2u(rl_cull_margin(r, mb, r_far)), 31u); /* member j ends up at bit N - 1 - j */ \
                mb = mb_next;                                                                                            \
            }                                                                                                            \
            passed = slot < count ? ~failed & ((1u << (N)) - 1u) : 0u; /* lanes beyond the round hold stale pairs (whose ray may have ended: NaN margins) */ \
            n_mine = (N);                                                                                                \
        }
#ifdef RL_CLUSTER_K // (a build that forces a cluster size, A/B: any size)
        if (n_members == 10u) {
            if (split) RL_MEMBERS(5u) else RL_MEMBERS(10u)
        } else if (n_members == 14u) {
            if (split) RL_MEMBERS(7u) else RL_MEMBERS(14u)
        } else RL_MEMBERS(n_members)
#else // RL_CLUSTER_K_CHOICES = {10, 14}: one compare decides (the three-way form was a tree of six scalar instructions per round; rl_scene_create refuses any other size)
        if (n_members == 10u) {
            if (split) RL_MEMBERS(5u) else RL_MEMBERS(10u)
        } else {
            if (split) RL_MEMBERS(7u) else RL_MEMBERS(14u)
        }
#endif
#undef RL_MEMBERS
        uint64_t any = __builtin_amdgcn_ballot_w64(passed != 0u);
        while (any != 0) {
            const uint32_t j = (n_mine - 1u) - (uint32_t)__builtin_ctz(passed | 0x80000000u); // (a lane with nothing left does not push; ctz(0) is undefined)
            RL_RING_PUSH(ring_b, passed != 0u, any, b_tail, ((first + j) << 6) | owner)
            b_tail += (uint32_t)__popcll(any);
            passed &= passed - 1u;
            if (RL_UNLIKELY(b_tail >= b_lim)) {
                process_spheres(64u);
                b_lim += 64u;
            }
            any = __builtin_amdgcn_ballot_w64(passed != 0u);
        }
        rl_wave_sync();
        RL_T1(RL_ST_T_A_ROUNDS, t_a);
    };

    // The cull table (rl_scene.h): level-1 bounds [clusters | prisms], then one group bound per group_gc clusters / RL_GROUP_GP prisms.
    // Every ray is tested against the GROUP bounds with wave-uniform records; the (group, ray) pairs that pass are
    // compacted into ring S and a ring-S round tests the group's members, one pair per lane with the
    // owner's cull terms fetched across lanes, pushing the members that pass to ring A (clusters or prisms).
    const uint32_t n_level1 = group_gc * n_cluster_groups + RL_GROUP_GP * n_prism_groups;
    // ---- ring S round.  PROCESS_A(count) runs a ring-A round; ITEM_BASE turns a cull-table index into the
    // cluster / prism number.
    /* one child of the pair's group: its bound (and, CYL, its cylinder) against the owner's ray; the pairs that pass go to ring A */ \
#define RL_GROUP_CHILD(J, COUNT, G, ITEM_BASE, PROCESS_A, CYL) RL_GROUP_CHILD_OF(cull[first + (J)], J, COUNT, G, ITEM_BASE, PROCESS_A, CYL)
#define RL_GROUP_CHILD_OF(BND, J, COUNT, G, ITEM_BASE, PROCESS_A, CYL)                                 \
    {                                                                                                   \
        const float lhs = rl_cull_lhs(r, (BND), r_far);                                                 \
        /* (no test for an empty mask: some pair of the round passes practically every child) */       \
        const uint32_t entry = ((first + (J) - (ITEM_BASE)) << 6) | owner;                              \
        RL_PUSH_##CYL(lhs, entry, PROCESS_A)                                                            \
    }
    /* a child that passed: to ring A ... */                                                            \
#define RL_PUSH_false(LHS, ENTRY, PROCESS_A)                                                            \
    RL_LE_PUSH(ring_a, LHS, r.q, a_tail, ENTRY)                                                         \
    if (RL_UNLIKELY(a_tail >= a_lim)) {                                                                 \
        PROCESS_A(64u);                                                                                 \
        a_lim += 64u;                                                                                   \
    }
    /* ... or, a prism of a scene whose prisms carry a second bound, to the cylinder round's ring (process_cylinders) */ \
#define RL_PUSH_CYL(LHS, ENTRY, PROCESS_A)                                                              \
    if (CYL) {                                                                                          \
        RL_LE_PUSH(ring_b, LHS, r.q, b_tail, ENTRY)                                                     \
        if (RL_UNLIKELY(b_tail >= b_lim)) {                                                             \
            process_cylinders(64u);                                                                     \
            b_lim += 64u;                                                                               \
        }                                                                                               \
    } else {                                                                                            \
        RL_PUSH_false(LHS, ENTRY, PROCESS_A)                                                            \
    }
    // The children's loop is unrolled where the scene is staged in LDS (the bounds' addresses become immediates, the loop's
    // counter and branch go: demo +0.8 %, glass +1.6 %, 513 objects +1.1 %; the round handlers inlined behind every child
    // double the kernel's code to ~110 KB, which the instruction cache takes) and rolled in the global-fetch variants, which
    // lose 2.5 % unrolled (their scalar registers are the tight resource).
#define RL_GROUP_ROUND(COUNT, G, ITEM_BASE, PROCESS_A, CYL)                                                    \
    {                                                                                                   \
        RL_STAT(RL_ST_S_ROUNDS, 1);                                                                     \
        RL_STAT(RL_ST_S_LANES, COUNT);                                                                  \
        RL_T0(t_s);                                                                                     \
        rl_wave_sync();                                                                                 \
        const uint32_t e = ring_s[(s_lim - 64u + lane) & 127u];                                              \
        const uint32_t owner = e & 63u;                                                                 \
        const uint32_t first = (ITEM_BASE) + __umul24((G), (lane < (COUNT)) ? (e >> 6) : 0u); /* stale entries: group 0 */ \
        /* (UNROLL_S: the children's bounds depend on the ring entry alone and are requested ahead of the cross-lane fetch, \
           whose wait covers them -- three LDS round trips less per round; the table has slack behind its last group) */ \
        RlF4 bnd_[4];                                                                                   \
        if (UNROLL_S && HOIST_S) {                                                                       \
            bnd_[0] = cull[first]; bnd_[1] = cull[first + 1u]; bnd_[2] = cull[first + 2u];              \
            if ((G) == 4u) bnd_[3] = cull[first + 3u];                                                  \
        }                                                                                               \
        /* (the table in global memory: the rolled loop below keeps two bounds requested ahead -- a child's bound is a per-lane    \
           gather from L2 / HBM -- the first two ahead of the cross-lane fetch) */                       \
        constexpr bool S_AHEAD = !UNROLL_S && !TABLES_IN_LDS && SPLIT && RL_S_AHEAD2; /* (plain launches: the open ones have no registers for it) */ \
        constexpr bool S_AHEAD_TWO = S_AHEAD;                                                           \
        RlF4 pa_ = RlF4(), pb_ = RlF4();                                                                \
        if (S_AHEAD) pa_ = cull[first];                                                                 \
        if (S_AHEAD_TWO) pb_ = cull[first + 1u];                                                        \
        RlCullRay r;                                                                                    \
        float r_far;                                                                                    \
        rl_fetch_cull_ray<PF>(ws, owner, r, r_far);                                                    \
        if (lane >= (COUNT)) r.q = -__builtin_inff(); /* lanes beyond the round never pass */           \
        if (S_AHEAD) {                                                                                  \
            _Pragma("nounroll") for (uint32_t j = 0; j < (G); ++j) {                                    \
                const RlF4 cur_ = pa_;                                                                  \
                if (S_AHEAD_TWO) pa_ = pb_, pb_ = cull[first + j + 2u]; /* (behind the last child: the next group's, the group bounds or the table's slack) */ \
                else pa_ = cull[first + j + 1u];                                                        \
                RL_GROUP_CHILD_OF(cur_, j, COUNT, G, ITEM_BASE, PROCESS_A, CYL)                         \
            }                                                                                           \
        } else if (UNROLL_S) {                                                                          \
            _Pragma("unroll") for (uint32_t j = 0; j < 4u; ++j) {                                       \
                if (j >= 3u && j >= (G)) break; /* groups hold 3 or 4 bounds (rl_scene.cpp) */          \
                if (HOIST_S) RL_GROUP_CHILD_OF(bnd_[j], j, COUNT, G, ITEM_BASE, PROCESS_A, CYL)          \
                else RL_GROUP_CHILD(j, COUNT, G, ITEM_BASE, PROCESS_A, CYL)                             \
            }                                                                                           \
        } else {                                                                                        \
            _Pragma("nounroll") for (uint32_t j = 0; j < (G); ++j) RL_GROUP_CHILD(j, COUNT, G, ITEM_BASE, PROCESS_A, CYL) \
        }                                                                                               \
        rl_wave_sync();                                                                                 \
        RL_T1(RL_ST_T_S_ROUNDS, t_s);                                                                   \
    }
    // ---- level 2, wave-uniform: group bounds [FIRST, FIRST + COUNT) of the cull table -> ring S ----
    /* Level 2 in two passes, for the PRISM groups: every group bound of a chunk of up to 31 tested first, the margins' signs      \
       shifted into a lane-private mask (as the cluster members' are); then the pairs that passed pushed, the lowest set bit of   \
       every lane per step.  A ray passes ~0.6 prism groups, so two or three steps replace eight pushes and the test loop is a   \
       bare one: built-in scene +0.65 %, glass +1.9 %.  The CLUSTER groups keep the push per group below: a ray passes 1.75 of   \
       them, five or six steps, and the same form lost 1.2 % there (round 5, tools/ab3.sh). */                                   \
#define RL_GROUP_CULLS_MASK(FIRST_GROUP, N_GROUPS, G, ITEM_BASE, PROCESS_A, CYL)                                \
    {                                                                                                   \
        for (uint32_t c0 = 0; c0 < (N_GROUPS); c0 += 31u) {                                             \
            const uint32_t n_here = (N_GROUPS) - c0 < 31u ? (N_GROUPS) - c0 : 31u;                      \
            const RlF4* gb = cull + n_level1 + (FIRST_GROUP) + c0;                                      \
            if (c0 != 0u) RL_REFRESH_FAR()                                                              \
            uint32_t failed = 0;                                                                        \
            _Pragma("unroll 4") for (uint32_t k = 0; k < n_here; ++k)                                   \
                failed = __builtin_amdgcn_alignbit(failed, rl_f2u(rl_cull_margin(cr, gb[k], far)), 31u); /* group k at bit n_here - 1 - k */ \
            uint32_t passed = idle_bit != 0u ? 0u : (~failed & ((1u << n_here) - 1u));                  \
            uint64_t any = __builtin_amdgcn_ballot_w64(passed != 0u);                                   \
            while (any != 0) {                                                                          \
                const uint32_t k = (n_here - 1u) - (uint32_t)__builtin_ctz(passed | 0x80000000u);       \
                RL_RING_PUSH(ring_s, passed != 0u, any, s_tail, ((c0 + k) << 6) | lane)          \
                s_tail += (uint32_t)__popcll(any);                                                      \
                passed &= passed - 1u;                                                                  \
                if (RL_UNLIKELY(s_tail >= s_lim)) {                                                     \
                    RL_GROUP_ROUND(64u, G, ITEM_BASE, PROCESS_A, CYL)                                   \
                    s_lim += 64u;                                                                       \
                }                                                                                       \
                any = __builtin_amdgcn_ballot_w64(passed != 0u);                                        \
            }                                                                                           \
        }                                                                                               \
        if (s_tail != s_lim - 64u) {                                                                    \
            const uint32_t left = s_tail - (s_lim - 64u);                                               \
            RL_GROUP_ROUND(left, G, ITEM_BASE, PROCESS_A, CYL)                                          \
            s_lim = s_tail + 64u;                                                                       \
        }                                                                                               \
    }
#define RL_GROUP_CULLS(FIRST_GROUP, N_GROUPS, G, ITEM_BASE, PROCESS_A, CYL)                                     \
    {                                                                                                   \
        const RlF4* gb = cull + n_level1 + (FIRST_GROUP);                                               \
        RlF4 g0 = gb[0];                                                                                \
        uint32_t g_entry = lane; /* (g << 6) | lane, kept running: one v_add per group instead of a shift-or from a scalar */ \
        for (const RlF4* const g_end = gb + (N_GROUPS); gb != g_end;) {                                 \
            const float lhs = rl_cull_lhs_apart(cr, g0, far);                                           \
            /* the next bound goes into the registers this one has just left (the table has slack at its end): loaded one   \
               test ahead into registers of its own it had to be COPIED over g0 every time round -- four moves and an address \
               per eleven-instruction test; the push below and the other waves cover the load (demo +1.0 %, glass +2 %) */  \
            g0 = *++gb;                                                                                 \
            /* (no `if (m != 0)` around the push: with 64 rays per wave some lane passes practically every group bound, and the   \
               test would be one more instruction per group) */                                         \
            RL_LE_PUSH(ring_s, lhs, cr.q, s_tail, g_entry) /* group number within its kind */            \
            g_entry += 64u;                                                                             \
            if (RL_UNLIKELY(s_tail >= s_lim)) {                                                         \
                RL_GROUP_ROUND(64u, G, ITEM_BASE, PROCESS_A, CYL)                                       \
                s_lim += 64u;                                                                           \
                RL_REFRESH_FAR()                                                                        \
            }                                                                                           \
        }                                                                                               \
        if (s_tail != s_lim - 64u) {                                                                    \
            const uint32_t left = s_tail - (s_lim - 64u);                                               \
            RL_GROUP_ROUND(left, G, ITEM_BASE, PROCESS_A, CYL)                                          \
            s_lim = s_tail + 64u;                                                                       \
        }                                                                                               \
    }
    // ---- level 3 (SUPER; round 6): the cluster groups of a scene with thousands of spheres.  Every ray is tested against the SUPER
    // bounds (one per super_g consecutive groups, behind the group bounds in the table) with wave-uniform records; the
    // (super, ray) pairs that pass are compacted into ring T, and a ring-T round tests the super's group bounds, one pair per
    // lane, pushing the groups that pass to ring S -- from where everything goes on as above.  A round's lane state (the
    // owner's cull terms, the running addresses) is re-derived from the ring entry after every nested ring-S round instead
    // of being held across it: the nesting T -> S -> A -> B is one level deeper than the registers were budgeted for.
    // One copy of the round (the loop is written so that the full rounds and the final partial one share it), and inside it
    // one copy of the ring-S round.
#define RL_SUPER_ROUND(COUNT, G, ITEM_BASE, PROCESS_A, CYL)                                             \
    {                                                                                                   \
        rl_wave_sync();                                                                                 \
        for (uint32_t j_ = 0;;) { /* j_: children done (wave-uniform) */                                \
            const uint32_t te = ring_t[(rt_lim - 64u + lane) & 127u];                                    \
            const uint32_t towner = te & 63u;                                                           \
            const uint32_t tgroup = __umul24(super_g, lane < (COUNT) ? (te >> 6) : 0u) + j_; /* stale entries: super 0 */ \
            const RlF4* tp = cull + n_level1 + tgroup;                                                  \
            RlF4 tb = tp[0]; /* ahead of the cross-lane fetch: one wait for both */                     \
            RlCullRay tr;                                                                               \
            float tr_far;                                                                               \
            rl_fetch_cull_ray<PF>(ws, towner, tr, tr_far);                                                  \
            if (lane >= (COUNT)) tr.q = -__builtin_inff(); /* lanes beyond the round never pass */      \
            uint32_t s_entry = (tgroup << 6) | towner;                                                  \
            bool s_full = false;                                                                        \
            while (j_ < super_g) {                                                                      \
                const float lhs = rl_cull_lhs_apart(tr, tb, tr_far);                                    \
                tb = *++tp; /* (behind the last group: the prism groups, the supers or the table's slack) */ \
                RL_LE_PUSH(ring_s, lhs, tr.q, s_tail, s_entry)                                          \
                s_entry += 64u;                                                                         \
                j_ += 1u;                                                                               \
                if (RL_UNLIKELY(s_tail >= s_lim)) {                                                     \
                    s_full = true;                                                                      \
                    break;                                                                              \
                }                                                                                       \
            }                                                                                           \
            if (!s_full) break;                                                                         \
            RL_GROUP_ROUND(64u, G, ITEM_BASE, PROCESS_A, CYL)                                           \
            s_lim += 64u;                                                                               \
            if (j_ >= super_g) break;                                                                   \
        }                                                                                               \
        rl_wave_sync();                                                                                 \
    }
#define RL_SUPER_CULLS(N_SUPERS, G, ITEM_BASE, PROCESS_A, CYL)                                          \
    {                                                                                                   \
        /* (one scalar -- the next super's number -- lives across a round; addresses, the prefetched bound and the running ring entry  \
           are derived from it again behind each) */                                                    \
        for (uint32_t s_next = 0;;) {                                                                   \
            const RlF4* sb = cull + (n_level1 + n_cluster_groups + n_prism_groups + s_next);            \
            RlF4 g0 = *sb;                                                                              \
            uint32_t t_entry = (s_next << 6) | lane;                                                    \
            while (s_next != (N_SUPERS) && rt_tail < rt_lim) {                                            \
                const float lhs = rl_cull_lhs_apart(cr, g0, far);                                       \
                g0 = *++sb; /* (behind the last super: the table's slack) */                            \
                RL_LE_PUSH_T(lhs, cr.q, rt_tail, t_entry)                                                \
                t_entry += 64u;                                                                         \
                s_next += 1u;                                                                           \
            }                                                                                           \
            const uint32_t waiting = rt_tail - (rt_lim - 64u);                                            \
            if (waiting == 0u) break; /* (only once the supers are exhausted) */                        \
            const uint32_t count = waiting < 64u ? waiting : 64u;                                       \
            RL_SUPER_ROUND(count, G, ITEM_BASE, PROCESS_A, CYL)                                         \
            rt_lim += count;                                                                             \
            RL_REFRESH_FAR()                                                                            \
        }                                                                                               \
        if (s_tail != s_lim - 64u) {                                                                    \
            const uint32_t left = s_tail - (s_lim - 64u);                                               \
            RL_GROUP_ROUND(left, G, ITEM_BASE, PROCESS_A, CYL)                                          \
            s_lim = s_tail + 64u;                                                                       \
        }                                                                                               \
    }
    // ---- sphere clusters: group culls -> ring S -> cluster bounds -> ring A -> members -> ring B ----
    if (n_cluster_groups != 0) {
        if (SUPER) asm volatile("" : "+s"(n_cluster_supers), "+s"(super_g)); // (opaque here, where they are used: see the counts above)
        if (SUPER && n_cluster_supers != 0u) {
            RL_SUPER_CULLS(n_cluster_supers, group_gc, 0u, process_clusters, false)
        } else {
            RL_GROUP_CULLS(0u, n_cluster_groups, group_gc, 0u, process_clusters, false)
        }
        RL_STAT(RL_ST_S_ITEMS, s_tail);
        if (a_tail != a_lim - 64u) process_clusters(a_tail - (a_lim - 64u));
        a_lim = a_tail + 64u;
        RL_STAT(RL_ST_A_ITEMS, a_tail);
    }
#ifdef RL_STATS
    const uint32_t a_head_after_clusters = a_tail;
#endif
#undef RL_SPHERE_REJECT
    RL_T1(RL_ST_T_CLUSTER, t_cluster);
    RL_T0(t_tailflush);
    if (b_tail != b_lim - 64u) process_spheres(b_tail - (b_lim - 64u));
    b_lim = b_tail + 64u;
    RL_T1(RL_ST_T_TAIL, t_tailflush);
    RL_T0(t_prism);
    far = rl_u2f((uint32_t)(keys[lane] >> 32)) * cr.len; // every sphere has been merged by now (process_spheres ends in a wave sync)
    if (!PF) ((RlLdsF*)&ws->far[0])[lane] = far;

    // ---- hexagonal prisms: cull -> compact -> evaluate -> merge ----
    auto process_prisms = [&](uint32_t count) {
        RL_STAT(RL_ST_P_ROUNDS, 1);
        RL_STAT(RL_ST_P_LANES, count);
        RL_T0(t_p);
        rl_wave_sync();
        uint32_t e = ring_a[(a_lim - 64u + lane) & 127u];
        uint32_t owner = e & 63u;
        const uint32_t prism = e >> 6;
        const RlF4* pr = sv.prisms + __umul24((uint32_t)RL_PRISM_STRIDE, lane < count ? prism : 0u);
        // (the plain launches of a scene whose prisms are staged in LDS: the first plane's records depend on the ring entry alone
        // and are requested ahead of the cross-lane fetch, whose wait covers them)
        constexpr bool PRE = HOIST_S && SPLIT && RL_W_PR;
        RlF4 pre_n = RlF4(), pre_off = RlF4();
        if (PRE) pre_n = pr[0], pre_off = pr[1];
        RlF3 ro, rd;
        rl_fetch_ray(ws, owner, dir, ro, rd);
        // The shortcut of rl_core.h decides all but ~0.1 % of the pairs (near an edge, grazing, within rounding of a face);
        // a round that holds one of those evaluates the reference's Compound tree instead -- for every lane, the branch is
        // wave-uniform, and with the same result for the lanes the shortcut had decided.
        RlCand c;
        int status = rl_hex_prism_fast<HOIST_S && SPLIT, PRE>(pr, ro, rd, &c, pre_n, pre_off);
        if (lane >= count) status = RL_PRISM_MISS;
        if (__builtin_amdgcn_ballot_w64(status == RL_PRISM_UNSURE) != 0) {
            RL_STAT(RL_ST_P_SLOW, 1);
            c = rl_hex_prism(pr, ro, rd);
            status = ((lane < count) & (c.t >= 0.0f)) ? RL_PRISM_HIT : RL_PRISM_MISS;
            // (the tree holds 8 normals, 8 offsets and 8 distances at once and sets the kernel's register count: this lane's own
            // cull terms are read back from the wave's scratch behind it instead of being held across it ...)
            {
                asm volatile("" ::: "memory");
                const float keep_len = cr.len;
                rl_fetch_cull_ray<PF>(ws, lane, cr, far);
                cr.len = keep_len;
            }
            // (... and so is the pair's ring entry)
            uint32_t again = a_lim - 64u; // (opaque: computed here, not carried across the tree either)
            asm volatile("" : "+s"(again) : : "memory");
            e = ring_a[(again + lane) & 127u];
            owner = e & 63u;
            pr = sv.prisms + __umul24((uint32_t)RL_PRISM_STRIDE, lane < count ? (e >> 6) : 0u);
        }
        if (status == RL_PRISM_HIT) {
            const uint32_t obj = rl_f2u(pr[1].w);
            const unsigned long long k = ((unsigned long long)rl_f2u(c.t) << 32) | (unsigned long long)((obj << 3) | c.k);
            __hip_atomic_fetch_min(keys + owner, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        rl_wave_sync();
        RL_T1(RL_ST_T_P_ROUNDS, t_p);
    };
    // ---- the prisms' second bound (CYL), one (prism, ray) pair per lane: the pairs that passed a prism's bounding sphere in a
    // ring-S round wait in ring B -- empty while the prisms are scanned -- and the ones whose ray's line comes within the
    // prism's cylinder go on to ring A.  Tested here, compacted, instead of behind every sphere test of the ring-S rounds (26
    // instructions for each of a group's three children whatever the sphere said): glass scene +0.7 %.
    auto process_cylinders = [&](uint32_t count) {
        rl_wave_sync();
        const uint32_t e = ring_b[(b_lim - 64u + lane) & 127u];
        const uint32_t owner = e & 63u;
        const RlF4* cy = prism_cyl + 2u * (lane < count ? (e >> 6) : 0u);
        const RlF4 cy0 = cy[0], cy1 = cy[1]; // (ahead of the cross-lane fetch: one wait for both)
        RlCullRay r;
        float r_far_unused;
        rl_fetch_cull_ray<PF>(ws, owner, r, r_far_unused);
        const bool pass = lane < count && rl_cyl_pass(r, cy0, rl_xyz(cy1));
        const uint64_t m = __builtin_amdgcn_ballot_w64(pass);
        if (m != 0) {
            RL_RING_PUSH(ring_a, pass, m, a_tail, e)
            a_tail += (uint32_t)__popcll(m);
            if (RL_UNLIKELY(a_tail >= a_lim)) {
                process_prisms(64u);
                a_lim += 64u;
            }
        }
        rl_wave_sync();
    };
    if (n_prism_groups != 0) {
        RL_GROUP_CULLS_MASK(n_cluster_groups, n_prism_groups, RL_GROUP_GP, group_gc * n_cluster_groups, process_prisms, CYL)
    }
#undef RL_REFRESH_FAR
#undef RL_SUPER_CULLS
#undef RL_SUPER_ROUND
#undef RL_LE_PUSH_T
#undef RL_GROUP_CULLS
#undef RL_GROUP_CULLS_MASK
#undef RL_GROUP_ROUND
#undef RL_GROUP_CHILD
#undef RL_GROUP_CHILD_OF
#undef RL_PUSH_CYL
#undef RL_PUSH_false
#undef RL_RING_SLOT
#undef RL_RING_PUSH
#undef RL_LE_PUSH
    if (CYL && b_tail != b_lim - 64u) {
        process_cylinders(b_tail - (b_lim - 64u));
        b_lim = b_tail + 64u;
    }
    if (a_tail != a_lim - 64u) process_prisms(a_tail - (a_lim - 64u));
    RL_T1(RL_ST_T_PRISM, t_prism);
#ifdef RL_STATS
    {
        uint32_t cluster_items = 0;
        if (n_cluster_groups != 0) cluster_items = a_head_after_clusters;
        RL_STAT(RL_ST_P_ITEMS, a_tail - cluster_items);
    }
#endif

    rl_wave_sync();
    const unsigned long long k = keys[lane];
    const uint32_t low = (uint32_t)k;
    best.t = rl_u2f((uint32_t)(k >> 32));
    if (low == 0xffffffffu) {
        best.obj = RL_HIT_NONE;
        best.sub = 0;
    } else {
        best.obj = low >> 3;
        best.sub = low & 7u;
    }
    return best;
}

// queue[0] = next unassigned path offset of this launch (zeroed before each launch),
// queue[1] = cumulative segments, queue[2] = cumulative paths.
// Dynamic LDS: [scene blob (RL_STAGE_ALL) or its tables (RL_STAGE_TABLES)][RlWaveScratch x 16].
// FUSED: paths that end on a light are splatted into `plot` (photons unused); otherwise every path's
// MappedPhoton goes to `photons` (plot unused).  A compile-time switch so neither variant carries the
// other's code and registers.
// OPEN: an open launch (see RlOpenDev above): the paths come from a job table that grows while the kernel runs, every lane
// remembers which job its path belongs to, and finished paths are counted per job.  A compile-time switch so that a
// plain launch -- every bulk launch -- carries none of the bookkeeping (measured 1 % on the fused kernel).
// CYL: the scene's prisms carry a second bound (RlFlatScene::prism_cylinders) -- a compile-time switch so that scenes
// without it run exactly the code they ran before it existed.
template <int STAGE, bool FUSED, bool OPEN, bool CYL>
__device__ __forceinline__ void rl_trace_body(const RlF4* __restrict__ scene, const RlSceneLayout& lay, const RlTraceJob& job, RlMappedPhoton* __restrict__ photons,
                                              float* __restrict__ plot, unsigned long long* __restrict__ queue, const RlJobEntry* jobs, RlOpenDev* od, RlOpenCtl* ctl) {
    extern __shared__ __attribute__((aligned(512))) RlF4 smem[]; // (512: the ring pushes OR slot offsets into a wave's scratch address, RL_RING_SLOT)
    const RlF4* base = scene; // the tables
    const RlF4* big = scene;  // the per-sphere and per-object arrays
    RlWaveScratch* scratch = (RlWaveScratch*)smem;
    if (STAGE == RL_STAGE_ALL) {
        for (uint32_t i = threadIdx.x; i < lay.total_f4; i += RL_TRACE_BLOCK) smem[i] = scene[i];
        __syncthreads();
        base = big = smem;
        scratch = (RlWaveScratch*)(smem + ((lay.total_f4 + 31u) & ~31u)); // 512-byte aligned (rl_scan_wave's ring addressing; stage_of() rounds up likewise)
    } else if (STAGE == RL_STAGE_TABLES) {
        const uint32_t n_staged = lay.off_objects - lay.off_planes;
        for (uint32_t i = threadIdx.x; i < n_staged; i += RL_TRACE_BLOCK) smem[i] = scene[lay.off_planes + i];
        __syncthreads();
        base = smem;
        scratch = (RlWaveScratch*)(smem + ((n_staged + 31u) & ~31u));
    }
    // A scene whose cull table has a third level (never one that is staged whole) gets 512 bytes per wave in front of the scratch
    // blocks: ring T (rl_scan_wave, SUPER).  The host sizes the launch's LDS accordingly (rl_api.hip: ring_t_bytes).
    RlLdsU32* ring_t = nullptr;
    if (STAGE != RL_STAGE_ALL && lay.n_cluster_supers != 0u) {
        ring_t = (RlLdsU32*)scratch + 128u * (threadIdx.x >> 6);
        scratch = (RlWaveScratch*)((RlF4*)scratch + 32u * (RL_TRACE_BLOCK / 64));
    }
    const uint32_t tab0 = STAGE == RL_STAGE_TABLES ? lay.off_planes : 0u; // blob offset of `base`'s first record

    RlSceneView sv;
    sv.spheres = big;
    sv.planes = base + (lay.off_planes - tab0);
    sv.parabs = base + (lay.off_parabs - tab0);
    sv.prisms = base + (lay.off_prisms - tab0);
    sv.objects = big + lay.off_objects;
    sv.cie = big + lay.off_cie;
    sv.sphere_obj = (const uint32_t*)(big + lay.off_sphere_obj);
    sv.sphere_r2 = (const float*)(big + lay.off_sphere_r2);
    sv.n_direct = lay.n_direct;
    sv.n_direct_padded = lay.n_direct_padded;
    sv.cluster_base = lay.cluster_base;
    sv.n_clusters = lay.n_clusters;
    sv.cluster_k = lay.cluster_k;
    sv.n_planes = lay.n_planes;
    sv.n_parabs = lay.n_parabs;
    sv.n_prisms = lay.n_prisms;
    sv.n_objects = lay.n_objects;
    sv.camera_rec = base + (lay.off_camera - tab0);
    sv.records = big; // (a tables-only stage completes its hits from the blob in global memory, like its spheres)
    const uint32_t lane = threadIdx.x & 63u;
    RlWaveScratch* ws = &scratch[threadIdx.x >> 6];
#if defined(RL_STATS) || defined(RL_DEBUG_ALIGN)
    // (the ring addressing is only right on a 512-byte aligned scratch, i.e. with the dynamic LDS at offset 0 of the
    // workgroup's allocation and nothing static in front of it; the diagnostic builds stop here if that ever changes)
    if (((uint32_t)(size_t)(RlLdsU32*)ws & 511u) != 0u) __builtin_trap();
#endif
    typedef __attribute__((address_space(3))) float RlLdsF32;
    RlLdsF32* stash = (RlLdsF32*)&ws->stash[0][0];

    uint64_t chunk_next = 0, chunk_end = 0;     // wave-uniform: this wave's slice of the global queue
    uint32_t stash_head = 0, stash_count = 0;   // wave-uniform
    uint64_t stash_path0 = 0;                   // wave-uniform: slot s of the stash holds path stash_path0 + s of the RNG stream ...
    uint32_t stash_valid = 0;                   // ... if s < stash_valid (the tail of a launch / call: fewer than 64 paths left)
    bool drained = false;                       // wave-uniform: the queue has no more paths for this wave
    bool active = false;
    uint64_t my_path = 0;  // the path's index in its RNG stream
    uint32_t my_job = 0;   // open launches: which job the path belongs to
    uint32_t stash_job = 0;               // wave-uniform, open launches: the job of the stash's current content
    uint64_t stash_first = job.first_path; // wave-uniform: path index of launch offset 0 as seen by that job
    RlPath p;
    p.origin = rl_f3(0.0f, 0.0f, 0.0f); // a lane without a path scans a null ray; rl_scan_wave's idle_bit mutes it
    p.direction = rl_f3(0.0f, 0.0f, 0.0f);
    p.wavelength = 0.0f;
    p.intensity = 0.0f;
    p.continue_chance = 0.0f;
    p.sx = p.sy = 0.0f;
    p.ior = 1.0f;
    p.bounce = 0;
    uint32_t segments = 0, paths_done = 0;
    // OPEN.  Finished paths are counted per job in the workgroup's LDS area and reported to the device-wide counters by
    // one wave at a time, every ~30 us: every wave telling RlOpenDev itself (thousands of same-address atomics per job
    // and more) measured 15-35 % of the kernel's time -- the returning atomics queue up and stall the waves.
    // The results a count stands for must be visible to whatever kernel the host launches once the job is reported
    // complete.  They are written as agent-scope atomics (write-through stores, memory-side float adds), and a path
    // is counted one iteration after its result was issued, behind an explicit s_waitcnt vmcnt(0) (settle() below; a release
    // at agent scope would write the whole L2 back, buffer_wbl2, every time).
    uint32_t known_local = 0;          // wave-uniform: jobs known to this wave
    bool pend_me = false;              // this lane's path finished in the last iteration (my_job is still its job)
    bool pend_any = false;             // wave-uniform: some lane's did
    uint32_t emit_pend = 0; // wave-uniform: the size of the emitter batch splatted in the last iteration (its calls: ring B)
    RlOpenWg* wgp = (RlOpenWg*)(scratch + RL_TRACE_BLOCK / 64);
    RlLdsU32* wg_fin = (RlLdsU32*)&wgp->fin[0];
    RlLdsU32* wg_seg = (RlLdsU32*)&wgp->seg[0];
    RlLdsU32* wg_flushed_at = (RlLdsU32*)&wgp->flushed_at;
    RlLdsU32* wg_poll = (RlLdsU32*)&wgp->poll[0];
    RlLdsF32* emit = (RlLdsF32*)&ws->emit[0][0];
    auto emit_put = [&](uint32_t row, uint32_t slot, float v) {
        emit[row * 64u + slot] = v;
    };
    auto emit_get = [&](uint32_t row, uint32_t slot) -> float {
        return emit[row * 64u + slot];
    };
    if (OPEN) {
        for (uint32_t i = threadIdx.x; i < sizeof(RlOpenWg) / 4; i += RL_TRACE_BLOCK) ((RlLdsU32*)wgp)[i] = 0;
        __syncthreads();
    }
    // counts the paths whose results were issued an iteration ago
    auto settle = [&]() {
        if (!pend_any && emit_pend == 0) return;
        // The results these counts stand for (write-through photon stores, memory-side float adds) must be acknowledged
        // before the count can reach the host.  A workgroup-scope release fence compiles to a wait on the LDS counter only
        // on this target (one variant reported paths before their stores had landed), so the wait on the
        // vector-memory counter is spelled out; tests/test_kernel_resources.py looks for it in every OPEN variant.
        asm volatile("s_waitcnt vmcnt(0) ; rl_settle: results acknowledged" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (pend_me) __hip_atomic_fetch_add(wg_fin + my_job, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane < emit_pend)
            __hip_atomic_fetch_add(wg_fin + (((RlLdsU32*)ws->ring_b)[lane] >> 24), 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        pend_me = false;
        pend_any = false;
        emit_pend = 0;
    };
    // reports the workgroup's counts to RlOpenDev if nobody did in the last `gap` ticks; whoever completes a job tells the host
    auto flush = [&](uint32_t gap) {
        const uint32_t now = (uint32_t)wall_clock64();
        uint32_t mine = gap == 0u ? 1u : 0u; // a wave that leaves reports whatever there is
        if (lane == 0 && gap != 0u) {
            uint32_t at = *wg_flushed_at;
            if (now - at >= gap)
                mine = __hip_atomic_compare_exchange_strong(wg_flushed_at, &at, now | 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
        }
        if (__builtin_amdgcn_readfirstlane(mine) == 0u) return;
        for (uint32_t base_job = 0; base_job < known_local; base_job += 64u) {
            const uint32_t jb = base_job + lane;
            uint32_t f = 0, sg = 0;
            if (jb < known_local) { // fin before seg: every path counted in f has its segments in sg or in an earlier report
                f = __hip_atomic_exchange(wg_fin + jb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                sg = __hip_atomic_exchange(wg_seg + jb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (__builtin_amdgcn_ballot_w64((f | sg) != 0u) == 0) continue;
            if (sg != 0) __hip_atomic_fetch_add(&od->seg[jb], sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the segment counts are in before the path counts
            if (f != 0) {
                const uint32_t before = __hip_atomic_fetch_add(&od->fin[jb], f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (before + f == (uint32_t)od->jobs[jb].end) {
                    __hip_atomic_fetch_add(&od->completed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t total = __hip_atomic_load(&od->seg[jb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&ctl->segs[jb], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(&ctl->done[jb], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    };
    uint32_t e_head = 0, e_tail = 0; // wave-uniform ring indices of the emitter queue
    uint32_t emit_age = 0; // wave-uniform: iterations since the queue last ran (open un-fused launches)
#ifdef RL_STATS
    unsigned long long st[RL_ST_COUNT];
    for (int k = 0; k < RL_ST_COUNT; ++k) st[k] = 0;
#endif
    RL_T0(t_total);

    // Evaluates and splats `count` queued paths, one per lane: EmissiveMaterial::get_intensity
    // (material.rs:101-105), cie1931::get_tristimulus and PlotUnit::plot_pixel (plot_unit.rs:56-84).
    auto process_emitted = [&](uint32_t count) {
        RL_STAT(RL_ST_EMIT_BATCHES, 1);
        RL_STAT(RL_ST_EMIT_LANES, count);
        if (OPEN && emit_pend != 0) settle(); // (a second batch in one iteration -- 64 paths ending at once: count the first before its tags go)
        rl_wave_sync();
        if (!FUSED) {
            // Un-fused: the batch's paths ended on a light; their records are written HERE, with the emitter term (f64 Planck)
            // evaluated for the whole batch at once instead of in the iteration the path ended in, for the two lanes it ended
            // in (round 6: ~120 instructions per iteration at 3 % of the lanes).  Queue rows: 0 = the photon's index in its
            // unit, 3 = the path's intensity, 4 = the emitter object (| call << 24 in open launches).  x, y and the wavelength are
            // the path's first draws again (trace_unit.rs:152-158 as rl_begin_path has it): same words, same conversions.
            uint32_t idx = 0, tagged = 0;
            float value = 0.0f;
            RlMappedPhoton* dst_base = photons;
            if (lane < count) {
                const uint32_t slot = (e_head + lane) & 63u;
                idx = rl_f2u(emit_get(0, slot));
                tagged = rl_f2u(emit_get(4, slot));
                if (OPEN) ((RlLdsU32*)ws->ring_b)[lane] = tagged; // (settle() counts these paths per call: see the fused branch)
                uint64_t first = job.first_path;
                if (OPEN) {
                    const RlJobEntry e = jobs[tagged >> 24];
                    first = e.first_path, dst_base = (RlMappedPhoton*)e.target;
                }
                const RlRngBlock b0 = rl_rng_block(job.seed, job.stream, first + idx, 0);
                RlMappedPhoton ph;
                ph.wavelength = rl_get_wavelength(b0.w[0]);
                ph.x = rl_get_bi_unit(b0.w[1]);
                ph.y = rl_get_bi_unit(b0.w[2]) / job.aspect_ratio;
                value = rl_emission(sv, emit_get(3, slot), ph.wavelength, OPEN ? (tagged & 0xffffffu) : tagged);
                ph.probability = value;
                if (OPEN) { // write-through: the reader is a kernel launched while this one still runs
                    uint32_t* dst = (uint32_t*)&dst_base[idx];
                    __hip_atomic_store(dst + 0, rl_f2u(ph.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 1, rl_f2u(ph.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 2, rl_f2u(ph.probability), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 3, rl_f2u(ph.wavelength), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    dst_base[idx] = ph;
                }
            }
        } else if (lane < count) {
            const uint32_t slot = (e_head + lane) & 63u;
            const float sx = emit_get(0, slot), sy = emit_get(1, slot), wavelength = emit_get(2, slot);
            const uint32_t tagged = rl_f2u(emit_get(4, slot));
            // (OPEN: settle() counts these paths per call an iteration from now, when the queue's slots may hold newer entries:
            // the tags wait in ring B, which is empty between two scans)
            if (OPEN) ((RlLdsU32*)ws->ring_b)[lane] = tagged;
            float* target = plot;
            if (OPEN) target = (float*)jobs[tagged >> 24].target;
            const float value = rl_emission(sv, emit_get(3, slot), wavelength, OPEN ? (tagged & 0xffffffu) : tagged);
            if (value != 0.0f) { // adding +0 is the identity
                const RlF3 cie = rl_mul(rl_tristimulus(sv.cie, wavelength), value);
                uint32_t width = job.width, height = job.height; // opaque: `width - 1` etc. are re-derived here instead of living in
                asm volatile("" : "+s"(width), "+s"(height));    // scalar registers across the persistent loop (and spilling)
                const RlSplat sp = rl_splat_weights(width, height, job.wm1, job.hm1, job.aspect_ratio, sx, sy);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float* px = target + 3ull * sp.idx[k];
                    unsafeAtomicAdd(px + 0, cie.x * sp.w[k]);
                    unsafeAtomicAdd(px + 1, cie.y * sp.w[k]);
                    unsafeAtomicAdd(px + 2, cie.z * sp.w[k]);
                }
            }
        }
        if (OPEN) emit_pend = count; // these paths are finished once the adds above are acknowledged: settle() counts them
        rl_wave_sync();
    };

    for (;;) {
        if (OPEN) {
            settle();
            flush(3000u);
        }
#if defined(RL_EXP_EXTRA)
        // sensitivity probe (timing only, tools/ab3.sh): what does one more instruction of a kind cost the kernel?
        {
            uint32_t xs = lane, xv = lane;
#if RL_EXP_EXTRA == 1 // 256 scalar instructions
            uint32_t ss = 1;
#pragma unroll
            for (int k = 0; k < 256; ++k) asm volatile("s_add_u32 %0, %0, 3" : "+s"(ss));
            xs = ss;
#elif RL_EXP_EXTRA == 2 // 256 vector instructions (four chains)
            uint32_t a = lane, b = lane + 1, c = lane + 2, d = lane + 3;
#pragma unroll
            for (int k = 0; k < 64; ++k) asm volatile("v_add_u32 %0, 3, %0\n\tv_add_u32 %1, 5, %1\n\tv_add_u32 %2, 7, %2\n\tv_add_u32 %3, 9, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            xv = a ^ b ^ c ^ d;
#elif RL_EXP_EXTRA == 3 // 16 exposed LDS round trips
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0xfc, %0" : "+v"(xv) : : "memory");
#elif RL_EXP_EXTRA == 4 // 256 s_nop 0
#pragma unroll
            for (int k = 0; k < 256; ++k) asm volatile("s_nop 0");
#elif RL_EXP_EXTRA == 5 // 256 vector instructions in ONE dependent chain
#pragma unroll
            for (int k = 0; k < 256; ++k) asm volatile("v_add_u32 %0, 3, %0" : "+v"(xv));
#elif RL_EXP_EXTRA == 8 // 64 taken branches
#pragma unroll
            for (int k = 0; k < 64; ++k) asm volatile("s_branch 0");
#elif RL_EXP_EXTRA == 9 // 64 conditional branches that are not taken
            asm volatile("s_cmp_eq_u32 0, 1");
#pragma unroll
            for (int k = 0; k < 64; ++k) asm volatile("s_cbranch_scc1 0");
#elif RL_EXP_EXTRA == 6 // 128 packed FMAs (four chains)
            {
                typedef float F2 __attribute__((ext_vector_type(2)));
                F2 pa = {(float)lane, 1.0f}, pb = {2.0f, (float)lane}, pc = {3.0f, 1.5f}, pd = {0.5f, 0.25f};
#pragma unroll
                for (int k = 0; k < 32; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_fma_f32 %1, %1, %2, %3\n\tv_pk_fma_f32 %2, %2, %3, %0\n\tv_pk_fma_f32 %3, %3, %0, %1" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd));
                xv = rl_f2u(pa.x + pb.y + pc.x + pd.y);
            }
#elif RL_EXP_EXTRA == 7 // 256 scalar-operand-free FMAs (four chains)
            {
                float fa = (float)lane, fb = 2.0f, fc = 3.0f, fd = 0.5f;
#pragma unroll
                for (int k = 0; k < 64; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %0\n\tv_fma_f32 %3, %3, %0, %1" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));
                xv = rl_f2u(fa + fb + fc + fd);
            }
#elif RL_EXP_EXTRA == 10 // 4 batches of nine ds_bpermute_b32 with one wait each (how rounds 1-4 fetched a pair's ray)
            {
                float t0 = (float)lane;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t addr = ((lane * 7u + 3u) & 63u) << 2;
                    float r0, r1, r2, r3, r4, r5, r6, r7, r8;
                    asm volatile("ds_bpermute_b32 %0, %9, %10\n\tds_bpermute_b32 %1, %9, %10\n\tds_bpermute_b32 %2, %9, %10\n\t"
                                 "ds_bpermute_b32 %3, %9, %10\n\tds_bpermute_b32 %4, %9, %10\n\tds_bpermute_b32 %5, %9, %10\n\t"
                                 "ds_bpermute_b32 %6, %9, %10\n\tds_bpermute_b32 %7, %9, %10\n\tds_bpermute_b32 %8, %9, %10\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(r8)
                                 : "v"(addr), "v"(t0) : "memory");
                    t0 = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + r8;
                }
                xv = rl_f2u(t0);
            }
#elif RL_EXP_EXTRA == 12 // 4 batches of two 16-byte gathers + one 4-byte gather from the wave's scratch, one wait each (round 5's fetch)
            {
                float t0 = (float)lane;
                uint32_t who = (lane * 7u + 3u) & 63u;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    RlCullRay r;
                    float f;
                    rl_fetch_cull_ray(ws, who, r, f);
                    t0 += r.d.x + r.d.y + r.d.z + r.p + r.m.x + r.m.y + r.m.z + r.q + f;
                    who = (who * 5u + (rl_f2u(t0) & 1u)) & 63u;
                    asm volatile("" ::: "memory");
                }
                xv = rl_f2u(t0);
            }
#elif RL_EXP_EXTRA == 11 // 64 s_and_saveexec / s_or exec pairs
#pragma unroll
            for (int k = 0; k < 64; ++k) asm volatile("s_and_saveexec_b64 s[2:3], exec\n\ts_or_b64 exec, exec, s[2:3]" ::: "s2", "s3");
#endif
            if (xs == 0xdeadbeefu || xv == 0xdeadbeefu) segments += 1; // keep the probe alive
        }
#endif
        // ---- hand new paths to the lanes whose path ended (trace_unit.rs:152-167) ----
        RL_T0(t_refill);
        // Two steps, written straight-line (at most: refill, hand out, and -- when more lanes asked than the stash held -- once
        // more): as a loop the compiler kept the path's sixteen registers in loop-carried copies and moved all of them at the end of
        // every hand-out (16 v_mov per iteration); a lane that still has no path after the second hand-out -- the tail of a launch --
        // asks again in the next iteration.
        // refill(): all 64 lanes generate one camera ray each (full exec mask) into the stash; false if there are no paths to take.
        auto refill = [&]() -> bool {
            if (drained) return false;
                // Refill: all 64 lanes generate one camera ray each (full exec mask) into the stash.
                uint32_t open_n = 0; // OPEN: size of the job the refill comes from
                if (OPEN) {
                    bool got = false;
                    for (;;) {
                        if (chunk_next < chunk_end) { // the rest of the slice this wave took last time
                            open_n = (uint32_t)jobs[stash_job].end;
                            got = true;
                            break;
                        }
                        if (stash_job < known_local) { // take paths of the oldest job that has any left
                            open_n = (uint32_t)jobs[stash_job].end;
                            // One same-address atomic per 64 paths is about all the memory system does at the chip's path
                            // rate (the waits queue up behind one another and stall the waves): with more calls waiting
                            // behind this one a wave takes four refills at a time; a call on its own is spread thin so
                            // that every wave gets some of it.
                            const uint32_t take = known_local - stash_job >= 2u ? 256u : 64u;
                            uint32_t b = 0;
                            if (lane == 0) b = __hip_atomic_fetch_add(&od->next[stash_job], take, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            b = __builtin_amdgcn_readfirstlane(b);
                            if (b < open_n) {
                                chunk_end = b + take < open_n ? b + take : ((open_n + 63u) & ~63u);
                                chunk_next = b;
                                got = true;
                                break;
                            }
                            stash_job += 1u;
                            continue;
                        }
                        // Out of known work.  Thousands of waves get here at about the same time and look again until
                        // there is more: the device-wide words are read at most once per workgroup and 10 us (the
                        // workgroup's last reading and its time are kept in LDS), or every such read queues up behind
                        // the others' and the waves that are still tracing slow down several-fold.
                        const uint32_t now = (uint32_t)wall_clock64();
                        const uint32_t seen = __builtin_amdgcn_readfirstlane(wg_poll[0]), seen_at = __builtin_amdgcn_readfirstlane(wg_poll[1]);
                        if (seen > known_local) {
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                            known_local = seen;
                            continue;
                        }
                        if (now - seen_at < 1000u) break;
                        if (lane == 0) wg_poll[1] = now;
                        uint32_t v = 0;
                        if (lane == 0) v = __hip_atomic_load(&od->known, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v = __builtin_amdgcn_readfirstlane(v);
                        if (v > known_local) {
                            if (lane == 0) wg_poll[0] = v;
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                            known_local = v;
                            continue;
                        }
                        if (lane == 0) v = __hip_atomic_load(&od->closed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (__builtin_amdgcn_readfirstlane(v) != 0u) {
                            drained = true;
                            break;
                        }
                        // Every known job is handed out: one wave at a time asks the host for more, or closes the launch.
                        if (lane == 0) {
                            uint32_t expected = 0;
                            v = __hip_atomic_compare_exchange_strong(&od->closing, &expected, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                                                     __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
                        }
                        if (__builtin_amdgcn_readfirstlane(v) == 0u) break; // somebody else is at it: look again next iteration
                        uint32_t pub = 0;
                        if (lane == 0) pub = __hip_atomic_load(&ctl->published, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                        pub = __builtin_amdgcn_readfirstlane(pub);
                        if (pub > RL_OPEN_CAP) pub = RL_OPEN_CAP;
                        if (pub > known_local) { // new calls: copy their entries to device memory, then make them known
                            for (uint32_t k = known_local + lane; k < pub; k += 64u) {
                                unsigned long long* dst = (unsigned long long*)&od->jobs[k];
                                const unsigned long long* src = (const unsigned long long*)&ctl->jobs[k];
                                for (int w = 0; w < 4; ++w)
                                    __hip_atomic_store(dst + w, __hip_atomic_load(src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM),
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                            if (lane == 0) {
                                od->idle_since = 0;
                                od->stuck_since = 0;
                                wg_poll[0] = pub;
                                __hip_atomic_store(&od->known, pub, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(&od->closing, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            known_local = pub;
                            continue;
                        }
                        // Nothing new.  While calls of this launch are still finishing their last paths the kernel is
                        // running anyway, and their callers come back with the next batch the moment they are told: stay
                        // open until every call is complete and then for a grace period (job.grace_ticks).
                        uint32_t stay = 0;
                        if (lane == 0 && known_local < RL_OPEN_CAP) {
                            const bool finishing = __hip_atomic_load(&od->completed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < known_local;
                            const unsigned long long now = wall_clock64(), since = finishing ? od->stuck_since : od->idle_since;
                            // (a call that never completes would be a bug in the counting above: after 5 s without one
                            // the launch ends regardless, and the host reports the call that is missing)
                            uint32_t grace = job.grace_ticks;
                            asm volatile("" : "+s"(grace));
                            const unsigned long long limit = finishing ? 500000000ull : (unsigned long long)grace;
                            if (since == 0) (finishing ? od->stuck_since : od->idle_since) = now, stay = 1;
                            else if (now - since < limit) stay = 1;
                            if (!finishing) od->stuck_since = 0;
                            if (stay) __hip_atomic_store(&od->closing, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        if (__builtin_amdgcn_readfirstlane(stay) != 0u) break;
                        uint32_t again = 0;
                        if (lane == 0) {
                            __hip_atomic_store(&ctl->closed_at, known_local, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_SYSTEM);
                            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");
                            again = __hip_atomic_load(&ctl->published, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                        again = __builtin_amdgcn_readfirstlane(again);
                        if (again == known_local || known_local >= RL_OPEN_CAP) { // nothing arrived: the launch is closed for good
                            if (lane == 0) {
                                __hip_atomic_store(&ctl->final_at, known_local, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_SYSTEM);
                                __hip_atomic_store(&od->closed, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            drained = true;
                            break;
                        }
                        if (lane == 0) { // a call arrived in between: re-open and pick it up in the next round
                            __hip_atomic_store(&ctl->closed_at, RL_OPEN_NONE, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_SYSTEM);
                            __hip_atomic_store(&od->closing, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    if (!got) return false; // no paths to hand out right now (or ever again, if drained)
                }
                if (!OPEN && chunk_next == chunk_end) {
                    // Small launches (fewer than 16 paths per lane of the grid: the reference's 524,288-path
                    // batch is 2 per lane) take one stash refill at a time, so that every wave gets work;
                    // with RL_CHUNK the first half of the waves would take everything.
                    const unsigned long long chunk =
                        job.n_paths >= (unsigned long long)gridDim.x * (RL_TRACE_BLOCK * 16ull) ? RL_CHUNK : 64ull;
                    unsigned long long b = 0;
                    if (lane == 0) b = atomicAdd(&queue[0], chunk);
                    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
                    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
                    chunk_next = ((uint64_t)hi << 32) | lo;
                    chunk_end = chunk_next + chunk;
                }
                RL_STAT(RL_ST_REFILLS, 1);
                if (OPEN) stash_first = jobs[stash_job].first_path;
                const uint64_t limit = OPEN ? (uint64_t)open_n : job.n_paths;
                stash_path0 = stash_first + chunk_next;
                stash_valid = chunk_next >= limit ? 0u : (limit - chunk_next >= 64ull ? 64u : (uint32_t)(limit - chunk_next));
                chunk_next += 64;
                const bool valid = lane < stash_valid;
                const uint64_t path_index = stash_path0 + lane;
                RlPath fresh = p;
                RL_T0(t_cam);
                if (valid) rl_begin_path(sv, job.aspect_ratio, job.seed, job.stream, path_index, &fresh);
                RL_T1(RL_ST_T_CAMERA, t_cam);
                rl_wave_sync();
                stash[0 * 64 + lane] = fresh.origin.x;
                stash[1 * 64 + lane] = fresh.origin.y;
                stash[2 * 64 + lane] = fresh.origin.z;
                stash[3 * 64 + lane] = fresh.direction.x;
                stash[4 * 64 + lane] = fresh.direction.y;
                stash[5 * 64 + lane] = fresh.direction.z;
                stash[6 * 64 + lane] = fresh.wavelength;
                stash[7 * 64 + lane] = fresh.sx;
                stash[8 * 64 + lane] = fresh.sy;
                stash[9 * 64 + lane] = fresh.ior;
                rl_wave_sync();
                stash_head = 0;
                stash_count = 64;
                if (!OPEN && chunk_next >= job.n_paths) drained = true;
            return true;
        };
        // hand_out(need): the lanes of `need` (no path) take the stash's next slots; true if every one of them was served a slot.
        auto hand_out = [&](const uint64_t need) -> bool {
            const uint32_t avail = stash_count - stash_head;
            const uint32_t rank = rl_mbcnt(need);
            const uint32_t slot = stash_head + rank;
            // One condition, combined without branches, and selects instead of assignments under it: with the sixteen assignments
            // inside (nested) ifs the compiler copied the whole path state at every join -- 64 v_mov per iteration.  Every lane
            // reads a slot (the lanes that take nothing: any one), the takers keep what they read.
            const bool take = !active & (rank < avail) & (slot < stash_valid);
            {
                const uint32_t sl = slot & 63u;
                const float s0 = stash[0 * 64 + sl], s1 = stash[1 * 64 + sl], s2 = stash[2 * 64 + sl], s3 = stash[3 * 64 + sl], s4 = stash[4 * 64 + sl];
                const float s5 = stash[5 * 64 + sl], s6 = stash[6 * 64 + sl], s7 = stash[7 * 64 + sl], s8 = stash[8 * 64 + sl], s9 = stash[9 * 64 + sl];
                my_path = take ? stash_path0 + slot : my_path;
                if (OPEN) my_job = take ? stash_job : my_job;
                p.origin = rl_f3(take ? s0 : p.origin.x, take ? s1 : p.origin.y, take ? s2 : p.origin.z);
                p.direction = rl_f3(take ? s3 : p.direction.x, take ? s4 : p.direction.y, take ? s5 : p.direction.z);
                p.wavelength = take ? s6 : p.wavelength;
                p.sx = take ? s7 : p.sx;
                p.sy = take ? s8 : p.sy;
                p.ior = take ? s9 : p.ior;
                p.intensity = take ? 1.0f : p.intensity;
                p.continue_chance = take ? 1.0f : p.continue_chance;
                p.bounce = take ? 0u : p.bounce;
                active = active | take;
            }
            const uint32_t wanted = (uint32_t)__popcll(need);
            stash_head += wanted < avail ? wanted : avail;
            return wanted <= avail;
        };
        {
            const uint64_t need = __builtin_amdgcn_ballot_w64(!active);
            if (need != 0) { // (one call site for refill(): it holds the camera's code)
                bool served = false;
                if (RL_LIKELY(stash_count != stash_head)) served = hand_out(need);
                if (RL_UNLIKELY(!served)) {
                    if (refill()) hand_out(__builtin_amdgcn_ballot_w64(!active));
                }
            }
        }
        RL_T1(RL_ST_T_REFILL, t_refill);
        if (__builtin_amdgcn_ballot_w64(active) == 0) {
            if (!OPEN || drained) break;
            // an open launch with nothing to hand out at the moment: report what is finished (the host may be waiting
            // for exactly that before it sends more) and look again
            settle();
            if (e_tail != e_head) {
                process_emitted(e_tail - e_head);
                e_head = e_tail;
                settle();
            }
            flush(500u);
            for (int k = 0; k < 4; ++k) __builtin_amdgcn_s_sleep(127); // ~15 us: thousands of waves poll the same few words
            continue;
        }
        // (ring-S rounds unrolled wherever the cull table is in LDS -- except in the fused open launches of a tables-only scene, the
        // instantiation with both LDS and 64-bit global addresses to hold: unrolled it spills two vector registers to scratch)
        // (... and their children's bounds requested ahead of the fetch wherever that leaves the instantiation spill-free: not in the
        // open launches of a tables-only scene, 64-bit global addresses again)
        // (round 6: the Compound tree's lean form freed ~15 registers, and the open launches of a scene that is staged whole now take
        // every one of these options inside their 120 registers; the tables-only and global-fetch open variants still do not)
        constexpr bool FULL = !OPEN || (RL_OPEN_FULL && STAGE == RL_STAGE_ALL && (FUSED || !CYL)); // (the un-fused open variant of a scene with prism cylinders spills one scalar register with them)
        const RlHit hit = rl_scan_wave<CYL, FULL && RL_LEAN_SPLIT, STAGE != RL_STAGE_NONE && (!(FUSED && OPEN) || FULL), STAGE != RL_STAGE_NONE && (!(OPEN && (FUSED || STAGE == RL_STAGE_TABLES)) || FULL) && RL_W_S && RL_LEAN_HOIST, STAGE == RL_STAGE_ALL, STAGE != RL_STAGE_NONE, STAGE != RL_STAGE_ALL>(sv, base + (lay.off_cull - tab0), CYL ? base + (lay.off_prism_cyl - tab0) : nullptr, lay.group_gc, lay.small_ordered, lay.cull_cmax2, lay.n_cluster_groups, lay.n_prism_groups,
                                       lay.n_cluster_supers, lay.super_g, ring_t, p.origin, p.direction, active ? 0u : 0x80000000u, ws, lane RL_TACC_ARG);
#ifdef RL_STATS
        {
            const uint32_t mk = (active && hit.obj != RL_HIT_NONE) ? rl_object_material(rl_f2u(sv.objects[hit.obj].w)) : 99u;
            const uint64_t m_act = __builtin_amdgcn_ballot_w64(active);
            const uint64_t m_void = __builtin_amdgcn_ballot_w64(active && hit.obj == RL_HIT_NONE);
            const uint64_t m_emit = __builtin_amdgcn_ballot_w64(mk == RL_MATERIAL_BLACK_BODY);
            const uint64_t m_grey = __builtin_amdgcn_ballot_w64(mk == RL_MATERIAL_DIFFUSE_GREY);
            const uint64_t m_col = __builtin_amdgcn_ballot_w64(mk == RL_MATERIAL_DIFFUSE_COLOURED);
            const uint64_t m_gls = __builtin_amdgcn_ballot_w64(mk == RL_MATERIAL_GLOSSY_MIRROR);
            const uint64_t m_glass = __builtin_amdgcn_ballot_w64(mk == RL_MATERIAL_SF10_GLASS);
            const uint64_t m_soap = __builtin_amdgcn_ballot_w64(mk == RL_MATERIAL_SOAP_BUBBLE);
            RL_STAT(RL_ST_ITER, 1);
            RL_STAT(RL_ST_SCAN_LANES, __popcll(m_act));
            RL_STAT(RL_ST_END_VOID, __popcll(m_void));
            RL_STAT(RL_ST_END_EMITTER, __popcll(m_emit));
            RL_STAT(RL_ST_SHADE_DIFFUSE, __popcll(m_grey | m_col | m_gls));
            RL_STAT(RL_ST_SHADE_GLASS, __popcll(m_glass));
            RL_STAT(RL_ST_SHADE_SOAP, __popcll(m_soap));
            RL_STAT(RL_ST_ANY_DIFFUSE, (m_grey | m_col | m_gls) != 0);
            RL_STAT(RL_ST_ANY_GLASS, m_glass != 0);
            RL_STAT(RL_ST_ANY_SOAP, m_soap != 0);
            RL_STAT(RL_ST_ANY_COLOURED, m_col != 0);
            RL_STAT(RL_ST_ANY_GLOSSY, m_gls != 0);
        }
#endif
        RL_T0(t_shade);
        // (declared per iteration: every one of these is set and consumed between here and the queue push below; at function scope
        // they were loop-carried values the compiler kept in registers across the scan)
        bool ended_on_emitter = false, ended_now = false;
        uint32_t emit_obj = 0, emit_idx = 0;
        if (active) {
            segments += 1;
            float value;
            uint32_t emitter = 0;
            const int status = rl_bounce(sv, job.seed, job.stream, my_path, &p, hit, &value, &emitter);
            if (status != RL_PATH_CONTINUES) {
                active = false;
                p.direction = rl_f3(0.0f, 0.0f, 0.0f);
                paths_done += 1;
                ended_on_emitter = status == RL_PATH_ENDED_ON_EMITTER; // only these can contribute (trace_unit.rs:94-101,131)
                emit_obj = OPEN ? (emitter | (my_job << 24)) : emitter; // open launches: the call, i.e. the target
                if (!FUSED && !ended_on_emitter) { // The Void, roulette: the record (probability 0) is written here
                    RlMappedPhoton ph;
                    ph.x = p.sx;
                    ph.y = p.sy;
                    ph.probability = value;
                    ph.wavelength = p.wavelength;
                    if (OPEN) { // write-through (see `report`): the reader is a kernel launched while this one still runs
                        const RlJobEntry e = jobs[my_job];
                        uint32_t* dst = (uint32_t*)&((RlMappedPhoton*)e.target)[my_path - e.first_path];
                        __hip_atomic_store(dst + 0, rl_f2u(ph.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 1, rl_f2u(ph.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 2, rl_f2u(ph.probability), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 3, rl_f2u(ph.wavelength), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        photons[my_path - job.first_path] = ph;
                    }
                }
                if (!FUSED && ended_on_emitter) emit_idx = (uint32_t)(my_path - (OPEN ? jobs[my_job].first_path : job.first_path));
                if (OPEN) { // trace_unit.rs:92-131: the Void and a light end the path in the scan's iteration, roulette after the bounce
                    ended_now = true;
                    __hip_atomic_fetch_add(wg_seg + my_job, p.bounce + ((hit.obj == RL_HIT_NONE || status == RL_PATH_ENDED_ON_EMITTER) ? 1u : 0u),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (OPEN) {
            // a path that ended on a light is finished when it is splatted / its record is written (process_emitted), every other one now
            pend_me = ended_now && !ended_on_emitter;
            pend_any = __builtin_amdgcn_ballot_w64(pend_me) != 0;
            ended_now = false;
        }
        RL_T1(RL_ST_T_SHADE, t_shade);
        RL_T0(t_emit);
        {
            // ---- fused splat (plot_unit.rs:56-95) / un-fused record of a contributing path, deferred: queue the paths that ended on a light ----
            const uint64_t m = __builtin_amdgcn_ballot_w64(ended_on_emitter);
            if (m != 0) {
                // The queue holds 64 paths.  A batch runs when it is exactly full -- or, before these paths would overflow it,
                // with the (nearly 64) it has: ~2 paths end on a light per iteration, so batches are ~97 % full.
                const uint32_t n_new = (uint32_t)__popcll(m);
                if (RL_UNLIKELY(e_tail - e_head + n_new > 64u)) {
                    process_emitted(e_tail - e_head);
                    e_head = e_tail;
                }
                if (ended_on_emitter) {
                    const uint32_t slot = rl_mbcnt_from(m, e_tail) & 63u;
                    if (FUSED) {
                        emit_put(0, slot, p.sx);
                        emit_put(1, slot, p.sy);
                        emit_put(2, slot, p.wavelength);
                    } else {
                        emit_put(0, slot, rl_u2f(emit_idx));
                    }
                    emit_put(3, slot, p.intensity);
                    emit_put(4, slot, rl_u2f(emit_obj));
                    ended_on_emitter = false;
                }
                e_tail += n_new;
                if (RL_UNLIKELY(e_tail - e_head == 64u)) {
                    process_emitted(64u);
                    e_head += 64u;
                }
            }
            // Open un-fused launches: a call is complete -- and its caller, who may be BLOCKED in TraceUnit::render, released -- only
            // when its last queued record is written, and a queue fills in ~30 iterations (half a millisecond); a batch that has
            // waited RL_EMIT_MAX_AGE iterations runs with what it has.
            if (OPEN && !FUSED) {
                emit_age = e_tail != e_head ? emit_age + 1u : 0u;
                if (RL_UNLIKELY(emit_age >= RL_EMIT_MAX_AGE)) {
                    process_emitted(e_tail - e_head);
                    e_head = e_tail;
                    emit_age = 0;
                }
            }
        }
        RL_T1(RL_ST_T_EMIT, t_emit);
    }
    if (e_tail != e_head) process_emitted(e_tail - e_head);
    if (OPEN) {
        settle();
        flush(0u);
    }
    RL_T1(RL_ST_T_TOTAL, t_total);
#ifdef RL_STATS
    if (lane == 0)
        for (int k = 0; k < RL_ST_COUNT; ++k) atomicAdd(&rl_stat_counters[k], st[k]);
#endif
    // One atomic per wave for the counters.
    uint32_t s = segments, d = paths_done;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off);
        d += __shfl_down(d, off);
    }
    if (lane == 0) {
        atomicAdd(&queue[1], (unsigned long long)s);
        atomicAdd(&queue[2], (unsigned long long)d);
    }
}

// The two entry points.  Four waves per SIMD either way (one workgroup of 16 waves per CU).
//   * plain launches -- every bulk launch, the bench -- may use all 128 vector registers a wave can have at that occupancy;
//   * OPEN launches stay resident while the host appends calls to them, so they keep to 120: four waves per SIMD then leave 32 of
//     the 512 registers per lane, which is what lets the small kernels of the other units (plot, gather, tonemap, clears) run
//     BESIDE a resident trace kernel instead of behind it.
// (amdgpu_num_vgpr counts in units of two registers on this target.)
template <int STAGE, bool FUSED, bool CYL>
__global__ __launch_bounds__(RL_TRACE_BLOCK, RL_TRACE_WPS) __attribute__((amdgpu_num_vgpr(RL_TRACE_VGPRS / 2))) void rl_trace_kernel(const RlF4* __restrict__ scene, RlSceneLayout lay, RlTraceJob job,
                                                                                                       RlMappedPhoton* __restrict__ photons, float* __restrict__ plot,
                                                                                                       unsigned long long* __restrict__ queue, const RlJobEntry* jobs,
                                                                                                       RlOpenDev* od, RlOpenCtl* ctl) {
    rl_trace_body<STAGE, FUSED, false, CYL>(scene, lay, job, photons, plot, queue, jobs, od, ctl);
}
template <int STAGE, bool FUSED, bool CYL>
__global__ __launch_bounds__(RL_TRACE_BLOCK, RL_TRACE_WPS) __attribute__((amdgpu_num_vgpr(RL_TRACE_VGPRS_OPEN / 2))) void rl_trace_kernel_open(const RlF4* __restrict__ scene, RlSceneLayout lay, RlTraceJob job,
                                                                                                            RlMappedPhoton* __restrict__ photons, float* __restrict__ plot,
                                                                                                            unsigned long long* __restrict__ queue, const RlJobEntry* jobs,
                                                                                                            RlOpenDev* od, RlOpenCtl* ctl) {
    rl_trace_body<STAGE, FUSED, true, CYL>(scene, lay, job, photons, plot, queue, jobs, od, ctl);
}

// PlotUnit::plot (plot_unit.rs:87-95) for the un-fused path: one photon per thread.
__global__ __launch_bounds__(RL_BLOCK) void rl_plot_kernel(const RlMappedPhoton* __restrict__ photons, uint32_t n,
                                                           const RlF4* __restrict__ cie, uint32_t width, uint32_t height,
                                                           float aspect_ratio, float* __restrict__ plot) {
    __shared__ RlF4 s_cie[RL_CIE_SAMPLES];
    for (uint32_t i = threadIdx.x; i < RL_CIE_SAMPLES; i += RL_BLOCK) s_cie[i] = cie[i];
    __syncthreads();
    for (uint32_t i = blockIdx.x * RL_BLOCK + threadIdx.x; i < n; i += gridDim.x * RL_BLOCK) {
        const RlMappedPhoton ph = photons[i];
        if (ph.probability == 0.0f) continue;
        const RlF3 c = rl_mul(rl_tristimulus(s_cie, ph.wavelength), ph.probability);
        const RlSplat s = rl_splat_weights(width, height, aspect_ratio, ph.x, ph.y);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float* px = plot + 3ull * s.idx[k];
            unsafeAtomicAdd(px + 0, c.x * s.w[k]);
            unsafeAtomicAdd(px + 1, c.y * s.w[k]);
            unsafeAtomicAdd(px + 2, c.z * s.w[k]);
        }
    }
}

// GatherUnit::accumulate + PlotUnit::clear (gather_unit.rs:49-64, plot_unit.rs:98-102), one float
// component per lane-element, 16 bytes per lane per buffer.  Kahan order is fixed; nothing here may
// be re-associated (no fast-math).
__global__ __launch_bounds__(RL_BLOCK) void rl_gather_kernel(float* __restrict__ acc, float* __restrict__ comp,
                                                             float* __restrict__ px, uint64_t n_floats) {
    const uint64_t n4 = n_floats / 4;
    float4* acc4 = (float4*)acc;
    float4* comp4 = (float4*)comp;
    float4* px4 = (float4*)px;
    for (uint64_t i = (uint64_t)blockIdx.x * RL_BLOCK + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * RL_BLOCK) {
        float4 a = acc4[i], c = comp4[i];
        const float4 p = px4[i];
        float av[4] = {a.x, a.y, a.z, a.w}, cv[4] = {c.x, c.y, c.z, c.w};
        const float pv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float extra = pv[k] - cv[k];
            const float sum = av[k] + extra;
            cv[k] = (sum - av[k]) - extra;
            av[k] = sum;
        }
        acc4[i] = make_float4(av[0], av[1], av[2], av[3]);
        comp4[i] = make_float4(cv[0], cv[1], cv[2], cv[3]);
        px4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n_floats & 3)) {
        const uint64_t i = n4 * 4 + threadIdx.x;
        const float extra = px[i] - comp[i];
        const float sum = acc[i] + extra;
        comp[i] = (sum - acc[i]) - extra;
        acc[i] = sum;
        px[i] = 0.0f;
    }
}

// TonemapUnit::find_exposure (tonemap_unit.rs:55-69).  The reference sums Y and Y^2 sequentially in
// f32 over all pixels; a tree reduction would move max_intensity at the 1e-4 level and with it every
// output pixel, so the two sums keep the reference's order: lane 0 of wave 0 accumulates sum(Y) and lane 1
// sum(Y*Y) in pixel order.  Everything that is not the dependent chain is done by the other three waves of the
// workgroup, one tile ahead: they stream the next tile of Y into LDS and square it there (y * y rounds exactly as in
// the reference's `sq + y * y`, contraction is off) while the chains run over the current one, so that the serial
// phase is 16-byte LDS reads issued four ahead of their use and one dependent add per pixel.  Runs once per tonemap
// (every 30 s in the reference).
#define RL_EXPOSURE_TILE 2048
#define RL_EXPOSURE_BLOCK 256
__global__ __launch_bounds__(RL_EXPOSURE_BLOCK) void rl_exposure_kernel(const float* __restrict__ xyz, uint32_t n_pixels,
                                                                        float n_as_float, float* __restrict__ out_max) {
    __shared__ __attribute__((aligned(16))) float tile[2][2][RL_EXPOSURE_TILE]; // [buffer][0 = Y, 1 = Y * Y][pixel of the tile]
    const uint32_t t = threadIdx.x;
    const uint32_t n_tiles = (n_pixels + RL_EXPOSURE_TILE - 1) / RL_EXPOSURE_TILE;
    auto fill = [&](uint32_t k) { // waves 1..3: tile k into buffer k & 1
        const uint32_t start = k * RL_EXPOSURE_TILE;
        const uint32_t count = min((uint32_t)RL_EXPOSURE_TILE, n_pixels - start);
        float* y_out = tile[k & 1u][0];
        float* yy_out = tile[k & 1u][1];
#pragma unroll 8
        for (uint32_t i = t - 64u; i < count; i += RL_EXPOSURE_BLOCK - 64u) {
            const float y = xyz[3ull * (start + i) + 1];
            y_out[i] = y;
            yy_out[i] = y * y;
        }
    };
    if (t >= 64u && n_tiles != 0) fill(0);
    __syncthreads();
    float total = 0.0f;
    for (uint32_t k = 0; k < n_tiles; ++k) {
        if (t >= 64u) {
            if (k + 1 < n_tiles) fill(k + 1);
        } else if (t < 2u) {
            const uint32_t count = min((uint32_t)RL_EXPOSURE_TILE, n_pixels - k * RL_EXPOSURE_TILE);
            const float* mine = tile[k & 1u][t];
            // Four 16-byte LDS reads stay in flight ahead of the 16 dependent adds that consume the previous four.  The
            // reads are inline assembly because the build's register-minimising scheduler otherwise sinks every read to
            // just before its use (one LDS round trip per four pixels: 22 ms at 1080p instead of 10); the s_waitcnt
            // statements take `total` as an operand so that the adds cannot move across them.
            typedef float RlV4 __attribute__((ext_vector_type(4)));
            uint32_t addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)mine;
            const uint32_t n16 = count / 16u;
            RlV4 a0, a1, a2, a3, b0, b1, b2, b3; // two register sets used in turn: nothing is ever copied while a read is in flight
#define RL_EXP_READ(R0, R1, R2, R3)                                                                                         \
    asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:16\n\tds_read_b128 %2, %5 offset:32\n\tds_read_b128 %3, %5 offset:48" \
                 : "=&v"(R0), "=&v"(R1), "=&v"(R2), "=&v"(R3), "+v"(total) : "v"(addr) : "memory");                          \
    addr += 64u;
#define RL_EXP_WAIT(R0, R1, R2, R3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R0), "+v"(R1), "+v"(R2), "+v"(R3), "+v"(total));
#define RL_EXP_ADD(R0, R1, R2, R3)                                                                      \
    total = total + R0.x; total = total + R0.y; total = total + R0.z; total = total + R0.w;             \
    total = total + R1.x; total = total + R1.y; total = total + R1.z; total = total + R1.w;             \
    total = total + R2.x; total = total + R2.y; total = total + R2.z; total = total + R2.w;             \
    total = total + R3.x; total = total + R3.y; total = total + R3.z; total = total + R3.w;
            if (n16 != 0) {
                RL_EXP_READ(a0, a1, a2, a3)
                uint32_t q = 1;
                for (; q + 1u < n16; q += 2u) {
                    RL_EXP_WAIT(a0, a1, a2, a3)
                    RL_EXP_READ(b0, b1, b2, b3)
                    RL_EXP_ADD(a0, a1, a2, a3)
                    RL_EXP_WAIT(b0, b1, b2, b3)
                    RL_EXP_READ(a0, a1, a2, a3)
                    RL_EXP_ADD(b0, b1, b2, b3)
                }
                RL_EXP_WAIT(a0, a1, a2, a3)
                if (q < n16) {
                    RL_EXP_READ(b0, b1, b2, b3)
                    RL_EXP_ADD(a0, a1, a2, a3)
                    RL_EXP_WAIT(b0, b1, b2, b3)
                    RL_EXP_ADD(b0, b1, b2, b3)
                } else {
                    RL_EXP_ADD(a0, a1, a2, a3)
                }
            }
#undef RL_EXP_READ
#undef RL_EXP_WAIT
#undef RL_EXP_ADD
            for (uint32_t i = 16u * n16; i < count; ++i) total = total + mine[i];
        }
        __syncthreads();
    }
    __shared__ float sums[2];
    if (t < 2u) sums[t] = total;
    __syncthreads();
    if (t == 0) {
        const float mean = sums[0] / n_as_float;
        const float sqr_mean = sums[1] / n_as_float;
        const float variance = sqr_mean - mean * mean;
        out_max[0] = mean + sqrtf(variance);
    }
}

// srgb.rs:20-26
__device__ __forceinline__ float rl_gamma_correct(float f) {
    if (f <= 0.0031308f) return 12.92f * f;
    return 1.055f * rl_powf_full(f, 1.0f / 2.4f) - 0.055f;
}
__device__ __forceinline__ float rl_clamp01(float x) { // tonemap_unit.rs:34-38
    if (x < 0.0f) return 0.0f;
    if (1.0f < x) return 1.0f;
    return x;
}

// TonemapUnit::tonemap's pixel loop (tonemap_unit.rs:79-99) + srgb::transform (srgb.rs:29-41).
__global__ __launch_bounds__(RL_BLOCK) void rl_tonemap_kernel(const float* __restrict__ xyz, uint32_t n_pixels,
                                                              const float* __restrict__ max_intensity_ptr,
                                                              uint8_t* __restrict__ rgb, float* __restrict__ srgb) {
    const float max_intensity = max_intensity_ptr[0];
    const float ln_4 = rl_logf(4.0f);
    for (uint32_t i = blockIdx.x * RL_BLOCK + threadIdx.x; i < n_pixels; i += gridDim.x * RL_BLOCK) {
        const float cx = rl_logf_full(xyz[3ull * i + 0] / max_intensity + 1.0f) / ln_4;
        const float cy = rl_logf_full(xyz[3ull * i + 1] / max_intensity + 1.0f) / ln_4;
        const float cz = rl_logf_full(xyz[3ull * i + 2] / max_intensity + 1.0f) / ln_4;
        const float r = rl_clamp01(rl_gamma_correct(3.2406f * cx - 1.5372f * cy - 0.4986f * cz));
        const float g = rl_clamp01(rl_gamma_correct(-0.9689f * cx + 1.8758f * cy + 0.0415f * cz));
        const float b = rl_clamp01(rl_gamma_correct(0.0557f * cx - 0.2040f * cy + 1.0570f * cz));
        srgb[3ull * i + 0] = r;
        srgb[3ull * i + 1] = g;
        srgb[3ull * i + 2] = b;
        rgb[3ull * i + 0] = rl_to_u8(r * 255.0f);
        rgb[3ull * i + 1] = rl_to_u8(g * 255.0f);
        rgb[3ull * i + 2] = rl_to_u8(b * 255.0f);
    }
}

// Evaluates rl_math.h on the device so tests can compare it bit-for-bit with the host build.
// fn: 0 sin 1 cos 2 tan 3 exp 4 log 5 acos 6 sf10 ior 7 sqrt 8 a/b (y = x[i] / x[i+1 mod n]) 9 powf(x, 1/2.4)
// 10: rl_roulette_ends(unit = x[i], continue_chance = x[m + i], intensity = x[2m + i]) for i < m = n / 3 (1 or 0)
// 11: rl_normalise((x[i], x[m + i], x[2m + i])) -> (y[i], y[m + i], y[2m + i])
// 12 sin 13 cos 14 exp 15 acos in their f64-evaluated forms (rl_*_d: scene construction, out-of-domain arguments)
// 16: rl_sqrtf, 17: rl_recipf, 18: rl_div200f (their short forms where the whole wave's arguments allow them)
__global__ void rl_math_probe_kernel(int fn, const float* __restrict__ x, float* __restrict__ y, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (fn == 10) { // whole waves take part: the fast path is a wave-uniform decision
        const uint32_t m = n / 3, j = i < m ? i : 0;
        const bool ends = rl_roulette_ends(x[j], x[m + j], x[2 * m + j]);
        if (i < m) y[i] = ends ? 1.0f : 0.0f;
        return;
    }
    if (fn == 11) { // rl_normalise of (x[i], x[m + i], x[2m + i]); whole waves take part: its shortcut is a wave-uniform decision
        const uint32_t m = n / 3, j = i < m ? i : 0;
        const RlF3 u = rl_normalise(rl_f3(x[j], x[m + j], x[2 * m + j]));
        if (i < m) y[i] = u.x, y[m + i] = u.y, y[2 * m + i] = u.z;
        return;
    }
    if (i >= n) return;
    const float v = x[i];
    float r = 0.0f;
    switch (fn) {
    case 0: r = rl_sinf(v); break;
    case 1: r = rl_cosf(v); break;
    case 2: r = rl_tanf(v); break;
    case 3: r = rl_expf(v); break;
    case 4: r = rl_logf(v); break;
    case 5: r = rl_acosf(v); break;
    case 6: r = rl_sf10_ior(v); break;
    case 7: r = sqrtf(v); break;
    case 8: r = v / x[(i + 1) % n]; break;
    case 9: r = rl_powf(v, 1.0f / 2.4f); break;
    case 16: r = rl_sqrtf(v); break;
    case 17: r = rl_recipf(v); break;
    case 18: r = rl_div200f(v); break;
    case 12: r = rl_sinf_d(v); break;
    case 13: r = rl_cosf_d(v); break;
    case 14: r = rl_expf_d(v); break;
    case 15: r = rl_acosf_d(v); break;
    }
    y[i] = r;
}

// Diagnostics (robigo_luculenta_debug.h): the short forms of rl_math.h / rl_core.h against the compiler's correctly rounded
// expansions for EVERY float whose bits lie in [lo, hi) (and, both_signs, its negative): what tools/sqrt_exhaustive.hip did with
// its own copies of the formulas, here with the functions the kernels call -- 1.9 G / 6.7 G arguments in seconds, no host traffic.
// fn: 16 rl_sqrtf vs sqrtf, 17 rl_recipf vs 1 / x, 18 rl_div200f vs x / 200.  Consecutive floats share a wave, so a wave takes the
// short form exactly where the range test of the function lets it.  bad[0] = mismatches, bad[1] = arguments compared.
__global__ __launch_bounds__(RL_BLOCK) void rl_math_sweep_kernel(int fn, uint32_t lo, uint32_t hi, int both_signs, unsigned long long* __restrict__ bad,
                                                                 uint32_t* __restrict__ example) {
    const uint64_t stride = (uint64_t)gridDim.x * RL_BLOCK;
    unsigned long long mine = 0, seen = 0;
    uint32_t last = 0;
    // (whole waves iterate together: the short forms decide per wave)
    const uint64_t n = (uint64_t)hi - lo, rounds = (n + stride - 1) / stride;
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t off = r * stride + (uint64_t)blockIdx.x * RL_BLOCK + threadIdx.x;
        const bool live = off < n;
        const uint32_t bits = lo + (uint32_t)(live ? off : 0);
        for (int sign = 0; sign <= (both_signs ? 1 : 0); ++sign) {
            const float x = rl_u2f(bits | ((uint32_t)sign << 31));
            float got, want;
            if (fn == 16) got = rl_sqrtf(x), want = sqrtf(x);
            else if (fn == 17) got = rl_recipf(x), want = 1.0f / x;
            else got = rl_div200f(x), want = x / 200.0f;
            if (live) {
                seen += 1;
                if (rl_f2u(got) != rl_f2u(want) && !(got != got && want != want)) mine += 1, last = rl_f2u(x);
            }
        }
    }
    if (mine) atomicAdd(&bad[0], mine), example[0] = last;
    atomicAdd(&bad[1], seen);
}

// Diagnostics (robigo_luculenta_debug.h): the prism shortcut and the Compound tree for n rays against one prism of the scene.
__global__ void rl_prism_probe_kernel(const RlF4* __restrict__ prisms, const float* __restrict__ rays, uint32_t n, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RlF3 o = rl_f3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    const RlF3 d = rl_f3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    RlCand fast;
    const int status = rl_hex_prism_fast(prisms, o, d, &fast);
    const RlCand tree = rl_hex_prism(prisms, o, d);
    out[5 * i] = (uint32_t)status;
    out[5 * i + 1] = rl_f2u(fast.t);
    out[5 * i + 2] = fast.k;
    out[5 * i + 3] = tree.t >= 0.0f ? rl_f2u(tree.t) : 0xffffffffu;
    out[5 * i + 4] = tree.k;
}

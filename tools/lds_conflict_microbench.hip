// lds_conflict_microbench.hip -- TOOL: what each class of LDS access of the trace kernel costs in LDS-array cycles and bank
// conflicts, measured with the same shape as the kernel (one workgroup of 16 waves per CU).  One __global__ per access class so
// that a counter pass reports them by kernel name:
//   hipcc -O2 --offload-arch=gfx950 -o tools/lds_mb tools/lds_conflict_microbench.hip
//   rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS -f csv -d out -o p -- tools/lds_mb
// Every kernel executes ITER x 8 LDS instructions of its class per wave; the access pattern of iteration i is a hash of (lane, i)
// shaped like the kernel's: ring entries are pushed in ascending lane order per ballot, a round mixes a few pushes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITER 4096
typedef __attribute__((address_space(3))) uint32_t LdsU32;
typedef __attribute__((address_space(3))) unsigned long long LdsU64;
struct alignas(16) F4 { float x, y, z, w; };

__device__ __forceinline__ uint32_t hash(uint32_t a, uint32_t b) {
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
    h ^= h >> 15; h *= 0xC2B2AE3Du; h ^= h >> 13;
    return h;
}
// owners of one round: four pushes of ~16 ascending lanes each (what the ring holds after a few ballots)
__device__ __forceinline__ uint32_t owner_like_a_round(uint32_t lane, uint32_t i) {
    const uint32_t push = lane >> 4, within = lane & 15u;
    const uint32_t start = hash(push, i) & 15u;           // each push: a run of ascending lanes with random gaps
    return (start + within * 3u + (hash(lane, i) & 1u)) & 63u;
}

extern __shared__ F4 smem[];

#define KERNEL(NAME, BODY)                                                                   \
    __global__ __launch_bounds__(1024) void NAME(float* out) {                               \
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;                    \
        F4* base = smem + wave * 512; /* 8 KB per wave */                                    \
        for (uint32_t k = lane; k < 512; k += 64) base[k] = F4{(float)k, 1.0f, 2.0f, 3.0f};  \
        __syncthreads();                                                                     \
        float acc = 0.0f;                                                                    \
        for (uint32_t i = 0; i < ITER; ++i) {                                                \
            _Pragma("unroll") for (uint32_t u = 0; u < 8; ++u) { BODY }                      \
        }                                                                                    \
        if (acc == 123.456f) out[threadIdx.x] = acc;                                         \
    }

KERNEL(mb_bpermute_identity, { acc += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)(lane << 2), __builtin_bit_cast(int, acc + (float)u))); })
KERNEL(mb_bpermute_round, { const uint32_t o = owner_like_a_round(lane, i * 8 + u); acc += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)(o << 2), __builtin_bit_cast(int, acc + (float)u))); })
KERNEL(mb_bpermute_random, { const uint32_t o = hash(lane, i * 8 + u) & 63u; acc += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)(o << 2), __builtin_bit_cast(int, acc + (float)u))); })
KERNEL(mb_read128_broadcast, { const F4 v = base[(i * 8 + u) & 255u]; acc += v.x; asm volatile("" ::: "memory"); })
KERNEL(mb_read128_members_22clusters, { const uint32_t ci = hash(lane >> 1, i) % 22u; const F4 v = base[1 + 15 * ci + u]; acc += v.x; asm volatile("" ::: "memory"); })
KERNEL(mb_read128_objects_stride2, { const uint32_t ob = hash(lane, i * 8 + u) % 250u; const F4 v = base[2 * ob]; acc += v.x; asm volatile("" ::: "memory"); })
KERNEL(mb_read128_objects_stride1, { const uint32_t ob = hash(lane, i * 8 + u) % 250u; const F4 v = base[ob]; acc += v.x; asm volatile("" ::: "memory"); })
KERNEL(mb_read128_prisms_stride17, { const uint32_t pr = hash(lane >> 1, i) % 22u; const F4 v = base[17 * pr + u]; acc += v.x; asm volatile("" ::: "memory"); })
KERNEL(mb_read32_random, { const uint32_t p = hash(lane, i * 8 + u) % 330u; acc += ((const float*)base)[p]; asm volatile("" ::: "memory"); })
KERNEL(mb_read32_consecutive, { acc += ((const float*)base)[(i * 8 + u + lane) & 127u]; asm volatile("" ::: "memory"); })
KERNEL(mb_write32_consecutive, { ((float*)base)[(i * 8 + u + lane) & 127u] = acc + (float)u; asm volatile("" ::: "memory"); })
KERNEL(mb_min_u64_round, { const uint32_t o = owner_like_a_round(lane, i * 8 + u); __hip_atomic_fetch_min((LdsU64*)base + o, (unsigned long long)hash(lane, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); asm volatile("" ::: "memory"); })
KERNEL(mb_min_u64_distinct, { __hip_atomic_fetch_min((LdsU64*)base + lane, (unsigned long long)hash(lane, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); asm volatile("" ::: "memory"); })

int main() {
    float* out;
    hipMalloc(&out, 4096);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
#define RUN(NAME)                                                                                                        \
    {                                                                                                                    \
        hipFuncSetAttribute((const void*)NAME, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                  \
        hipEvent_t a, b;                                                                                                 \
        hipEventCreate(&a); hipEventCreate(&b);                                                                          \
        hipLaunchKernelGGL(NAME, dim3(cus), dim3(1024), 128 * 1024, 0, out);                                             \
        hipEventRecord(a, 0);                                                                                            \
        hipLaunchKernelGGL(NAME, dim3(cus), dim3(1024), 128 * 1024, 0, out);                                             \
        hipEventRecord(b, 0);                                                                                            \
        hipEventSynchronize(b);                                                                                          \
        float ms = 0;                                                                                                    \
        hipEventElapsedTime(&ms, a, b);                                                                                  \
        /* 16 waves per CU share the LDS: cycles per wave-instruction of LDS time = time x clock / (ITER x 8 x 16) */     \
        printf("%-32s %8.3f ms  %6.1f CU-cycles per wave-instruction at 2.4 GHz (16 waves per CU)\n", #NAME, ms, ms * 1e-3 * 2.4e9 / (ITER * 8.0 * 16.0)); \
    }
    RUN(mb_bpermute_identity) RUN(mb_bpermute_round) RUN(mb_bpermute_random) RUN(mb_read128_broadcast) RUN(mb_read128_members_22clusters)
    RUN(mb_read128_objects_stride2) RUN(mb_read128_objects_stride1) RUN(mb_read128_prisms_stride17) RUN(mb_read32_random)
    RUN(mb_read32_consecutive) RUN(mb_write32_consecutive) RUN(mb_min_u64_round) RUN(mb_min_u64_distinct)
    return 0;
}

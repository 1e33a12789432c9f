// sqrt_exhaustive.hip -- TOOL: candidate short forms of a correctly rounded f32 square root against the compiler's IEEE expansion
// (-fhip-fp32-correctly-rounded-divide-sqrt) for EVERY positive normal float from 2^-96 up: counts and examples of mismatches.
//   hipcc -O2 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o tools/sqrt_ex tools/sqrt_exhaustive.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

__device__ __forceinline__ float m1(float x) { // rsq, one residual correction
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y, h = 0.5f * y;
    return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}
__device__ __forceinline__ float m2(float x) { // rsq, one Goldschmidt step, residual correction
    const float y = __builtin_amdgcn_rsqf(x);
    float s = x * y, h = 0.5f * y;
    const float e = __builtin_fmaf(-h, s, 0.5f);
    s = __builtin_fmaf(s, e, s);
    h = __builtin_fmaf(h, e, h);
    return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}
__device__ __forceinline__ float m3(float x) { // hardware sqrt (1 ulp), residual correction with rsq / 2
    const float s = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}
__device__ __forceinline__ float m5(float x) { return __builtin_amdgcn_sqrtf(x); } // control: the hardware's 1-ulp square root alone
__device__ __forceinline__ float m6(float x) { const float y = __builtin_amdgcn_rsqf(x); return x * y; } // control: x * rsq(x)
__device__ __forceinline__ float m4(float x) { // m1, then a second residual correction
    const float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    const float h = 0.5f * y;
    s = __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
    return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}
// one-operand divisions: the reciprocal and the quotient by the constant 200 (material.rs:224, :283, camera.rs:100)
__device__ __forceinline__ float r1(float x) { // 1 / x: rcp, one Newton step
    const float y = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, y, 1.0f), y, y);
}
__device__ __forceinline__ float r2(float x) { // 1 / x: rcp, Newton step, residual correction of the quotient 1 * y
    float y = __builtin_amdgcn_rcpf(x);
    y = __builtin_fmaf(__builtin_fmaf(-x, y, 1.0f), y, y);
    return __builtin_fmaf(__builtin_fmaf(-x, y, 1.0f), y, y);
}
__device__ __forceinline__ float d200a(float x) { // x / 200: product with fl(1/200), one residual correction
    const float c = 0.005f;
    const float q = x * c;
    return __builtin_fmaf(__builtin_fmaf(-200.0f, q, x), c, q);
}
__device__ __forceinline__ float d200b(float x) { // ... two residual corrections
    const float c = 0.005f;
    float q = x * c;
    q = __builtin_fmaf(__builtin_fmaf(-200.0f, q, x), c, q);
    return __builtin_fmaf(__builtin_fmaf(-200.0f, q, x), c, q);
}
__global__ void sweep_div(uint32_t lo, uint32_t hi, unsigned long long* bad, uint32_t* example) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long b[4] = {0, 0, 0, 0};
    for (uint64_t bits = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; bits < hi; bits += stride) {
        for (uint32_t sign = 0; sign < 2; ++sign) {
            const float x = __builtin_bit_cast(float, (uint32_t)bits | (sign << 31));
            const float want_r = 1.0f / x, want_d = x / 200.0f;
            const float got[4] = {r1(x), r2(x), d200a(x), d200b(x)};
            const float want[4] = {want_r, want_r, want_d, want_d};
            for (int k = 0; k < 4; ++k)
                if (__builtin_bit_cast(uint32_t, got[k]) != __builtin_bit_cast(uint32_t, want[k])) {
                    b[k] += 1;
                    example[k] = (uint32_t)bits | (sign << 31);
                }
        }
    }
    for (int k = 0; k < 4; ++k)
        if (b[k]) atomicAdd(&bad[k], b[k]);
}
__global__ void sweep(uint32_t lo, uint32_t hi, unsigned long long* bad, uint32_t* example) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long b[6] = {0, 0, 0, 0, 0, 0};
    for (uint64_t bits = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; bits < hi; bits += stride) {
        const float x = __builtin_bit_cast(float, (uint32_t)bits);
        const float want = sqrtf(x);
        const float got[6] = {m1(x), m2(x), m3(x), m4(x), m5(x), m6(x)};
        for (int k = 0; k < 6; ++k)
            if (__builtin_bit_cast(uint32_t, got[k]) != __builtin_bit_cast(uint32_t, want)) {
                b[k] += 1;
                example[k] = (uint32_t)bits;
            }
    }
    for (int k = 0; k < 6; ++k)
        if (b[k]) atomicAdd(&bad[k], b[k]);
}
int main() {
    unsigned long long* bad;
    uint32_t* ex;
    hipMalloc(&bad, 48);
    hipMalloc(&ex, 24);
    hipMemset(bad, 0, 48);
    hipMemset(ex, 0, 24);
    const uint32_t lo = 0x0f800000u, hi = 0x7f800000u;
    hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, 0, lo, hi, bad, ex);
    unsigned long long hb[6];
    uint32_t he[6];
    hipMemcpy(hb, bad, 48, hipMemcpyDeviceToHost);
    hipMemcpy(he, ex, 24, hipMemcpyDeviceToHost);
    const char* names[6] = {"m1 rsq + residual", "m2 rsq + Goldschmidt + residual", "m3 sqrt + residual (h from rsq)", "m4 rsq + two residual corrections", "control: v_sqrt_f32 alone", "control: x * v_rsq_f32(x)"};
    for (int k = 0; k < 6; ++k) {
        float x;
        memcpy(&x, &he[k], 4);
        printf("%-36s mismatches %llu of %u (last at 0x%08x = %g)\n", names[k], hb[k], hi - lo, he[k], x);
    }
    // one-operand divisions over 2^-100 <= |x| < 2^100, both signs
    hipMemset(bad, 0, 48);
    hipMemset(ex, 0, 24);
    const uint32_t dlo = 0x0d800000u, dhi = 0x71800000u;
    hipLaunchKernelGGL(sweep_div, dim3(4096), dim3(256), 0, 0, dlo, dhi, bad, ex);
    hipMemcpy(hb, bad, 48, hipMemcpyDeviceToHost);
    hipMemcpy(he, ex, 24, hipMemcpyDeviceToHost);
    const char* dn[4] = {"1/x: rcp + Newton", "1/x: rcp + Newton + correction", "x/200: x fl(1/200) + one correction", "x/200: two corrections"};
    for (int k = 0; k < 4; ++k) {
        float x;
        memcpy(&x, &he[k], 4);
        printf("%-36s mismatches %llu of %llu (last at 0x%08x = %g)\n", dn[k], hb[k], 2ull * (dhi - dlo), he[k], x);
    }
    return 0;
}

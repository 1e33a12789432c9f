// mfma_mix_microbench.hip -- does a v_mfma_f32_16x16x4_f32 issued between independent VALU work cost VALU
// issue slots on gfx950?  Each loop step runs 32 independent v_mul_f32 and NM independent MFMAs (four
// accumulator chains).  If the matrix pipe runs beside the VALU, time(NM) ~ time(0) until the matrix pipe
// itself saturates (32 cycles per MFMA per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_mix_microbench.hip -o /tmp/mfma_mb
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));
#define ITERS 4096

template <int NM>
__global__ __launch_bounds__(256) void mix(float* out, float a, float b) {
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = a + threadIdx.x * 1e-3f + i;
    v4f acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = (v4f){0.f, 0.f, 0.f, 0.f};
    float am = a + threadIdx.x, bm = b - threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (NM > 0 && (i % (32 / (NM > 32 ? 32 : NM))) == 0) {
#pragma unroll
                for (int r = 0; r < (NM > 32 ? NM / 32 : 1); ++r)
                    acc[(i + r) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, acc[(i + r) & 3], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 32; ++i) s += x[i];
    for (int j = 0; j < 4; ++j) s += acc[j].x + acc[j].y + acc[j].z + acc[j].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM>
void run(int waves_per_simd) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * waves_per_simd;
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    mix<NM><<<blocks, 256>>>(out, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mix<NM><<<blocks, 256>>>(out, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double steps_per_simd = (double)waves_per_simd * ITERS;          // each step = 32 VALU + NM MFMA per wave
    const double cycles_per_step = ms * 1e-3 * 2.4e9 / steps_per_simd;
    printf("waves/SIMD=%d  32 v_mul + %2d mfma_16x16x4 per step: %.3f ms, %.1f cycles per step per SIMD (VALU alone would be ~%.0f)\n",
           waves_per_simd, NM, ms, cycles_per_step, 32 * 2.8);
    hipFree(out);
}

int main() {
    for (int w : {4}) {
        run<0>(w);
        run<1>(w);
        run<2>(w);
        run<4>(w);
        run<8>(w);
        run<16>(w);
    }
    return 0;
}

// valu_microbench.hip -- measures the f32 VALU issue rate of gfx950 for the instruction mixes the
// trace kernel uses, so the roofline in DESIGN.md is priced against measured, not assumed, rates.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_microbench.hip -o /tmp/valu_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 4096
#define UNROLL 16

template <int MODE>
__global__ __launch_bounds__(256) void mb(float* out, float a, float b) {
    float x[UNROLL];
    float2 p[UNROLL];
    for (int i = 0; i < UNROLL; ++i) {
        x[i] = a + threadIdx.x * 1e-3f + i;
        p[i] = make_float2(x[i], x[i] + 0.5f);
    }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (MODE == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (MODE == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) % UNROLL]), "v"(p[(i + 2) % UNROLL]));
            if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) % UNROLL]));
            if (MODE == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) % UNROLL]));
            if (MODE == 6) asm volatile("v_sub_f32 %0, s4, %0" : "+v"(x[i]) : : "s4");
            if (MODE == 7) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc");
            if (MODE == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : "vcc");
            if (MODE == 9) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i]));
            if (MODE == 10) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
            if (MODE == 11) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(*(double*)&p[i]) : "v"(*(double*)&p[(i + 1) % UNROLL]), "v"(*(double*)&p[(i + 2) % UNROLL]));
            if (MODE == 12) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (MODE == 13) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (MODE == 14) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(x[i]) : "v"(a) : "s10", "s11");
            if (MODE == 15) asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %1" : : "v"(x[i]), "v"(a) : "s10", "s11");
            if (MODE == 16) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(a));
            if (MODE == 17) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (MODE == 18) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : "vcc");
            if (MODE == 19) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_and_saveexec_b64 s[10:11], vcc\n\tv_mov_b32 %0, %1\n\ts_mov_b64 exec, s[10:11]" : "+v"(x[i]) : "v"(a) : "vcc", "s10", "s11");
            if (MODE == 20) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (MODE == 21) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (MODE == 22) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x[i]) : "v"(a), "v"(b) : "vcc");
            if (MODE == 23) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "s"(b));
            if (MODE == 24) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
            if (MODE == 25) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, 0" : "=v"(*(unsigned long long*)&p[i]) : "v"(x[i]), "v"(a) : "s10", "s11");
            if (MODE == 26) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        }
    }
    float s = 0;
    for (int i = 0; i < UNROLL; ++i) s += x[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float flops_per_lane_inst, int waves_per_simd) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * waves_per_simd; // 256 threads = 4 waves = 1 per SIMD
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    mb<MODE><<<blocks, 256>>>(out, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mb<MODE><<<blocks, 256>>>(out, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_insts = (double)blocks * 4 * ITERS * UNROLL;
    const double per_simd_per_s = wave_insts / (cus * 4.0) / (ms * 1e-3);
    printf("%-14s waves/SIMD=%d  %.3f ms  %.3f G wave-inst/s/SIMD  (%.2f cycles/inst @2.4GHz)  %.1f TFLOP/s\n", name,
           waves_per_simd, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s,
           wave_insts * 64 * flops_per_lane_inst / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    for (int w : {4}) {
        run<0>("v_fma_f32", 2, w);
        run<1>("v_mul_f32", 1, w);
        run<2>("v_add_f32", 1, w);
        run<3>("v_pk_fma_f32", 4, w);
        run<4>("v_pk_mul_f32", 2, w);
        run<5>("v_pk_add_f32", 2, w);
        run<6>("v_sub_f32 sgpr", 1, w);
        run<7>("v_cmp_lt_f32", 1, w);
        run<8>("v_cndmask_b32", 1, w);
        run<9>("v_sqrt_f32", 1, w);
        run<10>("v_rcp_f32", 1, w);
        run<11>("v_fma_f64", 2, w);
        run<12>("v_mul_lo_u32", 1, w);
        run<13>("v_mul_hi_u32", 1, w);
        run<14>("cndmask e64 sgpr", 1, w);
        run<15>("v_cmp e64 sgpr", 1, w);
        run<16>("v_mov_b32", 1, w);
        run<17>("v_min_f32", 1, w);
        run<18>("cmp+cndmask", 1, w);
        run<19>("cmp+saveexec+mov", 1, w);
        run<20>("v_add_u32", 1, w);
        run<21>("v_and_b32", 1, w);
        run<22>("cndmask 3reg", 1, w);
        run<23>("v_sub_f32 s-arg", 1, w);
        run<24>("v_fma acc", 2, w);
        run<25>("v_mad_u64_u32", 1, w);
        run<26>("v_xor_b32", 1, w);
    }
    return 0;
}

// placement_probe.hip -- TOOL: how does the dispatcher spread the waves of workgroups over a CU's four SIMDs when a wave's
// registers allow five waves per SIMD?  (round 6: the trace kernel at 2 x 640 threads per CU ran at the speed of 2.5 waves per SIMD.)
// Every wave allocates `96` VGPRs (v95 is touched), stays resident ~3 ms and records where it ran (HW_ID / XCC_ID) and when;
// the host prints, per workgroup size, how many waves each SIMD held AT THE SAME TIME (overlap by timestamps).
//   hipcc -O2 --offload-arch=gfx950 -o tools/placement_probe tools/placement_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
struct Rec { unsigned hw_id, xcc_id; unsigned long long t0, t1; };
template <int THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_num_vgpr(48))) void probe(Rec* out, unsigned ticks, unsigned lds_floats) {
    extern __shared__ float smem[];
    if (threadIdx.x < lds_floats) smem[threadIdx.x] = 1.0f;
    asm volatile("v_mov_b32 v95, 0" ::: "v95");
    Rec r;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(r.hw_id), "=s"(r.xcc_id));
    r.t0 = wall_clock64();
    while (wall_clock64() - r.t0 < ticks) __builtin_amdgcn_s_sleep(32);
    r.t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * THREADS + threadIdx.x) / 64] = r;
}
template <int THREADS>
void run(int cus, int wgs_per_cu, size_t lds) {
    const int blocks = cus * wgs_per_cu, waves = blocks * THREADS / 64;
    Rec* out;
    hipMalloc(&out, sizeof(Rec) * waves);
    hipFuncSetAttribute((const void*)probe<THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe<THREADS>, THREADS, lds);
    hipLaunchKernelGGL(probe<THREADS>, dim3(blocks), dim3(THREADS), lds, 0, out, 300000u, 64u); // 3 ms at 100 MHz
    hipDeviceSynchronize();
    std::vector<Rec> h(waves);
    hipMemcpy(h.data(), out, sizeof(Rec) * waves, hipMemcpyDeviceToHost);
    unsigned long long first = ~0ull;
    for (auto& r : h) first = std::min(first, r.t0);
    // waves resident 1 ms after the first wave started, per SIMD
    std::map<unsigned, int> per_simd, per_cu;
    int resident = 0;
    for (auto& r : h)
        if (r.t0 <= first + 100000ull && r.t1 > first + 100000ull) {
            const unsigned hh = r.hw_id;
            const unsigned cu = ((((r.xcc_id & 15u) * 8 + ((hh >> 13) & 7u)) * 2 + ((hh >> 12) & 1u)) * 16 + ((hh >> 8) & 15u));
            per_simd[cu * 4 + ((hh >> 4) & 3u)] += 1;
            per_cu[cu] += 1;
            resident += 1;
        }
    std::map<int, int> hist, hist_cu;
    for (auto& kv : per_simd) hist[kv.second] += 1;
    for (auto& kv : per_cu) hist_cu[kv.second] += 1;
    std::printf("%4d threads x %d per CU (%zu B LDS each, occupancy API says %d): %d of %d waves resident at +1 ms; waves per SIMD:", THREADS, wgs_per_cu, lds, occ, resident, waves);
    for (auto& kv : hist) std::printf("  %d waves on %d SIMDs", kv.first, kv.second);
    std::printf(" ; waves per CU:");
    for (auto& kv : hist_cu) std::printf("  %d on %d CUs", kv.first, kv.second);
    std::printf("\n");
    hipFree(out);
}
// two launches on two streams: a 1024-thread workgroup per CU first, then a 256-thread one beside it -- five waves on every SIMD?
void run_pair(int cus, size_t lds_a, size_t lds_b) {
    const int waves_a = cus * 16, waves_b = cus * 4;
    Rec *oa, *ob;
    hipMalloc(&oa, sizeof(Rec) * waves_a);
    hipMalloc(&ob, sizeof(Rec) * waves_b);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipFuncSetAttribute((const void*)probe<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)probe<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe<1024>, dim3(cus), dim3(1024), lds_a, sa, oa, 300000u, 64u);
    hipLaunchKernelGGL(probe<256>, dim3(cus), dim3(256), lds_b, sb, ob, 300000u, 64u);
    hipDeviceSynchronize();
    std::vector<Rec> h(waves_a + waves_b);
    hipMemcpy(h.data(), oa, sizeof(Rec) * waves_a, hipMemcpyDeviceToHost);
    hipMemcpy(h.data() + waves_a, ob, sizeof(Rec) * waves_b, hipMemcpyDeviceToHost);
    unsigned long long first = ~0ull;
    for (auto& r : h) first = std::min(first, r.t0);
    std::map<unsigned, int> per_simd, b_per_cu;
    int resident = 0;
    for (size_t i = 0; i < h.size(); ++i) {
        const Rec& r = h[i];
        if (r.t0 <= first + 100000ull && r.t1 > first + 100000ull) {
            const unsigned hh = r.hw_id;
            const unsigned cu = ((((r.xcc_id & 15u) * 8 + ((hh >> 13) & 7u)) * 2 + ((hh >> 12) & 1u)) * 16 + ((hh >> 8) & 15u));
            per_simd[cu * 4 + ((hh >> 4) & 3u)] += 1;
            if (i >= (size_t)waves_a) b_per_cu[cu] += 1;
            resident += 1;
        }
    }
    std::map<int, int> hist, hb;
    for (auto& kv : per_simd) hist[kv.second] += 1;
    for (auto& kv : b_per_cu) hb[kv.second] += 1;
    std::printf("pair 1024 (%zu B) + 256 (%zu B): %d of %d waves resident at +1 ms; waves per SIMD:", lds_a, lds_b, resident, waves_a + waves_b);
    for (auto& kv : hist) std::printf("  %d waves on %d SIMDs", kv.first, kv.second);
    std::printf(" ; waves of the 256-thread launch per CU:");
    for (auto& kv : hb) std::printf("  %d on %d CUs", kv.first, kv.second);
    std::printf("\n");
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    run_pair(cus, 18 * 1024 + 16 * 6144, 18 * 1024 + 4 * 6144);
    run_pair(cus, 16 * 6144, 4 * 6144);
    run_pair(cus, 16 * 8192 - 2048, 4 * 8192 - 2048);
    run_pair(cus, 16 * 8192 - 512, 4 * 8192 - 512);
    run_pair(cus, 16 * 8192, 4 * 8192);
    run<256>(cus, 5, 31 * 1024);
    run<256>(cus, 5, 32 * 1024 - 512);
    run<256>(cus, 5, 30 * 1024);
    run<128>(cus, 10, 15 * 1024);
    run<320>(cus, 4, 39 * 1024);
    run<640>(cus, 2, 80 * 1024);
    run<640>(cus, 2, 1024);
    run<320>(cus, 4, 40 * 1024);
    run<256>(cus, 5, 32 * 1024);
    run<512>(cus, 2, 64 * 1024);
    run<1024>(cus, 1, 128 * 1024);
    run<1024>(cus, 2, 64 * 1024);
    run<768>(cus, 1, 64 * 1024);
    run<384>(cus, 3, 48 * 1024);
    run<128>(cus, 10, 16 * 1024);
    return 0;
}

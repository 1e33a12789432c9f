// Test infrastructure (not the product): fills every CU's LDS with a pattern, so that a trace kernel launched afterwards finds
// that pattern -- not whatever the previous kernel happened to leave -- in the ring slots it has not written yet.  A round of
// fewer than 64 pairs reads such slots (rl_scan_wave: the lanes beyond the round are pointed at record 0 before anything is
// loaded through them); with all ones in them a kernel that loaded through a stale entry from global memory would fault.
#include <hip/hip_runtime.h>

#include <cstdint>

__global__ __launch_bounds__(1024) void poison_kernel(uint32_t pattern, uint32_t n_words, unsigned long long* sink) {
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < n_words; i += 1024) lds[i] = pattern;
    __syncthreads();
    // (read one word back so that the stores are not dead code)
    if (threadIdx.x == 0 && lds[(blockIdx.x * 977u) % n_words] != pattern) atomicAdd(sink, 1ull);
}

// Returns 0 on success.  `blocks` workgroups of 1024 threads with the CU's whole 160 KB each: one per CU at a time, so a grid of a
// few times the CU count reaches every CU.
extern "C" int lds_poison(int device, uint32_t pattern, uint32_t blocks) {
    if (hipSetDevice(device) != hipSuccess) return 1;
    const size_t bytes = 160 * 1024;
    if (hipFuncSetAttribute((const void*)poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return 2;
    unsigned long long* sink = nullptr;
    if (hipMalloc((void**)&sink, sizeof *sink) != hipSuccess) return 3;
    (void)hipMemset(sink, 0, sizeof *sink);
    hipLaunchKernelGGL(poison_kernel, dim3(blocks), dim3(1024), bytes, 0, pattern, (uint32_t)(bytes / 4), sink);
    int rc = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess ? 0 : 4;
    unsigned long long bad = 0;
    if (rc == 0 && hipMemcpy(&bad, sink, sizeof bad, hipMemcpyDeviceToHost) != hipSuccess) rc = 5;
    (void)hipFree(sink);
    return rc != 0 ? rc : (bad != 0 ? 6 : 0);
}

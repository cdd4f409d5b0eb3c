// Calibration for the L2 memory-side byte counters: a streaming copy with the
// engine's access shape (8 bytes per lane, consecutive lanes on consecutive
// doubles) over a known byte count.  Run under `rocprofv3 --pmc FETCH_SIZE`
// and `--pmc WRITE_SIZE`; reported / known gives the correction factor for
// 8-byte-per-lane traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_copy8(const double *__restrict__ in, double *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i] + 1.0;
}

int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1024;  // bytes read = bytes written = mib MiB
    const size_t n = mib * (1u << 20) / 8;
    double *a, *b;
    if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&b, n * 8) != hipSuccess) return 1;
    hipMemset(a, 0, n * 8);
    hipMemset(b, 0, n * 8);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_copy8, dim3(8192), dim3(256), 0, 0, a, b, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("copy8 %zu MiB read + %zu MiB written: %.3f ms  %.1f GB/s\n", mib, mib, ms, 2.0 * n * 8 / ms / 1e6);
    }
    hipFree(a);
    hipFree(b);
    return 0;
}

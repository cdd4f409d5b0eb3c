// Sustained rate of unfused fp64 multiply + add (the blur kernels' arithmetic)
// with 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_dp(double *out, double a, double b, int iters) {
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = x[i] * a;
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = x[i] + b;
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    double *out;
    (void)hipMalloc(&out, 8192 * 1024 * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;  // 256-thread blocks: 4 waves each, one per SIMD
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_dp, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double ops = (double)blocks * 256 * iters * 128.0;
            if (rep) printf("waves/SIMD %d: %.3f ms, %.2f T fp64 ops/s (mul or add), %.1f%% of 256 CU x 64 lanes x 2.4 GHz\n", wps, ms,
                            ops / ms / 1e9, 100.0 * ops / ms / 1e9 / 39.32);
        }
    }
    return 0;
}

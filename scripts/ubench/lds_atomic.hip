// Micro-benchmark: cycles per LDS atomic wave-instruction for several types,
// conflict-free (each lane its own address) and 4 lanes per address.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <class T, int SHARE>
__global__ void k(T *out, unsigned long long *cyc, int iters) {
    __shared__ T buf[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = T(0);
    __syncthreads();
    const int lane = threadIdx.x;
    T v = T(lane + 1);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) atomicAdd(&buf[(lane / SHARE) + u * 64 + (threadIdx.x / 64) * 512], v);
    }
    __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = buf[threadIdx.x];
}

template <class T, int SHARE>
void run(const char *name, int threads) {
    T *out; unsigned long long *cyc;
    hipMalloc(&out, 1024 * 256 * sizeof(T)); hipMalloc(&cyc, 1024 * 8);
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL((k<T, SHARE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<T, SHARE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < blocks; i++) avg += h[i]; avg /= blocks;
    const double waves = threads / 64.0;
    printf("%-22s share=%d waves/block=%g: %.1f cycles per wave-instruction (per CU: %.1f)\n", name, SHARE, waves,
           avg / (iters * 8.0), avg / (iters * 8.0 * waves));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<double, 1>("ds_add_f64", 64); run<double, 1>("ds_add_f64", 256);
    run<double, 4>("ds_add_f64", 256); run<double, 16>("ds_add_f64", 256);
    run<unsigned long long, 1>("ds_add_u64", 64); run<unsigned long long, 1>("ds_add_u64", 256);
    run<unsigned long long, 4>("ds_add_u64", 256); run<unsigned long long, 16>("ds_add_u64", 256);
    run<float, 1>("ds_add_f32", 256); run<float, 4>("ds_add_f32", 256); run<float, 16>("ds_add_f32", 256);
    run<unsigned, 1>("ds_add_u32", 256); run<unsigned, 4>("ds_add_u32", 256); run<unsigned, 16>("ds_add_u32", 256);
    return 0;
}

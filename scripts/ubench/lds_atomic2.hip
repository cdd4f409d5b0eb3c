// Cost of ds_add_f64 under the descriptor kernel's access pattern: every lane
// adds to a pseudo-random bin of a 128-bin tensor copy (copies 129 doubles
// apart), with a given number of copies per wave and of active lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// COPIES: tensor copies per wave (lane % COPIES picks one); ACTIVE: lanes (spread evenly) that issue;
// SPREAD: bins drawn from [0, SPREAD) (small = neighbouring lanes often share a bin)
template <int COPIES, int ACTIVE, int SPREAD>
__global__ void k(double *out, unsigned long long *cyc, int iters) {
    extern __shared__ double buf[];
    const int nwaves = blockDim.x / 64;
    const int total = nwaves * COPIES * 129;
    for (int i = threadIdx.x; i < total; i += blockDim.x) buf[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int addr[8];
#pragma unroll
    for (int u = 0; u < 8; u++)
        addr[u] = (wv * COPIES + lane % COPIES) * 129 + (mix(threadIdx.x * 8 + u + blockIdx.x * 977) % SPREAD) * (128 / SPREAD);
    const bool on = (lane % (64 / ACTIVE)) == 0;
    double v = lane + 1.0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (on) {
#pragma unroll
            for (int u = 0; u < 8; u++) atomicAdd(&buf[addr[u]], v);
        }
    }
    __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = buf[threadIdx.x];
}

template <int COPIES, int ACTIVE, int SPREAD>
void run(int threads) {
    double *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 1024 * 256 * sizeof(double)); (void)hipMalloc(&cyc, 1024 * 8);
    const int iters = 2000, blocks = 256;
    const size_t lds = (size_t)(threads / 64) * COPIES * 129 * 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<COPIES, ACTIVE, SPREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; rep++)
        hipLaunchKernelGGL((k<COPIES, ACTIVE, SPREAD>), dim3(blocks), dim3(threads), lds, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < blocks; i++) avg += h[i]; avg /= blocks;
    const double waves = threads / 64.0;
    printf("copies/wave %2d active lanes %2d bins %3d waves %g: %.1f ticks per wave-instruction per CU\n", COPIES, ACTIVE, SPREAD,
           waves, avg / (iters * 8.0 * waves));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<8, 64, 128>(256); run<8, 32, 128>(256); run<8, 16, 128>(256); run<8, 8, 128>(256);
    run<8, 64, 16>(256); run<8, 64, 4>(256); run<8, 64, 1>(256);
    run<16, 64, 128>(256); run<16, 64, 16>(256); run<16, 64, 4>(256);
    run<32, 64, 128>(256); run<32, 64, 4>(256);
    run<8, 32, 4>(256); run<8, 16, 4>(256);
    return 0;
}

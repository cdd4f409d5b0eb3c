// How much vector-ALU work fits beside a stream of v_mfma_i32_16x16x64_i8 on one SIMD, and what clock the part
// holds while it runs?  Per MFMA the wave issues K independent v_min3_i32 (K = 0..4), one or two waves per
// SIMD, every CU busy.  Prints wall time, shader clocks per MFMA (s_memtime) and the clock frequency that
// follows from s_memrealtime (100 MHz) over the same interval.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using v4i = __attribute__((ext_vector_type(4))) int;

template <int K, int BT>
__global__ __launch_bounds__(BT) void k_mix(const int *__restrict__ in, int *__restrict__ out, long long *clk, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    v4i a[4], b[4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            a[i][j] = in[(tid * 32 + i * 4 + j) & 0xfffff];
            b[i][j] = in[(tid * 32 + 16 + i * 4 + j) & 0xfffff];
        }
    int f0 = in[tid & 0xfff], f1 = in[(tid + 7) & 0xfff], f2 = in[(tid + 9) & 0xfff], f3 = in[(tid + 11) & 0xfff];
    v4i acc[16] = {};
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 16; q++)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                __builtin_amdgcn_sched_barrier(0);
                acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(k + q) & 3], b[(k + (q >> 2)) & 3], acc[q], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (K >= 1) asm volatile("v_min3_i32 %0, %1, %2, %0" : "+v"(f0) : "v"(f1), "v"(f2));
                if (K >= 2) asm volatile("v_min3_i32 %0, %1, %2, %0" : "+v"(f1) : "v"(f2), "v"(f3));
                if (K >= 3) asm volatile("v_min3_i32 %0, %1, %2, %0" : "+v"(f2) : "v"(f3), "v"(f0));
                if (K >= 4) asm volatile("v_min3_i32 %0, %1, %2, %0" : "+v"(f3) : "v"(f0), "v"(f1));
            }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    int s = f0 + f1 + f2 + f3;
    for (int q = 0; q < 16; q++)
        for (int r = 0; r < 4; r++) s += acc[q][r];
    out[tid] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = w1 - w0;
    }
}

template <int K, int BT>
void run(const int *d_in, int *d_out, long long *d_clk, int blocks, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    long long clk[2] = {0, 0};
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_mix<K, BT>), dim3(blocks), dim3(BT), 0, 0, d_in, d_out, d_clk, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) {
            best = ms;
            (void)hipMemcpy(clk, d_clk, sizeof(clk), hipMemcpyDeviceToHost);
        }
    }
    const double mfma_per_wave = (double)iters * 32;
    const double ops = (double)blocks * (BT / 64) * mfma_per_wave * 2.0 * 16 * 16 * 64;
    const double waves_per_simd = blocks / 256.0 * (BT / 256);
    printf("K=%d waves/SIMD %.0f: %.2f ms  %.0f T op/s  %.1f shader clocks per MFMA and SIMD  clock %.2f GHz\n", K,
           waves_per_simd, best, ops / best / 1e9, clk[0] / mfma_per_wave / waves_per_simd,
           clk[0] / (clk[1] / 100e6) / 1e9);
}

int main() {
    const int n = 1 << 20;
    std::vector<int> h(n);
    srand(1);
    for (int &v : h) v = rand() ^ (rand() << 16);
    int *d_in, *d_out;
    long long *d_clk;
    if (hipMalloc(&d_in, n * 4) || hipMalloc(&d_out, 4096 * 256 * 4) || hipMalloc(&d_clk, 16)) return 1;
    (void)hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice);
    run<0, 256>(d_in, d_out, d_clk, 256, 10000);
    run<1, 256>(d_in, d_out, d_clk, 256, 10000);
    run<2, 256>(d_in, d_out, d_clk, 256, 10000);
    run<3, 256>(d_in, d_out, d_clk, 256, 10000);
    run<4, 256>(d_in, d_out, d_clk, 256, 10000);
    run<0, 512>(d_in, d_out, d_clk, 256, 10000);
    run<1, 512>(d_in, d_out, d_clk, 256, 10000);
    run<2, 512>(d_in, d_out, d_clk, 256, 10000);
    run<3, 512>(d_in, d_out, d_clk, 256, 10000);
    run<4, 512>(d_in, d_out, d_clk, 256, 10000);
    run<0, 1024>(d_in, d_out, d_clk, 256, 10000);
    run<2, 1024>(d_in, d_out, d_clk, 256, 10000);
    run<4, 1024>(d_in, d_out, d_clk, 256, 10000);
    return 0;
}

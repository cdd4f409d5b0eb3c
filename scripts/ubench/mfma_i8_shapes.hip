// Do the two int8 MFMA shapes sustain the same rate on this part under its clock management?  Bare loops on
// random operands held in registers, one and two waves per SIMD, every CU busy; prints T op/s per shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using v4i = __attribute__((ext_vector_type(4))) int;
using v16i = __attribute__((ext_vector_type(16))) int;

template <int SHAPE>
__global__ __launch_bounds__(256) void k_loop(const int *__restrict__ in, int *__restrict__ out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    v4i a[4], b[4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            a[i][j] = in[(tid * 32 + i * 4 + j) & 0xfffff];
            b[i][j] = in[(tid * 32 + 16 + i * 4 + j) & 0xfffff];
        }
    if constexpr (SHAPE == 32) {
        v16i acc[4] = {};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int k = 0; k < 4; k++) acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[k], b[(k + q) & 3], acc[q], 0, 0, 0);
        }
        int s = 0;
        for (int q = 0; q < 4; q++)
            for (int r = 0; r < 16; r++) s += acc[q][r];
        out[tid] = s;
    } else {
        v4i acc[16] = {};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < 16; q++)
#pragma unroll
                for (int k = 0; k < 2; k++) acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(k + q) & 3], b[(k + (q >> 2)) & 3], acc[q], 0, 0, 0);
        }
        int s = 0;
        for (int q = 0; q < 16; q++)
            for (int r = 0; r < 4; r++) s += acc[q][r];
        out[tid] = s;
    }
}

template <int SHAPE>
void run(const int *d_in, int *d_out, int blocks, int iters, const char *name) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_loop<SHAPE>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per wave and iteration: 16 MFMAs of 2*32*32*32 ops, or 32 of 2*16*16*64
    const double ops = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 32;
    printf("%-10s blocks %5d: %.2f ms  %.0f T op/s\n", name, blocks, best, ops / best / 1e9);
}

int main() {
    const int n = 1 << 20;
    std::vector<int> h(n);
    srand(1);
    for (int &v : h) v = rand() ^ (rand() << 16);
    int *d_in, *d_out;
    if (hipMalloc(&d_in, n * 4) || hipMalloc(&d_out, 4096 * 256 * 4)) return 1;
    (void)hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int blocks : {256, 512}) {
        run<32>(d_in, d_out, blocks, 20000, "32x32x32");
        run<16>(d_in, d_out, blocks, 20000, "16x16x64");
        run<32>(d_in, d_out, blocks, 20000, "32x32x32");
        run<16>(d_in, d_out, blocks, 20000, "16x16x64");
    }
    return 0;
}

// Which lanes of a 64-bit LDS integer atomic conflict with which: the cost of ds_add_u64 for lane -> address patterns
// given from the host.  Every pattern is run with one wave and with four waves per workgroup (one workgroup per CU);
// the figure is s_memtime ticks per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <functional>

__global__ void k(const int *pattern, unsigned long long *out, unsigned long long *cyc, int iters) {
    __shared__ unsigned long long buf[4 * 4096];
    for (int i = threadIdx.x; i < 4 * 4096; i += blockDim.x) buf[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long *mine = buf + wv * 4096;
    const int a = pattern[lane];
    const bool on = a >= 0;
    unsigned long long v = lane + 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (on) {
#pragma unroll
            for (int u = 0; u < 8; u++) atomicAdd(&mine[a + u * 512], v);
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = buf[threadIdx.x];
}

static void run(const char *name, std::function<int(int)> f) {
    int h[64];
    for (int l = 0; l < 64; l++) h[l] = f(l);
    int *dp; unsigned long long *out, *cyc;
    (void)hipMalloc(&dp, sizeof(h)); (void)hipMalloc(&out, 256 * 256 * 8); (void)hipMalloc(&cyc, 256 * 8);
    (void)hipMemcpy(dp, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 2000, blocks = 256;
    double res[2];
    int idx = 0;
    for (int threads : {64, 256}) {
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, dp, out, cyc, iters);
        (void)hipDeviceSynchronize();
        unsigned long long c[256];
        (void)hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        double avg = 0;
        for (int i = 0; i < blocks; i++) avg += c[i];
        avg /= blocks;
        res[idx++] = avg / (iters * 8.0 * (threads / 64));
    }
    printf("%-58s 1 wave %6.1f   4 waves %6.1f  clocks per wave-instruction per CU\n", name, res[0], res[1]);
    (void)hipFree(dp); (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run("distinct consecutive (no conflict)", [](int l) { return l; });
    run("32 lanes active (even lanes), distinct", [](int l) { return l % 2 ? -1 : l; });
    run("32 lanes active (first half), distinct", [](int l) { return l < 32 ? l : -1; });
    run("16 lanes active (first 16), distinct", [](int l) { return l < 16 ? l : -1; });
    run("8 lanes active (first 8), distinct", [](int l) { return l < 8 ? l : -1; });
    run("8 lanes active (every 8th), distinct", [](int l) { return l % 8 ? -1 : l; });
    for (int d : {1, 2, 4, 8, 16, 32}) {
        char nm[96];
        snprintf(nm, sizeof nm, "same address: lanes l and l^%d", d);
        run(nm, [d](int l) { return l & ~d; });
    }
    for (int d : {1, 2, 4, 8, 16, 32}) {
        char nm[96];
        snprintf(nm, sizeof nm, "same bank, other address: lanes l and l^%d", d);
        run(nm, [d](int l) { return (l & ~d) + ((l & d) ? 32 * 3 : 0); });  // +96 words of 8 bytes = +768 B: same bank whether 32 or 64 banks
    }
    run("4 lanes per address: l/4", [](int l) { return l / 4; });
    run("4 lanes per address: l%16", [](int l) { return l % 16; });
    run("8 lanes per address: l%8", [](int l) { return l % 8; });
    run("2 lanes per address: l%32", [](int l) { return l % 32; });
    run("stride 2 words (16 B)", [](int l) { return 2 * l; });
    run("stride 4 words (32 B)", [](int l) { return 4 * l; });
    run("stride 16 words (128 B)", [](int l) { return 16 * l; });
    run("stride 32 words (256 B)", [](int l) { return 32 * l; });
    run("stride 33 words", [](int l) { return 33 * l; });
    run("copy-major 16 copies stride 130, same bin", [](int l) { return (l % 16) * 130; });
    run("copy-minor 16 copies, same bin", [](int l) { return (l % 16); });
    // random bins among 16 (two cells' worth of orientation bins), copy = lane % 16
    auto rnd = [](int l) { unsigned x = l * 2654435761u; x ^= x >> 13; x *= 0x85ebca6bu; x ^= x >> 16; return (int)(x % 16); };
    run("copy-major 16 copies stride 130, random 16 bins", [rnd](int l) { return (l % 16) * 130 + rnd(l); });
    run("copy-minor 16 copies, random 16 bins", [rnd](int l) { return rnd(l) * 16 + (l % 16); });
    run("copy-major 8 copies stride 130, random 16 bins", [rnd](int l) { return (l % 8) * 130 + rnd(l); });
    run("copy-minor 8 copies, random 16 bins", [rnd](int l) { return rnd(l) * 8 + (l % 8); });
    return 0;
}

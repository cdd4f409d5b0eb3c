// 1 read : 2 write streaming (the blur kernels' mix) as a function of the byte
// offsets between the three streams modulo large powers of two: does it matter
// whether the input plane and the two output planes start at the same position
// of the memory-channel interleave?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_mix(const double *__restrict__ in, double *__restrict__ o1, double *__restrict__ o2, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const double v = in[i];
        o1[i] = v + 1.0;
        o2[i] = v;
    }
}

int main() {
    const size_t bytes = 1ull << 30;
    char *buf;
    if (hipMalloc(&buf, 3 * bytes + (64u << 20)) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, 3 * bytes + (64u << 20));
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const size_t offs[] = {0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 4352, 12544, 45312};
    for (size_t off : offs) {
        const double *in = (const double *)buf;
        double *o1 = (double *)(buf + bytes + off), *o2 = (double *)(buf + 2 * bytes + 2 * off);
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_mix, dim3(8192), dim3(256), 0, 0, in, o1, o2, bytes / 8);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("stream offsets 0 / %8zu / %8zu bytes: %.3f ms  %.0f GB/s\n", off, 2 * off, best, 3.0 * bytes / best / 1e6);
    }
    return 0;
}

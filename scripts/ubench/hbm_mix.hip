// Achievable HBM rates for the access mixes the pyramid kernels have:
// read-only, 1 read : 1 write and 1 read : 2 writes, at 8 and 16 bytes per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <class T>
__device__ __forceinline__ T add1(T v);
template <>
__device__ __forceinline__ double add1(double v) { return v + 1.0; }
template <>
__device__ __forceinline__ double2 add1(double2 v) { return make_double2(v.x + 1.0, v.y + 1.0); }
__device__ __forceinline__ double first(double v) { return v; }
__device__ __forceinline__ double first(double2 v) { return v.x + v.y; }

template <class T, int NW>
__global__ void k_mix(const T *__restrict__ in, T *__restrict__ o1, T *__restrict__ o2, double *sink, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0;
    for (; i < n; i += stride) {
        const T v = in[i];
        if (NW >= 1) o1[i] = add1(v);
        if (NW >= 2) o2[i] = v;
        if (NW == 0) acc += first(v);
    }
    if (NW == 0 && acc == 12345.678) *sink = acc;
}

template <class T, int NW>
void run(const char *name, void *a, void *b, void *c, double *sink, size_t bytes, int grid) {
    const size_t n = bytes / sizeof(T);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_mix<T, NW>), dim3(grid), dim3(256), 0, 0, (const T *)a, (T *)b, (T *)c, sink, n);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-28s grid %6d: %.3f ms  %.0f GB/s total traffic\n", name, grid, best, (1.0 + NW) * bytes / best / 1e6);
}

int main(int argc, char **argv) {
    const size_t bytes = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 1024) << 20;
    void *a, *b, *c;
    double *sink;
    if (hipMalloc(&a, bytes) || hipMalloc(&b, bytes) || hipMalloc(&c, bytes) || hipMalloc(&sink, 8)) return 1;
    (void)hipMemset(a, 0, bytes);
    (void)hipMemset(b, 0, bytes);
    (void)hipMemset(c, 0, bytes);
    (void)hipDeviceSynchronize();
    for (int grid : {2048, 8192, 32768}) {
        run<double, 0>("read 8B/lane", a, b, c, sink, bytes, grid);
        run<double2, 0>("read 16B/lane", a, b, c, sink, bytes, grid);
        run<double, 1>("1R:1W 8B/lane", a, b, c, sink, bytes, grid);
        run<double2, 1>("1R:1W 16B/lane", a, b, c, sink, bytes, grid);
        run<double, 2>("1R:2W 8B/lane", a, b, c, sink, bytes, grid);
        run<double2, 2>("1R:2W 16B/lane", a, b, c, sink, bytes, grid);
    }
    return 0;
}

// HBM ceiling calibration: plain 16-B/lane copy of n bytes, several shapes.  Build:
//   hipcc --offload-arch=gfx950 -O3 -o tools/dev/copy_bench tools/ceilings/copy_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NT, int UNROLL>
__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ a, u32x4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    u32x4 v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) if (i + k * 256 < n) v[k] = NT ? __builtin_nontemporal_load(a + i + k * 256) : a[i + k * 256];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) if (i + k * 256 < n) { if (NT) __builtin_nontemporal_store(v[k], b + i + k * 256); else b[i + k * 256] = v[k]; }
}
template <int NT, int UNROLL>
__global__ __launch_bounds__(256) void k_copy_persist(const u32x4* __restrict__ a, u32x4* __restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    for (size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; base < n; base += stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) if (base + k * 256 < n) v[k] = NT ? __builtin_nontemporal_load(a + base + k * 256) : a[base + k * 256];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) if (base + k * 256 < n) { if (NT) __builtin_nontemporal_store(v[k], b + base + k * 256); else b[base + k * 256] = v[k]; }
    }
}
template <typename F> float time_it(F f, int reps) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 3; ++i) f(i);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int i = 0; i < reps; ++i) f(i);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms * 1e3f / reps;
}
int main(int argc, char** argv) {
    const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)65536 * 2000;
    const int SLOTS = 8;   // rotate buffers so nothing is served from the Infinity Cache
    u32x4 *a[SLOTS], *b[SLOTS];
    for (int i = 0; i < SLOTS; ++i) { hipMalloc(&a[i], bytes); hipMalloc(&b[i], bytes); hipMemset(a[i], 1, bytes); hipMemset(b[i], 0, bytes); }
    const size_t n = bytes / 16;
#define RUN(NAME, KERN, GRID)                                                                        \
    { float us = time_it([&](int i) { hipLaunchKernelGGL(KERN, dim3(GRID), dim3(256), 0, 0, a[i % SLOTS], b[i % SLOTS], n); }, 40); \
      printf("%-34s %8.1f us  %6.2f TB/s (r+w)\n", NAME, us, 2.0 * bytes / us / 1e6); }
    RUN("copy u1", (k_copy<0, 1>), (n + 255) / 256);
    RUN("copy u4", (k_copy<0, 4>), (n + 1023) / 1024);
    RUN("copy u8", (k_copy<0, 8>), (n + 2047) / 2048);
    RUN("copy nt u4", (k_copy<1, 4>), (n + 1023) / 1024);
    RUN("copy nt u8", (k_copy<1, 8>), (n + 2047) / 2048);
    RUN("persist u4 grid 2048", (k_copy_persist<0, 4>), 2048);
    RUN("persist u4 grid 4096", (k_copy_persist<0, 4>), 4096);
    RUN("persist nt u4 grid 2048", (k_copy_persist<1, 4>), 2048);
    RUN("persist nt u8 grid 2048", (k_copy_persist<1, 8>), 2048);
    return 0;
}

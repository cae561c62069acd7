// Dev micro-benchmark: HBM READ rate of plain streaming kernels (16 bytes per lane), by buffer size (the 256 MB Infinity
// Cache serves repeated passes over smaller buffers) and workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ in, size_t n4, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const uint4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n4; i += stride) { const uint4 a = in[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    uint32_t* sink; hipMalloc(&sink, 4);
    for (size_t mb : {128, 512, 2048}) {
        const size_t n = mb << 20;
        uint4* d; hipMalloc(&d, n); hipMemset(d, 1, n);
        for (int wg_per_cu : {4, 8, 16, 32}) {
            auto launch = [&] { hipLaunchKernelGGL(k_read, dim3(256 * wg_per_cu), dim3(256), 0, 0, d, n / 16, sink); };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%5zu MB, %2d workgroups per CU: %.2f TB/s\n", mb, wg_per_cu, n * 10.0 / (ms * 1e-3) / 1e12);
        }
        hipFree(d);
    }
    return 0;
}

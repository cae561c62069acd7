// Dev micro-benchmark: HBM write rate by store width and by the K3 pattern (a wave writes 256-byte pieces of 64 rows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ void k_dword(uint32_t* out, size_t n) {            // contiguous, 4 B per lane
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (uint32_t)i;
}
__global__ void k_dwordx4(uint4* out, size_t n4) {            // contiguous, 16 B per lane
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = make_uint4(i, i, i, i);
}
// K3 pattern: workgroup = 64 rows (envs) x N columns; wave w takes 64-column groups g = w, w + 8, ...; per group 64
// stores of 256 B, one per row
__global__ __launch_bounds__(512) void k_rows256(uint32_t* out, int N) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t* base = out + (size_t)blockIdx.x * 64 * N;
    for (int g = w; g < N / 64; g += 8)
#pragma unroll 8
        for (int e = 0; e < 64; ++e) base[(size_t)e * N + g * 64 + lane] = e + lane;
}
// same bytes, but a wave covers 4 rows x 256 B... per store 16 B per lane: lanes 0-15 row e, 16-31 row e+1, ...
__global__ __launch_bounds__(512) void k_rows256x4(uint32_t* out, int N) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t* base = out + (size_t)blockIdx.x * 64 * N;
    for (int g = w; g < N / 64; g += 8)
#pragma unroll 8
        for (int e = 0; e < 64; e += 4)
            *reinterpret_cast<uint4*>(base + (size_t)(e + (lane >> 4)) * N + g * 64 + (lane & 15) * 4) = make_uint4(e, lane, e, lane);
}

// round 6: a wave writes, per env, 256 COLUMNS as one 1 KB store of 16 B per lane -- the row-staged K3: blocks of
// 256 columns b = w, w + 8, ...; 64 stores of 1 KB per block, one per row (9 lines touched, 2 partial when the row starts mid-line)
__global__ __launch_bounds__(512) void k_rows1k(uint32_t* out, int N) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t* base = out + (size_t)blockIdx.x * 64 * N;
    for (int b = w; b * 256 < N; b += 8)
#pragma unroll 8
        for (int e = 0; e < 64; ++e)
            if (b * 256 + lane * 4 < N) *reinterpret_cast<uint4*>(base + (size_t)e * N + b * 256 + lane * 4) = make_uint4(e, lane, e, lane);
}
// ... and the tile's [64, N] output as ONE contiguous run cut on 128-byte lines: piece p of the run = 1 KB = 8 whole lines, wave w
// takes pieces p = w, w + 8, ... (what a tile-wide staging of the counts -- 8N bytes of LDS -- would allow; the run starts on a line
// when 64 * 4N is a multiple of 128, i.e. always)
__global__ __launch_bounds__(512) void k_tile_run(uint32_t* out, int N) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint4* base = reinterpret_cast<uint4*>(out + (size_t)blockIdx.x * 64 * N);
    const int pieces = 64 * N / 256;                               // 1 KB pieces of the tile's run
    for (int p = w; p < pieces; p += 8) base[(size_t)p * 64 + lane] = make_uint4(p, lane, p, lane);
}

// a wave writes RUN contiguous KB (16 B per lane, 1 KB per instruction), then the next run somewhere else:
// run r of wave w lives at ((r * nwaves + w) * RUN) KB  (S1 writes 8 KB rows this way)
template <int RUN>
__global__ __launch_bounds__(256) void k_runs(uint4* out, size_t n16) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    const size_t run16 = (size_t)RUN * 64;                       // 16-byte units per run
    for (size_t r = wave; (r + 1) * run16 <= n16; r += nwaves) {
        uint4* p = out + r * run16 + lane;
#pragma unroll
        for (int i = 0; i < RUN; ++i) p[i * 64] = make_uint4(i, lane, i, lane);
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 2048;   // row length in dwords
    printf("row length %d dwords (%d bytes), 65536 rows\n", N, N * 4);
    const int B = 65536;
    const size_t n = (size_t)B * N;
    uint32_t* d;
    hipMalloc(&d, n * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %.2f TB/s\n", name, n * 4.0 * 10 / (ms * 1e-3) / 1e12);
    };
    run("contiguous dword", [&] { hipLaunchKernelGGL(k_dword, dim3(8192), dim3(256), 0, 0, d, n); });
    run("contiguous dwordx4", [&] { hipLaunchKernelGGL(k_dwordx4, dim3(8192), dim3(256), 0, 0, (uint4*)d, n / 4); });
    run("runs of 1 KB per wave", [&] { hipLaunchKernelGGL(k_runs<1>, dim3(4096), dim3(256), 0, 0, (uint4*)d, n / 4); });
    run("runs of 8 KB per wave", [&] { hipLaunchKernelGGL(k_runs<8>, dim3(4096), dim3(256), 0, 0, (uint4*)d, n / 4); });
    run("runs of 32 KB per wave", [&] { hipLaunchKernelGGL(k_runs<32>, dim3(4096), dim3(256), 0, 0, (uint4*)d, n / 4); });
    run("64 rows x 256 B, dword", [&] { hipLaunchKernelGGL(k_rows256, dim3(B / 64), dim3(512), 0, 0, d, N); });
    run("4 rows x 256 B, dwordx4", [&] { hipLaunchKernelGGL(k_rows256x4, dim3(B / 64), dim3(512), 0, 0, d, N); });
    run("64 rows x 1 KB, dwordx4", [&] { hipLaunchKernelGGL(k_rows1k, dim3(B / 64), dim3(512), 0, 0, d, N); });
    run("tile run, 1 KB on lines", [&] { hipLaunchKernelGGL(k_tile_run, dim3(B / 64), dim3(512), 0, 0, d, N); });
    return 0;
}

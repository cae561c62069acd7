// K11: dense-QUBO coordinate local search + value (MCPG/sampling.py:323-370).
//
// Functional coverage of the reference's QUBO sampler, not a roofline kernel: Q is a dense shared
// [n, n] f32 matrix and the sweep is Gauss-Seidel (variable i sees the already-updated variables
// j < i), so the work is n^2 multiply-adds per chain per sweep with a sequential dependency over i.
// 64 chains per wave as a bit tile in LDS; row i of Q is staged in LDS (one coalesced read shared by
// the 64 chains), every lane accumulates its own chain's dot product.  Exact w.r.t. the reference
// whenever Q is integer-valued with |partial sums| < 2^24 (nbiq instances: entries +-[10, 100]):
// then f32 sums are order independent.  A GEMM-shaped (MFMA) formulation would need a different
// algorithm (block Jacobi) and is out of the stated roofline.
#include "rls_tile.h"

namespace rls {

template <bool BIN>
__global__ __launch_bounds__(kWave) void k_qubo_ls_value(const float* __restrict__ Q, int64_t n,
                                                         const float* __restrict__ xs_in, float* __restrict__ xs_out,
                                                         int64_t C, int64_t num_ls, float* __restrict__ value) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    float* qrow = reinterpret_cast<float*>(words + n);
    const uint32_t* w32 = reinterpret_cast<const uint32_t*>(smem);
    const int lane = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * kWave;
    const int64_t c = c0 + lane;
    const bool valid = c < C;
    const int half = lane >> 5, sh = lane & 31;
    for (int64_t n0 = 0; n0 < n; n0 += kWave) {
        const int lim = (int)((n - n0) < kWave ? (n - n0) : kWave);
        uint64_t mine = 0;
        for (int k = 0; k < lim; ++k) {
            const float v = valid ? xs_in[(n0 + k) * C + c] : 0.0f;
            const uint64_t w = ballot64(v > 0.0f);
            if (lane == k) mine = w;
        }
        if (lane < lim) words[n0 + lane] = mine;
    }
    __syncthreads();
    auto spin = [&](int64_t j) -> float {
        const uint32_t b = (w32[(j << 1) + half] >> sh) & 1u;
        return BIN ? (float)b : (b ? 1.0f : -1.0f);
    };
    auto dot_row = [&](int64_t i, bool skip_diag) -> float {
        for (int64_t j = lane; j < n; j += kWave) qrow[j] = Q[i * n + j];
        __syncthreads();
        float acc = 0.0f;
#pragma unroll 4
        for (int64_t j = 0; j < i; ++j) acc += qrow[j] * spin(j);
        if (!skip_diag) acc += qrow[i] * spin(i);
#pragma unroll 4
        for (int64_t j = i + 1; j < n; ++j) acc += qrow[j] * spin(j);
        return acc;
    };
    for (int64_t cnt = 0; cnt < num_ls; ++cnt) {
        for (int64_t i = 0; i < n; ++i) {
            const float res = dot_row(i, true);                  // samples[index] = 0; Q[index] @ samples
            const float thr = BIN ? (-qrow[i] / 2.0f) : 0.0f;    // res > -Q_ii / 2   |   res > 0
            const uint64_t nw = ballot64(res > thr);
            __syncthreads();
            if (lane == 0) words[i] = nw;
            __syncthreads();
        }
    }
    float total = 0.0f;                                          // sum_i s_i (Q s)_i
    for (int64_t i = 0; i < n; ++i) {
        const float r = dot_row(i, false);
        total += spin(i) * r;
        __syncthreads();
    }
    if (valid) {
        value[c] = total;
        for (int64_t j = 0; j < n; ++j) xs_out[j * C + c] = (float)((w32[(j << 1) + half] >> sh) & 1u);
    }
}

}  // namespace rls

using namespace rls;

extern "C" int rls_qubo_local_search_value(const float* Q, int64_t n, const float* xs_in, float* xs_out, int64_t C,
                                           int64_t num_ls, int binary, float* value, void* stream) {
    RLS_REQUIRE(n > 0 && C >= 0 && num_ls >= 0, RLS_EINVAL, "bad sizes n=%lld C=%lld", (long long)n, (long long)C);
    if (C == 0) return RLS_OK;
    RLS_REQUIRE(Q && xs_in && xs_out && value, RLS_EINVAL, "NULL pointer");
    const size_t lds = (size_t)n * 12 + 16;
    RLS_REQUIRE(lds <= (size_t)kLdsBytes, RLS_EUNSUPPORTED, "n=%lld needs %zu B of LDS (max %d)", (long long)n, lds,
                kLdsBytes);
    const dim3 grid((unsigned)ceil_div(C, kWave)), block(kWave);
    if (binary) {
        auto kern = k_qubo_ls_value<true>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, block, lds, as_stream(stream), Q, n, xs_in, xs_out, C, num_ls, value);
    } else {
        auto kern = k_qubo_ls_value<false>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, block, lds, as_stream(stream), Q, n, xs_in, xs_out, C, num_ls, value);
    }
    return check_launch("k_qubo_ls_value");
}

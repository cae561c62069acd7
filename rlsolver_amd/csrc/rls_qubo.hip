// K11: dense-QUBO coordinate local search + value (MCPG/sampling.py:323-370), and its sparse (CSR) form.
//
// The reference sweeps the variables one at a time, each step a dense row dot product  Q[i, :] . s  (s_i := 0) over
// the CURRENT spins: n^2 multiply-adds per chain per sweep with a sequential dependency over i.  Only one term of
// step i + 1 depends on step i, though -- so the sweep is done in BLOCKS of W variables (block Gauss-Seidel, the
// same results): the W waves of a workgroup each take one row of the block and compute its dot product against the
// spins as they stand at the block's start (the n-long part, all waves in parallel, Q rows read once each,
// coalesced), then one wave resolves the block in order with the W x W in-block corrections
//     res_k += sum_{j < k in block} Q[i0+k, i0+j] * (s_j_new - s_j_old)
// -- two barriers per W variables instead of two per variable, and W times the arithmetic in flight.
//
// 64 chains per workgroup as a bit tile in LDS (words[j] bit c = variable j of chain c0 + c).  A wave turns each
// 64-variable chunk of the tile into "lane = chain" form with one 64 x 64 bit transpose (rls_tile.h) and then spends
// 3 VALU instructions per (variable, chain-lane): bit extract, convert, multiply-add with the row entry broadcast by
// v_readlane.  Sums run in a different order than torch.mv's: exact w.r.t. the reference whenever Q is integer-valued
// with |partial sums| < 2^24 (nbiq instances: entries +-[10, 100]), as before.
//
// Roofline: this IS a dense contraction ([n, n] x [n, C] per sweep); on the f32 VALU it is bounded by
// 3 instructions per multiply-add, i.e. ~1/3 of the 157 TFLOP/s vector peak.  An MFMA formulation would need the
// spins expanded to f32 fragments per step and is outside the stated (sparse, HBM-bound) roofline -- SURVEY.md 8d.
#include "rls_tile.h"

namespace rls {

constexpr int kQuboWaves = 8;

// sum over j of Q[i, j] * b_j for the lane's chain (b = the tile's bits), j != skip; also the plain row sum.
// `qrow` points at Q[i, 0]; the row is read in 64-entry chunks, one coalesced load per chunk, next chunk in flight.
__device__ __forceinline__ void qubo_row_dot(const float* __restrict__ qrow, int64_t n, int64_t skip,
                                             const uint64_t* __restrict__ words, int lane, const BitXpose& xc,
                                             float& ones_dot, float& row_sum) {
    float acc = 0.0f, rs = 0.0f;
    const int64_t nchunk = (n + 63) >> 6;
    int64_t j = lane;
    float qn = (j < n && j != skip) ? qrow[j] : 0.0f;
    for (int64_t c = 0; c < nchunk; ++c) {
        const float qv = qn;
        const int64_t jn = ((c + 1) << 6) + lane;
        qn = (c + 1 < nchunk && jn < n && jn != skip) ? qrow[jn] : 0.0f;
        const int64_t jw = (c << 6) + lane;                      // lane l holds the word of variable 64c + l (bits = chains) ...
        const uint64_t wd = jw < n ? words[jw] : 0ull;
        uint32_t r0 = (uint32_t)wd, r1 = (uint32_t)(wd >> 32);
        bit_transpose64(r0, r1, xc);                             // ... now lane p holds chain p: bit jj = variable 64c + jj
        rs += qv;
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) {
            const float q = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qv), jj));
            acc += q * (float)((r0 >> jj) & 1u);
        }
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) {
            const float q = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qv), 32 + jj));
            acc += q * (float)((r1 >> jj) & 1u);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) rs += __shfl_xor(rs, m, 64);
    ones_dot = acc;
    row_sum = rs;
}

template <bool BIN>
__global__ __launch_bounds__(kQuboWaves * kWave) void k_qubo_ls_value(const float* __restrict__ Q, int64_t n,
                                                                      const float* __restrict__ xs_in, float* __restrict__ xs_out,
                                                                      int64_t C, int64_t num_ls, float* __restrict__ value) {
    constexpr int W = kQuboWaves;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    float* part = reinterpret_cast<float*>(words + n);           // [W][64] partial results of a block
    uint64_t* neww = reinterpret_cast<uint64_t*>(part + W * kWave);   // [W] the block's new words
    const uint32_t* w32 = reinterpret_cast<const uint32_t*>(smem);
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t c0 = (int64_t)blockIdx.x * kWave;
    const int64_t c = c0 + lane;
    const bool valid = c < C;
    const int half = lane >> 5, sh = lane & 31;
    const BitXpose xc = bit_xpose_consts(lane);
    // tile load: node-major f32 rows, one ballot per variable
    for (int64_t n0 = (int64_t)w * kWave; n0 < n; n0 += (int64_t)W * kWave) {
        const int lim = (int)((n - n0) < kWave ? (n - n0) : kWave);
        uint64_t mine = 0;
        for (int k = 0; k < lim; ++k) {
            const float v = valid ? xs_in[(n0 + k) * C + c] : 0.0f;
            const uint64_t wd = ballot64(v > 0.0f);
            if (lane == k) mine = wd;
        }
        if (lane < lim) words[n0 + lane] = mine;
    }
    __syncthreads();
    // dot of row i with the current spins, s_i := 0:  +-1 spins: 2 * sum_{b_j = 1} Q_ij - sum_j Q_ij;  0/1: sum_{b_j = 1} Q_ij
    auto row_result = [&](int64_t i, bool skip_diag) -> float {
        float od, rs;
        qubo_row_dot(Q + i * n, n, skip_diag ? i : -1, words, lane, xc, od, rs);
        return BIN ? od : (2.0f * od - rs);
    };
    for (int64_t cnt = 0; cnt < num_ls; ++cnt) {
        for (int64_t i0 = 0; i0 < n; i0 += W) {
            const int64_t i = i0 + w;
            if (i < n) part[w * kWave + lane] = row_result(i, true);
            __syncthreads();
            if (w == 0) {   // resolve the block in order: variable i0 + k sees the new values of i0 .. i0 + k - 1
                const int kb = (int)((n - i0) < W ? (n - i0) : W);
                // the block's W x W corner of Q and its diagonal, one entry per lane
                const int rk = lane / W, cj = lane % W;
                const float qc = (rk < kb && cj < kb) ? Q[(i0 + rk) * n + i0 + cj] : 0.0f;
                float delta[W];                                   // s_new - s_old of the block's earlier variables (this chain)
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    if (k < kb) {
                        float res = part[k * kWave + lane];
#pragma unroll
                        for (int j = 0; j < W; ++j)
                            if (j < k) res += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qc), k * W + j)) * delta[j];
                        const float qii = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qc), k * W + k));
                        const float thr = BIN ? (-qii / 2.0f) : 0.0f;          // res > -Q_ii / 2   |   res > 0
                        const bool nb = res > thr;
                        const uint32_t ob = (w32[((i0 + k) << 1) + half] >> sh) & 1u;
                        delta[k] = ((float)nb - (float)ob) * (BIN ? 1.0f : 2.0f);
                        const uint64_t nw = ballot64(nb);
                        if (lane == 0) neww[k] = nw;              // the tile itself keeps the OLD bits until the block is resolved
                    } else {
                        delta[k] = 0.0f;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                if (lane < kb) words[i0 + lane] = neww[lane];
            }
            __syncthreads();
        }
    }
    // value[c] = sum_i s_i (Q s)_i: the waves split the rows
    float total = 0.0f;
    for (int64_t i = w; i < n; i += W) {
        const float r = row_result(i, false);
        const uint32_t b = (w32[(i << 1) + half] >> sh) & 1u;
        total += (BIN ? (float)b : (b ? 1.0f : -1.0f)) * r;
    }
    __syncthreads();
    part[w * kWave + lane] = total;
    __syncthreads();
    if (w == 0 && valid) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < W; ++k) t += part[k * kWave + lane];
        value[c] = t;
    }
    if (valid)
        for (int64_t j = w; j < n; j += W) xs_out[j * C + c] = (float)((w32[(j << 1) + half] >> sh) & 1u);
}

// Sparse QUBO: the same coordinate search and value on Q in CSR form (rowptr / col / val,
// the diagonal included as ordinary entries).  lane = chain; a wave walks its row's entries through broadcast reads,
// one workgroup (W waves) per 64 chains, the variables strictly in order: step i costs O(nnz_i), not O(n).
template <bool BIN>
__global__ __launch_bounds__(kWave) void k_qubo_sparse_ls_value(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                const float* __restrict__ val, int64_t n,
                                                                const float* __restrict__ xs_in, float* __restrict__ xs_out,
                                                                int64_t C, int64_t num_ls, float* __restrict__ value) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
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
            const uint64_t wd = ballot64(v > 0.0f);
            if (lane == k) mine = wd;
        }
        if (lane < lim) words[n0 + lane] = mine;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    auto spin = [&](int64_t j) -> float {
        const uint32_t b = (w32[(j << 1) + half] >> sh) & 1u;
        return BIN ? (float)b : (b ? 1.0f : -1.0f);
    };
    auto row = [&](int64_t i, bool skip_diag, float& diag) -> float {
        const int r0 = rowptr[i], r1 = rowptr[i + 1];
        float acc = 0.0f;
        diag = 0.0f;
        for (int base = r0; base < r1; base += kWave) {
            const int here = (r1 - base) < kWave ? (r1 - base) : kWave;
            const int mc = lane < here ? col[base + lane] : 0;
            const float mv = lane < here ? val[base + lane] : 0.0f;
            for (int k = 0; k < here; ++k) {
                const int j = __builtin_amdgcn_readlane(mc, k);
                const float q = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mv), k));
                if (j == i) {
                    diag += q;
                    if (skip_diag) continue;
                }
                acc += q * spin(j);
            }
        }
        return acc;
    };
    for (int64_t cnt = 0; cnt < num_ls; ++cnt)
        for (int64_t i = 0; i < n; ++i) {
            float qii;
            const float res = row(i, true, qii);
            const uint64_t nw = ballot64(res > (BIN ? (-qii / 2.0f) : 0.0f));
            if (lane == 0) words[i] = nw;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    float total = 0.0f, qii;
    for (int64_t i = 0; i < n; ++i) total += spin(i) * row(i, false, qii);
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
    const size_t lds = (size_t)n * 8 + (size_t)kQuboWaves * kWave * 4 + (size_t)kQuboWaves * 8;
    RLS_REQUIRE(lds <= (size_t)kLdsBytes, RLS_EUNSUPPORTED, "n=%lld needs %zu B of LDS (max %d)", (long long)n, lds,
                kLdsBytes);
    const dim3 grid((unsigned)ceil_div(C, kWave)), block(kQuboWaves * kWave);
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

extern "C" int rls_qubo_sparse_local_search_value(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n,
                                                  const float* xs_in, float* xs_out, int64_t C, int64_t num_ls, int binary,
                                                  float* value, void* stream) {
    RLS_REQUIRE(n > 0 && C >= 0 && num_ls >= 0, RLS_EINVAL, "bad sizes n=%lld C=%lld", (long long)n, (long long)C);
    if (C == 0) return RLS_OK;
    RLS_REQUIRE(rowptr && col && val && xs_in && xs_out && value, RLS_EINVAL, "NULL pointer");
    const size_t lds = (size_t)n * 8 + 16;
    RLS_REQUIRE(lds <= (size_t)kLdsBytes, RLS_EUNSUPPORTED, "n=%lld needs %zu B of LDS (max %d)", (long long)n, lds, kLdsBytes);
    const dim3 grid((unsigned)ceil_div(C, kWave)), block(kWave);
    if (binary) {
        auto kern = k_qubo_sparse_ls_value<true>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, block, lds, as_stream(stream), rowptr, col, val, n, xs_in, xs_out, C, num_ls, value);
    } else {
        auto kern = k_qubo_sparse_ls_value<false>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, block, lds, as_stream(stream), rowptr, col, val, n, xs_in, xs_out, C, num_ls, value);
    }
    return check_launch("k_qubo_sparse_ls_value");
}

// K11: dense-QUBO coordinate local search + value (MCPG/sampling.py:323-370), and its sparse (CSR) form.
//
// The reference sweeps the variables one at a time, each step a dense row dot product  Q[i, :] . s  (s_i := 0) over
// the CURRENT spins: n^2 multiply-adds per chain per sweep with a sequential dependency over i.  Only one term of
// step i + 1 depends on step i, though -- so the sweep is done in BLOCKS of 32 variables (block Gauss-Seidel, the
// same results): the n-long part of a block's products against the spins as they stand at the block's start IS a GEMM,
//     D[32 variables, chains] = Q[i0 .. i0+31, :] (32 x n)  x  S (n x chains, entries +-1 or 0/1),
// and the block is then resolved in order with the in-block corrections
//     res_k += sum_{j < k in block} Q[i0+k, i0+j] * (s_j_new - s_j_old).
//
// Both parts run on the matrix cores as v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: the arithmetic of the
// reference's f32 matmul; sums run in a different order than torch.mv's, exact w.r.t. the reference whenever Q is
// integer-valued with |partial sums| < 2^24 -- nbiq instances: entries +-[10, 100]).
//   * A workgroup owns NT tiles of 32 chains; its W waves split the n columns, each streaming its slice of the block's
//     32 rows from L2 (64 contiguous bytes per lane and batch: a full 128-byte line per row), expanding the spins from a
//     bit tile in LDS (one 32 NT-bit word per variable; 2 VALU per operand) and accumulating independent 32 x 32 tiles
//     (NT = 1: even / odd column pairs), so consecutive MFMAs never wait for each other.  The diagonal term is zeroed in
//     the A operand (s_i := 0 of sampling.py:333), not subtracted afterwards.
//   * The partial tiles meet in LDS and wave 0 resolves the block: the residual tile R stays in accumulator layout,
//     step k reads row k, decides, and applies the rank-1 update  R[m, c] += Q[i0+m, i0+k] * delta_k[c]  as ONE more
//     MFMA (A = column k of the block's 32 x 32 corner of Q, held one row per lane; B = delta in one k-slot, zero in the
//     other) -- 32 dependent MFMAs and ~10 VALU per step, where broadcast multiply-adds cost k per step plus a scalar
//     load, LDS read or SGPR each.
// MFMA operand layout (32x32x2): lane = (r = lane % 32, h = lane / 32); A[r][h], B[h][r]; D register v of lane (c, h) is
// row 8 (v / 4) + 4 h + v % 4, column c.  A batch is 32 columns: half h takes the 16 contiguous columns kb + 16 h + e and
// MFMA e pairs column kb + e (h = 0) with kb + 16 + e (h = 1) -- any pairing works as long as A and B agree.
//
// Roofline: a dense contraction, 2 n^2 flops per chain per sweep (+ one more for the value) against the f32 matrix
// peak (157 TFLOP/s); Q is re-read from L2 by every workgroup, 4 n^2 bytes per (workgroup, sweep), which is what NT = 2
// halves once there are enough chains to fill the chip with 64-chain workgroups.
#include "rls_tile.h"

namespace rls {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

template <int NT> struct QmWord;
template <> struct QmWord<1> { using type = uint32_t; };
template <> struct QmWord<2> { using type = uint64_t; };
template <int NT> constexpr int qm_part_stride() { return 32 * NT + 8; }   // the lane halves (rows r, r + 4) hit disjoint banks

template <bool BIN>
__device__ __forceinline__ float qm_spin(uint32_t word, int c) {
    const uint32_t t = word << (31 - c);                        // bit c -> sign position
    if (BIN) return __builtin_bit_cast(float, (uint32_t)((int32_t)t >> 31) & 0x3f800000u);      // 1.0 or 0.0
    return __builtin_bit_cast(float, (~t & 0x80000000u) | 0x3f800000u);                          // +1.0 or -1.0
}

// acc[.] += Q[i0 + r, k] * s[k, chains] over the batches [kb0, kb1) (multiples of 32) of this wave.
// NT = 1: acc[0] / acc[1] take the even / odd MFMAs of the one chain tile; NT = 2: acc[t] is chain tile t.
template <bool BIN, int NT, bool MASK_DIAG>
__device__ __forceinline__ void qm_block_dot(const float* __restrict__ Q, int64_t n, int64_t i0, int64_t kb0, int64_t kb1,
                                             const typename QmWord<NT>::type* __restrict__ words, int r, int h,
                                             f32x16 (&acc)[2]) {
    const int64_t row = i0 + r;
    const bool row_ok = row < n;
    const float* qrow = Q + (row_ok ? row : 0) * n;
    auto load_a = [&](int64_t kb, float (&a)[16]) {
        const int64_t kk = kb + 16 * h;
        if (row_ok && kk + 16 <= n) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4u v = *reinterpret_cast<const f32x4u*>(qrow + kk + 4 * q4);
                a[4 * q4] = v.x; a[4 * q4 + 1] = v.y; a[4 * q4 + 2] = v.z; a[4 * q4 + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) a[e] = (row_ok && kk + e < n) ? qrow[kk + e] : 0.0f;
        }
    };
    float a[16], an[16];
    if (kb0 < kb1) load_a(kb0, a);
    for (int64_t kb = kb0; kb < kb1; kb += 32) {
        if (kb + 32 < kb1) load_a(kb + 32, an);
        const int64_t kk = kb + 16 * h;
        typename QmWord<NT>::type wd[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) wd[e] = words[kk + e];     // 16-byte LDS reads, one address per lane half
        if (MASK_DIAG && kb < i0 + 32 && kb + 32 > i0) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (kk + e == row) a[e] = 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if constexpr (NT == 1) {
                acc[e & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], qm_spin<BIN>(wd[e], r), acc[e & 1], 0, 0, 0);
            } else {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], qm_spin<BIN>((uint32_t)wd[e], r), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], qm_spin<BIN>((uint32_t)(wd[e] >> 32), r), acc[1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = an[e];
    }
}

template <bool BIN, int NT, int W>
__global__ __launch_bounds__(W * kWave) void k_qubo_ls_value_mfma(const float* __restrict__ Q, int64_t n, int64_t n_pad,
                                                                  const float* __restrict__ xs_in, float* __restrict__ xs_out,
                                                                  int64_t C, int64_t num_ls, float* __restrict__ value) {
    using Word = typename QmWord<NT>::type;
    constexpr int PS = qm_part_stride<NT>();
    constexpr int CH = 32 * NT;                                  // chains per workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Word* words = reinterpret_cast<Word*>(smem);                                 // [n_pad] bit c = chain c0 + c
    float* part = reinterpret_cast<float*>(words + n_pad);                       // [W][32][PS]
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int r = lane & 31, h = lane >> 5;                      // r doubles as the chain column of B / D
    const int64_t c0 = (int64_t)blockIdx.x * CH;
    // tile load from the node-major f32 surface: NT = 2: one variable (64 chains) per ballot; NT = 1: two variables
    if constexpr (NT == 2) {
        const bool valid = c0 + lane < C;
        for (int64_t j = w; j < n_pad; j += W) {
            const float v = (valid && j < n) ? xs_in[j * C + c0 + lane] : 0.0f;
            const uint64_t b = ballot64(v > 0.0f);
            if (lane == 0) words[j] = b;
        }
    } else {
        const bool valid = c0 + r < C;
        for (int64_t j0 = (int64_t)w * 2; j0 < n_pad; j0 += (int64_t)W * 2) {
            const int64_t j = j0 + h;
            const float v = (valid && j < n) ? xs_in[j * C + c0 + r] : 0.0f;
            const uint64_t b = ballot64(v > 0.0f);
            if (lane == 0) { words[j0] = (uint32_t)b; words[j0 + 1] = (uint32_t)(b >> 32); }
        }
    }
    __syncthreads();
    const int64_t kper = n_pad / W;                              // a multiple of 32
    const int64_t kb0 = (int64_t)w * kper, kb1 = kb0 + kper;
    float* mypart = part + (size_t)w * 32 * PS;
    auto tile_word = [&](Word wd, int t) -> uint32_t { return NT == 1 ? (uint32_t)wd : (uint32_t)((uint64_t)wd >> (32 * t)); };
    for (int64_t cnt = 0; cnt < num_ls; ++cnt) {
        for (int64_t i0 = 0; i0 < n; i0 += 32) {
            // wave 0 fetches its rows of the block's 32 x 32 corner of Q first (lane (r, .) holds Q[i0 + r, i0 .. i0 + 31]):
            // the loads are long done when the resolution below uses column k as the A operand of step k
            float qc[32];
            if (w == 0) {
                const int64_t row = i0 + r;
                if (row < n && i0 + 32 <= n) {
#pragma unroll
                    for (int q4 = 0; q4 < 8; ++q4) {
                        const f32x4u v = *reinterpret_cast<const f32x4u*>(Q + row * n + i0 + 4 * q4);
                        qc[4 * q4] = v.x; qc[4 * q4 + 1] = v.y; qc[4 * q4 + 2] = v.z; qc[4 * q4 + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 32; ++k) qc[k] = (row < n && i0 + k < n) ? Q[row * n + i0 + k] : 0.0f;
                }
            }
            f32x16 acc[2] = {{0}, {0}};
            qm_block_dot<BIN, NT, true>(Q, n, i0, kb0, kb1, words, r, h, acc);
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float* dst = mypart + (8 * (v / 4) + 4 * h + (v % 4)) * PS + r;
                if constexpr (NT == 1) dst[0] = acc[0][v] + acc[1][v];
                else { dst[0] = acc[0][v]; dst[32] = acc[1][v]; }
            }
            __syncthreads();
            if (w == 0) {
                // Resolve the block in order, variable i0 + k seeing the new values of i0 .. i0 + k - 1.  Step k reads row k
                // of R (lanes of half hk = (k / 4) % 2, register 4 (k / 8) + k % 4), decides, and updates R by one MFMA
                // per chain tile.  Rows <= k collect updates nobody reads.
                const int kbk = (int)((n - i0) < 32 ? (n - i0) : 32);
                f32x16 R[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        float sum = 0.0f;
#pragma unroll
                        for (int ww = 0; ww < W; ++ww) sum += part[(size_t)(ww * 32 + 8 * (v / 4) + 4 * h + (v % 4)) * PS + 32 * t + r];
                        R[t][v] = sum;
                    }
                Word mine = 0;                                   // lane k ends up with the new word of variable i0 + k
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    const int hk = (k / 4) & 1;
                    const float thr = BIN ? (-qc[k] / 2.0f) : 0.0f;               // lane r = k holds Q_kk
                    const float thr_k = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, thr), k));
                    const Word old = words[i0 + k];              // the tile keeps the OLD bits until the block is resolved
                    Word nw = 0;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const bool nb = R[t][4 * (k / 8) + (k % 4)] > thr_k;      // res > -Q_ii / 2   |   res > 0
                        const uint64_t bal = ballot64(nb);
                        nw |= (Word)(hk ? (uint32_t)(bal >> 32) : (uint32_t)bal) << (NT == 1 ? 0 : 32 * t);
                        const uint32_t ob = (tile_word(old, t) >> r) & 1u;
                        float d = ((float)nb - (float)ob) * (BIN ? 1.0f : 2.0f);
                        if (h != hk) d = 0.0f;
                        R[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(qc[k], d, R[t], 0, 0, 0);
                    }
                    if (lane == k) mine = nw;
                }
                if (lane < kbk) words[i0 + lane] = mine;
            }
            __syncthreads();
        }
    }
    // value[c] = sum_i s_i (Q s)_i: every wave weighs its partial products with the rows' spins
    float total[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) total[t] = 0.0f;
    for (int64_t i0 = 0; i0 < n; i0 += 32) {
        f32x16 acc[2] = {{0}, {0}};
        qm_block_dot<BIN, NT, false>(Q, n, i0, kb0, kb1, words, r, h, acc);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const Word wd = words[i0 + 8 * (v / 4) + 4 * h + (v % 4)];
            if constexpr (NT == 1) total[0] += qm_spin<BIN>(wd, r) * (acc[0][v] + acc[1][v]);
            else {
                total[0] += qm_spin<BIN>((uint32_t)wd, r) * acc[0][v];
                total[1] += qm_spin<BIN>((uint32_t)(wd >> 32), r) * acc[1][v];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float both = total[t] + __shfl_xor(total[t], 32, 64);
        if (h == 0) part[(w * NT + t) * 32 + r] = both;
    }
    __syncthreads();
    if (w == 0 && lane < CH && c0 + lane < C) {
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < W; ++k) sum += part[(k * NT + (lane >> 5)) * 32 + r];
        value[c0 + lane] = sum;
    }
    if constexpr (NT == 2) {
        if (c0 + lane < C)
            for (int64_t j = w; j < n; j += W) xs_out[j * C + c0 + lane] = (float)((words[j] >> lane) & 1ull);
    } else {
        if (c0 + r < C)
            for (int64_t j = (int64_t)w * 2 + h; j < n; j += (int64_t)W * 2) xs_out[j * C + c0 + r] = (float)((words[j] >> r) & 1u);
    }
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
    int* rp = reinterpret_cast<int*>(words + n);                 // rowptr[0 .. n]
    const int lane = threadIdx.x;
    for (int64_t i = lane; i <= n; i += kWave) rp[i] = rowptr[i];
    const int64_t c0 = (int64_t)blockIdx.x * kWave;
    const int64_t c = c0 + lane;
    const bool valid = c < C;
    const int half = lane >> 5, sh = lane & 31;
    for (int64_t n0 = 0; n0 < n; n0 += kWave) {
        const int lim = (int)((n - n0) < kWave ? (n - n0) : kWave);
        uint64_t mine = 0;
        for (int k8 = 0; k8 < lim; k8 += 8) {                    // 8 rows' loads in flight, then their ballots
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (valid && k8 + u < lim) ? xs_in[(n0 + k8 + u) * C + c] : 0.0f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint64_t wd = ballot64(v[u] > 0.0f);
                if (lane == k8 + u) mine = wd;
            }
        }
        if (lane < lim) words[n0 + lane] = mine;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // Row i + 1's entries are fetched while row i is being summed (they do not depend on the chains), and a row is
    // summed 8 entries at a time: 8 broadcast lane reads, 8 LDS reads in flight, 8 multiply-adds.  One entry per trip was
    // a chain of LDS round trips behind a chain of L2 round trips per variable.
    struct Row { int r0, r1, mc; float mv; };                    // mc = LDS byte offset of the neighbour's word (this lane's half)
    const int half4 = half * 4;
    auto fetch = [&](int64_t i) -> Row {                         // i wraps: the look-ahead past the last row is harmless
        while (i >= n) i -= n;
        Row rw;
        rw.r0 = __builtin_amdgcn_readfirstlane(rp[i]);
        rw.r1 = __builtin_amdgcn_readfirstlane(rp[i + 1]);
        const bool in = lane < rw.r1 - rw.r0;
        rw.mc = in ? col[rw.r0 + lane] : 0;
        rw.mv = in ? val[rw.r0 + lane] : 0.0f;
        return rw;
    };
    auto spin_bits = [&](int j) -> uint32_t { return (w32[(j << 1) + half] >> sh) & 1u; };
    // One chunk (<= 64 entries, entry k in lane k; lanes past cnt hold (0, 0.0) and add nothing): ones += sum of the
    // values whose neighbour bit is set, all += sum of the values; the diagonal entries (col == i) go to diag and, with
    // skip_diag, out of both sums -- picked out per lane, once per chunk, so the entry loop carries no compare.
    // Per entry: 2 v_readlane, address add, ds_read, v_bfe_i32 (bit -> 0 / -1 mask), and, 2 adds.
    auto chunk_sum = [&](int mc, float mv, int cnt, int64_t i, bool skip_diag, float& ones, float& all, float& diag) {
        uint64_t dm = ballot64(lane < cnt && mc == (int)i);
        while (dm) {
            const int l = __builtin_ctzll(dm);
            diag += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mv), l));
            dm &= dm - 1;
        }
        if (skip_diag && mc == (int)i) mv = 0.0f;
        const int off = mc * 8;
        for (int k = 0; k < cnt; k += 8) {
            float q[8];
            uint32_t wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int jo = __builtin_amdgcn_readlane(off, k + u);
                q[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mv), k + u));
                wv[u] = *reinterpret_cast<const uint32_t*>(smem + (uint32_t)(jo + half4));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int m = __builtin_amdgcn_sbfe((int)wv[u], sh, 1);       // 0 or -1
                ones += __builtin_bit_cast(float, __builtin_bit_cast(int, q[u]) & m);
                all += q[u];
            }
        }
    };
    // sum_j Q_ij s_j over the row's entries (+-1: 2 * ones - all; 0/1: ones)
    auto row = [&](const Row& rw, int64_t i, bool skip_diag, float& diag) -> float {
        float ones = 0.0f, all = 0.0f;
        diag = 0.0f;
        const int deg = rw.r1 - rw.r0;
        chunk_sum(rw.mc, rw.mv, deg < kWave ? deg : kWave, i, skip_diag, ones, all, diag);
        for (int base = rw.r0 + kWave; base < rw.r1; base += kWave) {       // rows longer than a wave: the rest inline
            const int here = (rw.r1 - base) < kWave ? (rw.r1 - base) : kWave;
            const int mc = lane < here ? col[base + lane] : 0;
            const float mv = lane < here ? val[base + lane] : 0.0f;
            chunk_sum(mc, mv, here, i, skip_diag, ones, all, diag);
        }
        return BIN ? ones : (2.0f * ones - all);
    };
    // three rows of look-ahead: a row takes a fraction of the L2 round trip its entries need
    for (int64_t cnt = 0; cnt < num_ls; ++cnt) {
        Row cur = fetch(0), n1 = fetch(1), n2 = fetch(2);
        for (int64_t i = 0; i < n; ++i) {
            const Row n3 = fetch(i + 3);
            float qii;
            const float res = row(cur, i, true, qii);
            const uint64_t nw = ballot64(res > (BIN ? (-qii / 2.0f) : 0.0f));
            if (lane == 0) words[i] = nw;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            cur = n1; n1 = n2; n2 = n3;
        }
    }
    float total = 0.0f, qii;
    {
        Row cur = fetch(0), n1 = fetch(1), n2 = fetch(2);
        for (int64_t i = 0; i < n; ++i) {
            const Row n3 = fetch(i + 3);
            const uint32_t bi = spin_bits((int)i);
            total += (BIN ? (float)bi : (bi ? 1.0f : -1.0f)) * row(cur, i, false, qii);
            cur = n1; n1 = n2; n2 = n3;
        }
    }
    if (valid) {
        value[c] = total;
        for (int64_t j = 0; j < n; ++j) xs_out[j * C + c] = (float)((w32[(j << 1) + half] >> sh) & 1u);
    }
}

// Sparse QUBO by LEVELS (round 6).  The kernel above is one wave per 64 chains walking the n rows in order: 2^13 chains are
// 128 waves on a 1024-SIMD chip, 0.38 us per row -- 2.5 x slower than the dense MFMA kernel on a 2 %-filled 1000 x 1000 Q.  Row i
// needs the NEW bits of its neighbours j < i and the OLD bits of its neighbours j > i, so with level(i) = 1 + max level of the
// neighbours below i (host: MCPG_qubo.qubo_levels) the rows of one level have no entry between them and every neighbour above
// a row sits in a later level: the W waves of a 64-chain workgroup take a level's rows side by side, one barrier per level, and
// the result is the sequential sweep's bit for bit (the sums are order-independent for integer-valued Q, as everywhere in K11).
// (Measured and not kept: the schedule and row bounds staged in LDS + the next level's first row requested before the barrier --
// 487 -> 539 us at n = 1000, 2 % fill, 2^13 chains: the scalar cache already holds them, the time is the dependent VALU chain of a
// row with one wave per SIMD.)
template <bool BIN, int W>
__global__ __launch_bounds__(W * kWave) void k_qubo_sparse_levels(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                  const float* __restrict__ val, int64_t n,
                                                                  const int32_t* __restrict__ lv_ptr, const int32_t* __restrict__ lv_rows,
                                                                  int32_t L, const float* __restrict__ xs_in, float* __restrict__ xs_out,
                                                                  int64_t C, int64_t num_ls, float* __restrict__ value) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    const uint32_t* w32 = reinterpret_cast<const uint32_t*>(smem);
    float* part = reinterpret_cast<float*>(words + n);           // [W][64]
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t c0 = (int64_t)blockIdx.x * kWave;
    const int64_t c = c0 + lane;
    const bool valid = c < C;
    const int half = lane >> 5, sh = lane & 31, half4 = half * 4;
    // the tile: wave w turns rows 8 (w + W t) .. + 7 into words, 8 loads in flight
    for (int64_t j0 = (int64_t)w * 8; j0 < n; j0 += (int64_t)W * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (valid && j0 + u < n) ? xs_in[(j0 + u) * C + c] : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint64_t wd = ballot64(v[u] > 0.0f);
            if (lane == 0 && j0 + u < n) words[j0 + u] = wd;
        }
    }
    __syncthreads();
    struct Row { int i, r0, r1, mc; float mv; };
    auto fetch = [&](int i) -> Row {                             // i is wave-uniform: scalar loads of the row's bounds
        Row rw;
        rw.i = i;
        rw.r0 = rowptr[i];
        rw.r1 = rowptr[i + 1];
        const bool in = lane < rw.r1 - rw.r0;
        rw.mc = in ? col[rw.r0 + lane] : 0;
        rw.mv = in ? val[rw.r0 + lane] : 0.0f;
        return rw;
    };
    // one chunk of <= 64 entries (entry k in lane k): the sequential kernel's chunk_sum
    auto chunk_sum = [&](int mc, float mv, int cnt, int i, bool skip_diag, float& ones, float& all, float& diag) {
        uint64_t dm = ballot64(lane < cnt && mc == i);
        while (dm) {
            const int l = __builtin_ctzll(dm);
            diag += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mv), l));
            dm &= dm - 1;
        }
        if (skip_diag && mc == i) mv = 0.0f;
        const int off = mc * 8;
        for (int k = 0; k < cnt; k += 8) {
            float q[8];
            uint32_t wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int jo = __builtin_amdgcn_readlane(off, k + u);
                q[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mv), k + u));
                wv[u] = *reinterpret_cast<const uint32_t*>(smem + (uint32_t)(jo + half4));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int m = __builtin_amdgcn_sbfe((int)wv[u], sh, 1);       // 0 or -1
                ones += __builtin_bit_cast(float, __builtin_bit_cast(int, q[u]) & m);
                all += q[u];
            }
        }
    };
    auto row = [&](const Row& rw, bool skip_diag, float& diag) -> float {
        float ones = 0.0f, all = 0.0f;
        diag = 0.0f;
        const int deg = rw.r1 - rw.r0;
        chunk_sum(rw.mc, rw.mv, deg < kWave ? deg : kWave, rw.i, skip_diag, ones, all, diag);
        for (int base = rw.r0 + kWave; base < rw.r1; base += kWave) {
            const int here = (rw.r1 - base) < kWave ? (rw.r1 - base) : kWave;
            const int mc = lane < here ? col[base + lane] : 0;
            const float mv = lane < here ? val[base + lane] : 0.0f;
            chunk_sum(mc, mv, here, rw.i, skip_diag, ones, all, diag);
        }
        return BIN ? ones : (2.0f * ones - all);
    };
    for (int64_t cnt = 0; cnt < num_ls; ++cnt) {
        for (int lv = 0; lv < L; ++lv) {
            const int e0 = lv_ptr[lv], e1 = lv_ptr[lv + 1];
            int e = e0 + w;
            Row cur;
            if (e < e1) cur = fetch(lv_rows[e]);
            for (; e < e1; e += W) {
                Row nxt = cur;
                if (e + W < e1) nxt = fetch(lv_rows[e + W]);        // the wave's next row of this level, behind the current one's sums
                float qii;
                const float res = row(cur, true, qii);
                const uint64_t nw = ballot64(res > (BIN ? (-qii / 2.0f) : 0.0f));
                if (lane == 0) words[cur.i] = nw;                   // no row of this level reads word i
                cur = nxt;
            }
            __syncthreads();
        }
    }
    // value = s^T Q s: the rows dealt over the waves, the partial sums through LDS
    float total = 0.0f;
    {
        int i = w;
        Row cur;
        if (i < n) cur = fetch(i);
        for (; i < n; i += W) {
            Row nxt = cur;
            if (i + W < n) nxt = fetch(i + W);
            float qii;
            const uint32_t bi = (w32[(i << 1) + half] >> sh) & 1u;
            total += (BIN ? (float)bi : (bi ? 1.0f : -1.0f)) * row(cur, false, qii);
            cur = nxt;
        }
    }
    part[w * kWave + lane] = total;
    __syncthreads();
    if (w == 0 && valid) {
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < W; ++k) sum += part[k * kWave + lane];
        value[c] = sum;
    }
    if (valid)
        for (int64_t j = w; j < n; j += W) xs_out[j * C + c] = (float)((w32[(j << 1) + half] >> sh) & 1u);
}

}  // namespace rls

using namespace rls;

extern "C" int rls_qubo_local_search_value(const float* Q, int64_t n, const float* xs_in, float* xs_out, int64_t C,
                                           int64_t num_ls, int binary, float* value, void* stream) {
    RLS_REQUIRE(n > 0 && C >= 0 && num_ls >= 0, RLS_EINVAL, "bad sizes n=%lld C=%lld", (long long)n, (long long)C);
    if (C == 0) return RLS_OK;
    RLS_REQUIRE(Q && xs_in && xs_out && value, RLS_EINVAL, "NULL pointer");
    // 64-chain workgroups (half the L2 traffic and resolution work per chain) once they fill the chip; 8 waves (two per
    // SIMD, for the L2 latency) while a CU holds a single workgroup, 4 when several can share it
    const bool nt2 = C >= (int64_t)2 * 64 * num_cus();
    const bool w8 = !nt2 && ceil_div(C, 32) <= (int64_t)2 * num_cus();
    const int W = w8 ? 8 : 4;
    const int64_t n_pad = ceil_div(n, 32 * W) * 32 * W;
    const size_t lds = (size_t)n_pad * (nt2 ? 8 : 4) + (size_t)W * 32 * (nt2 ? qm_part_stride<2>() : qm_part_stride<1>()) * 4;
    RLS_REQUIRE(lds <= (size_t)kLdsBytes, RLS_EUNSUPPORTED, "n=%lld needs %zu B of LDS (max %d)", (long long)n, lds,
                kLdsBytes);
    const dim3 grid((unsigned)ceil_div(C, nt2 ? 64 : 32)), block(W * kWave);
#define RLS_QM_LAUNCH(BIN_, NT_, W_)                                                                                  \
    do {                                                                                                              \
        auto kern = k_qubo_ls_value_mfma<BIN_, NT_, W_>;                                                              \
        if (lds > 64 * 1024)                                                                                          \
            ensure_dyn_lds((const void*)kern, lds);       \
        hipLaunchKernelGGL(kern, grid, block, lds, as_stream(stream), Q, n, n_pad, xs_in, xs_out, C, num_ls, value);  \
    } while (0)
    if (binary) {
        if (nt2) RLS_QM_LAUNCH(true, 2, 4);
        else { if (w8) RLS_QM_LAUNCH(true, 1, 8); else RLS_QM_LAUNCH(true, 1, 4); }
    } else {
        if (nt2) RLS_QM_LAUNCH(false, 2, 4);
        else { if (w8) RLS_QM_LAUNCH(false, 1, 8); else RLS_QM_LAUNCH(false, 1, 4); }
    }
#undef RLS_QM_LAUNCH
    return check_launch("k_qubo_ls_value_mfma");
}

extern "C" int rls_qubo_sparse_local_search_value(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n,
                                                  const int32_t* lv_ptr, const int32_t* lv_rows, int32_t num_levels,
                                                  const float* xs_in, float* xs_out, int64_t C, int64_t num_ls, int binary,
                                                  float* value, void* stream) {
    RLS_REQUIRE(n > 0 && C >= 0 && num_ls >= 0, RLS_EINVAL, "bad sizes n=%lld C=%lld", (long long)n, (long long)C);
    if (C == 0) return RLS_OK;
    RLS_REQUIRE(rowptr && col && val && xs_in && xs_out && value, RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE((lv_ptr != nullptr) == (lv_rows != nullptr) && (!lv_ptr || num_levels >= 1), RLS_EINVAL,
                "lv_ptr / lv_rows / num_levels come together");
    if (lv_ptr && knob(KN_QUBO_LEVELS, 1) != 0) {
        // the level schedule: W waves per 64-chain tile -- 16 while that leaves at most two waves per SIMD (a row's sums are a
        // dependent VALU chain: a second wave fills its issue gaps), 8 up to four, then 4
        const int64_t tiles = ceil_div(C, kWave);
        const int kw = (int)knob(KN_QUBO_LEVELS, 1);
        const int Wv = kw == 4 || kw == 8 || kw == 16 ? kw : (tiles * 16 <= (int64_t)8 * num_cus() ? 16 : (tiles * 8 <= (int64_t)16 * num_cus() ? 8 : 4));
        const size_t ldl = (size_t)n * 8 + (size_t)Wv * kWave * 4;
        if (ldl <= (size_t)kLdsBytes) {
            const dim3 gl((unsigned)tiles), bl(Wv * kWave);
#define RLS_QL_LAUNCH(BIN_, W_)                                                                                                  \
    do {                                                                                                                         \
        auto kern = k_qubo_sparse_levels<BIN_, W_>;                                                                              \
        if (ldl > 64 * 1024) ensure_dyn_lds((const void*)kern, ldl);                                                             \
        hipLaunchKernelGGL(kern, gl, bl, ldl, as_stream(stream), rowptr, col, val, n, lv_ptr, lv_rows, num_levels, xs_in, xs_out, \
                           C, num_ls, value);                                                                                    \
    } while (0)
#define RLS_QL_W(BIN_) do { if (Wv == 16) RLS_QL_LAUNCH(BIN_, 16); else if (Wv == 8) RLS_QL_LAUNCH(BIN_, 8); else RLS_QL_LAUNCH(BIN_, 4); } while (0)
            if (binary) RLS_QL_W(true); else RLS_QL_W(false);
#undef RLS_QL_W
#undef RLS_QL_LAUNCH
            return check_launch("k_qubo_sparse_levels");
        }
    }
    const size_t lds = (size_t)n * 8 + (size_t)(n + 1) * 4;
    RLS_REQUIRE(lds <= (size_t)kLdsBytes, RLS_EUNSUPPORTED, "n=%lld needs %zu B of LDS (max %d)", (long long)n, lds, kLdsBytes);
    const dim3 grid((unsigned)ceil_div(C, kWave)), block(kWave);
    if (binary) {
        auto kern = k_qubo_sparse_ls_value<true>;
        if (lds > 64 * 1024) ensure_dyn_lds((const void*)kern, lds);
        hipLaunchKernelGGL(kern, grid, block, lds, as_stream(stream), rowptr, col, val, n, xs_in, xs_out, C, num_ls, value);
    } else {
        auto kern = k_qubo_sparse_ls_value<false>;
        if (lds > 64 * 1024) ensure_dyn_lds((const void*)kern, lds);
        hipLaunchKernelGGL(kern, grid, block, lds, as_stream(stream), rowptr, col, val, n, xs_in, xs_out, C, num_ls, value);
    }
    return check_launch("k_qubo_sparse_ls_value");
}

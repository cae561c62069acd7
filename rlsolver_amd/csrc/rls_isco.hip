// ISCO sampler steps as single kernels (VERDICT r1 "missing" #7):
//   rls_isco_maxcut_step : ISCO_maxcut.step   rlsolver/envs/env_ISCO.py:26-49
//                          (get_local_dist :51-63, proposal :37-49, ll_y2x :65-77)
//   rls_isco_tsp_step    : ISCO_TSP.step      rlsolver/envs/env_ISCO.py:188-236 (opt_2 :238-335, switch :337-344)
// with the sampler helpers of rlsolver/methods/util.py:498-570 (gumbel, log1mexp,
// noreplacement_sampling_renormalize, multinomial, bernoulli_logp, mh_step) folded in.  The reference runs each
// step as ~25-40 torch ops on [B, N] tensors (log_softmax, two sorts, argsort, gathers, cumsum, scatter, where);
// here one wave owns one env and keeps its row in LDS from the first read to the accepted sample.
//
// What "Gumbel top-k without replacement" needs per env is only: which L entries have the largest perturbed
// log-probability, in which order, and the running sum of their probabilities in that order -- so the two full
// sorts of the reference become one radix select (32 ballot rounds) + a bitonic sort of the L selected entries.
//
// Floating point: f32 like the reference, libm-accurate expf / logf / log1pf / expm1f; sums over a row are wave
// tree reductions / scans, not torch's order: results agree with the reference to ~1e-6 relative (tests: 1e-5,
// the tolerance north_star states for floating-point results).
#include "rls_tile.h"
#include "rls_draw.h"
#include <cmath>
#include <cstdlib>

namespace rls {

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ float wave_sum_f32x(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ void lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// monotone map float -> uint32 (larger float <=> larger key)
__device__ __forceinline__ uint32_t fkey(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// log(1 - exp(-|x|))   methods/util.py:502-505
__device__ __forceinline__ float log1mexp(float x) {
    x = -fabsf(x);
    return x > -0.693f ? logf(-expm1f(x)) : log1pf(-expf(x));
}

// sum over k of  min(ll_k - log1mexp(log(cum_k - p_k) + base), 0)   with p_k = exp(ll_k - base), cum = inclusive
// running sum of p in list order  (noreplacement_sampling_renormalize, methods/util.py:507-512, restricted to
// the entries that are summed afterwards).  ll(k) gives the k-th log-probability of the ordered list.
template <typename F>
__device__ __forceinline__ float noreplacement_ll_sum(int count, float base, int lane, F ll) {
    float carry = 0.0f, total = 0.0f;
    for (int k0 = 0; k0 < count; k0 += kWave) {
        const int k = k0 + lane;
        const bool in = k < count;
        const float l = in ? ll(k) : 0.0f;
        const float p = in ? expf(l - base) : 0.0f;
        float inc = p;                                   // inclusive scan across the wave
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const float o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        const float cum = carry + inc;
        const float ll_delta = logf(cum - p) + base;     // log(0) = -inf for the first entry -> log1mexp = 0
        const float out = fminf(l - log1mexp(ll_delta), 0.0f);
        total += in ? out : 0.0f;
        carry += __shfl(inc, kWave - 1, 64);
    }
    return wave_sum_f32x(total);
}

// Order of a selection that does not fit the LDS list (count > capacity: path lengths in the thousands on a graph whose
// rows nearly fill LDS -- the reference draws Poisson(~10) path lengths, main_ISCO_maxcut.py:22-30): repeated extraction of
// the maximum, O(count N / 64) per sample, ONE wave.  The node ids leave in order into `ord` -- global scratch, the sample's
// own row of y_out reinterpreted (written last by the step) -- `taken` is N bytes of LDS that are free at this point (yb).
// Equal keys: lowest node first.
__device__ __noinline__ void isco_order_by_extraction(const float* pert, uint8_t* taken, int64_t N, uint32_t prefix, int count,
                                                      int32_t* ord, int lane) {
    for (int64_t i = lane; i < N; i += kWave) taken[i] = 0;
    lds_fence();
    for (int k = 0; k < count; ++k) {
        uint32_t bk = 0;
        int bi = 0x7FFFFFFF;
        for (int64_t i = lane; i < N; i += kWave) {
            const uint32_t key = fkey(pert[i]);
            if (key >= prefix && !taken[i] && key > bk) { bk = key; bi = (int)i; }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const uint32_t ok = (uint32_t)__shfl_xor((int)bk, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            if (ok > bk || (ok == bk && oi < bi)) { bk = ok; bi = oi; }
        }
        if (lane == 0) { ord[k] = bi; taken[bi] = 1; }
        lds_fence();
    }
    __threadfence();      // the list is read back by every lane (and, in the workgroup kernel, by every wave after a barrier)
}

// ----------------------------------------------------------------------------------------------- MaxCut
// LDS per wave: xb[N] yb[N] bytes (padded to 8) | lp[N] f32 | pert[N] f32 | sel_key[P] f32 | sel_idx[P] i32,
// P = the list's capacity: the next power of two >= N while that fits LDS, else the largest power of two that does
// (N = 10^4, G70: 4096 entries).  A selection larger than P takes isco_order_by_extraction.
struct IscoMcArgs {
    const int32_t* rowptr; const int32_t* col;
    const int32_t* ell_ptr; const int32_t* ell;          // lane-per-node slabs of the symmetric CSR (rls_graph_ell) or NULL
    const float* x; float* y_out; int64_t B, N;
    const int64_t* path_length; float temperature;
    const float* u_gumbel; const float* u_accept; uint64_t seed; int64_t env_offset;
    float* energy_out; float* acc_out; float* terms_out; uint8_t* mask_out;
    int P;
    float* rows;                                         // caller's scratch f32 [B, 2, N] (lp, pert) for rows past the LDS, else NULL
};

// log-probabilities of the single-flip proposal distribution of state `s`: lp_i = log_softmax(gain_i / (2T)),
// gain_i = #same - #differing neighbours (the closed form of the reference's autograd, env_ISCO.py:51-63).
// Returns cut(s) (every edge counted from both ends -> / 2).
__device__ __forceinline__ int isco_local_dist(const uint8_t* s, float* lp, int64_t N, const int32_t* __restrict__ rowptr,
                                               const int32_t* __restrict__ col, const int32_t* __restrict__ ell_ptr,
                                               const int32_t* __restrict__ ell, float temperature, int lane) {
    int differ = 0;
    float mx = -INFINITY;
    if (ell_ptr) {
        // neighbour ids from the lane-per-node slabs: round k of the 64 nodes of a group is one coalesced load that does
        // not depend on anything, eight rounds in flight -- walking col[rowptr[i] ..] per lane was a chain of ~deg L2 round
        // trips per node (155 of the step's 207 us for one sample on a G22-sized graph)
        const int64_t G = (N + 63) >> 6;
        int ep0 = 0, ep1 = 0;                                                // slab offsets of 64 groups at a time, one per lane
        for (int64_t g = 0; g < G; ++g) {
            if ((g & 63) == 0) {
                ep0 = ell_ptr[g + lane <= G ? g + lane : G];
                ep1 = ell_ptr[g + lane + 1 <= G ? g + lane + 1 : G];
            }
            const int64_t i = (g << 6) + lane;
            const bool in = i < N;
            const uint8_t si = in ? s[i] : 0;
            const int e0 = __builtin_amdgcn_readlane(ep0, (int)(g & 63)), e1 = __builtin_amdgcn_readlane(ep1, (int)(g & 63));
            const int deg = in ? rowptr[i + 1] - rowptr[i] : 0;             // requested before the rounds, used after them
            int d = 0;
            const int self = (int)(in ? i : 0);
            int nbn[8];                                                      // the next eight rounds, in flight while these are summed
#pragma unroll
            for (int q = 0; q < 8; ++q) nbn[q] = (e0 + q * kWave < e1) ? ell[e0 + q * kWave + lane] : self;
            for (int k = e0; k < e1; k += 8 * kWave) {
                int nb[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) nb[q] = nbn[q];
                const int kn = k + 8 * kWave;
                if (kn < e1) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) nbn[q] = (kn + q * kWave < e1) ? ell[kn + q * kWave + lane] : self;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) d += (s[nb[q]] != si);          // past a row's end the slab holds the node itself
            }
            if (in) {
                differ += d;
                const float sc = (float)(deg - 2 * d) / (2.0f * temperature);
                lp[i] = sc;
                mx = fmaxf(mx, sc);
            }
        }
    } else {
        for (int64_t i = lane; i < N; i += kWave) {
            const int r0 = rowptr[i], r1 = rowptr[i + 1];
            const uint8_t si = s[i];
            int d = 0;
            for (int j = r0; j < r1; ++j) d += (s[col[j]] != si);
            differ += d;
            const float sc = (float)((r1 - r0) - 2 * d) / (2.0f * temperature);
            lp[i] = sc;
            mx = fmaxf(mx, sc);
        }
    }
    mx = wave_max_f(mx);
    float se = 0.0f;
    for (int64_t i = lane; i < N; i += kWave) se += expf(lp[i] - mx);
    const float lse = logf(wave_sum_f32x(se));
    for (int64_t i = lane; i < N; i += kWave) lp[i] = (lp[i] - mx) - lse;
    lds_fence();
    return wave_sum_i32(differ) >> 1;
}

__global__ __launch_bounds__(512) void k_isco_maxcut_step(IscoMcArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x / kWave;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x / kWave) + wib;
    if (b >= a.B) return;
    const int64_t N = a.N;
    const int P = a.P;
    const size_t nb8 = ((size_t)N + 7) & ~(size_t)7;
    const size_t per_wave = 2 * nb8 + (size_t)N * 8 + (size_t)P * 8;
    unsigned char* base = smem + (size_t)wib * per_wave;
    uint8_t* xb = base;
    uint8_t* yb = base + nb8;
    float* lp = reinterpret_cast<float*>(base + 2 * nb8);
    float* pert = lp + N;
    float* skey = pert + N;
    int32_t* sidx = reinterpret_cast<int32_t*>(skey + P);
    const float* xr = a.x + b * N;
    const float T = a.temperature;
    const uint64_t genv = (uint64_t)(b + a.env_offset);

    for (int64_t i = lane; i < N; i += kWave) xb[i] = xr[i] > 0.0f ? 1 : 0;
    lds_fence();
    // ---- forward: ll_x, proposal distribution, Gumbel perturbation (env_ISCO.py:51-63, util.py:498-516)
    const float ll_x = (float)isco_local_dist(xb, lp, N, a.rowptr, a.col, a.ell_ptr, a.ell, T, lane) / T;
    float lmax = -INFINITY;
    for (int64_t i = lane; i < N; i += kWave) {
        const float u = a.u_gumbel ? a.u_gumbel[b * N + i] : isco_unit(isco_draw(a.seed, genv, (uint32_t)i, 0, 1));
        const float l = lp[i];
        pert[i] = l - logf(-logf(u));
        lmax = fmaxf(lmax, l);
    }
    lmax = wave_max_f(lmax);                                   // ll_base of the forward renormalisation: max over ALL entries
    lds_fence();
    // ---- threshold = L-th largest perturbed value (util.py:518-523): radix select on the monotone key
    int64_t L = a.path_length[b];
    L = L < 1 ? 1 : (L > N ? N : L);
    uint32_t prefix = 0;
    int want = (int)L;
    if (N <= 32 * kWave) {
        // the lane's <= 32 keys in registers for the 32 bit passes (re-reading them from LDS put an LDS round trip on
        // every element of every pass); key 0 (out of range) has no bit set and is never counted
        uint32_t kk[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int64_t i = lane + (int64_t)j * kWave;
            kk[j] = i < N ? fkey(pert[i]) : 0u;
        }
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t hi_mask = bit == 31 ? 0u : (0xFFFFFFFFu << (bit + 1));
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) cnt += ((kk[j] & hi_mask) == prefix) && ((kk[j] >> bit) & 1u);
            cnt = wave_sum_i32(cnt);
            if (cnt >= want) prefix |= 1u << bit;              // the L-th largest has this bit set
            else want -= cnt;
        }
    } else {
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t hi_mask = bit == 31 ? 0u : (0xFFFFFFFFu << (bit + 1));
            int cnt = 0;
            for (int64_t i = lane; i < N; i += kWave) {
                const uint32_t k = fkey(pert[i]);
                cnt += ((k & hi_mask) == prefix) && ((k >> bit) & 1u);
            }
            cnt = wave_sum_i32(cnt);
            if (cnt >= want) prefix |= 1u << bit;              // the L-th largest has this bit set
            else want -= cnt;
        }
    }
    // ---- selected set (perturbed >= threshold), compacted then sorted by perturbed value, descending
    int count = 0;
    for (int64_t i0 = 0; i0 < N; i0 += kWave) {
        const int64_t i = i0 + lane;
        const bool sel = i < N && fkey(pert[i]) >= prefix;
        const uint64_t m = ballot64(sel);
        if (sel) {
            const int at = count + __popcll(m & ((1ull << lane) - 1ull));
            if (at < P) {
                skey[at] = pert[i];
                sidx[at] = (int)i;
            }
        }
        count += __popcll(m);
    }
    const bool big = count > P;                                  // (wave-uniform) the list does not hold the selection
    int32_t* ord = reinterpret_cast<int32_t*>(a.y_out + b * N);  // its order then lives in this sample's output row until the end
    if (big) {
        isco_order_by_extraction(pert, yb, N, prefix, count, ord, lane);
    } else {
        int P2 = 1;
        while (P2 < count) P2 <<= 1;
        for (int k = count + lane; k < P2; k += kWave) { skey[k] = -INFINITY; sidx[k] = -1; }
        lds_fence();
        for (int size = 2; size <= P2; size <<= 1) {
            for (int stride = size >> 1; stride >= 1; stride >>= 1) {
                for (int t = lane; t < (P2 >> 1); t += kWave) {
                    const int lo = ((t / stride) * (stride << 1)) + (t % stride);
                    const int hi = lo + stride;
                    const bool desc = ((lo & size) == 0);
                    const float k0 = skey[lo], k1 = skey[hi];
                    if ((k0 < k1) == desc) {
                        skey[lo] = k1; skey[hi] = k0;
                        const int t0 = sidx[lo]; sidx[lo] = sidx[hi]; sidx[hi] = t0;
                    }
                }
                lds_fence();
            }
        }
    }
    auto sel_at = [&](int k) -> int { return big ? ord[k] : sidx[k]; };
    // ---- ll_x2y: log-probability of drawing the selected set in that order (util.py:530-552)
    const float ll_x2y = noreplacement_ll_sum(count, lmax, lane, [&](int k) { return lp[sel_at(k)]; });
    // ---- y = x with the selected nodes flipped (env_ISCO.py:41-43)
    for (int64_t i = lane; i < N; i += kWave) yb[i] = xb[i];
    lds_fence();
    for (int k = lane; k < count; k += kWave) yb[sel_at(k)] ^= 1;
    lds_fence();
    if (a.mask_out) {
        for (int64_t i = lane; i < N; i += kWave) a.mask_out[b * N + i] = xb[i] ^ yb[i];
    }
    // ---- backward: ll_y and the probability of undoing the selection in reverse order (env_ISCO.py:65-77)
    const float ll_y = (float)isco_local_dist(yb, lp, N, a.rowptr, a.col, a.ell_ptr, a.ell, T, lane) / T;
    float bmax = -INFINITY;
    for (int k = lane; k < count; k += kWave) bmax = fmaxf(bmax, lp[sel_at(k)]);
    bmax = wave_max_f(bmax);
    const float ll_y2x = noreplacement_ll_sum(count, bmax, lane, [&](int k) { return lp[sel_at(count - 1 - k)]; });
    // ---- Metropolis-Hastings accept (env_ISCO.py:31-33, util.py:556-570)
    const float log_acc = fminf(((ll_y + ll_y2x) - ll_x) - ll_x2y, 0.0f);
    const float ua = a.u_accept ? a.u_accept[b] : isco_unit(isco_draw(a.seed, genv, 0xFFFFFFFFu, 0, 2));
    const bool accept = logf(ua + 1e-24f) < log_acc;
    float* yo = a.y_out + b * N;
    for (int64_t i = lane; i < N; i += kWave) yo[i] = (float)(accept ? yb[i] : xb[i]);
    if (lane == 0) {
        if (a.energy_out) a.energy_out[b] = ll_y * T;              // the reference returns ll_y * temperature of the PROPOSAL
        if (a.acc_out) a.acc_out[b] = expf(log_acc);
        if (a.terms_out) {
            float* t = a.terms_out + b * 5;
            t[0] = ll_x; t[1] = ll_x2y; t[2] = ll_y; t[3] = ll_y2x; t[4] = log_acc;
        }
    }
}

// The same step with a WORKGROUP per sample, for the few-chain regime the reference's own configs run in (one chain:
// a lone wave is instruction-issue bound, ~155 us per step on a G22-sized graph).  The node loops are split over the
// kIscoWgWaves waves; maxima / sums meet through LDS; the L-th largest perturbed value comes from four 8-bit histogram
// passes (LDS atomics) instead of 32 one-bit passes; the <= L selected entries are sorted and their path
// log-probabilities summed by wave 0 as before.  Same arithmetic per element; row sums are reduced wave by wave.
constexpr int kIscoWgWaves = 16;

struct IscoWgScratch { float f[kIscoWgWaves]; int i[kIscoWgWaves]; int hist[256]; int count; uint32_t prefix; int want; };

__device__ __forceinline__ float isco_block_max(float v, IscoWgScratch* sc, int lane, int w) {
    v = wave_max_f(v);
    __syncthreads();
    if (lane == 0) sc->f[w] = v;
    __syncthreads();
    float m = sc->f[0];
#pragma unroll
    for (int k = 1; k < kIscoWgWaves; ++k) m = fmaxf(m, sc->f[k]);
    return m;
}
__device__ __forceinline__ float isco_block_sum(float v, IscoWgScratch* sc, int lane, int w) {
    v = wave_sum_f32x(v);
    __syncthreads();
    if (lane == 0) sc->f[w] = v;
    __syncthreads();
    float m = 0.0f;
#pragma unroll
    for (int k = 0; k < kIscoWgWaves; ++k) m += sc->f[k];
    return m;
}
__device__ __forceinline__ int isco_block_sum_i(int v, IscoWgScratch* sc, int lane, int w) {
    v = wave_sum_i32(v);
    __syncthreads();
    if (lane == 0) sc->i[w] = v;
    __syncthreads();
    int m = 0;
#pragma unroll
    for (int k = 0; k < kIscoWgWaves; ++k) m += sc->i[k];
    return m;
}

// isco_local_dist over the whole workgroup: wave w takes the 64-node groups w, w + W, ...
__device__ __forceinline__ int isco_local_dist_wg(const uint8_t* s, float* lp, int64_t N, const IscoMcArgs& a, float temperature,
                                                  IscoWgScratch* sc, int lane, int w) {
    int differ = 0;
    float mx = -INFINITY;
    const int64_t G = (N + 63) >> 6;
    for (int64_t g = w; g < G; g += kIscoWgWaves) {
        const int64_t i = (g << 6) + lane;
        const bool in = i < N;
        const uint8_t si = in ? s[i] : 0;
        int d = 0;
        const int deg = in ? a.rowptr[i + 1] - a.rowptr[i] : 0;
        if (a.ell_ptr) {
            const int e0 = a.ell_ptr[g], e1 = a.ell_ptr[g + 1];
            const int self = (int)(in ? i : 0);
            for (int k = e0; k < e1; k += 8 * kWave) {
                int nb[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) nb[q] = (k + q * kWave < e1) ? a.ell[k + q * kWave + lane] : self;
#pragma unroll
                for (int q = 0; q < 8; ++q) d += (s[nb[q]] != si);
            }
        } else if (in) {
            for (int j = a.rowptr[i]; j < a.rowptr[i + 1]; ++j) d += (s[a.col[j]] != si);
        }
        if (in) {
            differ += d;
            const float v = (float)(deg - 2 * d) / (2.0f * temperature);
            lp[i] = v;
            mx = fmaxf(mx, v);
        }
    }
    mx = isco_block_max(mx, sc, lane, w);
    float se = 0.0f;
    for (int64_t g = w; g < G; g += kIscoWgWaves) {
        const int64_t i = (g << 6) + lane;
        if (i < N) se += expf(lp[i] - mx);
    }
    const float lse = logf(isco_block_sum(se, sc, lane, w));
    for (int64_t g = w; g < G; g += kIscoWgWaves) {
        const int64_t i = (g << 6) + lane;
        if (i < N) lp[i] = (lp[i] - mx) - lse;
    }
    const int tot = isco_block_sum_i(differ, sc, lane, w);        // (its barriers also publish lp[])
    return tot >> 1;
}

// GROWS: the two f32 rows (lp, pert) live in the caller's scratch (global memory, L2-resident while a sample is worked on) and
// LDS holds the byte rows and the list only -- N past ~15 900, up to ~81 000.  The workgroup's barriers order its global
// stores and loads as they do the LDS ones (__syncthreads fences both at workgroup scope; the L1 is the CU's).
template <bool GROWS>
__global__ __launch_bounds__(kIscoWgWaves * kWave) void k_isco_maxcut_step_wg(IscoMcArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int tid = threadIdx.x, NT = kIscoWgWaves * kWave;
    const int64_t b = blockIdx.x;
    const int64_t N = a.N;
    const int P = a.P;
    const size_t nb8 = ((size_t)N + 7) & ~(size_t)7;
    uint8_t* xb = smem;
    uint8_t* yb = smem + nb8;
    float* lp = GROWS ? a.rows + b * 2 * N : reinterpret_cast<float*>(smem + 2 * nb8);
    float* pert = lp + N;
    float* skey = GROWS ? reinterpret_cast<float*>(smem + 2 * nb8) : pert + N;
    int32_t* sidx = reinterpret_cast<int32_t*>(skey + P);
    IscoWgScratch* sc = reinterpret_cast<IscoWgScratch*>(sidx + P);
    const float* xr = a.x + b * N;
    const float T = a.temperature;
    const uint64_t genv = (uint64_t)(b + a.env_offset);

    for (int64_t i = tid; i < N; i += NT) xb[i] = xr[i] > 0.0f ? 1 : 0;
    __syncthreads();
    // ---- forward: ll_x, proposal distribution, Gumbel perturbation
    const float ll_x = (float)isco_local_dist_wg(xb, lp, N, a, T, sc, lane, w) / T;
    float lmax = -INFINITY;
    for (int64_t i = tid; i < N; i += NT) {
        const float u = a.u_gumbel ? a.u_gumbel[b * N + i] : isco_unit(isco_draw(a.seed, genv, (uint32_t)i, 0, 1));
        const float l = lp[i];
        pert[i] = l - logf(-logf(u));
        lmax = fmaxf(lmax, l);
    }
    lmax = isco_block_max(lmax, sc, lane, w);                  // ll_base of the forward renormalisation (also publishes pert[])
    // ---- threshold = L-th largest perturbed value: radix select, 8 bits per pass
    int64_t L = a.path_length[b];
    L = L < 1 ? 1 : (L > N ? N : L);
    uint32_t prefix = 0;
    int want = (int)L;
    for (int shift = 24; shift >= 0; shift -= 8) {
        const uint32_t hi_mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int k = tid; k < 256; k += NT) sc->hist[k] = 0;
        __syncthreads();
        for (int64_t i = tid; i < N; i += NT) {
            const uint32_t k = fkey(pert[i]);
            if ((k & hi_mask) == prefix) atomicAdd(&sc->hist[(k >> shift) & 255u], 1);
        }
        __syncthreads();
        if (w == 0) {   // the largest byte value v whose suffix count (entries with byte >= v) reaches `want`
            int c[4], tot = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { c[q] = sc->hist[lane * 4 + q]; tot += c[q]; }
            int suf = tot;                                        // inclusive suffix sum over lanes: lanes >= this one
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const int o = __shfl_down(suf, d, 64);
                if (lane + d < kWave) suf += o;
            }
            const int above = suf - tot;                          // entries in bins of higher lanes
            // inside this lane's four bins, from the top
            int byte_found = -1, want_left = 0, run = above;
#pragma unroll
            for (int q = 3; q >= 0; --q) {
                if (byte_found < 0 && run < want && run + c[q] >= want) { byte_found = lane * 4 + q; want_left = want - run; }
                run += c[q];
            }
            const uint64_t m = ballot64(byte_found >= 0);
            if (byte_found >= 0) { sc->prefix = prefix | ((uint32_t)byte_found << shift); sc->want = want_left; }
            (void)m;
        }
        __syncthreads();
        prefix = sc->prefix;
        want = sc->want;
    }
    // ---- selected set (perturbed >= threshold), compacted (any order: sorted next)
    if (tid == 0) sc->count = 0;
    __syncthreads();
    for (int64_t i = tid; i < N; i += NT) {
        if (fkey(pert[i]) >= prefix) {
            const int at = atomicAdd(&sc->count, 1);
            if (at < P) {
                skey[at] = pert[i];
                sidx[at] = (int)i;
            }
        }
    }
    __syncthreads();
    const int count = sc->count;
    const bool big = count > P;                                  // the list does not hold the selection: order by extraction
    int32_t* ord = reinterpret_cast<int32_t*>(a.y_out + b * N);  // (this sample's output row, written last)
    auto sel_at = [&](int k) -> int { return big ? ord[k] : sidx[k]; };
    float ll_x2y = 0.0f;
    if (w == 0 && big) {
        isco_order_by_extraction(pert, yb, N, prefix, count, ord, lane);
        ll_x2y = noreplacement_ll_sum(count, lmax, lane, [&](int k) { return lp[ord[k]]; });
    } else if (w == 0) {
        int P2 = 1;
        while (P2 < count) P2 <<= 1;
        for (int k = count + lane; k < P2; k += kWave) { skey[k] = -INFINITY; sidx[k] = -1; }
        lds_fence();
        for (int size = 2; size <= P2; size <<= 1) {
            for (int stride = size >> 1; stride >= 1; stride >>= 1) {
                for (int t = lane; t < (P2 >> 1); t += kWave) {
                    const int lo = ((t / stride) * (stride << 1)) + (t % stride);
                    const int hi = lo + stride;
                    const bool desc = ((lo & size) == 0);
                    const float k0 = skey[lo], k1 = skey[hi];
                    // equal keys: order by node id, so that the result does not depend on the compaction order
                    const int i0 = sidx[lo], i1 = sidx[hi];
                    const bool lt = (k0 < k1) || (k0 == k1 && i0 > i1);
                    if (lt == desc) {
                        skey[lo] = k1; skey[hi] = k0;
                        sidx[lo] = i1; sidx[hi] = i0;
                    }
                }
                lds_fence();
            }
        }
        ll_x2y = noreplacement_ll_sum(count, lmax, lane, [&](int k) { return lp[sidx[k]]; });
    }
    // ---- y = x with the selected nodes flipped
    __syncthreads();                                              // (wave 0 may have used yb as the extraction's marks)
    for (int64_t i = tid; i < N; i += NT) yb[i] = xb[i];
    __syncthreads();
    for (int k = tid; k < count; k += NT) yb[sel_at(k)] ^= 1;
    __syncthreads();
    if (a.mask_out) {
        for (int64_t i = tid; i < N; i += NT) a.mask_out[b * N + i] = xb[i] ^ yb[i];
    }
    // ---- backward
    const float ll_y = (float)isco_local_dist_wg(yb, lp, N, a, T, sc, lane, w) / T;
    if (w == 0) {
        float bmax = -INFINITY;
        for (int k = lane; k < count; k += kWave) bmax = fmaxf(bmax, lp[sel_at(k)]);
        bmax = wave_max_f(bmax);
        const float ll_y2x = noreplacement_ll_sum(count, bmax, lane, [&](int k) { return lp[sel_at(count - 1 - k)]; });
        const float log_acc = fminf(((ll_y + ll_y2x) - ll_x) - ll_x2y, 0.0f);
        const float ua = a.u_accept ? a.u_accept[b] : isco_unit(isco_draw(a.seed, genv, 0xFFFFFFFFu, 0, 2));
        const bool accept = logf(ua + 1e-24f) < log_acc;
        if (lane == 0) {
            sc->want = accept ? 1 : 0;
            if (a.energy_out) a.energy_out[b] = ll_y * T;
            if (a.acc_out) a.acc_out[b] = expf(log_acc);
            if (a.terms_out) {
                float* t = a.terms_out + b * 5;
                t[0] = ll_x; t[1] = ll_x2y; t[2] = ll_y; t[3] = ll_y2x; t[4] = log_acc;
            }
        }
    }
    __syncthreads();
    const bool accept = sc->want != 0;
    float* yo = a.y_out + b * N;
    for (int64_t i = tid; i < N; i += NT) yo[i] = (float)(accept ? yb[i] : xb[i]);
}

// ----------------------------------------------------------------------------------------------- TSP
struct IscoTspArgs {
    const float* dist; int64_t N; int32_t K; int32_t random_stride; float near_threshold;
    const int32_t* nearest; const int32_t* random;       // [N, K], [N, N-K-1]
    const int64_t* perm_in; int64_t* perm_out; int64_t B; int32_t path_length; float temperature;
    const float* u_partner; const int64_t* r_near; const int64_t* r_rand; const float* u_gumbel;   // [L, B, N] or NULL
    const float* u_accept;                                // [B] or NULL
    uint64_t seed; int64_t env_offset;
    float* log_acc_out; float* acc_out; int64_t* cur_out;
};

template <bool LDS_D>
__global__ __launch_bounds__(256) void k_isco_tsp_step(IscoTspArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x / kWave;
    const int n = (int)a.N;
    float* dl = reinterpret_cast<float*>(smem);
    const float* D = a.dist;
    if constexpr (LDS_D) {
        for (int64_t c = threadIdx.x; c < a.N * a.N; c += blockDim.x) dl[c] = a.dist[c];
        __syncthreads();
        D = dl;
    }
    unsigned char* wbase = smem + (LDS_D ? (size_t)a.N * a.N * 4 : 0) + (size_t)wib * ((size_t)n * 16);
    int32_t* Pm = reinterpret_cast<int32_t*>(wbase);      // tour
    int32_t* INV = Pm + n;                                  // city -> position
    float* lg = reinterpret_cast<float*>(INV + n);          // logits of this iteration
    int32_t* jb = reinterpret_cast<int32_t*>(lg + n);       // partner position | banned << 31, per position
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x / kWave) + wib;
    if (b >= a.B) return;
    const uint64_t genv = (uint64_t)(b + a.env_offset);
    const int K = a.K, NR = n - K - 1;
    const float T = a.temperature;
    const int64_t* pin = a.perm_in + b * a.N;
    for (int k = lane; k < n; k += kWave) {
        const int city = (int)pin[k];
        Pm[k] = city;
        INV[city] = k;
    }
    lds_fence();
    auto DD = [&](int u, int v) { return D[(int64_t)u * n + v]; };
    float s_delta = 0.0f, s_fwd = 0.0f, s_bwd = 0.0f;       // the three rows of `traj` (env_ISCO.py:190-196)
    for (int it = 0; it < a.path_length; ++it) {
        const int64_t row = ((int64_t)it * a.B + b) * a.N;
        // ---- opt_2 (env_ISCO.py:238-335): partner city per position, swap delta, ban mask; logits = logratio / 2
        float mx = -INFINITY;
        for (int i = lane; i < n; i += kWave) {
            const int city = Pm[i];
            float up;
            int rn, rr;
            if (a.u_partner) {
                up = a.u_partner[row + i];
                rn = (int)a.r_near[row + i];
                rr = (int)a.r_rand[row + i];
            } else {
                up = isco_unit(isco_draw(a.seed, genv, (uint32_t)i, (uint32_t)it, 3));
                rn = (int)(((uint64_t)isco_draw(a.seed, genv, (uint32_t)i, (uint32_t)it, 4) * (uint64_t)K) >> 32);
                rr = NR > 0 ? (int)(((uint64_t)isco_draw(a.seed, genv, (uint32_t)i, (uint32_t)it, 5) * (uint64_t)NR) >> 32) : 0;
            }
            const bool near = up < a.near_threshold;                              // rand < K / (K + 1)
            const int sel = near ? a.nearest[(int64_t)city * K + rn] : a.random[(int64_t)city * a.random_stride + rr];
            const int j = INV[sel];
            const int i0 = (i == 0) ? n - 1 : i - 1;
            const int i1 = (i + 1 == n) ? 0 : i + 1;
            const int i2 = (i1 + 1 == n) ? 0 : i1 + 1;
            const int j0 = (j == 0) ? n - 1 : j - 1;
            const int j1 = (j + 1 == n) ? 0 : j + 1;
            const int s_m1 = Pm[i1], s_m0 = Pm[i0];
            const bool banned = (s_m1 == sel) || (s_m0 == sel);
            const int s_i0 = Pm[j0], s_i1 = Pm[j1], s_i = sel;
            const bool c3 = (s_m1 == s_i0);
            const int nm = city, nm1 = s_m1, nm2 = Pm[i2];
            float delta;
            if (banned) delta = 0.0f;
            else if (c3) delta = -(DD(nm, nm1) + DD(s_i, s_i1)) + (DD(nm, s_i) + DD(s_i0, s_i1));
            else delta = -(((DD(nm, nm1) + DD(nm1, nm2)) + DD(s_i0, s_i)) + DD(s_i, s_i1)) +
                         (((DD(nm, s_i) + DD(s_i, nm2)) + DD(s_i0, nm1)) + DD(nm1, s_i1));
            const float logratio = banned ? -1e6f : (-delta) / T;                 // :197 logratio[ban_mask] = -1e6
            const float logit = logratio / 2.0f;                                  // apply_weight_function_logscale
            lg[i] = logit;
            jb[i] = j | (banned ? (int)0x80000000u : 0);
            mx = fmaxf(mx, logit);
        }
        lds_fence();
        mx = wave_max_f(mx);
        float se = 0.0f;
        for (int i = lane; i < n; i += kWave) se += expf(lg[i] - mx);
        const float lse = logf(wave_sum_f32x(se));
        // ---- multinomial with one draw: the position with the largest Gumbel-perturbed log-probability
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = lane; i < n; i += kWave) {
            const float u = a.u_gumbel ? a.u_gumbel[row + i] : isco_unit(isco_draw(a.seed, genv, (uint32_t)i, (uint32_t)it, 6));
            const float pv = ((lg[i] - mx) - lse) - logf(-logf(u));
            if (pv > best) { best = pv; bi = i; }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ob = __shfl_xor(best, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        const int q = bi;
        const float lq = lg[q];
        const float ll_x2y = (lq - mx) - lse;                                      // first draw: no renormalisation term
        const uint32_t jq = (uint32_t)jb[q];
        const bool q_banned = jq >> 31;
        const float delta_yx = q_banned ? -1e6f : lq * 2.0f;                       // logratio of the selected position
        // ---- reverse move: log_softmax of the logits with the selected one negated, at the selected position (:214-226)
        const float nq = -lq;
        float mx2 = -INFINITY;
        for (int i = lane; i < n; i += kWave) mx2 = fmaxf(mx2, i == q ? nq : lg[i]);
        mx2 = wave_max_f(mx2);
        float se2 = 0.0f;
        for (int i = lane; i < n; i += kWave) se2 += expf((i == q ? nq : lg[i]) - mx2);
        const float ll_y2x = (nq - mx2) - logf(wave_sum_f32x(se2));
        s_delta += delta_yx;
        s_fwd += -ll_x2y;
        s_bwd += ll_y2x;
        // ---- switch (:337-344): swap the cities at positions q + 1 and j(q) unless banned
        lds_fence();
        if (!q_banned && lane == 0) {
            const int j = (int)(jq & 0x7fffffffu);
            const int p1 = (q + 1 == n) ? 0 : q + 1;
            const int c1 = Pm[p1], c2 = Pm[j];
            Pm[p1] = c2; Pm[j] = c1;
            INV[c2] = p1; INV[c1] = j;
        }
        lds_fence();
    }
    const float log_acc = fminf((s_delta + s_fwd) + s_bwd, 0.0f);
    const float ua = a.u_accept ? a.u_accept[b] : isco_unit(isco_draw(a.seed, genv, 0xFFFFFFFFu, 0, 7));
    const bool accept = logf(ua + 1e-24f) < log_acc;
    int64_t* po = a.perm_out + b * a.N;
    for (int k = lane; k < n; k += kWave) {
        const int64_t cur = Pm[k];
        if (a.cur_out) a.cur_out[b * a.N + k] = cur;
        po[k] = accept ? cur : pin[k];
    }
    if (lane == 0) {
        if (a.log_acc_out) a.log_acc_out[b] = log_acc;
        if (a.acc_out) a.acc_out[b] = expf(log_acc);
    }
}

// K10' evolutionary_replacement  methods/util.py:87-94, the row moves: xs[dst[k]] = xs[src[k]], vs likewise.
// The reference's  xs[replace_ids] = xs[low_ids]  gathers the right-hand side first; dst and src are disjoint
// by construction (top_ids vs low_ids), so the copy may run in place.  One wave per moved row, 16-byte lanes.
__global__ __launch_bounds__(256) void k_copy_rows(uint8_t* __restrict__ xs, int64_t* __restrict__ vs, int64_t N,
                                                   const int64_t* __restrict__ dst, const int64_t* __restrict__ src,
                                                   int64_t K, int vec) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t k = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (k >= K) return;
    const int64_t d = dst[k], s = src[k];
    if (vec) {
        const u32x4* sp = reinterpret_cast<const u32x4*>(xs + s * N);
        u32x4* dp = reinterpret_cast<u32x4*>(xs + d * N);
        for (int64_t i = lane; i < N / 16; i += kWave) dp[i] = sp[i];
    } else {
        for (int64_t i = lane; i < N; i += kWave) xs[d * N + i] = xs[s * N + i];
    }
    if (lane == 0 && vs) vs[d] = vs[s];
}

}  // namespace rls

using namespace rls;

extern "C" {

// bytes of LDS a sample needs with its two f32 rows in LDS and the smallest list (64 entries)
static inline size_t isco_mc_lds_min(int64_t N) {
    return 2 * (((size_t)N + 7) & ~(size_t)7) + (size_t)N * 8 + 64 * 8 + sizeof(IscoWgScratch) + 16;
}

int64_t rls_isco_maxcut_scratch_bytes(const rls_graph* g, int64_t B) {
    if (!g || B <= 0 || (isco_mc_lds_min(g->num_nodes) <= (size_t)kLdsBytes && !knob_on(KN_ISCO_GLOBAL_ROWS))) return 0;
    return B * g->num_nodes * 8;
}

int rls_isco_maxcut_step(const rls_graph* g, const float* x, float* y_out, int64_t B, const int64_t* path_length,
                         float temperature, const float* u_gumbel, const float* u_accept, uint64_t seed,
                         int64_t env_offset, float* energy_out, float* acc_out, float* terms_out, uint8_t* mask_out,
                         void* scratch, int64_t scratch_bytes, void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(B >= 0, RLS_EINVAL, "B < 0");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(x && y_out && path_length, RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE((u_gumbel == nullptr) == (u_accept == nullptr), RLS_EINVAL, "u_gumbel and u_accept must both be given or both be NULL");
    RLS_REQUIRE(temperature > 0.0f, RLS_EINVAL, "temperature must be > 0");
    const int64_t N = g->num_nodes;
    RLS_REQUIRE((const void*)x != (const void*)y_out, RLS_EINVAL, "y_out must not alias x (a rejected proposal restores x; the row also serves as scratch)");
    // capacity of the selected-set list in LDS: a power of two >= N while that fits (N <= 4096: as before), else the largest
    // power of two the rows leave room for, at least 64 (N = 10^4: 4096; N = 15 000: 512) -- a larger selection takes the
    // extraction path.  The rows themselves (two byte rows, two f32 rows) bound N at ~15 900.
    const int force_cap = (int)knob(KN_ISCO_SEL_CAP, 0);   // dev / test knob
    const size_t fixed = sizeof(IscoWgScratch) + 16;
    const bool use_ell = g->ell_sym_ptr && g->ell_sym && !g->wgt;
    if (isco_mc_lds_min(N) > (size_t)kLdsBytes || knob_on(KN_ISCO_GLOBAL_ROWS)) {
        // rows past the LDS: a workgroup per sample with lp / pert in the caller's scratch (rls_isco_maxcut_scratch_bytes), the byte
        // rows and a list of up to 4096 entries in LDS
        const size_t brow = 2 * (((size_t)N + 7) & ~(size_t)7);
        RLS_REQUIRE(brow + 64 * 8 + fixed <= (size_t)kLdsBytes, RLS_EUNSUPPORTED, "N=%lld: the two byte rows of a sample need %zu B of LDS (max %d)",
                    (long long)N, brow + 64 * 8 + fixed, kLdsBytes);
        RLS_REQUIRE(scratch && scratch_bytes >= B * N * 8, RLS_EINVAL,
                    "N=%lld needs %lld B of scratch (rls_isco_maxcut_scratch_bytes), got %lld", (long long)N, (long long)(B * N * 8),
                    (long long)(scratch ? scratch_bytes : 0));
        int Pg = 64;
        while (Pg < 4096 && Pg < N && brow + (size_t)Pg * 16 + fixed <= (size_t)kLdsBytes) Pg <<= 1;
        if (force_cap >= 2 && force_cap < Pg && (force_cap & (force_cap - 1)) == 0) Pg = force_cap;
        const size_t lds_g = brow + (size_t)Pg * 8 + fixed;
        IscoMcArgs ag{g->rowptr, g->col, use_ell ? g->ell_sym_ptr : nullptr, use_ell ? g->ell_sym : nullptr, x, y_out, B, N,
                      path_length, temperature, u_gumbel, u_accept, seed, env_offset, energy_out, acc_out, terms_out, mask_out, Pg,
                      static_cast<float*>(scratch)};
        ensure_dyn_lds((const void*)k_isco_maxcut_step_wg<true>, lds_g);
        hipLaunchKernelGGL(k_isco_maxcut_step_wg<true>, dim3((unsigned)B), dim3(kIscoWgWaves * kWave), lds_g, as_stream(stream), ag);
        return check_launch("k_isco_maxcut_step_wg<global rows>");
    }
    const size_t rows = 2 * (((size_t)N + 7) & ~(size_t)7) + (size_t)N * 8;
    int P = 1;
    while (P < N) P <<= 1;
    while (P > 64 && rows + (size_t)P * 8 + fixed > (size_t)kLdsBytes) P >>= 1;
    if (force_cap >= 2 && force_cap < P && (force_cap & (force_cap - 1)) == 0) P = force_cap;
    const size_t per_wave = rows + (size_t)P * 8;
    RLS_REQUIRE(per_wave + fixed <= (size_t)kLdsBytes, RLS_EUNSUPPORTED, "N=%lld needs %zu B of LDS per env (max %d)", (long long)N,
                per_wave + fixed, kLdsBytes);
    const int force_wg = (int)knob(KN_ISCO_FORCE_WG, -1);      // dev knob: 0 wave per sample, 1 workgroup per sample
    // The workgroup kernel (8-bit radix passes, node loops over 16 waves) also wins at LARGE batches from ~1500 nodes on: a wave's
    // select is 32 one-bit passes over all its keys (from LDS past 2048 nodes) and its rows leave a CU ever fewer samples -- 4096
    // samples (tools/timing/isco_kernels.py): N = 2000 485 vs 427 us, N = 4000 2121 vs 527, G70's 10^4 8410 vs 852, N = 15 000 14 526
    // vs 1320; at N = 800 the wave kernel keeps the large batches (163 vs 301)
    const bool use_wg = force_wg >= 0 ? (force_wg != 0 && N >= 4 * kWave)
                                      : (B <= (int64_t)2 * num_cus() || N >= 1536) && N >= 4 * kWave && per_wave + fixed <= (size_t)kLdsBytes;
    // wave-per-sample kernel: one wave per SIMD is latency-bound (round 3: 4 samples per CU), so the list gives up capacity for
    // resident samples while it stays >= 512 entries -- the reference draws Poisson(~10) path lengths, a longer selection takes
    // the extraction path -- G22-sized rows: 2048 entries x 4 samples -> 512 x 6 per CU, 4096 samples 606 -> 508 us
    const int max_waves = (int)knob(KN_ISCO_WAVES, 8);     // dev knob (<= 8)
    auto waves_for = [&](int cap) {
        int wv = (int)((size_t)kLdsBytes / (rows + (size_t)cap * 8));
        return wv > max_waves ? max_waves : wv;
    };
    int Pw = P;
    if (!use_wg && force_cap == 0)
        while (Pw > 512 && waves_for(Pw >> 1) > waves_for(Pw)) Pw >>= 1;
    const int waves = waves_for(Pw) < 1 ? 1 : waves_for(Pw);
    const size_t lds = (rows + (size_t)Pw * 8) * waves;
    // few chains (the reference's configs run one): a workgroup per sample; from ~4 samples per CU on the wave-per-sample
    // kernel has the throughput
    const size_t lds_wg = per_wave + sizeof(IscoWgScratch) + 16;
    if (use_wg) {
        IscoMcArgs aw{g->rowptr, g->col, use_ell ? g->ell_sym_ptr : nullptr, use_ell ? g->ell_sym : nullptr, x, y_out, B, N,
                      path_length, temperature, u_gumbel, u_accept, seed, env_offset, energy_out, acc_out, terms_out, mask_out, P, nullptr};
        if (lds_wg > 64 * 1024)
            ensure_dyn_lds((const void*)k_isco_maxcut_step_wg<false>, lds_wg);
        hipLaunchKernelGGL(k_isco_maxcut_step_wg<false>, dim3((unsigned)B), dim3(kIscoWgWaves * kWave), lds_wg, as_stream(stream), aw);
        return check_launch("k_isco_maxcut_step_wg");
    }
    IscoMcArgs a{g->rowptr, g->col, use_ell ? g->ell_sym_ptr : nullptr, use_ell ? g->ell_sym : nullptr, x, y_out, B, N, path_length, temperature, u_gumbel, u_accept, seed, env_offset,
                 energy_out, acc_out, terms_out, mask_out, Pw, nullptr};
    if (lds > 64 * 1024)
        ensure_dyn_lds((const void*)k_isco_maxcut_step, lds);
    hipLaunchKernelGGL(k_isco_maxcut_step, dim3((unsigned)ceil_div(B, waves)), dim3(waves * kWave), lds, as_stream(stream), a);
    return check_launch("k_isco_maxcut_step");
}

int rls_isco_tsp_step(const float* dist, int64_t N, const int32_t* nearest, int32_t K, float near_threshold, const int32_t* random,
                      int32_t random_stride,
                      const int64_t* perm_in, int64_t* perm_out, int64_t B, int32_t path_length, float temperature,
                      const float* u_partner, const int64_t* r_near, const int64_t* r_rand, const float* u_gumbel,
                      const float* u_accept, uint64_t seed, int64_t env_offset, float* log_acc_out, float* acc_out,
                      int64_t* cur_out, void* stream) {
    RLS_REQUIRE(N > 2 && N < (1 << 24) && B >= 0 && path_length >= 0, RLS_EINVAL, "bad sizes N=%lld B=%lld", (long long)N, (long long)B);
    RLS_REQUIRE(K >= 1 && K < N - 1, RLS_EINVAL, "K=%d outside [1, N-2]", K);
    RLS_REQUIRE(random_stride >= N - K - 1, RLS_EINVAL, "random_stride=%d < N - K - 1", random_stride);
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(dist && nearest && random && perm_in && perm_out, RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE(perm_in != perm_out, RLS_EINVAL, "perm_out must not alias perm_in (a rejected move restores it)");
    const bool test = u_partner || r_near || r_rand || u_gumbel || u_accept;
    RLS_REQUIRE(!test || (u_partner && r_near && r_rand && u_gumbel && u_accept), RLS_EINVAL,
                "test draws must be given all together");
    RLS_REQUIRE(temperature > 0.0f, RLS_EINVAL, "temperature must be > 0");
    const size_t per_wave = (size_t)N * 16;
    const size_t dbytes = (size_t)N * N * 4;
    const bool lds_d = dbytes + 4 * per_wave <= (size_t)kLdsBytes - 1024;
    int waves = 4;
    if (!lds_d) {
        RLS_REQUIRE(per_wave <= (size_t)kLdsBytes, RLS_EUNSUPPORTED, "N=%lld too large for the per-env tour scratch", (long long)N);
        while (waves > 1 && per_wave * waves > (size_t)kLdsBytes) waves >>= 1;
    }
    const size_t lds = (lds_d ? dbytes : 0) + per_wave * waves;
    IscoTspArgs a{dist, N, K, random_stride, near_threshold, nearest, random, perm_in, perm_out, B, path_length, temperature, u_partner, r_near, r_rand,
                  u_gumbel, u_accept, seed, env_offset, log_acc_out, acc_out, cur_out};
    const dim3 grid((unsigned)ceil_div(B, waves)), block(waves * kWave);
    if (lds_d) {
        if (lds > 64 * 1024) ensure_dyn_lds((const void*)k_isco_tsp_step<true>, lds);
        hipLaunchKernelGGL(k_isco_tsp_step<true>, grid, block, lds, as_stream(stream), a);
    } else {
        if (lds > 64 * 1024) ensure_dyn_lds((const void*)k_isco_tsp_step<false>, lds);
        hipLaunchKernelGGL(k_isco_tsp_step<false>, grid, block, lds, as_stream(stream), a);
    }
    return check_launch("k_isco_tsp_step");
}

int rls_copy_rows(uint8_t* xs, int64_t* vs, int64_t N, const int64_t* dst, const int64_t* src, int64_t K, void* stream) {
    RLS_REQUIRE(N > 0 && K >= 0, RLS_EINVAL, "bad sizes");
    if (K == 0) return RLS_OK;
    RLS_REQUIRE(xs && dst && src, RLS_EINVAL, "NULL pointer");
    const int vec = ((((uintptr_t)xs) & 15) == 0 && N % 16 == 0) ? 1 : 0;
    hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)ceil_div(K, 4)), dim3(256), 0, as_stream(stream), xs, vs, N, dst, src, K, vec);
    return check_launch("k_copy_rows");
}

}  // extern "C"

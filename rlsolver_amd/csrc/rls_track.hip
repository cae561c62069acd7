// Best-solution tracking on the device: the argmax / compare / copy of
// Evaluator.record2 (rlsolver/methods/util_evaluator.py:90-107) as one launch that never synchronises -- the
// reference's  float(good_v)  forces a host read per call.  State lives in device memory: best_v (double[1]),
// best_x (uint8[N]), improved (uint8[1]) and a value log; the host reads them only when it prints.
#include "rls_common.h"

namespace rls {

constexpr int kTrackThreads = 1024;

template <typename V>
__global__ __launch_bounds__(kTrackThreads) void k_best_update(const uint8_t* __restrict__ xs, const V* __restrict__ vs,
                                                               int64_t B, int64_t N, int if_maximize,
                                                               uint8_t* __restrict__ best_x, double* __restrict__ best_v,
                                                               uint8_t* __restrict__ improved, double* __restrict__ log_v,
                                                               int64_t log_index, int force) {
    __shared__ double s_val[kTrackThreads / 64];
    __shared__ int64_t s_idx[kTrackThreads / 64];
    __shared__ int64_t s_win;
    __shared__ int s_take;
    const double sign = if_maximize ? 1.0 : -1.0;
    double bv = -INFINITY;
    int64_t bi = INT64_MAX;
    for (int64_t b = threadIdx.x; b < B; b += kTrackThreads) {           // first extremum, like torch.argmax / argmin
        const double v = sign * (double)vs[b];
        if (v > bv) { bv = v; bi = b; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ov = __shfl_xor(bv, m, 64);
        const int64_t oi = __shfl_xor(bi, m, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { s_val[threadIdx.x >> 6] = bv; s_idx[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kTrackThreads / 64; ++w)
            if (s_val[w] > bv || (s_val[w] == bv && s_idx[w] < bi)) { bv = s_val[w]; bi = s_idx[w]; }
        const double good = sign * bv;
        const bool take = force || (if_maximize ? good > best_v[0] : good < best_v[0]);   // strict, util_evaluator.py:103
        if (take) best_v[0] = good;
        improved[0] = take ? 1 : 0;
        if (log_v) log_v[log_index] = good;
        s_win = bi;
        s_take = take;
    }
    __syncthreads();
    if (s_take) {
        const uint8_t* row = xs + s_win * N;
        for (int64_t n = threadIdx.x; n < N; n += kTrackThreads) best_x[n] = row[n];
    }
}

// The key of the episode-boundary exchange (rlsolver_amd/dist.py): one launch instead of an argmax and
// half a dozen [1]-sized torch ops in front of the 8-byte all-reduce.  key[0] = (best << rank_bits) | low_code with best =
// max_b vs[b] (doubled for float inputs, which carry integers or half-integers: the bidirectional envs return count / 2),
// index[0] = its first position; flag[0] |= 1 when the value does not fit |best| < limit or is not a half-integer.
template <typename V>
__global__ __launch_bounds__(kTrackThreads) void k_best_key(const V* __restrict__ vs, int64_t B, int is_float, int rank_bits,
                                                            int64_t low_code, int64_t limit, int64_t* __restrict__ key,
                                                            int64_t* __restrict__ index, int32_t* __restrict__ flag) {
    __shared__ double s_val[kTrackThreads / 64];
    __shared__ int64_t s_idx[kTrackThreads / 64];
    double bv = -INFINITY;
    int64_t bi = INT64_MAX;
    // one workgroup (the key must come out of ONE launch with no scratch): 16 bytes per lane per load and four loads in flight,
    // or the 64 dependent 4-byte loads per thread of a 2^16-env batch are 15 us of latency on the only cross-rank path.  A
    // thread's indices grow with its iterations, so the strict compare keeps the FIRST maximum (torch.argmax).
    constexpr int VEC = 16 / (int)sizeof(V);
    int64_t done = 0;
    if ((reinterpret_cast<uintptr_t>(vs) & 15) == 0) {
        const int64_t nvec = B / VEC;
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* __restrict__ v4 = reinterpret_cast<const u32x4*>(vs);
#pragma unroll 4
        for (int64_t i = threadIdx.x; i < nvec; i += kTrackThreads) {
            const u32x4 raw = __builtin_nontemporal_load(v4 + i);
            V e[VEC];
            __builtin_memcpy(e, &raw, 16);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const double v = (double)e[j];
                if (v > bv) { bv = v; bi = i * VEC + j; }
            }
        }
        done = nvec * VEC;
    }
    for (int64_t b = done + threadIdx.x; b < B; b += kTrackThreads) {
        const double v = (double)vs[b];
        if (v > bv) { bv = v; bi = b; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ov = __shfl_xor(bv, m, 64);
        const int64_t oi = __shfl_xor(bi, m, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { s_val[threadIdx.x >> 6] = bv; s_idx[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kTrackThreads / 64; ++w)
            if (s_val[w] > bv || (s_val[w] == bv && s_idx[w] < bi)) { bv = s_val[w]; bi = s_idx[w]; }
        const double scaled = is_float ? bv * 2.0 : bv;
        const double r = rint(scaled);
        const bool ok = r == scaled && fabs(r) < (double)limit;
        const int64_t best = ok ? (int64_t)r : 0;
        key[0] = (int64_t)((uint64_t)best << rank_bits) | low_code;      // (arithmetic meaning: best * 2^rank_bits + low_code)
        if (index) index[0] = bi;
        if (!ok) atomicOr(flag, 1);
    }
}

// After the all-reduce: the reduced key back into (objective, owner rank) in ONE launch (the host-side alternative is a shift,
// a mask and a subtraction of [1]-sized tensors, three launches on the only cross-rank path).  obj_out is int64[1], or
// double[1] = key's objective / 2 for float inputs; owner = world - 1 - low code.  A key equal to empty_key (what a rank
// without envs contributes) that survived the reduction means no rank had an env: flag bit 1.
__global__ void k_key_unpack(const int64_t* __restrict__ key, int rank_bits, int64_t world, int as_float, void* __restrict__ obj_out,
                             int64_t* __restrict__ owner_out, int64_t empty_key, int32_t* __restrict__ flag) {
    if (threadIdx.x) return;
    const int64_t k = key[0];
    const int64_t obj = k >> rank_bits;                                   // arithmetic: negative objectives survive
    const int64_t low = k & ((1ll << rank_bits) - 1);
    if (as_float) ((double*)obj_out)[0] = (double)obj * 0.5;
    else ((int64_t*)obj_out)[0] = obj;
    if (owner_out) owner_out[0] = world - 1 - low;
    if (flag && k == empty_key) atomicOr(flag, 2);
}

// C2 without a host read (round 6): after the key's all-reduce every rank writes ITS candidate message -- the winner (the rank whose
// low code survived in the reduced key) its global env index and its row bit-packed, everyone else zeros -- and a SUM all-reduce of
// the messages is the winner's message on every rank.  msg: uint8 [8 + ceil(N / 8)] = { global index, little endian | bit k of
// byte j = x[8 j + k] }.  xs is this rank's [B, N] (row index[0]) or a single row (B = 1, index ignored).
__global__ __launch_bounds__(256) void k_winner_message(const uint8_t* __restrict__ xs, int64_t B, int64_t N, const int64_t* __restrict__ index,
                                                        const int64_t* __restrict__ key, int rank_bits, int64_t my_low_code,
                                                        int64_t env_offset, uint8_t* __restrict__ msg, int64_t msg_bytes) {
    const bool mine = xs != nullptr && (key[0] & ((1ll << rank_bits) - 1)) == my_low_code;
    const int64_t li = (B > 1 && index) ? index[0] : 0;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < msg_bytes; j += (int64_t)gridDim.x * blockDim.x) {
        uint8_t v = 0;
        if (mine) {
            if (j < 8) {
                v = (uint8_t)((uint64_t)((index ? index[0] : 0) + env_offset) >> (8 * j));
            } else {
                const int64_t n0 = (j - 8) * 8;
                const uint8_t* row = xs + li * N;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (n0 + k < N && row[n0 + k]) v |= (uint8_t)(1u << k);
            }
        }
        msg[j] = v;
    }
}
__global__ __launch_bounds__(256) void k_winner_unpack(const uint8_t* __restrict__ msg, int64_t N, uint8_t* __restrict__ x_out,
                                                       int64_t* __restrict__ index_out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && index_out) {
        uint64_t g = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) g |= (uint64_t)msg[j] << (8 * j);
        index_out[0] = (int64_t)g;
    }
    if (x_out)
        for (int64_t n = t; n < N; n += (int64_t)gridDim.x * blockDim.x) x_out[n] = (msg[8 + (n >> 3)] >> (n & 7)) & 1u;
}

}  // namespace rls

using namespace rls;

extern "C" int rls_winner_message(const uint8_t* xs, int64_t B, int64_t N, const int64_t* index, const int64_t* key, int32_t rank_bits,
                                  int64_t my_low_code, int64_t env_offset, uint8_t* msg, void* stream) {
    RLS_REQUIRE(N >= 1 && B >= 0, RLS_EINVAL, "bad sizes B=%lld N=%lld", (long long)B, (long long)N);
    RLS_REQUIRE(key && msg, RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE(B == 0 || xs, RLS_EINVAL, "xs is NULL");
    RLS_REQUIRE(rank_bits >= 0 && rank_bits < 32 && my_low_code >= 0 && my_low_code < (1ll << rank_bits), RLS_EINVAL, "bad key layout");
    const int64_t nb = 8 + (N + 7) / 8;
    hipLaunchKernelGGL(k_winner_message, dim3((unsigned)grid_for(nb, 256)), dim3(256), 0, as_stream(stream), B > 0 ? xs : nullptr, B, N, index,
                       key, (int)rank_bits, my_low_code, env_offset, msg, nb);
    return check_launch("k_winner_message");
}

extern "C" int rls_winner_unpack(const uint8_t* msg, int64_t N, uint8_t* x_out, int64_t* index_out, void* stream) {
    RLS_REQUIRE(N >= 1 && msg, RLS_EINVAL, "bad arguments");
    hipLaunchKernelGGL(k_winner_unpack, dim3((unsigned)grid_for(N, 256)), dim3(256), 0, as_stream(stream), msg, N, x_out, index_out);
    return check_launch("k_winner_unpack");
}

extern "C" int rls_key_unpack(const int64_t* key, int32_t rank_bits, int64_t world, int as_float, void* obj_out, int64_t* owner_out,
                              int64_t empty_key, int32_t* flag, void* stream) {
    RLS_REQUIRE(key && obj_out, RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE(rank_bits >= 0 && rank_bits < 32 && world >= 1 && world <= (1ll << rank_bits), RLS_EINVAL, "bad key layout");
    hipLaunchKernelGGL(k_key_unpack, dim3(1), dim3(64), 0, as_stream(stream), key, (int)rank_bits, world, as_float, obj_out, owner_out,
                       empty_key, flag);
    return check_launch("k_key_unpack");
}

extern "C" int rls_best_key(const void* vs, int vs_kind, int64_t B, int32_t rank_bits, int64_t low_code, int64_t limit, int64_t* key,
                            int64_t* index, int32_t* flag, void* stream) {
    RLS_REQUIRE(B >= 1, RLS_EINVAL, "B < 1");
    RLS_REQUIRE(vs && key && flag, RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE(vs_kind >= 0 && vs_kind <= 3, RLS_EINVAL, "vs_kind must be 0 (int64), 1 (float32), 2 (float64) or 3 (int32)");
    RLS_REQUIRE(rank_bits >= 0 && rank_bits < 32 && low_code >= 0 && low_code < (1ll << rank_bits) && limit > 0 &&
                    limit <= (1ll << (62 - rank_bits)), RLS_EINVAL, "bad key layout");
    hipStream_t s = as_stream(stream);
    const dim3 grid(1), block(kTrackThreads);
    if (vs_kind == 0)
        hipLaunchKernelGGL(k_best_key<int64_t>, grid, block, 0, s, (const int64_t*)vs, B, 0, (int)rank_bits, low_code, limit, key, index, flag);
    else if (vs_kind == 3)
        hipLaunchKernelGGL(k_best_key<int32_t>, grid, block, 0, s, (const int32_t*)vs, B, 0, (int)rank_bits, low_code, limit, key, index, flag);
    else if (vs_kind == 1)
        hipLaunchKernelGGL(k_best_key<float>, grid, block, 0, s, (const float*)vs, B, 1, (int)rank_bits, low_code, limit, key, index, flag);
    else
        hipLaunchKernelGGL(k_best_key<double>, grid, block, 0, s, (const double*)vs, B, 1, (int)rank_bits, low_code, limit, key, index, flag);
    return check_launch("k_best_key");
}

extern "C" int rls_best_update(const uint8_t* xs, const void* vs, int vs_kind, int64_t B, int64_t N, int if_maximize,
                               uint8_t* best_x, double* best_v, uint8_t* improved, double* log_v, int64_t log_index,
                               int force, void* stream) {
    RLS_REQUIRE(B >= 1 && N >= 1, RLS_EINVAL, "bad sizes B=%lld N=%lld", (long long)B, (long long)N);
    RLS_REQUIRE(xs && vs && best_x && best_v && improved, RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE(vs_kind >= 0 && vs_kind <= 2, RLS_EINVAL, "vs_kind must be 0 (int64), 1 (float32) or 2 (float64)");
    RLS_REQUIRE(!log_v || log_index >= 0, RLS_EINVAL, "negative log_index");
    hipStream_t s = as_stream(stream);
    const dim3 grid(1), block(kTrackThreads);
    if (vs_kind == 0)
        hipLaunchKernelGGL(k_best_update<int64_t>, grid, block, 0, s, xs, (const int64_t*)vs, B, N, if_maximize, best_x, best_v,
                           improved, log_v, log_index, force);
    else if (vs_kind == 1)
        hipLaunchKernelGGL(k_best_update<float>, grid, block, 0, s, xs, (const float*)vs, B, N, if_maximize, best_x, best_v,
                           improved, log_v, log_index, force);
    else
        hipLaunchKernelGGL(k_best_update<double>, grid, block, 0, s, xs, (const double*)vs, B, N, if_maximize, best_x, best_v,
                           improved, log_v, log_index, force);
    return check_launch("k_best_update");
}

// K4: the gym step -- the headline kernel (HBM-bound: read state, write next state).
//
// One wave owns EPW consecutive envs, i.e. one contiguous run of EPW*N spins.
//   gain : the wave's lanes split the action node's CSR row, fetch the neighbours' spins of
//          that env and reduce.  Unweighted graphs need one ballot + popcount (cutdeg c,
//          gain = deg - 2c); weighted graphs a wave shuffle reduction.
//   emit : the run is streamed with 16-byte vectors, the flipped spin patched in flight.
//
// Three emit structures (which one runs where: see the launcher):
//   MODE 0  gain first (gathers from global), then copy           -- first version
//   MODE 1  issue the copy's loads first, compute the gain while they fly, patch, store
//   MODE 2  stage the run in LDS with direct global->LDS loads (no VGPR round trip), gather
//           the neighbours from LDS, patch the byte in LDS, stream LDS -> HBM
#include "rls_tile.h"
#include <cstdlib>

namespace rls {

constexpr int kStepPad = 128;   // spare bytes per wave's LDS stage: MODE 2 shifts a run by up to 112 bytes (see there)

template <typename T> __device__ __forceinline__ T spin_flip(T v);
template <> __device__ __forceinline__ uint8_t spin_flip<uint8_t>(uint8_t v) { return v == 0 ? 1 : 0; }
template <> __device__ __forceinline__ float spin_flip<float>(float v) { return v == 0.0f ? 1.0f : 0.0f; }  // logical_not

// cut gain of flipping node a, neighbours' spins read through `row` (global or LDS)
template <typename T, bool WEIGHTED>
__device__ __forceinline__ int flip_gain(const T* row, int64_t a, const int32_t* __restrict__ rowptr,
                                         const int32_t* __restrict__ col, const int32_t* __restrict__ wgt,
                                         int lane) {
    const int r0 = rowptr[a], r1 = rowptr[a + 1];
    const bool xa = spin_is_set(row[a]);
    int acc = 0;
    for (int j = r0 + lane; j < r1; j += kWave) {
        const bool xn = spin_is_set(row[col[j]]);
        if constexpr (WEIGHTED) acc += (xn == xa) ? wgt[j] : -wgt[j];
        else acc += (xn != xa) ? 1 : 0;
    }
    if constexpr (WEIGHTED) {
        return wave_sum_i32(acc);
    } else {
        int c;
        if (r1 - r0 <= kWave) c = __popcll(ballot64(acc != 0));
        else c = wave_sum_i32(acc);
        return (r1 - r0) - 2 * c;
    }
}

template <bool NT, typename V> __device__ __forceinline__ V ld_vec(const V* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT, typename V> __device__ __forceinline__ void st_vec(V* p, V v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// s_waitcnt vmcnt(k) for a wave-uniform RUN-TIME k (the instruction takes an immediate): a scalar jump over 64 one-line cases
__device__ __forceinline__ void wait_vmcnt_le(int k) {
    k = __builtin_amdgcn_readfirstlane(k);
#define RLS_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
#define RLS_W8(b) RLS_W(b + 0) RLS_W(b + 1) RLS_W(b + 2) RLS_W(b + 3) RLS_W(b + 4) RLS_W(b + 5) RLS_W(b + 6) RLS_W(b + 7)
    switch (k) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        RLS_W(1) RLS_W(2) RLS_W(3) RLS_W(4) RLS_W(5) RLS_W(6) RLS_W(7)
        RLS_W(8) RLS_W(9) RLS_W(10) RLS_W(11) RLS_W(12) RLS_W(13) RLS_W(14) RLS_W(15)
        RLS_W(16) RLS_W(17) RLS_W(18) RLS_W(19) RLS_W(20) RLS_W(21) RLS_W(22) RLS_W(23)
        RLS_W(24) RLS_W(25) RLS_W(26) RLS_W(27) RLS_W(28) RLS_W(29) RLS_W(30) RLS_W(31)
        RLS_W(32) RLS_W(33) RLS_W(34) RLS_W(35) RLS_W(36) RLS_W(37) RLS_W(38) RLS_W(39)
        RLS_W(40) RLS_W(41) RLS_W(42) RLS_W(43) RLS_W(44) RLS_W(45) RLS_W(46) RLS_W(47)
        RLS_W(48) RLS_W(49) RLS_W(50) RLS_W(51) RLS_W(52) RLS_W(53) RLS_W(54) RLS_W(55)
        RLS_W(56) RLS_W(57) RLS_W(58) RLS_W(59) RLS_W(60) RLS_W(61) RLS_W(62)
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
#undef RLS_W8
#undef RLS_W
}

template <typename T, int EPW, int MODE, bool EMIT, bool VEC, bool WEIGHTED, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_maxcut_step(const T* __restrict__ xin, T* __restrict__ xout,
                                                     int64_t B, int64_t N,
                                                     const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ col,
                                                     const int32_t* __restrict__ wgt,
                                                     const int64_t* __restrict__ action,
                                                     int32_t* __restrict__ obj, float* __restrict__ reward,
                                                     float* __restrict__ cur, float* __restrict__ done,
                                                     float done_value, int align_lines) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    using V = typename SpinVec<T>::type;
    constexpr int PER = SpinVec<T>::n;
    // a wave takes runs wave, wave + (waves of the grid), ...: with a grid that covers every run this is one trip (the plain
    // launch); with a grid that just fills the chip -- MODE 2 on long rows -- the wave is persistent: the stores of run k are
    // still in flight while the loads of run k + 1 are issued (the LDS stage is free once its ds_reads have returned), and no
    // workgroup is dispatched or retired mid-kernel
    const int64_t wave_stride = (int64_t)gridDim.x * (blockDim.x / kWave);
  for (int64_t wave = (int64_t)blockIdx.x * (blockDim.x / kWave) + wib; wave * EPW < B; wave += wave_stride) {
    const int64_t b0 = wave * EPW;
    const int nenv = (int)((B - b0) < EPW ? (B - b0) : EPW);

    int64_t act[EPW];
#pragma unroll
    for (int k = 0; k < EPW; ++k) {
        // an action outside [0, N) leaves its env untouched and reports reward = NaN (the reference raises an
        // IndexError there; a kernel cannot, but it must not read or write another env's bytes)
        const int64_t a = (k < nenv) ? action[b0 + k] : -1;
        act[k] = ((uint64_t)a < (uint64_t)N) ? a : -1;
    }

    int my_delta = 0;  // lane k keeps env k's gain
    auto publish = [&]() {
        if (lane < nenv) {
            const int64_t b = b0 + lane;
            bool ok = false;
#pragma unroll
            for (int k = 0; k < EPW; ++k) if (lane == k) ok = act[k] >= 0;
            const int v = obj[b] + my_delta;
            obj[b] = v;
            reward[b] = ok ? (float)my_delta : __builtin_nanf("");
            if (cur) cur[b] = (float)v;
            if (done) done[b] = done_value;
        }
    };

    if constexpr (!EMIT) {
#pragma unroll
        for (int k = 0; k < EPW; ++k)
            if (act[k] >= 0) {
                const int d = flip_gain<T, WEIGHTED>(xin + (b0 + k) * N, act[k], rowptr, col, wgt, lane);
                if (lane == k) my_delta = d;
            }
        publish();
        if (lane < nenv) {  // in place: only the flipped spins are written
            int64_t a = -1;
#pragma unroll
            for (int k = 0; k < EPW; ++k) if (lane == k) a = act[k];
            if (a >= 0) {
                T* p = xout + (b0 + lane) * N + a;
                *p = spin_flip<T>(*p);
            }
        }
        return;
    } else if constexpr (!VEC) {
#pragma unroll
        for (int k = 0; k < EPW; ++k)
            if (act[k] >= 0) {
                const int d = flip_gain<T, WEIGHTED>(xin + (b0 + k) * N, act[k], rowptr, col, wgt, lane);
                if (lane == k) my_delta = d;
            }
        publish();
        const T* src = xin + b0 * N;
        T* dst = xout + b0 * N;
        const int64_t nel = (int64_t)nenv * N;
        for (int64_t i = lane; i < nel; i += kWave) {
            T v = src[i];
#pragma unroll
            for (int k = 0; k < EPW; ++k)
                if (act[k] >= 0 && i == (int64_t)k * N + act[k]) v = spin_flip<T>(v);
            dst[i] = v;
        }
        continue;
    } else {
        // VEC: a full run of EPW rows is a whole number of 16-byte vectors and starts 16-byte aligned (the rows
        // themselves need not be: N = 1000 bytes works with EPW = 4).  Only the last run of a batch can be
        // short; its nel % PER trailing elements go element-wise (tail_copy).
        const V* src = reinterpret_cast<const V*>(xin + b0 * N);
        V* dst = reinterpret_cast<V*>(xout + b0 * N);
        const int64_t nel = (int64_t)nenv * N;
        const int64_t nvec = nel / PER;
        auto tail_copy = [&]() {   // register modes: copy + flip the elements past the last full vector
            for (int64_t i = nvec * PER + lane; i < nel; i += kWave) {
                T v = xin[b0 * N + i];
#pragma unroll
                for (int k = 0; k < EPW; ++k)
                    if (act[k] >= 0 && i == (int64_t)k * N + act[k]) v = spin_flip<T>(v);
                xout[b0 * N + i] = v;
            }
        };
        int64_t fvec[EPW];
        int fidx[EPW];
#pragma unroll
        for (int k = 0; k < EPW; ++k) {
            const int64_t rel = (act[k] < 0) ? -1 : (int64_t)k * N + act[k];
            fvec[k] = rel < 0 ? -1 : rel / PER;
            fidx[k] = (int)(rel < 0 ? 0 : rel % PER);
        }

        if constexpr (MODE == 0) {
#pragma unroll
            for (int k = 0; k < EPW; ++k)
                if (act[k] >= 0) {
                    const int d = flip_gain<T, WEIGHTED>(xin + (b0 + k) * N, act[k], rowptr, col, wgt, lane);
                    if (lane == k) my_delta = d;
                }
            publish();
#pragma unroll 4
            for (int64_t i = lane; i < nvec; i += kWave) {
                V v = ld_vec<NTL>(src + i);
#pragma unroll
                for (int k = 0; k < EPW; ++k)
                    if (i == fvec[k]) v = SpinVec<T>::flip_at(v, fidx[k]);
                st_vec<NTS>(dst + i, v);
            }
            tail_copy();
        } else if constexpr (MODE == 1) {
            // (two batches per wave in flight -- the next one requested before this one is stored -- were measured in round 6: N = 10^5,
            // 4096 envs 0.574 -> 0.548 of HBM: 133 registers, three waves per SIMD instead of four)
            constexpr int BATCH = 8;  // 8 x 16 B per lane = 8 KB per wave in flight
            V buf[BATCH];
            bool first = true;
            for (int64_t base = 0; base < nvec; base += BATCH * kWave) {
#pragma unroll
                for (int q = 0; q < BATCH; ++q) {
                    const int64_t i = base + q * kWave + lane;
                    if (i < nvec) buf[q] = ld_vec<NTL>(src + i);
                }
                if (first) {  // gains while the first batch is in flight
                    first = false;
#pragma unroll
                    for (int k = 0; k < EPW; ++k)
                        if (act[k] >= 0) {
                            const int d = flip_gain<T, WEIGHTED>(xin + (b0 + k) * N, act[k], rowptr, col, wgt, lane);
                            if (lane == k) my_delta = d;
                        }
                    publish();
                }
#pragma unroll
                for (int q = 0; q < BATCH; ++q) {
                    const int64_t i = base + q * kWave + lane;
                    if (i < nvec) {
                        V v = buf[q];
#pragma unroll
                        for (int k = 0; k < EPW; ++k)
                            if (i == fvec[k]) v = SpinVec<T>::flip_at(v, fidx[k]);
                        st_vec<NTS>(dst + i, v);
                    }
                }
            }
            tail_copy();
        } else if constexpr (MODE == 3) {
            // MODE 3: MODE 2's staging with the stores CHASING the loads.  The flip needs no gain (a gym step flips its action node
            // whatever the reward), so 1 KB piece j of the run can leave as soon as it has landed: the wave's stores are in flight
            // beside its own later loads instead of starting when the last load is back.  The vector-memory counter completes in
            // order; the ops younger than load j are the later loads, the neighbour-id loads and the stores of the pieces before
            // j -- (n - 1 - j) + E + j = n - 1 + E of them whatever j, so ONE wait value serves every trip.  The gain is read from
            // the staged (unpatched: the flip is applied in registers) row afterwards.
            // (instruction boundaries on the global side's cache lines, as in MODE 2 below -- when the input and the output run start
            // the same distance into a line, so that store piece j still needs exactly load piece j: the wait arithmetic stands)
            unsigned char* region = smem + (size_t)wib * ((size_t)EPW * N * sizeof(T) + kStepPad);
            const int h_in = (int)((reinterpret_cast<uintptr_t>(src) >> 4) & 7), h_out = (int)((reinterpret_cast<uintptr_t>(dst) >> 4) & 7);
            const int h = (align_lines && h_in == h_out) ? h_in : 0;
            V* region_v = reinterpret_cast<V*>(region);
            V* stage_v = region_v + h;
            T* stage = reinterpret_cast<T*>(stage_v);
            const int nch = (int)((nvec + h + kWave - 1) / kWave);
            for (int64_t slot0 = 0; slot0 < nvec + h; slot0 += kWave) {
                const int64_t i = slot0 + lane - h;
                if (i >= 0 && i < nvec) glds16<NTL>(src + i, region_v + slot0);
            }
            if (nvec * PER < nel) {   // a short last run: its elements past the last whole vector are staged by hand (the gain reads them)
                for (int64_t i = nvec * PER + lane; i < nel; i += kWave) stage[i] = xin[b0 * N + i];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            int r0[EPW], deg[EPW], nb[EPW], nlo = 0;
#pragma unroll
            for (int k = 0; k < EPW; ++k) {
                r0[k] = 0; deg[k] = 0; nb[k] = 0;
                if (act[k] >= 0) {
                    r0[k] = rowptr[act[k]];
                    deg[k] = rowptr[act[k] + 1] - r0[k];
                    // hand-issued (and hand-waited, below): a compiler-tracked load pending at the store loop's head makes the
                    // compiler drain the vector-memory counter there -- which waits for the whole run and ends the chase.
                    // Only for a node that HAS neighbours (deg is wave-uniform, so `nlo` stays exact): an isolated node at the
                    // end of the CSR has r0 == nnz, and col + nnz is one entry past the array (col may be NULL when nnz == 0)
                    if (deg[k] > 0) {
                        const int32_t* pn = col + r0[k] + (lane < deg[k] ? lane : 0);
                        asm volatile("global_load_dword %0, %1, off" : "=v"(nb[k]) : "v"(pn) : "memory");
                        ++nlo;
                    }
                }
            }
            const int keep = nch - 1 + nlo;
            for (int j = 0; j < nch; ++j) {
                wait_vmcnt_le(keep);
                const int64_t i = (int64_t)j * kWave + lane - h;
                if (i >= 0 && i < nvec) {
                    V v = stage_v[i];
#pragma unroll
                    for (int k = 0; k < EPW; ++k)
                        if (i == fvec[k]) v = SpinVec<T>::flip_at(v, fidx[k]);
                    st_vec<NTS>(dst + i, v);
                }
            }
            tail_copy();                                                     // (a short last run only)
            wait_vmcnt_le(nch < 62 ? nch : 62);                              // the neighbour ids are older than every store
            // the compiler does not know that nb[k] only became valid at the wait above: every use below goes through a value it
            // must treat as (re)defined HERE, after the wait
#pragma unroll
            for (int k = 0; k < EPW; ++k) asm volatile("" : "+v"(nb[k]) :: "memory");
#pragma unroll
            for (int k = 0; k < EPW; ++k)
                if (act[k] >= 0) {
                    const T* row = stage + (int64_t)k * N;
                    const bool xa = spin_is_set(row[act[k]]);
                    int d;
                    if (deg[k] <= kWave) {
                        const bool on = lane < deg[k];
                        const bool xn = on ? spin_is_set(row[nb[k]]) : xa;
                        d = deg[k] - 2 * __popcll(ballot64(xn != xa));
                    } else {
                        d = flip_gain<T, WEIGHTED>(row, act[k], rowptr, col, wgt, lane);
                    }
                    if (lane == k) my_delta = d;
                }
            publish();
        } else {
            // MODE 2: LDS staged.  Per-wave region of EPW*N*sizeof(T) bytes (16-byte multiple) + 128.
            // A run starts 16-byte aligned, not 128-byte aligned (rows of 2000 or 10 000 bytes: a run starts 0, 16, ... 112 bytes
            // into a cache line), and a 1 KB wave-instruction that starts mid-line touches NINE lines instead of eight -- the
            // texture path prices lines: rows of 12 000 / 15 984 / 16 016 bytes ran at 0.63 of HBM, rows of 12 288 / 16 000 /
            // 16 128 at 0.71 - 0.73.  So the instruction boundaries are put on the lines of the GLOBAL side: vector i of the run
            // lives in LDS slot i + h (h = 16-byte units between the line's start and the run's, 0..7; the region has 128
            // spare bytes), the first load instruction carries 64 - h vectors, every later one starts on a line; the stores likewise
            // with the output's own h.
            unsigned char* region = smem + (size_t)wib * ((size_t)EPW * N * sizeof(T) + kStepPad);
            const int h_in = align_lines ? (int)((reinterpret_cast<uintptr_t>(src) >> 4) & 7) : 0;
            const int h_out = align_lines ? (int)((reinterpret_cast<uintptr_t>(dst) >> 4) & 7) : 0;
            V* region_v = reinterpret_cast<V*>(region);
            V* stage_v = region_v + h_in;
            T* stage = reinterpret_cast<T*>(stage_v);
            for (int64_t slot0 = 0; slot0 < nvec + h_in; slot0 += kWave) {
                const int64_t i = slot0 + lane - h_in;
                if (i >= 0 && i < nvec) glds16<NTL>(src + i, region_v + slot0);  // LDS dst = wave base + lane*16
            }
            for (int64_t i = nvec * PER + lane; i < nel; i += kWave) stage[i] = xin[b0 * N + i];   // short last run only
            // while the rows fly: CSR row bounds (scalar loads) and this lane's neighbour id / weight
            int r0[EPW], deg[EPW], nb[EPW], wv[EPW];
#pragma unroll
            for (int k = 0; k < EPW; ++k) {
                r0[k] = 0; deg[k] = 0; nb[k] = 0; wv[k] = 0;
                if (act[k] >= 0) {
                    r0[k] = rowptr[act[k]];
                    deg[k] = rowptr[act[k] + 1] - r0[k];
                    if (lane < deg[k]) {
                        nb[k] = col[r0[k] + lane];
                        if constexpr (WEIGHTED) wv[k] = wgt[r0[k] + lane];
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < EPW; ++k)
                if (act[k] >= 0) {
                    const T* row = stage + (int64_t)k * N;
                    const bool xa = spin_is_set(row[act[k]]);
                    int d;
                    if (deg[k] <= kWave) {
                        const bool on = lane < deg[k];
                        const bool xn = on ? spin_is_set(row[nb[k]]) : xa;
                        if constexpr (WEIGHTED) d = wave_sum_i32(on ? ((xn == xa) ? wv[k] : -wv[k]) : 0);
                        else d = deg[k] - 2 * __popcll(ballot64(xn != xa));
                    } else {
                        d = flip_gain<T, WEIGHTED>(row, act[k], rowptr, col, wgt, lane);
                    }
                    if (lane == k) my_delta = d;
                }
            publish();
            __builtin_amdgcn_wave_barrier();
            if (lane < nenv) {
                int64_t a = -1;
#pragma unroll
                for (int k = 0; k < EPW; ++k) if (lane == k) a = act[k];
                if (a >= 0) {
                    T* p = stage + (int64_t)lane * N + a;
                    *p = spin_flip<T>(*p);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll 4
            for (int64_t slot0 = 0; slot0 < nvec + h_out; slot0 += kWave) {
                const int64_t i = slot0 + lane - h_out;
                if (i >= 0 && i < nvec) st_vec<NTS>(dst + i, stage_v[i]);
            }
            for (int64_t i = nvec * PER + lane; i < nel; i += kWave) xout[b0 * N + i] = stage[i];
        }
    }
  }   // runs of this wave
}

// tuning knobs (rls_tuning_set; no environment variable is read by the production library): nts = -1 automatic | 0 | 1
// (nontemporal stores), epw / wpb = 0: automatic, persist = -1 automatic | 0 one run per wave | k: k workgroup rounds per CU
// resident, waves loop; chase = -1 automatic | 0 MODE 2 (stores after the last load) | 1 MODE 3 (stores chase the loads);
// align = 0: load / store instructions start where the run starts (before round 4's fix)
struct StepKnobs { int nts, epw, wpb, persist, chase, align; };
static StepKnobs step_knobs() {
    return StepKnobs{(int)knob(KN_STEP_NTS, -1), (int)knob(KN_STEP_EPW, 0), (int)knob(KN_STEP_WPB, 0), (int)knob(KN_STEP_PERSIST, -1),
                     (int)knob(KN_STEP_CHASE, -1), (int)knob(KN_STEP_ALIGN, 1)};
}

}  // namespace rls

using namespace rls;

// Structures in production (round 3; the A/B history is in DESIGN_HISTORY.md section 6):
//   emit, vectorisable runs   MODE 2: the run staged in LDS by LDS-DMA with nontemporal loads (every input byte is read once),
//                             up to the whole 160 KB of a CU per workgroup (f32 rows of 10^4 nodes: 2 waves x 40 KB);
//                             stores nontemporal when the batch is larger than the 256 MB Infinity Cache (nothing of it
//                             can still be there when the next step reads it: +3 points at G70 size), plain otherwise
//                             (the next step's reads hit what this step wrote: +6 points at G22 size)
//   emit, rows too long       MODE 1 (register batches, gathers from global), nontemporal stores
//   emit, odd row lengths     MODE 0 element-wise
//   in place                  O(deg) bytes
// Unweighted graphs only: the reference's gym env counts cut EDGES (envs/env_PPO.py:108-121).
extern "C" int rls_maxcut_step(const rls_graph* g, const void* x_in, void* x_out, int spin_bytes, int64_t B,
                               const int64_t* action, int32_t* obj, float* reward, float* cur, float* done,
                               float done_value, void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(B >= 0, RLS_EINVAL, "B < 0");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(x_in && x_out && action && obj && reward, RLS_EINVAL, "x_in/x_out/action/obj/reward is NULL");
    RLS_REQUIRE(spin_bytes == 1 || spin_bytes == 4, RLS_EINVAL, "spin_bytes must be 1 or 4");
    RLS_REQUIRE(!g->wgt, RLS_EUNSUPPORTED, "rls_maxcut_step counts cut edges (env_PPO.py:108-121): build the graph without weights");
    const int64_t N = g->num_nodes;
    const bool emit = (x_in != x_out);
    RLS_REQUIRE(!emit || (const char*)x_in + (size_t)B * N * spin_bytes <= (const char*)x_out ||
                    (const char*)x_out + (size_t)B * N * spin_bytes <= (const char*)x_in,
                RLS_EINVAL, "x_in and x_out overlap partially");
    const StepKnobs knobs = step_knobs();
    // envs per wave: the largest of {8,4,2,1} whose staged run fits ~8 KB of LDS per wave (G22: 4) ...
    int epw_auto = 8;
    while (epw_auto > 1 && (int64_t)epw_auto * N * spin_bytes > 8192) epw_auto >>= 1;
    // ... but at least enough rows for a run to be a whole number of 16-byte vectors (N = 1000 bytes: >= 2 rows,
    // N = 7003 bytes: none of {1..8} works and the element-wise kernel runs)
    while (epw_auto < 8 && ((int64_t)epw_auto * N * spin_bytes) % 16 != 0) epw_auto <<= 1;
    const int epw = knobs.epw ? knobs.epw : (emit ? epw_auto : 4);   // in place: nothing is staged
    // flat runs of EPW rows start 16-byte aligned when one RUN is a multiple of 16 bytes
    const bool vec = ((((uintptr_t)x_in) | ((uintptr_t)x_out)) & 15) == 0 && ((int64_t)epw * N * spin_bytes) % 16 == 0;
    const size_t run_bytes = (size_t)epw * N * spin_bytes;
    // waves per workgroup: 4, fewer while the staged runs of a workgroup would take more than half a CU's LDS
    int waves_per_block = knobs.wpb ? knobs.wpb : 4;
    if (!knobs.wpb && emit && vec && !knob_on(KN_STEP_NOSTAGE))
        while (waves_per_block > 1 && (size_t)waves_per_block * (run_bytes + kStepPad) > (size_t)kLdsBytes / 2) waves_per_block >>= 1;
    hipStream_t s = as_stream(stream);
    // rows so long that ONE staged run is all a CU's LDS holds (N > ~81 900 bytes) go unstaged: a single wave per CU alternates
    // between filling and draining its run (tools/timing/k4_long_rows.py: N = 10^5, 4096 envs staged 0.46 of HBM, unstaged 0.57 before
    // its double batch); two runs per CU (N = 80 000) still stage at 0.70-0.71
    const bool lone_run = !knobs.wpb && emit && vec && waves_per_block == 1 && 2 * (run_bytes + kStepPad) > (size_t)kLdsBytes;
    if (lone_run) waves_per_block = 4;
    const bool staged = emit && vec && !lone_run && (size_t)waves_per_block * (run_bytes + kStepPad) <= (size_t)kLdsBytes &&
                        !knob_on(KN_STEP_NOSTAGE);
    const size_t lds = staged ? (size_t)waves_per_block * (run_bytes + kStepPad) : 0;
    int64_t nblocks = ceil_div(ceil_div(B, epw), waves_per_block);
    if (staged && knobs.persist != 0) {
        // persistent form: as many workgroups as are resident at once (LDS-limited, at most 8 per CU), each wave looping over runs
        const int64_t per_cu = (int64_t)((size_t)kLdsBytes / lds) < 8 ? (int64_t)((size_t)kLdsBytes / lds) : 8;
        const int64_t resident = (int64_t)num_cus() * per_cu;
        const bool want = knobs.persist > 0 || (knobs.persist < 0 && false);
        if (want && nblocks > resident) nblocks = resident;
    }
    const dim3 grid((unsigned)nblocks), block(waves_per_block * kWave);
    // MODE 3 needs its run's pieces + the neighbour-id loads to stay within the 6-bit counter: <= 54 KB runs
    const int64_t nch_run = ceil_div((int64_t)run_bytes, 16 * kWave);
    // measured (tools/sweep_step.py, SW_CHASE=0,1): f32 rows gain 3 - 6 % (N = 10^4: 0.674 -> 0.701; N = 2000: 0.707 -> 0.752 of
    // 8 TB/s), 1-byte rows nothing (0.694 / 0.694, 0.738 / 0.739): their runs are 8 - 10 pieces, back before the first could leave
    const bool chase = staged && nch_run + 1 + epw <= 62 && (knobs.chase > 0 || (knobs.chase < 0 && spin_bytes == 4));   // (+ 1: a shifted run's extra piece)
    const bool nts = knobs.nts >= 0 ? knobs.nts != 0 : ((size_t)B * N * spin_bytes > ((size_t)256 << 20));

#define LAUNCH_STEP_E(T, EPW, MODE, EMIT, VEC, NTL, NTS)                                                       \
    do {                                                                                                       \
        auto kern = k_maxcut_step<T, EPW, MODE, EMIT, VEC, false, NTL, NTS>;                                   \
        if (lds > 64 * 1024)                                                                                   \
            ensure_dyn_lds((const void*)kern, lds); \
        hipLaunchKernelGGL(kern, grid, block, lds, s, (const T*)x_in, (T*)x_out, B, N, g->rowptr, g->col, g->wgt, \
                           action, obj, reward, cur, done, done_value, knobs.align);                           \
    } while (0)
#define LAUNCH_STEP(T, MODE, EMIT, VEC, NTL, NTS)                           \
    do {                                                                    \
        if (epw == 2) LAUNCH_STEP_E(T, 2, MODE, EMIT, VEC, NTL, NTS);       \
        else if (epw == 8) LAUNCH_STEP_E(T, 8, MODE, EMIT, VEC, NTL, NTS);  \
        else if (epw == 1) LAUNCH_STEP_E(T, 1, MODE, EMIT, VEC, NTL, NTS);  \
        else LAUNCH_STEP_E(T, 4, MODE, EMIT, VEC, NTL, NTS);                \
    } while (0)
#define DISPATCH_T(T)                                                                       \
    do {                                                                                    \
        if (!emit) LAUNCH_STEP_E(T, 4, 0, false, false, false, false);                      \
        else if (!vec) LAUNCH_STEP(T, 0, true, false, false, false);                        \
        else if (!staged) LAUNCH_STEP(T, 1, true, true, false, true);                       \
        else if (chase && nts) LAUNCH_STEP(T, 3, true, true, true, true);                   \
        else if (chase) LAUNCH_STEP(T, 3, true, true, true, false);                         \
        else if (nts) LAUNCH_STEP(T, 2, true, true, true, true);                            \
        else LAUNCH_STEP(T, 2, true, true, true, false);                                    \
    } while (0)
    RLS_REQUIRE(emit || epw == 4, RLS_EINVAL, "the in-place step runs 4 envs per wave");
    if (spin_bytes == 1) DISPATCH_T(uint8_t);
    else DISPATCH_T(float);
#undef DISPATCH_T
#undef LAUNCH_STEP
#undef LAUNCH_STEP_E
    return check_launch("k_maxcut_step");
}

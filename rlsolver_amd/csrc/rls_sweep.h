// The greedy single-flip sweep over a 64-env bit tile, as a device function shared by the stand-alone
// K5 kernel and the fused local-search kernel.  One wave; see rls_maxcut.hip for the design notes.
#pragma once
#include "rls_ring.h"
#include "rls_tile.h"

namespace rls {

// words[0..N) = tile (words[N] must be 0: the sentinel), rp[0..N] = CSR rowptr in LDS, ring = kRing
// int32 entries of LDS.  `wbytes` is the byte address of words[0].  Returns this lane's total gain.
// The caller must have made words / rp visible (barrier) and must not have ring loads in flight.
__device__ __forceinline__ int64_t sweep_tile(uint64_t* words, const int32_t* rp, int32_t* ring,
                                              const int32_t* __restrict__ col, int64_t nnz, int64_t N, int lane) {
    const unsigned char* wbytes = reinterpret_cast<const unsigned char*>(words);
    int64_t F;
    ring_prime(col, nnz, F, ring, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int sh = lane & 31;
    const uint32_t half4 = (uint32_t)(lane >> 5) * 4u;   // byte offset of this lane's dword inside a word
    int64_t gain = 0;
    // software pipeline: node i+1's row bounds and this lane's ring entry are fetched while node i runs
    int r0 = rp[0], r1 = rp[1];
    const int sentinel = (int)N;   // words[N] == 0
    int my_nb = (r0 + lane < r1) ? ring[(r0 + lane) & (kRing - 1)] : sentinel;
    for (int64_t i = 0; i < N; ++i) {
        ring_advance(col, nnz, F, r0, ring, lane);     // wave-uniform; refills once per ~kRefill entries
        const int r2 = (i + 2 <= N) ? rp[i + 2] : r1;
        const int nxt_nb = (r1 + lane < r2) ? ring[(r1 + lane) & (kRing - 1)] : sentinel;
        const uint32_t xi = (*reinterpret_cast<const uint32_t*>(wbytes + ((uint32_t)i * 8u + half4)) >> sh) & 1u;
        int acc = 0;   // #neighbours with spin 1 in this lane's env
        const int deg = r1 - r0;
        const int first = deg < kWave ? deg : kWave;
        // 8 neighbours per trip, written out by hand (readlane is convergent: hipcc will not unroll
        // it, and a rolled loop pays one full LDS round trip per neighbour).  Lanes past the row end
        // hold the sentinel id N whose word is always zero, so no tail predication is needed.
        for (int j = 0; j < first; j += 8) {
            uint32_t w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t nb = (uint32_t)__builtin_amdgcn_readlane(my_nb, j + k);
                w[k] = *reinterpret_cast<const uint32_t*>(wbytes + (nb * 8u + half4));
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += (int)((w[k] >> sh) & 1u);
        }
        for (int j = r0 + kWave; j < r1; ++j) {        // hubs: the rest of the row straight from the ring
            const uint32_t nb = (uint32_t)ring[j & (kRing - 1)];
            acc += (int)((*reinterpret_cast<const uint32_t*>(wbytes + (nb * 8u + half4)) >> sh) & 1u);
        }
        const int same_minus_diff = xi ? (2 * acc - deg) : (deg - 2 * acc);   // sum_j (x_i == x_j ? +1 : -1)
        const bool flip = same_minus_diff >= 0;
        gain += flip ? same_minus_diff : 0;
        const uint64_t fm = ballot64(flip);
        if (lane == 0) words[i] ^= fm;
        // one wave: DS ops execute in issue order, so a compiler barrier is all the next node needs
        asm volatile("" ::: "memory");
        r0 = r1;
        r1 = r2;
        my_nb = nxt_nb;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    return gain;
}

// ---------------------------------------------------------------------------------------------------
// Level-scheduled sweep.  The sequential pass only orders ADJACENT nodes: with level(i) = 1 + max level of
// i's lower-numbered neighbours, the nodes of one level are pairwise non-adjacent, every earlier neighbour of
// a node lies in a lower level and every later one in a higher level -- so deciding the levels in order, the
// nodes inside a level in any order, reproduces the sequential result bit for bit (47 levels of ~43 nodes for
// G22 instead of 2000 dependent steps).  The host (rls_graph_sweep_schedule) sorts the nodes by (level, id),
// cuts levels into batches the LDS ring can hold and writes one int32 stream: for every schedule position
// the node id followed by its neighbours.  rpf[k] = offset of position k in the stream, bit 31 = first
// position of a batch.  The WA working waves of the workgroup take the positions of a batch round-robin, one
// workgroup barrier per batch; a batch of <= 768 entries stays inside the ring window wave 0 maintains.
// Returns this wave's partial gain for the lane's env (sum the W partials).
template <int W, int WA = W>   // WA = waves that take nodes; waves WA..W-1 only keep the barrier schedule
__device__ __forceinline__ int64_t sweep_tile_batched(uint64_t* words, const int32_t* rpf, int32_t* ring,
                                                      const int32_t* __restrict__ stream, int64_t len, int64_t N,
                                                      int lane, int w) {
    const unsigned char* wbytes = reinterpret_cast<const unsigned char*>(words);
    constexpr uint32_t M = 0x7fffffffu;
    // wave 0 alone feeds the ring: it requests ahead of the NEXT batch before the barrier that ends the current
    // one, so what a batch reads (<= 768 entries from its first) was complete before that barrier
    // (ring_advance waits for every refill but the newest, and the newest starts >= 1024 entries ahead).
    int64_t F = 0;
    if (w == 0) ring_prime(stream, len, F, ring, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int sh = lane & 31;
    const uint32_t half4 = (uint32_t)(lane >> 5) * 4u;
    const int sentinel = (int)N;
    int64_t gain = 0;
    int64_t bstart = 0;
    while (bstart < N) {
        // batch end = first flagged position after bstart (or N): 64 candidates checked at once
        const int64_t cand = bstart + 1 + lane;
        const bool is_end = (cand >= N) || (((uint32_t)rpf[cand]) >> 31);
        const int64_t bend = bstart + 1 + __builtin_ctzll(ballot64(is_end));
        for (int64_t p = (w < WA ? bstart + w : bend); p < bend; p += WA) {
            const int r0 = (int)((uint32_t)rpf[p] & M), r1 = (int)((uint32_t)rpf[p + 1] & M);
            const uint32_t i = (uint32_t)ring[r0 & (kRing - 1)];              // node at this position (broadcast read)
            const int deg = r1 - r0 - 1;
            const int my_nb = (lane < deg) ? ring[(r0 + 1 + lane) & (kRing - 1)] : sentinel;
            const uint32_t xi = (*reinterpret_cast<const uint32_t*>(wbytes + (i * 8u + half4)) >> sh) & 1u;
            int acc = 0;
            const int first = deg < kWave ? deg : kWave;
            for (int j = 0; j < first; j += 8) {
                uint32_t wv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t nb = (uint32_t)__builtin_amdgcn_readlane(my_nb, j + k);
                    wv[k] = *reinterpret_cast<const uint32_t*>(wbytes + (nb * 8u + half4));
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += (int)((wv[k] >> sh) & 1u);
            }
            for (int j = r0 + 1 + kWave; j < r1; ++j) {
                const uint32_t nb = (uint32_t)ring[j & (kRing - 1)];
                acc += (int)((*reinterpret_cast<const uint32_t*>(wbytes + (nb * 8u + half4)) >> sh) & 1u);
            }
            const int same_minus_diff = xi ? (2 * acc - deg) : (deg - 2 * acc);
            const bool flip = same_minus_diff >= 0;
            gain += flip ? same_minus_diff : 0;
            const uint64_t fm = ballot64(flip);
            if (lane == 0) words[i] ^= fm;
        }
        if (w == 0) ring_advance(stream, len, F, (int64_t)((uint32_t)rpf[bend] & M), ring, lane);
        __syncthreads();   // the batch's flips (and the next batch's ring entries) are visible to every wave
        bstart = bend;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    return gain;
}

// ---------------------------------------------------------------------------------------------------
// Level-PARALLEL sweep (lane = node).  All nodes of a dependency level are independent, so a wave decides
// 64 of them at once on 64-env words instead of one node at a time on 64 lanes = envs:
//   c(e) = #{neighbours j : x_j != x_i}   bit-sliced: one random ds_read_b64 + XOR + carry-save add per neighbour,
//   flip(e) = [deg - 2 c(e) >= 0] = [c(e) <= deg / 2]   bit-sliced compare of the vertical counter with a per-lane
//   constant, words[i] ^= flip.
// ~10 instructions per (node, neighbour) for all 64 envs (the lane = env form: ~4.5 per env) and one pass per
// level instead of one per node: 45 passes of ~1.2 us for G22 instead of 2000 node steps of ~0.9 us over the
// waves.  The schedule (rls_graph_sweep_levels) is read straight from global memory: group k belongs to wave
// k % W, which loads its NEXT group's header and first rounds before it waits at the level barrier, so the L2
// latency of the schedule hides behind the other waves' levels.  lvp = LDS copy of lv_ptr (G + 1 entries).
// words[N] must be 0 (idle lanes point there).  The caller recounts the objective afterwards.
template <int NP>
__device__ __forceinline__ uint64_t lv_count_le(const uint64_t (&pl)[8], uint32_t thr) {
    // mask of envs whose NP-bit vertical counter is <= thr (per lane): the carry chain of c + ~thr (rls_tile.h: lv_le_const)
    return lv_le_const<NP, 8>(pl, thr);
}

// One group of the level schedule: vertical counters of "neighbour differs" over its rounds (entries = LDS byte
// offsets of the neighbours' words -- the tile sits at LDS address 0, so an entry goes into the read as it was loaded; whole
// blocks of 8 rounds, loaded unguarded), summed across the 2 / 4 / 8 adjacent lanes a long row is spread over (lcode = log2 of
// that, sorted so that lane 0 has the group's largest), then the bit-sliced compare with the per-lane threshold.
// NC = counter planes above `fours` that the rounds can reach.  NB = the group's blocks when that is 1 or 2 (straight-line code
// on the registers that came with the header), 0: a loop that requests block b + 2 while block b is counted (blk = this lane's
// slab of block 2).
typedef const uint64_t __attribute__((address_space(3))) sweep_lds_cu64;
__device__ __forceinline__ uint64_t sweep_word_at(uint32_t a) { return *(sweep_lds_cu64*)(uintptr_t)a; }

template <int NC>
__device__ __forceinline__ void sweep_count_block(const uint32_t (&nb)[8], uint64_t own, uint64_t& ones, uint64_t& twos, uint64_t& fours,
                                                  uint64_t (&c)[5]) {
    uint64_t d[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) d[q] = sweep_word_at(nb[q]) ^ own;   // padding = the node itself: 0
    uint64_t twosA, twosB, foursA, foursB, carry;
    csa(twosA, ones, ones, d[0], d[1]);
    csa(twosB, ones, ones, d[2], d[3]);
    csa(foursA, twos, twos, twosA, twosB);
    csa(twosA, ones, ones, d[4], d[5]);
    csa(twosB, ones, ones, d[6], d[7]);
    csa(foursB, twos, twos, twosA, twosB);
    csa(carry, fours, fours, foursA, foursB);
#pragma unroll
    for (int p = 0; p < NC; ++p) {
        const uint64_t t = c[p] & carry;
        c[p] ^= carry;
        carry = t;
    }
}

template <int NB, int NC, int NP>
__device__ __forceinline__ uint64_t sweep_group_flips(const int32_t* __restrict__ blk, int rounds, const uint32_t (&nb0)[8],
                                                      const uint32_t (&nb1)[8], uint64_t own, uint32_t thr, uint32_t lcode) {
    uint64_t ones = 0, twos = 0, fours = 0, c[5] = {0, 0, 0, 0, 0};
    if constexpr (NB == 1) {
        sweep_count_block<NC>(nb0, own, ones, twos, fours, c);
    } else if constexpr (NB == 2) {
        sweep_count_block<NC>(nb0, own, ones, twos, fours, c);
        sweep_count_block<NC>(nb1, own, ones, twos, fours, c);
    } else {
        uint32_t nb[8], nx[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { nb[q] = nb0[q]; nx[q] = nb1[q]; }
        for (int r0 = 0; r0 < rounds; r0 += 8, blk += 512) {
            u32x4 na, nbv;                                   // block r0 / 8 + 2, requested before block r0 / 8 is counted
            if (r0 + 16 < rounds) {
                na = *reinterpret_cast<const u32x4*>(blk);
                nbv = *reinterpret_cast<const u32x4*>(blk + 256);
            }
            sweep_count_block<NC>(nb, own, ones, twos, fours, c);
#pragma unroll
            for (int q = 0; q < 8; ++q) nb[q] = nx[q];
            nx[0] = na.x; nx[1] = na.y; nx[2] = na.z; nx[3] = na.w;
            nx[4] = nbv.x; nx[5] = nbv.y; nx[6] = nbv.z; nx[7] = nbv.w;
        }
    }
    const int gl = __builtin_amdgcn_readlane((int)lcode, 0);
    if (gl > 0) {   // degrees < 256: the sum over a row's lanes stays below 2^8
        uint64_t pv[8] = {ones, twos, fours, c[0], c[1], c[2], c[3], c[4]};
        lv_merge_planes<1>(pv, 0ull - (uint64_t)(lcode >= 1u));
        if (gl > 1) lv_merge_planes<2>(pv, 0ull - (uint64_t)(lcode >= 2u));
        if (gl > 2) lv_merge_planes<4>(pv, 0ull - (uint64_t)(lcode >= 3u));
        return lv_count_le<8>(pv, thr);
    }
    const uint64_t pl[8] = {ones, twos, fours, c[0], c[1], c[2], c[3], c[4]};
    return lv_count_le<NP>(pl, thr);
}

// A hub (a row of 256 ... 4095 entries) is a group of its own with lane = neighbour: per-lane vertical counters over its
// rounds (64 neighbours each; the first eight rounds arrive prefetched), every plane transposed across the wave and
// popcounted -- lane e then holds env e's count -- and the same rule c <= deg / 2.
__device__ __forceinline__ uint64_t sweep_hub_flips(const int32_t* __restrict__ ent, int rounds, const uint32_t (&nb0)[8], uint64_t own,
                                                    uint32_t deg, int lane) {       // ent = the record's entries (behind its 64 header words)
    uint64_t cv[7] = {0, 0, 0, 0, 0, 0, 0};                      // rounds <= 64
    auto add = [&](uint32_t off) {
        uint64_t carry = sweep_word_at(off) ^ own;
#pragma unroll
        for (int p = 0; p < 7; ++p) { const uint64_t t = cv[p] & carry; cv[p] ^= carry; carry = t; }
    };
#pragma unroll
    for (int q = 0; q < 8; ++q) add(nb0[q]);
    for (int r = 8; r < rounds; ++r) add((uint32_t)ent[(r >> 3) * 512 + ((r >> 2) & 1) * 256 + lane * 4 + (r & 3)]);
    const BitXpose xc = bit_xpose_consts(lane);
    int cnt = 0;
#pragma unroll
    for (int p = 0; p < 7; ++p) {
        uint32_t r0 = (uint32_t)cv[p], r1 = (uint32_t)(cv[p] >> 32);
        bit_transpose64(r0, r1, xc);
        cnt += (__builtin_popcount(r0) + __builtin_popcount(r1)) << p;
    }
    return ballot64((uint32_t)cnt <= (deg >> 1));
}

template <int W>
__device__ __forceinline__ void sweep_tile_levels(uint64_t* words, const int32_t* lvp, const int32_t* __restrict__ data,
                                                  int64_t G, int64_t N, int lane, int w) {
    constexpr uint32_t M = 0x3fffffffu;
    // table entries are used as LDS addresses: the tile must sit at LDS address 0 (every caller puts it first in its dynamic LDS
    // and has no static LDS)
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)words != 0u) __builtin_trap();
    // prefetched head of this wave's next group: header word + the first TWO blocks of 8 rounds, five loads (unguarded: rounds
    // are whole blocks of 8 and the table ends in sixteen spare rows)
    uint32_t hdr = (uint32_t)N;
    uint32_t nb0[8], nb1[8];
    auto prefetch = [&](int64_t k) {
        if (k < G) {
            const int32_t* rec = data + ((uint32_t)lvp[k] & M);
            hdr = (uint32_t)rec[lane];
            const u32x4 a = *reinterpret_cast<const u32x4*>(rec + 64 + 4 * lane), b = *reinterpret_cast<const u32x4*>(rec + 320 + 4 * lane);
            const u32x4 c = *reinterpret_cast<const u32x4*>(rec + 576 + 4 * lane), d = *reinterpret_cast<const u32x4*>(rec + 832 + 4 * lane);
            nb0[0] = a.x; nb0[1] = a.y; nb0[2] = a.z; nb0[3] = a.w; nb0[4] = b.x; nb0[5] = b.y; nb0[6] = b.z; nb0[7] = b.w;
            nb1[0] = c.x; nb1[1] = c.y; nb1[2] = c.z; nb1[3] = c.w; nb1[4] = d.x; nb1[5] = d.y; nb1[6] = d.z; nb1[7] = d.w;
        }
    };
    prefetch(w);
    // level boundaries of the pass, and per group: a wave walks ITS groups (w, w + W, ...) and meets the others once per
    // boundary it crosses -- the level of group k = the number of level-start flags up to k, from a ballot over the 64
    // offsets the wave holds in registers
    int num_levels = 0;
    for (int64_t k0 = 0; k0 < G; k0 += kWave)
        num_levels += __builtin_popcountll(ballot64(k0 + lane < G && (lvp[k0 + lane < G ? k0 + lane : G] >> 31) != 0));
    int chunk = 0, chunk_next = 0;
    int64_t cbase = -1;
    uint64_t lmask = 0;
    int lev_base = 0, passed = 0;
    for (int64_t k = w; k < G; k += W) {
        if ((k & ~(int64_t)63) != cbase) {
            cbase = k & ~(int64_t)63;
            lev_base += __builtin_popcountll(lmask);
            const int64_t a0 = cbase + lane <= G ? cbase + lane : G, a1 = cbase + 1 + lane <= G ? cbase + 1 + lane : G;
            chunk = lvp[a0];
            chunk_next = lvp[a1];
            lmask = ballot64(cbase + lane < G && (chunk >> 31) != 0);
        }
        const int need = lev_base + __builtin_popcountll(lmask & ((2ull << (k & 63)) - 1ull));
        for (; passed < need; ++passed) __syncthreads();     // a new level starts: every earlier flip is visible
        const int64_t p0 = (uint32_t)__builtin_amdgcn_readlane(chunk, (int)(k & 63)) & M;
        const int64_t p1 = (uint32_t)__builtin_amdgcn_readlane(chunk_next, (int)(k & 63)) & M;
        const int rounds = (int)((p1 - p0) >> 6) - 1;        // a multiple of 8
        if (((uint32_t)__builtin_amdgcn_readlane(chunk, (int)(k & 63)) >> 30) & 1u) {   // a hub: one node, lane = neighbour
            const uint32_t hnode = (uint32_t)__builtin_amdgcn_readlane((int)hdr, 0), hdeg = (uint32_t)__builtin_amdgcn_readlane((int)hdr, 1);
            const uint64_t hown = words[hnode];
            const uint64_t hflip = sweep_hub_flips(data + p0 + 64, rounds, nb0, hown, hdeg, lane);
            if (lane == 0) words[hnode] = hown ^ hflip;
            prefetch(k + W);
            continue;
        }
        const uint32_t node = hdr & 0xFFFFFu, thr = (hdr >> 20) & 0xFFu, lcode = (hdr >> 28) & 3u;
        const uint64_t own = words[node];
        const int32_t* blk = data + p0 + 64 + 1024 + 4 * lane;      // this lane's slab of block 2
        uint64_t flip;
        if (rounds == 8) flip = sweep_group_flips<1, 1, 4>(blk, rounds, nb0, nb1, own, thr, lcode);
        else if (rounds == 16) flip = sweep_group_flips<2, 2, 5>(blk, rounds, nb0, nb1, own, thr, lcode);
        else if (rounds <= 24) flip = sweep_group_flips<0, 2, 5>(blk, rounds, nb0, nb1, own, thr, lcode);     // (0: isolated nodes; 24)
        else flip = sweep_group_flips<0, 4, 7>(blk, rounds, nb0, nb1, own, thr, lcode);
        if (node < (uint32_t)N && (lane & ((1 << lcode) - 1)) == 0) words[node] = own ^ flip;
        prefetch(k + W);
    }
    for (; passed < num_levels; ++passed) __syncthreads();
    __syncthreads();
}

}  // namespace rls

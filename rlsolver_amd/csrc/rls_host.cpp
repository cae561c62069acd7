// Host-only half of the C ABI: error plumbing, rls_version and the [host] schedule builders (rls_graph_sweep_schedule,
// rls_graph_sweep_batches, rls_graph_sweep_levels, rls_mcpg_visit_levels, rls_graph_ell).  No HIP here: this translation
// unit is plain C++17, compiled into librlsolver_hip.so with the rest AND, on its own, with
// -fsanitize=address,undefined by tests/test_sanitize.py (tools/host_sanitize.cpp drives it over random graphs).
#include "rls_host.h"
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace rls {

static thread_local char g_err[512] = "";

static const char* const kKnobNames[KN_COUNT] = {
#define RLS_X(n) "RLS_" #n,
    RLS_KNOB_LIST(RLS_X)
#undef RLS_X
};
std::atomic<int64_t> g_knobs[KN_COUNT];
static int knob_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < KN_COUNT; ++i)
        if (std::strcmp(name, kKnobNames[i]) == 0 || std::strcmp(name, kKnobNames[i] + 4) == 0) return i;
    return -1;
}
namespace {
struct KnobInit {
    KnobInit() {
        for (int i = 0; i < KN_COUNT; ++i) g_knobs[i].store(kKnobUnset, std::memory_order_relaxed);
#ifdef RLS_DEV   // development builds only: the environment seeds the table once, at load
        for (int i = 0; i < KN_COUNT; ++i)
            if (const char* v = std::getenv(kKnobNames[i])) g_knobs[i].store(std::atoll(v), std::memory_order_relaxed);
#endif
    }
};
static KnobInit g_knob_init;
}  // namespace

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---- lanes per row of a lane = node group (K5 / K7 level schedules) ------------------------------------------------
// A group costs its LONGEST lane and the waves of a workgroup meet at every level boundary, so one long row among short
// ones stalls the level and pads 63 lanes: a row longer than `cap` entries is spread over 2, 4 or 8 adjacent lanes
// (lane j of them takes entries j, j + L, ...; the kernel adds the lanes' bit-sliced counters before the compare,
// ~110 VALU per group that has any).  Per level the cap (none / 32 / 16 / 8 entries per lane; a lane never holds
// more than 64) is the one with the least estimated time: the level's work over the waves plus its longest group, in
// VALU instructions (~150 per group, ~50 per 8 rounds, ~60 for the cross-lane sums: see plan_lane_groups).
struct LaneGroup { int64_t k0, k1; int32_t rounds; bool multi; };   // rows [k0, k1) of the level's degree-descending order

static inline int32_t lanes_log2_for(int32_t deg, int32_t cap) {
    int32_t lc = 0;
    while (lc < 3 && ((cap > 0 && deg > (cap << lc)) || ((deg + (1 << lc) - 1) >> lc) > 64)) ++lc;
    return lc;
}

// groups of <= 64 lanes over rows [a, b) (degrees non-increasing: lanes of a row stay aligned to their count);
// rounds come in whole blocks of 8 (the kernels load them unguarded).  Returns the estimate.
template <class DegAt>
static int64_t plan_lane_groups(int64_t a, int64_t b, int32_t cap, DegAt deg_at, std::vector<LaneGroup>* out) {
    int64_t total = 0, longest = 0, k = a;
    while (k < b) {
        LaneGroup g{k, k, 0, false};
        int32_t used = 0;
        while (g.k1 < b) {
            const int32_t deg = deg_at(g.k1), lc = lanes_log2_for(deg, cap), L = 1 << lc;
            if (used + L > 64) break;
            used += L;
            const int32_t r = (deg + L - 1) / L;
            if (r > g.rounds) g.rounds = r;
            g.multi = g.multi || lc > 0;
            ++g.k1;
        }
        g.rounds = (g.rounds + 7) / 8 * 8;
        // (round 5: 150 / 50 / 60 -- a block of 8 rounds costs the kernels ~45 instructions since the tables are lane-major, a
        // cross-lane merge step ~64; with round 4's 150 / 85 / 110 a G22-sized graph got 129 groups, now 101: K5 110.6 -> 103 us,
        // K7 at BA-1e4 and seven other graph shapes unchanged within 1 %: tools/timing/plan_ab.py, plan_shapes.py)
        const int64_t cost = knob(KN_PLAN_FIXED, 150) + knob(KN_PLAN_BLOCK, 50) * (int64_t)(g.rounds / 8) + (g.multi ? knob(KN_PLAN_MERGE, 60) : 0);
        total += cost;
        if (cost > longest) longest = cost;
        if (out) out->push_back(g);
        k = g.k1;
    }
    return total / 8 + longest;
}

template <class DegAt>
static int32_t best_lane_cap(int64_t a, int64_t b, DegAt deg_at) {
    int32_t best_cap = 0;
    int64_t best = -1;
    for (int32_t cap : {0, 32, 16, 8}) {
        const int64_t c = plan_lane_groups(a, b, cap, deg_at, nullptr);
        if (best < 0 || c < best) { best = c; best_cap = cap; }
    }
    return best_cap;
}

// ---- LDS banks of a lane = node group ---------------------------------------------------------------------------------
// A group's round is one ds_read_b64 per lane at the byte offset the table holds: word n of the tile covers banks 2 (n mod 32)
// and + 1, a half-wave (32 lanes) is served in one LDS cycle when its 32 words fall in 32 different classes n mod 32 (equal
// addresses broadcast), and every further word of a class costs a cycle.  Graph neighbours are as good as random (SQ counters,
// round 3: 1.6 - 2.4 conflict cycles per active LDS cycle in K3 / K7 / K9 / the ISCO step) -- but the ORDER in which a lane
// visits its neighbours is free (a count does not depend on it), so each lane's entries are dealt to the rounds greedily: round
// by round, every lane of a half-wave takes, of the entries it still holds, the one whose class is least used in this round.
// rows = [rounds][64] table entries (bit 31 may carry a flag); in place; a pure permutation inside every lane's column.
static void spread_banks(int32_t* rows, int64_t rounds, int shift = 3) {   // shift 3: entries are byte offsets of 8-byte words; 0: word indices
    if (rounds < 2 || rounds > 128) return;     // (rows of hundreds of entries are hub groups: other kernels, and O(rounds^2) here)
    for (int half = 0; half < 2; ++half) {
        std::vector<std::vector<int32_t>> have(32);
        for (int l = 0; l < 32; ++l) {
            have[(size_t)l].resize((size_t)rounds);
            for (int64_t r = 0; r < rounds; ++r) have[(size_t)l][(size_t)r] = rows[r * 64 + half * 32 + l];
        }
        for (int64_t r = 0; r < rounds; ++r) {
            int load[32] = {0};
            std::vector<uint32_t> placed;                       // addresses already in this round (a repeat broadcasts)
            placed.reserve(32);
            for (int q = 0; q < 32; ++q) {
                const int l = (int)((q + r) & 31);              // (rotate who chooses first)
                std::vector<int32_t>& h = have[(size_t)l];
                size_t best = 0;
                int best_cost = 1 << 30;
                for (size_t k = 0; k < h.size(); ++k) {
                    const uint32_t addr = (uint32_t)h[k] & 0x7fffffffu;
                    int cost = load[(addr >> shift) & 31];
                    for (uint32_t a : placed)
                        if (a == addr) { cost = -1; break; }
                    if (cost < best_cost) { best_cost = cost; best = k; }
                }
                const int32_t e = h[best];
                h.erase(h.begin() + (std::ptrdiff_t)best);
                rows[r * 64 + half * 32 + l] = e;
                const uint32_t addr = (uint32_t)e & 0x7fffffffu;
                if (best_cost >= 0) { ++load[(addr >> shift) & 31]; placed.push_back(addr); }
            }
        }
    }
}

}  // namespace rls

extern "C" {

int rls_version(void) { return RLS_ABI_VERSION; }

const char* rls_last_error_string(void) { return rls::g_err; }

int rls_tuning_set(const char* name, int64_t value) {
    const int i = rls::knob_index(name);
    if (i < 0) return rls::fail(RLS_EINVAL, "rls_tuning_set: unknown knob '%s'", name ? name : "(null)");
    if (value == rls::kKnobUnset) return rls::fail(RLS_EINVAL, "rls_tuning_set: INT64_MIN is the 'unset' marker");
    rls::g_knobs[i].store(value, std::memory_order_relaxed);
    return RLS_OK;
}

int rls_tuning_unset(const char* name) {
    if (name == nullptr) {   // all of them
        for (int i = 0; i < rls::KN_COUNT; ++i) rls::g_knobs[i].store(rls::kKnobUnset, std::memory_order_relaxed);
        return RLS_OK;
    }
    const int i = rls::knob_index(name);
    if (i < 0) return rls::fail(RLS_EINVAL, "rls_tuning_unset: unknown knob '%s'", name);
    rls::g_knobs[i].store(rls::kKnobUnset, std::memory_order_relaxed);
    return RLS_OK;
}

int rls_tuning_get(const char* name, int64_t* value, int32_t* is_set) {
    const int i = rls::knob_index(name);
    if (i < 0 || !value || !is_set) return rls::fail(RLS_EINVAL, "rls_tuning_get: unknown knob or NULL output");
    const int64_t v = rls::g_knobs[i].load(std::memory_order_relaxed);
    *is_set = v != rls::kKnobUnset;
    *value = *is_set ? v : 0;
    return RLS_OK;
}

int rls_tuning_name(int32_t index, const char** name) {
    if (!name || index < 0 || index >= rls::KN_COUNT) return rls::fail(RLS_EINVAL, "rls_tuning_name: index out of range");
    *name = rls::kKnobNames[index];
    return RLS_OK;
}

int rls_graph_sweep_schedule(const int32_t* rowptr, const int32_t* col, int64_t N, int32_t max_nodes,
                             int32_t max_entries, int32_t* rowptr_flagged, int32_t* stream, int64_t* num_batches,
                             int64_t* num_levels) {
    if (!rowptr || !rowptr_flagged || !stream || N < 0 || max_nodes < 1 || max_nodes > 64 || max_entries < 1 ||
        (N > 0 && rowptr[N] > 0 && !col))
        return rls::fail(RLS_EINVAL, "rls_graph_sweep_schedule: bad arguments");
    if ((int64_t)rowptr[N > 0 ? N : 0] + N >= (int64_t)0x7fffffff)
        return rls::fail(RLS_EUNSUPPORTED, "rls_graph_sweep_schedule: stream offsets need 31 bits");
    std::vector<int32_t> level((size_t)(N > 0 ? N : 1), 0);
    int32_t nlev = 0;
    for (int64_t i = 0; i < N; ++i) {
        int32_t l = 0;
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j)
            if (col[j] < i && level[(size_t)col[j]] + 1 > l) l = level[(size_t)col[j]] + 1;
        level[(size_t)i] = l;
        if (l + 1 > nlev) nlev = l + 1;
    }
    // counting sort by level (stable: ids ascending inside a level)
    std::vector<int64_t> start((size_t)nlev + 1, 0);
    for (int64_t i = 0; i < N; ++i) ++start[(size_t)level[(size_t)i] + 1];
    for (int32_t l = 0; l < nlev; ++l) start[(size_t)l + 1] += start[(size_t)l];
    std::vector<int32_t> order((size_t)(N > 0 ? N : 1));
    {
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (int64_t i = 0; i < N; ++i) order[(size_t)fill[(size_t)level[(size_t)i]]++] = (int32_t)i;
    }
    int64_t nb = 0, off = 0;
    int32_t nodes = 0, entries = 0, cur_level = -1;
    for (int64_t k = 0; k < N; ++k) {
        const int32_t i = order[(size_t)k];
        const int32_t len = 1 + rowptr[i + 1] - rowptr[i];
        const bool first = level[(size_t)i] != cur_level || nodes >= max_nodes || entries + len > max_entries;
        if (first) { ++nb; nodes = 0; entries = 0; cur_level = level[(size_t)i]; }
        ++nodes;
        entries += len;
        rowptr_flagged[k] = (int32_t)((uint32_t)off | (first ? 0x80000000u : 0u));
        stream[off++] = i;
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) stream[off++] = col[j];
    }
    rowptr_flagged[N] = (int32_t)off;
    if (num_batches) *num_batches = nb;
    if (num_levels) *num_levels = nlev;
    return RLS_OK;
}

int rls_graph_sweep_batches(const int32_t* rowptr, const int32_t* col, int64_t N, int32_t max_nodes,
                            int32_t max_entries, int32_t* rowptr_flagged, int64_t* num_batches) {
    if (!rowptr || !rowptr_flagged || N < 0 || max_nodes < 1 || max_nodes > 64 || max_entries < 1 ||
        (N > 0 && rowptr[N] > 0 && !col))
        return rls::fail(RLS_EINVAL, "rls_graph_sweep_batches: bad arguments");
    std::vector<int32_t> stamp((size_t)(N > 0 ? N : 1), -1);
    int64_t nb = 0;
    int32_t cur = -1, nodes = 0, entries = 0;
    for (int64_t i = 0; i < N; ++i) {
        const int32_t r0 = rowptr[i], r1 = rowptr[i + 1];
        bool start = (i == 0) || nodes >= max_nodes || entries + (r1 - r0) > max_entries;
        for (int32_t j = r0; j < r1 && !start; ++j) start = stamp[(size_t)col[j]] == cur;   // adjacent to a member
        if (start) { ++cur; ++nb; nodes = 0; entries = 0; }
        stamp[(size_t)i] = cur;
        ++nodes;
        entries += r1 - r0;
        rowptr_flagged[i] = (int32_t)((uint32_t)r0 | (start ? 0x80000000u : 0u));
    }
    rowptr_flagged[N] = rowptr[N];
    if (num_batches) *num_batches = nb;
    return RLS_OK;
}

int rls_graph_sweep_levels(const int32_t* rowptr, const int32_t* col, int64_t N, int32_t* lv_ptr, int64_t ptr_capacity,
                           int32_t* lv_data, int64_t data_capacity, int64_t* num_groups, int64_t* total) {
    if (!rowptr || N < 0 || (N > 0 && rowptr[N] > 0 && !col) || !num_groups || !total)
        return rls::fail(RLS_EINVAL, "rls_graph_sweep_levels: bad arguments");
    if (N >= (1 << 20)) return rls::fail(RLS_EUNSUPPORTED, "rls_graph_sweep_levels: N >= 2^20");
    std::vector<int32_t> level((size_t)(N > 0 ? N : 1), 0);
    int32_t nlev = 0;
    for (int64_t i = 0; i < N; ++i) {
        if (rowptr[i + 1] - rowptr[i] >= 4096) return rls::fail(RLS_EUNSUPPORTED, "rls_graph_sweep_levels: degree >= 4096");
        int32_t l = 0;
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j)
            if (col[j] < i && level[(size_t)col[j]] + 1 > l) l = level[(size_t)col[j]] + 1;
        level[(size_t)i] = l;
        if (l + 1 > nlev) nlev = l + 1;
    }
    std::vector<int64_t> start((size_t)nlev + 1, 0);
    for (int64_t i = 0; i < N; ++i) ++start[(size_t)level[(size_t)i] + 1];
    for (int32_t l = 0; l < nlev; ++l) start[(size_t)l + 1] += start[(size_t)l];
    std::vector<int32_t> order((size_t)(N > 0 ? N : 1));
    {
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (int64_t i = 0; i < N; ++i) order[(size_t)fill[(size_t)level[(size_t)i]]++] = (int32_t)i;
    }
    // nodes of a level are independent: longest rows first, so that the lanes a long row is spread over stay aligned.
    // A row of 256 or more entries (a hub) is a group of its own, lane = neighbour (bit 30 of its lv_ptr entry).
    constexpr int32_t kHub = 256;
    auto degn = [&](int32_t i) { return rowptr[i + 1] - rowptr[i]; };
    for (int32_t l = 0; l < nlev; ++l)
        std::stable_sort(order.begin() + start[(size_t)l], order.begin() + start[(size_t)l + 1],
                         [&](int32_t x, int32_t y) { return degn(x) > degn(y); });
    auto deg_at = [&](int64_t k) { return degn(order[(size_t)k]); };
    int64_t ng = 0, off = 0;
    std::vector<rls::LaneGroup> groups;
    std::vector<int32_t> rows;      // a group's rounds as [rounds][64] (what spread_banks permutes)
    // Record layout (round 5, as rls_mcpg_visit_levels): 64 header words (lane l's at l), then per block of 8 rounds two slabs of
    // [64 lanes][4 rounds] -- a lane fetches a block in two 16-byte loads.  Round r of lane l:
    auto slot = [](int64_t r, int64_t l) { return 64 + (r >> 3) * 512 + ((r >> 2) & 1) * 256 + l * 4 + (r & 3); };
    for (int32_t l = 0; l < nlev; ++l) {
        const int64_t a0 = start[(size_t)l], b = start[(size_t)l + 1];
        int64_t a = a0;                                    // [a0, a): the level's hubs (sorted first), [a, b): lane = node rows
        while (a < b && deg_at(a) >= kHub) ++a;
        const int32_t cap = rls::best_lane_cap(a, b, deg_at);
        groups.clear();
        for (int64_t h = a0; h < a; ++h) groups.push_back(rls::LaneGroup{h, h + 1, (int32_t)(((deg_at(h) + 63) / 64 + 7) & ~7), false});
        const size_t nhub = groups.size();
        rls::plan_lane_groups(a, b, cap, deg_at, &groups);
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            const rls::LaneGroup& g = groups[gi];
            const bool hub = gi < nhub;
            const int64_t len = (int64_t)(1 + g.rounds) * 64;
            if (off + len >= (int64_t)0x3fffffff) return rls::fail(RLS_EUNSUPPORTED, "rls_graph_sweep_levels: too large");
            if (lv_ptr) {
                if (ng + 1 >= ptr_capacity) return rls::fail(RLS_EINVAL, "rls_graph_sweep_levels: ptr capacity too small");
                lv_ptr[ng] = (int32_t)((uint32_t)off | (gi == 0 ? 0x80000000u : 0u) | (hub ? 0x40000000u : 0u));
            }
            if (lv_data) {
                if (off + len > data_capacity) return rls::fail(RLS_EINVAL, "rls_graph_sweep_levels: data capacity too small");
                int32_t* rec = lv_data + off;
                for (int64_t e = 0; e < 64; ++e) rec[e] = (int32_t)N;                          // idle lanes: node N (its word is zero)
                rows.assign((size_t)(g.rounds * 64), (int32_t)(N * 8));
                if (hub) {   // header: lane 0 = the node, lane 1 = its degree; then its neighbours 64 per round, padded with itself
                    const int32_t i = order[(size_t)g.k0], deg = degn(i);
                    rec[0] = i;
                    rec[1] = deg;
                    for (int64_t e = 0; e < (int64_t)g.rounds * 64; ++e)
                        rows[(size_t)e] = (int32_t)((uint32_t)(e < deg ? col[rowptr[i] + e] : i) * 8u);
                } else {
                    int32_t ln = 0;
                    for (int64_t k = g.k0; k < g.k1; ++k) {
                        const int32_t i = order[(size_t)k], deg = degn(i), lc = rls::lanes_log2_for(deg, cap), L = 1 << lc;
                        for (int32_t j = 0; j < L; ++j) {              // lane j of the node's L takes neighbours j, j + L, ...
                            rec[ln + j] = (int32_t)((uint32_t)i | ((uint32_t)(deg >> 1) << 20) | ((uint32_t)lc << 28));
                            for (int32_t r = 0; r < g.rounds; ++r) {   // short lanes end in the node itself: x_i ^ x_i adds nothing
                                const int32_t e = r * L + j;
                                rows[(size_t)((int64_t)r * 64 + ln + j)] = (int32_t)((uint32_t)(e < deg ? col[rowptr[i] + e] : i) * 8u);
                            }
                        }
                        ln += L;
                    }
                    rls::spread_banks(rows.data(), g.rounds);
                }
                for (int64_t r = 0; r < g.rounds; ++r)
                    for (int64_t q = 0; q < 64; ++q) rec[slot(r, q)] = rows[(size_t)(r * 64 + q)];
            }
            off += len;
            ++ng;
        }
    }
    if (lv_ptr) {
        if (ng >= ptr_capacity) return rls::fail(RLS_EINVAL, "rls_graph_sweep_levels: ptr capacity too small");
        lv_ptr[ng] = (int32_t)off;
    }
    // sixteen spare rows behind the last record: the kernel requests a group's first two blocks of rounds without looking
    if (lv_data) {
        if (off + 16 * 64 > data_capacity) return rls::fail(RLS_EINVAL, "rls_graph_sweep_levels: data capacity too small");
        for (int64_t e = 0; e < 16 * 64; ++e) lv_data[off + e] = (int32_t)(N * 8);
    }
    *num_groups = ng;
    *total = off + 16 * 64;
    return RLS_OK;
}

int rls_mcpg_visit_levels(const int32_t* rowptr, const int32_t* col, int64_t N, const int32_t* order, int32_t* lv_ptr,
                          int64_t ptr_capacity, int32_t* lv_data, int64_t data_capacity, int64_t* num_groups,
                          int64_t* total) {
    if (!rowptr || !order || N < 0 || (N > 0 && rowptr[N] > 0 && !col) || !num_groups || !total)
        return rls::fail(RLS_EINVAL, "rls_mcpg_visit_levels: bad arguments");
    if (N >= (1 << 20)) return rls::fail(RLS_EUNSUPPORTED, "rls_mcpg_visit_levels: N >= 2^20");
    constexpr int32_t kHubDeg = 128;             // longer rows get a group of their own, lane = neighbour
    std::vector<int32_t> pos_of((size_t)(N > 0 ? N : 1), -1);
    for (int64_t p = 0; p < N; ++p) {
        if (order[p] < 0 || order[p] >= N || pos_of[(size_t)order[p]] != -1)
            return rls::fail(RLS_EINVAL, "rls_mcpg_visit_levels: order is not a permutation");
        pos_of[(size_t)order[p]] = (int32_t)p;
    }
    std::vector<int32_t> level((size_t)(N > 0 ? N : 1), 0), nfresh((size_t)(N > 0 ? N : 1), 0);   // per position
    int32_t nlev = 0;
    for (int64_t p = 0; p < N; ++p) {
        const int32_t i = order[p];
        if (rowptr[i + 1] - rowptr[i] >= 1024) return rls::fail(RLS_EUNSUPPORTED, "rls_mcpg_visit_levels: degree >= 1024");
        int32_t l = 0, nf = 0;
        for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
            const int32_t q = pos_of[(size_t)col[j]];
            if (q < p) { if (level[(size_t)q] + 1 > l) l = level[(size_t)q] + 1; }
            else if (q > p) ++nf;
        }
        level[(size_t)p] = l;
        nfresh[(size_t)p] = nf;
        if (l + 1 > nlev) nlev = l + 1;
    }
    // positions by (level, hub last, degree descending, position)
    std::vector<int32_t> sp((size_t)(N > 0 ? N : 1));
    for (int64_t p = 0; p < N; ++p) sp[(size_t)p] = (int32_t)p;
    auto degp = [&](int32_t p) { return rowptr[order[p] + 1] - rowptr[order[p]]; };
    std::stable_sort(sp.begin(), sp.begin() + N, [&](int32_t a, int32_t b) {
        if (level[(size_t)a] != level[(size_t)b]) return level[(size_t)a] < level[(size_t)b];
        const bool ha = degp(a) > kHubDeg, hb = degp(b) > kHubDeg;
        if (ha != hb) return hb;
        return degp(a) > degp(b);
    });
    int64_t ng = 0, off = 0;
    auto header = [&](int32_t p, int32_t lcode, int32_t& h0, int32_t& h1) {
        const int32_t i = order[p], deg = degp(p), t0 = deg + nfresh[(size_t)p];
        h0 = (int32_t)((uint32_t)i | ((uint32_t)((deg + 1) >> 1) << 20) | ((uint32_t)lcode << 28) | ((deg & 1) ? 0u : 0x80000000u));
        h1 = (int32_t)((uint32_t)p | ((uint32_t)((t0 + 1) >> 1) << 20) | ((t0 & 1) ? 0u : 0x80000000u));
    };
    auto entry = [&](int32_t p, int32_t r) {   // r-th neighbour word of position p
        const int32_t i = order[p], nb = col[rowptr[i] + r];
        return (int32_t)(((uint32_t)nb * 8u) | (pos_of[(size_t)nb] > p ? 0x80000000u : 0u));   // LDS byte offset of the word
    };
    // Record layout (round 5): LANE-major, so that a lane fetches its data in a few wide loads instead of one dword per row --
    // dwords [2 l, 2 l + 1] = lane l's two header words (one 8-byte load), then per block of 8 rounds two slabs of [64 lanes][4
    // rounds] (two 16-byte loads per lane and block; a wave-instruction reads 1 KB contiguous).  Round r of lane l:
    auto slot = [](int64_t r, int64_t l) { return 128 + (r >> 3) * 512 + ((r >> 2) & 1) * 256 + l * 4 + (r & 3); };
    using Grp = rls::LaneGroup;
    auto deg_at = [&](int64_t k) { return degp(sp[(size_t)k]); };
    std::vector<Grp> groups;
    std::vector<int32_t> rows;      // a group's rounds as [rounds][64] (what spread_banks permutes)
    int64_t k0 = 0;
    while (k0 < N) {
        const int32_t lev = level[(size_t)sp[(size_t)k0]];
        // the level's lane = node rows [k0, kn) and its hubs [kn, ke)
        int64_t kn = k0, ke;
        while (kn < N && level[(size_t)sp[(size_t)kn]] == lev && degp(sp[(size_t)kn]) <= kHubDeg) ++kn;
        ke = kn;
        while (ke < N && level[(size_t)sp[(size_t)ke]] == lev) ++ke;
        const int32_t best_cap = rls::best_lane_cap(k0, kn, deg_at);
        // the level's groups, the costly ones first: waves take groups round-robin, the long ones should not queue up behind
        // a wave's earlier work
        groups.clear();
        for (int64_t h = kn; h < ke; ++h) groups.push_back(Grp{h, h + 1, (degp(sp[(size_t)h]) + 63) / 64, false});
        std::stable_sort(groups.begin(), groups.end(), [](const Grp& x, const Grp& y) { return x.rounds > y.rounds; });
        rls::plan_lane_groups(k0, kn, best_cap, deg_at, &groups);
        bool level_start = true;
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            const Grp& g = groups[gi];
            const bool hub = g.k0 >= kn;
            const int64_t rounds = (g.rounds + 7) / 8 * 8;      // whole blocks (a hub's own count of rounds is ceil(deg / 64): word 4)
            const int64_t len = (2 + rounds) * 64;
            if (off + len >= (int64_t)0x3fffffff / 2) return rls::fail(RLS_EUNSUPPORTED, "rls_mcpg_visit_levels: too large");
            if (lv_ptr) {
                if (ng + 1 >= ptr_capacity) return rls::fail(RLS_EINVAL, "rls_mcpg_visit_levels: ptr capacity too small");
                lv_ptr[ng] = (int32_t)((uint32_t)off | (level_start ? 0x80000000u : 0u) | (hub ? 0x40000000u : 0u));
            }
            if (lv_data) {
                if (off + len > data_capacity) return rls::fail(RLS_EINVAL, "rls_mcpg_visit_levels: data capacity too small");
                int32_t* rec = lv_data + off;
                for (int64_t e = 0; e < len; ++e) rec[e] = (int32_t)(e < 128 ? N : N * 8);      // idle header lanes: node N; padding: the zero word
                rows.assign((size_t)(rounds * 64), (int32_t)(N * 8));
                if (hub) {
                    const int32_t p = sp[(size_t)g.k0], md = degp(p);
                    header(p, 0, rec[0], rec[1]);
                    rec[4] = md;                                   // lane 2's first header word
                    for (int32_t r = 0; r < md; ++r) rows[(size_t)r] = entry(p, r);      // neighbour r: round r / 64, lane r % 64
                } else {
                    int32_t ln = 0;
                    for (int64_t k = g.k0; k < g.k1; ++k) {
                        const int32_t p = sp[(size_t)k], deg = degp(p), lc = rls::lanes_log2_for(deg, best_cap), L = 1 << lc;
                        for (int32_t j = 0; j < L; ++j) {          // lane j of the node's L takes neighbours j, j + L, ...
                            header(p, lc, rec[2 * (ln + j)], rec[2 * (ln + j) + 1]);
                            for (int32_t r = j; r < deg; r += L) rows[(size_t)((r / L) * 64 + ln + j)] = entry(p, r);
                        }
                        ln += L;
                    }
                    rls::spread_banks(rows.data(), rounds);
                }
                for (int64_t r = 0; r < rounds; ++r)
                    for (int64_t l = 0; l < 64; ++l) rec[slot(r, l)] = rows[(size_t)(r * 64 + l)];
            }
            off += len;
            ++ng;
            level_start = false;
        }
        k0 = ke;
    }
    if (lv_ptr) {
        if (ng >= ptr_capacity) return rls::fail(RLS_EINVAL, "rls_mcpg_visit_levels: ptr capacity too small");
        lv_ptr[ng] = (int32_t)off;
    }
    // sixteen spare rows behind the last record (the kernel requests a group's header and first two blocks without looking), then the
    // whole table AGAIN with the `fresh` flags cleared: passes >= 1 do not need them, and an entry that is nothing but the LDS
    // address of its word goes into the read instruction as it was loaded (one VALU less per neighbour and 64 chains)
    const int64_t half = off + 16 * 64;
    if (lv_data) {
        if (2 * half > data_capacity) return rls::fail(RLS_EINVAL, "rls_mcpg_visit_levels: data capacity too small");
        for (int64_t e = 0; e < 16 * 64; ++e) lv_data[off + e] = (int32_t)(N * 8);
        for (int64_t e = 0; e < half; ++e) lv_data[half + e] = lv_data[e];
        for (int64_t q = 0; q < ng && lv_ptr; ++q) {
            const int64_t a = (uint32_t)lv_ptr[q] & 0x3fffffffu, b = (uint32_t)lv_ptr[q + 1] & 0x3fffffffu;
            for (int64_t e = a + 128; e < b; ++e) lv_data[half + e] &= 0x7fffffff;
        }
        if (!lv_ptr) return rls::fail(RLS_EINVAL, "rls_mcpg_visit_levels: lv_data without lv_ptr");
    }
    *num_groups = ng;
    *total = 2 * half;
    return RLS_OK;
}

int rls_graph_ell(const int32_t* rowptr, const int32_t* col, int64_t N, int32_t* ell_ptr, int32_t* ell,
                  int64_t capacity, int64_t* total) {
    if (!rowptr || !ell_ptr || N < 0 || (N > 0 && rowptr[N] > 0 && !col))
        return rls::fail(RLS_EINVAL, "rls_graph_ell: bad arguments");
    const int64_t G = (N + 63) / 64;
    int64_t off = 0;
    for (int64_t g = 0; g < G; ++g) {
        const int64_t i0 = g * 64, i1 = (i0 + 64 < N) ? i0 + 64 : N;
        int32_t md = 0;
        for (int64_t i = i0; i < i1; ++i)
            if (rowptr[i + 1] - rowptr[i] > md) md = rowptr[i + 1] - rowptr[i];
        if (off + (int64_t)md * 64 >= (int64_t)0x7fffffff) return rls::fail(RLS_EUNSUPPORTED, "rls_graph_ell: too large");
        ell_ptr[g] = (int32_t)off;
        if (ell) {
            if (off + (int64_t)md * 64 > capacity) return rls::fail(RLS_EINVAL, "rls_graph_ell: capacity too small");
            for (int32_t k = 0; k < md; ++k)
                for (int64_t l = 0; l < 64; ++l) {
                    const int64_t i = i0 + l;
                    int32_t v = (int32_t)(i < N ? i : 0);
                    if (i < N && rowptr[i] + k < rowptr[i + 1]) v = col[rowptr[i] + k];
                    ell[off + (int64_t)k * 64 + l] = v;
                }
            rls::spread_banks(ell + off, md, 0);       // a lane's neighbours in the order that spreads each round over the LDS banks
        }
        off += (int64_t)md * 64;
    }
    ell_ptr[G] = (int32_t)off;
    if (total) *total = off;
    return RLS_OK;
}

}  // extern "C"

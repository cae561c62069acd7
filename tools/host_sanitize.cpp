// Sanitizer driver ("race / memory checking"): the host-only half of the C ABI (rls_host.cpp: the
// schedule builders every DeviceGraph and every MCPG visiting order go through) and the C oracle (oracle/oracle.c), run
// over random graphs -- G(n, m), stars and multi-hub graphs, paths, isolated nodes, N = 1, duplicate edges -- in
// EXACTLY-sized heap buffers, built with -fsanitize=address,undefined by tests/test_sanitize.py.  Every builder runs
// its sizing call first and then fills buffers of exactly that size, so a one-past-the-end write is an ASAN report;
// structural invariants (every node scheduled once, levels respect the dependency order, ELL rows complete) are checked
// on top.  Exit code 0 = clean.  CPU only: sanitizers do not run on the GPU box.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <vector>

#include "rlsolver_hip.h"

#include "../oracle/oracle.h"

#define CHECK(cond, ...)                                                   \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "host_sanitize: CHECK failed: %s | ", #cond); \
            std::fprintf(stderr, __VA_ARGS__);                             \
            std::fprintf(stderr, "\n");                                    \
            std::exit(1);                                                  \
        }                                                                  \
    } while (0)

struct Graph {
    int64_t n;
    std::vector<int32_t> eu, ev;              // stored edges, sorted by (eu, ev)
    std::vector<int32_t> erowptr;             // [n + 1]
    std::vector<int32_t> rowptr, col;         // symmetric CSR, rows sorted
    int32_t max_deg;
};

static Graph make_graph(std::mt19937_64& rng, int kind) {
    auto U = [&](int64_t lo, int64_t hi) { return (int64_t)(lo + rng() % (uint64_t)(hi - lo + 1)); };
    Graph g;
    std::set<std::pair<int32_t, int32_t>> e;
    switch (kind) {
        case 0: g.n = 1; break;                                                        // N = 1, no edges
        case 1: g.n = U(2, 70); break;                                                 // tiny
        case 2: g.n = U(64, 700); break;
        case 3: g.n = U(700, 5000); break;
        default: g.n = U(100, 3000); break;
    }
    const int64_t n = g.n;
    auto add = [&](int64_t a, int64_t b) { if (a != b) e.insert({(int32_t)std::min(a, b), (int32_t)std::max(a, b)}); };
    if (n > 1) {
        if (kind == 4) {                                                               // hubs (degree up to 4000) + isolated nodes
            const int hubs = (int)U(1, 5);
            for (int h = 0; h < hubs; ++h) {
                const int64_t hub = U(0, n - 1), d = std::min<int64_t>(n - 1, U(60, 4000));
                for (int64_t k = 0; k < d; ++k) add(hub, U(0, n - 1));
            }
            for (int64_t k = U(0, n); k > 0; --k) add(U(0, n / 2), U(0, n / 2));        // the upper half stays mostly isolated
        } else if (kind == 5) {                                                        // path + chords
            for (int64_t i = 0; i + 1 < n; ++i) add(i, i + 1);
            for (int64_t i = 0; i + 2 < n; i += U(1, 5)) add(i, i + 2);
        } else {
            const int64_t m = U(kind == 1 ? 0 : 1, std::min<int64_t>(n * (n - 1) / 2, n * U(1, 12)));
            for (int64_t k = 0; k < m; ++k) add(U(0, n - 1), U(0, n - 1));
        }
    }
    for (auto& p : e) { g.eu.push_back(p.first); g.ev.push_back(p.second); }
    g.erowptr.assign((size_t)n + 1, 0);
    for (int32_t u : g.eu) ++g.erowptr[(size_t)u + 1];
    for (int64_t i = 0; i < n; ++i) g.erowptr[(size_t)i + 1] += g.erowptr[(size_t)i];
    std::vector<std::vector<int32_t>> adj((size_t)n);
    for (size_t k = 0; k < g.eu.size(); ++k) { adj[(size_t)g.eu[k]].push_back(g.ev[k]); adj[(size_t)g.ev[k]].push_back(g.eu[k]); }
    g.rowptr.assign((size_t)n + 1, 0);
    g.max_deg = 0;
    for (int64_t i = 0; i < n; ++i) {
        std::sort(adj[(size_t)i].begin(), adj[(size_t)i].end());
        g.rowptr[(size_t)i + 1] = g.rowptr[(size_t)i] + (int32_t)adj[(size_t)i].size();
        g.max_deg = std::max(g.max_deg, (int32_t)adj[(size_t)i].size());
        for (int32_t v : adj[(size_t)i]) g.col.push_back(v);
    }
    return g;
}

// exact-size heap array: reading or writing element [size] is an ASAN report
template <class T>
struct Exact {
    T* p;
    size_t n;
    explicit Exact(size_t n_) : p(new T[n_ ? n_ : 0]()), n(n_) {}
    ~Exact() { delete[] p; }
    T& operator[](size_t i) { return p[i]; }
};

static void run_builders(const Graph& g, std::mt19937_64& rng) {
    const int64_t N = g.n, nnz = (int64_t)g.col.size();
    const int32_t* rp = g.rowptr.data();
    const int32_t* col = g.col.empty() ? nullptr : g.col.data();
    // ---- rls_graph_sweep_schedule: every node once, followed by its row; a batch never mixes levels
    {
        Exact<int32_t> flagged((size_t)N + 1), stream((size_t)(nnz + N));
        int64_t nb = -1, nl = -1;
        CHECK(rls_graph_sweep_schedule(rp, col, N, 64, 768, flagged.p, stream.p, &nb, &nl) == RLS_OK, "%s", rls_last_error_string());
        std::vector<char> seen((size_t)N, 0);
        std::vector<int32_t> level((size_t)N, -1);
        int32_t cur = -1;
        for (int64_t k = 0; k < N; ++k) {
            const uint32_t f = (uint32_t)flagged[(size_t)k];
            const int64_t off = f & 0x7fffffffu;
            const int32_t node = stream[(size_t)off];
            CHECK(node >= 0 && node < N && !seen[(size_t)node], "node %d scheduled twice", node);
            seen[(size_t)node] = 1;
            if (f >> 31) ++cur;
            level[(size_t)node] = cur;
            const int64_t end = (uint32_t)flagged[(size_t)k + 1] & 0x7fffffffu;
            CHECK(end - off == 1 + rp[node + 1] - rp[node], "row length of node %d", node);
            for (int64_t j = 0; j < rp[node + 1] - rp[node]; ++j) CHECK(stream[(size_t)(off + 1 + j)] == col[rp[node] + j], "row of node %d", node);
        }
        // batches respect the sequential order: a lower-numbered neighbour sits in an earlier batch
        for (int64_t i = 0; i < N; ++i)
            for (int32_t j = rp[i]; j < rp[i + 1]; ++j)
                if (col[j] < i) CHECK(level[(size_t)col[j]] < level[(size_t)i], "dependency %d -> %lld", col[j], (long long)i);
        CHECK(nb >= (N > 0) && nl >= (N > 0), "counts");
    }
    // ---- rls_graph_sweep_batches
    {
        Exact<int32_t> flagged((size_t)N + 1);
        int64_t nb = -1;
        const int32_t mn = (int32_t)(1 + rng() % 64), me = (int32_t)(1 + rng() % 1000);
        CHECK(rls_graph_sweep_batches(rp, col, N, mn, me, flagged.p, &nb) == RLS_OK, "%s", rls_last_error_string());
        CHECK(((uint32_t)flagged[(size_t)N] & 0x7fffffffu) == (uint32_t)rp[N], "end offset");
    }
    // ---- rls_graph_ell: sizing call, then an exactly-sized table
    {
        const int64_t groups = (N + 63) / 64;
        Exact<int32_t> ptr((size_t)groups + 1);
        int64_t total = -1;
        CHECK(rls_graph_ell(rp, col, N, ptr.p, nullptr, 0, &total) == RLS_OK, "%s", rls_last_error_string());
        Exact<int32_t> ell((size_t)total);
        CHECK(rls_graph_ell(rp, col, N, ptr.p, ell.p, total, &total) == RLS_OK, "%s", rls_last_error_string());
        CHECK(ptr[(size_t)groups] == total, "ell_ptr end");
        for (int64_t gq = 0; gq < groups; ++gq) {
            const int64_t rounds = (ptr[(size_t)gq + 1] - ptr[(size_t)gq]) / 64;
            for (int64_t l = 0; l < 64; ++l) {
                // a lane's column = its node's neighbours in the builder's (bank-spreading) order + the node itself as padding:
                // the same multiset as the CSR row, nothing else
                const int64_t node = gq * 64 + l;
                std::vector<int32_t> have, want;
                for (int64_t k = 0; k < rounds; ++k) have.push_back(ell[(size_t)(ptr[(size_t)gq] + 64 * k + l)]);
                if (node < N) {
                    for (int64_t j = rp[node]; j < rp[node + 1]; ++j) want.push_back(col[j]);
                    while ((int64_t)want.size() < rounds) want.push_back((int32_t)node);
                    std::sort(have.begin(), have.end());
                    std::sort(want.begin(), want.end());
                    CHECK(have == want, "ell column of node %lld", (long long)node);
                } else {
                    for (int32_t v : have) CHECK(v >= 0 && v < std::max<int64_t>(N, 1) + 64, "ell padding %d", v);
                }
            }
        }
        if (total > 0) CHECK(rls_graph_ell(rp, col, N, ptr.p, ell.p, total - 1, &total) != RLS_OK, "a short table must be refused");
    }
    // ---- rls_graph_sweep_levels (N < 2^20, degrees < 4096)
    if (g.max_deg < 4096) {
        int64_t ng = -1, tot = -1;
        CHECK(rls_graph_sweep_levels(rp, col, N, nullptr, 0, nullptr, 0, &ng, &tot) == RLS_OK, "%s", rls_last_error_string());
        Exact<int32_t> lvp((size_t)ng + 1), lvd((size_t)tot);
        CHECK(rls_graph_sweep_levels(rp, col, N, lvp.p, ng + 1, lvd.p, tot, &ng, &tot) == RLS_OK, "%s", rls_last_error_string());
        std::vector<char> seen((size_t)N, 0);
        for (int64_t q = 0; q < ng; ++q) {
            const int64_t off = (uint32_t)lvp[(size_t)q] & 0x3fffffffu, end = (uint32_t)lvp[(size_t)q + 1] & 0x3fffffffu;
            CHECK(off + 64 <= end && end <= tot, "group %lld bounds", (long long)q);
            const bool hub = ((uint32_t)lvp[(size_t)q] >> 30) & 1;
            if (hub) {
                const int32_t node = lvd[(size_t)off];
                CHECK(node >= 0 && node < N && !seen[(size_t)node], "hub node");
                seen[(size_t)node] = 1;
            } else {
                for (int l = 0; l < 64; ++l) {
                    const int32_t w = lvd[(size_t)(off + l)], node = w & 0xfffff, lg = (w >> 28) & 7;
                    if (node < N && (l & ((1 << lg) - 1)) == 0) {          // the first lane of a row's lanes
                        CHECK(!seen[(size_t)node], "node %d twice", node);
                        seen[(size_t)node] = 1;
                    }
                }
            }
        }
        for (int64_t i = 0; i < N; ++i) CHECK(seen[(size_t)i], "node %lld missing from the level schedule", (long long)i);
        if (tot > 0) {
            int64_t a, b;
            CHECK(rls_graph_sweep_levels(rp, col, N, lvp.p, ng + 1, lvd.p, tot - 1, &a, &b) != RLS_OK, "a short table must be refused");
        }
    }
    // ---- rls_mcpg_visit_levels (degrees < 1024) on a random visiting order and on the degree-descending one
    if (g.max_deg < 1024) {
        for (int pass = 0; pass < 2; ++pass) {
            std::vector<int32_t> order((size_t)N);
            for (int64_t i = 0; i < N; ++i) order[(size_t)i] = (int32_t)i;
            if (pass == 0) std::shuffle(order.begin(), order.end(), rng);
            else std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return rp[a + 1] - rp[a] > rp[b + 1] - rp[b]; });
            int64_t ng = -1, tot = -1;
            CHECK(rls_mcpg_visit_levels(rp, col, N, order.data(), nullptr, 0, nullptr, 0, &ng, &tot) == RLS_OK, "%s", rls_last_error_string());
            Exact<int32_t> lvp((size_t)ng + 1), lvd((size_t)tot);
            CHECK(rls_mcpg_visit_levels(rp, col, N, order.data(), lvp.p, ng + 1, lvd.p, tot, &ng, &tot) == RLS_OK, "%s", rls_last_error_string());
            for (int64_t q = 0; q < ng; ++q) {
                const int64_t off = (uint32_t)lvp[(size_t)q] & 0x3fffffffu, end = (uint32_t)lvp[(size_t)q + 1] & 0x3fffffffu;
                CHECK(off + 128 <= end && end <= tot, "visit group %lld bounds", (long long)q);
            }
            if (N > 1) {                      // a non-permutation is refused, not read out of bounds
                order[0] = order[1];
                CHECK(rls_mcpg_visit_levels(rp, col, N, order.data(), nullptr, 0, nullptr, 0, &ng, &tot) != RLS_OK, "duplicate order entry");
            }
        }
    }
}

static void run_oracle(const Graph& g, std::mt19937_64& rng) {
    const int64_t N = g.n, E = (int64_t)g.eu.size(), B = 1 + (int64_t)(rng() % 9);
    Exact<uint8_t> xs((size_t)(B * N));
    for (size_t i = 0; i < xs.n; ++i) xs[i] = (uint8_t)(rng() & 1);
    Exact<int64_t> obj((size_t)B), last((size_t)B), act((size_t)B), rew((size_t)B), before((size_t)B), cd((size_t)(B * N));
    const int32_t* eu = E ? g.eu.data() : nullptr;
    const int32_t* ev = E ? g.ev.data() : nullptr;
    orc_maxcut_obj(xs.p, B, N, eu, ev, E, 0, obj.p);
    orc_node_cutdeg(xs.p, B, N, g.erowptr.data(), ev, cd.p);
    for (int64_t b = 0; b < B; ++b) {
        int64_t s = 0;
        for (int64_t i = 0; i < N; ++i) s += cd[(size_t)(b * N + i)];
        CHECK(s == obj[(size_t)b], "sum of stored-adjacency cut degrees == cut");
        last[(size_t)b] = before[(size_t)b] = obj[(size_t)b];
        act[(size_t)b] = (int64_t)(rng() % (uint64_t)N);
    }
    orc_step_u8(xs.p, B, N, act.p, eu, ev, E, 0, last.p, rew.p);
    orc_maxcut_obj(xs.p, B, N, eu, ev, E, 0, obj.p);
    for (int64_t b = 0; b < B; ++b)
        CHECK(obj[(size_t)b] == last[(size_t)b] && rew[(size_t)b] == last[(size_t)b] - before[(size_t)b], "step keeps the objective");
    if (N <= 700) {
        orc_greedy_sweep(xs.p, B, N, eu, ev, E, 0, last.p);
        orc_maxcut_obj(xs.p, B, N, eu, ev, E, 0, obj.p);
        for (int64_t b = 0; b < B; ++b) CHECK(obj[(size_t)b] == last[(size_t)b], "sweep keeps the objective");
    }
    const int64_t T = 2 + (int64_t)(rng() % 30);
    Exact<float> dist((size_t)(T * T)), len((size_t)B);
    for (size_t i = 0; i < dist.n; ++i) dist[i] = (float)(rng() % 1000) / 7.0f;
    Exact<int64_t> perm((size_t)(B * T));
    for (int64_t b = 0; b < B; ++b) {
        for (int64_t k = 0; k < T; ++k) perm[(size_t)(b * T + k)] = k;
        std::shuffle(perm.p + b * T, perm.p + (b + 1) * T, rng);
    }
    orc_tsp_tour_length(dist.p, T, perm.p, B, len.p);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 200;
    std::mt19937_64 rng(argc > 2 ? (uint64_t)std::atoll(argv[2]) : 20261003ull);
    CHECK(rls_version() == RLS_ABI_VERSION, "version");
    // argument checks return codes instead of touching memory
    int64_t a = 0, b = 0;
    CHECK(rls_graph_sweep_schedule(nullptr, nullptr, 4, 64, 768, nullptr, nullptr, &a, &b) == RLS_EINVAL, "null rowptr");
    CHECK(rls_last_error_string()[0] != 0, "error text recorded");
    for (int it = 0; it < iters; ++it) {
        const int kind = it < 6 ? it : (int)(rng() % 6);
        Graph g = make_graph(rng, kind);
        run_builders(g, rng);
        run_oracle(g, rng);
    }
    std::printf("host_sanitize: %d random graphs through the host builders and the C oracle, clean\n", iters);
    return 0;
}

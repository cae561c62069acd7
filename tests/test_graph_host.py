"""Host-side graph layer: parsing, CSR construction, generators (CPU only)."""
import numpy as np
import pytest

from rlsolver_amd import graph as G


def test_read_write_roundtrip(tmp_path):
    g = [(0, 1, 1), (0, 2, 1), (2, 3, 5)]
    p = tmp_path / "g.txt"
    G.write_mygraph(str(p), g, 4)
    assert G.read_mygraph(str(p)) == g
    assert G.read_graph_header(str(p)) == (4, 3)
    # comments and blank lines are tolerated
    p.write_text("// header follows\n4 3\n1 2 1\n\n1 3 1 // inline\n3 4 5\n")
    assert G.read_mygraph(str(p)) == g
    n, eu, ev, w = G.read_edge_arrays(str(p))
    assert n == 4 and eu.tolist() == [0, 0, 2] and ev.tolist() == [1, 2, 3] and w.tolist() == [1, 1, 5]
    assert G.load_mygraph2(graph_name=str(p)) == g
    with pytest.raises(ValueError):
        G.load_mygraph2(dataDir=str(tmp_path), graph_name="nope")


def test_golden_graphs_csr(golden):
    z = golden("maxcut_obj")
    for name in z["names"]:
        arr = z[f"{name}/graph"]
        mg = [tuple(int(v) for v in r) for r in arr]
        n = G.calc_num_nodes_in_mygraph(mg)
        assert n == int(z[f"{name}/bidir0/num_nodes"])
        for bidir in (False, True):
            csr = G.build_csr(mg, num_nodes=n, if_bidirectional=bidir)
            assert csr.num_stored_edges == len(mg) * (2 if bidir else 1)
            assert np.array_equal(np.bincount(csr.eu, minlength=n)[None, :], z[f"{name}/bidir{int(bidir)}/n0_num_n1"])
            assert (np.diff(csr.eu) >= 0).all()
            # symmetric CSR: every edge appears in both rows, sorted
            assert csr.nnz == 2 * len(mg)
            for i in range(n):
                row = csr.col[csr.rowptr[i]:csr.rowptr[i + 1]]
                assert (np.diff(row) >= 0).all()
            deg = np.zeros(n, int)
            for a, b, _ in mg:
                deg[a] += 1
                deg[b] += 1
            assert np.array_equal(csr.degree, deg)
            n1s, _ = G.build_adjacency_indies(mg, bidir)
            off = np.concatenate([[0], np.cumsum([len(x) for x in n1s])])
            assert np.array_equal(np.concatenate(n1s), csr.ev)
            assert off[-1] == csr.num_stored_edges
        adj = G.build_adjacency_bool(mg, n, True)
        assert adj.sum() == 2 * len({(min(a, b), max(a, b)) for a, b, _ in mg})


def test_csr_edge_cases():
    csr = G.build_csr([(0, 1, 1), (0, 1, 1), (2, 2, 1)], num_nodes=4)
    assert csr.num_stored_edges == 3 and csr.nnz == 4      # self loop dropped from CSR, multi-edge kept
    assert csr.degree.tolist() == [2, 2, 0, 0] and csr.max_degree == 2
    with pytest.raises(ValueError):
        G.build_csr([(0, 5, 1)], num_nodes=3)
    e = G.build_csr([], num_nodes=3)
    assert e.nnz == 0 and e.rowptr.tolist() == [0, 0, 0, 0]


def test_generators_deterministic():
    a = G.generate_gnm(200, 1000, seed=3)
    assert a == G.generate_gnm(200, 1000, seed=3) and a != G.generate_gnm(200, 1000, seed=4)
    assert len(a) == 1000 and len({(u, v) for u, v, _ in a}) == 1000 and all(u < v for u, v, _ in a)
    b = G.generate_ba(300, 5, seed=5)
    assert len(b) == 5 * (300 - 5) and len(set(b)) == len(b)
    assert G.calc_num_nodes_in_mygraph(b) == 300
    g, n, m = G.generate_mygraph("ER", 50, seed=1)
    assert n == 50 and m == len(g)
    mg, n22, real = G.load_gset(22, data_dir="/nonexistent")
    assert (n22, len(mg), real) == (2000, 19990, False)
    with pytest.raises(ValueError):
        G.generate_gnm(4, 7, 0)


def test_tsp_tables_small():
    coords = np.array([[0, 0], [3, 4], [6, 8], [0, 1]], dtype=np.float64)
    d, near, rnd = G.tsp_tables(coords, K=2)
    assert d.dtype == np.float32 and d[0, 1] == 5 and d[0, 2] == 10 and d[1, 1] == 0
    assert near.tolist()[0] == [3, 1] and rnd.tolist()[2] == [0, 1, 3]
    assert G.generate_tsp_coords(100, 100).shape == (100, 2)


def test_sweep_level_schedule_is_a_valid_reordering_of_the_sequential_pass():
    """rls_graph_sweep_schedule (host, in the C library): positions are a permutation; nodes of one batch are
    pairwise non-adjacent; for every edge the lower-numbered endpoint sits in an earlier batch -- together
    these make the batched sweep equal to the sequential one (envs/env_L2A.py:109-116)."""
    import ctypes as C
    from rlsolver_amd import _abi
    from rlsolver_amd.graph import build_csr, generate_gnm, generate_ba
    for g, n in ((generate_gnm(300, 1500, seed=3), 300), (generate_ba(200, 4, seed=1), 200), ([(0, 1, 1)], 5),
                 (generate_gnm(2000, 19990, seed=22), 2000)):
        csr = build_csr(g, num_nodes=n, if_bidirectional=False)
        rp = np.ascontiguousarray(csr.rowptr, dtype=np.int32)
        col = np.ascontiguousarray(csr.col, dtype=np.int32)
        flagged = np.empty(n + 1, dtype=np.int32)
        stream = np.empty(csr.nnz + n, dtype=np.int32)
        nb, nl = C.c_int64(0), C.c_int64(0)
        _abi.call("rls_graph_sweep_schedule", rp.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p), n, 64, 768,
                  flagged.ctypes.data_as(C.c_void_p), stream.ctypes.data_as(C.c_void_p), C.byref(nb), C.byref(nl))
        off = flagged.view(np.uint32) & 0x7FFFFFFF
        first = (flagged.view(np.uint32)[:n] >> 31).astype(bool)
        assert first[0] and off[n] == csr.nnz + n and first.sum() == nb.value and 1 <= nl.value <= nb.value
        nodes = stream[off[:n]]
        assert sorted(nodes.tolist()) == list(range(n))
        batch_of = np.empty(n, np.int64)
        batch_of[nodes] = np.cumsum(first) - 1
        for k in range(n):   # the stream carries each node's CSR row
            i = nodes[k]
            assert off[k + 1] - off[k] == 1 + rp[i + 1] - rp[i]
            assert np.array_equal(stream[off[k] + 1: off[k + 1]], col[rp[i]: rp[i + 1]])
            nbrs = col[rp[i]: rp[i + 1]]
            assert (batch_of[nbrs[nbrs < i]] < batch_of[i]).all() and (batch_of[nbrs[nbrs > i]] > batch_of[i]).all()
        sizes = np.bincount(batch_of)
        ents = np.add.reduceat(np.diff(off.astype(np.int64)), np.flatnonzero(first))
        assert sizes.max() <= 64 and ents.max() <= 768


def test_sweep_level_groups_encode_the_same_schedule():
    """rls_graph_sweep_levels: every node appears once (on L = 1, 2, 4 or 8 adjacent, L-aligned lanes, the longest
    rows first), lower-numbered neighbours sit in earlier LEVELS, lane j of a node lists its CSR entries j, j + L, ...
    as byte offsets of the neighbours' words and ends in the node itself, rounds come in whole blocks of 8, no lane holds
    more than 64 entries, and the table ends in sixteen spare rows (the kernels request a group's first two blocks unguarded).  A row of 256 or more entries is a group of its own
    (bit 30 of its offset): node and degree in the header, its neighbours 64 per round, padded with itself."""
    import ctypes as C
    from rlsolver_amd import _abi
    from rlsolver_amd.graph import build_csr, generate_gnm, generate_ba
    for g, n in ((generate_gnm(300, 1500, seed=3), 300), (generate_ba(200, 4, seed=1), 200), ([(0, 1, 1)], 70),
                 (generate_gnm(2000, 19990, seed=22), 2000), (generate_gnm(260, 26000, seed=5), 260),
                 ([(0, j, 1) for j in range(1, 700)] + [(j, j + 1, 1) for j in range(1, 699, 2)], 700), (generate_ba(5000, 10, seed=4), 5000)):
        csr = build_csr(g, num_nodes=n, if_bidirectional=False)
        rp = np.ascontiguousarray(csr.rowptr, dtype=np.int32)
        col = np.ascontiguousarray(csr.col, dtype=np.int32)
        a = (rp.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p), n)
        ng, tot = C.c_int64(0), C.c_int64(0)
        _abi.call("rls_graph_sweep_levels", *a, None, 0, None, 0, C.byref(ng), C.byref(tot))
        lvp = np.empty(ng.value + 1, dtype=np.int32)
        lvd = np.empty(tot.value, dtype=np.int32)
        _abi.call("rls_graph_sweep_levels", *a, lvp.ctypes.data_as(C.c_void_p), lvp.size, lvd.ctypes.data_as(C.c_void_p),
                  lvd.size, C.byref(ng), C.byref(tot))
        off = (lvp.view(np.uint32) & 0x3FFFFFFF).astype(np.int64)
        first = (lvp.view(np.uint32)[:-1] >> 31).astype(bool)
        is_hub = ((lvp.view(np.uint32)[:-1] >> 30) & 1).astype(bool)
        assert is_hub.sum() == int((np.diff(rp) >= 256).sum())
        assert first[0] and off[-1] + 16 * 64 == tot.value and (lvd[off[-1]:] == n * 8).all()      # sixteen spare rows (ABI v11)
        level_of_group = np.cumsum(first) - 1
        level_of = np.full(n, -1)
        seen = []
        for k in range(ng.value):
            raw = lvd[off[k]: off[k + 1]]
            rounds = raw.size // 64 - 1
            assert rounds % 8 == 0 and rounds <= 64
            # ABI v11 record: 64 header words, then per block of 8 rounds two slabs of [64 lanes][4 rounds] -- round r of lane l at
            # 64 + 512 (r / 8) + 256 ((r / 4) % 2) + 4 l + r % 4.  Decoded back to [1 + rounds][64] for the checks below
            rec = np.empty((1 + rounds, 64), dtype=raw.dtype)
            rec[0] = raw[:64]
            for r in range(rounds):
                rec[1 + r] = raw[64 + 512 * (r // 8) + 256 * ((r // 4) % 2) + 4 * np.arange(64) + r % 4]
            if is_hub[k]:
                i, deg = int(rec[0, 0]), int(rec[0, 1])
                assert deg == rp[i + 1] - rp[i] >= 256 and (rec[0, 2:] == n).all() and rounds * 64 >= deg
                flat = rec[1:].reshape(-1)
                assert np.array_equal(flat[:deg], col[rp[i]: rp[i + 1]] * 8) and (flat[deg:] == i * 8).all()
                seen.append(i)
                level_of[i] = level_of_group[k]
                continue
            hdr = rec[0].view(np.uint32)
            nodes, half, lcode = (hdr & 0xFFFFF).astype(np.int64), (hdr >> 20) & 0xFF, (hdr >> 28) & 3
            live = nodes < n
            assert (np.diff(lcode[live].astype(int)) <= 0).all()             # longest rows (most lanes) first
            ln = 0
            while ln < 64 and live[ln]:
                i, L = nodes[ln], 1 << int(lcode[ln])
                assert ln % L == 0 and (nodes[ln: ln + L] == i).all() and (lcode[ln: ln + L] == lcode[ln]).all()
                deg = rp[i + 1] - rp[i]
                assert (half[ln: ln + L] == deg // 2).all()
                row = col[rp[i]: rp[i + 1]]
                for j in range(L):
                    mine = row[j::L]
                    assert len(mine) <= rounds
                    # the lane's column = its share of the row in the builder's (LDS-bank-spreading) order, padded with the node itself:
                    # the same multiset, nothing else
                    want = np.sort(np.concatenate([mine * 8, np.full(rounds - len(mine), i * 8, dtype=mine.dtype)]))
                    assert np.array_equal(np.sort(rec[1:, ln + j]), want)
                seen.append(int(i))
                level_of[i] = level_of_group[k]
                ln += L
            assert not live[ln:].any() and (rec[1:, ~live] == n * 8).all()
        assert sorted(seen) == list(range(n))
        for i in range(n):
            nbrs = col[rp[i]: rp[i + 1]]
            assert (level_of[nbrs[nbrs < i]] < level_of[i]).all() and (level_of[nbrs[nbrs > i]] > level_of[i]).all()


def test_streaming_reader_matches_line_reader_and_scales(tmp_path):
    """the chunked reader gives what the per-line reader gives (comments, blank lines, float
    weights, rows without a weight, blocks cut anywhere) and parses a 2*10^6-edge file in a few seconds."""
    import time
    from rlsolver_amd.graph import read_edge_arrays, read_mygraph
    p = tmp_path / "g.txt"
    p.write_text("// a comment\n5 6\n1 2 1\n2 3 -1\n\n3 4 1.0  // trailing\n4 5 2\n5 1 -3\n1 3 1\n")
    n, eu, ev, w = read_edge_arrays(str(p), chunk_bytes=7)           # blocks far smaller than a line
    mg = read_mygraph(str(p))
    assert n == 5 and list(zip(eu.tolist(), ev.tolist(), w.tolist())) == mg
    q = tmp_path / "nw.txt"
    q.write_text("4 3\n1 2\n2 3\n3 4\n")
    n, eu, ev, w = read_edge_arrays(str(q))
    assert n == 4 and w.tolist() == [1, 1, 1] and eu.tolist() == [0, 1, 2]
    (tmp_path / "e.txt").write_text("7 0\n")
    assert read_edge_arrays(str(tmp_path / "e.txt"))[1].size == 0
    # 10^6 nodes, 2*10^6 edges
    rng = np.random.RandomState(0)
    N, E = 1_000_000, 2_000_000
    a = rng.randint(1, N + 1, size=(E, 2))
    big = tmp_path / "big.txt"
    with open(big, "w") as f:
        f.write(f"{N} {E}\n")
        np.savetxt(f, np.column_stack([a, rng.choice([-1, 1], size=E)]), fmt="%d")
    t0 = time.time()
    n, eu, ev, w = read_edge_arrays(str(big), chunk_bytes=8 << 20)
    dt = time.time() - t0
    assert n == N and eu.shape == (E,) and np.array_equal(eu, a[:, 0] - 1) and np.array_equal(ev, a[:, 1] - 1)
    assert set(np.unique(w)) == {-1, 1} and dt < 30


def test_read_edge_arrays_validates_rows_ids_and_count(tmp_path):
    """Rows with and without a weight mixed parse line by line (as the reference's reader does), never as a reshaped token
    soup; node ids outside [1, n] raise; an edge count that contradicts the header warns."""
    from rlsolver_amd.graph import read_edge_arrays
    p = tmp_path / "mixed.txt"
    p.write_text("4 4\n1 2 5\n2 3\n3 4\n1 4\n")                 # 9 tokens: used to become 3 bogus weighted rows
    n, eu, ev, w = read_edge_arrays(str(p))
    assert n == 4 and eu.tolist() == [0, 1, 2, 0] and ev.tolist() == [1, 2, 3, 3] and w.tolist() == [5, 1, 1, 1]
    q = tmp_path / "blank.txt"
    q.write_text("3 2\n\n1 2 7\n\n2 3 9\n")
    n, eu, ev, w = read_edge_arrays(str(q))
    assert (n, eu.tolist(), ev.tolist(), w.tolist()) == (3, [0, 1], [1, 2], [7, 9])
    for body in ("3 1\n0 2\n", "3 1\n1 4\n", "3 2\n1 2\n5 1 1\n"):
        r = tmp_path / "bad.txt"
        r.write_text(body)
        with pytest.raises(ValueError, match="outside"):
            read_edge_arrays(str(r))
    r = tmp_path / "bad4.txt"
    r.write_text("3 1\n1 2 3 4\n")
    with pytest.raises(ValueError):
        read_edge_arrays(str(r))
    s = tmp_path / "short.txt"
    s.write_text("3 5\n1 2\n2 3\n")
    with pytest.warns(UserWarning, match="announces 5 edges"):
        read_edge_arrays(str(s))


def test_load_data_builds_the_isco_tsp_params(tmp_path):
    """ISCO/util_TSP.py:5-23: TSPLIB file -> params_dict (distance f32, K nearest without self, all-but-self table)."""
    import torch
    from rlsolver_amd.graph import generate_tsp_coords, load_data, tsp_tables
    c = generate_tsp_coords(12, seed=4)
    p = tmp_path / "t12.tsp"
    p.write_text("NAME: t12\nTYPE: TSP\nDIMENSION: 12\nEDGE_WEIGHT_TYPE: EUC_2D\nNODE_COORD_SECTION\n"
                 + "".join(f"{i + 1} {x!r} {y!r}\n" for i, (x, y) in enumerate(c.tolist())) + "EOF\n")
    d = load_data(str(p), K=5)
    dist, near, rnd = tsp_tables(np.asarray(c, dtype=np.float32), 5)
    assert d["num_nodes"] == 12 and d["distance"].dtype == torch.float32 and d["nearest_indices"].dtype == torch.int64
    assert np.array_equal(d["distance"].numpy(), dist) and np.array_equal(d["nearest_indices"].numpy(), near)
    assert np.array_equal(d["random_indices"].numpy(), rnd) and d["random_indices"].shape == (12, 11)


def test_mcpg_visit_level_table_layout_abi_v11():
    """rls_mcpg_visit_levels (ABI v11): lane-major records -- words 2 l, 2 l + 1 = lane l's two header words, round r of lane l at
    128 + 512 (r / 8) + 256 ((r / 4) % 2) + 4 l + r % 4 --, every visiting position on L adjacent lanes once, a lane's entries =
    its share of the node's neighbours as LDS byte offsets with bit 31 set on the not-yet-visited ones, sixteen spare rows, then the
    SAME table again with the fresh flags cleared (headers untouched) at offset (lv_ptr[G] & 0x3fffffff) + 1024."""
    import ctypes as C
    from rlsolver_amd import _abi
    from rlsolver_amd.graph import build_csr, generate_gnm, generate_ba
    for g, n in ((generate_gnm(300, 1500, seed=3), 300), (generate_ba(400, 5, seed=1), 400), ([(0, 1, 1)], 70),
                 ([(0, j, 1) for j in range(1, 500)] + [(j, j + 1, 1) for j in range(1, 499, 2)], 500)):
        csr = build_csr(g, num_nodes=n, if_bidirectional=False)
        rp = np.ascontiguousarray(csr.rowptr, dtype=np.int32)
        col = np.ascontiguousarray(csr.col, dtype=np.int32)
        deg = np.diff(rp)
        order = np.argsort(-deg, kind="stable").astype(np.int32)
        pos_of = np.empty(n, dtype=np.int64)
        pos_of[order] = np.arange(n)
        a = (rp.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p), n, order.ctypes.data_as(C.c_void_p))
        ng, tot = C.c_int64(0), C.c_int64(0)
        _abi.call("rls_mcpg_visit_levels", *a, None, 0, None, 0, C.byref(ng), C.byref(tot))
        lvp = np.empty(ng.value + 1, dtype=np.int32)
        lvd = np.full(tot.value, -12345, dtype=np.int32)
        _abi.call("rls_mcpg_visit_levels", *a, lvp.ctypes.data_as(C.c_void_p), lvp.size, lvd.ctypes.data_as(C.c_void_p), lvd.size,
                  C.byref(ng), C.byref(tot))
        off = (lvp.view(np.uint32) & 0x3FFFFFFF).astype(np.int64)
        is_hub = ((lvp.view(np.uint32)[:-1] >> 30) & 1).astype(bool)
        half = off[-1] + 16 * 64
        assert tot.value == 2 * half and (lvd[off[-1]: half] == n * 8).all() and not (lvd == -12345).any()
        first, clean = lvd[:half].view(np.uint32), lvd[half:].view(np.uint32)
        seen = []
        for k in range(ng.value):
            raw, cl = first[off[k]: off[k + 1]], clean[off[k]: off[k + 1]]
            rounds = raw.size // 64 - 2
            assert rounds % 8 == 0 and rounds >= 0
            assert np.array_equal(raw[:128], cl[:128]) and np.array_equal(raw[128:] & 0x7FFFFFFF, cl[128:]) and not (cl[128:] >> 31).any()
            ent = np.empty((rounds, 64), dtype=np.uint32)
            for r in range(rounds):
                ent[r] = raw[128 + 512 * (r // 8) + 256 * ((r // 4) % 2) + 4 * np.arange(64) + r % 4]
            h0, h1 = raw[0:128:2], raw[1:128:2]
            if is_hub[k]:
                i, p, d = int(h0[0] & 0xFFFFF), int(h1[0] & 0xFFFFF), int(raw[4])
                assert order[p] == i and d == deg[i] > 128
                nbrs = col[rp[i]: rp[i + 1]]
                want = nbrs.astype(np.uint32) * 8 | np.where(pos_of[nbrs] > p, np.uint32(1 << 31), np.uint32(0))
                flat = ent.reshape(-1)                                   # neighbour e in round e / 64 of lane e % 64
                assert np.array_equal(np.sort(flat[:d]), np.sort(want)) and (flat[d:] == n * 8).all()
                seen.append(p)
                continue
            ln = 0
            while ln < 64 and (h0[ln] & 0xFFFFF) < n:
                i, p, L = int(h0[ln] & 0xFFFFF), int(h1[ln] & 0xFFFFF), 1 << int((h0[ln] >> 28) & 3)
                assert order[p] == i and ln % L == 0 and ((h0[ln: ln + L] & 0xFFFFF) == i).all()
                assert int((h0[ln] >> 20) & 0xFF) == (deg[i] + 1) // 2 and bool(h0[ln] >> 31) == (deg[i] % 2 == 0)
                nbrs = col[rp[i]: rp[i + 1]]
                nfresh = int((pos_of[nbrs] > p).sum())
                assert int((h1[ln] >> 20) & 0xFF) == (deg[i] + nfresh + 1) // 2 and bool(h1[ln] >> 31) == ((deg[i] + nfresh) % 2 == 0)
                want = nbrs.astype(np.uint32) * 8 | np.where(pos_of[nbrs] > p, np.uint32(1 << 31), np.uint32(0))
                got = np.concatenate([ent[:, ln + j] for j in range(L)])
                got = got[got != n * 8]
                assert np.array_equal(np.sort(got), np.sort(want))          # the row, dealt over its L lanes in any order, padding apart
                for j in range(L):
                    assert (ent[:, ln + j] != n * 8).sum() == len(nbrs[j::L])
                seen.append(p)
                ln += L
            assert ((h0[ln:] & 0xFFFFF) >= n).all() and (ent[:, ln:] == n * 8).all()
        assert sorted(seen) == list(range(n))

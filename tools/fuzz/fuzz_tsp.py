"""Differential fuzz of the TSP kernels against the numpy restatements: tour length (f32, 1e-5 relative), the 2-opt reversal
delta (f64), the best-improvement 2-opt pass in both rankings and the whole local_search_2_opt (routes and float64 distances
bit for bit), on Euclidean, integer (ties everywhere) and asymmetric matrices.  `python tools/fuzz/fuzz_tsp.py [seconds] [seed]`."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import oracle_np as onp
from rlsolver_amd import ops_mcpg_tsp as mops
from rlsolver_amd import _abi; _abi.tuning_from_env()   # RLS_<KNOB> variables -> rls_tuning_set (forced forms)
from rlsolver_amd.methods import tsp_opt_2 as t2

DEV = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    N = int(rng.choice([rng.randint(3, 9), rng.randint(9, 30), rng.randint(30, 70)]))
    kind = rng.choice(["euclid", "integer", "asym"])
    c = rng.rand(N, 2) * 100
    d = np.sqrt(((c[:, None] - c[None]) ** 2).sum(-1))
    if kind == "integer":
        d = np.rint(d / 20.0)
    elif kind == "asym":
        d = d + rng.rand(N, N) * 5
    np.fill_diagonal(d, 0)
    B = int(rng.randint(1, 40))
    perms = np.stack([rng.permutation(N) for _ in range(B)])
    dd = torch.from_numpy(d).to(DEV)
    pp = torch.from_numpy(perms).to(DEV)
    tag = f"it={it} N={N} kind={kind} B={B}"
    # K12
    d32 = d.astype(np.float32)
    got = mops.tsp_tour_length(torch.from_numpy(d32).to(DEV), pp).cpu().numpy()
    want = onp.tsp_tour_length_f64(d32, perms)
    assert np.allclose(got, want, rtol=1e-5, atol=0), "K12 " + tag
    # K12 / K13 / 2-opt delta at sizes on both sides of "D fits LDS" (N ~ 200), the swap with recorded partner draws
    if it % 3 == 0:
        N2 = int(rng.choice([rng.randint(4, 60), rng.randint(60, 200), rng.randint(200, 420)]))
        B2 = int(rng.choice([1, 5, 64, 65, 300]))
        c2 = rng.rand(N2, 2).astype(np.float32)
        dist2 = np.sqrt(((c2[:, None] - c2[None]) ** 2).sum(-1)).astype(np.float32)
        pn = np.stack([rng.permutation(N2) for _ in range(B2)])
        d2, p2 = torch.from_numpy(dist2).to(DEV), torch.from_numpy(pn).to(DEV)
        t2tag = f"{tag} N2={N2} B2={B2}"
        length = mops.tsp_tour_length(d2, p2).cpu().numpy()
        assert np.allclose(length, onp.tsp_tour_length_f64(dist2, pn), rtol=1e-5), "K12 large " + t2tag
        off = rng.randint(1, N2, size=(B2, N2))
        sel = np.take_along_axis(pn, (np.arange(N2)[None, :] + off) % N2, axis=1)
        Tt = float(rng.choice([0.1, 0.5, 2.0]))
        lr_w, idx_w, ban_w = onp.tsp_swap_delta_all(dist2, pn, sel, Tt)
        lr, idx, ban = mops.tsp_swap_delta_all(d2, p2, torch.from_numpy(sel).to(DEV), Tt)
        assert np.array_equal(idx.cpu().numpy(), idx_w) and np.array_equal(ban.cpu().numpy(), ban_w), "K13 indices / ban " + t2tag
        assert np.allclose(lr.cpu().numpy(), lr_w, rtol=1e-5, atol=1e-5 * length.max() / Tt), "K13 logratio " + t2tag
        # K13 as the reference runs it: the partners drawn in the kernel (round 6) -- what it drew, fed back through the recorded-draw
        # hook, reproduces it bit for bit; the byte tables (N <= 256) and the tables read from memory draw the same partners; every
        # partner comes from the position's own tables and is never the position's own city; a shard draws what the batch draws
        if N2 >= 4:
            from rlsolver_amd.graph import tsp_tables
            Kt = int(rng.randint(1, min(20, N2 - 2) + 1))
            _, near, rnd = tsp_tables(c2, K=Kt)
            n32, r32 = torch.from_numpy(near.astype(np.int32)).to(DEV), torch.from_numpy(rnd.astype(np.int32)).to(DEV)
            thr = float(np.float32(Kt / (Kt + 1)))
            sd, eo = int(rng.randint(0, 2 ** 62)), int(rng.randint(0, 10 ** 6))
            t8 = mops.tsp_tables8(n32, r32)
            a = mops.tsp_swap_delta_all(d2, p2, None, Tt, nearest=n32, random=r32, near_threshold=thr, seed=sd, env_offset=eo, return_selected=True)
            if t8 is not None:
                b = mops.tsp_swap_delta_all(d2, p2, None, Tt, nearest=n32, random=r32, near_threshold=thr, seed=sd, env_offset=eo,
                                            return_selected=True, tables8=t8)
                assert all(torch.equal(u, v) for u, v in zip(a, b)), "K13 draw: byte tables vs memory tables " + t2tag
            c = mops.tsp_swap_delta_all(d2, p2, a[3], Tt)
            assert all(torch.equal(u, v) for u, v in zip(a[:3], c)), "K13 draw vs its own partners fed back " + t2tag
            sel_d = a[3].cpu().numpy()
            assert (sel_d != pn).all(), "K13 draw: own city " + t2tag
            allowed = np.zeros((N2, N2), dtype=bool)
            allowed[np.arange(N2)[:, None], near] = True
            allowed[np.arange(N2)[:, None], rnd[:, :N2 - Kt - 1]] = True
            assert allowed[pn, sel_d].all(), "K13 draw: partner outside the tables " + t2tag
            lw, iw, bw = onp.tsp_swap_delta_all(dist2, pn, sel_d, Tt)
            assert np.array_equal(a[1].cpu().numpy(), iw) and np.array_equal(a[2].cpu().numpy(), bw), "K13 draw vs oracle " + t2tag
            if B2 > 1:
                h = B2 // 2
                e = mops.tsp_swap_delta_all(d2, p2[h:].contiguous(), None, Tt, nearest=n32, random=r32, near_threshold=thr, seed=sd,
                                            env_offset=eo + h, tables8=t8)
                assert all(torch.equal(u[h:], v) for u, v in zip(a[:3], e)), "K13 draw: shard " + t2tag
        pos = rng.randint(0, N2, size=B2)
        xw = onp.tsp_switch(pn, pos, idx_w)
        xg = p2.clone()
        mops.tsp_apply_swap(xg, torch.from_numpy(pos).to(DEV), idx)
        assert np.array_equal(xg.cpu().numpy(), xw), "switch " + t2tag
        i2 = rng.randint(0, N2 - 1, size=B2)
        j2 = np.array([rng.randint(a + 1, N2) for a in i2])
        dl = mops.tsp_2opt_delta(d2, p2, torch.from_numpy(i2).to(DEV), torch.from_numpy(j2).to(DEV)).cpu().numpy()
        nb = min(B2, 6)
        want_dl = onp.tsp_2opt_delta(dist2.astype(np.float64), pn, np.arange(nb), i2[:nb], j2[:nb])
        assert np.allclose(dl[:len(want_dl)], want_dl, rtol=0, atol=2e-5 * length.max()), "2-opt delta " + t2tag
    # one exact pass per tour vs the restated loops
    cur = np.array([onp.tsp_distance_calc(d, [int(v) + 1 for v in p] + [int(p[0]) + 1]) for p in perms])
    bi, bj, bv = mops.tsp_2opt_best(dd, pp, torch.from_numpy(cur).to(DEV), slices=int(rng.choice([1, 3, 8])))
    for b in range(min(B, 4)):
        tour = [int(v) + 1 for v in perms[b]] + [int(perms[b][0]) + 1]
        r, dist = onp.tsp_local_search_2_opt(d, tour, cur[b], 1)
        if int(bi[b]) < 0:
            assert r == tour and dist == cur[b], "2-opt none " + tag
        else:
            i, j = int(bi[b]), int(bj[b])
            cand = list(tour)
            cand[i:j + 1] = cand[i:j + 1][::-1]
            cand[-1] = cand[0]
            assert cand == r and float(bv[b]) == dist, f"2-opt pass b={b} " + tag
    # the whole search on one tour
    if N <= 40:
        tour = [int(v) + 1 for v in perms[0]] + [int(perms[0][0]) + 1]
        rs = int(rng.choice([-1, 1, 3]))
        r, dist = onp.tsp_local_search_2_opt(d, tour, cur[0], rs)
        r2, dist2 = t2.local_search_2_opt(d, [tour, cur[0]], recursive_seeding=rs, verbose=False, device=DEV)
        assert r2 == r and dist2 == dist, f"local_search_2_opt rs={rs} " + tag
    # delta ranking on symmetric matrices: the reported delta is the best of all reversal deltas
    if kind != "asym":
        bi, bj, bd = mops.tsp_2opt_best(dd, pp)
        for b in range(min(B, 3)):
            best = 0.0
            for i in range(N - 1):
                for j in range(i + 1, N):
                    if i == 0 and j == N - 1:
                        continue
                    t = perms[b]
                    a_, b_, c_, e_ = t[i - 1], t[i], t[j], t[(j + 1) % N]
                    best = min(best, (d[a_, c_] + d[b_, e_]) - (d[a_, b_] + d[c_, e_]))
            assert float(bd[b]) == best, f"2-opt delta b={b} " + tag
    it += 1
print(f"fuzz_tsp: {it} random configurations, no mismatch")

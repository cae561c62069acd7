"""Fuzz of the ISCO_maxcut sampler step: the workgroup-per-sample kernel (small batches) against the wave-per-sample kernel
(large batches) and both against the float64 oracle on random graphs with the same recorded draws -- the same selected
nodes, log-probability terms within the conditioning-aware tolerance of tests/isco_tol.py, the same accepted samples away from
the accept margin -- plus the invariants of a step (the proposal
differs from x exactly on the selected nodes, an accepted sample IS the proposal).  `python tools/fuzz/fuzz_isco.py [seconds] [seed]`."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import oracle_isco as oi
from tests.isco_tol import RTOL, assert_ll_close, ll_atol
from rlsolver_amd import graph as G
from rlsolver_amd import _abi; _abi.tuning_from_env()   # RLS_<KNOB> variables -> rls_tuning_set (forced forms)
from rlsolver_amd.envs.env_ISCO_maxcut import ISCO_maxcut

DEV = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    kind = rng.choice(["gnm", "ba", "hub"])
    n = int(rng.choice([rng.randint(20, 256), rng.randint(256, 900), rng.randint(900, 3000)]))
    if kind == "gnm":
        g = np.asarray(G.generate_gnm(n, int(rng.randint(n, n * 8)), int(rng.randint(1 << 30))), dtype=np.int64)
    elif kind == "ba":
        g = np.asarray(G.generate_ba(n, int(rng.randint(1, 7)), int(rng.randint(1 << 30))), dtype=np.int64)
    else:
        e = {(0, j) for j in rng.choice(np.arange(1, n), min(n - 1, int(rng.choice([100, 400, 1500]))), replace=False).tolist()}
        for _ in range(2 * n):
            a, b = rng.randint(0, n, 2)
            if a != b:
                e.add((min(a, b), max(a, b)))
        g = np.asarray([(a, b, 1) for a, b in sorted(e)], dtype=np.int64)
    B, Bs = 600, int(rng.choice([1, 7, 64]))
    T = float(rng.choice([0.3, 0.7, 1.5]))
    tag = f"it={it} kind={kind} n={n} E={len(g)} Bs={Bs} T={T}"
    if "-v" in sys.argv:
        print(tag, flush=True)
    mk = lambda b: ISCO_maxcut({"num_nodes": n, "num_edges": len(g), "edge_from": torch.from_numpy(g[:, 0].copy()).to(DEV),
                                "edge_to": torch.from_numpy(g[:, 1].copy()).to(DEV)}, batch_size=b, device=DEV)
    big, small = mk(B), mk(Bs)
    x = torch.from_numpy(rng.randint(0, 2, size=(B, n)).astype(np.float32)).to(DEV)
    pln = rng.randint(1, min(40, n), size=B).astype(np.int64)
    if n >= 1100 and rng.rand() < 0.4:      # a path longer than the wave kernel's selected-set list (512 entries once it trades capacity
        pln[int(rng.randint(0, Bs))] = int(rng.randint(520, n // 2))    # for resident samples): ordered by extraction there, sorted in the other kernel
    pl = torch.from_numpy(pln).to(DEV)
    draws = {"u_gumbel": torch.from_numpy(rng.rand(B, n).astype(np.float32).clip(1e-7, 1 - 1e-7)),
             "u_accept": torch.from_numpy(rng.rand(B).astype(np.float32))}
    _abi.tuning_set("RLS_ISCO_FORCE_WG", 0)      # (from ~1500 nodes on the library would take the workgroup kernel here too)
    try:
        yb, eb, ab, tb, mb = big.step(x, pl, T, draws=draws, want_terms=True)
    finally:
        _abi.tuning_unset("RLS_ISCO_FORCE_WG")
    ds = {k: v[:Bs] for k, v in draws.items()}
    ys, es, as_, ts, ms = small.step(x[:Bs].contiguous(), pl[:Bs].contiguous(), T, draws=ds, want_terms=True)
    if it % 3 == 0 and n >= 256:     # (n >= 256: the workgroup kernel either way) the same samples with the f32 rows in the step's
                                      # scratch (the form N > ~15 900 takes): the same bits
        _abi.tuning_set("RLS_ISCO_GLOBAL_ROWS", 1)
        try:
            got = small.step(x[:Bs].contiguous(), pl[:Bs].contiguous(), T, draws=ds, want_terms=True)
        finally:
            _abi.tuning_unset("RLS_ISCO_GLOBAL_ROWS")
        for a, b in zip((ys, es, as_, ts, ms), got):
            assert torch.equal(a, b), "rows in scratch vs rows in LDS " + tag
    assert bool((mb.sum(1) >= pl).all()), "path length " + tag     # ties at the threshold are all selected (util.py:514-555)
    # the path log-probabilities are ill-conditioned in the reference itself (tests/isco_tol.py): tolerance from the float64 oracle
    r = oi.maxcut_step(x[:Bs].cpu().numpy(), g[:, 0], g[:, 1], pl[:Bs].cpu().numpy(), T, ds["u_gumbel"].numpy(), ds["u_accept"].numpy())
    mass, plh = r["remaining_mass"].copy(), pl[:Bs].cpu().numpy()
    # the ORDER of the draws is the order of log_prob - log(-log(u)); where two of the L + 1 largest sit within a few ulps of
    # each other the order (and with it every path log-probability) depends on the platform's log -- in the reference too
    # (found by this fuzzer: two draws 1 ulp apart moved ll_x2y by 0.08).  Such envs are compared on nothing but the invariants.
    _, lp0 = oi.maxcut_local_dist(x[:Bs].cpu().numpy(), g[:, 0], g[:, 1], T)
    pert = (lp0 - np.log(-np.log(ds["u_gumbel"].numpy()))).astype(np.float32)
    top = -np.sort(-pert, axis=1)[:, :int(plh.max()) + 1]
    gaps = np.abs(np.diff(top, axis=1)) / np.maximum(np.abs(top[:, 1:]), 1e-3)
    near_tie = np.array([(gaps[e, :plh[e]] < 2e-6).any() for e in range(Bs)])
    mass[near_tie] = 0.0                                                        # excluded like ill-conditioned envs
    keep = torch.from_numpy(~near_tie).to(DEV)
    # (the two kernels reduce the softmax normaliser in different orders: their perturbed scores can differ in the last ulp,
    #  so even the SELECTED SET may differ where the L-th and (L+1)-th score nearly tie -- seen once in ~330 000 envs)
    assert torch.equal(mb[:Bs][keep], ms[keep]), "selected nodes " + tag
    assert np.array_equal(ms.cpu().numpy().astype(np.uint8)[~near_tie], r["mask"].astype(np.uint8)[~near_tie]), "selected nodes vs oracle " + tag
    for kern, tt in (("wave", tb[:Bs]), ("workgroup", ts)):
        tt = tt.cpu().numpy()
        np.testing.assert_allclose(tt[:, 0], r["ll_x"], rtol=1e-5, atol=1e-4, err_msg=kern + " ll_x " + tag)
        np.testing.assert_allclose(tt[~near_tie, 2], r["ll_y"][~near_tie], rtol=1e-5, atol=1e-4, err_msg=kern + " ll_y " + tag)
        for c, k in ((1, "ll_x2y"), (3, "ll_y2x")):   # 3 x the model of tests/isco_tol.py (fitted at N <= 2000; here N up to 3000)
            okm = mass >= 1e-6
            err = np.abs(tt[:, c] - r[k])
            tolv = 3 * ll_atol(mass, plh) + RTOL * np.abs(r[k])
            if not (err[okm] <= tolv[okm]).all():
                bad = int(np.flatnonzero(okm & (err > tolv))[0])
                raise AssertionError(f"{kern} {k} {tag} env {bad}: got {tt[bad, c]} oracle {r[k][bad]} other kernel "
                                     f"{(ts if kern == 'wave' else tb)[bad, c].item()} tol {tolv[bad]} mass {mass[bad]} L {plh[bad]} "
                                     f"selected {int(r['mask'][bad].sum())}")
        # log_acc = ll_y + ll_y2x - ll_x - ll_x2y in float32: it also carries the rounding of its (large) terms
        big_terms = 4e-7 * (np.abs(r["ll_x"]) + np.abs(r["ll_y"]) + np.abs(r["ll_x2y"]) + np.abs(r["ll_y2x"]))
        ok = mass >= 1e-6
        assert (np.abs(tt[:, 4] - r["log_acc"])[ok] <= (3 * ll_atol(mass, plh) + big_terms + RTOL * np.abs(r["log_acc"]))[ok]).all(), f"{kern} log_acc {tag}"
    sure = torch.from_numpy((mass >= 1e-6) & (r["accept_margin"] > 2 * (3 * ll_atol(mass, plh) + big_terms + RTOL * np.abs(r["log_acc"])))).to(DEV)
    assert torch.equal(yb[:Bs][sure], ys[sure]), "accepted samples " + tag
    assert np.array_equal(ys[sure].cpu().numpy(), r["y"][sure.cpu().numpy()].astype(np.float32)), "accepted samples vs oracle " + tag
    prop = torch.where(mb.bool(), 1 - x, x)
    changed = (yb != x).any(1)
    assert torch.equal(yb[changed], prop[changed]), "an accepted sample is the proposal " + tag
    it += 1
print(f"fuzz_isco: {it} random configurations, no mismatch")

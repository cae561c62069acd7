"""Differential fuzz of the spin-system env, per-env (dense) couplings and one shared graph (CSR), against the numpy restatement of the reference's batched PECO env
(oracle/oracle_spin.py, one instance per env on that env's own matrix): random sizes, densities, +-1 couplings with and
without diagonal entries, reward modes, visited-state memory, revisits; and the single-instance float64 surface with its options
(PASS, finite memory, CUT / ENERGY, reversible / irreversible spins, both bases, shared graph or generator) against the float64
restatement of the numpy env.  `python tools/fuzz/fuzz_spin.py [seconds] [seed]`."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle.oracle_spin import SpinSystemOracle, SpinSystemOracleF64
from rlsolver_amd.envs.spinsystem import (ECO_PECO_OBSERVABLES, ExtraAction, OptimisationTarget, RewardSignal, SpinBasis, SpinSystem,
                                          SpinSystemUnbiased)
from rlsolver_amd.envs.util_envs_PECO import SetGraphGenerator

DEV = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
it = singles = 0
while time.time() < t_end:
    n = int(rng.choice([rng.randint(3, 20), rng.randint(20, 70), rng.randint(64, 200)]))
    B = int(rng.randint(1, 10))
    T = int(rng.randint(3, 50))
    dens = rng.uniform(0.05, 0.9)
    W = np.zeros((B, n, n), np.float32)
    for b in range(B):
        while True:
            up = np.triu((rng.rand(n, n) < dens) * rng.choice([-1, 1], size=(n, n)), 1).astype(np.float32)
            m = up + up.T
            if rng.rand() < 0.4:                                               # self-loops, as the reference's BA seed clique
                k = int(rng.randint(1, min(n, 6) + 1))
                m[np.arange(k), np.arange(k)] = rng.choice([-1, 1], size=k)
            rs = m.sum(1)
            if np.abs(rs).sum() != 0 and rs.max() != 0:
                break
        W[b] = m
    mode = rng.choice(["DENSE", "BLS", "CUSTOM_BLS"])
    norm = bool(rng.rand() < 0.5)
    basin = None if rng.rand() < 0.4 else float(rng.choice([0.25, 1.0 / n, 0.5]))
    stag = None if rng.rand() < 0.5 else float(rng.choice([0.125, 0.25]))
    tag = f"it={it} n={n} B={B} T={T} mode={mode} norm={norm} basin={basin} stag={stag}"
    shared = bool(rng.rand() < 0.4)
    if shared:                                                                  # one graph for all envs: the CSR form of the step
        W[:] = W[0]
        W[:, np.arange(n), np.arange(n)] = 0
        rs = W[0].sum(1)
        if np.abs(rs).sum() == 0 or rs.max() == 0:
            continue
        mg = [(i, j, int(W[0, i, j])) for i in range(n) for j in range(i + 1, n) if W[0, i, j] != 0]
        env = SpinSystem(mg, n, B, max_steps=T, observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal[mode], norm_rewards=norm,
                         spin_basis=SpinBasis.BINARY, basin_reward=basin, stag_punishment=stag, device=DEV)
    else:
        env = SpinSystem(None, None, B, max_steps=T, observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal[mode], norm_rewards=norm,
                         spin_basis=SpinBasis.BINARY, basin_reward=basin, stag_punishment=stag, device=DEV,
                         graph_generator=SetGraphGenerator(W, device=DEV))
    tag += f" shared={shared}"
    oras = [SpinSystemOracle(W[b], 1, T, reward=mode, norm_rewards=norm, basin_reward=basin, stag_punishment=stag) for b in range(B)]
    s0 = (2 * rng.randint(0, 2, size=(B, n)) - 1).astype(np.float32)
    obs = env.reset(torch.from_numpy(s0))
    want = np.concatenate([o.reset(s0[b:b + 1]) for b, o in enumerate(oras)])
    assert np.array_equal(obs[:, :7].cpu().numpy(), want), "reset " + tag
    assert np.array_equal(obs[:, 7:].cpu().numpy(), W), "matrix rows " + tag
    acts = rng.randint(0, n, size=(T, B))
    for t in range(2, T, 3):
        acts[t] = acts[t - 1]
    for t in range(T):
        o, r, d = env.step(torch.from_numpy(acts[t]).to(DEV))
        res = [ora.step(acts[t, b:b + 1]) for b, ora in enumerate(oras)]
        assert np.array_equal(o[:, :7].cpu().numpy(), np.concatenate([x[0] for x in res])), f"obs t={t} " + tag
        assert np.array_equal(r.cpu().numpy(), np.concatenate([x[1] for x in res])), f"reward t={t} " + tag
        assert np.array_equal(env.score.cpu().numpy(), np.concatenate([ora.score for ora in oras])), f"score t={t} " + tag
        if t % 7 == 3:     # whoever reads env.state gets the rows a step does not store (rls_spin_materialize)
            st = env.state.cpu().numpy().copy()
            st[:, 0] = (1 - st[:, 0]) / 2
            assert np.array_equal(st, np.concatenate([x[0] for x in res])), f"state t={t} " + tag
    it += 1
    # the single-instance float64 surface with the reference's options: ExtraAction.PASS and / or a finite memory, CUT or ENERGY,
    # reversible or irreversible spins, either spin basis, a shared graph or a generator (per-env couplings, B = 1); the maximum
    # local reward follows the numpy env's rule (over the NONZERO entries: graphs whose row sums are all <= 0 are kept)
    if rng.rand() < 0.5:
        Wd = W[0].astype(np.float64).copy()
        Wd[np.arange(n), np.arange(n)] = 0
        if rng.rand() < 0.25:                                                     # an isolated node or two
            for i in rng.choice(n, size=min(n - 2, int(rng.randint(1, 3))), replace=False):
                Wd[i, :] = 0
                Wd[:, i] = 0
        if rng.rand() < 0.2:
            Wd = np.abs(Wd)                                                       # a positive graph: under ENERGY every row sum is <= 0
        if not np.any(Wd.sum(1) != 0):
            continue
        mg = [(i, j, int(Wd[i, j])) for i in range(n) for j in range(i + 1, n) if Wd[i, j] != 0]
        ep = bool(rng.rand() < 0.6)
        M = None if rng.rand() < 0.4 else int(rng.randint(2, 9))
        target = str(rng.choice(["CUT", "ENERGY"]))
        rev = bool(rng.rand() < 0.6)
        binary = bool(rng.rand() < 0.5)
        gen = bool(rng.rand() < 0.4)
        tag1 = (f"single: it={it} n={n} T={T} mode={mode} norm={norm} basin={basin} stag={stag} pass={ep} memory={M} target={target} "
                f"reversible={rev} binary={binary} generator={gen}")
        kw = dict(max_steps=T, observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal[mode], norm_rewards=norm,
                  spin_basis=SpinBasis.BINARY if binary else SpinBasis.SIGNED, basin_reward=basin, stag_punishment=stag, device=DEV,
                  extra_action=ExtraAction.PASS if ep else ExtraAction.NONE, memory_length=M, reversible_spins=rev,
                  optimisation_target=OptimisationTarget[target])
        if gen:
            class Fixed:
                n_spins, biased = n, False

                def get(self, with_padding=False):
                    return Wd.copy()
            e1 = SpinSystemUnbiased(None, None, graph_generator=Fixed(), **kw)
        else:
            e1 = SpinSystemUnbiased(mg, n, **kw)
        o1 = SpinSystemOracleF64(Wd, T, reward=mode, norm_rewards=norm, basin_reward=basin, stag_punishment=stag, extra_pass=ep, memory_length=M,
                                 target=target, reversible=rev, binary=binary)
        s1 = (2 * rng.randint(0, 2, size=n) - 1).astype(np.float64) if rev else None
        assert np.array_equal(e1.reset(s1), o1.reset(s1)), "reset " + tag1
        assert e1.max_local_reward_available == o1.max_local and e1.score == o1.score, "max local reward / score " + tag1
        assert np.array_equal(e1.get_immeditate_rewards_avaialable(), o1.gains()), "immediate rewards " + tag1
        prev = 0
        for t in range(T):
            a = int(rng.randint(0, n))
            a = prev if t % 3 == 2 else (n if (ep and rng.rand() < 0.2) else a)
            prev = a
            go, gr, gd, _ = e1.step(a)
            wo, wr, wd = o1.step(a)
            assert np.array_equal(go, wo) and gr == wr and gd == wd, f"t={t} a={a} " + tag1
            assert e1.best_obs_score == o1.best_obs_score and e1.score == o1.score and e1.best_score == o1.best_score, f"scores t={t} " + tag1
            if t % 5 == 1:
                assert np.array_equal(e1.state, o1.state), f"state t={t} " + tag1
            if gd:
                break
        singles += 1
print(f"fuzz_spin: {it} random configurations ({singles} also on the single-instance surface with its options), no mismatch")

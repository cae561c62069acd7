"""Statistical tests of the PRODUCTION draws ("distributional tests cover the production mode").

Parity is pinned with the reference's recorded draws supplied (the test hooks); in production every kernel draws from its own
counter-based generator, and "distributionally equivalent" needs evidence of its own.  Every generator on the path is
held here to moments, tails, uniformity (chi-square) and lag correlations across the three axes a key is built from
(node / round / env), from fixed seeds -- the outcomes are deterministic, the thresholds are ~5-6 sigma of the ideal
generator's sampling noise:

* the local search's normals (``normal4``: two 32-bit murmur finalisers per node quad, 16-bit radius x 16-bit angle
  Box-Muller, tail cut at 4.7 sigma), read through ``rls_maxcut_ls_normals`` -- the values the fused / round kernels use;
* ``rls_rand_actions`` and ``rls_rand_spins`` (Philox-4x32-10);
* the metro walk's node picks and Metropolis uniforms (K9), observed through walks whose outcome IS the draw;
* the tie coins of the level-parallel sampler (K7), observed on isolated nodes, where the new bit IS the coin;
* the ISCO Gumbel perturbation, observed at infinite temperature, where the selected set is a uniform random subset."""
import math

import numpy as np
import pytest
import torch

from gpu_util import DEV, gnm_arr
from rlsolver_amd import ops, ops_mcpg_tsp as mops

pytestmark = pytest.mark.gpu


def _corr(a, b):
    a = a.double().flatten() - a.double().mean()
    b = b.double().flatten() - b.double().mean()
    return float((a * b).mean() / (a.std(unbiased=False) * b.std(unbiased=False)))


def _chi2_uniform(counts):
    c = counts.double()
    e = c.sum() / c.numel()
    return float(((c - e) ** 2 / e).sum()), c.numel() - 1


def _chi2_ok(chi2, df, sigmas=6.0):
    return abs(chi2 - df) < sigmas * math.sqrt(2 * df)


def test_local_search_normals_moments_tails_and_uniformity():
    B, N, D = 2048, 2000, 4
    z = torch.stack([ops.maxcut_ls_normals(B, N, seed=0x5EED1234ABCD, draw=d, device=DEV, env_offset=7 * B) for d in range(D)])
    n = z.numel()
    zd = z.double()
    m1, m2 = float(zd.mean()), float((zd ** 2).mean())
    m3, m4 = float((zd ** 3).mean()), float((zd ** 4).mean())
    assert abs(m1) < 5 / math.sqrt(n), m1
    assert abs(m2 - 1) < 5 * math.sqrt(2 / n), m2
    assert abs(m3) < 5 * math.sqrt(15 / n), m3
    # the tail ends at sqrt(2 ln 2^16) = 4.71 sigma: E z^4 falls short of 3 by ~1e-3 (P(|z| > 4.71) z^4), nothing like a percent
    assert abs(m4 - 3) < 5 * math.sqrt(96 / n) + 2e-3, m4
    assert float(zd.abs().max()) <= math.sqrt(2 * math.log(65536.0)) + 1e-3
    for thr, p in ((1.0, 0.31731050786), (2.0, 0.04550026390), (3.0, 0.00269979606), (4.0, 6.3342484e-5)):
        k = float((zd.abs() > thr).sum())
        assert abs(k - n * p) < 5 * math.sqrt(n * p) + 2, (thr, k, n * p)
    # probability integral transform into 256 equiprobable bins
    u = 0.5 * (1 + torch.erf(zd / math.sqrt(2)))
    chi2, df = _chi2_uniform(torch.bincount((u * 256).long().clamp_(0, 255).flatten(), minlength=256))
    assert _chi2_ok(chi2, df), (chi2, df)
    # sign bits and the first mantissa-level structure: each of the 4 lanes of a quad by itself
    for k in range(4):
        zk = zd[..., k::4]
        assert abs(float(zk.mean())) < 5 / math.sqrt(zk.numel()) and abs(float((zk ** 2).mean()) - 1) < 5 * math.sqrt(2 / zk.numel())


def test_local_search_normals_are_uncorrelated_along_every_key_axis():
    B, N, D = 1024, 2000, 6
    z = torch.stack([ops.maxcut_ls_normals(B, N, seed=99, draw=d, device=DEV) for d in range(D)])
    n = z.numel()
    lim = 5.5 / math.sqrt(n / 2)
    assert abs(_corr(z[..., :-1], z[..., 1:])) < lim                     # adjacent nodes (within and across quads)
    assert abs(_corr(z[..., 0::4], z[..., 1::4])) < 2 * lim              # the cos / sin pair of one hash
    assert abs(_corr(z[..., 0::4], z[..., 2::4])) < 2 * lim              # the two hashes of a quad
    assert abs(_corr(z[..., :-4], z[..., 4:])) < lim                     # neighbouring quads
    assert abs(_corr(z[:-1], z[1:])) < lim                               # consecutive rounds
    assert abs(_corr(z[:, :-1], z[:, 1:])) < lim                         # consecutive envs
    sq = z.double() ** 2
    assert abs(_corr(sq[..., 0::4], sq[..., 1::4])) < 3 * lim            # squares of a Box-Muller pair: independent normals
    assert abs(_corr(sq[..., 0::4], sq[..., 2::4])) < 3 * lim
    # different seeds, different env offsets: unrelated streams; same (seed, env, node, draw): the same value
    a = ops.maxcut_ls_normals(256, N, seed=1, draw=0, device=DEV)
    assert abs(_corr(a, ops.maxcut_ls_normals(256, N, seed=2, draw=0, device=DEV))) < 5.5 / math.sqrt(a.numel())
    assert abs(_corr(a, ops.maxcut_ls_normals(256, N, seed=1, draw=0, device=DEV, env_offset=256))) < 5.5 / math.sqrt(a.numel())
    assert torch.equal(ops.maxcut_ls_normals(128, N, seed=1, draw=0, device=DEV, env_offset=128), a[128:])


def test_local_search_normals_quads_do_not_repeat():
    """Round 3 hashed both halves of a quad from ONE 32-bit key: of 2 * 10^6 quads ~500 pairs shared all four normals (n^2 / 2^33),
    and two envs whose 32-bit keys collide shared a whole call.  With independent keys per half a full repeat needs a 64-bit
    coincidence; halves still coincide at the rate any 32-bit draw does."""
    B, N = 4096, 2000
    z = ops.maxcut_ls_normals(B, N, seed=4242, draw=3, device=DEV).view(-1, 4)
    bits = z.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    full = (bits[:, 0] * 0x1E3779B97F4A7C15 + bits[:, 1]) ^ ((bits[:, 2] * 0x42B2AE3D27D4EB4F) + bits[:, 3] * 31)   # (int64 products wrap)
    # exact check on candidate duplicates of the mixed word
    srt, order = torch.sort(full)
    cand = (srt[1:] == srt[:-1]).nonzero().flatten()
    dup = sum(1 for i in cand.tolist() if torch.equal(bits[order[i]], bits[order[i + 1]]))
    assert dup == 0, f"{dup} quads repeat all four normals"
    half = torch.sort(bits[:, 0] << 32 | bits[:, 1])[0]
    nh = int((half[1:] == half[:-1]).sum())
    expect = z.shape[0] ** 2 / 2 / 2 ** 32
    assert 0.6 * expect < nh < 1.5 * expect, (nh, expect)                # what 32-bit draws of an ideal generator do


def test_rand_actions_uniform_and_uncorrelated():
    B, N, S = 1 << 16, 2000, 16
    a = torch.stack([ops.rand_actions(B, N, seed=31337, step=s, device=DEV, env_offset=3 * B) for s in range(S)])
    assert int(a.min()) >= 0 and int(a.max()) < N
    chi2, df = _chi2_uniform(torch.bincount(a.flatten(), minlength=N))
    assert _chi2_ok(chi2, df), (chi2, df)
    lim = 5.5 / math.sqrt(a.numel())
    assert abs(_corr(a[:, :-1], a[:, 1:])) < lim and abs(_corr(a[:-1], a[1:])) < lim
    # pairs (a_t, a_{t+1}) mod 16: 256 cells
    chi2, df = _chi2_uniform(torch.bincount(((a[:-1] % 16) * 16 + a[1:] % 16).flatten(), minlength=256))
    assert _chi2_ok(chi2, df), (chi2, df)


def test_rand_spins_fair_and_uncorrelated():
    B, N = 8192, 2000
    x = ops.rand_spins(B, N, seed=77, device=DEV, env_offset=12345).float()
    assert float(x[:, 0].sum()) == 0                                     # node 0 := 0 (env_L2A.py:84)
    x = x[:, 1:]
    n = x.numel()
    assert abs(float(x.mean()) - 0.5) < 5 * 0.5 / math.sqrt(n)
    lim = 5.5 / math.sqrt(n)
    assert abs(_corr(x[:, :-1], x[:, 1:])) < lim and abs(_corr(x[:-1], x[1:])) < lim
    assert abs(_corr(x[:, :-32], x[:, 32:])) < lim and abs(_corr(x[:, :-128], x[:, 128:])) < lim   # word / Philox-call strides
    col = x.double().mean(dim=0)
    assert float(((col - 0.5) ** 2).mean()) < 2.0 * 0.25 / B              # per-node bias: none beyond sampling noise


def _metro_one_round(N, C, p, t, seed):
    start = mops.PackedChains(torch.zeros((C // 64, N), dtype=torch.int64, device=DEV), C)
    out = mops.PackedChains.empty(N, C, DEV)
    probs = torch.full((N,), p, dtype=torch.float32, device=DEV)
    mops.mcpg_metro_rounds(out, probs, 1, seed=seed, t_offset=t, samples_in=start)
    return out.unpack()                                                  # f32 [N, C]: a 1 where the round flipped a node


def test_metro_walk_node_picks_and_uniforms():
    N, C, T = 1000, 1 << 15, 24
    idx = []
    for t in range(T):
        xs = _metro_one_round(N, C, 0.5, t, seed=2024)                   # p = 1/2: acceptance rate (1 - q) / q = 1, every pick flips
        assert bool((xs.sum(dim=0) == 1).all())
        idx.append(xs.argmax(dim=0))
    idx = torch.stack(idx)                                               # [T, C] the node pick of (round, chain)
    chi2, df = _chi2_uniform(torch.bincount(idx.flatten(), minlength=N))
    assert _chi2_ok(chi2, df), (chi2, df)
    lim = 5.5 / math.sqrt(idx.numel())
    assert abs(_corr(idx[:, :-1], idx[:, 1:])) < lim and abs(_corr(idx[:-1], idx[1:])) < lim
    # the Metropolis uniform: from the all-zero state the acceptance rate is p / (1 - p) =: r, so P(flip) = P(u < r)
    for r in (0.1, 0.25, 0.5, 0.75, 0.9):
        acc = torch.stack([_metro_one_round(N, C, r / (1 + r), t, seed=555).sum(dim=0) for t in range(8)])
        n = acc.numel()
        assert abs(float(acc.sum()) - n * r) < 5 * math.sqrt(n * r * (1 - r)), (r, float(acc.mean()))
        if r == 0.5:
            lim = 5.5 / math.sqrt(n)
            assert abs(_corr(acc[:, :-1], acc[:, 1:])) < lim and abs(_corr(acc[:-1], acc[1:])) < lim
            picks = torch.stack([_metro_one_round(N, C, 0.5, t, seed=555).argmax(dim=0) for t in range(8)])
            assert abs(_corr(acc, picks % 2)) < lim                      # the uniform is independent of the pick made with it


def test_level_sampler_tie_coins_are_fair():
    """On an isolated node the level-parallel sampler's decision is the tie coin itself: count = 0 = K, degree even."""
    from rlsolver_amd.methods import MCPG as amcpg
    N, C = 512, 8192
    data = amcpg.make_data(N, np.array([0]), np.array([1]), DEV)
    zeros = mops.PackedChains(torch.zeros((C // 64, N), dtype=torch.int64, device=DEV), C)

    def coins(num_ls, seed):
        out, _ = mops.mcpg_local_search_levels(data.graph, zeros, data._lv_ptr, data._lv_data, num_ls, seed)
        return out[2:]                                                   # f32 [N - 2, C]: the isolated nodes
    c1 = coins(1, 808)
    n = c1.numel()
    assert abs(float(c1.mean()) - 0.5) < 5 * 0.5 / math.sqrt(n)
    lim = 5.5 / math.sqrt(n)
    assert abs(_corr(c1[:, :-1], c1[:, 1:])) < lim                       # neighbouring chains: bits of one coin word
    assert abs(_corr(c1[:, :-64], c1[:, 64:])) < lim                     # neighbouring tiles
    assert abs(_corr(c1[:-1], c1[1:])) < lim                             # neighbouring visiting positions
    c2 = coins(2, 808)                                                   # the last pass decides: pass 1's coins
    assert abs(_corr(c1, c2)) < lim and abs(_corr(c1, coins(1, 809))) < lim
    word_pop = c1.view(N - 2, C // 64, 64).sum(dim=2)                    # popcount of each 64-coin word ~ Binomial(64, 1/2)
    assert abs(float(word_pop.var()) - 16.0) < 5 * 16.0 * math.sqrt(2 / word_pop.numel())


def test_isco_gumbel_selection_is_uniform_at_infinite_temperature():
    from rlsolver_amd.envs.env_ISCO_maxcut import ISCO_maxcut
    n, B, L = 200, 8192, 10
    g = gnm_arr(n, 800, 6)
    pd = {"edge_from": torch.from_numpy(g[:, 0].copy()).to(DEV), "edge_to": torch.from_numpy(g[:, 1].copy()).to(DEV),
          "num_nodes": n, "num_edges": g.shape[0]}
    s = ISCO_maxcut(pd, batch_size=B, device=DEV, seed=17)
    x = s.random_gen_init_sample()
    _, _, acc, terms, mask = s.step(x, L, 1e12, want_terms=True)          # all scores equal: the top-L of the Gumbel draws alone
    assert bool((mask.sum(dim=1) == L).all())
    counts = mask.sum(dim=0).double()
    e = B * L / n
    chi2 = float(((counts - e) ** 2 / (e * (1 - L / n))).sum())          # (fixed row sums: hypergeometric variance)
    assert _chi2_ok(chi2, n - 1), chi2
    m = mask.float()
    lim = 5.5 / math.sqrt(m.numel())
    assert abs(_corr(m[:-1], m[1:])) < lim                               # consecutive samples select independently
    assert bool((acc > 0.999).all())                                     # and every proposal is accepted at T = inf

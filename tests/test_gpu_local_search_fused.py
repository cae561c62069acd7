"""The fused local-search kernel (rls_maxcut_local_search): bit-exact against the reference's
golden traces in test mode (noise supplied), equal to the decomposed K2/K6/K5 path, and sane in
production mode (in-kernel Philox + Box-Muller noise)."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as onp
from tests.gpu_util import DEV, gnm_arr, to_dev_bool

pytestmark = pytest.mark.gpu


def mygraph_of(arr):
    return [tuple(int(v) for v in r) for r in arr]


@pytest.mark.parametrize("gname", ["PL_20_ID0", "BA_100_ID0", "ER_100_ID0", "gset_14_stub"])
@pytest.mark.parametrize("bidir", [False, True])
@pytest.mark.parametrize("fused", [True, False])
def test_local_search_inplace_golden_both_paths(golden, gname, bidir, fused):
    from rlsolver_amd import ops
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    z = golden("maxcut_local_search")
    env = EnvMaxcut(mygraph=mygraph_of(z[f"{gname}/graph"]), device=DEV, if_bidirectional=bidir)
    env.fused_local_search = fused
    tag = f"{gname}/bidir{int(bidir)}"
    num_spin = int(z[f"{tag}/ls/num_spin"])
    # (rows that are not dword-aligned -- the 14-node stub -- have no fused form: both legs take the decomposed path there)
    assert ops.local_search_fusable(env.graph, num_spin) == (env.num_nodes % 4 == 0)
    xs = to_dev_bool(z[f"{tag}/ls/xs_in"]).clone()
    noise = torch.from_numpy(z[f"{tag}/ls/noise"]).to(DEV)
    gx, gv = env.local_search_inplace(xs, torch.empty(()), num_iters=8, num_spin=num_spin, noise_std=0.3, noise=noise)
    assert np.array_equal(gx.cpu().numpy().astype(np.uint8), z[f"{tag}/ls/xs_out"])
    assert np.array_equal(gv.cpu().numpy(), z[f"{tag}/ls/vs_out"])
    # second call on the result with an explicit good_vs (the in/out objective path)
    xs2 = to_dev_bool(z[f"{tag}/ls/xs_in"]).clone()
    vs2 = env.calculate_obj_values(xs2)
    gx2, gv2 = env.local_search_inplace(xs2, vs2, num_iters=8, num_spin=num_spin, noise_std=0.3, noise=noise)
    assert gv2.data_ptr() == vs2.data_ptr() and torch.equal(gx2, gx) and torch.equal(gv2, gv)


@pytest.mark.parametrize("n,m,B,bidir", [(2000, 19990, 130, False), (333, 2000, 70, True), (64, 400, 64, False), (1000, 5000, 65, False),
                                         (3004, 9000, 64, True)])
def test_fused_equals_decomposed_random(n, m, B, bidir):
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    garr = gnm_arr(n, m, seed=n)
    env = EnvMaxcut(mygraph=mygraph_of(garr), device=DEV, if_bidirectional=bidir, num_nodes=n)
    g = torch.Generator(device="cpu").manual_seed(1)
    xs0 = torch.randint(0, 2, (B, n), generator=g, dtype=torch.bool).to(DEV)
    noise = torch.randn((6, B, n), generator=g).to(DEV)
    outs = []
    for fused in (True, False):
        env.fused_local_search = fused
        xs = xs0.clone()
        gx, gv = env.local_search_inplace(xs, torch.empty(()), num_iters=5, num_spin=6, noise_std=0.3, noise=noise)
        outs.append((gx.clone(), gv.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][1].cpu().numpy(), onp.maxcut_obj(outs[0][0].cpu().numpy(), garr, bidir))
    if n <= 64:   # and against the reference-shaped oracle
        wx, wv = onp.local_search_inplace(xs0.cpu().numpy(), garr, n, bidir, noise.cpu().numpy(), num_iters=5, num_spin=6)
        assert np.array_equal(outs[0][0].cpu().numpy(), wx) and np.array_equal(outs[0][1].cpu().numpy(), wv)


def test_fused_production_noise_is_sane():
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    n, m, B = 800, 4694, 512
    garr = gnm_arr(n, m, seed=14)
    env = EnvMaxcut(mygraph=mygraph_of(garr), device=DEV, num_nodes=n)
    torch.manual_seed(0)
    xs = env.generate_xs_randomly(B)
    v0 = env.calculate_obj_values(xs)
    vs = v0.clone()
    for _ in range(3):
        prev = vs.clone()
        env.local_search_inplace(xs, vs, num_iters=8, num_spin=8, noise_std=0.3)
        assert (vs >= prev).all()
    assert torch.equal(env.calculate_obj_values(xs), vs)
    assert float(vs.float().mean()) > 0.62 * m > float(v0.float().mean())
    # reproducible under torch.manual_seed, different otherwise
    res = []
    for seed in (5, 5, 6):
        torch.manual_seed(seed)
        x = env.generate_xs_randomly(64)
        v = env.calculate_obj_values(x)
        env.local_search_inplace(x, v, num_iters=8, num_spin=8)
        res.append(x.clone())
    assert torch.equal(res[0], res[1]) and not torch.equal(res[0], res[2])


@pytest.mark.parametrize("gname", ["BA_100_ID0", "PL_20_ID0"])
@pytest.mark.parametrize("fused", [True, False])
def test_local_search_class_golden_both_paths(golden, gname, fused):
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    from rlsolver_amd.methods.LocalSearch import LocalSearch
    z = golden("local_search_class")
    env = EnvMaxcut(mygraph=mygraph_of(z[f"{gname}/graph"]), device=DEV, if_bidirectional=False)
    env.fused_local_search = fused
    tag = f"{gname}/bidir0"
    ls = LocalSearch(simulator=env, num_nodes=env.num_nodes)
    ls.reset(to_dev_bool(z[f"{tag}/xs_in"]).clone())
    for r in range(2):
        gx, gv, nupd = ls.random_search(num_iters=4, num_spin=4, noise_std=0.3,
                                        noise=torch.from_numpy(z[f"{tag}/round{r}/noise"]).to(DEV))
        assert np.array_equal(gx.cpu().numpy().astype(np.uint8), z[f"{tag}/round{r}/xs"])
        assert np.array_equal(gv.cpu().numpy(), z[f"{tag}/round{r}/vs"])


@pytest.mark.parametrize("n,m,B,bidir,num_spin", [(2000, 19990, 130, False, 8), (320, 2000, 70, True, 6), (64, 400, 64, False, 3),
                                                  (1008, 5000, 65, False, 12), (3008, 9000, 200, True, 8),
                                                  (320, 2000, 16500, False, 8),       # (enough tiles for one workgroup each)
                                                  (333, 2000, 70, False, 5), (1001, 5000, 130, True, 8), (2004, 9000, 65, False, 8),
                                                  (1999, 9000, 200, False, 8)])       # rows that are not 16-byte multiples
def test_round_kernels_equal_the_fused_kernel_for_the_same_seed(n, m, B, bidir, num_spin):
    """rls_maxcut_ls_threshold + rls_maxcut_ls_propose per round + K5 draw what the fused kernel draws: under the same torch
    seed local_search_inplace gives the same spins and cuts through either form (the fused one is pinned by the golden
    fixtures with recorded draws and checked for sanity with its own); also LocalSearch.random_search's form, where the
    first draw already proposes."""
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    garr = gnm_arr(n, m, seed=n + 1)
    env = EnvMaxcut(mygraph=mygraph_of(garr), device=DEV, if_bidirectional=bidir, num_nodes=n)
    g = torch.Generator(device="cpu").manual_seed(2)
    xs0 = torch.randint(0, 2, (B, n), generator=g, dtype=torch.bool).to(DEV)
    for first in (False, True):
        outs = []
        for rounds in (False, True):
            env.force_ls_rounds = rounds
            env.force_ls_fused = not rounds
            torch.manual_seed(77)
            xs, vs = xs0.clone(), env.calculate_obj_values(xs0)
            env.local_search_pipeline(xs, vs, weight_mult=2 if first else 1, num_iters=5, num_spin=num_spin, noise_std=0.3, noise=None,
                                      first_draw_proposes=first)
            outs.append((xs, vs))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (first,)
        assert np.array_equal(outs[1][1].cpu().numpy(), onp.maxcut_obj(outs[1][0].cpu().numpy(), garr, bidir))
        assert not torch.equal(outs[1][0], xs0)


@pytest.mark.parametrize("n,m", [(10000, 49975), (12000, 30000), (9000, 27000), (9001, 27000)])
def test_round_kernels_beyond_the_fused_kernel(n, m):
    """Graphs the fused kernel does not fit (N = 10^4: rd_std in LDS; N = 12 000: rd_std read from global memory; N = 9000 -- a Gset
    size -- and 9001: rows that are not 16-byte multiples, read on a padded pitch / through the funnel-shift stage): every round
    keeps cuts non-decreasing and consistent, touches only rows it accepts, and proposes about num_spin flips per env."""
    from rlsolver_amd import ops
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    B, num_spin = 192, 8
    garr = gnm_arr(n, m, seed=5)
    env = EnvMaxcut(mygraph=mygraph_of(garr), device=DEV, num_nodes=n)
    assert not ops.local_search_fusable(env.graph, num_spin, B) and ops.ls_rounds_supported(env.graph, num_spin)
    torch.manual_seed(3)
    xs = env.generate_xs_randomly(B)
    vs = env.calculate_obj_values(xs)
    ws, span = ops.maxcut_ls_weights(env.graph, xs, 1, padded=True)
    pitch_b = ws.shape[1] * ws.element_size()       # rows a whole number of ops.LS_PITCH_BYTES apart (cache lines; >= the 16 bytes the kernels need)
    assert pitch_b % ops.LS_PITCH_BYTES == 0 and ops.LS_PITCH_BYTES % 16 == 0 and 0 <= pitch_b - n * ws.element_size() < ops.LS_PITCH_BYTES
    ws_flat, span_flat = ops.maxcut_ls_weights(env.graph, xs, 1)
    assert torch.equal(ws[:, :n], ws_flat) and torch.equal(span, span_flat)
    rd_std = (span.float() * 0.3).contiguous()
    thresh = ops.maxcut_ls_threshold(env.graph, ws, rd_std, seed=99, num_spin=num_spin)
    assert thresh.shape == (B,) and bool(torch.isfinite(thresh).all())
    assert bool((thresh > ws_flat.float().mean(dim=1)).all())       # the 9th largest of 10^4 noisy weights sits in the upper tail
    flips = []
    for it in range(1, 5):
        x0, v0 = xs.clone(), vs.clone()
        ops.maxcut_ls_propose(env.graph, xs, ws, rd_std, thresh, vs, seed=99, draw=it)
        changed = (xs != x0).any(dim=1)
        assert bool((vs >= v0).all()) and torch.equal(env.calculate_obj_values(xs), vs)
        assert bool((vs[~changed] == v0[~changed]).all())
        nf = (xs != x0).sum(dim=1)[changed].float()
        if nf.numel():
            flips.append(float(nf.mean()))
    assert flips and 0.25 * num_spin < np.mean(flips) < 4 * num_spin, flips
    # with the scratch buffer a small batch's noise passes are split over several workgroups: same threshold, same round
    scratch = ops.ls_scratch(env.graph, B, ws)
    assert scratch is not None and scratch.numel() >= (B + 63) // 64 * n * 8
    assert torch.equal(ops.maxcut_ls_threshold(env.graph, ws, rd_std, seed=99, num_spin=num_spin, scratch=scratch), thresh)
    p, q, vp, vq = xs.clone(), xs.clone(), vs.clone(), vs.clone()
    ops.maxcut_ls_propose(env.graph, p, ws, rd_std, thresh, vp, seed=99, draw=5)
    ops.maxcut_ls_propose(env.graph, q, ws, rd_std, thresh, vq, seed=99, draw=5, scratch=scratch)
    assert torch.equal(p, q) and torch.equal(vp, vq)
    assert ops.ls_scratch(env.graph, 1 << 16, ws) is None           # a batch that fills the chip by itself
    # the same seed and draw index give the same proposal; another draw index another one
    a, b, c = xs.clone(), xs.clone(), xs.clone()
    va, vb, vc = vs.clone(), vs.clone(), vs.clone()
    ops.maxcut_ls_propose(env.graph, a, ws, rd_std, thresh, va, seed=99, draw=7)
    ops.maxcut_ls_propose(env.graph, b, ws, rd_std, thresh, vb, seed=99, draw=7)
    ops.maxcut_ls_propose(env.graph, c, ws, rd_std, thresh, vc, seed=99, draw=8)
    assert torch.equal(a, b) and torch.equal(va, vb)
    # and the whole call through the env
    v1 = vs.clone()
    env.local_search_inplace(xs, vs, num_iters=3, num_spin=num_spin)
    assert bool((vs >= v1).all()) and torch.equal(env.calculate_obj_values(xs), vs)


@pytest.mark.parametrize("n", [20000, 20008])
def test_round_kernels_at_g81_size_where_the_tile_fills_lds(n):
    """N = 20 000 (Gset's G81): the 64-env tile takes 160 000 of LDS's 163 840 bytes, so the proposal rounds run as mask kernels
    + an apply kernel through the scratch buffer (which is required here) -- on half tiles (32 envs: rows of 16-byte multiples,
    N = 20 000) or on the bare 64-env tile with 4 waves (other rows, N = 20 008) --, the threshold as usual; same invariants, and
    all rounds at once == one round per call for the same seed."""
    from rlsolver_amd import ops
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    m, B, num_spin = 40000, 130, 8
    garr = gnm_arr(n, m, seed=81)
    env = EnvMaxcut(mygraph=mygraph_of(garr), device=DEV, num_nodes=n)
    assert not ops.local_search_fusable(env.graph, num_spin, B) and ops.ls_rounds_supported(env.graph, num_spin)
    torch.manual_seed(4)
    xs = env.generate_xs_randomly(B)
    vs = env.calculate_obj_values(xs)
    assert np.array_equal(vs.cpu().numpy(), onp.maxcut_obj(xs.cpu().numpy(), garr, False))              # K1 on the 4-wave tile form
    ws, span = ops.maxcut_ls_weights(env.graph, xs, 1, padded=True)
    c = ops.maxcut_node_cutdeg(env.graph, xs)
    deg = torch.from_numpy(np.bincount(env.graph.csr.eu, minlength=n)).to(DEV)
    assert torch.equal(ws[:, :n].long(), deg[None, :] - c)                                                # weights without the stage
    rd_std = (span.float() * 0.3).contiguous()
    scratch1 = ops.ls_scratch(env.graph, B, ws, num_draws=1)
    scratch4 = ops.ls_scratch(env.graph, B, ws, num_draws=4)
    assert scratch1 is not None and scratch4.numel() == 4 * scratch1.numel() == 4 * 3 * n * 8
    thresh = ops.maxcut_ls_threshold(env.graph, ws, rd_std, seed=7, num_spin=num_spin, scratch=scratch1)
    with pytest.raises(RuntimeError):
        ops.maxcut_ls_propose(env.graph, xs.clone(), ws, rd_std, thresh, vs.clone(), seed=7, draw=1)       # no scratch
    a, va = xs.clone(), vs.clone()
    for it in range(1, 5):
        x0, v0 = a.clone(), va.clone()
        ops.maxcut_ls_propose(env.graph, a, ws, rd_std, thresh, va, seed=7, draw=it, scratch=scratch1)
        changed = (a != x0).any(dim=1)
        assert bool((va >= v0).all()) and bool((va[~changed] == v0[~changed]).all())
    assert np.array_equal(va.cpu().numpy(), onp.maxcut_obj(a.cpu().numpy(), garr, False)) and bool((va > vs).any())
    b, vb = xs.clone(), vs.clone()
    ops.maxcut_ls_rounds(env.graph, b, ws, rd_std, thresh, vb, seed=7, first_draw=1, num_draws=4, scratch=scratch4)
    assert torch.equal(a, b) and torch.equal(va, vb)
    v1 = vb.clone()
    env.local_search_inplace(b, vb, num_iters=3, num_spin=num_spin)                                        # + K5 without the stage
    assert bool((vb >= v1).all()) and np.array_equal(vb.cpu().numpy(), onp.maxcut_obj(b.cpu().numpy(), garr, False))


@pytest.mark.parametrize("n,m", [(24000, 48000), (36000, 54000), (39936, 60000)])
def test_round_kernels_past_the_64_env_tile_on_half_tiles(n, m):
    """20 224 < N <= 39 936: no 64-env tile fits LDS; the weights pre-pass, K1, the apply kernel of the proposal rounds and the sweep
    run on HALF tiles (32 envs, 32-bit words: csrc/rls_tile32.h), the threshold / mask kernels as before (rd_std in LDS at 24 000,
    read from global memory beyond ~24 900).  Invariants of every round, all rounds at once == one round per call, and the whole
    local_search_inplace == the decomposed path (torch kthvalue + K6 per round on the kernels' own draws) for the same seed, bit
    for bit."""
    from rlsolver_amd import ops
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    B, num_spin = 100, 8
    garr = gnm_arr(n, m, seed=81)
    env = EnvMaxcut(mygraph=mygraph_of(garr), device=DEV, num_nodes=n, seed=11)
    assert not ops.local_search_fusable(env.graph, num_spin, B) and ops.ls_rounds_supported(env.graph, num_spin)
    xs = env.generate_xs_randomly(B)
    vs = env.calculate_obj_values(xs)
    assert np.array_equal(vs.cpu().numpy(), onp.maxcut_obj(xs.cpu().numpy(), garr, False))
    ws, span = ops.maxcut_ls_weights(env.graph, xs, 1, padded=True)
    c = ops.maxcut_node_cutdeg(env.graph, xs)
    deg = torch.from_numpy(np.bincount(env.graph.csr.eu, minlength=n)).to(DEV)
    assert torch.equal(ws[:, :n].long(), deg[None, :] - c)
    assert torch.equal(span.long(), (deg[None, :] - c).max(0)[0] - (deg[None, :] - c).min(0)[0])
    rd_std = (span.float() * 0.3).contiguous()
    scratch1 = ops.ls_scratch(env.graph, B, ws, num_draws=1)
    scratch4 = ops.ls_scratch(env.graph, B, ws, num_draws=4)
    assert scratch1 is not None and scratch4.numel() == 4 * scratch1.numel()
    thresh = ops.maxcut_ls_threshold(env.graph, ws, rd_std, seed=7, num_spin=num_spin, scratch=scratch1)
    # the threshold against torch on the kernels' own draws
    noisy0 = ws[:, :n].float() + ops.maxcut_ls_normals(B, n, 7, 0, DEV) * rd_std
    assert torch.equal(thresh, torch.kthvalue(noisy0, k=n - num_spin, dim=1)[0])
    a, va = xs.clone(), vs.clone()
    for it in range(1, 5):
        x0, v0 = a.clone(), va.clone()
        ops.maxcut_ls_propose(env.graph, a, ws, rd_std, thresh, va, seed=7, draw=it, scratch=scratch1)
        changed = (a != x0).any(dim=1)
        assert bool((va >= v0).all()) and bool((va[~changed] == v0[~changed]).all())
    assert np.array_equal(va.cpu().numpy(), onp.maxcut_obj(a.cpu().numpy(), garr, False)) and bool((va > vs).any())
    b, vb = xs.clone(), vs.clone()
    ops.maxcut_ls_rounds(env.graph, b, ws, rd_std, thresh, vb, seed=7, first_draw=1, num_draws=4, scratch=scratch4)
    assert torch.equal(a, b) and torch.equal(va, vb)
    # the whole call: round kernels (half tiles) == the decomposed path, same seed stream
    outs = []
    for fused in (True, False):
        e2 = EnvMaxcut(mygraph=mygraph_of(garr), device=DEV, num_nodes=n, seed=23)
        e2.fused_local_search = fused
        x2, v2 = xs.clone(), vs.clone()
        e2.local_search_inplace(x2, v2, num_iters=3, num_spin=num_spin)
        assert np.array_equal(v2.cpu().numpy(), onp.maxcut_obj(x2.cpu().numpy(), garr, False)) and bool((v2 >= vs).all())
        outs.append((x2, v2))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])

"""A dense graph (N = 3000, E = 1.2 M; stored edges 2.4 M when bidirectional): the MaxCut entry points run and match the C oracle
on a small batch -- looks for limits in the counter widths / degree caps (max degree ~900).  `python tools/sweeps/dense_graph.py`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import oracle_np as onp, oracle_c as oc
from rlsolver_amd import ops
from rlsolver_amd.envs.env_L2A import EnvMaxcut
from rlsolver_amd.graph import generate_gnm

dev = torch.device("cuda:0")
n, m, B = 3000, 1_200_000, 70
mg = generate_gnm(n, m, 3)
garr = np.asarray(mg, dtype=np.int64)
for bidir in (False, True):
    env = EnvMaxcut(mygraph=mg, device=dev, if_bidirectional=bidir, num_nodes=n)
    g = env.graph
    eu, ev = onp.stored_edges(garr, bidir)
    torch.manual_seed(0)
    xs = env.generate_xs_randomly(B)
    x_np = xs.cpu().numpy().astype(np.uint8)
    vs = env.calculate_obj_values(xs)
    want = oc.maxcut_obj(x_np, eu, ev, int(bidir))
    assert np.array_equal(vs.cpu().numpy(), want), "K1"
    d = ops.maxcut_delta_all(g, xs)
    # flip gain of node i = change of the cut when i flips: check a few against re-evaluation
    for i in (0, 17, n - 1):
        y = x_np.copy(); y[:, i] ^= 1
        assert np.array_equal(d[:, i].cpu().numpy(), oc.maxcut_obj(y, eu, ev, int(bidir)) - want), "K3"
    x5, v5 = xs.clone(), vs.clone()
    ops.maxcut_greedy_sweep(g, x5, v5)
    wx, wv = oc.greedy_sweep(x_np.copy(), want.astype(np.int64).copy(), eu, ev, int(bidir))
    assert np.array_equal(x5.cpu().numpy().astype(np.uint8), wx) and np.array_equal(v5.cpu().numpy(), wv), "K5"
    xl, vl = xs.clone(), vs.clone()
    env.local_search_inplace(xl, vl, num_iters=4, num_spin=8)
    assert bool((vl >= vs).all()) and np.array_equal(vl.cpu().numpy(), oc.maxcut_obj(xl.cpu().numpy().astype(np.uint8), eu, ev, int(bidir))), "LS"
    print(f"bidir={bidir}: max degree {g.csr.max_degree}, stored edges {len(eu)}: K1 K3 K5 LS match; fused LS: {ops.local_search_fusable(g, 8, B)}, "
          f"round kernels: {ops.ls_rounds_supported(g, 8)}, weights dtype {ops.ls_weight_dtype(g, 1)}")

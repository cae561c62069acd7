"""The tile kernels over graph SHAPES at one size (N = 2000-ish, 2^14 envs): G(n, m) at several densities, a torus (Gset G48-50),
a path, a star, a hub graph, BA.  Looks for schedules that degenerate (levels = N on a path, one giant row on a star).
`python tools/sweeps/graph_shape_sweep.py`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import ops
from rlsolver_amd.envs.env_L2A import EnvMaxcut
from rlsolver_amd.graph import generate_ba, generate_gnm
from rlsolver_amd.methods import MCPG as amcpg

dev = torch.device("cuda:0")
B = 1 << 14


def t_us(f, n=4):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def torus(a, b):
    idx = lambda i, j: (i % a) * b + (j % b)
    return [(min(idx(i, j), idx(i + 1, j)), max(idx(i, j), idx(i + 1, j)), 1) for i in range(a) for j in range(b)] + \
           [(min(idx(i, j), idx(i, j + 1)), max(idx(i, j), idx(i, j + 1)), 1) for i in range(a) for j in range(b)]


n = 2000
shapes = [("G(2000, 19990) = G22", n, generate_gnm(n, 19990, 22)), ("G(2000, 4000) sparse", n, generate_gnm(n, 4000, 1)),
          ("G(2000, 200000) dense, degree 200", n, generate_gnm(n, 200000, 2)), ("torus 40 x 50", n, torus(40, 50)),
          ("path", n, [(i, i + 1, 1) for i in range(n - 1)]), ("star", n, [(0, i, 1) for i in range(1, n)]),
          ("star + G(2000, 8000)", n, sorted(set((0, i, 1) for i in range(1, n)) | set((a, b, 1) for a, b, _ in generate_gnm(n, 8000, 3)))),
          ("BA m=4", n, generate_ba(n, 4, 3)), ("BA m=20", n, generate_ba(n, 20, 3))]
print(f"{'graph':38s} {'E':>7s} {'maxdeg':>6s} {'levels':>6s} | us per call at 2^14 envs: K1  K3  K5  K6  ls_w  LS  MCPG-sampler(2^14 chains, num_ls=2)")
for name, n, mg in shapes:
    env = EnvMaxcut(mygraph=mg, device=dev, num_nodes=n)
    g = env.graph
    x = torch.rand(B, n, device=dev) < 0.5
    v = ops.maxcut_obj(g, x)
    d = torch.empty((B, n), dtype=torch.int32, device=dev)
    m = torch.rand(B, n, device=dev) < 0.004
    arr = np.asarray(mg, dtype=np.int64)
    data = amcpg.make_data(n, arr[:, 0], arr[:, 1], dev)
    xs = (torch.rand((n, B), device=dev) < 0.5).float()
    ts = [t_us(lambda: ops.maxcut_obj(g, x)), t_us(lambda: ops.maxcut_delta_all(g, x, out=d)), t_us(lambda: ops.maxcut_greedy_sweep(g, x, v)),
          t_us(lambda: ops.maxcut_propose_accept(g, x, m, v)), t_us(lambda: ops.maxcut_ls_weights(g, x, 1)),
          t_us(lambda: env.local_search_inplace(x, v, num_iters=8, num_spin=8)),
          t_us(lambda: amcpg.sampler_func(data, xs, 2, B // 128, 128, dev))]
    print(f"{name:38s} {len(mg):7d} {g.csr.max_degree:6d} {g.num_sweep_levels:6d} | " + " ".join(f"{t:8.0f}" for t in ts), flush=True)

"""Entry points beyond the MaxCut tile kernels over awkward shapes (odd N, batch sizes off the tile): time per call, next to
the round shape.  `python tools/dev/shape_sweep.py`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import ops, ops_mcpg_tsp as mops
from rlsolver_amd.envs.spinsystem import SpinSystem
from rlsolver_amd.graph import generate_gnm, generate_tsp_coords, tsp_tables
from rlsolver_amd.methods import MCPG as amcpg, MCPG_qubo as mq

dev = torch.device("cuda:0")


def t_us(f, n=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def show(name, shapes, make):
    ts = []
    for sh in shapes:
        try:
            ts.append(t_us(make(*sh)))
        except Exception as e:   # noqa
            ts.append(float("nan")); print("   ", name, sh, type(e).__name__, str(e)[:90])
    flag = "   <-- slower than the round shape" if np.nanmax(ts[1:]) > 1.6 * ts[0] else ""
    print(f"{name:40s}" + " ".join(f"{str(sh):>16s}:{t:7.0f}" for sh, t in zip(shapes, ts)) + flag, flush=True)


# MCPG (chains C, nodes N)
def mk_sampler(n, C):
    arr = np.asarray(generate_gnm(n, 10 * n, 1), dtype=np.int64)
    data = amcpg.make_data(n, arr[:, 0], arr[:, 1], dev)
    x = (torch.rand((n, C), device=dev) < 0.5).float()
    return lambda: amcpg.sampler_func(data, x, 2, C // 128, 128, dev)
show("MCPG sampler_func (N, C)", [(2000, 16384), (1999, 16384), (2001, 16384), (2000, 16384 + 128)], mk_sampler)
def mk_metro(n, C):
    p = torch.full((n,), 0.5, device=dev)
    x = (torch.rand((n, C), device=dev) < 0.5).float()
    return lambda: amcpg.metro_sampling(p, x, n // 10, dev)
show("MCPG metro_sampling (N, C)", [(2000, 16384), (1999, 16384), (2000, 16384 + 1), (2000, 16000)], mk_metro)

# QUBO (n, chains)
def mk_qubo(n, C):
    Q = torch.randn(n, n, device=dev).round(); Q = Q + Q.T
    x = (torch.rand((n, C), device=dev) < 0.5).float()
    return lambda: mq.qubo_local_search_value(Q, x, 1, False)
show("K11 qubo_local_search_value (n, C)", [(1024, 8192), (1000, 8192), (999, 8192), (1001, 8192), (1024, 8192 + 5)], mk_qubo)

# TSP (N, B)
def mk_tour(N, B):
    D = torch.from_numpy(tsp_tables(generate_tsp_coords(N, 1), 5)[0]).to(dev)
    t = mops.rand_perms(B, N, 3, dev)
    return lambda: mops.tsp_tour_length(D, t)
show("K12 tsp_tour_length (N, B)", [(100, 65536), (101, 65536), (99, 65536), (100, 65537), (1000, 8192), (1001, 8192)], mk_tour)
def mk_swap(N, B):
    dist, near, rnd = tsp_tables(generate_tsp_coords(N, 1), min(20, N - 2))
    D, nn, rr = (torch.from_numpy(a).to(dev) for a in (dist, near, rnd))
    t = mops.rand_perms(B, N, 3, dev)
    return lambda: mops.tsp_swap_delta_all(D, nn, rr, t, 1.0, 7)
show("K13 tsp_swap_delta_all (N, B)", [(100, 65536), (101, 65536), (99, 65536), (100, 65537)], mk_swap)

# spin env (N, B)
def mk_spin(n, B):
    rng = np.random.RandomState(0)
    mg = [(a, b, int(rng.choice([-1, 1]))) for a, b, _ in generate_gnm(n, 10 * n, 2)]
    e = SpinSystem(mg, n, B, max_steps=10 ** 6, device=dev, include_adjacency=False)
    a = torch.randint(0, n, (B,), device=dev)
    return lambda: e.step(a)
show("spin step + rows-only observation (N, B)", [(2000, 16384), (1999, 16384), (2001, 16384), (2000, 16383)], mk_spin)

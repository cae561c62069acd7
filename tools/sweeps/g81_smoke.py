"""Classes of the boundary on a G81-sized graph (N = 20 000, E = 40 000): LocalSearch runs and stays consistent; ISCO_maxcut.step
reports its limit (its Gumbel top-k sorts next-power-of-two(N) keys in LDS per sample: N <= 8192; the reference runs it on BA-100)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import ops
from rlsolver_amd.graph import generate_gnm

dev = torch.device("cuda:0")
n, m = 20000, 40000
mg = generate_gnm(n, m, 81)


def t_us(f, k=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3


from rlsolver_amd.envs.env_ISCO_maxcut import ISCO_maxcut
from rlsolver_amd.methods.LocalSearch import LocalSearch
from rlsolver_amd.envs.env_L2A import EnvMaxcut
from rlsolver_amd.methods.util_evaluator import Evaluator
import inspect
print("ISCO_maxcut ctor:", inspect.signature(ISCO_maxcut.__init__))
try:
    arr = np.asarray(mg, dtype=np.int64)
    params = {"num_nodes": n, "num_edges": m, "edge_from": torch.from_numpy(arr[:, 0]).to(dev), "edge_to": torch.from_numpy(arr[:, 1]).to(dev)}
    smp = ISCO_maxcut(params, batch_size=1024, device=dev) if "batch_size" in inspect.signature(ISCO_maxcut.__init__).parameters else ISCO_maxcut(params)
    x = smp.random_gen_init_sample()
    print("ISCO_maxcut.step us:", round(t_us(lambda: smp.step(x, 8, torch.tensor(1.0, device=dev)))))
except Exception as e:
    print("ISCO_maxcut:", type(e).__name__, str(e)[:200])
try:
    env = EnvMaxcut(mygraph=mg, device=dev, num_nodes=n)
    ls = LocalSearch(env, n)
    xs = env.generate_xs_randomly(512)
    ls.reset(xs)
    v0 = ls.good_vs.clone()
    print("LocalSearch.random_search us:", round(t_us(lambda: ls.random_search(num_iters=8, num_spin=8))))
    assert bool((ls.good_vs >= v0).all()) and torch.equal(env.calculate_obj_values(ls.good_xs), ls.good_vs)
except Exception as e:
    print("LocalSearch:", type(e).__name__, str(e)[:200])
print("done")

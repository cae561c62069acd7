"""The TSP entry points over city counts either side of "the distance matrix fits LDS" (N = 200): us per call.
`python tools/sweeps/tsp_n_sweep.py`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import ops_mcpg_tsp as mops
from rlsolver_amd.envs.env_ISCO import ISCO_TSP
from rlsolver_amd.graph import generate_tsp_coords, tsp_tables

dev = torch.device("cuda:0")
NS = (20, 52, 100, 150, 200, 201, 300, 500, 1000)
B = 8192


def t_us(f, n=3):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


rows = {}
for N in NS:
    K = min(20, N - 2)
    dist, near, rnd = tsp_tables(generate_tsp_coords(N, 1), K)
    params = {"distance": torch.from_numpy(dist).to(dev), "nearest_indices": torch.from_numpy(near).to(dev),
              "random_indices": torch.from_numpy(rnd).to(dev), "num_nodes": N}
    env = ISCO_TSP(params, batch_size=B, K=K, device=dev)
    x = env.random_gen_init_sample(params)
    tests = (("K12 calculate_distance", lambda: env.calculate_distance(x)),
             ("K13 opt_2 (swap deltas)", lambda: env.opt_2(x, torch.tensor(1.0, device=dev))),
             ("I2 step (path_length 4)", lambda: env.step(x, 4, torch.tensor(1.0, device=dev))))
    for name, f in tests:
        try:
            rows.setdefault(name, []).append(t_us(f))
        except Exception as e:   # noqa
            rows.setdefault(name, []).append(float("nan")); print("   ", name, N, type(e).__name__, str(e)[:140])
print(f"{B} tours; us per call at N = " + "".join(f"{n:>8d}" for n in NS))
for name, ts in rows.items():
    print(f"{name:28s}" + "".join(f"{t:8.0f}" for t in ts))

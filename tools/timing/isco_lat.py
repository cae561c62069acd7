"""Dev: latency of one ISCO sampler step by batch size (the reference's own configs run BATCH_SIZE = 1)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from rlsolver_amd.graph import tsp_tables, generate_tsp_coords, generate_gnm
from rlsolver_amd.envs.env_ISCO import ISCO_TSP
from rlsolver_amd.envs.env_ISCO_maxcut import ISCO_maxcut
dev = torch.device("cuda:0")


def t(fn, it=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


N = 100
dist, near, rnd = tsp_tables(generate_tsp_coords(N, 100), K=20)
params = {"num_nodes": N, "distance": torch.from_numpy(dist).to(dev), "nearest_indices": torch.from_numpy(near).to(dev),
          "random_indices": torch.from_numpy(rnd).to(dev)}
for B in (1, 64, 4096, 65536):
    env = ISCO_TSP(params, batch_size=B, K=20, device=dev)
    x = env.random_gen_init_sample()
    for L in (1, 8):
        print("ISCO_TSP  B=%6d L=%2d: %8.1f us" % (B, L, t(lambda: env.step(x, L, 1.0))))
n, m = 2000, 19990
g = np.asarray(generate_gnm(n, m, 22), dtype=np.int64)
pm = {"num_nodes": n, "num_edges": m, "edge_from": torch.from_numpy(g[:, 0].copy()).to(dev), "edge_to": torch.from_numpy(g[:, 1].copy()).to(dev)}
for B in (1, 64, 4096):
    env = ISCO_maxcut(pm, batch_size=B, device=dev)
    x = env.random_gen_init_sample()
    for L in (1, 16):
        print("ISCO_maxcut B=%6d L=%2d: %8.1f us" % (B, L, t(lambda: env.step(x, L, 1.0))))

"""local_search_inplace, fused kernel vs round kernels (threshold + one mask launch for all rounds + apply + sweep) by batch size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import graph
from rlsolver_amd.envs.env_L2A import EnvMaxcut
dev = torch.device('cuda:0')


def t(f, K=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(K): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / K * 1e3


for name, g, n in (("G22-sized", graph.generate_gnm(2000, 19990, seed=22), 2000), ("BA-1e4", graph.generate_ba(10000, 5, seed=5), 10000),
                   ("G14-sized", graph.generate_gnm(800, 4694, seed=14), 800)):
    env = EnvMaxcut(mygraph=g, device=dev, num_nodes=n)
    for B in (2048, 8192, 16384, 32768, 65536):
        xs = env.generate_xs_randomly(B); vs = env.calculate_obj_values(xs)
        row = []
        for form in ("auto", "fused", "rounds"):
            env.force_ls_fused, env.force_ls_rounds = form == "fused", form == "rounds"
            try:
                row.append(f"{form} {t(lambda: env.local_search_inplace(xs, vs)):8.1f}")
            except Exception as ex:
                row.append(f"{form} n/a ({type(ex).__name__})")
        env.force_ls_fused = env.force_ls_rounds = False
        print(f"{name} B={B}: " + " | ".join(row) + " us", flush=True)

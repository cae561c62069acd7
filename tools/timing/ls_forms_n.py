"""local_search_inplace, fused kernel vs round kernels, by row length (the fused kernel's LDS layout holds N <= ~6500)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import graph
from rlsolver_amd.envs.env_L2A import EnvMaxcut
dev = torch.device('cuda:0')


def t(f, K=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(K): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / K * 1e3


for n in (1000, 3008, 4000, 5008, 6000, 6496):
    env = EnvMaxcut(mygraph=graph.generate_gnm(n, 5 * n, seed=n), device=dev, num_nodes=n)
    for B in (8192, 32768, 65536):
        xs = env.generate_xs_randomly(B); vs = env.calculate_obj_values(xs)
        row = []
        for form in ("auto", "fused", "rounds"):
            env.force_ls_fused, env.force_ls_rounds = form == "fused", form == "rounds"
            try:
                row.append(f"{form} {t(lambda: env.local_search_inplace(xs, vs)):8.1f}")
            except Exception as ex:
                row.append(f"{form} n/a ({type(ex).__name__})")
        env.force_ls_fused = env.force_ls_rounds = False
        print(f"N={n} B={B}: " + " | ".join(row) + " us", flush=True)

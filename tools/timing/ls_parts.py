"""local_search_inplace at G22 size (weights pre-pass + fused kernel), 4096 and 2^16 envs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import ops, graph
from rlsolver_amd.envs.env_L2A import EnvMaxcut
dev = torch.device('cuda:0')
n, m = 2000, 19990
g = graph.generate_gnm(n, m, seed=22)
env = EnvMaxcut(mygraph=g, device=dev, num_nodes=n)
def t(f, K=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(K): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / K * 1e3
for B in (4096, 65536):
    xs = env.generate_xs_randomly(B); vs = env.calculate_obj_values(xs)
    us_w = t(lambda: ops.maxcut_ls_weights(env.graph, xs, 1))
    us_all = t(lambda: env.local_search_inplace(xs.clone(), vs.clone()))
    us_clone = t(lambda: (xs.clone(), vs.clone()))
    print(f"B={B}: ls_weights op {us_w:.1f} us, whole local_search_inplace {us_all:.1f} us (clones {us_clone:.1f})")

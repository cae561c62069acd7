import sys, os
sys.path.insert(0, ".")
import torch, numpy as np
from rlsolver_amd import graph
from rlsolver_amd.envs.env_L2A import EnvMaxcut
dev = torch.device('cuda:0')
def t(f, reps=3):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
from rlsolver_amd import ops
for n, m in ((16000, 32000), (20000, 40000), (24000, 48000), (36000, 72000), (48000, 96000)):
    g = graph.generate_gnm(n, m, seed=1)
    env = EnvMaxcut(mygraph=g, device=dev, num_nodes=n)
    B = 4096
    xs = env.generate_xs_randomly(B)
    vs = env.calculate_obj_values(xs)
    ms = t(lambda: env.local_search_inplace(xs, vs))
    k3 = t(lambda: ops.maxcut_delta_all(env.graph, xs))
    k2 = t(lambda: env.calculate_obj_values_for_loop(xs))
    k5 = t(lambda: ops.maxcut_greedy_sweep(env.graph, xs, vs))
    print(f"N={n} B={B}: local_search_inplace {ms:.2f} ms, K3 delta_all {k3:.2f} ms, K2 {k2:.2f} ms, K5 {k5:.2f} ms", flush=True)

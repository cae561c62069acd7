"""Fused local search at G22 / 2^16: time against the number of proposal rounds (slope = one round, intercept = load + count + threshold
pass + sweep + store) and the weights pre-pass beside it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import graph, ops
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


env = EnvMaxcut(mygraph=graph.generate_gnm(2000, 19990, seed=22), device=dev, num_nodes=2000)
B = 1 << 16
xs = env.generate_xs_randomly(B); vs = env.calculate_obj_values(xs)
print("weights pre-pass", round(t(lambda: ops.maxcut_ls_weights(env.graph, xs, 1, padded=True, return_minmax=True)), 1), "us;  K5 alone",
      round(t(lambda: ops.maxcut_greedy_sweep(env.graph, xs, vs)), 1), "us;  K1 alone", round(t(lambda: ops.maxcut_obj(env.graph, xs)), 1), "us")
for k in (0, 1, 2, 4, 8, 16):
    print(f"num_iters={k}: {t(lambda: env.local_search_inplace(xs, vs, num_iters=k)):8.1f} us", flush=True)

#!/usr/bin/env python3
"""Where does local_search_inplace spend its time? (torch ops vs kernels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlsolver_amd import ops
from rlsolver_amd.envs.env_L2A import EnvMaxcut
from rlsolver_amd.graph import generate_gnm

dev = torch.device("cuda:0")
n, m = 2000, 19990
env = EnvMaxcut(mygraph=generate_gnm(n, m, 22), device=dev, num_nodes=n)


def T(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for B in (4096, 65536):
    torch.manual_seed(0)
    xs = env.generate_xs_randomly(B); vs = env.calculate_obj_values(xs)
    vs_raw = env.calculate_obj_values_for_loop(xs, if_sum=False)
    ws = env.n0_num_n1 - vs_raw
    ws_std = ws.max(dim=0, keepdim=True)[0] - ws.min(dim=0, keepdim=True)[0]
    rd_std = ws_std.float() * 0.3
    spin_rand = ws + torch.randn_like(ws, dtype=torch.float32) * rd_std
    thresh = torch.kthvalue(spin_rand, k=n - 8, dim=1)[0][:, None]
    mask = spin_rand.gt(thresh)
    print(f"B={B}")
    print("  cutdeg (K2)        %8.3f ms" % T(lambda: env.calculate_obj_values_for_loop(xs, if_sum=False)))
    print("  ws = deg - raw     %8.3f ms" % T(lambda: env.n0_num_n1 - vs_raw))
    print("  ws max/min dim0    %8.3f ms" % T(lambda: ws.max(dim=0, keepdim=True)[0] - ws.min(dim=0, keepdim=True)[0]))
    print("  randn_like         %8.3f ms" % T(lambda: torch.randn_like(ws, dtype=torch.float32)))
    print("  ws + randn*rd_std  %8.3f ms" % T(lambda: ws + torch.randn_like(ws, dtype=torch.float32) * rd_std))
    print("  kthvalue           %8.3f ms" % T(lambda: torch.kthvalue(spin_rand, k=n - 8, dim=1)))
    print("  gt(thresh)         %8.3f ms" % T(lambda: spin_rand.gt(thresh)))
    print("  propose_accept     %8.3f ms" % T(lambda: ops.maxcut_propose_accept(env.graph, xs, mask, vs)))
    print("  greedy_sweep       %8.3f ms" % T(lambda: ops.maxcut_greedy_sweep(env.graph, xs, vs)))
    print("  local_search total %8.3f ms" % T(lambda: env.local_search_inplace(xs, vs, 8, 8, 0.3), 3))

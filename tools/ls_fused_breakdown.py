#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlsolver_amd import ops
from rlsolver_amd.graph import build_csr, generate_gnm
dev = torch.device("cuda:0")
n, m = 2000, 19990
g = ops.DeviceGraph(build_csr(generate_gnm(n, m, 22), num_nodes=n), dev)

def T(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for B in (4096, 65536):
    xs = ops.rand_spins(B, n, 1, dev); vs = ops.maxcut_obj(g, xs)
    ws, wstd = ops.maxcut_ls_weights(g, xs, 1)
    rd = (wstd.float() * 0.3).contiguous()
    noise = torch.randn((9, B, n), device=dev) if B <= 4096 else None
    print(f"B={B}")
    print("  ls_weights pre-pass      %8.3f ms" % T(lambda: ops.maxcut_ls_weights(g, xs, 1)))
    print("  sweep alone              %8.3f ms" % T(lambda: ops.maxcut_greedy_sweep(g, xs, vs)))
    for it in (0, 1, 8):
        print(f"  fused iters={it} (prod rng) %8.3f ms" % T(lambda: ops.maxcut_local_search(g, xs, ws, rd, vs, it, 8, seed=3)))
    if noise is not None:
        print("  fused iters=8 (noise in) %8.3f ms" % T(lambda: ops.maxcut_local_search(g, xs, ws, rd, vs, 8, 8, noise=noise)))

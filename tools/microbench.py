#!/usr/bin/env python3
"""Quick per-kernel timing on one GPU (development aid; bench.py is the judged harness)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rlsolver_amd import _abi, ops
from rlsolver_amd.graph import build_csr, generate_gnm

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2000)
ap.add_argument("--m", type=int, default=19990)
ap.add_argument("--B", type=int, default=65536)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--slots", type=int, default=8)
ap.add_argument("--which", default="step,step_inplace,obj,sweep,propose")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = ops.DeviceGraph(build_csr(generate_gnm(a.n, a.m, 22), num_nodes=a.n), dev)
B, N = a.B, a.n


def timeit(fn, iters):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


which = a.which.split(",")
x = ops.rand_spins(B, N, 1, dev)
if "step" in which:
    slots = [torch.empty_like(x) for _ in range(a.slots)]
    slots[0].copy_(x)
    obj = ops.maxcut_obj(g, x).to(torch.int32)
    reward = torch.empty(B, dtype=torch.float32, device=dev)
    acts = [ops.rand_actions(B, N, 7, s, dev) for s in range(16)]
    by = B * (2 * N + 20)
    modes = [int(m) for m in os.environ.get("NTS", "0,1").split(",")]      # nontemporal stores off / on (rls_tuning_set)
    for rep in range(2):
        for mode in modes:
            _abi.tuning_set("RLS_STEP_NTS", mode)
            t = timeit(lambda i: ops.maxcut_step(g, slots[i % a.slots], slots[(i + 1) % a.slots], acts[i % 16], obj, reward), a.iters)
            print(f"step emit, nontemporal stores {mode}: {t*1e6:9.1f} us  {B/t:.3e} steps/s  {by/t/1e9:8.1f} GB/s algorithmic ({by/t/8e12*100:.1f}% of 8 TB/s)")
    _abi.tuning_unset("RLS_STEP_NTS")
if "step_inplace" in which:
    obj = ops.maxcut_obj(g, x).to(torch.int32)
    reward = torch.empty(B, dtype=torch.float32, device=dev)
    acts = [ops.rand_actions(B, N, 7, s, dev) for s in range(16)]
    xx = x.clone()
    t = timeit(lambda i: ops.maxcut_step(g, xx, xx, acts[i % 16], obj, reward), a.iters)
    print(f"step inplace: {t*1e6:9.1f} us  {B/t:.3e} steps/s")
if "obj" in which:
    out = torch.empty(B, dtype=torch.int64, device=dev)
    t = timeit(lambda i: ops.maxcut_obj(g, x, out), a.iters)
    by = B * (N + 8)
    print(f"obj         : {t*1e6:9.1f} us  {B/t:.3e} evals/s  {by/t/1e9:8.1f} GB/s algorithmic ({by/t/8e12*100:.1f}%)")
if "sweep" in which:
    Bs = min(B, 65536)
    xs = x[:Bs].clone()
    vs = ops.maxcut_obj(g, xs)
    t = timeit(lambda i: ops.maxcut_greedy_sweep(g, xs, vs), max(3, a.iters // 10))
    print(f"sweep B={Bs}: {t*1e6:9.1f} us  {Bs*N/t:.3e} candidate flips/s")
    xs = x[:4096].clone(); vs = ops.maxcut_obj(g, xs)
    t = timeit(lambda i: ops.maxcut_greedy_sweep(g, xs, vs), max(3, a.iters // 10))
    print(f"sweep B=4096: {t*1e6:9.1f} us  {4096*N/t:.3e} candidate flips/s")
if "propose" in which:
    mask = (torch.rand((B, N), device=dev) < 0.004)
    xs = x.clone(); vs = ops.maxcut_obj(g, xs)
    t = timeit(lambda i: ops.maxcut_propose_accept(g, xs, mask, vs), max(3, a.iters // 5))
    print(f"propose     : {t*1e6:9.1f} us  {B/t:.3e} proposals/s")

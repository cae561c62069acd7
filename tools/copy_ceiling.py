#!/usr/bin/env python3
"""What a plain device copy of the same bytes achieves on this box (the practical ceiling for K4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlsolver_amd import ops
from rlsolver_amd.graph import build_csr, generate_gnm

n, m, B = 2000, 19990, 65536
dev = torch.device("cuda:0")
g = ops.DeviceGraph(build_csr(generate_gnm(n, m, 22), num_nodes=n), dev)
obj = torch.zeros(B, dtype=torch.int32, device=dev)
reward = torch.empty(B, dtype=torch.float32, device=dev)
acts = [ops.rand_actions(B, n, 7, s, dev) for s in range(16)]


def timeit(fn, iters=200):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for S in (2, 4, 8, 32):
    slots = [ops.rand_spins(B, n, s, dev) for s in range(S)]
    f32 = [s_.view(torch.float32) for s_ in slots]
    t_copy = timeit(lambda i: f32[(i + 1) % S].copy_(f32[i % S]))
    t_step = timeit(lambda i: ops.maxcut_step(g, slots[i % S], slots[(i + 1) % S], acts[i % 16], obj, reward))
    by = 2 * B * n
    print(f"slots={S:2d}: torch copy_ {t_copy*1e6:6.1f} us ({by/t_copy/1e12:.2f} TB/s)   "
          f"step emit {t_step*1e6:6.1f} us ({B*(2*n+20)/t_step/1e12:.2f} TB/s)")
    del slots, f32

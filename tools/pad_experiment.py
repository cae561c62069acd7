#!/usr/bin/env python3
"""Does the distance between the read slot and the write slot matter (HBM channel mapping)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlsolver_amd import ops
from rlsolver_amd.graph import build_csr, generate_gnm

n, m, B, S = 2000, 19990, 65536, 8
dev = torch.device("cuda:0")
g = ops.DeviceGraph(build_csr(generate_gnm(n, m, 22), num_nodes=n), dev)
obj = torch.zeros(B, dtype=torch.int32, device=dev)
reward = torch.empty(B, dtype=torch.float32, device=dev)
acts = [ops.rand_actions(B, n, 7, s, dev) for s in range(16)]
slot_bytes = B * n


def timeit(fn, iters=300):
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


res = {}
pads = [0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, (3 << 20) + 12288, 123 * 4096 + 256]
for rep in range(2):
    for pad in pads:
        big = torch.empty(S * (slot_bytes + pad) + 256, dtype=torch.uint8, device=dev)
        slots = [big[s * (slot_bytes + pad): s * (slot_bytes + pad) + slot_bytes].view(B, n).view(torch.bool) for s in range(S)]
        ops.rand_spins(B, n, 1, dev, out=slots[0])
        t = timeit(lambda i: ops.maxcut_step(g, slots[i % S], slots[(i + 1) % S], acts[i % 16], obj, reward))
        res.setdefault(pad, []).append(t)
        del big, slots
for pad in pads:
    t = min(res[pad])
    print(f"pad {pad:>9d} B: {t*1e6:6.1f} us  {B*(2*n+20)/t/8e12*100:5.1f}% of 8 TB/s")

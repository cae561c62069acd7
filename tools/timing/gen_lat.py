"""Reset-path generators against the write ceiling: rand_spins [B, N] uint8, rand_perms [B, N] int64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import ops, ops_mcpg_tsp as mops
dev = torch.device("cuda:0")
def t_us(fn, n=50):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for B, N in ((65536, 2000), (131072, 10000), (65536, 800), (65536, 1000), (262144, 2000), (4096, 2000)):
    out = torch.empty((B, N), dtype=torch.bool, device=dev)
    us = t_us(lambda i: ops.rand_spins(B, N, seed=i, device=dev, out=out))
    print(f"rand_spins B={B} N={N}: {us:.1f} us {B * N / us / 8e6:.3f} of 8 TB/s")
for B, N in ((65536, 100), (262144, 100), (65536, 52), (16384, 200)):
    us = t_us(lambda i: mops.rand_perms(B, N, i, dev))
    print(f"rand_perms B={B} N={N}: {us:.1f} us {B * N * 8 / us / 8e6:.3f} of 8 TB/s")

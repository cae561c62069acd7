"""Dev timing of the small generator / reduction kernels (K14 rand_spins, rand_perms, get_return bit sums)."""
import sys
import torch
sys.path.insert(0, ".")
from rlsolver_amd import ops, ops_mcpg_tsp as mops

dev = torch.device("cuda:0")


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for B, N in ((1 << 16, 2000), (1 << 17, 10000), (1 << 12, 100)):
    x = torch.empty(B, N, dtype=torch.uint8, device=dev)
    us = timeit(lambda: ops.rand_spins(B, N, 7, dev, out=x))
    print("rand_spins B=%d N=%d: %.1f us  %.2f TB/s" % (B, N, us, B * N / us / 1e6))
us = timeit(lambda: mops.rand_perms(1 << 16, 100, 3, dev))
p = mops.rand_perms(1 << 16, 100, 3, dev)
print("rand_perms 2^16 x 100: %.1f us  %.2f TB/s" % (us, p.numel() * 8 / us / 1e6))
assert bool((p.sort(dim=1).values == torch.arange(100, device=dev)).all())
N, C = 10000, 1 << 18
words = torch.randint(-2**63, 2**63 - 1, (C // 64, N), dtype=torch.int64, device=dev)
pc = mops.PackedChains(words, C)
val = torch.randn(C, device=dev)
us = timeit(lambda: mops.mcpg_value_bit_sums(pc, val))
print("value_bit_sums BA-1e4 2^18: %.1f us  %.2f TB/s" % (us, words.numel() * 8 / us / 1e6))
A = mops.mcpg_value_bit_sums(pc, val)
bits = ((words[:64].unsqueeze(-1) >> torch.arange(64, device=dev)) & 1).double()      # [64 tiles, N, 64]
ref = torch.einsum("tne,te->n", bits, val[:64 * 64].view(64, 64).double())
A64 = mops.mcpg_value_bit_sums(mops.PackedChains(words[:64].contiguous(), 64 * 64), val[:64 * 64].contiguous())
print("bit sums max abs err vs f64 (64 tiles):", float((A64.double() - ref).abs().max()))

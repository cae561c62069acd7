"""Dev timing: bit-pack / unpack of the f32 [N, C] chain surface."""
import sys
import torch
sys.path.insert(0, ".")
from rlsolver_amd import ops_mcpg_tsp as mops
dev = torch.device("cuda:0")
N, C = 10000, 1 << 18
xs = (torch.rand(N, C, device=dev) < 0.5).float()


def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


pk = mops.PackedChains.pack(xs)
assert torch.equal(pk.unpack(), xs)
us = t(lambda: mops.PackedChains.pack(xs))
print("pack   f32 [1e4, 2^18]: %.0f us  %.2f TB/s" % (us, xs.numel() * 4 / us / 1e6))
us = t(lambda: pk.unpack())
print("unpack f32 [1e4, 2^18]: %.0f us  %.2f TB/s" % (us, xs.numel() * 4 / us / 1e6))

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


from rlsolver_amd import _abi
# RLS_MCPG_SHIM: 0 = the round-4 kernels; low nibble 1 = line-wide streams, bits 4..7 = 1 + log2 spans per wave, bits 8..11 = 1: unpack row-sequential
variants = [int(v, 0) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1"])]
out = torch.empty_like(xs)
for rep in range(2):
    for v in variants:
        _abi.tuning_set("RLS_MCPG_SHIM", v)
        pk = mops.PackedChains.pack(xs)
        assert torch.equal(pk.unpack(), xs)
        us = t(lambda: mops.PackedChains.pack(xs))
        us2 = t(lambda: pk.unpack())
        print("shim 0x%03x  pack f32 [1e4, 2^18]: %.0f us  %.2f TB/s   unpack: %.0f us  %.2f TB/s" % (v, us, xs.numel() * 4 / us / 1e6, us2, xs.numel() * 4 / us2 / 1e6))

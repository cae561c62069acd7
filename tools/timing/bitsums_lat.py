import sys, torch
sys.path.insert(0, ".")
from rlsolver_amd import ops_mcpg_tsp as mops
dev = torch.device("cuda:0")
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for N in (800, 832, 1000, 2000, 10000):
    for C in (4096, 65536, 262144):
        words = torch.randint(-2**63, 2**63 - 1, (C // 64, N), dtype=torch.int64, device=dev)
        pc = mops.PackedChains(words, C)
        val = torch.randn(C, device=dev)
        print("N=%5d C=%6d: bit sums %.1f us" % (N, C, t(lambda: mops.mcpg_value_bit_sums(pc, val))))

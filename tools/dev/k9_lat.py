import sys, torch, numpy as np
from rlsolver_amd import ops_mcpg_tsp as mops
from rlsolver_amd.methods import MCPG as amcpg
dev = torch.device('cuda:0')
n = 10000
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
xs = (torch.rand((n, C), device=dev) < 0.5).float()
probs = torch.rand(n, device=dev) * 0.6 + 0.2
acc = torch.zeros(5000, dtype=torch.int64, device=dev)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for T in (0, 100, 1000):
    for wb in (False, True):
        ms = t(lambda: mops.mcpg_metro_rounds(xs, probs, T, seed=1, write_back=wb, accepts=acc))
        print(f"T={T:5d} write_back={wb!s:5s} {ms:.3f} ms")
ms = t(lambda: amcpg.metro_sampling(probs, xs, 1000, device=dev))
print("metro_sampling(T=1000) whole call", f"{ms:.3f} ms")

"""Dev (build with RLS_EXTRA_CFLAGS=-DK9_PROF): shader-clock cycles the K9 walker spends per round."""
import sys
import torch
sys.path.insert(0, ".")
from rlsolver_amd import ops_mcpg_tsp as mops
dev = torch.device("cuda:0")
N, C, T = 10000, 1 << 18, 1000
probs = torch.rand(N, device=dev) * 0.6 + 0.2
kept = mops.PackedChains.empty(N, C, dev)
kept.words.random_(-2**62, 2**62)
acc = torch.zeros(T, dtype=torch.int64, device=dev)
for _ in range(2):
    acc.zero_()
    mops.mcpg_metro_rounds(kept, probs, T, seed=5, accepts=acc)
torch.cuda.synchronize()
a = acc.cpu().numpy()
print("walker cycles per round %.1f (shader clock), walker total %d, window loop total %d" % (a[T - 1] / T, a[T - 1], a[T - 2]))

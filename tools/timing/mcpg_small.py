"""Dev: an MCPG round at the reference's default-ish sizes (Gset-800-node graph, 512 x 128 chains)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from rlsolver_amd.graph import generate_gnm
from rlsolver_amd.methods import MCPG as amcpg
from rlsolver_amd.ops_mcpg_tsp import PackedChains
dev = torch.device("cuda:0")
for n, m, M, R in ((800, 4694, 512, 128), (2000, 19990, 512, 128), (800, 4694, 64, 64)):
    arr = np.asarray(generate_gnm(n, m, 14), dtype=np.int64)
    data = amcpg.make_data(n, arr[:, 0], arr[:, 1], dev)
    kept = PackedChains.pack((torch.rand(n, M, device=dev) < 0.5).float())
    rnd = amcpg.MCPGRound(data, kept, torch.zeros(M, device=dev), M, R, 5)
    probs = torch.full((n,), 0.5, device=dev)
    pr = torch.full((n,), 0.4, device=dev, requires_grad=True)
    for _ in range(3):
        rnd.step(probs)
    torch.cuda.synchronize()
    e0, e1, e2 = torch.cuda.Event(True), torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10):
        rnd.step(probs)
    e1.record()
    for _ in range(10):
        o = rnd.get_return(pr); o.backward()
    e2.record(); torch.cuda.synchronize()
    print("N=%d M=%d R=%d: round %.1f us, get_return fwd+bwd %.1f us" % (n, M, R, e0.elapsed_time(e1) * 100, e1.elapsed_time(e2) * 100))

# where does get_return's time go at N = 800?
from rlsolver_amd import ops_mcpg_tsp as mops
n, m, M, R = 800, 4694, 512, 128
arr = np.asarray(generate_gnm(n, m, 14), dtype=np.int64)
data = amcpg.make_data(n, arr[:, 0], arr[:, 1], dev)
kept = PackedChains.pack((torch.rand(n, M, device=dev) < 0.5).float())
rnd = amcpg.MCPGRound(data, kept, torch.zeros(M, device=dev), M, R, 5)
probs = torch.full((n,), 0.5, device=dev)
rnd.step(probs); torch.cuda.synchronize()


def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


val = rnd.value.detach().float().contiguous()
print("bit sums on the round's samples: %.1f us" % t(lambda: mops.mcpg_value_bit_sums(rnd.samples, val)))
rand_words = torch.randint(-2**63, 2**63 - 1, rnd.samples.words.shape, dtype=torch.int64, device=dev)
print("bit sums on random words:        %.1f us" % t(lambda: mops.mcpg_value_bit_sums(PackedChains(rand_words, M * R), val)))
print("value stats", float(val.abs().max()), float((val == 0).float().mean()))

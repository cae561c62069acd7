"""Dev: K12 / K13 by batch size (the 2^16-tour BASELINE row is a 13 us launch)."""
import sys, torch
sys.path.insert(0, ".")
from rlsolver_amd.graph import tsp_tables, generate_tsp_coords
from rlsolver_amd import ops_mcpg_tsp as mops
dev = torch.device("cuda:0")
N = 100
dist, near, rnd = tsp_tables(generate_tsp_coords(N, 100), K=20)
d = torch.from_numpy(dist).to(dev)


def t(fn, it=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for B in (1 << 16, 1 << 18, 1 << 20):
    perms = mops.rand_perms(B, N, 3, dev)
    sel = torch.roll(perms, 7, 1).contiguous()
    u12 = t(lambda: mops.tsp_tour_length(d, perms))
    u13 = t(lambda: mops.tsp_swap_delta_all(d, perms, sel, 0.5))
    print("B=2^%d: K12 %.1f us (%.3f of 8 TB/s) | K13 %.1f us (%.3f)" % (B.bit_length() - 1, u12, B * (8 * N + 4) / u12 / 8e6, u13, B * 29 * N / u13 / 8e6))

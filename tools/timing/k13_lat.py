"""K13 at TSP-100 / 2^16: partners drawn in the kernel (21N bytes per tour) vs the selected tensor given (29N)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rlsolver_amd import ops_mcpg_tsp as mops
from rlsolver_amd.graph import generate_tsp_coords, tsp_tables
dev = torch.device("cuda:0")
def t_us(fn, n=50):
    """n calls captured in ONE hipGraph and replayed: the kernels back to back, no Python / dispatcher time between them (a call of
    the 13-argument op costs the host more than the 25-45 us the kernel runs)."""
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n): fn(i)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for N, B in ((100, 1 << 16), (100, 1 << 18), (52, 1 << 16), (200, 1 << 15)):
    dist, near, rnd = tsp_tables(generate_tsp_coords(N, 100), K=20)
    d = torch.from_numpy(dist).to(dev)
    near32, rnd32 = torch.from_numpy(near.astype("int32")).to(dev), torch.from_numpy(rnd.astype("int32")).to(dev)
    tab8 = mops.tsp_tables8(near32, rnd32)
    perms = mops.rand_perms(B, N, 3, dev)
    sel = torch.roll(perms, 7, 1).contiguous()
    a = t_us(lambda i: mops.tsp_swap_delta_all(d, perms, None, 0.5, nearest=near32, random=rnd32, near_threshold=20 / 21, seed=i, tables8=tab8))
    b = t_us(lambda i: mops.tsp_swap_delta_all(d, perms, sel, 0.5))
    c = t_us(lambda i: mops.tsp_tour_length(d, perms))
    if N == 100 and B == 1 << 16:      # what the two-op form of rounds 1-5 paid before its kernel: ISCO_TSP.draw_partners (torch RNG + gathers)
        from rlsolver_amd.envs.env_ISCO import ISCO_TSP
        params = {"num_nodes": N, "distance": d, "nearest_indices": torch.from_numpy(near).to(dev), "random_indices": torch.from_numpy(rnd).to(dev)}
        env = ISCO_TSP(params, batch_size=B, K=20, device=dev)
        e = t_us(lambda i: env.draw_partners(perms), 10)
        print(f"N={N} B={B}: ISCO_TSP.draw_partners (3 torch draws + 2 gathers + where) {e:.1f} us -> the two-op opt_2 cost {e + b:.1f} us, the one-kernel opt_2 {a:.1f}")
    print(f"N={N} B={B}: K13 draw {a:.1f} us ({B * 21 * N / a / 8e6:.3f} of 8 TB/s on 21N) | selected given {b:.1f} us ({B * 29 * N / b / 8e6:.3f} on 29N) | K12 {c:.1f} us ({B * (8 * N + 4) / c / 8e6:.3f})")

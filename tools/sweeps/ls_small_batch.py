"""local_search_inplace at small batches: the fused kernel (one workgroup per tile) against the round kernels (a tile's noise
passes split over several workgroups).  `python tools/sweeps/ls_small_batch.py`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd.envs.env_L2A import EnvMaxcut
from rlsolver_amd.graph import generate_ba, generate_gnm


def timeit(f, n=10):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return float(np.median(ts))


def main():
    dev = torch.device("cuda:0")
    for name, n, mg in (("G22-sized", 2000, generate_gnm(2000, 19990, 22)), ("BA-2000 m=4", 2000, generate_ba(2000, 4, 3)),
                        ("G14-sized", 800, generate_gnm(800, 4694, 14)), ("G(3008, 9000)", 3008, generate_gnm(3008, 9000, 1)),
                        ("BA-1e4 m=5", 10000, generate_ba(10000, 5, 5))):
        env = EnvMaxcut(mygraph=mg, device=dev, num_nodes=n)
        for B in (256, 4096, 8192, 16384, 65536):
            torch.manual_seed(0)
            xs = env.generate_xs_randomly(B)
            vs = env.calculate_obj_values(xs)
            out = []
            for fused in (True, False):
                env.force_ls_rounds, env.force_ls_fused = not fused, fused
                out.append(timeit(lambda: env.local_search_inplace(xs, vs, num_iters=8, num_spin=8, noise_std=0.3)))
            print(f"{name} B={B}: fused {out[0]:.0f} us, round kernels {out[1]:.0f} us")


if __name__ == "__main__":
    main()

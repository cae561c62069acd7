"""K3 (delta_all) and the local-search weights on a hub graph (BA n = 10^4, m = 5) next to a flat graph of the same size: what
the hub groups cost the lane = node kernel.  `python tools/timing/k3_ba.py [log2 B]`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import ops
from rlsolver_amd.graph import build_csr, generate_ba, generate_gnm


def timeit(f, n=10):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return float(np.median(ts))


def main():
    dev = torch.device("cuda:0")
    B = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 16)
    n = 10000
    for name, mg in (("BA m=5", generate_ba(n, 5, 5)), ("G(n, 49975)", generate_gnm(n, 49975, 5))):
        csr = build_csr(mg, n, False)
        g = ops.DeviceGraph(csr, dev)
        xs = (torch.rand(B, n, device=dev) < 0.5)
        out = torch.empty((B, n), dtype=torch.int32, device=dev)
        t3 = timeit(lambda: ops.maxcut_delta_all(g, xs, out=out))
        tw = timeit(lambda: ops.maxcut_ls_weights(g, xs, 1))
        deg = np.diff(csr.rowptr)
        md = [int(deg[i:i + 64].max()) for i in range(0, n, 64)]
        print(f"{name}: max degree {deg.max()}, group rounds sum {sum(md)} max {max(md)}; K3 {t3:.0f} us ({B * n * 5 / t3 / 1e6 / 8000:.3f} of 8 TB/s), "
              f"ls_weights {tw:.0f} us")


if __name__ == "__main__":
    main()

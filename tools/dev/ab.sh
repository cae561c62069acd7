export PYTHONPATH=.
python tools/timing/k1_tile32.py 2>&1 | grep "N=\|TILE"
python - <<'PY'
import torch, numpy as np
from rlsolver_amd import graph as G, ops
from rlsolver_amd.graph import build_csr
dev = torch.device("cuda:0")
def t(f, reps=10):
    f(); f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for n, m in ((20000, 40000), (24000, 48000), (40000, 80000)):
    dg = ops.DeviceGraph(build_csr(G.generate_gnm(n, m, 22), num_nodes=n), dev)
    B = 1 << 14
    xs = ops.rand_spins(B, n, 3, dev)
    vs = ops.maxcut_obj(dg, xs)
    mask = (torch.rand((B, n), device=dev) < 0.01)
    us = t(lambda: ops.maxcut_propose_accept(dg, xs, mask, vs))
    print(f"K6 byte mask N={n} B=2^14: {us:.1f} us ({3 * B * n / us / 1e6 / 8:.3f} of 8 TB/s at 3N bytes per env)")
PY

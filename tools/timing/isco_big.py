"""ISCO_maxcut.step past the LDS rows (N > ~15 900: the f32 rows in the step's scratch, a workgroup per sample) beside the sizes below it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd.envs.env_ISCO_maxcut import ISCO_maxcut
from rlsolver_amd.graph import generate_gnm
dev = torch.device("cuda:0")


def t(f, reps=5):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for n, m in ((10000, 9999), (15000, 30000), (20000, 40000), (44000, 88000), (80000, 160000)):
    g = np.asarray(generate_gnm(n, m, 7), dtype=np.int64)
    row = []
    for B in (1, 64, 512, 4096):
        s = ISCO_maxcut({"num_nodes": n, "num_edges": len(g), "edge_from": torch.from_numpy(g[:, 0].copy()).to(dev),
                         "edge_to": torch.from_numpy(g[:, 1].copy()).to(dev)}, batch_size=B, device=dev)
        x = s.random_gen_init_sample()
        pl = torch.full((B,), 12, dtype=torch.int64, device=dev)
        row.append(f"B={B}: {t(lambda: s.step(x, pl, 0.5)):9.1f} us")
    print(f"N={n} E={m} path 12:  " + "   ".join(row), flush=True)

"""ISCO_maxcut.step: wave per sample vs workgroup per sample (RLS_ISCO_FORCE_WG = 0 | 1) by graph size and batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import _abi
from rlsolver_amd.envs.env_ISCO_maxcut import ISCO_maxcut
from rlsolver_amd.graph import generate_gnm
dev = torch.device("cuda:0")


def t(f, reps=4):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for n, m in ((800, 4694), (2000, 19990), (4000, 16000), (6000, 24000), (10000, 9999), (15000, 30000)):
    g = np.asarray(generate_gnm(n, m, 7), dtype=np.int64)
    for B in (256, 1024, 4096, 16384):
        s = ISCO_maxcut({"num_nodes": n, "num_edges": len(g), "edge_from": torch.from_numpy(g[:, 0].copy()).to(dev),
                         "edge_to": torch.from_numpy(g[:, 1].copy()).to(dev)}, batch_size=B, device=dev)
        x = s.random_gen_init_sample()
        pl = torch.full((B,), 12, dtype=torch.int64, device=dev)
        row = []
        for force in (0, 1):
            _abi.tuning_set("RLS_ISCO_FORCE_WG", force)
            row.append(f"{'wg  ' if force else 'wave'} {t(lambda: s.step(x, pl, 0.5)):9.1f}")
        _abi.tuning_unset("RLS_ISCO_FORCE_WG")
        print(f"N={n} B={B}: " + " | ".join(row) + " us", flush=True)

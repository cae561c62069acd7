"""K11 (dense QUBO coordinate search + value) over problem sizes: us per call and TFLOP/s.  `python tools/sweeps/qubo_n_sweep.py`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd.methods import MCPG_qubo as mq

dev = torch.device("cuda:0")


def t_us(f, n=3):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for C in (4096, 32768):
    row = []
    for n in (100, 333, 500, 1000, 1001, 2000, 3000, 5000):
        Q = torch.randn(n, n, device=dev).mul(20).round()
        Q = Q + Q.T
        x = (torch.rand((n, C), device=dev) < 0.5).float()
        try:
            t = t_us(lambda: mq.qubo_local_search_value(Q, x, 1, False))
            row.append(f"n={n}: {t:7.0f} us ({(2.0 * n * n * C * 2) / t / 1e6:5.1f} TFLOP/s)")
        except Exception as e:   # noqa
            row.append(f"n={n}: {type(e).__name__} {str(e)[:60]}")
    print(f"C = {C}:  " + "   ".join(row))

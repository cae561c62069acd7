import torch, numpy as np
from rlsolver_amd.graph import generate_gnm
from rlsolver_amd.envs.spinsystem import SpinSystem, ECO_PECO_OBSERVABLES, S2V_OBSERVABLES
dev = torch.device('cuda:0')
n, m = 2000, 19990
mg = generate_gnm(n, m, 22)
for name, obs_set in (("ECO 7 rows", ECO_PECO_OBSERVABLES), ("S2V 1 row", S2V_OBSERVABLES)):
    for B in (4096, 16384):
        env = SpinSystem(mg, n, B, max_steps=10 ** 6, observables=obs_set, device=dev, include_adjacency=False)
        env.reset()
        acts = [torch.randint(0, n, (B,), device=dev) for _ in range(8)]
        for i in range(5): env.step(acts[i % 8])
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        K = 50
        for i in range(K): env.step(acts[i % 8])
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / K * 1e3
        R = env.state.shape[1]
        print(f"{name:11s} B={B:6d} rows={R}: {us:8.1f} us/step  {B/us*1e6:.3g} env-steps/s  state bytes/env {R*4*n}  -> {B*R*4*n/us/1e6:.2f} TB/s if the whole state moved once")

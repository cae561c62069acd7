"""The MaxCut entry points over the Gset node counts (800 ... 20 000) and past them (24 000 ... 44 000) at two batch sizes: us per
call and ns per (env, node).  Looks for cliffs where a kernel form changes (fused local search <= ~7100 nodes, round kernels and
the 64-env tile <= ~15 500 / 20 224, half tiles of 32 envs <= 39 936 / 40 448, one env per wave beyond).
`python tools/sweeps/n_sweep.py`."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import ops
from rlsolver_amd.envs.env_L2A import EnvMaxcut
from rlsolver_amd.envs.env_PPO import EnvMaxcut as Gym
from rlsolver_amd.graph import generate_gnm

dev = torch.device("cuda:0")
NS = (800, 1000, 2000, 3000, 5000, 7000, 8000, 9000, 10000, 14000, 20000, 20240, 24000, 32000, 39936, 44000)


def t_us(f, n=4):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in (4096, 16384):
    print(f"B = {B}; graphs G(N, 3 N); us per call [ns per (env, node)]")
    print(f"{'':30s}" + "".join(f"{n:>15d}" for n in NS))
    rows = {}
    for n in NS:
        mg = generate_gnm(n, 3 * n, 7)
        env = EnvMaxcut(mygraph=mg, device=dev, num_nodes=n)
        g = env.graph
        x = torch.rand(B, n, device=dev) < 0.5
        v = ops.maxcut_obj(g, x)
        d = torch.empty((B, n), dtype=torch.int32, device=dev)
        m = torch.rand(B, n, device=dev) < 0.002
        gym = Gym(types.SimpleNamespace(num_nodes=n, num_envs=B, num_steps=10 ** 9), mygraph=mg, device=dev)
        gym.reset()
        a = torch.randint(0, n, (B,), device=dev)
        slot = torch.empty((B, n), dtype=torch.float32, device=dev)
        for name, f in (("K1 obj", lambda: ops.maxcut_obj(g, x)), ("K3 delta_all", lambda: ops.maxcut_delta_all(g, x, out=d)),
                        ("K5 sweep", lambda: ops.maxcut_greedy_sweep(g, x, v)), ("K6 propose", lambda: ops.maxcut_propose_accept(g, x, m, v)),
                        ("local_search_inplace", lambda: env.local_search_inplace(x, v, num_iters=8, num_spin=8)),
                        ("gym step(out=slot) f32", lambda: gym.step(a, out=slot))):
            try:
                rows.setdefault(name, []).append(t_us(f))
            except Exception as e:   # noqa
                rows.setdefault(name, []).append(float("nan")); print("   ", name, n, type(e).__name__, str(e)[:100])
        del env, g, x, v, d, m, gym, slot
        torch.cuda.empty_cache()
    for name, ts in rows.items():
        print(f"{name:30s}" + "".join(f"{t:8.0f} [{t * 1e3 / (B * n):4.2f}]" for t, n in zip(ts, NS)))

"""MCPG / ISCO / spin-env entry points over the Gset node counts: us per call.  `python tools/sweeps/n_sweep2.py`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd.envs.env_ISCO_maxcut import ISCO_maxcut
from rlsolver_amd.envs.spinsystem import SpinSystem
from rlsolver_amd.graph import generate_gnm
from rlsolver_amd.methods import MCPG as amcpg
from rlsolver_amd.ops_mcpg_tsp import PackedChains

dev = torch.device("cuda:0")
NS = (2000, 5000, 8000, 10000, 14000, 16000, 20000)
C = 16384


def t_us(f, n=3):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


rows = {}
for n in NS:
    mg = generate_gnm(n, 3 * n, 7)
    arr = np.asarray(mg, dtype=np.int64)
    data = amcpg.make_data(n, arr[:, 0], arr[:, 1], dev)
    probs = torch.full((n,), 0.5, device=dev)
    xs = (torch.rand((n, C), device=dev) < 0.5).float()
    kept = PackedChains.pack((torch.rand((n, C // 128), device=dev) < 0.5).float())
    out = PackedChains.empty(n, C, dev)
    T = n // 10
    rnd = amcpg.MCPGRound(data, (torch.rand((n, 128), device=dev) < 0.5).float(), torch.zeros(128, device=dev), 128, C // 128, 2)
    spin = SpinSystem([(a, b, 1) for a, b, _ in mg], n, 4096, max_steps=10 ** 6, device=dev, include_adjacency=False)
    act = torch.randint(0, n, (4096,), device=dev)
    tests = (("MCPG sampler_func f32 (num_ls=2)", lambda: amcpg.sampler_func(data, xs, 2, C // 128, 128, dev)),
             ("MCPG metro_sampling f32 (T=N/10)", lambda: amcpg.metro_sampling(probs, xs, T, dev)),
             ("MCPG metro_sampling_packed", lambda: amcpg.metro_sampling_packed(probs, kept, T, num_chains=C, out=out)),
             ("MCPG round on device (num_ls=2)", lambda: rnd.step(probs)),
             ("spin step + obs rows (4096 envs)", lambda: spin.step(act)))
    for name, f in tests:
        try:
            rows.setdefault(name, []).append(t_us(f))
        except Exception as e:   # noqa
            rows.setdefault(name, []).append(float("nan")); print("   ", name, n, type(e).__name__, str(e)[:120])
    del data, xs, kept, out, rnd, spin
    torch.cuda.empty_cache()
print(f"{C} chains; us per call at N = " + "".join(f"{n:>9d}" for n in NS))
for name, ts in rows.items():
    print(f"{name:36s}" + "".join(f"{t:9.0f}" for t in ts))

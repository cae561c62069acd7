"""Rows that are not 16-byte multiples: every batched MaxCut entry point at N = 2000 next to N = 1999 / 2001 / 2004 / 2008 (same
density), two batch sizes.  Looks for slow unaligned forms.  `python tools/sweeps/odd_rows.py`."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlsolver_amd import ops
from rlsolver_amd.envs.env_L2A import EnvMaxcut
from rlsolver_amd.envs.env_PPO import EnvMaxcut as Gym
from rlsolver_amd.graph import generate_gnm

dev = torch.device("cuda:0")


def t_us(f, n=6):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in (4096, 65536):
    print(f"B = {B}: us per call at N =      2000    1999    2001    2004    2008")
    rows = {}
    for n in (2000, 1999, 2001, 2004, 2008):
        mg = generate_gnm(n, 10 * n, 22)
        env = EnvMaxcut(mygraph=mg, device=dev, num_nodes=n)
        g = env.graph
        x = torch.rand(B, n, device=dev) < 0.5
        v = ops.maxcut_obj(g, x)
        d = torch.empty((B, n), dtype=torch.int32, device=dev)
        m = torch.rand(B, n, device=dev) < 0.004
        gym = Gym(types.SimpleNamespace(num_nodes=n, num_envs=B, num_steps=10 ** 9), mygraph=mg, device=dev)
        gym.reset()
        a = torch.randint(0, n, (B,), device=dev)
        slot = torch.empty((B, n), dtype=torch.float32, device=dev)
        u8 = ops.maxcut_step_launcher(g, x.clone(), torch.empty_like(x), v.to(torch.int32).clone()) if hasattr(ops, "maxcut_step_launcher") and False else None
        for name, f in (("K1 obj", lambda: ops.maxcut_obj(g, x)), ("K2 cutdeg", lambda: ops.maxcut_node_cutdeg(g, x)),
                        ("K3 delta_all", lambda: ops.maxcut_delta_all(g, x, out=d)), ("K5 sweep", lambda: ops.maxcut_greedy_sweep(g, x, v)),
                        ("K6 propose", lambda: ops.maxcut_propose_accept(g, x, m, v)), ("ls_weights", lambda: ops.maxcut_ls_weights(g, x, 1)),
                        ("local_search_inplace", lambda: env.local_search_inplace(x, v, num_iters=8, num_spin=8)),
                        ("K14 rand_spins", lambda: ops.rand_spins(B, n, 3, dev)),
                        ("gym step(out=slot) f32", lambda: gym.step(a, out=slot)), ("gym step in place", lambda: gym.step(a))):
            rows.setdefault(name, []).append(t_us(f))
    for name, ts in rows.items():
        flag = "   <-- unaligned rows much slower" if max(ts[1:]) > 1.6 * ts[0] else ""
        print(f"{name:36s}" + "".join(f"{t:8.0f}" for t in ts) + flag)

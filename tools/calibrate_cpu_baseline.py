#!/usr/bin/env python3
"""Calibrate the "ref-shaped" CPU baseline (oracle/oracle_torch.py) against the IMPORTED reference in the build
container (the reference cannot travel to the GPU box): both run env_PPO.EnvMaxcut.step on the same graph / envs /
actions on this container's cores.  Prints one JSON line; the numbers are quoted in BASELINE.md.

    PYTHONDONTWRITEBYTECODE=1 python tools/calibrate_cpu_baseline.py
"""
import json, os, sys, time, types
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
import numpy as np
import torch as th
from rlsolver.envs.env_PPO import EnvMaxcut as RefEnv
from oracle.oracle_torch import PPOEnvRefShaped
from rlsolver_amd.graph import generate_gnm

out = {"cores": os.cpu_count(), "torch_threads": th.get_num_threads()}
for tag, n, m, B, steps in (("G14-sized G(800,4694), 256 envs", 800, 4694, 256, 40), ("G22-sized G(2000,19990), 1024 envs", 2000, 19990, 1024, 6)):
    mg = generate_gnm(n, m, 14)
    rng = np.random.RandomState(0)
    x0 = rng.randint(0, 2, size=(B, n)).astype(bool)
    acts = th.from_numpy(rng.randint(0, n, size=(steps, B)).astype(np.int64))
    ref = RefEnv(types.SimpleNamespace(num_nodes=n, num_envs=B, num_steps=10 ** 9), mygraph=mg, device=th.device("cpu"))
    ref.xs = th.from_numpy(x0).float()
    ref.last_reward = ref.calculate_obj_values().float()
    mine = PPOEnvRefShaped(np.asarray(mg, dtype=np.int64), n, B, 10 ** 9)
    mine.reset_to(x0)
    ref.step(acts[0]); mine.step(acts[0])
    t0 = time.perf_counter()
    for t in range(1, steps):
        r = ref.step(acts[t])
    t_ref = (time.perf_counter() - t0) / (steps - 1)
    t0 = time.perf_counter()
    for t in range(1, steps):
        q = mine.step(acts[t])
    t_mine = (time.perf_counter() - t0) / (steps - 1)
    assert th.equal(r[1], q[1]) and th.equal(r[3], q[3])
    out[tag] = {"reference_env_steps_per_s": B / t_ref, "ref_shaped_oracle_env_steps_per_s": B / t_mine, "ratio": t_ref / t_mine}
print(json.dumps(out))

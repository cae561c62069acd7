"""CPU ORACLE, "ref-shaped" form -- TEST / BASELINE INFRASTRUCTURE, NOT PRODUCT CODE.

SURVEY.md section 8d asks for the CPU baseline in two forms.  oracle.c is the "port" (C + OpenMP, flip + full
re-evaluation).  This module is form (i): the SAME op chain as the reference's env_PPO.EnvMaxcut written with
torch-CPU ops in the reference's own shape, so that its timing on the GPU box's host cores stands in for the
reference's CPU env path (which cannot travel):
    rlsolver/envs/env_PPO.py:92-106   step: a Python loop over the envs doing two index ops each, then
    rlsolver/envs/env_PPO.py:108-121  calculate_obj_values: three int64 [B, E'] index tensors, two advanced-index
                                      gathers, XOR, sum.
Pinned against tests/golden/env_ppo.npz (tests/test_oracle_golden.py).  Only tests/ and bench.py's cpu_baseline leg
import it.
"""
from __future__ import annotations

import numpy as np
import torch as th


class PPOEnvRefShaped:
    def __init__(self, graph_arr, num_nodes: int, num_envs: int, num_steps: int, if_bidirectional: bool = False):
        g = np.asarray(graph_arr, dtype=np.int64).reshape(-1, 3)
        u, v = g[:, 0], g[:, 1]
        if if_bidirectional:
            u, v = np.concatenate([u, v]), np.concatenate([v, u])
        order = np.lexsort((v, u))                                   # n0-major, n1 ascending: the env's edge order
        self.n0_ids = th.from_numpy(u[order].copy())[None, :]
        self.n1_ids = th.from_numpy(v[order].copy())[None, :]
        self.sim_ids = th.zeros(self.n0_ids.shape[1], dtype=th.long)[None, :]
        self.if_bidirectional = if_bidirectional
        self.num_nodes, self.num_envs, self.num_steps = num_nodes, num_envs, num_steps
        self.action_count = 0
        self.xs = None
        self.last_reward = None

    def reset_to(self, xs_bool):
        self.xs = th.as_tensor(np.asarray(xs_bool)).to(th.float).clone()
        self.last_reward = self.calculate_obj_values().to(th.float)
        return self.xs

    def calculate_obj_values(self):
        xs = self.xs > 0
        num_sims = xs.shape[0]
        if num_sims != self.sim_ids.shape[0]:                        # the three cached [B, E'] index tensors
            self.n0_ids = self.n0_ids[0].repeat(num_sims, 1)
            self.n1_ids = self.n1_ids[0].repeat(num_sims, 1)
            self.sim_ids = self.sim_ids[0:1] + th.arange(num_sims, dtype=th.long)[:, None]
        values = (xs[self.sim_ids, self.n0_ids] ^ xs[self.sim_ids, self.n1_ids]).sum(1)
        return values // 2 if self.if_bidirectional else values

    def step(self, action):
        self.action_count += 1
        for n in range(self.num_envs):                               # env_PPO.py:94-95: O(B) tiny ops
            self.xs[n, action[n]] = th.logical_not(self.xs[n, action[n]])
        cur = self.calculate_obj_values().to(th.float)
        reward = cur - self.last_reward
        self.last_reward = cur
        if self.action_count == self.num_steps:
            self.action_count = 0
            done = th.ones(self.num_envs)
        else:
            done = th.zeros(self.num_envs)
        return self.xs, reward, done, cur

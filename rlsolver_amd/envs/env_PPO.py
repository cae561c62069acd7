"""EnvMaxcut, gym flavour -- drop-in for rlsolver/envs/env_PPO.py:63-126.

``reset() -> xs f32 [B, N]``; ``step(action) -> (xs, reward, done, cur)`` all float32, xs being the
env's own storage mutated in place, exactly like the reference.  One HIP kernel per step
(flip + cut gain via the action node's CSR row) instead of a Python loop over envs followed by a
full objective evaluation.
"""
from __future__ import annotations

from typing import Optional

import torch as th

from .. import ops
from ..graph import MyGraph, build_csr
from .env_L2A import _seed_from_torch

TEN = th.Tensor


class EnvMaxcut:
    def __init__(self, args, mygraph: MyGraph = (), device=th.device('cpu'), if_bidirectional: bool = False):
        self.device = th.device(device)
        if self.device.type != 'cuda':
            raise TypeError(f"rlsolver_amd.EnvMaxcut needs a HIP device (got {self.device}); there is no CPU path")
        self.int_type = th.long
        self.if_bidirectional = if_bidirectional
        self.num_nodes = args.num_nodes
        self.num_envs = args.num_envs
        self.xs = None
        self.action_count = 0
        self.last_reward = None
        self.num_steps = args.num_steps
        self.num_edges = len(mygraph)
        csr = build_csr(mygraph, num_nodes=self.num_nodes, if_bidirectional=if_bidirectional)
        self.graph = ops.DeviceGraph(csr, self.device)
        self.n0_ids = self.graph.eu.to(th.long)[None, :]
        self.n1_ids = self.graph.ev.to(th.long)[None, :]
        B = self.num_envs
        self._obj = th.zeros(B, dtype=th.int32, device=self.device)
        self._reward = th.zeros(B, dtype=th.float32, device=self.device)
        self._done = th.zeros(B, dtype=th.float32, device=self.device)

    def reset(self):
        xs = self.generate_xs_randomly(num_sims=self.num_envs)
        self.xs = xs.to(th.float)
        self._obj = ops.maxcut_obj(self.graph, self.xs).to(th.int32)
        self.last_reward = self._obj.to(th.float)
        return self.xs

    def step(self, action, out: Optional[TEN] = None):
        """env_PPO.py:92-106.  ``out`` (f32 [B, N]) makes the step emit the next state there (the
        rollout-buffer form); by default xs is updated in place like the reference."""
        self.action_count += 1
        action = action.to(device=self.device, dtype=th.int64).contiguous()
        cur = th.empty(self.num_envs, dtype=th.float32, device=self.device)
        reward = th.empty(self.num_envs, dtype=th.float32, device=self.device)
        if self.action_count == self.num_steps:
            self.action_count = 0
            done_value = 1.0
        else:
            done_value = 0.0
        next_done = th.empty(self.num_envs, dtype=th.float32, device=self.device)
        dst = self.xs if out is None else out
        ops.maxcut_step(self.graph, self.xs, dst, action, self._obj, reward, cur, next_done, done_value)
        self.xs = dst
        self.last_reward = cur
        return self.xs, reward, next_done, cur

    def calculate_obj_values(self, if_sum: bool = True) -> TEN:
        """env_PPO.py:108-121 (objective of the env's own state)."""
        if if_sum:
            return ops.maxcut_obj(self.graph, self.xs)
        values = ops.maxcut_edge_cut_mask(self.graph, self.xs > 0)
        if self.if_bidirectional:
            values = values // 2
        return values

    def generate_xs_randomly(self, num_sims):
        return ops.rand_spins(num_sims, self.num_nodes, _seed_from_torch(), self.device)

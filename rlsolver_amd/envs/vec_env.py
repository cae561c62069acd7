"""MaxCut as an elegantrl-style vectorised env (the contract of elegantrl/train/config.py:118-137 and
elegantrl/agents/AgentBase.py:141-170):

    reset() -> (state f32 [num_envs, state_dim], info)
    step(action) -> (state, reward f32 [num_envs], terminal bool [num_envs], truncate bool [num_envs], info)

with attributes env_name, num_envs, max_step, state_dim (= N), action_dim (= N), if_discrete (True).
Discrete actions arrive as int32 [num_envs] (AgentBase.py:149).  State is the spin vector as f32,
updated in place by the K4 kernel; an episode truncates every max_step steps (no auto-reset, as
the CO envs of the reference: the caller decides when to reset).
"""
from __future__ import annotations

import types

import torch as th

from .env_PPO import EnvMaxcut as _GymEnv


class MaxcutVecEnv:
    def __init__(self, mygraph, num_nodes: int, num_envs: int, max_step: int = 12345, gpu_id: int = 0,
                 if_bidirectional: bool = False, env_name: str = "maxcut", env_offset: int = 0, seed=None):
        """``env_offset`` / ``seed``: rlsolver_amd/seeding.py -- ``num_envs`` is this rank's share of a sharded batch."""
        if gpu_id < 0:
            raise TypeError("MaxcutVecEnv needs a HIP device (gpu_id >= 0); there is no CPU path")
        self.device = th.device(f"cuda:{gpu_id}")
        args = types.SimpleNamespace(num_nodes=num_nodes, num_envs=num_envs, num_steps=max_step)
        self._env = _GymEnv(args, mygraph=mygraph, device=self.device, if_bidirectional=if_bidirectional, env_offset=env_offset, seed=seed)
        self.env_name = env_name
        self.num_envs = num_envs
        self.max_step = max_step
        self.state_dim = num_nodes
        self.action_dim = num_nodes
        self.if_discrete = True

    def reset(self):
        state = self._env.reset()
        self._env.action_count = 0
        return state, {}

    def step(self, action):
        state, reward, done, cur = self._env.step(action.reshape(self.num_envs))
        truncate = done.bool()
        terminal = th.zeros_like(truncate)
        return state, reward, terminal, truncate, {"obj": cur}

    @property
    def best_obj(self):
        return self._env.last_reward.max()

    def state_dict(self):
        return self._env.state_dict()

    def load_state_dict(self, d):
        self._env.load_state_dict(d)

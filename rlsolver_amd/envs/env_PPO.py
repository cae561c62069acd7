"""EnvMaxcut, gym flavour -- drop-in for rlsolver/envs/env_PPO.py:63-126.

``reset() -> xs f32 [B, N]``; ``step(action) -> (xs, reward, done, cur)`` all float32, xs being the
env's own storage mutated in place, exactly like the reference.  One HIP kernel per step
(flip + cut gain via the action node's CSR row) instead of a Python loop over envs followed by a
full objective evaluation.
"""
from __future__ import annotations

from typing import Optional

import torch as th

from .. import ops, torch_ops
from ..graph import MyGraph, build_adjacency_indies, build_csr  # noqa: F401  (build_adjacency_indies: env_PPO.py:26-47 has its own copy)
from ..seeding import Sharded

TEN = th.Tensor


class EnvMaxcut(Sharded):
    def __init__(self, args, mygraph: MyGraph = (), device=th.device('cpu'), if_bidirectional: bool = False,
                 spin_dtype=th.float32, reuse_buffers: bool = False, env_offset: int = 0, seed: Optional[int] = None):
        """``spin_dtype``: torch.float32 is the reference's surface (xs cast to float, env_PPO.py:87); torch.bool keeps
        the state at 1 byte per spin (4x less HBM traffic per step) for callers that cast where they consume it.
        ``reuse_buffers``: reward / done / cur are written into two alternating pre-allocated sets instead of fresh
        tensors (no allocator call on the step path); a returned tensor then stays valid for ONE further step.
        ``env_offset`` / ``seed``: rlsolver_amd/seeding.py -- ``args.num_envs`` is this rank's share of a sharded batch and
        ``env_offset`` the global id of its env 0 (reset draws are keyed by the global env id; a step draws nothing)."""
        self._init_shard(env_offset, seed)
        self.device = th.device(device)
        if self.device.type != 'cuda':
            raise TypeError(f"rlsolver_amd.EnvMaxcut needs a HIP device (got {self.device}); there is no CPU path")
        self.int_type = th.long
        self.if_bidirectional = if_bidirectional
        self.num_nodes = args.num_nodes
        self.num_envs = args.num_envs
        self._xs = None
        self._stale = False
        self._carry = None
        self.action_count = 0
        self.last_reward = None
        self.num_steps = args.num_steps
        self.num_edges = len(mygraph)
        csr = build_csr(mygraph, num_nodes=self.num_nodes, if_bidirectional=if_bidirectional)
        self.graph = ops.DeviceGraph(csr, self.device)
        self._gh = torch_ops.graph_handle(self.graph)
        self._step_op = torch_ops.ops.maxcut_step       # native custom op: the launch-bound path of this class
        self.n0_ids = self.graph.eu.to(th.long)[None, :]
        self.n1_ids = self.graph.ev.to(th.long)[None, :]
        B = self.num_envs
        if spin_dtype not in (th.float32, th.bool):
            raise TypeError("spin_dtype must be torch.float32 (reference surface) or torch.bool")
        self.spin_dtype = spin_dtype
        self._obj = th.zeros(B, dtype=th.int32, device=self.device)
        self._sets = [tuple(th.zeros(B, dtype=th.float32, device=self.device) for _ in range(3)) for _ in range(2)] \
            if reuse_buffers else None
        self._flip = 0

    # The reference recomputes the cut from ``self.xs`` on every step (env_PPO.py:96-98); this class keeps it incrementally
    # (``_obj`` + the action node's gain).  A caller that changes the state behind the env's back must say so:
    #   * ``env.xs = tensor`` (assignment) is seen by the property below and re-synchronises by itself at the next step;
    #   * an IN-PLACE edit (``env.xs[3, 7] = 1``) cannot be seen: call ``env.resync()`` after it.
    # Either way the next step's reward is the reference's: new cut - last_reward, the edit's own change included.
    @property
    def xs(self):
        return self._xs

    @xs.setter
    def xs(self, value):
        self._xs = value
        self._stale = value is not None

    def resync(self):
        """Recompute the incremental objective from ``self.xs`` (call after editing ``env.xs`` in place between steps).  The
        cut change of the edit is carried into the next step's reward, as ``cur_reward - self.last_reward`` of
        env_PPO.py:97-98 would hold it."""
        self._stale = False
        if self._xs is None:
            return self
        if self._xs.dtype != self.spin_dtype or self._xs.device != self.device or not self._xs.is_contiguous():
            self._xs = self._xs.to(device=self.device, dtype=self.spin_dtype).contiguous()
        new = ops.maxcut_obj(self.graph, self._xs).to(th.int32)
        if self.last_reward is not None:
            carry = new.to(th.float32) - self.last_reward
            self._carry = carry if self._carry is None else self._carry + carry
        self._obj.copy_(new)
        return self

    def reset(self):
        xs = self.generate_xs_randomly(num_sims=self.num_envs)
        self._xs = xs.to(self.spin_dtype)
        self._stale, self._carry = False, None
        self._obj.copy_(ops.maxcut_obj(self.graph, self._xs))
        self.last_reward = self._obj.to(th.float)
        return self._xs

    def step(self, action, out: Optional[TEN] = None):
        """env_PPO.py:92-106.  ``out`` (f32 [B, N]) makes the step emit the next state there (the
        rollout-buffer form); by default xs is updated in place like the reference."""
        self.action_count += 1
        if action.dtype != th.int64 or action.device != self.device or not action.is_contiguous():
            action = action.to(device=self.device, dtype=th.int64).contiguous()
        if self._sets is not None:
            reward, next_done, cur = self._sets[self._flip]
            self._flip ^= 1
        else:
            reward, next_done, cur = (th.empty(self.num_envs, dtype=th.float32, device=self.device) for _ in range(3))
        if self.action_count == self.num_steps:
            self.action_count = 0
            done_value = 1.0
        else:
            done_value = 0.0
        if self._stale:
            self.resync()
        dst = self._xs if out is None else out
        self._step_op(self._gh, self._xs, dst, action, self._obj, reward, cur, next_done, done_value)
        self._xs = dst
        if self._carry is not None:            # only after a resync(): the edit's own cut change belongs to this reward
            reward += self._carry
            self._carry = None
        self.last_reward = cur
        return self._xs, reward, next_done, cur

    def calculate_obj_values(self, if_sum: bool = True) -> TEN:
        """env_PPO.py:108-121 (objective of the env's own state)."""
        if if_sum:
            return ops.maxcut_obj(self.graph, self.xs)
        values = ops.maxcut_edge_cut_mask(self.graph, self.xs if self.xs.dtype == th.bool else self.xs > 0)
        if self.if_bidirectional:
            values = values // 2
        return values

    def generate_xs_randomly(self, num_sims):
        return ops.rand_spins(num_sims, self.num_nodes, self._next_seed(), self.device, env_offset=self.env_offset)

    # ---- checkpoint of the env state
    def state_dict(self):
        return {"xs": self.xs.clone(), "obj": self._obj.clone(), "action_count": self.action_count,
                "last_reward": None if self.last_reward is None else self.last_reward.clone()}

    def load_state_dict(self, d):
        self._xs = d["xs"].to(device=self.device, dtype=self.spin_dtype).clone()
        self._stale, self._carry = False, None
        self._obj.copy_(d["obj"])
        self.action_count = int(d["action_count"])
        self.last_reward = None if d["last_reward"] is None else d["last_reward"].clone()

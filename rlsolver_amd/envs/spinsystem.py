"""SpinSystem -- the S2V / ECO / PECO environment surface (SURVEY.md section 8b "S2V/ECO obs
contract", 8f item 1) for ONE shared signed-weight graph, on a HIP device.

Mirrors the batched env of rlsolver/methods/ECO_S2V/src/envs/spinsystem_PECO.py (class
SpinSystemUnbiased) and its instance-wise inference twin inference_network_env.py:

    reset(spins=None) -> obs f32 [B, R + N, N]     rows 0..R-1 observables (row 0 = spins, {0,1}
                                                   under SpinBasis.BINARY), rows R.. = adjacency
    step(action int64 [B]) -> (obs, reward f32 [B], done bool [B])
    attrs: num_envs, n_spins, max_steps, current_step, score, best_score, best_spins,
           action_space.n, observation_space.shape, get_best_cut(), get_allowed_action_states(),
           get_observation(), matrix (dense [N, N] f32, built lazily)

What runs where: the flip, the score change, the all-node gain cache ("immediate cuts available",
kept as int32 [B, N] and updated in O(deg) instead of the reference's dense [B,N,N] matmul), reward,
best tracking and every observable row are ONE HIP kernel per step (rls_spin_step).  The visited
state memory behind stag_punishment / basin_reward is the reference's own packed-bit XOR compare
(util_envs_PECO.py:228-288) in torch.  Edge weights must be integers (EdgeType.DISCRETE / unweighted).
"""
from __future__ import annotations

import ctypes as C
from enum import Enum
from typing import Optional, Sequence

import numpy as np
import torch

from .. import _abi, ops
from ..graph import build_csr
from ..ops import _ptr, _stream
from .env_L2A import _seed_from_torch


class Observable(Enum):  # ECO_S2V/src/envs/util_envs.py:40-51
    SPIN_STATE = 1
    IMMEDIATE_REWARD_AVAILABLE = 2
    TIME_SINCE_FLIP = 3
    EPISODE_TIME = 4
    TERMINATION_IMMANENCY = 5
    NUMBER_OF_GREEDY_ACTIONS_AVAILABLE = 6
    DISTANCE_FROM_BEST_SCORE = 7
    DISTANCE_FROM_BEST_STATE = 8


ECO_PECO_OBSERVABLES = [Observable.SPIN_STATE, Observable.IMMEDIATE_REWARD_AVAILABLE, Observable.TIME_SINCE_FLIP,
                        Observable.DISTANCE_FROM_BEST_SCORE, Observable.DISTANCE_FROM_BEST_STATE,
                        Observable.NUMBER_OF_GREEDY_ACTIONS_AVAILABLE, Observable.TERMINATION_IMMANENCY]
S2V_OBSERVABLES = [Observable.SPIN_STATE]


class RewardSignal(Enum):
    DENSE = 1
    BLS = 2
    SINGLE = 3
    CUSTOM_BLS = 4


class SpinBasis(Enum):
    SIGNED = 1
    BINARY = 2


_ROW_ORDER = [Observable.IMMEDIATE_REWARD_AVAILABLE, Observable.TIME_SINCE_FLIP, Observable.EPISODE_TIME,
              Observable.TERMINATION_IMMANENCY, Observable.NUMBER_OF_GREEDY_ACTIONS_AVAILABLE,
              Observable.DISTANCE_FROM_BEST_SCORE, Observable.DISTANCE_FROM_BEST_STATE]
_REWARD_MODE = {RewardSignal.DENSE: 0, RewardSignal.BLS: 1, RewardSignal.CUSTOM_BLS: 2}


class _HistoryBuffer:
    """Visited-state memory, util_envs_PECO.py:228-288: spins packed 8 per byte, exact compare
    against every earlier state of the same env."""

    def __init__(self, num_envs, device):
        self.num_envs, self.device, self.buffer = num_envs, device, None
        self._w = 2 ** torch.arange(7, -1, -1, device=device)

    def update(self, spins_signed):
        b01 = ((spins_signed + 1) / 2).to(torch.int64)
        pad = (-b01.shape[1]) % 8
        if pad:
            b01 = torch.cat([b01, torch.zeros(b01.shape[0], pad, dtype=b01.dtype, device=b01.device)], dim=1)
        packed = (b01.view(b01.shape[0], -1, 8) * self._w).sum(dim=2).to(torch.uint8)
        if self.buffer is None:
            self.buffer = packed.unsqueeze(0)
            return torch.ones(self.num_envs, dtype=torch.bool, device=self.device)
        visited = ((self.buffer ^ packed.unsqueeze(0)).sum(dim=2) == 0).any(dim=0)
        self.buffer = torch.cat([self.buffer, packed.unsqueeze(0)], dim=0)
        return ~visited


class SpinSystem:
    class _ActionSpace:
        def __init__(self, n, device):
            self.n, self.device = n, device
            self.actions = torch.arange(n, device=device)

        def sample(self, n=1):
            return self.actions[torch.randint(0, self.n, (n,), device=self.device)].tolist()

    class _ObservationSpace:
        def __init__(self, n_spins, n_observables):
            self.shape = [n_spins, n_observables]

    def __init__(self, mygraph, num_nodes: int, num_envs: int, max_steps: int = 20,
                 observables: Sequence[Observable] = ECO_PECO_OBSERVABLES,
                 reward_signal: RewardSignal = RewardSignal.DENSE, spin_basis: SpinBasis = SpinBasis.SIGNED,
                 norm_rewards: bool = False, horizon_length: Optional[int] = None,
                 stag_punishment: Optional[float] = None, basin_reward: Optional[float] = None,
                 device=None, include_adjacency: bool = True):
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type != "cuda":
            raise TypeError(f"rlsolver_amd.SpinSystem needs a HIP device (got {self.device}); there is no CPU path")
        if observables[0] != Observable.SPIN_STATE:
            raise AssertionError("First observable must be Observation.SPIN_STATE.")
        if reward_signal not in _REWARD_MODE:
            raise NotImplementedError(f"reward_signal {reward_signal} is not supported on the batched env")
        self.observables = list(enumerate(observables))
        self.num_envs, self.n_spins, self.max_steps = num_envs, num_nodes, max_steps
        self.n_actions = num_nodes                       # extra_action = NONE
        self.reward_signal, self.norm_rewards, self.spin_basis = reward_signal, norm_rewards, spin_basis
        self.horizon_length = horizon_length if horizon_length is not None else max_steps
        self.stag_punishment, self.basin_reward = stag_punishment, basin_reward
        self.reversible_spins = True
        self.include_adjacency = include_adjacency
        self.action_space = self._ActionSpace(self.n_actions, self.device)
        self.observation_space = self._ObservationSpace(self.n_spins, len(observables))
        csr = build_csr(mygraph, num_nodes=num_nodes, if_bidirectional=False)
        if np.any(csr.wgt != csr.wgt.astype(np.int32)):
            raise ValueError("SpinSystem needs integer edge weights")
        self.graph = ops.DeviceGraph(csr, self.device, use_weights=True)
        wdeg = np.zeros(num_nodes, np.int64)
        np.add.at(wdeg, np.repeat(np.arange(num_nodes), np.diff(csr.rowptr)), csr.wgt)
        self._max_local = float(wdeg.max())              # host copy: reading it back from the device synced every step
        self.max_local_reward_available_ = torch.full((num_envs,), self._max_local, device=self.device)
        if float(wdeg.max()) == 0.0 or np.abs(wdeg).sum() == 0:
            raise ValueError("empty graph / zero max local reward (the reference re-draws the graph here)")
        self.max_local_reward_available = self.max_local_reward_available_.unsqueeze(1).expand(-1, num_nodes)
        self._rows = (C.c_int32 * 7)(*[next((i for i, o in self.observables if o == want), -1) for want in _ROW_ORDER])
        self._matrix = None
        R, B = len(observables), num_envs
        self.state = torch.zeros((B, R, num_nodes), dtype=torch.float32, device=self.device)
        self._delta = torch.zeros((B, num_nodes), dtype=torch.int32, device=self.device)
        self._num_nonpos = torch.zeros(B, dtype=torch.int32, device=self.device)
        self.score = torch.zeros(B, dtype=torch.float32, device=self.device)
        self.best_score = self.score.clone()
        self.best_spins = torch.zeros((B, num_nodes), dtype=torch.float32, device=self.device)
        self.current_step = 0
        self.reset()

    # ---- dense adjacency only when somebody asks for it (N^2 floats)
    @property
    def matrix(self):
        if self._matrix is None:
            csr = self.graph.csr
            m = np.zeros((self.n_spins, self.n_spins), dtype=np.float32)
            np.add.at(m, (np.repeat(np.arange(self.n_spins), np.diff(csr.rowptr)), csr.col), csr.wgt.astype(np.float32))
            self._matrix = torch.from_numpy(m).to(self.device)
        return self._matrix

    matrix_obs = matrix

    def reset(self, spins=None):
        """spinsystem_PECO.py:150-195.  spins: optional [B, N] in the env's spin basis."""
        self.current_step = 0
        B, N = self.num_envs, self.n_spins
        self.state.zero_()
        if spins is None:
            bits = ops.rand_spins(B, N, _seed_from_torch(), self.device)
            bits[:, 0] = torch.randint(0, 2, (B,), device=self.device, dtype=torch.bool)  # no gauge fixing here
            self.state[:, 0, :] = 2 * bits.float() - 1
        else:
            spins = torch.as_tensor(spins, device=self.device, dtype=torch.float32)
            self.state[:, 0, :] = (2 * spins - 1) if self.spin_basis == SpinBasis.BINARY and spins.min() >= 0 else spins
        _abi.call("rls_spin_delta_init", self.graph.ref, _ptr(self.state), B, self.state.shape[1], _ptr(self._delta),
                  _stream(self.device))
        # f32 quotients formed in f64 and rounded once: identical to a correctly rounded f32 division
        # (torch's GPU tensor/scalar division multiplies by the reciprocal, 1 ulp off the CPU result)
        imm = self._delta.double()
        for idx, obs in self.observables:
            if obs == Observable.IMMEDIATE_REWARD_AVAILABLE:
                self.state[:, idx, :] = (imm / self.max_local_reward_available.double()).float()
            elif obs == Observable.NUMBER_OF_GREEDY_ACTIONS_AVAILABLE:
                self.state[:, idx, :] = (1 - (torch.sum(imm <= 0, dim=-1).double() / N).float()).unsqueeze(-1)
        self.score = self.calculate_cut()
        self.best_score = self.score.clone()
        self.best_obs_score = self.best_score
        self.best_spins = self.state[:, 0, :].clone()
        self.best_obs_spins = self.best_spins
        self.history_buffer = _HistoryBuffer(B, self.device) if (self.stag_punishment is not None or
                                                                  self.basin_reward is not None) else None
        return self.get_observation()

    def calculate_cut(self, spins=None):
        """cut = 1/4 * sum_ij W_ij (1 - s_i s_j)  (spinsystem_PECO.py:564-566) = (sum(W)/2 - sum_i delta_i / 2) / 2,
        exact in integers."""
        if spins is not None:
            raise NotImplementedError("calculate_cut(spins) for foreign spins: use rlsolver_amd.ops.maxcut_obj")
        wsum = int(self.graph.csr.wgt.sum())             # = sum_ij W_ij over ordered pairs
        return (wsum - self._delta.sum(dim=1)).float() / 4

    def calculate_score(self, spins=None):
        return self.calculate_cut(spins)

    def step(self, action):
        """spinsystem_PECO.py:306-486 -> (obs, reward f32 [B], done bool [B])"""
        self.current_step += 1
        if self.current_step > self.max_steps:
            print("The environment has already returned done. Stop it!")
            raise NotImplementedError
        B = self.num_envs
        action = action.to(device=self.device, dtype=torch.int64).contiguous()
        rew = torch.empty(B, dtype=torch.float32, device=self.device)
        # max(0, (current_step - max_steps) / horizon_length + 1), evaluated in f32 like the reference's torch ops
        term = float(max(np.float32(0.0), np.float32((self.current_step - self.max_steps) / self.horizon_length) + np.float32(1)))
        _abi.call("rls_spin_step", self.graph.ref, _ptr(self.state), B, self.state.shape[1], self._rows,
                  _ptr(self._delta), _ptr(action), _ptr(self.score), _ptr(self.best_score), _ptr(self.best_spins),
                  _ptr(rew), _ptr(self._num_nonpos), self._max_local,
                  float(np.float32(1.0 / self.max_steps)), term, _REWARD_MODE[self.reward_signal],
                  float(self.n_spins) if self.norm_rewards else 1.0, _stream(self.device))
        if self.history_buffer is not None:
            visiting_new_state = self.history_buffer.update(self.state[:, 0, :])
            if self.stag_punishment is not None:
                rew[~visiting_new_state] -= self.stag_punishment
            if self.basin_reward is not None:
                rew[(self._num_nonpos == self.n_spins) & visiting_new_state] += self.basin_reward
        done = torch.full((B,), self.current_step == self.max_steps, dtype=torch.bool, device=self.device)
        return self.get_observation(), rew, done

    def get_observation(self):
        state = self.state.clone()
        if self.spin_basis == SpinBasis.BINARY:
            state[:, 0, :] = (1 - state[:, 0, :]) / 2
        if not self.include_adjacency:
            return state
        return torch.cat((state, self.matrix.unsqueeze(0).expand(state.shape[0], -1, -1)), dim=-2)

    def get_immeditate_rewards_avaialable(self, spins=None):
        return self._delta.float()

    def get_allowed_action_states(self):
        return (0, 1) if self.spin_basis == SpinBasis.BINARY else (1, -1)

    def get_best_cut(self):
        return self.best_score

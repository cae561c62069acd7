"""CPU ORACLE for the S2V / ECO / PECO spin system -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates rlsolver/methods/ECO_S2V/src/envs/spinsystem_PECO.py (SpinSystemUnbiased, batched) the
way the reference computes it: DENSE f32 matmul for the single-flip gains every step, per-step
observable rows, BLS / DENSE reward, basin reward via an exact visited-state list.  Shared graph
(one W for every env).  Pinned against tests/golden/spinsystem.npz (captured from the reference).
"""
from __future__ import annotations

import numpy as np

F = np.float32


class SpinSystemOracle:
    # observable row order = ECO_PECO_OBSERVABLES (ECO_S2V/src/envs/util_envs.py:53-59)
    SPIN, IMMEDIATE, TIME_SINCE_FLIP, DIST_SCORE, DIST_STATE, GREEDY, TERMINATION = range(7)

    def __init__(self, W, num_envs, max_steps, reward="DENSE", norm_rewards=False, basin_reward=None):
        self.W = np.asarray(W, F)
        self.n = self.W.shape[0]
        self.B, self.max_steps = num_envs, max_steps
        self.reward, self.norm_rewards, self.basin_reward = reward, norm_rewards, basin_reward
        ones = np.ones((num_envs, self.n), F)
        self.max_local = self._imm(ones).max(axis=-1)                       # spinsystem_PECO.py:160-170

    def _imm(self, spins):
        """_get_immeditate_cuts_avaialable: matmul(W, s) * s  (:660-661)"""
        return (spins @ self.W.T).astype(F) * spins

    def calculate_cut(self, spins):
        """:564-566"""
        return (F(0.25) * (-(spins @ self.W.T).astype(F) * spins).sum(-1, dtype=F) + F(0.25) * self.W.sum(dtype=F)).astype(F)

    def reset(self, spins_signed):
        self.t = 0
        self.state = np.zeros((self.B, 7, self.n), F)
        self.state[:, 0] = spins_signed
        imm = self._imm(self.state[:, 0])
        self.state[:, self.IMMEDIATE] = imm / self.max_local[:, None]
        self.state[:, self.GREEDY] = (F(1) - (imm <= 0).sum(-1).astype(F) / F(self.n))[:, None]
        self.score = self.calculate_cut(self.state[:, 0])
        self.best_score = self.score.copy()
        self.best_spins = self.state[:, 0].copy()
        self.visited = [set() for _ in range(self.B)] if self.basin_reward is not None else None
        return self.observation()

    def observation(self):
        s = self.state.copy()
        s[:, 0] = (F(1) - s[:, 0]) / F(2)                                      # SpinBasis.BINARY (:488-492)
        return s

    def step(self, action):
        self.t += 1
        idx = np.arange(self.B)
        new = self.state.copy()
        new[idx, 0, action] = -self.state[idx, 0, action]
        imm = self._imm(new[:, 0])
        delta = -imm[idx, action]                                              # :346-348
        self.score = (self.score + delta).astype(F)
        self.state = new
        improvement = self.score - self.best_score
        if self.reward == "BLS":
            rew = np.where(improvement > 0, improvement, F(0)).astype(F)
        else:
            rew = delta.astype(F)
        if self.norm_rewards:
            rew = (rew / F(self.n)).astype(F)
        if self.visited is not None:                                           # HistoryBuffer + basin (:383-397)
            fresh = np.zeros(self.B, bool)
            for b in range(self.B):
                key = self.state[b, 0].tobytes()
                fresh[b] = key not in self.visited[b]
                self.visited[b].add(key)
            rew = rew.copy()
            rew[np.all(imm <= 0, axis=-1) & fresh] += F(self.basin_reward)
        upd = self.score > self.best_score
        self.best_score = np.where(upd, self.score, self.best_score)
        self.best_spins = np.where(upd[:, None], self.state[:, 0], self.best_spins)
        st = self.state
        st[:, self.IMMEDIATE] = imm / self.max_local[:, None]
        st[:, self.TIME_SINCE_FLIP] += F(1.0 / self.max_steps)
        st[idx, self.TIME_SINCE_FLIP, action] = 0
        st[:, self.TERMINATION] = max(F(0), F((self.t - self.max_steps) / self.max_steps) + F(1))
        st[:, self.GREEDY] = (F(1) - (imm <= 0).sum(-1).astype(F) / F(self.n))[:, None]
        st[:, self.DIST_SCORE] = (np.abs(self.score - self.best_score) / self.max_local)[:, None]
        st[:, self.DIST_STATE] = np.count_nonzero(self.best_spins - st[:, 0], axis=-1)[:, None]
        done = np.full(self.B, self.t == self.max_steps)
        return self.observation(), rew, done

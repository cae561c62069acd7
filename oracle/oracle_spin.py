"""CPU ORACLE for the S2V / ECO / PECO spin system -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates rlsolver/methods/ECO_S2V/src/envs/spinsystem_PECO.py (SpinSystemUnbiased, batched) the
way the reference computes it: DENSE f32 matmul for the single-flip gains every step, per-step
observable rows, BLS / DENSE reward, basin reward via an exact visited-state list.  Shared graph
(one W for every env).  Pinned against tests/golden/spinsystem.npz (captured from the reference).
``inference=True`` restates its instance-wise twin, ECO_S2V/src/envs/inference_network_env.py (pinned against
tests/golden/spinsystem_inference.npz): best score / spins start from the best env of the batch (:203-206),
step() returns (obs, done).
"""
from __future__ import annotations

import numpy as np

F = np.float32


class SpinSystemOracle:
    # observable row order = ECO_PECO_OBSERVABLES (ECO_S2V/src/envs/util_envs.py:53-59)
    SPIN, IMMEDIATE, TIME_SINCE_FLIP, DIST_SCORE, DIST_STATE, GREEDY, TERMINATION = range(7)

    def __init__(self, W, num_envs, max_steps, reward="DENSE", norm_rewards=False, basin_reward=None,
                 stag_punishment=None, inference=False):
        self.inference = inference
        self.W = np.asarray(W, F)
        self.n = self.W.shape[0]
        self.B, self.max_steps = num_envs, max_steps
        self.reward, self.norm_rewards, self.basin_reward = reward, norm_rewards, basin_reward
        self.stag_punishment = stag_punishment
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
        if self.inference:                                                     # inference_network_env.py:203-206
            i = int(np.argmax(self.score))
            self.best_score = np.full(self.B, self.score[i], F)
            self.best_spins = np.repeat(self.state[i:i + 1, 0], self.B, axis=0)
        self.visited = [set() for _ in range(self.B)] if (not self.inference and (self.basin_reward is not None or
                                                              self.stag_punishment is not None)) else None
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
        elif self.reward == "CUSTOM_BLS":                                      # :372-374
            rew = np.where(improvement > 0, improvement / (improvement + F(0.1)), F(0)).astype(F)
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
            if self.stag_punishment is not None:
                rew[~fresh] -= F(self.stag_punishment)
            if self.basin_reward is not None:
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
        if self.inference:
            return self.observation(), done                                     # inference_network_env.py:444
        return self.observation(), rew, done


class SpinSystemOracleF64:
    """The numpy single-instance env, rlsolver/methods/ECO_S2V/src/envs/spinsystem.py (SpinSystemUnbiased :588-661,
    step :333-482, reset :176-252, observation :484-495), restated in float64 with a DENSE matvec for the gains and
    the action-parity-set visited memory of util_envs.py:355-381.  ECO_PECO_OBSERVABLES row order (``s2v=True``: S2V_OBSERVABLES,
    the spin row alone); OptimisationTarget.CUT or ENERGY (``target``: score = -E = s'Ws / 2, :531-533, :632-647; immediate
    rewards -2 s (W s), :498-499, :654-656); ExtraAction.NONE or PASS (``extra_pass``: n + 1 actions, every array carries the
    reference's padding column, :226-233, :252-262), infinite or finite memory (``memory_length``, :206-209, :398-404),
    reversible or irreversible spins (``reversible=False``: reset to all +1, :262-264; done once no spin is +1, :476-480).
    Pinned against tests/golden/spinsystem_cpu.npz, spinsystem_options.npz and spinsystem_s2v.npz."""
    SPIN, IMMEDIATE, TIME_SINCE_FLIP, DIST_SCORE, DIST_STATE, GREEDY, TERMINATION = range(7)

    def __init__(self, W, max_steps, reward="DENSE", norm_rewards=False, basin_reward=None, stag_punishment=None,
                 extra_pass=False, memory_length=None, target="CUT", reversible=True, s2v=False, binary=True):
        assert target in ("CUT", "ENERGY")
        self.target, self.reversible, self.s2v, self.binary = target, reversible, s2v, binary
        self.W = np.asarray(W, np.float64)
        self.n = self.W.shape[0]
        self.na = self.n + int(extra_pass)                                     # n_actions
        self.max_steps = max_steps
        self.reward, self.norm_rewards = reward, norm_rewards
        self.basin_reward, self.stag_punishment = basin_reward, stag_punishment
        self.memory_length = memory_length
        imm1 = self._imm(np.ones(self.n))
        self.max_local = np.max(imm1[np.nonzero(imm1)])                        # :190-196
        self.W_obs = np.zeros((self.na, self.na))                              # matrix_obs, zero-padded (:222-225)
        self.W_obs[:self.n, :self.n] = self.W

    def _imm(self, s):
        if self.target == "ENERGY":
            return -1 * (2 * s * (self.W @ s))                                 # :498-499 on :654-656
        return s * (self.W @ s)                                                # :659-661

    def _score(self, s):
        if self.target == "ENERGY":
            return -1. * (-(s @ (self.W @ s)) / 2)                             # :531-533 on :644-647
        return 0.25 * np.sum(self.W * (1 - np.outer(s, s)))                    # :601-607

    def reset(self, spins_signed=None):
        n = self.n
        self.t = 0
        st = np.zeros((7, self.na))
        if spins_signed is None:
            assert not self.reversible, "reversible spins start from a random draw: pass it"
            spins_signed = np.ones(n)                                          # :262-264
        st[0, :n] = np.asarray(spins_signed)[:n]
        imm = self._imm(st[0, :n])
        st[self.IMMEDIATE, :n] = imm / self.max_local
        st[self.GREEDY, :n] = 1 - np.sum(imm <= 0) / n                         # reset writes [:n_spins] only (:259-261)
        self.state = st
        self.score = self._score(st[0, :n])
        self.best_score = self.best_obs_score = self.score
        self.best_spins = st[0, :n].copy()
        self.best_obs_spins = st[0, :n].copy()
        if self.memory_length is not None:
            self.score_memory = np.array([self.best_score] * self.memory_length)
            self.spins_memory = np.array([self.best_spins] * self.memory_length)
            self.idx_memory = 1
        self.flipped = frozenset()                                             # HistoryBuffer.current_action_hist
        self.seen = set()
        return self.observation()

    def observation(self):
        s = self.state.copy()
        if self.binary:
            s[0] = (1 - s[0]) / 2                                              # SpinBasis.BINARY (the padding spin 0 -> 0.5)
        if self.s2v:
            s = s[:1]
        return np.vstack((s, self.W_obs))

    def gains(self):
        return self._imm(self.state[0, :self.n])

    def step(self, a):
        n = self.n
        self.t += 1
        new = self.state.copy()
        if a == n:                                                             # ExtraAction.PASS (:349-351)
            delta = 0
        else:
            new[0, a] = -self.state[0, a]
            if self.target == "ENERGY":
                delta = -1. * (-2 * new[0, a] * (new[0, :n] @ self.W[:, a]))   # :542-543 on :626
            else:
                delta = -1 * new[0, a] * (new[0, :n] @ self.W[:, a])           # _calculate_cut_change :631
            self.score += delta
        self.state = new
        imm = self._imm(new[0, :n])
        rew = 0
        if self.score > self.best_obs_score:
            if self.reward == "BLS":
                rew = self.score - self.best_obs_score
            elif self.reward == "CUSTOM_BLS":
                rew = self.score - self.best_obs_score
                rew = rew / (rew + 0.1)
        if self.reward == "DENSE":
            rew = delta
        if self.norm_rewards:
            rew /= n
        if self.stag_punishment is not None or self.basin_reward is not None:
            self.flipped = self.flipped ^ frozenset([a])
            fresh = self.flipped not in self.seen
            self.seen.add(self.flipped)
            if self.stag_punishment is not None and not fresh:
                rew -= self.stag_punishment
            if self.basin_reward is not None and np.all(imm <= 0) and fresh:
                rew += self.basin_reward
        if self.score > self.best_score:
            self.best_score = self.score
            self.best_spins = new[0, :n].copy()
        if self.memory_length is not None:                                     # :398-404
            self.score_memory[self.idx_memory] = self.score
            self.spins_memory[self.idx_memory] = new[0, :n]
            self.idx_memory = (self.idx_memory + 1) % self.memory_length
            self.best_obs_score = self.score_memory.max()
            self.best_obs_spins = self.spins_memory[self.score_memory.argmax()].copy()
        else:
            self.best_obs_score = self.best_score
            self.best_obs_spins = self.best_spins.copy()
        st = self.state
        st[self.IMMEDIATE, :n] = imm / self.max_local
        st[self.TIME_SINCE_FLIP] += 1. / self.max_steps
        st[self.TIME_SINCE_FLIP, a] = 0
        st[self.TERMINATION] = max(0, ((self.t - self.max_steps) / self.max_steps) + 1)
        st[self.GREEDY] = 1 - np.sum(imm <= 0) / n
        st[self.DIST_SCORE] = np.abs(self.score - self.best_obs_score) / self.max_local
        st[self.DIST_STATE, :n] = np.count_nonzero(self.best_obs_spins - st[0, :n])
        done = self.t == self.max_steps
        if not self.reversible and not np.any(st[0, :n] > 0):                  # :476-480
            done = True
        return self.observation(), rew, done

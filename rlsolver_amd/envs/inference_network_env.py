"""The instance-wise inference env -- drop-in for rlsolver/methods/ECO_S2V/src/envs/inference_network_env.py
(``SpinSystemFactory.get`` -> ``SpinSystemUnbiased``; built by inference_PECO.py:84-99, stepped by
``peco_test_network``, ECO_S2V/util.py:20-63) on a HIP device.

It is the batched PECO env on ONE shared [N, N] graph with three differences, all kept:

* ``step(action) -> (obs, done)`` -- no reward (:444);
* ``best_score`` / ``best_spins`` start from the BEST env of the batch, not from each env's own start (:203-206: a
  0-dim ``torch.max`` over the batch, expanded), so the two distance-from-best observables of every env refer to it
  until that env beats it; ``get_best_cut()`` is that 0-dim value before the first step (``peco_test_network`` calls
  ``.item()`` on it) and the per-env vector afterwards (the reference's ``torch.where`` broadcasts it there);
* the observation's adjacency rows are one matrix expanded over the batch (:455).

The reference computes ``matmul(spins [B, N], matrix [N, N]) * spins`` every step; here a step is the O(deg) kernel of
envs/spinsystem.py (rls_spin_step) on the shared CSR graph.  ``use_tensor_core=True`` (float16 state) is refused.
"""
from __future__ import annotations

import numpy as np
import torch

from .spinsystem import (ECO_PECO_OBSERVABLES, ExtraAction, Observable, OptimisationTarget, RewardSignal, SpinBasis,  # noqa: F401
                         SpinSystem)


def _edges_of(matrix) -> tuple:
    """A symmetric integer-valued [N, N] coupling matrix -> (edge list, N)."""
    m = torch.as_tensor(matrix).detach().cpu().numpy().astype(np.float64)
    if m.ndim != 2 or m.shape[0] != m.shape[1]:
        raise ValueError(f"graph_generator.get() must return one [N, N] matrix for the inference env, got {m.shape}")
    if not np.array_equal(m, m.T) or np.any(np.diagonal(m) != 0) or np.any(m != np.rint(m)):
        raise ValueError("the inference env needs a symmetric integer-valued matrix with an empty diagonal")
    iu, ju = np.nonzero(np.triu(m, 1))
    return [(int(a), int(b), int(m[a, b])) for a, b in zip(iu, ju)], int(m.shape[0])


class SpinSystemUnbiased(SpinSystem):
    """inference_network_env.py:541-595 on SpinSystem's kernels.  Positional parameters as there (:81-99).

    Defaults: the reference's own defaults for ``extra_action`` / ``optimisation_target`` / ``reversible_spins`` (PASS, ENERGY,
    False) select configurations no MaxCut caller uses -- every agent passes ExtraAction.NONE, OptimisationTarget.CUT,
    reversible_spins=True explicitly (select_best_neural_network.py:121-136, ECO_S2V/util.py:146-160) and dqn_PECO.py:252
    asserts them -- and this env refuses them.  So that ``SpinSystemUnbiased(gg, num_envs=B)`` is a working env rather than an
    exception, the defaults here are the supported values; passing the reference's defaults explicitly still raises
    NotImplementedError, naming the option."""

    def __init__(self, graph_generator=None, max_steps=20, observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal.DENSE,
                 extra_action=ExtraAction.NONE, optimisation_target=OptimisationTarget.CUT, spin_basis=SpinBasis.SIGNED,
                 norm_rewards=False, memory_length=None, horizon_length=None, stag_punishment=None, basin_reward=None,
                 reversible_spins=True, init_snap=None, seed=None, device=None, num_envs=None, use_tensor_core=False):
        unsupported = [name for name, bad in (("extra_action", extra_action.name != "NONE"),
                                              ("optimisation_target", optimisation_target.name != "CUT"),
                                              ("memory_length", memory_length is not None), ("reversible_spins", not reversible_spins),
                                              ("init_snap", init_snap is not None), ("use_tensor_core", bool(use_tensor_core)),
                                              ("biased graphs", bool(getattr(graph_generator, "biased", False)))) if bad]
        if unsupported:
            raise NotImplementedError("the inference SpinSystem on the device supports ExtraAction.NONE, OptimisationTarget.CUT, infinite "
                                      f"memory, reversible spins, unbiased graphs, float32 state; got {', '.join(unsupported)}")
        if num_envs is None:
            raise ValueError("num_envs is required (inference_PECO.py:91-98 passes the size of the mini-batch)")
        if seed is not None:
            np.random.seed(seed)                               # inference_network_env.py:113-114
        mygraph, n = _edges_of(graph_generator.get())
        same = lambda enum, v: enum[v.name]
        self.use_tensor_core = False
        self._fresh = True
        # stag_punishment / basin_reward only enter the reward, which this env does not return: no visited-state ring
        super().__init__(mygraph, n, num_envs, max_steps, [same(Observable, o) for o in observables], same(RewardSignal, reward_signal),
                         same(SpinBasis, spin_basis), norm_rewards, horizon_length, None, None, device)
        self.gg = graph_generator         # (stag_punishment / basin_reward stay None on the env: see above)

    def reset(self, spins=None):
        obs = super().reset(spins)        # (the rows that refer to the best are zero at a reset: nothing below changes obs)
        # :203-206: every env starts from the batch's best score and spins (first maximum, torch.argmax's documented choice)
        i = torch.argmax(self.score)
        sp = self._state[:, 0, :]
        self.best_score.copy_(self.score[i].expand(self.num_envs))
        self.best_spins.copy_(sp[i].expand(self.num_envs, self.n_spins))
        self._dist_best.copy_((sp != sp[i]).sum(dim=1).to(torch.int32))     # the step kernel keeps this Hamming distance incrementally
        self._fresh = True
        return obs

    def step(self, action):
        obs, _, done = super().step(action)
        self._fresh = False
        return obs, done

    def get_best_cut(self):
        return self.best_score[0] if self._fresh else self.best_score


class SpinSystemFactory:
    """inference_network_env.py:19-55."""

    @staticmethod
    def get(graph_generator=None, max_steps=20, observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal.DENSE,
            extra_action=ExtraAction.NONE, optimisation_target=OptimisationTarget.CUT, spin_basis=SpinBasis.SIGNED,
            norm_rewards=False, memory_length=None, horizon_length=None, stag_punishment=None, basin_reward=None,
            reversible_spins=True, init_snap=None, seed=None, device=None, num_envs=None, if_greedy=False, use_tensor_core=False):
        return SpinSystemUnbiased(graph_generator, max_steps, observables, reward_signal, extra_action, optimisation_target, spin_basis,
                                  norm_rewards, memory_length, horizon_length, stag_punishment, basin_reward, reversible_spins,
                                  init_snap, seed, device, num_envs, use_tensor_core)

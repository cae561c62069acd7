#!/usr/bin/env python3
"""S2V-DQN's environment loop -- rlsolver/methods/ECO_S2V/train_and_inference/train_S2V.py:37-82 (env_args) and the acting
part of src/agents/dqn.py:419-430, 480-492 (an action is drawn among the spins whose value still equals
`env.get_allowed_action_states()`) -- on rlsolver_amd's drop-in env, with the import swapped and nothing else:

    - import rlsolver.methods.ECO_S2V.src.envs.core as ising_env
    + import rlsolver_amd.envs.spinsystem as ising_env

The env is the single-instance one with IRREVERSIBLE spins: an episode starts from all spins +1 (BINARY basis: 0), every spin
may be flipped once, `done` comes when none is left (or after max_steps), the reward is the DENSE cut change / n_spins.  The
policy here is a stand-in for the MPNN Q-network (greedy on the env's own immediate cut gains, epsilon-random): the point is the
env surface the agent sees.

    python examples/s2v_episodes.py [--nodes 40] [--episodes 5]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class RandomBAGraphs:
    """What train_S2V.py hands the env: an object with n_spins and get() -> [N, N] couplings, a fresh graph per reset (the
    reference's util_envs.RandomBAGraphGenerator is a networkx wrapper; any such object works)."""
    biased = False

    def __init__(self, n_spins, m, seed):
        self.n_spins, self.m, self.rng = n_spins, m, np.random.RandomState(seed)

    def get(self, with_padding=False):
        n, m = self.n_spins, self.m
        W = np.zeros((n, n))
        targets = list(range(m))
        repeated = []
        for v in range(m, n):
            for t in set(targets):
                W[v, t] = W[t, v] = self.rng.choice([-1.0, 1.0])       # EdgeType.DISCRETE
            repeated.extend(set(targets))
            repeated.extend([v] * m)
            targets = [repeated[i] for i in self.rng.randint(0, len(repeated), m)]
        return W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=40)
    ap.add_argument("--episodes", type=int, default=5)
    ap.add_argument("--epsilon", type=float, default=0.1)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()

    import torch
    import rlsolver_amd.envs.spinsystem as ising_env
    from rlsolver_amd.envs.spinsystem import (S2V_OBSERVABLES, ExtraAction, OptimisationTarget, RewardSignal, SpinBasis)

    env_args = {'observables': S2V_OBSERVABLES,                      # train_S2V.py:37-47, verbatim
                'reward_signal': RewardSignal.DENSE,
                'extra_action': ExtraAction.NONE,
                'optimisation_target': OptimisationTarget.CUT,
                'spin_basis': SpinBasis.BINARY,
                'norm_rewards': True,
                'memory_length': None,
                'horizon_length': None,
                'stag_punishment': None,
                'basin_reward': None,
                'reversible_spins': False}
    gen = RandomBAGraphs(a.nodes, 4, a.seed)
    env = ising_env.make("SpinSystem", gen, int(a.nodes * 1), device=torch.device("cuda:0"), **env_args)
    rng = np.random.RandomState(a.seed + 1)
    allowed = env.get_allowed_action_states()                         # dqn.py:254 -- 0 under the BINARY basis
    assert not env.reversible_spins and allowed == 0
    for ep in range(a.episodes):
        obs = env.reset()                                             # [1 + N, N]: the spin row, then the couplings
        score0, ret, steps, done = env.score, 0.0, 0, False
        while not done:
            flippable = np.nonzero(obs[0, :] == allowed)[0]           # dqn.py:422
            assert flippable.size > 0
            if rng.rand() < a.epsilon:
                action = int(rng.choice(flippable))
            else:
                gains = env.get_immeditate_rewards_avaialable()       # stand-in for the Q-values
                action = int(flippable[np.argmax(gains[flippable])])
            obs, rew, done, _ = env.step(action)
            ret += rew
            steps += 1
        assert steps == a.nodes and not (obs[0, :] == allowed).any()  # every spin flipped exactly once: that is what ended it
        assert abs(ret * a.nodes - (env.score - score0)) < 1e-9       # the DENSE rewards sum to the cut change
        print(f"episode {ep}: {steps} steps, cut {env.score:.0f} (best on the way {env.best_score:.0f}), return {ret:.4f}")
    print("s2v_episodes: ok")


if __name__ == "__main__":
    main()

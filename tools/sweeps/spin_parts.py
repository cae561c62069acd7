"""Dev: S1 step time with the full ECO observable set vs only the O(deg) rows (spin state + immediate reward): the
difference is the streaming of the five rows that change everywhere."""
import sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, ".")
from rlsolver_amd import ops, _abi
from rlsolver_amd.graph import generate_gnm
from rlsolver_amd.envs.spinsystem import ECO_PECO_OBSERVABLES, Observable, RewardSignal, SpinBasis, SpinSystem
dev = torch.device("cuda:0")
n, m, B, T = 2000, 19990, 1 << 14, 64
rng = np.random.RandomState(1)
mg = [(u, v, int(rng.choice([-1, 1]))) for u, v, _ in generate_gnm(n, m, 22)]
for label, obs in (("ECO observables", ECO_PECO_OBSERVABLES), ("spin state + immediate reward only", [Observable.SPIN_STATE, Observable.IMMEDIATE_REWARD_AVAILABLE])):
    env = SpinSystem(mg, n, B, max_steps=T, observables=obs, reward_signal=RewardSignal.BLS, norm_rewards=True,
                     spin_basis=SpinBasis.BINARY, device=dev, include_adjacency=False)
    acts = [ops.rand_actions(B, n, 11, s, dev) for s in range(8)]
    rew = torch.empty(B, device=dev)
    R = len(obs)

    def one(i):
        if env.current_step >= T:
            env.current_step = 0
        env.current_step += 1
        torch.ops.rlsolver_hip.spin_step(env.graph.handle, env._env_handle, env._state, env._rows, acts[i % 8], rew, None, env._max_local, 1.0,
                                         1, float(n), env.current_step - 1, False, 0.0, False, 0.0)
    for i in range(3):
        one(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for i in range(40):
        one(i)
    e1.record(); torch.cuda.synchronize()
    print("%-36s %.1f us per bare kernel step" % (label, e0.elapsed_time(e1) / 40 * 1e3))

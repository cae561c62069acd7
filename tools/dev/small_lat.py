"""Dev: latency of the class surface at the small sizes the reference's own scripts default to."""
import sys, types
import numpy as np, torch
sys.path.insert(0, ".")
from rlsolver_amd.graph import generate_gnm, generate_ba
from rlsolver_amd.envs.env_L2A import EnvMaxcut
from rlsolver_amd.methods.LocalSearch import LocalSearch
dev = torch.device("cuda:0")


def t(fn, it=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for name, n, m in (("G14-sized", 800, 4694), ("G22-sized", 2000, 19990)):
    mg = generate_gnm(n, m, 14)
    env = EnvMaxcut(mygraph=mg, device=dev, num_nodes=n)
    for B in (64, 256, 1024, 4096):
        xs = env.generate_xs_randomly(B)
        vs = env.calculate_obj_values(xs)
        print("%s B=%5d: obj %6.1f us | local_search_inplace(8 iters) %7.1f us | for_loop %6.1f us" % (
            name, B, t(lambda: env.calculate_obj_values(xs)), t(lambda: env.local_search_inplace(xs.clone(), vs.clone())),
            t(lambda: env.calculate_obj_values_for_loop(xs, True))))
from rlsolver_amd.envs.spinsystem import ECO_PECO_OBSERVABLES, RewardSignal, SpinBasis, SpinSystem
from rlsolver_amd import ops
rng = np.random.RandomState(1)
n = 200
mg = [(u, v, int(rng.choice([-1, 1]))) for u, v, _ in generate_ba(n, 4, 3)]
for B in (1, 50, 1024):
    env = SpinSystem(mg, n, B, max_steps=2 * n, observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal.BLS, norm_rewards=True,
                     spin_basis=SpinBasis.BINARY, device=dev, include_adjacency=True)
    env.reset()
    acts = [ops.rand_actions(B, n, 11, s, dev) for s in range(8)]
    k = [0]

    def one():
        if env.current_step >= 2 * n - 1:
            env.reset()
        k[0] += 1
        env.step(acts[k[0] % 8])
    print("SpinSystem BA-200 B=%5d: step (with the [B, 7 + N, N] observation) %7.1f us" % (B, t(one, 50)))

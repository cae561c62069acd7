"""2^20 + 2^16 envs of a G22-sized graph on ONE GPU (2.1e9 spins: past 2^31 elements): every MaxCut entry point runs and a sample of
rows matches the oracle -- looks for 32-bit index arithmetic.  `python tools/sweeps/huge_batch.py`."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import oracle_np as onp
from rlsolver_amd import ops
from rlsolver_amd.envs.env_L2A import EnvMaxcut
from rlsolver_amd.envs.env_PPO import EnvMaxcut as Gym
from rlsolver_amd.graph import generate_gnm

dev = torch.device("cuda:0")
n, m, B = 2000, 19990, (1 << 20) + (1 << 16)
assert B * n > 2 ** 31
mg = generate_gnm(n, m, 22)
garr = np.asarray(mg, dtype=np.int64)
env = EnvMaxcut(mygraph=mg, device=dev, num_nodes=n)
g = env.graph
xs = ops.rand_spins(B, n, 3, dev)
rows = np.array([0, 1, 63, 64, 4097, B // 2, B // 2 + 1, (1 << 20) - 65, (1 << 20) + 7, 1073742, 1073743, B - 2, B - 1])
sub = xs[rows].cpu().numpy()
vs = ops.maxcut_obj(g, xs)
assert np.array_equal(vs[rows].cpu().numpy(), onp.maxcut_obj(sub, garr, False)), "K1"
d = ops.maxcut_delta_all(g, xs)
assert np.array_equal(d[rows].cpu().numpy(), onp.maxcut_delta_all(sub, garr, n, None)), "K3"
del d
c = ops.maxcut_node_cutdeg(g, xs)
assert np.array_equal(c[rows].cpu().numpy(), onp.maxcut_node_cutdeg(sub, garr, n, False)), "K2"
del c
x5, v5 = xs.clone(), vs.clone()
ops.maxcut_greedy_sweep(g, x5, v5)
wx, wv = onp.greedy_sweep(sub.astype(bool), onp.maxcut_obj(sub, garr, False), garr, False)
assert np.array_equal(x5[rows].cpu().numpy(), wx.astype(np.uint8)) and np.array_equal(v5[rows].cpu().numpy(), wv), "K5"
del x5
mask = torch.zeros_like(xs)
mask[:, :5] = True
x6, v6 = xs.clone(), vs.clone()
ops.maxcut_propose_accept(g, x6, mask, v6)
prop = sub ^ mask[rows].cpu().numpy().astype(np.uint8)
pv = onp.maxcut_obj(prop, garr, False)
acc = pv >= vs[rows].cpu().numpy()
assert np.array_equal(x6[rows].cpu().numpy(), np.where(acc[:, None], prop, sub)), "K6"
del x6, mask
xl, vl = xs.clone(), vs.clone()
env.local_search_inplace(xl, vl, num_iters=2, num_spin=8)
assert bool((vl >= vs).all()) and np.array_equal(vl[rows].cpu().numpy(), onp.maxcut_obj(xl[rows].cpu().numpy(), garr, False)), "LS"
del xl
gym = Gym(types.SimpleNamespace(num_nodes=n, num_envs=B, num_steps=10 ** 9), mygraph=mg, device=dev, spin_dtype=torch.bool)
gym.reset()
a = torch.randint(0, n, (B,), device=dev)
obs, r, dn, cur = gym.step(a)
assert np.array_equal(cur[rows].cpu().numpy().astype(np.int64), onp.maxcut_obj(obs[rows].cpu().numpy().astype(np.uint8), garr, False)), "K4"
slot = torch.empty_like(obs)
obs2, r2, dn2, cur2 = gym.step(a, out=slot)
assert obs2.data_ptr() == slot.data_ptr()
assert np.array_equal(cur2[rows].cpu().numpy().astype(np.int64), onp.maxcut_obj(obs2[rows].cpu().numpy().astype(np.uint8), garr, False)), "K4 emit"
print("huge_batch: 2^20 + 2^16 envs x 2000 nodes: K1 K2 K3 K5 K6 LS", "K4", "match the oracle on sampled rows")

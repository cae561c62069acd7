import torch, types
from rlsolver_amd.graph import generate_gnm
from rlsolver_amd.envs.env_PPO import EnvMaxcut as GymEnv
from rlsolver_amd.envs.vec_env import MaxcutVecEnv
dev = torch.device('cuda:0')
n, m, B = 2000, 19990, 65536
mg = generate_gnm(n, m, 22)
def t(f, K=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(K): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / K * 1e3
args = types.SimpleNamespace(num_nodes=n, num_envs=B, num_steps=10 ** 9)
env = GymEnv(args, mg, dev, False)
env.reset()
acts = [torch.randint(0, n, (B,), device=dev) for _ in range(8)]
k = [0]
def st():
    k[0] += 1
    env.step(acts[k[0] % 8])
us = t(st)
print(f"env_PPO.EnvMaxcut.step  (f32 state, in place): {us:8.1f} us  {B/us*1e6:.3g} env-steps/s")
try:
    venv = MaxcutVecEnv(mg, num_envs=B, max_step=10 ** 9, device=dev)
    venv.reset()
    def st2():
        k[0] += 1
        venv.step(acts[k[0] % 8].to(torch.int32))
    us = t(st2)
    print(f"MaxcutVecEnv.step: {us:8.1f} us  {B/us*1e6:.3g} env-steps/s")
except Exception as ex:
    print("vec env:", repr(ex)[:200])

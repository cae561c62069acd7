"""Differential fuzz of the gym env class (env_PPO.EnvMaxcut: reset / step with the done rule, in place and emitting into
rollout slots, f32 reference surface and 1-byte spins, reuse_buffers) against the numpy restatement of the reference's env on
random graphs, env counts and episode lengths.  `python tools/fuzz/fuzz_gym.py [seconds] [seed]`."""
import sys, time, types
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import oracle_np as onp
from rlsolver_amd import graph as G
from rlsolver_amd import _abi; _abi.tuning_from_env()   # RLS_<KNOB> variables -> rls_tuning_set (forced forms)
from rlsolver_amd.envs.env_PPO import EnvMaxcut

DEV = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    kind = rng.choice(["gnm", "ba", "hub"])
    n = int(rng.choice([rng.randint(5, 80), rng.randint(80, 700), rng.randint(700, 2600)]))
    if kind == "gnm":
        graph = np.asarray(G.generate_gnm(n, int(rng.randint(n - 1, min(n * 8, n * (n - 1) // 2))), int(rng.randint(1 << 30))), dtype=np.int64)
    elif kind == "ba":
        graph = np.asarray(G.generate_ba(n, int(rng.randint(1, min(7, n - 1))), int(rng.randint(1 << 30))), dtype=np.int64)
    else:
        e = {(0, j) for j in rng.choice(np.arange(1, n), min(n - 1, int(rng.choice([70, 300, 900]))), replace=False).tolist()}
        for _ in range(n):
            a, b = rng.randint(0, n, 2)
            if a != b:
                e.add((min(a, b), max(a, b)))
        graph = np.asarray([(a, b, 1) for a, b in sorted(e)], dtype=np.int64)
    B = int(rng.choice([1, 3, 16, 64, 65, 256, 1000]))
    num_steps = int(rng.randint(1, 12))
    bidir = bool(rng.rand() < 0.4)
    dt = torch.float32 if rng.rand() < 0.6 else torch.bool
    reuse = bool(rng.rand() < 0.5)
    tag = f"it={it} kind={kind} n={n} E={len(graph)} B={B} num_steps={num_steps} bidir={bidir} dtype={dt} reuse={reuse}"
    if "-v" in sys.argv:
        print(tag, flush=True)
    torch.manual_seed(int(rng.randint(1 << 30)))
    env = EnvMaxcut(types.SimpleNamespace(num_nodes=n, num_envs=B, num_steps=num_steps), mygraph=[tuple(int(v) for v in r) for r in graph],
                    device=DEV, if_bidirectional=bidir, spin_dtype=dt, reuse_buffers=reuse)
    obs = env.reset()
    assert obs.shape == (B, n) and obs.dtype == dt and not bool(obs[:, 0].any()), "reset " + tag
    ora = onp.PPOEnvOracle(graph, n, num_steps, bidir)
    ora.reset_to((obs > 0).cpu().numpy() if dt == torch.float32 else obs.cpu().numpy())
    slots = [torch.empty((B, n), dtype=dt, device=DEV) for _ in range(3)]
    for t in range(2 * num_steps + 3):
        a = rng.randint(0, n, B)
        if rng.rand() < 0.5:
            xs, r, d, c = env.step(torch.from_numpy(a).to(DEV))
        else:
            xs, r, d, c = env.step(torch.from_numpy(a).to(DEV), out=slots[t % 3])
            assert xs.data_ptr() == slots[t % 3].data_ptr(), "emit slot " + tag
        wx, wr, wd, wc = ora.step(a)
        got = xs.float().cpu().numpy()
        assert np.array_equal(got, wx) and np.array_equal(r.cpu().numpy(), wr), f"step {t} state / reward " + tag
        assert np.array_equal(d.cpu().numpy(), wd) and np.array_equal(c.cpu().numpy(), wc), f"step {t} done / cur " + tag
    it += 1
print(f"fuzz_gym: {it} random configurations, no mismatch")

"""The workload of tests/test_gpu_two_process.py: run as ``python shard_child.py <outdir>`` by every rank of a world
(RANK / WORLD_SIZE in the environment; all ranks share cuda:0 and talk over a gloo group -- the 8-byte keys, per-node
ranges and winner rows go through host copies) and imported by the test itself for the one-process run.

Every section runs a sharded drop-in class on this rank's contiguous share of the batch and stores what it computed."""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_NODES, N_EDGES, BATCH, STEPS = 300, 1500, 512, 12
MCPG_N, MCPG_E, MCPG_M, MCPG_R = 400, 1800, 256, 4


def workload(rank: int, world: int, group, dev):
    from rlsolver_amd import dist as rdist, ops
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    from rlsolver_amd.envs.env_PPO import EnvMaxcut as Gym
    from rlsolver_amd.graph import generate_gnm
    from rlsolver_amd.methods import MCPG as amcpg
    from rlsolver_amd.methods.LocalSearch import LocalSearch
    from rlsolver_amd.methods.util_evaluator import Evaluator
    out = {}
    g = generate_gnm(N_NODES, N_EDGES, 31)
    off, cnt = rdist.env_shard(BATCH, rank, world)

    # ---- L2A-style search: random rows -> local search -> evaluator -> everyone restarts from the global best
    env = EnvMaxcut(mygraph=g, device=dev, num_nodes=N_NODES, env_offset=off, group=group)
    torch.manual_seed(1234)                                      # every rank seeds alike: the kernel seeds are shared,
    xs = env.generate_xs_randomly(cnt)                           # the env ids are not
    xs, vs = env.local_search_inplace(xs, torch.empty(()), num_iters=6, num_spin=8)
    out["ls_xs"], out["ls_vs"] = xs.clone(), vs.clone()
    ev = Evaluator(save_dir=os.path.join(os.environ.get("RLS_OUT", "/tmp"), f"ev_{world}_{rank}"), num_bits=N_NODES,
                   x=torch.zeros(N_NODES, dtype=torch.bool, device=dev), v=0, if_maximize=True)
    ev.record2(1, vs, xs, group=group)
    out["ev_best_v"], out["ev_best_x"] = torch.tensor(float(ev.best_v)), ev.best_x.clone()
    solver = LocalSearch(env, N_NODES)
    solver.reset(xs)
    gx, gv, _ = solver.random_search(num_iters=4, num_spin=6)
    out["rs_xs"], out["rs_vs"] = gx.clone(), gv.clone()
    best, owner = rdist.share_best(gx, gv, group=group)          # env_MCPG.py:452-458, sharded
    out["shared_best"], out["shared_row"] = best.clone(), gx[0].clone()
    assert bool((gv == best).all()) and bool((gx == gx[0]).all())

    # ---- the gym env (K4): reset + STEPS steps, best objective of the episode exchanged at its end (C1 + C2)
    gym = Gym(types.SimpleNamespace(num_nodes=N_NODES, num_envs=cnt, num_steps=STEPS), mygraph=g, device=dev,
              spin_dtype=torch.bool, env_offset=off)
    torch.manual_seed(99)
    gym.reset()
    for t in range(STEPS):
        act = ops.rand_actions(cnt, N_NODES, seed=5, step=t, device=dev, env_offset=off)
        _, rew, done, cur = gym.step(act)
    out["gym_xs"], out["gym_cur"] = gym.xs.clone(), cur.clone()
    b, o, bx, gi = rdist.global_best(gym._obj, gym.xs, want_solution=True, group=group, env_offset=off)
    out["gym_best"], out["gym_best_x"], out["gym_best_index"] = b.clone(), bx.clone(), gi.clone()

    # ---- MCPG: three rounds of a sharded MCPGRound at fixed probabilities, then run_mcpg with its policy updates
    gm = np.asarray(generate_gnm(MCPG_N, MCPG_E, 12), dtype=np.int64)
    data = amcpg.make_data(MCPG_N, gm[:, 0].copy(), gm[:, 1].copy(), dev)
    m0, ml = rdist.env_shard(MCPG_M // 64, rank, world)
    m0, ml = m0 * 64, ml * 64
    gen = torch.Generator().manual_seed(7)
    xs_init = (torch.rand(MCPG_N, MCPG_M, generator=gen) < 0.5).float().to(dev)
    probs = (torch.rand(MCPG_N, generator=gen) * 0.6 + 0.2).to(dev)
    rnd = amcpg.MCPGRound(data, xs_init[:, m0:m0 + ml].contiguous(), torch.zeros(ml, device=dev), ml, MCPG_R, 2,
                          kept_offset=m0, total_kept=MCPG_M, group=group)
    torch.manual_seed(6)
    for r in range(3):
        value, best = rnd.step(probs)
        out[f"mcpg{r}_samples"], out[f"mcpg{r}_expected"] = rnd.samples.words.clone(), rnd.expected.clone()
        out[f"mcpg{r}_res"], out[f"mcpg{r}_info"] = rnd.now_max_res.clone(), rnd.now_max_info.words.clone()
        out[f"mcpg{r}_best"] = best.clone()
    v, x = rnd.best_solution()
    out["mcpg_best_v"], out["mcpg_best_x"] = torch.tensor(v), x.clone()
    torch.manual_seed(8)
    logs = []
    v, x, _ = amcpg.run_mcpg(data, xs_init[:, m0:m0 + ml].contiguous(), torch.zeros(ml, device=dev), ml, MCPG_R, 2, num_rounds=3,
                             sample_epoch_num=2, log=lambda *a: logs.append(" ".join(str(s) for s in a)), kept_offset=m0,
                             total_kept=MCPG_M, group=group)
    out["run_best_v"], out["run_best_x"] = torch.tensor(v), x.clone()
    out["run_cut_of_best_x"] = ops.maxcut_obj(data.graph, x[None, :].contiguous()).clone()
    return {k: t.detach().cpu().numpy() for k, t in out.items()}, {"off": off, "cnt": cnt, "m0": m0, "ml": ml}


def main():
    outdir = sys.argv[1]
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)                               # both ranks on the one GPU of the box
    torch.cuda.set_device(dev)
    os.environ["RLS_OUT"] = outdir
    arrays, meta = workload(rank, world, dist.group.WORLD, dev)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **arrays)
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump(meta, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

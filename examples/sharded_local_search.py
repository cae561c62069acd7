#!/usr/bin/env python3
"""The reference's search loop -- rlsolver/envs/env_MCPG.py:407-493 (search_and_evaluate_local_search: restart every env from
the best row, perturb, LocalSearch.random_search, keep the better rows, track the best) -- on rlsolver_amd's drop-in classes,
with the env batch SHARDED over the GPUs of a node: one process per GPU, every rank runs this same file.

    python examples/sharded_local_search.py                                      # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/sharded_local_search.py
    python -m torch.distributed.run ... --nproc-per-node 2 examples/sharded_local_search.py --share-gpu     # 2 ranks, ONE GPU (gloo)

What the sharded loop adds to the reference's lines -- and nothing else:
  * `env_offset=` / `group=` on the simulator: this rank's rows are rows [off, off + cnt) of the one-process batch;
  * the perturbation's node picks come from a generator keyed by the global env id (`ops.rand_actions`);
  * `best_xs[:] = best_xs[best_vs.argmax()]`  ->  `dist.share_best(best_xs, best_vs)`   (C1 + C2: 8 + N / 8 bytes);
  * `evaluator.record2(..., group=)`: the batch's best row is found over all ranks before it is recorded.
The printed result -- best cut and its solution string -- is the SAME for any number of ranks (tests/test_gpu_rccl.py runs it
with 1 and 2 ranks and compares)."""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=800)
    ap.add_argument("--edges", type=int, default=4694, help="G(n, m) stand-in of that size (gset_14: 800 / 4694) unless --graph-file")
    ap.add_argument("--graph-file", default="", help="a Gset-format file (data/gset/gset_14.txt ...)")
    ap.add_argument("--num-sims", type=int, default=2 ** 12)
    ap.add_argument("--num-iter1", type=int, default=8)
    ap.add_argument("--num-iter0", type=int, default=4)
    ap.add_argument("--ls-iters", type=int, default=2 ** 6)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--share-gpu", action="store_true", help="all ranks on cuda:0, gloo between them (a box with one GPU)")
    a = ap.parse_args()

    import torch as th
    import torch.distributed as dist
    from rlsolver_amd import dist as rdist, ops
    from rlsolver_amd.envs.env_MCPG import EnvMaxcut
    from rlsolver_amd.graph import generate_gnm, read_mygraph
    from rlsolver_amd.methods.LocalSearch import LocalSearch
    from rlsolver_amd.methods.util_evaluator import Evaluator
    from rlsolver_amd.methods.util_read_data import update_xs_by_vs

    rank, local_rank, world = rdist.init_from_env(backend="gloo" if a.share_gpu else None)
    device = th.device("cuda", 0 if a.share_gpu else local_rank)
    th.cuda.set_device(device)
    group = dist.group.WORLD if dist.is_initialized() else None
    off, cnt = rdist.env_shard(a.num_sims, rank, world)               # this rank's rows of the batch

    mygraph = read_mygraph(a.graph_file) if a.graph_file else generate_gnm(a.nodes, a.edges, seed=14)
    th.manual_seed(a.seed)                                            # the SAME seed on every rank
    sim = EnvMaxcut(sim_name="sharded", mygraph=mygraph, device=device, num_nodes=0 if a.graph_file else a.nodes,
                    env_offset=off, group=group)
    num_nodes = sim.num_nodes
    save_dir = os.path.join(tempfile.gettempdir(), f"sharded_ls_{os.getpid()}")
    evaluator = Evaluator(save_dir=save_dir, num_bits=num_nodes, x=th.zeros(num_nodes, dtype=th.bool, device=device), v=0, if_maximize=True)
    solver = LocalSearch(simulator=sim, num_nodes=num_nodes)

    best_xs = sim.generate_xs_randomly(cnt)
    best_vs = sim.calculate_obj_values(best_xs)
    rows = th.arange(cnt, device=device)
    update_j1 = 0
    for j1 in range(a.num_iter1):
        rdist.share_best(best_xs, best_vs, group=group)               # env_MCPG.py:452-454: everyone restarts from the best row
        xs = best_xs.clone()
        for k in range(a.num_iter0):                                  # :457-460: a few random single flips per env
            sample = ops.rand_actions(cnt, num_nodes, seed=a.seed + 1, step=j1 * a.num_iter0 + k, device=device, env_offset=off)
            xs[rows, sample] = th.logical_not(xs[rows, sample])
        solver.reset(xs)                                              # :463
        for j0 in range(a.num_iter0):                                 # :466-471 (update_xs_by_vs returns the batch size there: no early exit)
            solver.random_search(num_iters=a.ls_iters, num_spin=4)
            update_xs_by_vs(best_xs, best_vs, solver.good_xs, solver.good_vs)
        if_update1 = evaluator.record2(i=j1, vs=solver.good_vs, xs=solver.good_xs, group=group)     # :481-486, over all ranks
        if rank == 0:
            evaluator.logging_print(show_str=f"{evaluator.best_v:6}", if_show_x=False)
        if if_update1:
            update_j1 = j1
        elif j1 - update_j1 > 3:
            break
    if rank == 0:
        print(json.dumps({"ranks": world, "num_sims": a.num_sims, "num_nodes": num_nodes, "best": evaluator.best_v,
                          "x_str": evaluator.best_x_str, "cut_of_x": int(sim.calculate_obj_values(evaluator.best_x[None, :].contiguous())[0])}),
              flush=True)
    if dist.is_initialized():
        dist.barrier() if dist.get_backend() != "nccl" else dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

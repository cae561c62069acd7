"""A sharded run on the hardware a test box has: TWO fresh processes share cuda:0, form a gloo group (the 8-byte keys, the
per-node ranges and the winners' rows travel through host copies) and each run their contiguous share of the batch through
the drop-in classes -- EnvMaxcut.generate_xs_randomly / local_search_inplace, Evaluator.record2(group=), LocalSearch,
dist.share_best, the gym env's K4 loop with the episode-end exchange, MCPGRound and run_mcpg (tests/shard_child.py).
The union of what the two ranks computed must be what ONE process computes on the whole batch, bit for bit, and both ranks
must hold the same global best (process model of rlsolver/methods/S2V_PPO/launch.py:17)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("two_proc"))
    world = 2
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RLS_FORCE_PG", None)
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "shard_child.py"), out], env=e, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=600)
            logs.append(o)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, logs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-4000:]}"
    parts = [(dict(np.load(os.path.join(out, f"rank{r}.npz"))), json.load(open(os.path.join(out, f"rank{r}.json")))) for r in range(world)]
    sys.path.insert(0, HERE)
    import shard_child
    os.environ["RLS_OUT"] = out
    whole, _ = shard_child.workload(0, 1, None, torch.device("cuda", 0))
    return whole, parts


def test_local_search_union_equals_one_process(runs):
    whole, parts = runs
    for k in ("ls_xs", "ls_vs", "rs_xs", "rs_vs", "gym_xs", "gym_cur"):
        union = np.concatenate([p[k] for p, _ in parts])
        assert union.shape == whole[k].shape and np.array_equal(union, whole[k]), k
    assert [m["off"] for _, m in parts] == [0, whole["ls_xs"].shape[0] // 2]


def test_both_ranks_hold_the_global_best(runs):
    whole, parts = runs
    for k in ("ev_best_v", "ev_best_x", "shared_best", "shared_row", "gym_best", "gym_best_x", "gym_best_index"):
        for p, _ in parts:
            assert np.array_equal(p[k], whole[k]), k
    # the evaluator's best is the batch's first best row
    i = int(np.argmax(whole["ls_vs"]))
    assert float(whole["ev_best_v"]) == float(whole["ls_vs"][i]) and np.array_equal(whole["ev_best_x"], whole["ls_xs"][i])
    assert int(whole["gym_best_index"]) == int(np.argmax(whole["gym_cur"]))


def test_mcpg_rounds_union_equals_one_process(runs):
    import shard_child as sc
    whole, parts = runs
    M, R = sc.MCPG_M, sc.MCPG_R
    for r in range(3):
        for p, meta in parts:
            m0, ml = meta["m0"], meta["ml"]
            tl = ml // 64
            gt = [q * (M // 64) + m0 // 64 + t for q in range(R) for t in range(tl)]
            gc = [q * M + m0 + j for q in range(R) for j in range(ml)]
            assert np.array_equal(p[f"mcpg{r}_samples"], whole[f"mcpg{r}_samples"][gt]), f"round {r}: chains differ"
            assert np.array_equal(p[f"mcpg{r}_expected"], whole[f"mcpg{r}_expected"][gc])
            assert np.array_equal(p[f"mcpg{r}_res"], whole[f"mcpg{r}_res"][m0:m0 + ml])
            assert np.array_equal(p[f"mcpg{r}_info"], whole[f"mcpg{r}_info"][m0 // 64:m0 // 64 + tl])
            assert np.array_equal(p[f"mcpg{r}_best"], whole[f"mcpg{r}_best"])
    for p, _ in parts:
        assert float(p["mcpg_best_v"]) == float(whole["mcpg_best_v"]) and np.array_equal(p["mcpg_best_x"], whole["mcpg_best_x"])


def test_run_mcpg_ranks_agree(runs):
    """run_mcpg updates its policy from float sums whose order is not fixed (atomics), so later rounds are compared between the
    ranks -- which share every sum through the exchange and must agree exactly -- and, against one process, by what they
    guarantee: the reported value is the cut of the reported solution."""
    whole, parts = runs
    a, b = parts[0][0], parts[1][0]
    assert float(a["run_best_v"]) == float(b["run_best_v"]) and np.array_equal(a["run_best_x"], b["run_best_x"])
    for p in (a, b, whole):
        assert float(p["run_best_v"]) == float(p["run_cut_of_best_x"][0])

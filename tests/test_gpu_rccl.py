"""The episode-boundary exchange over RCCL (backend "nccl") on the one GPU a test box has: a fresh child process
forms a 1-rank group (RLS_FORCE_PG=1 keeps it on the collective path) and runs C1 / C2 on device tensors; a second
child runs bench.py itself under the same group.  SURVEY.md section 8e; the multi-rank logic is covered by the gloo
tests (tests/test_dist_gloo.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path.insert(0, os.environ["RLS_ROOT"])
import torch, torch.distributed as dist
from rlsolver_amd import dist as rdist, ops
from rlsolver_amd.graph import build_csr, generate_gnm
rank, local_rank, world = rdist.init_from_env()
assert dist.is_initialized() and dist.get_backend() == "nccl" and (rank, world) == (0, 1)
dev = torch.device("cuda", local_rank)
n, B = 203, 777
mygraph = generate_gnm(n, 900, seed=3)
g = ops.DeviceGraph(build_csr(mygraph, num_nodes=n, if_bidirectional=False), dev)
xs = ops.rand_spins(B, n, seed=5, device=dev)
obj = ops.maxcut_obj(g, xs)
best, owner, bx = rdist.global_best(obj, xs, want_solution=True)     # C1 all_reduce(MAX) + C2 broadcast, on device
i = int(obj.argmax())
ok = int(best) == int(obj.max()) and int(owner) == 0 and torch.equal(bx, xs[i]) and bx.is_cuda
# the float (bidirectional) form and the no-solution form
bf, of, _ = rdist.global_best(obj.to(torch.float32) / 2)
ok = ok and float(bf) == float(obj.max()) / 2 and int(of) == 0
# C2 on device tensors = candidate messages + SUM all-reduce, no host read: with the global index, a callable row source, f32 rows
b4, o4, x4, g4 = rdist.global_best(obj, xs, want_solution=True, env_offset=1000)
ok = ok and int(b4) == int(obj.max()) and int(g4) == 1000 + i and torch.equal(x4, xs[i]) and x4.dtype == xs.dtype
b5, o5, x5, g5 = rdist.global_best(obj, lambda li: xs[li], want_solution=True, env_offset=7, num_nodes=n)
ok = ok and int(g5) == 7 + i and torch.equal(x5, xs[i].bool())
b6, o6, x6 = rdist.global_best(obj, xs.float(), want_solution=True)
ok = ok and x6.dtype == torch.float32 and torch.equal(x6, xs[i].float())
b7, o7, x7, g7 = rdist.global_best(obj, None, env_offset=5)
ok = ok and x7 is None and int(g7) == 5 + i
# the lean form (VERDICT r5 item 1): one rls_best_key launch + one all_reduce per exchange, unpack / flag check deferred
ex = rdist.BestExchange(dev, depth=4)
for rep in range(9):                                                   # (more than the ring's depth: slots are reused)
    key = ex.exchange(obj + rep)
o, w = ex.unpack(key)
ok = ok and int(o) == int(obj.max()) + 8 and int(w) == 0 and int(ex.last_index[0]) == i
ex.check()
# ... and captured in a hipGraph with the work before it, as bench.py's regions do at N > 1
obj2 = obj.clone()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    obj2.add_(1)
    k2 = ex.exchange(obj2)
gr.replay(); gr.replay()
torch.cuda.synchronize()
ok = ok and int(ex.unpack(k2)[0]) == int(obj.max()) + 2
ex.exchange(torch.full((4,), 1 << 50, dtype=torch.int64, device=dev))  # outside the key's range: the sticky flag, read lazily
try:
    ex.check(); ok = False
except ValueError:
    pass
# a rank WITHOUT envs joins every collective (env_shard gives some ranks nothing when B < world): here it is the only rank, so the
# reduced key is the empty key -- zeros come back and the sticky flag (bit 1: "no rank had an env") trips at the lazy check
be, oe, xe, ge = rdist.global_best(torch.empty(0, dtype=torch.int64, device=dev), torch.empty((0, n), dtype=torch.bool, device=dev),
                                   want_solution=True, env_offset=0)
ok = ok and xe.shape == (n,) and not bool(xe.any()) and int(ge) == 0
try:
    rdist._site(dev, None).check(); ok = False
except ValueError:
    pass
t = torch.arange(8, dtype=torch.int64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
ok = ok and t.tolist() == list(range(8))
dist.barrier(device_ids=[local_rank])
dist.destroy_process_group()
print(json.dumps({"ok": bool(ok), "best": int(best), "argmax": i}))
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(RLS_FORCE_PG="1", RLS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
             HSA_ENABLE_IPC_MODE_LEGACY="0")
    return e


def test_global_best_over_one_rank_rccl_group():
    p = subprocess.run([sys.executable, "-c", CHILD], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["ok"] is True


def test_bench_under_one_rank_rccl_group():
    """bench.py with an initialised RCCL group: barrier(device_ids), the all_reduce(MAX) of the region times and the
    in-region global_best all run; the JSON line is the last stdout line."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--repeats", "3",
                        "--envs-per-gpu", "4096", "--no-cpu-baseline", "--no-config5", "--no-configs"], env=_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["repeats"] == 3 and len(out["ms_per_step_all"]) == 3
    assert out["value"] > 0 and 0 < out["roofline"]["frac"] < 1


def test_bench_line_carries_the_exchange_probe():
    """The N = 1 line's `exchange_probe`: a child process (started before bench.py touches the GPU) times the exchange on a
    1-rank RCCL group -- eager, alone, and inside a hipGraph."""
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RLS_FORCE_PG"):
        e.pop(k, None)
    e["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--repeats", "3",
                        "--envs-per-gpu", "4096", "--no-cpu-baseline", "--no-config5", "--no-configs"], env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    pr = out["exchange_probe"]
    assert "error" not in pr, pr
    assert pr["backend"] == "nccl" and pr["check"] == "ok" and 0 < pr["exchange_us"] < 1000 and pr["exchange_us_single"] > 0
    assert pr["graph_capturable"] in (True, False)
    assert out["roofline"]["hbm_only"]["frac"] > 0 and "bound_detail" in out["roofline"]


@pytest.mark.parametrize("launch", ["self-spawn", "torchrun"])
def test_bench_two_ranks_sharing_the_gpu(launch):
    """The driver's N > 1 command line as far as ONE GPU can run it: `bench.py --gpus 2` (started by itself, or under
    torch.distributed.run as the contract launches it) with --share-gpu -- both ranks on cuda:0, gloo between them.  Real
    kernels, real shard offsets (rank 1's envs are keyed 8192 ..), the exchange at the end of every timed region, the MAX over
    ranks, the end-of-run parity check on both ranks and rank 0's JSON as the last stdout line.  (The RCCL calls themselves are
    the 1-rank group's tests above; a number printed in this mode is not a scaling measurement and says so.)"""
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RLS_FORCE_PG"):
        e.pop(k, None)
    e["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    args = ["--gpus", "2", "--share-gpu", "--steps", "20", "--warmup", "5", "--repeats", "3", "--envs-per-gpu", "8192",
            "--no-cpu-baseline", "--no-config5"]
    if launch == "self-spawn":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *args]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), *args]
    p = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, p.stdout[-1500:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and len(out["ms_per_step_all"]) == 3 and out["scaling"] == "weak"
    assert out["config"]["envs_per_gpu"] == 8192 and out["config"]["global_envs"] == 16384
    assert "TEST MODE" in out["config"]["parallelism"] and out["value"] > 0
    # value = the units ALL ranks processed / the slowest rank's region
    assert abs(out["value"] - 16384 * 20 / (out["ms_per_step"] * 20 * 1e-3)) / out["value"] < 1e-6


def test_bench_eight_ranks_sharing_the_gpu_strong_scaling_line_explains_itself():
    """the 8-rank wiring once on the one GPU -- `--gpus 8 --share-gpu --global-envs 16384` (strong scaling: 2048
    envs per rank): eight rendezvous, eight shard offsets, and a line that can attribute a shortfall: every rank's kernel time
    per step (HIP events), its exchange time, its region time, MAX - MIN of the region times; the shards and the MCPG chain ids of
    the same batch cover it exactly once."""
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RLS_FORCE_PG"):
        e.pop(k, None)
    e["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-gpu", "--steps", "20", "--warmup", "5", "--repeats", "3",
           "--global-envs", "16384", "--no-cpu-baseline", "--no-config5"]
    p = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    out = json.loads([ln for ln in p.stdout.strip().splitlines() if ln.startswith("{") and '"metric"' in ln][-1])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["config"]["global_envs"] == 16384
    assert abs(out["value"] - 16384 * 20 / (out["ms_per_step"] * 20 * 1e-3)) / out["value"] < 1e-6
    rb = out["rank_breakdown"]
    assert [r["rank"] for r in rb["ranks"]] == list(range(8))
    assert [(r["env_offset"], r["envs"]) for r in rb["ranks"]] == [(2048 * k, 2048) for k in range(8)]
    assert all(r["kernel_us_per_step"] > 0 and r["exchange_us"] > 0 and r["region_ms"] > 0 for r in rb["ranks"])
    assert rb["skew_ms"] >= 0 and len(rb["skew_ms_all_regions"]) == 3
    # the region that counts is the slowest rank's
    assert abs(max(r["region_ms"] for r in rb["ranks"]) - out["ms_per_step"] * 20) < 1e-6 * out["ms_per_step"] * 20 + 1e-9
    cov = out["shard_cover"]
    assert cov["envs"] == "each exactly once" and cov["global"] == 16384
    assert "configs" not in out and "cpu_baseline" not in out          # rank-0, N = 1 legs only


def test_sharded_search_loop_example_is_rank_count_invariant():
    """examples/sharded_local_search.py -- the reference's search loop (env_MCPG.py:407-493) on the sharded classes -- with one
    rank and with two ranks sharing the GPU: the same best cut AND the same solution string."""
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RLS_FORCE_PG"):
        e.pop(k, None)
    e["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    script = os.path.join(ROOT, "examples", "sharded_local_search.py")
    args = ["--nodes", "300", "--edges", "1500", "--num-sims", "512", "--num-iter1", "4", "--num-iter0", "2", "--ls-iters", "8", "--seed", "3"]
    outs = []
    for world in (1, 2):
        cmd = [sys.executable, script, *args] if world == 1 else \
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", str(_free_port()), script, *args, "--share-gpu"]
        p = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
        assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
        outs.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"x_str"' in ln][-1]))
    one, two = outs
    assert (one["ranks"], two["ranks"]) == (1, 2)
    assert one["best"] == two["best"] == one["cut_of_x"] == two["cut_of_x"] and one["x_str"] == two["x_str"]
    assert one["best"] > 1500 * 0.6                                   # and the search found something: well above a random cut


def test_s2v_episode_example_runs():
    """examples/s2v_episodes.py: S2V-DQN's env_args (train_S2V.py:37-47: irreversible spins) through ising_env.make and the
    agent's action masking (dqn.py:254, 419-430), asserting inside that an episode ends after every spin was flipped once and that
    the DENSE rewards sum to the cut change."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "s2v_episodes.py"), "--nodes", "30", "--episodes", "3"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "s2v_episodes: ok", r.stdout[-3000:]

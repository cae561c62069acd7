#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the MaxCut gym step (K4, rls_maxcut_step) on a
Gset-G22-sized graph with 2^16 parallel envs per GPU, next-state emitted into a rollout ring
(the PPO `obs[t+1] = next_obs` pattern), 1-byte spins in and out.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks itself (a fresh
`python -m torch.distributed.run` child, started before this process touches the GPU), relays rank 0's JSON line
as its last stdout line and exits with the children's return code.  `--dry-run` exercises the same launch, shard
and exchange wiring on CPU (gloo) without a kernel.

One "step" = one K4 pass over the whole batch: for every env flip node a_b, compute the cut gain
from the action node's CSR row, update obj, write reward, and emit the next state.  Algorithmic
HBM bytes per env-step = 2N + 20.  Envs are sharded over ranks with no
data-path collective (weak scaling: 2^16 envs per GPU); the episode-boundary best-objective
exchange (8-byte RCCL all-reduce) runs once at the end of every timed region when N > 1.

Timing: W warm-up steps, then R (`--repeats`, default 5) timed regions of EXACTLY K steps each, every region
bracketed by barrier + synchronize on both sides; a rank's time runs from the release of the opening barrier to its own
synchronize after the last step and the exchange, and the region counts as the MAX over ranks (the closing barrier
itself -- a second collective latency that no step waits for -- is not inside any rank's interval).  The line reports
the MEDIAN region (`ms_per_step`, `value`) and all of them (`ms_per_step_all`).

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events on the launch stream;
`cpu_baseline` is the C oracle (reference algorithm: flip + full objective re-evaluation, OpenMP)
timed on the host cores on a bounded sample (rank 0, N = 1 only); `cpu_baseline_ref_shaped` is the reference's
own op chain in torch-CPU ops (oracle/oracle_torch.py).  `config5_shard` is BASELINE config #5's per-GPU shard
(G70-sized graph, 2^17 envs per GPU) measured the same way after the headline (skip with --no-config5); as a
headline of its own: `--gset 70 --envs-per-gpu 131072`.
"""
import argparse
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
METRIC = "env-steps/sec (all instances) on Gset G22 MaxCut; achieved HBM GB/s % peak"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--envs-per-gpu", type=int, default=1 << 16)
    ap.add_argument("--gset", type=int, default=22, help="Gset id whose (n, m) the graph has (22 = headline)")
    ap.add_argument("--slots", type=int, default=8, help="rollout ring depth (slots of B*N bytes)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config5", action="store_true", help="skip the secondary G70 / 2^17-envs-per-GPU measurement")
    ap.add_argument("--via-env", action="store_true",
                    help="time the drop-in class surface, EnvMaxcutGym.step(action, out=slot) with 1-byte spins, instead of "
                         "the pre-validated C-ABI launcher (same kernel, plus the Python / ctypes path of the class)")
    ap.add_argument("--graph", choices=("auto", "on", "off"), default="auto",
                    help="launcher mode: enqueue each timed region of --steps launches as ONE hipGraph (captured before the region, "
                         "replayed inside it).  auto = on when --steps <= 100, where the first launch's latency is a visible share of "
                         "the region; the kernels, buffers and the end-of-run parity check are the same either way")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST MODE for boxes with one GPU: every rank runs on cuda:0 and the ranks talk over gloo (keys and timings "
                         "staged through the host).  Exercises the whole N > 1 path -- shard offsets, per-region exchange, MAX over "
                         "ranks, rank 0's JSON line -- with real kernels; the number it prints is not a scaling measurement and the "
                         "line says so")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU only (gloo): launch, shard and exchange wiring of the N-rank run, no kernels")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------- #
# self-spawn: `python bench.py --gpus N` without torchrun
# --------------------------------------------------------------------------- #
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_spawn(a, argv):
    """Parent of an N-rank run.  Nothing here may touch the GPU (the children are fresh processes; a process that
    has initialised HIP must never exec or be replaced)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    rest = []
    for ln in p.stdout.splitlines():
        s = ln.strip()
        if s.startswith("{") and '"metric"' in s:
            try:
                json.loads(s)
                line = s
                continue
            except ValueError:
                pass
        rest.append(ln)
    if rest:
        print("\n".join(rest), file=sys.stderr, flush=True)
    if line is None:
        print(f"bench.py: the {a.gpus}-rank child run printed no result line (rc={p.returncode})", file=sys.stderr)
        raise SystemExit(p.returncode or 1)
    print(line, flush=True)
    raise SystemExit(p.returncode)


# --------------------------------------------------------------------------- #
# CPU baselines (rank 0, N = 1 only; outside every timed region)
# --------------------------------------------------------------------------- #
def cpu_baseline(graph_arr, n, seconds):
    """The reference's env step on the host cores: flip + full cut re-evaluation per env
    (oracle/oracle.c: orc_step_u8, the algorithm of envs/env_PPO.py:92-121), OpenMP over envs."""
    import numpy as np
    from oracle import oracle_c as oc
    from oracle import oracle_np as onp
    eu, ev = onp.stored_edges(graph_arr, False)
    Bs = 4096
    rng = np.random.RandomState(0)
    xs = rng.randint(0, 2, size=(Bs, n)).astype(np.uint8)
    last = oc.maxcut_obj(xs, eu, ev, 0)
    acts = rng.randint(0, n, size=(64, Bs)).astype(np.int64)
    oc.step_u8(xs, acts[0], eu, ev, 0, last)  # warm-up (thread pool, caches)
    steps, t0 = 0, time.perf_counter()
    while True:
        oc.step_u8(xs, acts[steps % 64], eu, ev, 0, last)
        steps += 1
        el = time.perf_counter() - t0
        if el >= seconds or steps >= 100000:
            break
    return {"value": Bs * steps / el, "unit": "env-steps/s", "cores": oc.num_threads(), "kind": "port",
            "sample": f"{steps} steps x {Bs} envs of the same graph in {el:.1f} s; C/OpenMP restatement of "
                      f"env_PPO.step (flip + full cut re-evaluation over E={len(eu)} edges)"}


def cpu_baseline_ref_shaped(graph_arr, n, seconds):
    """SURVEY.md section 8d form (i): the reference's env_PPO.step op chain itself (Python loop over envs + three int64
    [B, E'] index tensors + two advanced-index gathers, envs/env_PPO.py:92-121) in torch-CPU ops on all host cores
    (oracle/oracle_torch.py, pinned on the reference's trace).  B is reduced so that the 24*B*E' bytes of index
    tensors fit comfortably in host memory."""
    import numpy as np
    import torch as th
    from oracle.oracle_torch import PPOEnvRefShaped
    cores = os.cpu_count() or 1
    th.set_num_threads(cores)
    Bs = 1024
    rng = np.random.RandomState(0)
    env = PPOEnvRefShaped(graph_arr, n, Bs, 10 ** 9, False)
    env.reset_to(rng.randint(0, 2, size=(Bs, n)).astype(bool))
    acts = th.from_numpy(rng.randint(0, n, size=(16, Bs)).astype(np.int64))
    env.step(acts[0])
    steps, t0 = 0, time.perf_counter()
    while True:
        env.step(acts[steps % 16])
        steps += 1
        el = time.perf_counter() - t0
        if el >= seconds or steps >= 10000:
            break
    return {"value": Bs * steps / el, "unit": "env-steps/s", "cores": cores, "kind": "ref-shaped",
            "sample": f"{steps} steps x {Bs} envs of the same graph in {el:.1f} s; torch-CPU restatement of env_PPO.step in "
                      f"the reference's shape (per-env Python loop, int64 [B,E'] index gathers, {th.get_num_threads()} threads)"}


def pmc_traffic_per_launch(envs, nodes, slots):
    """HBM bytes per k_maxcut_step launch from the committed rocprofv3 PMC summary (separate
    FETCH_SIZE / WRITE_SIZE passes of this same command, gfx950-corrected; tools/summarize_prof.py).
    PMC counters cannot be read from inside the process, so this is the profiled figure, not live.
    The summary holds one entry per kernel instantiation the profiled command launched; tools/summarize_prof.py tags
    the step kernel's entries with the workload they were measured on (envs per GPU, nodes, ring slots -- from the
    profiled run's own JSON line).  Only an entry whose workload is EXACTLY this run's is reported; a run with other
    --envs-per-gpu / --gset / --slots gets None (null in the line), never another configuration's traffic."""
    import glob
    best = None
    want = {"envs": int(envs), "nodes": int(nodes), "slots": int(slots)}
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            d = json.load(open(p))
            for k, v in d.get("kernels", {}).items():
                hb = v.get("hbm_bytes_per_launch")
                if "k_maxcut_step<unsigned char" in k and hb and v.get("workload") == want:
                    best = (hb, f"{os.path.basename(p)}: {k.replace('rls::', '')}, grid {v.get('grid')}")
        except Exception:
            pass
    return best


# --------------------------------------------------------------------------- #
# dry run: the N-rank wiring on CPU
# --------------------------------------------------------------------------- #
def dry_run(a):
    import torch
    import torch.distributed as dist
    from rlsolver_amd import dist as rdist
    rank, local_rank, world = rdist.init_from_env(backend="gloo")
    B = a.envs_per_gpu
    env_offset = rank * B
    off, cnt = rdist.env_shard(world * B, rank, world)
    assert (off, cnt) == (env_offset, B), (off, cnt, env_offset, B)
    # a stand-in objective vector whose global maximum sits in a known shard: env id e scores (e * 7919) % 1009
    ids = torch.arange(env_offset, env_offset + min(B, 4096), dtype=torch.int64)
    obj = (ids * 7919) % 1009
    xs = ((ids[:, None] + torch.arange(16)[None, :]) % 3 == 0)
    best, owner, bx = rdist.global_best(obj, xs, want_solution=True)
    mine = {"rank": rank, "local_rank": local_rank, "env_offset": env_offset, "envs": B,
            "local_best": int(obj.max())}
    if world > 1:
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        dist.barrier()
        dist.destroy_process_group()
    else:
        allr = [mine]
    if rank == 0:
        print(json.dumps({"metric": METRIC, "dry_run": True, "n_gpus": world, "ranks": allr,
                          "global_best": int(best), "owner": int(owner), "best_x": [int(v) for v in bx.tolist()]}),
              flush=True)


# --------------------------------------------------------------------------- #
# the timed workload
# --------------------------------------------------------------------------- #
def measure(a, gset, B, steps, warmup, repeats, dev, rank, local_rank, world, via_env=False, verify=True):
    """W warm-up steps, then `repeats` regions of exactly `steps` K4 launches.  Returns a dict with the per-region
    wall times (MAX over ranks) and HIP-event kernel times of this rank."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from rlsolver_amd import dist as rdist
    from rlsolver_amd import ops
    from rlsolver_amd.graph import build_csr, load_gset

    mygraph, n, is_real = load_gset(gset, os.path.join(ROOT, "data", "gset"))
    csr = build_csr(mygraph, num_nodes=n, if_bidirectional=False)
    g = ops.DeviceGraph(csr, dev)
    N, S = n, a.slots
    env_offset = rank * B

    ring = torch.empty((S, B, N), dtype=torch.bool, device=dev)
    ops.rand_spins(B, N, seed=0, device=dev, env_offset=env_offset, out=ring[0])
    obj = ops.maxcut_obj(g, ring[0]).to(torch.int32)
    reward = torch.empty(B, dtype=torch.float32, device=dev)
    A = 64
    actions = [ops.rand_actions(B, N, seed=1, step=s, device=dev, env_offset=env_offset) for s in range(A)]
    slots = [ring[s] for s in range(S)]

    # one pre-validated launcher per (slot, action vector) pair of the cycle: the timed loop is then a
    # bare C-ABI call per step (the ring and the action pool are fixed buffers)
    period = S * A // math.gcd(S, A)
    if via_env:
        import types
        from rlsolver_amd.envs.env_PPO import EnvMaxcut as EnvMaxcutGym
        env = EnvMaxcutGym(types.SimpleNamespace(num_nodes=N, num_envs=B, num_steps=10 ** 9), mygraph=mygraph, device=dev,
                           spin_dtype=torch.bool, reuse_buffers=True)
        env.xs, env._obj = slots[0], obj        # same initial state as the launcher mode

        def step(t):
            env.step(actions[t % A], out=slots[(t + 1) % S])
    else:
        launchers = [ops.maxcut_step_launcher(g, slots[t % S], slots[(t + 1) % S], actions[t % A], obj, reward)
                     for t in range(period)]

        def step(t):
            launchers[t % period]()

    t = 0
    for _ in range(warmup):
        step(t)
        t += 1

    use_pg = dist.is_initialized()
    # (auto needs at least one eager warm-up launch: a kernel's first launch loads its code object, which a capture must not do)
    use_graph = (not via_env) and (a.graph == "on" or (a.graph == "auto" and steps <= 100 and warmup > 0))
    graphs = []
    if use_graph:
        # one hipGraph per timed region: region r runs steps t_r .. t_r + steps - 1 of the ring / action cycle (capture enqueues
        # nothing: the launchers resolve torch's current stream at every call, which is the capture stream here)
        try:
            torch.cuda.synchronize(dev)
            tc = t
            for _ in range(repeats):
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for k in range(steps):
                        step(tc + k)
                graphs.append(gr)
                tc += steps
        except Exception as ex:          # a capture that fails must not fail the measurement: the eager launcher loop is the same work
            if a.graph == "on":
                raise
            print(f"bench.py: hipGraph capture failed ({type(ex).__name__}: {ex}); timing the eager launcher loop", file=sys.stderr)
            graphs, use_graph = [], False
            torch.cuda.synchronize(dev)

    nccl = use_pg and dist.get_backend() == "nccl"

    def barrier():
        if use_pg:
            dist.barrier(device_ids=[local_rank]) if nccl else dist.barrier()

    if use_pg:   # the first collective builds the communicator (16 ms on a 1-rank RCCL group): not part of any region
        rdist.global_best(obj)
        barrier()
    wall, kern = [], []
    for rep in range(repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        e0.record()
        if use_graph:
            graphs[rep].replay()
            t += steps
        else:
            for _ in range(steps):
                step(t)
                t += 1
        e1.record()
        if use_pg:  # episode boundary: best objective over all shards (C1, 8 bytes over RCCL)
            rdist.global_best(obj)
            e2 = torch.cuda.Event()
            e2.record()
        else:
            e2 = e1
        while not e2.query():     # busy-poll the last event, then synchronize (which returns at once): a blocking wait adds
            pass                  # its wake-up latency (tens of us) to a region that may be under a millisecond long
        torch.cuda.synchronize(dev)
        wall.append(time.perf_counter() - t0)     # this rank's region; the slowest rank's is what counts (MAX below)
        barrier()
        torch.cuda.synchronize(dev)
        kern.append(e0.elapsed_time(e1) * 1e-3 / max(steps, 1))

    if use_pg:
        tt = torch.tensor(wall, dtype=torch.float64, device=dev if nccl else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = [float(v) for v in tt.tolist()]

    if verify:  # size-independent parity property at full size: incremental obj == recomputed
        final = slots[t % S]
        if not torch.equal(ops.maxcut_obj(g, final).to(torch.int32), obj):
            raise SystemExit("PARITY FAILURE: incremental objective != recomputed objective")

    res = {"wall": wall, "kernel_s": kern, "N": N, "E": len(mygraph), "is_real": is_real, "B": B, "graph": use_graph,
           "graph_arr": np.asarray(mygraph, dtype=np.int64)}
    return res          # the ring and the launchers die with this frame: the next workload gets the memory back


def summarize(res, steps, world):
    B, N = res["B"], res["N"]
    el = statistics.median(res["wall"])
    ks = statistics.median(res["kernel_s"])
    bytes_per_launch = B * (2 * N + 20)
    achieved = bytes_per_launch / ks / 1e9
    return {"value": world * B * steps / el, "ms_per_step": el / steps * 1e3,
            "ms_per_step_all": [w / steps * 1e3 for w in res["wall"]],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                         "kernel": "k_maxcut_step<u8, emit>", "us_per_launch": ks * 1e6,
                         "us_per_launch_all": [k * 1e6 for k in res["kernel_s"]],
                         "algorithmic_bytes_per_launch": bytes_per_launch}}


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(a, argv)          # never returns
    if int(os.environ.get("WORLD_SIZE", "1")) != a.gpus:      # before any rendezvous: a mismatch must not hang
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}: launch with torch.distributed.run "
                         f"--nproc-per-node {a.gpus} (or plain `python bench.py --gpus {a.gpus}`, which starts the ranks itself)")
    if a.dry_run:
        return dry_run(a)

    import torch
    import torch.distributed as dist
    from rlsolver_amd import dist as rdist

    rank, local_rank, world = rdist.init_from_env(backend="gloo" if a.share_gpu else None)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if a.share_gpu:
        local_rank = 0                                  # test mode: all ranks on the one GPU of the box
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    R = max(1, a.repeats)

    res = measure(a, a.gset, a.envs_per_gpu, a.steps, a.warmup, R, dev, rank, local_rank, world,
                  via_env=a.via_env, verify=not a.no_verify)
    torch.cuda.empty_cache()
    res5 = None
    if not a.no_config5 and not (a.gset == 70 and a.envs_per_gpu == 131072):
        # BASELINE config #5's per-GPU shard, measured the same way (never mixed into `value`)
        res5 = measure(a, 70, 131072, max(1, min(a.steps, 200)), min(a.warmup, 20), R, dev, rank, local_rank, world,
                       verify=not a.no_verify)

    out = None
    if rank == 0:
        B, N = res["B"], res["N"]
        s = summarize(res, a.steps, world)
        out = {
            "metric": METRIC,
            "value": s["value"],
            "unit": "env-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": s["ms_per_step"],
            "repeats": R, "ms_per_step_all": s["ms_per_step_all"],
            "timing": f"median of {R} regions of exactly {a.steps} steps, each bracketed by barrier + synchronize, MAX over ranks",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # the graph: the real Gset file when data/gset/gset_<k>.txt is present, else the G(n, m) stand-in of its size; spins
            # and actions are drawn either way
            "dtype": "u8", "data": "real Gset graph, synthetic spins and actions" if res["is_real"] else "synthetic",
            "config": {"workload": f"Gset G{a.gset}{'' if res['is_real'] else '-sized G(n,m) stand-in'} MaxCut "
                                   f"(N={N}, E={res['E']}), {B} envs per GPU, K4 gym step emitting the next "
                                   f"state into a {a.slots}-slot rollout ring, uniform random actions"
                                   + ("; through EnvMaxcutGym.step(action, out=slot)" if a.via_env else "")
                                   + ("; BASELINE config #5 shard (2^20 envs over 8 GPUs = 131072 per GPU)"
                                      if (a.gset == 70 and B == 131072) else ""),
                       "entry": "EnvMaxcutGym.step" if a.via_env else ("rls_maxcut_step launcher, each timed region of "
                                                                        f"{a.steps} launches enqueued as one hipGraph" if res["graph"]
                                                                        else "rls_maxcut_step launcher"),
                       "num_nodes": N, "num_edges": res["E"], "envs_per_gpu": B, "global_envs": world * B, "slots": a.slots,
                       "parallelism": f"env-shard x{world}" + (" (TEST MODE --share-gpu: all ranks on one GPU over gloo; not a scaling "
                                                                "measurement)" if a.share_gpu else "")},
            "roofline": s["roofline"],
        }
        tr = pmc_traffic_per_launch(B, N, a.slots)
        if tr is not None:
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = tr[0], f"profiles/{tr[1]} (rocprofv3 --pmc)"
        if res5 is not None:
            st5 = max(1, min(a.steps, 200))
            s5 = summarize(res5, st5, world)
            tr5 = pmc_traffic_per_launch(res5["B"], res5["N"], a.slots)
            if tr5 is not None:
                s5["roofline"]["traffic"], s5["roofline"]["traffic_source"] = tr5[0], f"profiles/{tr5[1]} (rocprofv3 --pmc)"
            out["config5_shard"] = {
                "workload": f"Gset G70{'' if res5['is_real'] else '-sized G(n,m) stand-in'} MaxCut (N={res5['N']}, "
                            f"E={res5['E']}), 131072 envs per GPU = 2^20 over 8 GPUs (BASELINE config #5), same K4 loop",
                "value": s5["value"], "unit": "env-steps/s", "steps": st5, "ms_per_step": s5["ms_per_step"],
                "global_envs": world * 131072, "roofline": s5["roofline"]}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(res["graph_arr"], N, a.cpu_seconds)
            out["cpu_baseline_ref_shaped"] = cpu_baseline_ref_shaped(res["graph_arr"], N, max(3.0, a.cpu_seconds / 2))

    if dist.is_initialized():
        dist.barrier(device_ids=[local_rank]) if dist.get_backend() == "nccl" else dist.barrier()
        dist.destroy_process_group()
    if rank == 0:   # the JSON line is the last thing on stdout (RCCL prints its banner at init/teardown)
        sys.stdout.flush()
        try:   # RCCL writes its banner through C stdio; drain it so the JSON really is the last line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

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
METRIC = "env-steps/sec (all instances) on Gset G22 MaxCut; achieved HBM GB/s % peak"     # BASELINE.json's, for --gset 22
MALL_BYTES = 256 << 20   # Infinity Cache


def metric_for(gset):
    """BASELINE.json's metric string names G22; a headline on another graph size (--gset 70 = config #5) says so."""
    return METRIC.replace("G22", f"G{gset}")


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
    ap.add_argument("--no-hbm-only", action="store_true", help="skip the nontemporal-store repeat of the headline loop (roofline.hbm_only)")
    ap.add_argument("--no-config5", action="store_true", help="skip the secondary G70 / 2^17-envs-per-GPU measurement")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` block: BASELINE configs #2 (dREINFORCE-shaped local search), #3 (BA MCMC) and #4 (TSP) at "
                         "their stated sizes, each a bounded wall-clock measurement after the headline (never mixed into `value`)")
    ap.add_argument("--global-envs", type=int, default=0,
                    help="STRONG scaling: this many envs in total, split over the ranks (rlsolver_amd.dist.env_shard) -- "
                         "`--gset 70 --global-envs 1048576 --gpus 8` is BASELINE config #5 as the headline; 0 = --envs-per-gpu each (weak)")
    ap.add_argument("--via-env", action="store_true",
                    help="time the drop-in class surface, EnvMaxcutGym.step(action, out=slot) with 1-byte spins, instead of "
                         "the pre-validated C-ABI launcher (same kernel, plus the Python / ctypes path of the class)")
    ap.add_argument("--graph", choices=("auto", "on", "off"), default="auto",
                    help="launcher mode: enqueue each timed region of --steps launches as ONE hipGraph (captured before the region, "
                         "replayed inside it).  auto = on when --steps <= 100, where the first launch's latency is a visible share of "
                         "the region; the kernels, buffers and the end-of-run parity check are the same either way")
    ap.add_argument("--graph-exchange", choices=("auto", "off"), default="auto",
                    help="N > 1 over RCCL: capture the region's exchange in the region's hipGraph (auto) or launch it eagerly (off)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST MODE for boxes with one GPU: every rank runs on cuda:0 and the ranks talk over gloo (keys and timings "
                         "staged through the host).  Exercises the whole N > 1 path -- shard offsets, per-region exchange, MAX over "
                         "ranks, rank 0's JSON line -- with real kernels; the number it prints is not a scaling measurement and the "
                         "line says so")
    ap.add_argument("--exchange-probe", action="store_true",
                    help="child mode of the N = 1 line: time the episode-boundary exchange (rls_best_key + 8-byte all_reduce) on a "
                         "1-rank RCCL group (RLS_FORCE_PG=1), print one JSON line and exit")
    ap.add_argument("--no-exchange-probe", action="store_true")
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
# exchange probe: what the N > 1 line's per-region exchange costs, measured on the 1-rank RCCL group a 1-GPU box can form
# --------------------------------------------------------------------------- #
def exchange_probe_child(a):
    """Runs in a fresh process (RLS_FORCE_PG=1, WORLD_SIZE=1): the exchange of one timed region -- BestExchange.exchange =
    one rls_best_key launch + one 8-byte all_reduce(MAX) over RCCL -- eager and inside a hipGraph, timed with HIP events on
    the launch stream.  A 1-rank group has no peer: this is the launch + collective-kernel floor, not xGMI latency."""
    import torch
    import torch.distributed as dist
    from rlsolver_amd import dist as rdist
    rank, local_rank, world = rdist.init_from_env()
    dev = torch.device("cuda", local_rank)
    B = a.envs_per_gpu
    obj = (torch.arange(B, dtype=torch.int64, device=dev) * 7919 % 10007).to(torch.int32)
    ex = rdist.BestExchange(dev)
    for _ in range(20):                         # communicator set-up + code objects
        ex.exchange(obj)
    torch.cuda.synchronize(dev)
    out = {"backend": dist.get_backend(), "world": world, "envs": B, "ops_per_exchange": "1 launch (rls_best_key) + 1 all_reduce(MAX, 8 B)"}

    def timed(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        e0.record()
        fn()
        e1.record()
        th = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) * 1e3 / n, th * 1e6 / n

    n = 200
    runs = [timed(lambda: [ex.exchange(obj) for _ in range(n)], n) for _ in range(5)]
    out["exchange_us"] = statistics.median(r[0] for r in runs)            # device time per exchange, back to back
    out["exchange_host_us"] = statistics.median(r[1] for r in runs)       # host time to enqueue one
    # one exchange on an idle stream (what a region's end pays: nothing to overlap with)
    singles = []
    for _ in range(30):
        singles.append(timed(lambda: ex.exchange(obj), 1)[0])
    out["exchange_us_single"] = statistics.median(singles)
    key = ex.exchange(obj)
    o, w = ex.unpack(key)
    ex.check()
    out["check"] = "ok" if (int(o) == int(obj.max()) and int(w) == 0) else "BROKEN"
    try:            # the same inside a hipGraph (torch's NCCL ops are capturable): what bench.py does at N > 1
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                ex.exchange(obj)
        g.replay()
        torch.cuda.synchronize(dev)
        out["exchange_us_in_graph"] = statistics.median(timed(g.replay, 20)[0] for _ in range(5))
        out["graph_capturable"] = True
    except Exception as e:
        out["graph_capturable"] = False
        out["graph_error"] = f"{type(e).__name__}: {e}"[:200]
    torch.cuda.synchronize(dev)
    dist.barrier(device_ids=[local_rank])
    dist.destroy_process_group()
    sys.stdout.flush()
    print("EXCHANGE_PROBE " + json.dumps(out), flush=True)


def exchange_probe_parent(a):
    """Called by the N = 1 run BEFORE it touches the GPU: start the probe as a child process, wait for it, return its line
    (or why there is none -- the probe never fails the benchmark)."""
    env = dict(os.environ)
    env.update({"RLS_FORCE_PG": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(_free_port())})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__), "--exchange-probe", "--envs-per-gpu", str(a.envs_per_gpu)]
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        return {"error": "probe child timed out"}
    for ln in p.stdout.splitlines():
        if ln.startswith("EXCHANGE_PROBE "):
            out = json.loads(ln[len("EXCHANGE_PROBE "):])
            tr = exchange_trace()
            if tr is not None:
                out["kernel_trace"] = tr
            return out
    return {"error": f"probe child printed no line (rc={p.returncode})", "stderr_tail": p.stderr[-300:]}


def exchange_trace():
    """Launches per exchange from the committed rocprofv3 kernel trace of this same probe (profiles/rNN_exchange.json: written by
    tools/timing/exchange_trace.sh; a trace cannot be taken from inside the process)."""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_exchange.json")), reverse=True):
        try:
            d = json.load(open(p))
            return {"launches_per_exchange": d["launches_per_exchange"], "kernels": d["kernels"], "source": f"profiles/{os.path.basename(p)}"}
        except Exception:
            pass
    return None


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
# BASELINE configs #2, #3, #4 (rank 0, N = 1 only; after the headline, never mixed into `value`)
# --------------------------------------------------------------------------- #
def _profiled(kernel_sub, row_sub, field):
    """A figure that cannot be measured live (SQ / TCC counters), from the newest committed per-kernel table
    (profiles/rNN_kernels.json: tools/kernel_table.py over the round's rocprofv3 passes), or None."""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernels.json")), reverse=True):
        try:
            for grp in json.load(open(p)).get("groups", []):
                for r in (grp if isinstance(grp, list) else grp.get("rows", [grp])):     # (a flat list of row dicts since r02)
                    if kernel_sub in r.get("kernel", "") and row_sub in r.get("row", "") and r.get(field) is not None:
                        return {"value": r[field], "source": f"profiles/{os.path.basename(p)}: {r['kernel']}"}
        except Exception:
            pass
    return None


def _time_calls(fn, iters, warm=2):
    import torch
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def config2_local_search(dev):
    """BASELINE config #2's outer-loop body: dREINFORCE's `local_search_inplace` (envs/env_L2A.py:87-116: 8 noisy
    top-k multi-flip proposals + one greedy sweep) on B = 64 x 1024 rows of a G22-sized graph.  The reference evaluates
    N + 8 full objectives per env per call; a call is ONE pre-pass + ONE fused kernel here."""
    import torch
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    from rlsolver_amd.graph import load_gset
    mygraph, n, is_real = load_gset(22, os.path.join(ROOT, "data", "gset"))
    B = 64 * 1024
    env = EnvMaxcut(mygraph=mygraph, device=dev, num_nodes=n)
    torch.manual_seed(0)
    xs = env.generate_xs_randomly(B)
    vs = env.calculate_obj_values(xs)
    v0 = vs.clone()
    t = _time_calls(lambda i: env.local_search_inplace(xs, vs, num_iters=8, num_spin=8, noise_std=0.3), 30, warm=2)
    ok = bool((vs >= v0).all()) and torch.equal(env.calculate_obj_values(xs), vs)
    if not ok:
        raise SystemExit("PARITY FAILURE (config #2): local_search_inplace lost value or its vs != recomputed objective")
    out = {"workload": f"G22{'' if is_real else '-sized G(n,m) stand-in'} (N={n}, E={len(mygraph)}), {B} envs: "
                       "EnvMaxcut.local_search_inplace(num_iters=8, num_spin=8) -- dREINFORCE's outer-loop body",
           "ms_per_call": t * 1e3, "candidate_evaluations_per_s": B * (n + 8) / t, "env_sweeps_per_s": B / t,
           "unit_note": "the reference performs N + 8 full objective evaluations per env per call (env-steps in its sense)",
           "bound": "valu", "check": "32 calls chained; vs never drops and equals the recomputed objective of xs: ok",
           "mean_gain_over_random": float((vs - v0).float().mean())}
    vf = _profiled("k_maxcut_local_search", "G22", "valu_frac")
    if vf:
        out["valu_issue_frac"] = vf
    return out


def config3_mcpg(dev):
    """BASELINE config #3: BA n = 10^4, m = 5, 2^18 chains = 2048 kept x 128 repeats, num_ls = 8 sweeps in degree-descending
    order (methods/MCPG.py:120-166) + metro_sampling (:88-118) + the best-merge (:376-391).  `num_samples_per_second` is the
    reference's own print (MCPG.py:409-412): kept chains per round time."""
    import numpy as np
    import torch
    from rlsolver_amd import ops
    from rlsolver_amd.graph import generate_ba
    from rlsolver_amd.methods import MCPG as amcpg
    from rlsolver_amd.ops_mcpg_tsp import PackedChains
    n, C, R, num_ls = 10000, 1 << 18, 128, 8
    M, T = C // R, n // 10
    arr = np.asarray(generate_ba(n, 5, seed=5), dtype=np.int64)
    data = amcpg.make_data(n, arr[:, 0], arr[:, 1], dev)
    torch.manual_seed(0)
    probs = torch.full((n,), 0.5, device=dev)
    kept = PackedChains.pack((torch.rand((n, M), device=dev) < 0.5).float())
    chains = PackedChains.empty(n, C, dev)
    t_metro = _time_calls(lambda i: amcpg.metro_sampling_packed(probs, kept, T, num_chains=C, out=chains), 12, warm=2)
    res = {}

    def ls(i):
        res["r"] = amcpg.sampler_func_packed(data, chains, num_ls, M, R)
    t_ls = _time_calls(ls, 12, warm=2)
    vs_good, xs_good, value, _ = res["r"]
    cut = ops.maxcut_obj(data.graph, (xs_good.unpack().t() > 0).contiguous())
    if not (torch.equal(cut.float(), vs_good) and float(vs_good.mean()) > 0.6 * data.num_edges):
        raise SystemExit("PARITY FAILURE (config #3): a kept chain's reported value is not its cut")
    rnd = amcpg.MCPGRound(data, kept.clone(), torch.zeros(M, device=dev), M, R, num_ls)
    best = []

    def step(i):
        best.append(rnd.step(probs)[1])
    t_round = _time_calls(step, 12, warm=2)
    bv = torch.cat(best).cpu()
    if not bool((bv[1:] >= bv[:-1]).all()):
        raise SystemExit("PARITY FAILURE (config #3): the incumbent of the MCPG round got worse")
    out = {"workload": f"BA n={n} m=5 (E={data.num_edges}), {C} chains = {M} kept x {R} repeats, num_ls={num_ls}, T={T} walk rounds",
           "sampler_func_packed_ms": t_ls * 1e3, "chain_sweeps_per_s": C * num_ls / t_ls, "node_updates_per_s": C * num_ls * n / t_ls,
           "metro_sampling_packed_ms": t_metro * 1e3, "proposals_per_s": C * T / t_metro,
           "mcpg_round_ms": t_round * 1e3, "num_samples_per_second": M / t_round, "bound": "valu",
           "check": "every kept chain's value == its recomputed cut; 14 rounds chained, incumbent never worse: ok"}
    # the reference-shaped surface: f32 [N, C] in and out (MCPG.py:88-166 as the training loop calls it)
    try:
        xs = torch.empty((n, C), device=dev)
        for c0 in range(0, C, 1 << 15):
            xs[:, c0:c0 + (1 << 15)] = (torch.rand((n, 1 << 15), device=dev) < 0.5).float()
        out["metro_sampling_f32_ms"] = _time_calls(lambda i: amcpg.metro_sampling(probs, xs, T, dev), 5, warm=1) * 1e3
        out["sampler_func_f32_ms"] = _time_calls(lambda i: amcpg.sampler_func(data, xs, num_ls, M, R, dev), 5, warm=1) * 1e3
        del xs
    except torch.cuda.OutOfMemoryError:
        out["metro_sampling_f32_ms"] = out["sampler_func_f32_ms"] = None
    for key, ksub in (("k7_valu_issue_frac", "valu_frac"), ("k7_wait_any_share", "wait_any_share_of_wave_cycles")):
        vf = _profiled("k_mcpg_local_search_levels", "BA-1e4", ksub)
        if vf:
            out[key] = vf
    return out


def config4_tsp(dev):
    """BASELINE config #4: TSP-100 random Euclidean, 2^16 tours: tour length (K12, envs/env_ISCO.py:346-350) and the swap /
    2-opt delta of every position (K13, :232-296)."""
    import torch
    from rlsolver_amd import ops_mcpg_tsp as mops
    from rlsolver_amd.graph import generate_tsp_coords, tsp_tables
    N, B = 100, 1 << 16
    dist, near, rnd = tsp_tables(generate_tsp_coords(N, 100), K=20)
    d = torch.from_numpy(dist).to(dev)
    K = near.shape[1]
    near32, rnd32 = torch.from_numpy(near.astype("int32")).to(dev), torch.from_numpy(rnd.astype("int32")).to(dev)
    tab8 = mops.tsp_tables8(near32, rnd32)
    thr = K / (K + 1)
    perms = mops.rand_perms(B, N, 3, dev)
    t12 = _time_calls(lambda i: mops.tsp_tour_length(d, perms), 300, warm=5)
    # K13 as the reference runs it (env_ISCO.py:238-335): partners drawn inside the call -- 8N bytes in, 13N out per tour
    t13 = _time_calls(lambda i: mops.tsp_swap_delta_all(d, perms, None, 0.5, nearest=near32, random=rnd32, near_threshold=thr, seed=i, tables8=tab8),
                      300, warm=5)
    sel = torch.roll(perms, 7, 1).contiguous()
    t13s = _time_calls(lambda i: mops.tsp_swap_delta_all(d, perms, sel, 0.5), 100, warm=3)       # the recorded-draw hook: + 8N in
    length = mops.tsp_tour_length(d, perms)
    rel = 0.0
    for other in (torch.roll(perms, 17, 1).contiguous(), torch.flip(perms, [1]).contiguous()):
        rel = max(rel, float(((mops.tsp_tour_length(d, other) - length).abs() / length).max()))
    lr, idx, ban, drawn = mops.tsp_swap_delta_all(d, perms, None, 0.5, nearest=near32, random=rnd32, near_threshold=thr, seed=11, tables8=tab8,
                                                  return_selected=True)
    if not torch.equal(torch.gather(perms, 1, idx), drawn):
        raise SystemExit("PARITY FAILURE (config #4): indices do not point at the drawn partner cities")
    ar = torch.arange(B, device=dev)
    pos = torch.argmin(ban.to(torch.uint8), dim=1)
    okm = ~ban[ar, pos]
    x = perms.clone()
    mops.tsp_apply_swap(x, torch.where(okm, pos, torch.full_like(pos, -1)), idx)
    err = ((mops.tsp_tour_length(d, x) - length) - (-lr[ar, pos] * 0.5)).abs()
    rel13 = float((err[okm] / length[okm]).max())
    if not (rel <= 1e-5 and rel13 <= 2e-5):
        raise SystemExit(f"PARITY FAILURE (config #4): rotation / reversal {rel:.2e}, swap delta vs length difference {rel13:.2e}")
    b12, b13 = B * (8 * N + 4), B * (8 * N + 13 * N)            # K13 = 8N in + 13N out
    return {"workload": f"TSP-{N} uniform Euclidean, {B} tours (int64 [B, N] permutations, f32 distances)",
            "k12_tour_length_us": t12 * 1e6, "k12_tours_per_s": B / t12, "k12_hbm_frac": b12 / t12 / 1e9 / HBM_PEAK_GBS,
            "k13_swap_delta_all_us": t13 * 1e6, "k13_candidate_moves_per_s": B * N / t13, "k13_hbm_frac": b13 / t13 / 1e9 / HBM_PEAK_GBS,
            "k13_with_selected_tensor_us": t13s * 1e6,
            "bound": "hbm", "algorithmic_bytes_per_tour": {"k12": 8 * N + 4, "k13": 21 * N},
            "check": f"lengths invariant under rotation / reversal to {rel:.1e} relative (tolerance 1e-5); partners drawn in the kernel, "
                     f"indices point at them; swap delta == length difference of the applied swap to {rel13:.1e}: ok"}


def extra_configs(dev):
    import torch
    out = {}
    for key, fn in (("config2_local_search", config2_local_search), ("config3_mcpg", config3_mcpg), ("config4_tsp", config4_tsp)):
        t0 = time.perf_counter()
        out[key] = fn(dev)
        torch.cuda.synchronize()
        out[key]["wall_s_incl_setup"] = round(time.perf_counter() - t0, 2)
        torch.cuda.empty_cache()
    return out


# --------------------------------------------------------------------------- #
# N > 1: the shards cover the global batch exactly once (envs, and the MCPG chain ids of the same batch)
# --------------------------------------------------------------------------- #
def shard_cover(G, rank, world, repeat_times=128):
    """This rank's env interval (rlsolver_amd.dist.env_shard) and -- when the batch can be an MCPG batch of G = M_total x
    `repeat_times` chains whose kept chains split in whole 64-chain tiles -- the global ids its chains get through
    rls_chain_ids (offset, period, skip): id(c) = (c // period) * (period + skip) + offset + c % period, the kernels' rule
    (include/rlsolver_hip.h).  Rank 0 checks that all ranks' sets tile [0, G) exactly once (check_cover)."""
    import numpy as np
    from rlsolver_amd import dist as rdist
    off, cnt = rdist.env_shard(G, rank, world)
    rec = {"rank": rank, "env_offset": off, "envs": cnt, "chain_ids": None}
    if G % repeat_times == 0 and (G // repeat_times) % (64 * world) == 0:
        m_total = G // repeat_times
        koff, m = rdist.env_shard(m_total // 64, rank, world)
        koff, m = koff * 64, m * 64                     # kept chains [koff, koff + m) of m_total, whole tiles
        rec["chain_ids"] = (koff, m, m_total - m)       # what MCPGRound(kept_offset=koff, total_kept=m_total) passes to the kernels
        c = np.arange(m * repeat_times, dtype=np.int64)
        rec["_ids"] = (c // m) * m_total + koff + c % m
    return rec


def check_cover(recs, G):
    import numpy as np
    iv = sorted((r["env_offset"], r["env_offset"] + r["envs"]) for r in recs)
    envs_ok = iv[0][0] == 0 and iv[-1][1] == G and all(iv[i][1] == iv[i + 1][0] for i in range(len(iv) - 1))
    out = {"global": G, "ranks": len(recs), "envs": "each exactly once" if envs_ok else "BROKEN"}
    if all(r.get("_ids") is not None for r in recs):
        cnt = np.bincount(np.concatenate([r["_ids"] for r in recs]), minlength=G)
        out["mcpg_chain_ids"] = "each exactly once" if (cnt.size == G and bool((cnt == 1).all())) else "BROKEN"
        out["chain_ids_per_rank"] = [list(r["chain_ids"]) for r in sorted(recs, key=lambda r: r["rank"])]
    if "BROKEN" in out.values():
        raise SystemExit(f"shard cover broken: {out}")
    return out


# --------------------------------------------------------------------------- #
# dry run: the N-rank wiring on CPU
# --------------------------------------------------------------------------- #
def dry_run(a):
    import torch
    import torch.distributed as dist
    from rlsolver_amd import dist as rdist
    rank, local_rank, world = rdist.init_from_env(backend="gloo")
    G = a.global_envs if a.global_envs > 0 else world * a.envs_per_gpu
    env_offset, B = rdist.env_shard(G, rank, world)
    if a.global_envs <= 0:
        assert (env_offset, B) == (rank * a.envs_per_gpu, a.envs_per_gpu), (env_offset, B)
    # a stand-in objective vector whose global maximum sits in a known shard: env id e scores (e * 7919) % 1009
    ids = torch.arange(env_offset, env_offset + min(B, 4096), dtype=torch.int64)
    obj = (ids * 7919) % 1009
    xs = ((ids[:, None] + torch.arange(16)[None, :]) % 3 == 0)
    best, owner, bx = rdist.global_best(obj, xs, want_solution=True)
    mine = {"rank": rank, "local_rank": local_rank, "env_offset": env_offset, "envs": B,
            "local_best": int(obj.max())}
    cov = shard_cover(G, rank, world)
    if world > 1:
        allr, covs = [None] * world, [None] * world
        dist.all_gather_object(allr, mine)
        dist.all_gather_object(covs, cov)
        dist.barrier()
        dist.destroy_process_group()
    else:
        allr, covs = [mine], [cov]
    if rank == 0:
        cover = sorted((r["env_offset"], r["env_offset"] + r["envs"]) for r in allr)     # the shards tile [0, G) exactly once
        assert cover[0][0] == 0 and cover[-1][1] == G and all(cover[i][1] == cover[i + 1][0] for i in range(world - 1)), cover
        print(json.dumps({"metric": metric_for(a.gset), "dry_run": True, "n_gpus": world, "global_envs": G,
                          "scaling": "strong" if a.global_envs > 0 else "weak", "ranks": allr, "shard_cover": check_cover(covs, G),
                          "global_best": int(best), "owner": int(owner), "best_x": [int(v) for v in bx.tolist()]}),
              flush=True)


# --------------------------------------------------------------------------- #
# the timed workload
# --------------------------------------------------------------------------- #
def measure(a, gset, B, steps, warmup, repeats, dev, rank, local_rank, world, via_env=False, verify=True, env_offset=None):
    """W warm-up steps, then `repeats` regions of exactly `steps` K4 launches.  Returns a dict with the per-region
    wall times (MAX over ranks), the HIP-event kernel times of this rank and -- N > 1 -- every rank's own kernel / exchange /
    region times (what an N > 1 line needs to attribute a shortfall: slow kernels on one rank, the exchange, or skew).
    `env_offset`: global id of this rank's env 0 (default rank * B: weak scaling, equal shards)."""
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
    env_offset = rank * B if env_offset is None else env_offset

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
        env._xs, env._obj = slots[0], obj       # same initial state as the launcher mode (env._xs: no resync needed, obj IS its cut)

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
    nccl = use_pg and dist.get_backend() == "nccl"
    # the region's exchange: ONE rls_best_key launch + ONE 8-byte all_reduce(MAX) (rlsolver_amd.dist.BestExchange); the key is
    # unpacked and the range flag read after the last region, never inside one
    ex = rdist.BestExchange(dev) if use_pg else None
    if use_pg:   # the first collective builds the communicator (16 ms on a 1-rank RCCL group): not part of any region, and
        ex.exchange(obj)          # not something a capture may do
        torch.cuda.synchronize(dev)
    # (auto needs at least one eager warm-up launch: a kernel's first launch loads its code object, which a capture must not do)
    use_graph = (not via_env) and (a.graph == "on" or (a.graph == "auto" and steps <= 100 and warmup > 0))
    graphs = []
    if use_graph:
        # one hipGraph per timed region: region r runs steps t_r .. t_r + steps - 1 of the ring / action cycle (capture enqueues
        # nothing: the launchers resolve torch's current stream at every call, which is the capture stream here)
        def capture(with_exchange):
            out, tc = [], t
            torch.cuda.synchronize(dev)
            for _ in range(repeats):
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for k in range(steps):
                        step(tc + k)
                    if with_exchange:
                        ex.exchange(obj)
                out.append(gr)
                tc += steps
            return out
        # over RCCL the region's exchange is captured with its steps (torch's NCCL ops are capturable): a region is then ONE
        # graph launch.  Keys staged through the host (gloo: --share-gpu) cannot be captured: steps in the graph, exchange eager
        exch_in_graph = False
        if nccl and a.graph_exchange != "off":
            try:
                graphs = capture(True)
                exch_in_graph = True
            except Exception as e:
                print(f"bench.py: capturing the exchange failed ({type(e).__name__}: {e}); exchange stays eager", file=sys.stderr)
                graphs = []
                torch.cuda.synchronize(dev)
        if not graphs:
            try:
                graphs = capture(False)
            except Exception as e:       # a capture that fails must not fail the measurement: the eager launcher loop is the same work
                if a.graph == "on":
                    raise
                print(f"bench.py: hipGraph capture failed ({type(e).__name__}: {e}); timing the eager launcher loop", file=sys.stderr)
                graphs, use_graph = [], False
                torch.cuda.synchronize(dev)
    else:
        exch_in_graph = False

    def barrier():
        if use_pg:
            dist.barrier(device_ids=[local_rank]) if nccl else dist.barrier()

    if use_pg:
        barrier()
    wall, kern, exch = [], [], []
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
        if use_pg and not exch_in_graph:  # episode boundary: best objective over all shards (C1, 8 bytes over RCCL)
            key = ex.exchange(obj)
            e2 = torch.cuda.Event(enable_timing=True)
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
        exch.append(e1.elapsed_time(e2) * 1e-3 if (use_pg and not exch_in_graph) else 0.0)   # last step's end -> the exchange's end

    per_rank = None
    exch_probe_s = None
    if use_pg:
        # the exchange by itself, outside every region (what `exchange_us` is when the regions hold it inside their graph): 50
        # back to back, HIP events, all ranks in step
        barrier()
        torch.cuda.synchronize(dev)
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        for _ in range(50):
            key = ex.exchange(obj)
        eb.record()
        torch.cuda.synchronize(dev)
        exch_probe_s = ea.elapsed_time(eb) * 1e-3 / 50
        if exch_in_graph:                    # a graph region's kernel time holds its exchange: take the stand-alone figure out
            kern = [max(0.0, (k * steps - exch_probe_s) / max(steps, 1)) for k in kern]
            exch = [exch_probe_s] * len(exch)
        # the exchange is checked here, after the regions: the key is the maximum over every rank's objectives, the flag clean
        best, owner = ex.unpack(key)
        ex.check()
        mx = obj.max().to(torch.int64).reshape(1)
        if nccl:
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        else:
            hm = mx.cpu()
            dist.all_reduce(hm, op=dist.ReduceOp.MAX)
            mx = hm
        if int(best) != int(mx[0]) or not (0 <= int(owner) < world):
            raise SystemExit(f"PARITY FAILURE: exchanged best {int(best)} (owner {int(owner)}) != max over ranks {int(mx[0])}")
    if use_pg:
        # every rank's own figures (outside every timed region), then the region time that counts: MAX over ranks
        mine = {"rank": rank, "device": str(dev), "envs": int(B), "env_offset": int(env_offset), "region_s": list(wall),
                "kernel_s_per_step": list(kern), "exchange_s": list(exch), "exchange_in_graph": bool(exch_in_graph),
                "exchange_alone_s": exch_probe_s}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        tt = torch.tensor(wall, dtype=torch.float64, device=dev if nccl else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = [float(v) for v in tt.tolist()]

    if verify:  # size-independent parity property at full size: incremental obj == recomputed
        final = slots[t % S]
        if not torch.equal(ops.maxcut_obj(g, final).to(torch.int32), obj):
            raise SystemExit("PARITY FAILURE: incremental objective != recomputed objective")

    res = {"wall": wall, "kernel_s": kern, "exchange_s": exch, "per_rank": per_rank, "N": N, "E": len(mygraph), "is_real": is_real,
           "B": B, "graph": use_graph, "graph_arr": np.asarray(mygraph, dtype=np.int64)}
    return res          # the ring and the launchers die with this frame: the next workload gets the memory back


def rank_breakdown(res, steps):
    """Per rank, for the MEDIAN region (the one `value` is computed from): kernel time per step (HIP events around the region's
    launches on that rank), the exchange (device time from the last step's end to the end of the best-objective all-reduce) and
    the rank's own wall time of the region; `skew_ms` = MAX - MIN of those wall times.  A shortfall at N > 1 then reads off:
    kernel_us_per_step up on every rank = the kernels themselves (clocks / HBM sharing), one rank's region_ms high = a straggler,
    exchange_us high = the collective."""
    pr = res.get("per_rank")
    if not pr:
        return None
    order = sorted(range(len(res["wall"])), key=lambda i: res["wall"][i])
    mid = order[len(order) // 2]
    ranks = [{"rank": r["rank"], "device": r["device"], "envs": r["envs"], "env_offset": r["env_offset"],
              "kernel_us_per_step": r["kernel_s_per_step"][mid] * 1e6, "exchange_us": r["exchange_s"][mid] * 1e6,
              "region_ms": r["region_s"][mid] * 1e3} for r in sorted(pr, key=lambda r: r["rank"])]
    reg = [r["region_ms"] for r in ranks]
    return {"region_index": mid, "ranks": ranks, "skew_ms": max(reg) - min(reg),
            "kernel_us_per_step_max": max(r["kernel_us_per_step"] for r in ranks),
            "kernel_us_per_step_min": min(r["kernel_us_per_step"] for r in ranks),
            "exchange_us_max": max(r["exchange_us"] for r in ranks),
            "exchange": "1 rls_best_key launch + 1 all_reduce(MAX, 8 B) per region"
                        + (", captured in the region's hipGraph (exchange_us = the same exchange timed alone, 50 back to back)"
                           if pr[0].get("exchange_in_graph") else ", eager after the region's last step"),
            "exchange_alone_us_max": max((r.get("exchange_alone_s") or 0.0) for r in pr) * 1e6,
            "skew_ms_all_regions": [(max(r["region_s"][i] for r in pr) - min(r["region_s"][i] for r in pr)) * 1e3
                                    for i in range(len(res["wall"]))]}


def summarize(res, steps, world, total_envs=None):
    B, N = res["B"], res["N"]
    el = statistics.median(res["wall"])
    ks = statistics.median(res["kernel_s"])
    bytes_per_launch = B * (2 * N + 20)
    achieved = bytes_per_launch / ks / 1e9
    total = world * B if total_envs is None else total_envs
    return {"value": total * steps / el, "ms_per_step": el / steps * 1e3, "rank_breakdown": rank_breakdown(res, steps),
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
    if a.exchange_probe:
        return exchange_probe_child(a)
    probe = None
    if a.gpus == 1 and not a.no_exchange_probe and os.environ.get("RLS_FORCE_PG") != "1":
        probe = exchange_probe_parent(a)        # a child process, started and finished before this one touches the GPU

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

    strong = a.global_envs > 0
    G = a.global_envs if strong else world * a.envs_per_gpu
    env_offset, B_rank = rdist.env_shard(G, rank, world)          # weak: rank * envs_per_gpu, envs_per_gpu
    res = measure(a, a.gset, B_rank, a.steps, a.warmup, R, dev, rank, local_rank, world,
                  via_env=a.via_env, verify=not a.no_verify, env_offset=env_offset)
    torch.cuda.empty_cache()
    res5 = None
    is_c5 = a.gset == 70 and (G == world * 131072)
    if not a.no_config5 and not is_c5:
        # BASELINE config #5's per-GPU shard, measured the same way (never mixed into `value`)
        res5 = measure(a, 70, 131072, max(1, min(a.steps, 200)), min(a.warmup, 20), R, dev, rank, local_rank, world,
                       verify=not a.no_verify)
        torch.cuda.empty_cache()
    res_nts = None
    if world == 1 and not a.no_hbm_only and res["B"] * res["N"] <= MALL_BYTES:
        # the same loop with nontemporal stores: a ring slot of B*N bytes fits the 256 MB Infinity Cache, and with plain stores
        # (the launcher's choice at this size) the next step's reads hit what this step wrote -- FETCH_SIZE counts those hits.
        # With nontemporal stores nothing a step writes is kept for the next one to find: the HBM-only figure.
        from rlsolver_amd import _abi
        _abi.tuning_set("RLS_STEP_NTS", 1)
        try:
            res_nts = measure(a, a.gset, B_rank, max(1, min(a.steps, 200)), min(a.warmup, 20), R, dev, rank, local_rank, world,
                              verify=not a.no_verify, env_offset=env_offset)
        finally:
            _abi.tuning_unset("RLS_STEP_NTS")
        torch.cuda.empty_cache()
    cfgs = None
    if rank == 0 and world == 1 and not a.no_configs:
        cfgs = extra_configs(dev)
    cover = None
    if world > 1:
        covs = [None] * world
        dist.all_gather_object(covs, shard_cover(G, rank, world))
        if rank == 0:
            cover = check_cover(covs, G)

    out = None
    if rank == 0:
        B, N = res["B"], res["N"]
        s = summarize(res, a.steps, world, total_envs=G)
        out = {
            "metric": metric_for(a.gset),
            "value": s["value"],
            "unit": "env-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": s["ms_per_step"],
            "repeats": R, "ms_per_step_all": s["ms_per_step_all"],
            "timing": f"median of {R} regions of exactly {a.steps} steps, each bracketed by barrier + synchronize, MAX over ranks",
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            # the graph: the real Gset file when data/gset/gset_<k>.txt is present, else the G(n, m) stand-in of its size; spins
            # and actions are drawn either way
            "dtype": "u8", "data": "real Gset graph, synthetic spins and actions" if res["is_real"] else "synthetic",
            "config": {"workload": f"Gset G{a.gset}{'' if res['is_real'] else '-sized G(n,m) stand-in'} MaxCut "
                                   f"(N={N}, E={res['E']}), {B} envs per GPU, K4 gym step emitting the next "
                                   f"state into a {a.slots}-slot rollout ring, uniform random actions"
                                   + ("" if a.via_env else "; the launcher passes cur = done = NULL (the rollout loop reads reward and keeps "
                                      "obj: 12 of the 20 scalar bytes per env-step are written; --via-env writes all four vectors)")
                                   + (f"; STRONG scaling: {G} envs in total split over {world} ranks" if strong else "")
                                   + ("; through EnvMaxcutGym.step(action, out=slot)" if a.via_env else "")
                                   + ("; BASELINE config #5 (2^20 envs over 8 GPUs = 131072 per GPU)" if is_c5 else ""),
                       "entry": "EnvMaxcutGym.step" if a.via_env else ("rls_maxcut_step launcher, each timed region of "
                                                                        f"{a.steps} launches enqueued as one hipGraph" if res["graph"]
                                                                        else "rls_maxcut_step launcher"),
                       "num_nodes": N, "num_edges": res["E"], "envs_per_gpu": B, "global_envs": G, "slots": a.slots,
                       "parallelism": f"env-shard x{world}" + (" (TEST MODE --share-gpu: all ranks on one GPU over gloo; not a scaling "
                                                                "measurement)" if a.share_gpu else "")},
            "roofline": s["roofline"],
        }
        if s["rank_breakdown"] is not None:
            out["rank_breakdown"] = s["rank_breakdown"]
        if cover is not None:
            out["shard_cover"] = cover
        tr = pmc_traffic_per_launch(B, N, a.slots)
        if tr is not None:
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = tr[0], f"profiles/{tr[1]} (rocprofv3 --pmc)"
        slot = B * N
        if slot <= MALL_BYTES:
            out["roofline"]["bound_detail"] = (f"hbm + infinity cache: a ring slot is {slot / 1e6:.0f} MB < 256 MB of MALL and the launcher "
                                               "stores plainly at this size, so part of a step's reads are served by what the previous step "
                                               "wrote (FETCH_SIZE counts MALL hits); `hbm_only` is the same loop with nontemporal stores")
        else:
            out["roofline"]["bound_detail"] = (f"hbm only: a ring slot is {slot / 1e6:.0f} MB > 256 MB of MALL, stores are nontemporal")
        if res_nts is not None:
            sn = summarize(res_nts, max(1, min(a.steps, 200)), world, total_envs=G)["roofline"]
            out["roofline"]["hbm_only"] = {"frac": sn["frac"], "achieved": sn["achieved"], "unit": "GB/s", "us_per_launch": sn["us_per_launch"],
                                           "how": "rls_tuning_set(\"RLS_STEP_NTS\", 1): the same workload, nontemporal stores",
                                           "kernel": "k_maxcut_step<u8, emit, nontemporal stores>"}
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
            if s5["rank_breakdown"] is not None:
                out["config5_shard"]["rank_breakdown"] = s5["rank_breakdown"]
        if cfgs is not None:
            out["configs"] = cfgs
        if probe is not None:
            out["exchange_probe"] = probe
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

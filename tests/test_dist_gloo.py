"""world_size-2 and -4 gloo tests of the episode-boundary exchange (C1/C2) and the env sharding."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rlsolver_amd import dist as rdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, lr, w = rdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    N, Bg = 33, 11
    off, cnt = rdist.env_shard(Bg, rank, world)
    g = torch.Generator().manual_seed(1234)
    xs_all = torch.randint(0, 2, (Bg, N), generator=g, dtype=torch.bool)
    vs_all = torch.tensor([5, 9, -3, 9, 2, 7, 1, 0, 9, 4, 8])
    case = []
    for shift in (0, 3, 7):  # move the maxima around, incl. a tie across ranks (lowest rank wins)
        v = torch.roll(vs_all, shift)
        obj, owner, bx = rdist.global_best(v[off:off + cnt], xs_all[off:off + cnt], want_solution=True)
        case.append((int(obj), int(owner), bx.clone()))
        gi = int(v.argmax())  # first maximum globally == lowest rank, first local index
        assert int(obj) == int(v.max())
        exp_owner = next(rk for rk in range(world) if rdist.env_shard(Bg, rk, world)[0] <= gi
                         < sum(rdist.env_shard(Bg, rk, world)))
        assert int(owner) == exp_owner
        assert torch.equal(bx, xs_all[gi])
    # negative objectives survive the packing
    obj, owner, _ = rdist.global_best(torch.tensor([-7 - rank, -9]))
    assert int(obj) == -7 and int(owner) == 0
    # the winner's GLOBAL index travels with its row (one message); rows may come from a callable (bit-packed storage)
    for shift in (0, 4):
        v = torch.roll(vs_all, shift)
        obj, owner, bx, gi = rdist.global_best(v[off:off + cnt], lambda li: xs_all[off + int(li)], want_solution=True,
                                               env_offset=off, num_nodes=N)
        assert int(gi) == int(v.argmax()) and torch.equal(bx, xs_all[int(gi)]) and bx.dtype == torch.bool
        obj, owner, bx, gi = rdist.global_best(-v[off:off + cnt], env_offset=off)             # MINLOC by negation, no row
        assert int(gi) == int(v.argmin()) and bx is None and int(obj) == -int(v.min())
    # whole-batch statistics of a sharded batch: per-node (min, max) in ONE collective, float64 sums
    mm_all = torch.stack([vs_all[:, None] - torch.arange(5)[None, :], vs_all[:, None] + torch.arange(5)[None, :]]).to(torch.int32)
    mm = torch.stack([mm_all[0, off:off + cnt].min(dim=0)[0], mm_all[1, off:off + cnt].max(dim=0)[0]])
    out = rdist.all_reduce_minmax(mm)
    assert out is mm and torch.equal(mm[0], mm_all[0].min(dim=0)[0]) and torch.equal(mm[1], mm_all[1].max(dim=0)[0])
    t = vs_all[off:off + cnt].to(torch.float64).sum().reshape(1)
    assert float(rdist.all_reduce_sum(t)[0]) == float(vs_all.sum())
    # the Sharded mixin routes through the group it was given; share_best = "everyone restarts from the best row"
    from rlsolver_amd.seeding import Sharded
    sh = Sharded()
    sh._init_shard(off, None, dist.group.WORLD)
    mm2 = torch.stack([mm_all[0, off:off + cnt].min(dim=0)[0], mm_all[1, off:off + cnt].max(dim=0)[0]])
    assert torch.equal(sh._global_minmax(mm2), mm)
    xs_l, vs_l = xs_all[off:off + cnt].clone(), vs_all[off:off + cnt].clone()
    best, owner = rdist.share_best(xs_l, vs_l)
    assert int(best) == 9 and bool((vs_l == 9).all()) and bool((xs_l == xs_all[1]).all())
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = True


def test_seed_stream_and_chain_id_numbering():
    """SeedStream: a private stream is a function of (seed, call count) only; the default draws from torch's generator.  The
    chain numbering of rls_chain_ids as MCPGRound builds it covers the global batch exactly once over the shards."""
    from rlsolver_amd.seeding import SeedStream
    a, b = SeedStream(5), SeedStream(5)
    sa = [a.next() for _ in range(4)]
    assert sa == [b.next() for _ in range(4)] and len(set(sa)) == 4 and all(0 <= s < 2 ** 62 for s in sa)
    assert SeedStream(6).next() != sa[0]
    assert a.derive(sa[0], 0) != a.derive(sa[0], 1) and a.derive(sa[0], 3) == b.derive(sa[0], 3)
    st = a.state_dict()
    nxt = a.next()
    c = SeedStream()
    c.load_state_dict(st)
    assert c.next() == nxt
    torch.manual_seed(3)
    d = SeedStream().next()
    torch.manual_seed(3)
    assert SeedStream().next() == d
    M, R = 512, 3
    seen = []
    for W in (1, 2, 4, 8):
        ids = set()
        for r in range(W):
            m0, ml = rdist.env_shard(M // 64, r, W)
            m0, ml = m0 * 64, ml * 64
            off, per, skip = m0, ml, M - ml
            for c_ in range(ml * R):
                ids.add(off + c_ + (c_ // per) * skip if per else off + c_)
        assert ids == set(range(M * R))
        seen.append(ids)


def test_global_best_two_ranks():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))


def _worker4(rank, world, port, ret):
    """4 ranks, uneven shards (B = 6 -> 2, 2, 1, 1), an EMPTY shard (B = 3), float (bidirectional) objectives,
    bit-packed solution broadcast for an N that is not a multiple of 8."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    rdist.init_from_env(backend="gloo")
    N = 77
    g = torch.Generator().manual_seed(99)
    for Bg in (6, 3, 13):
        xs_all = torch.randint(0, 2, (Bg, N), generator=g, dtype=torch.bool)
        vs_all = torch.randint(-50, 50, (Bg,), generator=g)
        off, cnt = rdist.env_shard(Bg, rank, world)
        if Bg == 3:
            assert cnt == (1 if rank < 3 else 0)          # rank 3 owns nothing and must not hang the others
        obj, owner, bx = rdist.global_best(vs_all[off:off + cnt], xs_all[off:off + cnt], want_solution=True)
        gi = int(vs_all.argmax())
        assert int(obj) == int(vs_all.max()) and torch.equal(bx, xs_all[gi]) and bx.dtype == torch.bool
        spans = [rdist.env_shard(Bg, rk, world) for rk in range(world)]
        assert int(owner) == next(rk for rk, (o, c) in enumerate(spans) if o <= gi < o + c)
        # float objectives of a bidirectional env (count / 2: integers or half-integers) keep their value
        vf = vs_all.to(torch.float32) / 2
        objf, ownerf, _ = rdist.global_best(vf[off:off + cnt])
        assert objf.dtype == torch.float64 and float(objf) == float(vf.max()) and int(ownerf) == int(owner)
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = True


def test_global_best_four_ranks_uneven_and_empty_shards():
    world = 4
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker4, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))


def test_pack_bits_roundtrip():
    g = torch.Generator().manual_seed(5)
    for n in (1, 7, 8, 9, 77, 2000):
        x = torch.randint(0, 2, (n,), generator=g, dtype=torch.bool)
        p = rdist.pack_bits(x)
        assert p.numel() == (n + 7) // 8 and torch.equal(rdist.unpack_bits(p, n), x)
    with pytest.raises(Exception):
        rdist.global_best(torch.tensor([1 << 50]))
    with pytest.raises(Exception):
        rdist.global_best(torch.tensor([0.25]))
    assert float(rdist.global_best(torch.tensor([1.5, -2.0]))[0]) == 1.5


def test_env_shard_partition():
    for B in (0, 1, 7, 64, 65537):
        for W in (1, 2, 3, 8):
            spans = [rdist.env_shard(B, r, W) for r in range(W)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == B
            for (o0, c0), (o1, _) in zip(spans, spans[1:]):
                assert o0 + c0 == o1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        rdist.env_shard(8, 2, 2)


def test_pack_unpack_roundtrip():
    for W in (1, 2, 8):
        for r in range(W):
            for v in (-5, 0, 13359, 2 ** 40):
                k = rdist.pack_key(torch.tensor(v), r, W)
                o, owner = rdist.unpack_key(k, W)
                assert int(o) == v and int(owner) == r
    # ordering: larger obj wins; equal obj -> lower rank wins
    assert rdist.pack_key(torch.tensor(10), 1, 2) > rdist.pack_key(torch.tensor(9), 0, 2)
    assert rdist.pack_key(torch.tensor(10), 0, 2) > rdist.pack_key(torch.tensor(10), 1, 2)


def _worker_empty(rank, world, port, ret):
    """An all-empty world must fail the same way on every rank (it used to broadcast an uninitialised buffer)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    rdist.init_from_env(backend="gloo")
    errs = 0
    for want in (True, False):
        try:
            rdist.global_best(torch.empty(0, dtype=torch.int64), torch.empty((0, 9), dtype=torch.bool), want_solution=want)
        except (ValueError, RuntimeError) as e:
            assert "no envs at all" in str(e)
            errs += 1
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = errs


def test_global_best_all_empty_world_raises_everywhere():
    world = 2
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker_empty, args=(world, port, ret), nprocs=world, join=True)
    assert [ret.get(r) for r in range(world)] == [2, 2]


def _run_bench(*args, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, env=e, timeout=600)
    return p, (json.loads(p.stdout.strip().splitlines()[-1]) if p.stdout.strip() else None)


@pytest.mark.parametrize("gpus", [1, 2, 4])
def test_bench_self_spawn_dry_run(gpus):
    """`python bench.py --gpus N` with no torchrun environment starts the N ranks itself; the JSON line is the LAST
    stdout line; rank r owns envs [r * B, (r + 1) * B); the exchange finds the global maximum and its lowest owner."""
    B = 3000
    p, out = _run_bench("--gpus", str(gpus), "--dry-run", "--envs-per-gpu", str(B))
    assert p.returncode == 0, p.stderr[-2000:]
    assert out["dry_run"] is True and out["n_gpus"] == gpus
    assert [r["rank"] for r in out["ranks"]] == list(range(gpus))
    assert [r["local_rank"] for r in out["ranks"]] == list(range(gpus))
    assert [r["env_offset"] for r in out["ranks"]] == [r * B for r in range(gpus)]
    score = lambda e: (e * 7919) % 1009
    per_rank = [max(score(e) for e in range(r * B, (r + 1) * B)) for r in range(gpus)]
    assert [r["local_best"] for r in out["ranks"]] == per_rank
    assert out["global_best"] == max(per_rank) and out["owner"] == per_rank.index(max(per_rank))
    e_star = next(e for e in range(out["owner"] * B, (out["owner"] + 1) * B) if score(e) == max(per_rank))
    assert out["best_x"] == [int((e_star + k) % 3 == 0) for k in range(16)]


def test_bench_eight_rank_dry_run_is_config5():
    """BASELINE config #5 as the driver would launch it, wiring only: `--gset 70 --global-envs 1048576 --gpus 8` (strong scaling:
    2^20 envs split over the ranks) -- eight rendezvous, shard offsets of 2^17, and the shards AND the MCPG chain ids of the same
    batch (rls_chain_ids: 8192 kept chains x 128 repeats, 1024 kept per rank) cover 2^20 exactly once."""
    p, out = _run_bench("--gpus", "8", "--dry-run", "--gset", "70", "--global-envs", str(1 << 20))
    assert p.returncode == 0, p.stderr[-2000:]
    assert out["n_gpus"] == 8 and out["global_envs"] == 1 << 20 and out["scaling"] == "strong"
    assert [(r["env_offset"], r["envs"]) for r in out["ranks"]] == [(k << 17, 1 << 17) for k in range(8)]
    cov = out["shard_cover"]
    assert cov["envs"] == cov["mcpg_chain_ids"] == "each exactly once" and cov["global"] == 1 << 20
    assert cov["chain_ids_per_rank"] == [[1024 * k, 1024, 7168] for k in range(8)]
    # a global batch that does not divide: the first (G % W) ranks take one env more, still an exact cover
    p, out = _run_bench("--gpus", "4", "--dry-run", "--global-envs", "10003")
    assert p.returncode == 0, p.stderr[-2000:]
    assert [r["envs"] for r in out["ranks"]] == [2501, 2501, 2501, 2500] and out["shard_cover"]["envs"] == "each exactly once"
    assert "mcpg_chain_ids" not in out["shard_cover"]


def test_bench_refuses_world_mismatch():
    p, _ = _run_bench("--gpus", "1", "--dry-run", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0",
                                                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    assert p.returncode != 0

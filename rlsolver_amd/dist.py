"""Multi-GPU sharding of the environment batch.

Envs are independent: rank r owns global envs [r*B_local, (r+1)*B_local); graph tables are
replicated; there is NO data-path collective during steps.  The only exchange is at episode
boundaries:

  C1  all_reduce(MAX) of ONE int64 key per rank, key = (best_obj << RANK_BITS) | (W-1-rank)
      (MAXLOC emulation: the largest objective wins, ties go to the lowest rank);
  C2  the winner's solution (bit-packed, ceil(N / 8) bytes + its global index) when the caller wants it -- on device tensors a
      SUM all-reduce of per-rank candidate messages (only the winner's is non-zero: no rank needs to know the owner on the host),
      on host tensors (the gloo CPU tests) a broadcast from the owner
      (the single-device analogue is best_xs[best_vs.argmax()] / Evaluator.record2,
      rlsolver/methods/L2A/demo_instance.py:165, rlsolver/methods/util_evaluator.py:90-107).

Backend: "nccl" (= RCCL over xGMI on ROCm) on GPUs, "gloo" in the CPU tests.  Payloads are 8 B
and <= N bytes, i.e. latency-bound.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist

RANK_BITS = 20  # up to 2^20 ranks; objectives up to 2^43


def env_shard(num_envs_global: int, rank: int, world_size: int) -> Tuple[int, int]:
    """(offset, count) of the contiguous shard of rank `rank`; the first (B % W) ranks get one more."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, rem = divmod(num_envs_global, world_size)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def pack_key(best_obj: torch.Tensor, rank: int, world_size: int) -> torch.Tensor:
    if world_size > (1 << RANK_BITS):
        raise ValueError("world too large for the packed key")
    return (best_obj.to(torch.int64) << RANK_BITS) | (world_size - 1 - rank)


def unpack_key(key: torch.Tensor, world_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
    obj = key >> RANK_BITS                       # arithmetic shift: negative objectives survive
    owner = (world_size - 1) - (key & ((1 << RANK_BITS) - 1))
    return obj, owner


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from torchrun's env; initialises the process group if W > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RLS_FORCE_PG=1 initialises a 1-rank group too (lets the RCCL code path be exercised on a 1-GPU box)
    if (world > 1 or os.environ.get("RLS_FORCE_PG") == "1") and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def _through_host(t: torch.Tensor, group) -> bool:
    """Device tensors under a backend that only moves host memory (gloo: the CPU tests, and the two-processes-on-one-GPU
    test) are staged through the host; RCCL takes them as they are."""
    return t.is_cuda and dist.get_backend(group) != "nccl"


def _all_reduce(t: torch.Tensor, op, group) -> torch.Tensor:
    if _through_host(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def all_reduce_minmax(mm: torch.Tensor, group=None) -> torch.Tensor:
    """mm int32 [2, N] = per-node (min, max) of a whole-batch statistic over the LOCAL envs -> over every rank's envs, in
    place and returned.  One collective of 8 N bytes: the minima travel negated under MAX.  The local search's
    ``rd_std = (max_b ws - min_b ws) * noise_std`` (envs/env_L2A.py:93-94) is the one such statistic on the path."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return mm
    mm[0].neg_()
    _all_reduce(mm, dist.ReduceOp.MAX, group)
    mm[0].neg_()
    return mm


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """Sum over the ranks, in place and returned (float64 / int64 tensors: callers convert first so that the result does not
    depend on how a float32 sum would have been ordered)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    return _all_reduce(t, dist.ReduceOp.SUM, group)


OBJ_LIMIT = 1 << 42   # |objective| (doubled for float inputs) must stay below this for the packed key
_EMPTY_KEY = -(1 << 63)  # what a rank with no envs contributes: loses against every real key


def _broadcast(t: torch.Tensor, src: int, group) -> torch.Tensor:
    if _through_host(t, group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def pack_bits(x: torch.Tensor) -> torch.Tensor:
    """bool/uint8 [N] (0|1) -> uint8 [ceil(N/8)], bit k of byte j = x[8j + k] (control-plane sized: <= 10 KB)."""
    n = x.numel()
    pad = (-n) % 8
    b = x.reshape(-1).to(torch.uint8)
    if pad:
        b = torch.cat([b, torch.zeros(pad, dtype=torch.uint8, device=b.device)])
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=b.device)
    return (b.view(-1, 8) * w).sum(dim=1, dtype=torch.uint8)


def unpack_bits(p: torch.Tensor, n: int, dtype=torch.bool) -> torch.Tensor:
    sh = torch.arange(8, dtype=torch.uint8, device=p.device)
    return ((p[:, None] >> sh) & 1).reshape(-1)[:n].to(dtype)


_FLAG_MSG = ("global_best: objective outside the packed-key range (|obj| < 2^42) or not a half-integer (flag bit 0), or no rank "
             "had an env (flag bit 1)")


class BestExchange:
    """The episode-boundary exchange of ONE call site as exactly two device operations: one ``rls_best_key`` launch (first
    argmax + packed key + range / half-integer check) into a slot of a persistent ring, and the 8-byte ``all_reduce(MAX)``
    on that slot (C1; the single-device analogue is ``best_vs.argmax()``, rlsolver/methods/L2A/demo_instance.py:165).
    Nothing is allocated, no [1]-sized torch op runs, nothing is read by the host:

        ex = BestExchange(device, group)          # once per call site
        key = ex.exchange(local_vs)               # per episode boundary: 1 launch + 1 collective
        ...
        obj, owner = ex.unpack(key)               # when the caller wants the numbers: 1 launch (rls_key_unpack)
        ex.check()                                # lazily: one host read of the sticky flag word

    A slot is reused after ``depth`` exchanges: a key the caller still holds then changes under it (``unpack`` writes fresh
    tensors).  The flag word is sticky and shared by the ring: it is asserted asynchronously every ``depth``-th exchange and
    on ``check()``, never per call.  The slot's layout is {key, index of the local maximum, spare}."""

    def __init__(self, device, group=None, depth: int = 64):
        self.device = torch.device(device)
        self.group = group
        self.depth = int(depth)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if self.world > (1 << RANK_BITS):
            raise ValueError("world too large for the packed key")
        self.ring = torch.zeros((self.depth, 4), dtype=torch.int64, device=self.device)
        self.flag = torch.zeros(2, dtype=torch.int32, device=self.device)[0:1]
        self._keys = [self.ring[i, 0:1] for i in range(self.depth)]      # views made once: a slice is a host-side cost per call
        self._idx = [self.ring[i, 1:2] for i in range(self.depth)]
        self._n = 0
        self._collective = (dist.is_initialized() and (self.world > 1 or os.environ.get("RLS_FORCE_PG") == "1"))
        self._host = self._collective and dist.get_backend(group) != "nccl"
        self.last_index = None

    def exchange(self, local_vs: torch.Tensor) -> torch.Tensor:
        """-> the REDUCED key, int64 [1] (a ring slot).  ``last_index`` = int64 [1] view of the local maximum's position."""
        from .torch_ops import ops as _t
        i = self._n % self.depth
        self._n += 1
        key, idx = self._keys[i], self._idx[i]
        if local_vs.numel():
            _t.best_key(local_vs if local_vs.is_contiguous() else local_vs.contiguous(), RANK_BITS, self.world - 1 - self.rank,
                        OBJ_LIMIT, key, idx, self.flag)
        else:
            key.fill_(_EMPTY_KEY)
        self.last_index = idx
        if self._collective:
            if self._host:
                h = key.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
                key.copy_(h)
            else:
                dist.all_reduce(key, op=dist.ReduceOp.MAX, group=self.group)            # C1: 8 bytes
        if i == self.depth - 1:
            torch._assert_async(self.flag[0] == 0, _FLAG_MSG)
        return key

    def unpack(self, key: torch.Tensor, as_float: bool = False):
        """(best objective 0-dim -- int64, or float64 = half the doubled key for float inputs --, owner rank int64 0-dim) as
        fresh tensors, one launch."""
        from .torch_ops import ops as _t
        obj = torch.empty(1, dtype=torch.float64 if as_float else torch.int64, device=self.device)
        owner = torch.empty(1, dtype=torch.int64, device=self.device)
        _t.key_unpack(key, RANK_BITS, self.world, obj, owner, _EMPTY_KEY, self.flag)
        return obj[0], owner[0]

    def check(self):
        """One host read of the sticky flag: raises what the per-call assert of earlier rounds raised."""
        f = int(self.flag[0])
        if f:
            raise ValueError(_FLAG_MSG + f" [flag = {f}]")


_SITES = {}


def _site(device, group) -> BestExchange:
    """global_best()'s own call site per (device, group, world): created on first use, lives as long as the process."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    k = (str(device), id(group) if group is not None else None, world, dist.is_initialized())
    ex = _SITES.get(k)
    if ex is None or ex.rank != (dist.get_rank(group) if dist.is_initialized() else 0):     # (a group torn down and rebuilt)
        ex = _SITES[k] = BestExchange(device, group)
    return ex


def _on_device(local_vs: torch.Tensor) -> bool:
    return local_vs.is_cuda and local_vs.dtype in (torch.int64, torch.int32, torch.float32, torch.float64)


def _local_best(local_vs: torch.Tensor, rank: int, world: int):
    """Host-tensor form (gloo CPU tests, dry runs): (key int64 [1] or None for an empty shard, first argmax index 0-dim
    tensor or None).  Device tensors take BestExchange."""
    is_float = local_vs.is_floating_point()
    if not local_vs.numel():
        return None, None
    li = local_vs.argmax()
    raw = local_vs[li]
    lbest = torch.round(raw.to(torch.float64) * 2).to(torch.int64) if is_float else raw.to(torch.int64)
    ok = lbest.abs() < OBJ_LIMIT
    if is_float:
        ok = ok & (lbest.to(torch.float64) == raw.to(torch.float64) * 2)
    torch._assert_async(ok, "global_best: objective outside the packed-key range (|obj| < 2^42) or not a half-integer")
    return pack_key(lbest, rank, world).reshape(1), li


def _row(local_xs, li):
    """The winner's row: local_xs is [B_local, N], or a callable li -> [N] (rows that live bit-packed, MCPG's kept chains)."""
    return local_xs(li) if callable(local_xs) else local_xs[li]


def global_best(local_vs: torch.Tensor, local_xs=None, want_solution: bool = False, group=None,
                env_offset: Optional[int] = None, num_nodes: Optional[int] = None):
    """Episode-boundary exchange.  local_vs [B_local] integer, or float holding integers / half-integers (the
    bidirectional envs return ``count / 2`` as float); local_xs [B_local, N] (bool/uint8), or a callable
    ``index -> row [N]`` with ``num_nodes`` = N.  B_local may be 0 (env_shard gives some ranks nothing when
    B < world): such a rank still joins every collective.

    Returns (best_obj 0-dim tensor -- int64, or float64 for float input --, owner_rank int64 0-dim, best_x or
    None), plus -- when ``env_offset`` (the global id of this rank's env 0) is given -- the winner's GLOBAL env index as
    a fourth element (int64 0-dim).  The solution travels bit-packed (ceil(N/8) bytes), the index in the same message.
    Single-process (no group initialised) degenerates to argmax over the local batch.
    """
    is_float = local_vs.is_floating_point()
    n_local = local_vs.numel()
    dev = local_vs.device
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world > (1 << RANK_BITS):
        raise ValueError("world too large for the packed key")
    finish = (lambda o: o.to(torch.float64) / 2) if is_float else (lambda o: o)
    want_x = want_solution and local_xs is not None
    want_i = env_offset is not None
    tail = (lambda gi: (gi,)) if want_i else (lambda gi: ())
    # RLS_FORCE_PG=1 keeps a 1-rank group on the collective path (the RCCL calls get exercised on a 1-GPU box)
    single = not dist.is_initialized() or (world == 1 and os.environ.get("RLS_FORCE_PG") != "1")
    if _on_device(local_vs):
        # device path: rls_best_key + all_reduce (+ rls_key_unpack): two launches and one collective, nothing allocated but the
        # two result words, nothing read by the host (BestExchange); the range / half-integer flag is checked lazily
        if single and not n_local:
            raise ValueError("global_best: no envs at all")
        ex = _site(dev, group)
        key = ex.exchange(local_vs)
        li = ex.last_index[0] if n_local else None
        obj, owner = ex.unpack(key, as_float=is_float)
        if single:
            return (obj, owner, (_row(local_xs, li).clone() if want_x else None)) + tail(li + env_offset if want_i else None)
        if want_x or want_i:
            # C2 without a host read: every rank writes its candidate message (the winner its global index + bit-packed row,
            # the others zeros: rls_winner_message), a SUM all-reduce is the winner's message everywhere, rls_winner_unpack
            # turns it back -- 2 launches + 1 collective, where the broadcast form needed int(owner) on the host first
            from .torch_ops import ops as _t
            n = (num_nodes if callable(local_xs) else local_xs.shape[1]) if want_x else 1
            if n is None:
                raise ValueError("global_best: a callable local_xs needs num_nodes")
            xs_arg = None
            if n_local:
                if not want_x:
                    xs_arg = torch.zeros(1, dtype=torch.uint8, device=dev)                 # (only the index travels)
                elif callable(local_xs):
                    xs_arg = local_xs(li).reshape(-1)
                else:
                    xs_arg = local_xs
                if xs_arg.dtype not in (torch.bool, torch.uint8) or not xs_arg.is_contiguous():
                    xs_arg = (xs_arg[li] if xs_arg.dim() == 2 else xs_arg).gt(0).contiguous()
            msg = torch.empty(8 + (n + 7) // 8, dtype=torch.uint8, device=dev)
            _t.winner_message(xs_arg, ex.last_index if n_local else None, key, RANK_BITS, world - 1 - rank,
                              int(env_offset) if want_i else 0, n, msg)
            _all_reduce(msg, dist.ReduceOp.SUM, group)                                     # C2: 8 + ceil(N / 8) bytes
            best_x = gi = None
            if want_x:
                dt = torch.bool if callable(local_xs) else local_xs.dtype
                best_x = torch.empty(n, dtype=dt if dt in (torch.bool, torch.uint8) else torch.uint8, device=dev)
            gi1 = torch.empty(1, dtype=torch.int64, device=dev) if want_i else None
            _t.winner_unpack(msg, n, best_x, gi1)
            if want_x and best_x.dtype != dt:
                best_x = best_x.to(dt)
            return (obj, owner, best_x) + tail(gi1[0] if want_i else None)
        finish = lambda o: o                                            # (rls_key_unpack has halved a float key already)
    else:
        key, li = _local_best(local_vs, rank, world)
        if single:
            if not n_local:
                raise ValueError("global_best: no envs at all")
            return (finish(key[0] >> RANK_BITS), torch.zeros((), dtype=torch.int64, device=dev),
                    (_row(local_xs, li).clone() if want_x else None)) + tail(li + env_offset if want_i else None)
        if key is None:
            key = torch.full((1,), _EMPTY_KEY, dtype=torch.int64, device=dev)
        _all_reduce(key, dist.ReduceOp.MAX, group)                      # C1: 8 bytes
        obj, owner = unpack_key(key[0], world)
    best_x = gi = None
    if want_x or want_i:
        if int(key[0]) == _EMPTY_KEY:                                   # (the host read below, taken one line early)
            raise ValueError("global_best: no envs at all")             # every rank fails the same way, before C2
        src = int(owner)                                                # one host read per episode boundary
        if want_x:
            n = num_nodes if callable(local_xs) else local_xs.shape[1]
            if n is None:
                raise ValueError("global_best: a callable local_xs needs num_nodes")
        nb = ((n + 7) // 8) if want_x else 0
        # one message: [8 bytes: the winner's global index (little endian) | ceil(N / 8) bytes: its solution, bit-packed]
        buf = torch.zeros(8 * want_i + nb, dtype=torch.uint8, device=dev)
        if rank == src and n_local:
            if want_i:
                g = (li + env_offset).to(torch.int64).reshape(1)
                buf[:8] = ((g[:, None] >> (8 * torch.arange(8, device=dev))) & 0xFF).to(torch.uint8).reshape(8)
            if want_x:
                buf[8 * want_i:] = pack_bits(_row(local_xs, li))
        _broadcast(buf, src, group)                                     # C2: (8 +) ceil(N/8) bytes
        if want_i:
            gi = (buf[:8].to(torch.int64) << (8 * torch.arange(8, device=dev))).sum()
        if want_x:
            dt = torch.bool if callable(local_xs) else local_xs.dtype
            best_x = unpack_bits(buf[8 * want_i:], n, dt)
    elif not n_local and not _on_device(local_vs):
        # no host read on this path: an all-empty world trips the same error asynchronously on every rank (a rank that has
        # envs knows the reduced key is a real one; the device path raises flag bit 1 in rls_key_unpack instead)
        torch._assert_async(key[0] != _EMPTY_KEY, "global_best: no envs at all")
    return (finish(obj), owner, best_x) + tail(gi)


def share_best(xs: torch.Tensor, vs: torch.Tensor, group=None):
    """The sharded form of the reference's "everyone restarts from the best" lines (envs/env_MCPG.py:452-458:
    ``best_i = best_vs.argmax(); best_xs[:] = best_xs[best_i]; best_vs[:] = best_vs[best_i]``; the same move in
    methods/L2A/demo_instance.py): C1 + C2, then every local row becomes the GLOBAL best.  In place; returns
    (best value 0-dim, owner rank 0-dim)."""
    best, owner, bx = global_best(vs, xs, want_solution=True, group=group)
    if xs.shape[0]:
        xs[:] = bx.to(xs.dtype)
        vs[:] = best.to(vs.dtype)
    return best, owner

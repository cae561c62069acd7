"""Multi-GPU sharding of the environment batch.

Envs are independent: rank r owns global envs [r*B_local, (r+1)*B_local); graph tables are
replicated; there is NO data-path collective during steps.  The only exchange is at episode
boundaries:

  C1  all_reduce(MAX) of ONE int64 key per rank, key = (best_obj << RANK_BITS) | (W-1-rank)
      (MAXLOC emulation: the largest objective wins, ties go to the lowest rank);
  C2  broadcast of the winner's solution (N bytes) when the caller wants it
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


OBJ_LIMIT = 1 << 42   # |objective| (doubled for float inputs) must stay below this for the packed key
_KEY_BUFFERS = {}


def _key_buffers(device):
    """Per-device scratch of the exchange: key int64 [1], index int64 [1], flag int32 [1] (zeroed once; the kernel only sets it)."""
    b = _KEY_BUFFERS.get(device)
    if b is None:
        b = _KEY_BUFFERS[device] = {"key": torch.zeros(1, dtype=torch.int64, device=device),
                                    "index": torch.zeros(1, dtype=torch.int64, device=device),
                                    "flag": torch.zeros(1, dtype=torch.int32, device=device)}
    return b
_EMPTY_KEY = -(1 << 63)  # what a rank with no envs contributes: loses against every real key


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


def global_best(local_vs: torch.Tensor, local_xs: Optional[torch.Tensor] = None, want_solution: bool = False,
                group=None):
    """Episode-boundary exchange.  local_vs [B_local] integer, or float holding integers / half-integers (the
    bidirectional envs return ``count / 2`` as float); local_xs [B_local, N] (bool/uint8).  B_local may be 0
    (env_shard gives some ranks nothing when B < world): such a rank still joins every collective.

    Returns (best_obj 0-dim tensor -- int64, or float64 for float input --, owner_rank int64 0-dim, best_x or
    None).  The solution travels bit-packed (ceil(N/8) bytes).  Single-process (no group initialised)
    degenerates to argmax over the local batch.
    """
    is_float = local_vs.is_floating_point()
    n_local = local_vs.numel()
    dev = local_vs.device
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world > (1 << RANK_BITS):
        raise ValueError("world too large for the packed key")
    li = lbest = key_dev = None
    if n_local and local_vs.is_cuda and local_vs.dtype in (torch.int64, torch.int32, torch.float32, torch.float64):
        # one launch: first argmax + the packed key (+ range / half-integer check) -- rls_best_key
        from .torch_ops import ops as _t
        bufs = _key_buffers(dev)
        _t.best_key(local_vs.contiguous(), RANK_BITS, world - 1 - rank, OBJ_LIMIT, bufs["key"], bufs["index"], bufs["flag"])
        torch._assert_async(bufs["flag"][0] == 0,
                            "global_best: objective outside the packed-key range (|obj| < 2^42) or not a half-integer")
        key_dev, li = bufs["key"], bufs["index"][0]
    elif n_local:
        li = local_vs.argmax()
        raw = local_vs[li]
        lbest = torch.round(raw.to(torch.float64) * 2).to(torch.int64) if is_float else raw.to(torch.int64)
        ok = lbest.abs() < OBJ_LIMIT
        if is_float:
            ok = ok & (lbest.to(torch.float64) == raw.to(torch.float64) * 2)
        torch._assert_async(ok, "global_best: objective outside the packed-key range (|obj| < 2^42) or not a half-integer")
    finish = (lambda o: o.to(torch.float64) / 2) if is_float else (lambda o: o)
    # RLS_FORCE_PG=1 keeps a 1-rank group on the collective path (the RCCL calls get exercised on a 1-GPU box)
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and os.environ.get("RLS_FORCE_PG") != "1"):
        if not n_local:
            raise ValueError("global_best: no envs at all")
        if key_dev is not None:
            lbest = key_dev[0] >> RANK_BITS
        return finish(lbest), torch.zeros((), dtype=torch.int64, device=dev), \
            (local_xs[li].clone() if (want_solution and local_xs is not None) else None)
    if key_dev is not None:
        key = key_dev                                                   # (a per-device buffer: consumed before the next call)
    else:
        key = (pack_key(lbest, rank, world) if n_local else torch.full((), _EMPTY_KEY, dtype=torch.int64, device=dev)).reshape(1)
    dist.all_reduce(key, op=dist.ReduceOp.MAX, group=group)            # C1: 8 bytes
    obj, owner = unpack_key(key[0], world)
    best_x = None
    if want_solution and local_xs is not None:
        n = local_xs.shape[1]
        if int(key[0]) == _EMPTY_KEY:                                   # (the host read below, taken one line early)
            raise ValueError("global_best: no envs at all")             # every rank fails the same way, before C2
        src = int(owner)                                                # one host read per episode boundary
        buf = pack_bits(local_xs[li]) if (rank == src and n_local) else \
            torch.empty((n + 7) // 8, dtype=torch.uint8, device=local_xs.device)
        dist.broadcast(buf, src=src, group=group)                      # C2: ceil(N/8) bytes
        best_x = unpack_bits(buf, n, local_xs.dtype)
    elif not n_local:
        # no host read on this path: an all-empty world trips the same error asynchronously on every rank (a rank that has
        # envs knows the reduced key is a real one)
        torch._assert_async(key[0] != _EMPTY_KEY, "global_best: no envs at all")
    return finish(obj), owner, best_x

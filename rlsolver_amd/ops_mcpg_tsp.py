"""Tensor-level wrappers for the MCPG and TSP entry points (same conventions as ops.py)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _abi
from .ops import DeviceGraph, _check, _ptr, _stream

TEN = torch.Tensor
_NM_DTYPES = (torch.float32, torch.uint8, torch.bool)


def _u64(v: int) -> C.c_uint64:
    return C.c_uint64(v & (2 ** 64 - 1))


# ------------------------------------------------------------------------------ MCPG
def mcpg_metro_rounds(samples: TEN, probs: TEN, T: int, index: Optional[TEN] = None, u: Optional[TEN] = None,
                      seed: int = 0, t_limit: Optional[TEN] = None, write_back: bool = True,
                      accepts: Optional[TEN] = None, t_offset: int = 0, samples_in: Optional[TEN] = None) -> None:
    """K9 (see include/rlsolver_hip.h).  samples [N, C] f32/uint8 in place, or read from ``samples_in`` (same
    shape / dtype) and written to ``samples``."""
    _check(samples, "samples", _NM_DTYPES)
    dev = samples.device
    if samples.dim() != 2:
        raise ValueError("samples must be [N, C]")
    N, Cc = samples.shape
    _check(probs, "probs", (torch.float32,), dev, (N,))
    if (index is None) != (u is None):
        raise ValueError("index and u must be given together")
    if index is not None:
        _check(index, "index", (torch.int64,), dev)
        _check(u, "u", (torch.float32,), dev)
        if index.shape[0] < t_offset + T or u.shape[0] < t_offset + T or index.shape[1:] != (Cc,) or u.shape[1:] != (Cc,):
            raise ValueError("index/u must be [>= t_offset + T, C]")
    if t_limit is not None:
        _check(t_limit, "t_limit", (torch.int64,), dev)
    if accepts is not None:
        _check(accepts, "accepts", (torch.int64,), dev)
        if accepts.numel() < T:
            raise ValueError("accepts must hold T entries")
    if samples_in is not None:
        _check(samples_in, "samples_in", (samples.dtype,), dev, (N, Cc))
        if not write_back:
            raise ValueError("samples_in needs write_back")
    _abi.call("rls_mcpg_metro_rounds", _ptr(samples), _ptr(samples_in), 4 if samples.dtype == torch.float32 else 1, N, Cc, _ptr(probs),
              T, t_offset, _ptr(index), _ptr(u), _u64(seed), _ptr(t_limit), int(bool(write_back)), _ptr(accepts),
              _stream(dev))


def mcpg_local_search(g: DeviceGraph, xs_in: TEN, order: TEN, num_ls: int, uniforms: Optional[TEN] = None,
                      seed: int = 0, visit_stream: Optional[TEN] = None):
    """K7 + expected cut.  Returns (xs_out f32 [N, C], expected f32 [C])."""
    _check(xs_in, "xs_in", _NM_DTYPES, g.device)
    if xs_in.dim() != 2 or xs_in.shape[0] != g.num_nodes:
        raise ValueError(f"xs_in must be [{g.num_nodes}, C]")
    Cc = xs_in.shape[1]
    _check(order, "order", (torch.int32,), g.device, (g.num_nodes,))
    if uniforms is not None:
        _check(uniforms, "uniforms", (torch.float32,), g.device, (num_ls, g.num_nodes, Cc))
    xs_out = torch.empty((g.num_nodes, Cc), dtype=torch.float32, device=g.device)
    expected = torch.empty(Cc, dtype=torch.float32, device=g.device)
    if visit_stream is not None:
        _check(visit_stream, "visit_stream", (torch.int32,), g.device)
    _abi.call("rls_mcpg_local_search", g.ref, _ptr(xs_in), 4 if xs_in.dtype == torch.float32 else 1, _ptr(xs_out), Cc,
              _ptr(order), _ptr(visit_stream), 0 if visit_stream is None else visit_stream.numel(), num_ls,
              _ptr(uniforms), _u64(seed), _ptr(expected), _stream(g.device))
    return xs_out, expected


def mcpg_local_search_levels(g: DeviceGraph, xs_in: TEN, lv_ptr: TEN, lv_data: TEN, num_ls: int, seed: int = 0,
                             coins: Optional[TEN] = None):
    """K7 + expected cut on the level-parallel schedule (rls_mcpg_visit_levels).  ``coins`` int64 (bit pattern of
    uint64) [num_ls * N, ceil(C / 64)]: the tie coins "u < 1/2" -- test hook; None = counter hash keyed by seed.
    Returns (xs_out f32 [N, C], expected f32 [C])."""
    _check(xs_in, "xs_in", _NM_DTYPES, g.device)
    if xs_in.dim() != 2 or xs_in.shape[0] != g.num_nodes:
        raise ValueError(f"xs_in must be [{g.num_nodes}, C]")
    Cc = xs_in.shape[1]
    _check(lv_ptr, "lv_ptr", (torch.int32,), g.device)
    _check(lv_data, "lv_data", (torch.int32,), g.device)
    if coins is not None:
        _check(coins, "coins", (torch.int64,), g.device, (num_ls * g.num_nodes, (Cc + 63) // 64))
    xs_out = torch.empty((g.num_nodes, Cc), dtype=torch.float32, device=g.device)
    expected = torch.empty(Cc, dtype=torch.float32, device=g.device)
    _abi.call("rls_mcpg_local_search_levels", g.ref, _ptr(xs_in), 4 if xs_in.dtype == torch.float32 else 1, _ptr(xs_out),
              Cc, _ptr(lv_ptr), _ptr(lv_data), lv_ptr.numel() - 1, num_ls, _ptr(coins), _u64(seed), _ptr(expected),
              _stream(g.device))
    return xs_out, expected


def mcpg_pick_best(expected: TEN, xs: TEN, total_mcmc_num: int, repeat_times: int, num_edges: int):
    """K8 second half.  Returns (best_index int64 [M], vs_good f32 [M], xs_good f32 [N, M])."""
    _check(xs, "xs", (torch.float32,))
    dev = xs.device
    N, Cc = xs.shape
    if Cc != total_mcmc_num * repeat_times:
        raise ValueError("xs must be [N, total_mcmc_num * repeat_times]")
    _check(expected, "expected", (torch.float32,), dev, (Cc,))
    idx = torch.empty(total_mcmc_num, dtype=torch.int64, device=dev)
    vs = torch.empty(total_mcmc_num, dtype=torch.float32, device=dev)
    xg = torch.empty((N, total_mcmc_num), dtype=torch.float32, device=dev)
    _abi.call("rls_mcpg_pick_best", _ptr(expected), _ptr(xs), N, total_mcmc_num, repeat_times, num_edges, _ptr(idx),
              _ptr(vs), _ptr(xg), _stream(dev))
    return idx, vs, xg


# ------------------------------------------------------------------------------ TSP
def _perm(p: TEN, name="perm"):
    _check(p, name, (torch.int64,))
    if p.dim() != 2:
        raise ValueError(f"{name} must be [B, N]")
    return p.shape


def tsp_tour_length(dist: TEN, perm: TEN) -> TEN:
    B, N = _perm(perm)
    _check(dist, "dist", (torch.float32,), perm.device, (N, N))
    out = torch.empty(B, dtype=torch.float32, device=perm.device)
    _abi.call("rls_tsp_tour_length", _ptr(dist), N, _ptr(perm), B, _ptr(out), _stream(perm.device))
    return out


def tsp_swap_delta_all(dist: TEN, perm: TEN, selected: TEN, temperature: float):
    B, N = _perm(perm)
    dev = perm.device
    _check(dist, "dist", (torch.float32,), dev, (N, N))
    _check(selected, "selected", (torch.int64,), dev, (B, N))
    logratio = torch.empty((B, N), dtype=torch.float32, device=dev)
    indices = torch.empty((B, N), dtype=torch.int64, device=dev)
    ban = torch.empty((B, N), dtype=torch.bool, device=dev)
    _abi.call("rls_tsp_swap_delta_all", _ptr(dist), N, _ptr(perm), B, _ptr(selected), float(temperature),
              _ptr(logratio), _ptr(indices), _ptr(ban), _stream(dev))
    return logratio, indices, ban


def tsp_apply_swap(perm: TEN, pos: TEN, indices: TEN) -> None:
    B, N = _perm(perm)
    _check(pos, "pos", (torch.int64,), perm.device, (B,))
    _check(indices, "indices", (torch.int64,), perm.device, (B, N))
    _abi.call("rls_tsp_apply_swap", _ptr(perm), B, N, _ptr(pos), _ptr(indices), _stream(perm.device))


def tsp_2opt_delta(dist: TEN, perm: TEN, i: TEN, j: TEN) -> TEN:
    B, N = _perm(perm)
    dev = perm.device
    _check(dist, "dist", (torch.float32,), dev, (N, N))
    _check(i, "i", (torch.int64,), dev, (B,))
    _check(j, "j", (torch.int64,), dev, (B,))
    out = torch.empty(B, dtype=torch.float32, device=dev)
    _abi.call("rls_tsp_2opt_delta", _ptr(dist), N, _ptr(perm), B, _ptr(i), _ptr(j), _ptr(out), _stream(dev))
    return out


def rand_perms(B: int, N: int, seed: int, device, env_offset: int = 0) -> TEN:
    device = torch.device(device)
    if device.type != "cuda":
        raise TypeError("rand_perms needs a HIP device")
    out = torch.empty((B, N), dtype=torch.int64, device=device)
    _abi.call("rls_rand_perms", _ptr(out), B, N, _u64(seed), env_offset, _stream(device))
    return out

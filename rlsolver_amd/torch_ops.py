"""The kernels as PyTorch custom ops: ``torch.ops.rlsolver_hip.*``.

north_star asks for "hand-written HIP kernels through PyTorch-ROCm custom ops over a thin C-ABI":
this module registers the hot-path entry points with ``torch.library`` (dispatch key CUDA = HIP on
ROCm) so they appear in the dispatcher, profiler traces and ``torch.ops``.  Only the CUDA/HIP key
gets an implementation -- calling an op with CPU tensors raises NotImplementedError from the
dispatcher (no CPU fallback).  The shared graph is passed as an integer handle from
``register_graph`` because op schemas cannot carry a struct of device pointers.

The env classes call ``rlsolver_amd.ops`` directly (same C ABI, one dispatcher hop less).
"""
from __future__ import annotations

import itertools
from typing import Dict

import torch

from . import ops

_graphs: Dict[int, ops.DeviceGraph] = {}
_next = itertools.count(1)


def register_graph(g: ops.DeviceGraph) -> int:
    h = next(_next)
    _graphs[h] = g
    return h


def release_graph(handle: int) -> None:
    _graphs.pop(handle, None)


def _g(handle: int) -> ops.DeviceGraph:
    try:
        return _graphs[handle]
    except KeyError:
        raise RuntimeError(f"unknown rlsolver_hip graph handle {handle}") from None


_lib = torch.library.Library("rlsolver_hip", "DEF")
_lib.define("maxcut_obj(int graph, Tensor xs) -> Tensor")
_lib.define("maxcut_delta_all(int graph, Tensor xs) -> Tensor")
_lib.define("maxcut_node_cutdeg(int graph, Tensor xs) -> Tensor")
_lib.define("maxcut_step(int graph, Tensor x_in, Tensor(a!) x_out, Tensor action, Tensor(b!) obj, "
            "Tensor(c!) reward) -> ()")
_lib.define("maxcut_greedy_sweep(int graph, Tensor(a!) xs, Tensor(b!) obj) -> ()")
_lib.define("maxcut_propose_accept(int graph, Tensor(a!) xs, Tensor mask, Tensor(b!) obj) -> ()")
_lib.define("select_better_rows(Tensor(a!) xs0, Tensor(b!) vs0, Tensor xs1, Tensor vs1, bool if_maximize) -> ()")
_lib.define("tsp_tour_length(Tensor dist, Tensor perm) -> Tensor")

_impl = torch.library.Library("rlsolver_hip", "IMPL", "CUDA")
_impl.impl("maxcut_obj", lambda graph, xs: ops.maxcut_obj(_g(graph), xs))
_impl.impl("maxcut_delta_all", lambda graph, xs: ops.maxcut_delta_all(_g(graph), xs))
_impl.impl("maxcut_node_cutdeg", lambda graph, xs: ops.maxcut_node_cutdeg(_g(graph), xs))
_impl.impl("maxcut_step", lambda graph, x_in, x_out, action, obj, reward:
           ops.maxcut_step(_g(graph), x_in, x_out, action, obj, reward))
_impl.impl("maxcut_greedy_sweep", lambda graph, xs, obj: ops.maxcut_greedy_sweep(_g(graph), xs, obj))
_impl.impl("maxcut_propose_accept", lambda graph, xs, mask, obj: ops.maxcut_propose_accept(_g(graph), xs, mask, obj))
_impl.impl("select_better_rows", lambda xs0, vs0, xs1, vs1, if_maximize:
           ops.select_better_rows(xs0, vs0, xs1, vs1, if_maximize))


def _tsp_len(dist, perm):
    from . import ops_mcpg_tsp as mops
    return mops.tsp_tour_length(dist, perm)


_impl.impl("tsp_tour_length", _tsp_len)

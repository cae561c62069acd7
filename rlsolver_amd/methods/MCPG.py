"""MCPG sampling functions -- drop-in for rlsolver/methods/MCPG.py:88-166 (metro_sampling,
sampler_func) and the graph part of maxcut_dataloader (:187-232), on HIP tensors.

Shapes and dtypes follow the reference: probs f32 [N]; chains node-major f32 [N, C] holding 0/1;
sampler_func returns (vs_good f32 [M], xs_good f32 [N, M], value f32 [C]).
"""
from __future__ import annotations

import types
from typing import Optional

import numpy as np
import torch

from .. import _abi, ops, ops_mcpg_tsp as mops
from ..graph import build_csr, read_edge_arrays

TEN = torch.Tensor


def _seed_from_torch() -> int:
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def maxcut_dataloader(path, device):
    """The parts of maxcut_dataloader (MCPG.py:187-232) that sampling needs: edge_index int64 [2, E],
    per-node degree, sorted_degree_nodes = argsort(|weighted degree|, descending).  Built with numpy
    in O(E) instead of the reference's Python O(E * deg) list surgery; the shared graph goes to the
    device once as a CSR.  Returns (data, num_nodes) like the reference."""
    num_nodes, eu, ev, w = read_edge_arrays(path)
    return make_data(num_nodes, eu, ev, device), num_nodes


def make_data(num_nodes: int, eu, ev, device, sorted_degree_nodes=None):
    device = torch.device(device)
    eu = np.asarray(eu, dtype=np.int64)
    ev = np.asarray(ev, dtype=np.int64)
    csr = build_csr((eu, ev, np.ones_like(eu)), num_nodes=num_nodes, if_bidirectional=False)
    # the objective kernel walks the edge list in file order semantics (order is irrelevant to a sum)
    data = types.SimpleNamespace()
    data.num_nodes = num_nodes
    data.edge_index = torch.from_numpy(np.stack([eu, ev])).to(device)
    data.num_edges = int(eu.shape[0])
    deg = np.bincount(np.concatenate([eu, ev]), minlength=num_nodes).astype(np.float64)
    data.single_degree = deg.astype(np.int64).tolist()
    data.weighted_degree = deg.tolist()
    if sorted_degree_nodes is None:
        # torch.argsort(descending=True) is not stable; any tie order is a valid reference outcome
        sorted_degree_nodes = torch.argsort(torch.from_numpy(deg), descending=True, stable=True)
    data.sorted_degree_nodes = torch.as_tensor(sorted_degree_nodes).to(torch.int64)
    data.graph = ops.DeviceGraph(csr, device)
    data._order_i32 = data.sorted_degree_nodes.to(device=device, dtype=torch.int32).contiguous()
    data._visit_stream = torch.from_numpy(build_visit_stream(csr, data.sorted_degree_nodes.cpu().numpy())).to(device)
    lv = build_visit_levels(csr, data.sorted_degree_nodes.cpu().numpy())
    data._lv_ptr, data._lv_data = (None, None) if lv is None else (torch.from_numpy(lv[0]).to(device),
                                                                  torch.from_numpy(lv[1]).to(device))
    return data


def build_visit_levels(csr, order: np.ndarray):
    """Level-parallel (lane = node) form of the visiting order for the production K7 kernel
    (include/rlsolver_hip.h: rls_mcpg_visit_levels).  None when the graph is outside its limits."""
    import ctypes as C
    from .. import _abi
    n = csr.num_nodes
    if n >= (1 << 20) or csr.max_degree >= 1024:
        return None
    rp = np.ascontiguousarray(csr.rowptr, dtype=np.int32)
    col = np.ascontiguousarray(csr.col, dtype=np.int32)
    od = np.ascontiguousarray(order, dtype=np.int32)
    a = (rp.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p), n, od.ctypes.data_as(C.c_void_p))
    ng, tot = C.c_int64(0), C.c_int64(0)
    _abi.call("rls_mcpg_visit_levels", *a, None, 0, None, 0, C.byref(ng), C.byref(tot))
    lvp = np.empty(int(ng.value) + 1, dtype=np.int32)
    lvd = np.empty(max(int(tot.value), 1), dtype=np.int32)
    _abi.call("rls_mcpg_visit_levels", *a, lvp.ctypes.data_as(C.c_void_p), lvp.size, lvd.ctypes.data_as(C.c_void_p),
              lvd.size, C.byref(ng), C.byref(tot))
    return lvp, lvd


def build_visit_stream(csr, order: np.ndarray, max_nodes: int = 32, max_entries: int = 400) -> np.ndarray:
    """Flatten the visiting order for the streaming K7 kernel (format: rls_mcpg.hip / include/rlsolver_hip.h).

    The sequential pass (MCPG.py:136-142) only orders ADJACENT nodes, so the visiting positions are
    level-scheduled by the library's host pass (rls_graph_sweep_schedule, on the graph relabelled by visiting
    position): positions sorted by (dependency level, position), levels cut into batches of <= max_nodes
    nodes / <= max_entries stream entries.  Nodes of a batch are pairwise non-adjacent and every neighbour
    visited earlier sits in an earlier batch: deciding batch after batch reproduces the sequential result."""
    import ctypes as C
    from .. import _abi
    n = csr.num_nodes
    order = np.asarray(order, dtype=np.int64)
    pos_of = np.empty(n, dtype=np.int64)
    pos_of[order] = np.arange(n)
    deg = np.diff(csr.rowptr).astype(np.int64)
    deg_o = deg[order]
    nnz = int(deg_o.sum())
    cum = np.concatenate([[0], np.cumsum(deg_o)])
    # rows gathered in visiting order
    starts = csr.rowptr[order].astype(np.int64)
    idx = np.repeat(starts - cum[:-1], deg_o) + np.arange(nnz)
    nbr = csr.col[idx].astype(np.int64)
    row_pos = np.repeat(np.arange(n), deg_o)
    fresh = pos_of[nbr] > row_pos
    nfresh = np.bincount(row_pos, weights=fresh, minlength=n).astype(np.int64)
    # level schedule over visiting positions (graph relabelled by position); a record is 3 entries longer than
    # the scheduler's (1 + deg) count
    rp_p = np.ascontiguousarray(cum, dtype=np.int32)
    col_p = np.ascontiguousarray(pos_of[nbr], dtype=np.int32)
    flagged = np.empty(n + 1, dtype=np.int32)
    tmp = np.empty(nnz + n, dtype=np.int32)
    nb, nl = C.c_int64(0), C.c_int64(0)
    _abi.call("rls_graph_sweep_schedule", rp_p.ctypes.data_as(C.c_void_p), col_p.ctypes.data_as(C.c_void_p), n,
              max_nodes, max(max_entries - 3 * max_nodes, 1), flagged.ctypes.data_as(C.c_void_p),
              tmp.ctypes.data_as(C.c_void_p), C.byref(nb), C.byref(nl))
    off = (flagged.view(np.uint32) & 0x7FFFFFFF).astype(np.int64)
    sp = tmp[off[:n]].astype(np.int64)                               # visiting position at schedule slot k
    is_start = (flagged[:n].view(np.uint32) >> 31).astype(bool)
    bid = np.cumsum(is_start) - 1                                   # batch of each slot
    first_slot = np.flatnonzero(is_start)
    m = np.diff(np.concatenate([first_slot, [n]]))                   # nodes per batch
    deg_s = deg_o[sp]
    rec_len = deg_s + 4
    rec_total = np.add.reduceat(rec_len, first_slot)
    batch_size = 3 + m + rec_total
    H = np.concatenate([[0], np.cumsum(batch_size)])                 # header offset of each batch
    rec_cum = np.cumsum(rec_len) - rec_len
    within = rec_cum - rec_cum[first_slot][bid]
    rec_off = H[bid] + 3 + m[bid] + within
    stream = np.zeros(int(H[-1]), dtype=np.int64)
    stream[H[:-1]] = m
    stream[H[:-1] + 1] = H[1:]
    k_in_batch = np.arange(n) - first_slot[bid]
    stream[H[bid] + 3 + k_in_batch] = rec_off
    stream[rec_off], stream[rec_off + 1], stream[rec_off + 2], stream[rec_off + 3] = order[sp], deg_s, nfresh[sp], sp
    # rows: entries of visiting position sp[k] copied to rec_off[k] + 4 ...
    src = np.repeat(cum[:-1][sp], deg_s) + (np.arange(int(deg_s.sum())) - np.repeat(np.cumsum(deg_s) - deg_s, deg_s))
    dst = np.repeat(rec_off + 4, deg_s) + (np.arange(int(deg_s.sum())) - np.repeat(np.cumsum(deg_s) - deg_s, deg_s))
    stream[dst] = nbr[src] | (fresh[src].astype(np.int64) << 31)
    return (stream & 0xFFFFFFFF).astype(np.uint32).view(np.int32)


def metro_sampling(probs: TEN, start_status: TEN, max_transfer_time: int, device=None,
                   index: Optional[TEN] = None, u: Optional[TEN] = None) -> TEN:
    """MCPG.py:88-117.  Up to 5*T proposal rounds per chain, stopping after the first round whose
    cumulative accept count reaches C*T -- evaluated on the device (two kernel launches, no host
    sync), where the reference syncs once per round.  ``index``/``u`` ([>=5T, C]) replace the
    torch.randint / torch.rand draws (test hook)."""
    device = start_status.device if device is None else torch.device(device)
    start = start_status.to(device=device, dtype=torch.float32).contiguous()
    # the first chunk reads the caller's start state and writes the result buffer: no copy of the [N, C] state
    samples = torch.empty_like(start) if start.data_ptr() == start_status.data_ptr() else start
    probs = probs.detach().to(device=device, dtype=torch.float32).contiguous()
    N, Cc = samples.shape
    Tmax = max_transfer_time * 5
    if index is not None:
        Tmax = min(Tmax, index.shape[0])
    if Tmax <= 0:   # no rounds (N < 10 gives T = int(N / 10) = 0): the reference returns start_status.bool().float()
        return start.clone() if samples is not start else start
    seed = _seed_from_torch() if index is None else 0
    # Walk the rounds in chunks of T.  A round accepts at most C proposals, so the cumulative count cannot reach
    # C*T before the LAST round of the first chunk: that chunk is applied directly (one pass, counting as it
    # goes).  Later chunks: dry pass -> accept counts -> stop round (on the device) -> apply.  Chunks after the
    # stop round see a zero limit and return at once.
    chunk = max(1, max_transfer_time)
    target = Cc * max_transfer_time
    cum_prev = torch.zeros((), dtype=torch.int64, device=device)
    live = torch.ones((), dtype=torch.bool, device=device)
    zero = torch.zeros((), dtype=torch.int64, device=device)
    for t0 in range(0, Tmax, chunk):
        tk = min(chunk, Tmax - t0)
        tk_dev = torch.full((), tk, dtype=torch.int64, device=device)
        accepts = torch.zeros(tk, dtype=torch.int64, device=device)
        if t0 == 0:
            mops.mcpg_metro_rounds(samples, probs, tk, index, u, seed, None, True, accepts, t_offset=0,
                                   samples_in=None if samples is start else start)
            cum = accepts.cumsum(0)
            hit = cum[-1] >= target
        else:
            limit = torch.where(live, tk_dev, zero).reshape(1).contiguous()
            mops.mcpg_metro_rounds(samples, probs, tk, index, u, seed, limit, False, accepts, t_offset=t0)
            cum = cum_prev + accepts.cumsum(0)
            reached = cum >= target
            hit = reached.any()
            t_stop = torch.where(hit, reached.to(torch.int64).argmax() + 1, tk_dev)
            apply_limit = torch.minimum(limit[0], t_stop).reshape(1).contiguous()
            mops.mcpg_metro_rounds(samples, probs, tk, index, u, seed, apply_limit, True, None, t_offset=t0)
        cum_prev = cum[-1]
        live = live & ~hit
    return samples


def sampler_func(data, xs_sample: TEN, num_ls: int, total_mcmc_num: int, repeat_times: int, device=None,
                 uniforms: Optional[TEN] = None):
    """MCPG.py:120-166: node-sequential stochastic local search (K7), expected cut (K8), best of
    repeats.  ``uniforms`` f32 [num_ls, N, C] replaces torch.rand (test hook)."""
    xs_sample = xs_sample.contiguous()
    seed = _seed_from_torch() if uniforms is None else 0
    if uniforms is None and getattr(data, '_lv_ptr', None) is not None and _abi.lib().rls_mcpg_local_search_levels_supported(
            data.graph.ref, data._lv_ptr.numel() - 1):
        # production path: level-parallel kernel (the draws only ever decide ties, so it carries coins, not uniforms)
        xs_loc, expected = mops.mcpg_local_search_levels(data.graph, xs_sample, data._lv_ptr, data._lv_data, num_ls, seed)
    else:
        xs_loc, expected = mops.mcpg_local_search(data.graph, xs_sample, data._order_i32, num_ls, uniforms, seed,
                                                  visit_stream=getattr(data, '_visit_stream', None))
    _, vs_good, xs_good = mops.mcpg_pick_best(expected, xs_loc, total_mcmc_num, repeat_times, data.num_edges)
    value = expected - expected.mean()
    return vs_good, xs_good, value

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
from ..ops_mcpg_tsp import PackedChains
from ..graph import build_csr, read_edge_arrays
from ..seeding import Sharded, seed_from_torch as _seed_from_torch

TEN = torch.Tensor


def maxcut_dataloader(path, device):
    """The parts of maxcut_dataloader (MCPG.py:187-232) that sampling needs: edge_index int64 [2, E],
    per-node degree, sorted_degree_nodes = argsort(|weighted degree|, descending).  Built with numpy
    in O(E) instead of the reference's Python O(E * deg) list surgery; the shared graph goes to the
    device once as a CSR.  Returns (data, num_nodes) like the reference."""
    num_nodes, eu, ev, w = read_edge_arrays(path)
    return make_data(num_nodes, eu, ev, device), num_nodes


class MCPGData(types.SimpleNamespace):
    """The ``Data`` object of maxcut_dataloader (MCPG.py:187-232).  What the samplers here read is set eagerly by
    make_data; the per-node / per-edge Python lists the reference also hangs on it (``neighbors``, ``neighbor_edges``,
    ``add_items``, ``sorted_degree_edges``: MCPG.py:205-231, 235-252) are built on first access -- nothing in the HIP
    path reads them, a caller's own code might."""

    _LAZY = ("neighbors", "neighbor_edges", "add_items", "sorted_degree_edges")

    def __getattr__(self, name):            # only reached for attributes that are not set yet
        if name in MCPGData._LAZY:
            append_neighbors(self)
            return self.__dict__[name]
        raise AttributeError(name)


def append_neighbors(data, device=None):
    """MCPG.py:235-289: ``data.neighbors[i]`` = node i's neighbours in edge-file order (both directions of every edge, int64
    tensor), ``data.neighbor_edges[i]`` = their weights (all ones) as [1, deg] -- the shape maxcut_dataloader leaves them in
    (:227-228) --, plus the dataloader's ``add_items`` [3, E] and ``sorted_degree_edges`` (:216-231).  One stable sort and
    two device tensors; every list entry is a view.  (The reference's per-edge n0 / n1 neighbour lists -- O(E deg) memory,
    read by nothing on the MaxCut path -- are not built.)"""
    dev = torch.device(device) if device is not None else data.edge_index.device
    ei = data.edge_index.cpu().numpy()
    eu, ev = ei[0], ei[1]
    E, N = eu.shape[0], data.num_nodes
    rows = np.stack([eu, ev], axis=1).reshape(-1)               # edge k contributes (eu -> ev) then (ev -> eu)
    nbrs = np.stack([ev, eu], axis=1).reshape(-1)
    order = np.argsort(rows, kind="stable")
    counts = np.bincount(rows, minlength=N).tolist()
    flat = torch.from_numpy(nbrs[order]).to(dev)
    data.neighbors = list(torch.split(flat, counts))
    data.neighbor_edges = [t.unsqueeze(0) for t in torch.split(torch.ones(2 * E, dtype=torch.int64, device=dev), counts)]
    wd = np.asarray(data.weighted_degree, dtype=np.float64)
    add = np.stack([1 - wd[eu] / 2 - 0.05, 1 - wd[ev] / 2 - 0.05, np.full(E, 1 + 0.05)]).astype(np.float32)
    data.add_items = torch.from_numpy(add).to(dev)
    absdeg = np.abs(wd).astype(np.float32)
    edge_degree = torch.from_numpy(absdeg[eu] + absdeg[ev])
    data.sorted_degree_edges = torch.argsort(edge_degree, descending=True, stable=True)
    return data


def make_data(num_nodes: int, eu, ev, device, sorted_degree_nodes=None):
    device = torch.device(device)
    eu = np.asarray(eu, dtype=np.int64)
    ev = np.asarray(ev, dtype=np.int64)
    csr = build_csr((eu, ev, np.ones_like(eu)), num_nodes=num_nodes, if_bidirectional=False)
    # the objective kernel walks the edge list in file order semantics (order is irrelevant to a sum)
    data = MCPGData()
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


def build_visit_stream(csr, order: np.ndarray, max_nodes: int = 32, max_entries: int = 400, weighted: bool = False) -> np.ndarray:
    """Flatten the visiting order for the streaming K7 kernel (format: rls_mcpg.hip / include/rlsolver_hip.h).

    The sequential pass (MCPG.py:136-142) only orders ADJACENT nodes, so the visiting positions are
    level-scheduled by the library's host pass (rls_graph_sweep_schedule, on the graph relabelled by visiting
    position): positions sorted by (dependency level, position), levels cut into batches of <= max_nodes
    nodes / <= max_entries stream entries.  Nodes of a batch are pairwise non-adjacent and every neighbour
    visited earlier sits in an earlier batch: deciding batch after batch reproduces the sequential result.

    ``weighted``: records carry the CSR weights (node, deg, Wfresh, position, Wdeg, then (neighbour, weight) pairs) for
    the weighted sampler of rlsolver/methods/MCPG/sampling.py:89-127."""
    import ctypes as C
    from .. import _abi
    n = csr.num_nodes
    hdr, ent = (5, 2) if weighted else (4, 1)
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
    wgt = csr.wgt[idx].astype(np.int64)
    row_pos = np.repeat(np.arange(n), deg_o)
    fresh = pos_of[nbr] > row_pos
    nfresh = np.bincount(row_pos, weights=fresh * (wgt if weighted else 1), minlength=n).astype(np.int64)
    wdeg = np.bincount(row_pos, weights=wgt, minlength=n).astype(np.int64)
    # level schedule over visiting positions (graph relabelled by position); a record is hdr - 1 entries longer than
    # the scheduler's (1 + deg) count per neighbour entry
    rp_p = np.ascontiguousarray(cum, dtype=np.int32)
    col_p = np.ascontiguousarray(pos_of[nbr], dtype=np.int32)
    flagged = np.empty(n + 1, dtype=np.int32)
    tmp = np.empty(nnz + n, dtype=np.int32)
    nb, nl = C.c_int64(0), C.c_int64(0)
    budget = max((max_entries - 3 * max_nodes - (hdr - 4) * max_nodes) // ent, 1)
    _abi.call("rls_graph_sweep_schedule", rp_p.ctypes.data_as(C.c_void_p), col_p.ctypes.data_as(C.c_void_p), n,
              max_nodes, budget, flagged.ctypes.data_as(C.c_void_p),
              tmp.ctypes.data_as(C.c_void_p), C.byref(nb), C.byref(nl))
    off = (flagged.view(np.uint32) & 0x7FFFFFFF).astype(np.int64)
    sp = tmp[off[:n]].astype(np.int64)                               # visiting position at schedule slot k
    is_start = (flagged[:n].view(np.uint32) >> 31).astype(bool)
    bid = np.cumsum(is_start) - 1                                   # batch of each slot
    first_slot = np.flatnonzero(is_start)
    m = np.diff(np.concatenate([first_slot, [n]]))                   # nodes per batch
    deg_s = deg_o[sp]
    rec_len = ent * deg_s + hdr
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
    if weighted:
        stream[rec_off + 4] = wdeg[sp]
    # rows: entries of visiting position sp[k] copied behind the header of its record
    j_in_row = np.arange(int(deg_s.sum())) - np.repeat(np.cumsum(deg_s) - deg_s, deg_s)
    src = np.repeat(cum[:-1][sp], deg_s) + j_in_row
    dst = np.repeat(rec_off + hdr, deg_s) + ent * j_in_row
    stream[dst] = nbr[src] | (fresh[src].astype(np.int64) << 31)
    if weighted:
        stream[dst + 1] = wgt[src]
    return (stream & 0xFFFFFFFF).astype(np.uint32).view(np.int32)


ACCEPT_ROWS = 64      # rows the kernels spread their per-round accept counts over (one row = thousands of atomics per address)


def _walk_chunks(samples, start, probs: TEN, max_transfer_time: int, Tmax: int, Cc: int, index, u, seed: int,
                 chain_ids=None, stats=None, total_chains: Optional[int] = None) -> None:
    """The rounds of metro_sampling in chunks, with the reference's stop rule (MCPG.py:103,115) evaluated on the device.
    A round accepts at most C proposals, so the cumulative count cannot reach C*T before round T: chunks inside the first T
    rounds are applied directly (one pass, counting as it goes; the first one reads the caller's start state and writes the
    result buffer).  Later chunks: dry pass -> accept counts -> stop round (rls_mcpg_metro_stop: one launch) -> apply.  Chunks
    after the stop round see a zero limit and return at once.  A chunk is T rounds, or fewer where a launch cannot take that many
    (the node-major kernels keep the accept counts in LDS beside the tile: 956 rounds at N = 20 000).
    A shard of the chains (``chain_ids``: their global ids; ``stats``: the Sharded object whose group the ranks share;
    ``total_chains``: the GLOBAL chain count): the stop rule is a statistic of the whole batch, so each chunk's per-round
    accept counts are summed over the ranks before the stop kernel reads them -- every rank then stops at the round the
    one-process run stops at."""
    device = probs.device
    st, sb, N, _ = mops._chains(samples, "samples")
    cap = mops.mcpg_metro_max_rounds(N, sb)
    if cap <= 0:
        raise RuntimeError(f"metro_sampling: {N} nodes do not fit this layout's tile in LDS")
    chunk = max(1, min(max_transfer_time, cap))
    target = (Cc if total_chains is None else total_chains) * max_transfer_time

    def counts(acc):        # what the stop kernel reads: this rank's rows, or one row of whole-batch counts
        return acc if stats is None else stats._global_sum(acc.sum(dim=0, keepdim=True))
    starts = list(range(0, Tmax, chunk))
    sizes = [min(chunk, Tmax - t0) for t0 in starts]
    accepts = torch.zeros((len(starts), ACCEPT_ROWS, chunk), dtype=torch.int64, device=device)
    ctl = torch.empty(3, dtype=torch.int64, device=device)            # {accepts so far, live, limit of the next dry pass}
    apply_limit = torch.empty(1, dtype=torch.int64, device=device)
    scratch = mops.metro_scratch(N, Cc, device) if sb == 0 else None     # the packed walk's draw windows, one buffer for every chunk
    for k, (t0, tk) in enumerate(zip(starts, sizes)):
        acc = accepts[k] if tk == chunk else accepts[k].reshape(-1)[: ACCEPT_ROWS * tk].view(ACCEPT_ROWS, tk)
        next_tk = sizes[k + 1] if k + 1 < len(sizes) else 0
        if t0 + tk <= max(1, max_transfer_time):                       # inside the first T rounds: direct
            mops.mcpg_metro_rounds(samples, probs, tk, index, u, seed, None, True, acc, t_offset=t0,
                                   samples_in=None if (k > 0 or samples is start) else start, chain_ids=chain_ids, scratch=scratch)
            mops.mcpg_metro_stop(counts(acc), target, 1 if k == 0 else 2, next_tk, ctl)
        else:
            mops.mcpg_metro_rounds(samples, probs, tk, index, u, seed, ctl[2:3], False, acc, t_offset=t0, chain_ids=chain_ids, scratch=scratch)
            mops.mcpg_metro_stop(counts(acc), target, 0, next_tk, ctl, apply_limit)
            mops.mcpg_metro_rounds(samples, probs, tk, index, u, seed, apply_limit, True, None, t_offset=t0, chain_ids=chain_ids, scratch=scratch)


def metro_sampling_packed(probs: TEN, start: PackedChains, max_transfer_time: int, num_chains: Optional[int] = None,
                          index: Optional[TEN] = None, u: Optional[TEN] = None, out: Optional[PackedChains] = None,
                          seed: Optional[int] = None, chain_ids=None, stats=None, total_chains: Optional[int] = None) -> PackedChains:
    """metro_sampling (MCPG.py:88-117) on bit-packed chains.  ``start`` may hold fewer chains than ``num_chains`` (a
    multiple of 64): chain c starts from chain c % start.num_chains, the reference's ``xs_bool.repeat(1, repeat_times)``.
    Up to 5*T proposal rounds per chain, stopping after the first round whose cumulative accept count reaches C*T --
    evaluated on the device (no host sync), where the reference syncs once per round.
    ``seed``: the kernel seed (default: one draw of torch's generator); ``chain_ids`` / ``stats`` / ``total_chains``: a shard
    of a larger batch, see _walk_chunks."""
    device = start.device
    Cc = start.num_chains if num_chains is None else num_chains
    N = start.num_nodes
    probs = probs.detach().to(device=device, dtype=torch.float32).contiguous()
    samples = out if out is not None else PackedChains.empty(N, Cc, device)
    Tmax = max_transfer_time * 5
    if index is not None:
        Tmax = min(Tmax, index.shape[0])
    if index is not None:
        seed = 0
    elif seed is None:
        seed = _seed_from_torch()
    if Tmax <= 0:   # no rounds (N < 10 gives T = int(N / 10) = 0): the start state, broadcast
        mops.mcpg_metro_rounds(samples, probs, 0, None, None, 0, None, True, None, samples_in=start)
        return samples
    _walk_chunks(samples, start, probs, max_transfer_time, Tmax, Cc, index, u, seed, chain_ids, stats, total_chains)
    return samples


def metro_sampling(probs: TEN, start_status: TEN, max_transfer_time: int, device=None,
                   index: Optional[TEN] = None, u: Optional[TEN] = None) -> TEN:
    """MCPG.py:88-117 with the reference's surface (float32 [N, C] in and out, the caller's tensor untouched): pack ->
    the packed walk -> unpack while the tile plus the producers' window fit LDS (N <= 16 000; at BA-10^4 / 2^18 chains
    the two surface conversions are 4.1 of the 5.3 ms), the node-major f32 kernel beyond that.  The sync-free form of a
    sampling round is metro_sampling_packed.  Up to 5*T proposal rounds per chain, stopping after the first round whose
    cumulative accept count reaches C*T -- evaluated on the device, where the reference syncs once per round.
    ``index``/``u`` ([>=5T, C]) replace the torch.randint / torch.rand draws (test hook)."""
    device = start_status.device if device is None else torch.device(device)
    if start_status.shape[0] * 8 + 2 * 64 * 64 * 4 + 16 <= 160 * 1024:
        start = start_status.to(device=device, dtype=torch.float32).contiguous()
        return metro_sampling_packed(probs, PackedChains.pack(start), max_transfer_time, index=index, u=u).unpack()
    return _metro_sampling_nodemajor(probs, start_status, max_transfer_time, device, index, u)


def _metro_sampling_nodemajor(probs: TEN, start_status: TEN, max_transfer_time: int, device, index: Optional[TEN],
                              u: Optional[TEN]) -> TEN:
    """metro_sampling on the node-major f32 kernel (16 loader waves overlap the 2 x 4N bytes per chain with the walk)."""
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
    _walk_chunks(samples, start, probs, max_transfer_time, Tmax, Cc, index, u, seed)
    return samples


def _levels_ok(data) -> bool:
    return getattr(data, '_lv_ptr', None) is not None and bool(
        _abi.lib().rls_mcpg_local_search_levels_supported(data.graph.ref, data._lv_ptr.numel() - 1))


def sampler_func_packed(data, xs: PackedChains, num_ls: int, total_mcmc_num: int, repeat_times: int,
                        num_chains: Optional[int] = None, in_place: bool = True, seed: Optional[int] = None, chain_ids=None):
    """sampler_func (MCPG.py:120-166) on bit-packed chains, production draws: level-parallel K7 (in place on ``xs``
    unless told otherwise), expected cut, best of repeats.  Returns (vs_good f32 [M], xs_good PackedChains of M
    chains, value f32 [C], xs after the local search)."""
    if not _levels_ok(data):
        raise _abi.RlsError("sampler_func_packed", -2, "the level-parallel K7 kernel does not cover this graph")
    Cc = xs.num_chains if num_chains is None else num_chains
    out = xs if (in_place and xs.num_chains == Cc) else PackedChains.empty(xs.num_nodes, Cc, xs.device)
    xs_loc, expected = mops.mcpg_local_search_levels(data.graph, xs, data._lv_ptr, data._lv_data, num_ls,
                                                     _seed_from_torch() if seed is None else seed,
                                                     out=out, num_chains=Cc, chain_ids=chain_ids)
    _, vs_good, xs_good = mops.mcpg_pick_best(expected, xs_loc, total_mcmc_num, repeat_times, data.num_edges)
    return vs_good, xs_good, expected - expected.mean(), xs_loc


def tie_coins_from_uniforms(data, uniforms: TEN) -> TEN:
    """Recorded torch.rand draws f32 [num_ls, N (visiting position), C] -> the tie coins of the level-parallel kernel,
    int64 words [num_ls * N, ceil(C / 64)].

    The draw of MCPG.py:139-141 only ever decides a TIE: the neighbour sum s is a multiple of 1/2, so
    (s + u/4) < (deg + 1/4)/2 is settled by s alone unless s == deg/2, and then it reads
    fl(deg/2 + fl(u/4)) < deg/2 + 1/8 in float32 -- "u < 1/2" except within a few ulps below 1/2, where the sum rounds
    up to the threshold (the larger deg, the wider that band).  The coin is that exact float32 expression per
    (position, chain), so the kernel reproduces the reference for EVERY recorded draw, not only away from 1/2."""
    num_ls, n, C = uniforms.shape
    deg = torch.tensor(data.single_degree, dtype=torch.float32, device=uniforms.device)[data.sorted_degree_nodes.to(uniforms.device)]
    k = torch.tensor(0.25, dtype=torch.float32, device=uniforms.device)
    half = (deg / 2).view(1, n, 1)
    thr = ((deg + k) / 2).view(1, n, 1)
    coin = (half + uniforms.to(torch.float32) * k) < thr
    words = PackedChains.pack(coin.reshape(num_ls * n, C).contiguous()).words          # [ceil(C / 64), num_ls * N]
    return words.t().contiguous()


def sampler_func(data, xs_sample: TEN, num_ls: int, total_mcmc_num: int, repeat_times: int, device=None,
                 uniforms: Optional[TEN] = None):
    """MCPG.py:120-166: node-sequential stochastic local search (K7), expected cut (K8), best of
    repeats.  ``uniforms`` f32 [num_ls, N, C] replaces torch.rand (test hook)."""
    xs_sample = xs_sample.contiguous()
    if _levels_ok(data):
        # production path: level-parallel kernel (the draws only ever decide ties, so it carries coins, not uniforms;
        # recorded draws become coins by the reference's own float32 expression); the f32 [N, C] input is read once,
        # everything after it is bit-packed, xs_good leaves as f32 [N, M]
        # (packing first and running the 8-wave packed kernel beats letting the kernel's own loader read the f32 surface:
        # 6.7 vs 8.5 ms at BA-10^4 / 2^18 chains)
        packed = PackedChains.pack(xs_sample)
        coins = None if uniforms is None else tie_coins_from_uniforms(data, uniforms)
        xs_loc, expected = mops.mcpg_local_search_levels(data.graph, packed, data._lv_ptr, data._lv_data, num_ls,
                                                         0 if uniforms is not None else _seed_from_torch(), coins=coins, out=packed)
        _, vs_good, xs_good = mops.mcpg_pick_best(expected, xs_loc, total_mcmc_num, repeat_times, data.num_edges)
        return vs_good, xs_good.unpack(), expected - expected.mean()
    xs_loc, expected = mops.mcpg_local_search(data.graph, xs_sample, data._order_i32, num_ls, uniforms,
                                              0 if uniforms is not None else _seed_from_torch(),
                                              visit_stream=getattr(data, '_visit_stream', None))
    _, vs_good, xs_good = mops.mcpg_pick_best(expected, xs_loc, total_mcmc_num, repeat_times, data.num_edges)
    value = expected - expected.mean()
    return vs_good, xs_good, value


class _ReturnFn(torch.autograd.Function):
    """get_return (MCPG.py:292-302) with the samples bit-packed: objective = mean_c(log_prob_sum_c * value_c) where
    log_prob_sum_c = sum_n log(s p + (1 - s)(1 - p)).  Forward and backward need only A_n = sum_c value_c s_nc
    (one kernel over the packed chains) and V = sum_c value_c."""

    @staticmethod
    def forward(ctx, probs, samples, value, sums=None):
        # sums = (A, V) of an earlier call on the same samples and value (they do not depend on probs: the eight policy epochs
        # of a round share them, MCPG.py:397-403)
        A, V = sums[:2] if sums is not None else (mops.mcpg_value_bit_sums(samples, value), value.sum())
        C = sums[2] if (sums is not None and len(sums) > 2) else samples.num_chains    # (a shard: the GLOBAL sums and chain count)
        lp, l1p = probs.log(), (1 - probs).log()
        ctx.save_for_backward(probs, A, V)
        ctx.C = C
        return (l1p.sum() * V + ((lp - l1p) * A).sum()) / C

    @staticmethod
    def backward(ctx, g):
        probs, A, V = ctx.saved_tensors
        return g * (A / probs - (V - A) / (1 - probs)) / ctx.C, None, None, None


def get_return(probs: TEN, samples, value: TEN, total_mcmc_num: int = 0, repeat_times: int = 0):
    """MCPG.py:292-302.  ``samples``: PackedChains (the metro output of the round), or the reference's float
    [C, N] tensor (packed on the way in)."""
    if not isinstance(samples, PackedChains):
        samples = PackedChains.pack(samples.t().contiguous())
    return _ReturnFn.apply(probs, samples, value.detach().to(torch.float32).contiguous())


def _get_column(pc: PackedChains, i) -> TEN:
    """Chain i of a PackedChains as bool [N]; i a python int or a 0-dim device tensor (no host read)."""
    i = torch.as_tensor(i, dtype=torch.int64, device=pc.device)
    return ((pc.words.index_select(0, (i // 64).reshape(1))[0] >> (i % 64)) & 1).bool()


def _set_column(pc: PackedChains, i: int, bits: TEN) -> None:
    sh = i % 64
    mask = torch.ones((), dtype=torch.int64, device=pc.device) << sh
    row = pc.words[i // 64]
    pc.words[i // 64] = (row & ~mask) | (bits.to(torch.int64) << sh)


class MCPGRound(Sharded):
    """One sampling round of the MCPG outer loop (methods/MCPG.py:366-413) kept on the device, sync-free:

        xs_sample = metro_sampling(xs_prob, xs_bool, change_times)                    K9, bit-packed
        temp_max, temp_max_info, value = sampler_func(...)                            K7 + K8, bit-packed
        best-merge into now_max_res / now_max_info, min/max replacement (:376-391)    rls_mcpg_merge_best
        xs_bool = temp_max_info.repeat(1, repeat_times)                               never materialised (C_in broadcast)
        get_return(xs_prob, xs_sample.t(), value, ...) for the policy update          rls_mcpg_value_bit_sums

    State: now_max_res f32 [M], now_max_info / start PackedChains of M chains, samples PackedChains of C = M * R.

    Sharded over ranks: the KEPT chains are split -- ``total_mcmc_num`` is this rank's share M_local
    (a multiple of 64), ``kept_offset`` the global index of its kept chain 0 (shards contiguous in rank order),
    ``total_kept`` the global M, ``group`` the process group.  All repeats of a kept chain live on its rank, so sampling,
    local search, best-of-repeats and the per-chain merge need no exchange, and every draw is keyed by the chain's GLOBAL id
    (rls_chain_ids) -- a shard's chains are bit for bit the chains of the one-process run.  What the reference computes over
    the WHOLE batch is reduced over the group: the walk's stop rule (per-round accept counts, MCPG.py:103,115), the mean of
    ``expected`` (:165), the best / worst incumbent of the min/max replacement (:383-391: C1 twice + C2 once per round) and
    get_return's two chain sums (:292-302).  Every rank then holds the same loss and gradient: the policy parameters stay in
    step without a gradient all-reduce."""

    def __init__(self, data, now_max_info, now_max_res: TEN, total_mcmc_num: int, repeat_times: int, num_ls: int,
                 change_times: Optional[int] = None, kept_offset: int = 0, total_kept: Optional[int] = None, group=None,
                 seed: Optional[int] = None):
        self._init_shard(kept_offset, seed, group)
        self.data, self.M, self.R, self.num_ls = data, total_mcmc_num, repeat_times, num_ls
        if total_mcmc_num % 64 != 0:
            raise ValueError("total_mcmc_num must be a multiple of 64 (one bit tile holds 64 chains)")
        self.M_total = total_mcmc_num if total_kept is None else int(total_kept)
        if self.M_total % 64 != 0 or kept_offset % 64 != 0 or kept_offset + total_mcmc_num > self.M_total:
            raise ValueError("kept_offset / total_kept: shards of the kept chains are whole 64-chain tiles inside [0, total_kept)")
        self.sharded = self.M_total != self.M or group is not None
        # global id of local chain c = repeat (c // M) * M_total + kept_offset + c % M
        self.chain_ids = (kept_offset, self.M, self.M_total - self.M) if self.M_total != self.M else None
        self.N = data.num_nodes
        self.change_times = int(self.N / 10) if change_times is None else change_times     # MCPG.py:331
        info = now_max_info if isinstance(now_max_info, PackedChains) else PackedChains.pack(now_max_info.contiguous())
        self.now_max_info = info
        self.now_max_res = now_max_res.to(torch.float32).clone()
        self.start = info.clone()                                                            # xs_bool, before the repeat
        packed_fits = mops.mcpg_metro_max_rounds(self.N, 0) > 0
        self.samples = PackedChains.empty(self.N, self.M * self.R, info.device) if packed_fits else None   # the round's metro output
        self.work = PackedChains.empty(self.N, self.M * self.R, info.device) if packed_fits else None      # after the local search
        self.value = None
        self._sums = None
        self.best_value = self.best_index = self.best_x = None
        # the bit-packed walk keeps a 32 KB window of draws beside the tile: up to N ~ 16 000.  Beyond (G81: 20 000 nodes) the round
        # runs on the node-major kernels through the f32 surface and packs what it keeps
        self._nodemajor = mops.mcpg_metro_max_rounds(self.N, 0) == 0
        if self._nodemajor and self.sharded:
            raise NotImplementedError("a sharded MCPGRound needs the bit-packed walk (N <= ~16 000)")
        if self.sharded and self.now_max_res.numel():
            # the incumbents travel between ranks as a packed MAXLOC key of DOUBLED integers (dist.global_best): cut values of
            # this build's integer-weighted graphs are integers or half-integers.  Anything else would trip a device-side assert
            # rounds later -- refuse it here, by name (one host read, at construction only)
            twice = self.now_max_res.to(torch.float64) * 2
            if not bool(((twice == twice.round()) & (twice.abs() < float(1 << 43))).all()):
                raise ValueError("a sharded MCPGRound exchanges its incumbents as integer MAXLOC keys: now_max_res must hold integers "
                                 "or half-integers below 2^42 in magnitude (cut values of an integer-weighted graph); run graphs "
                                 "with non-integer weights unsharded")

    def _global_best(self, values: TEN, rows: Optional[PackedChains], maximize: bool):
        """(best value 0-dim, GLOBAL index 0-dim int64, its chain bool [N] or None) over every rank's ``values``; first index on
        ties (shards are contiguous in rank order)."""
        vs = values if maximize else -values
        row_of = (lambda li: _get_column(rows, li)) if rows is not None else None
        if self.stat_hook is not None:
            return self.stat_hook("best", (vs, row_of, self.env_offset, maximize))
        from .. import dist
        v, _, x, gi = dist.global_best(vs, row_of, want_solution=rows is not None, group=self.group,
                                       env_offset=self.env_offset, num_nodes=self.N)
        return (v if maximize else -v), gi, x

    def _replace_global_worst(self, temp_info: PackedChains) -> None:
        """MCPG.py:383-391 over the WHOLE batch: the first worst incumbent (and the start state of its chain) becomes the
        first best one.  C1 for the best (+ C2: its chain, N / 8 bytes), C1 for the worst; the owner of the worst rewrites it."""
        hi_v, hi_g, x_hi = self._global_best(self.now_max_res, self.now_max_info, True)
        _, lo_g, _ = self._global_best(self.now_max_res, None, False)
        lo = int(lo_g) - self.env_offset                                  # (a host read: the sharded round has them anyway)
        if 0 <= lo < self.M:
            self.now_max_res[lo] = hi_v.to(torch.float32)
            _set_column(self.now_max_info, lo, x_hi)
            _set_column(temp_info, lo, x_hi)
        self.best_value = hi_v.to(torch.float32).reshape(1)
        self.best_index, self.best_x = hi_g.reshape(1), x_hi

    def step(self, xs_prob: TEN):
        """One round; returns (value f32 [C], best value so far f32 [1]) -- device tensors, nothing is read back (a sharded
        round reads the two owners of its exchange)."""
        C = self.M * self.R
        if self.M_total != self.M and self.group is None and self.stat_hook is None:
            raise RuntimeError("this MCPGRound holds a shard of the kept chains (total_kept > total_mcmc_num): it needs group= (the process "
                               "group of the other shards) for the whole-batch statistics -- without it the stop rule, the mean and the "
                               "best / worst incumbent would silently be those of the shard")
        if self._nodemajor:
            xs_sample = metro_sampling(xs_prob, self.start.unpack().repeat(1, self.R), self.change_times)
            temp_max, temp_f32, value = sampler_func(self.data, xs_sample, self.num_ls, self.M, self.R)
            self.samples = PackedChains.pack(xs_sample)
            temp_info = PackedChains.pack(temp_f32.contiguous())
            self.best_value, self.best_index = mops.mcpg_merge_best(temp_max, temp_info, self.now_max_res, self.now_max_info)
            self.start, self.value, self._sums = temp_info, value, None
            return self.value, self.best_value
        Cg = self.M_total * self.R
        metro_sampling_packed(xs_prob, self.start, self.change_times, num_chains=C, out=self.samples, seed=self._next_seed(),
                              chain_ids=self.chain_ids, stats=self if self.sharded else None, total_chains=Cg)
        xs_loc, expected = mops.mcpg_local_search_levels(self.data.graph, self.samples, self.data._lv_ptr, self.data._lv_data,
                                                         self.num_ls, self._next_seed(), out=self.work, chain_ids=self.chain_ids)
        _, temp_max, temp_info = mops.mcpg_pick_best(expected, xs_loc, self.M, self.R, self.data.num_edges)
        if not self.sharded:
            self.best_value, self.best_index = mops.mcpg_merge_best(temp_max, temp_info, self.now_max_res, self.now_max_info)
            self.value = expected - expected.mean()
        else:
            mops.mcpg_merge_best(temp_max, temp_info, self.now_max_res, self.now_max_info, replace_worst=False)
            self._replace_global_worst(temp_info)
            total = self._global_sum(expected.sum(dtype=torch.float64).reshape(1))          # the mean over the whole batch (:165)
            self.value = expected - (total[0] / Cg).to(torch.float32)
        self.expected = expected
        self.start = temp_info
        self._sums = None
        return self.value, self.best_value

    def get_return(self, xs_prob: TEN):
        """get_return of the round's samples; the two sums over the chains are formed once per round, not once per epoch (and
        over every rank's chains when the round is a shard)."""
        if self._sums is None:
            A, V = mops.mcpg_value_bit_sums(self.samples, self.value), self.value.sum()
            if self.sharded:
                tot = self._global_sum(torch.cat([A.to(torch.float64), V.to(torch.float64).reshape(1)]))
                self._sums = (tot[:-1].to(torch.float32), tot[-1].to(torch.float32), self.M_total * self.R)
            else:
                self._sums = (A, V)
        return _ReturnFn.apply(xs_prob, self.samples, self.value, self._sums)

    def best_solution(self):
        """(value: float, x: bool [N]) of the best incumbent -- a host read, for the end of a run."""
        if self.sharded:
            return float(self.best_value.item()), self.best_x
        i = int(self.best_index.item())
        word = self.now_max_info.words[i // 64]
        return float(self.best_value.item()), ((word >> (i % 64)) & 1).bool()


def run_mcpg(data, xs_init: TEN, vs_init: TEN, total_mcmc_num: int, repeat_times: int, num_ls: int, num_rounds: int,
             sample_epoch_num: int = 8, lr: float = 8e-2, log=print, kept_offset: int = 0, total_kept: Optional[int] = None,
             group=None, seed: Optional[int] = None):
    """The sampling loop of mcpg() (methods/MCPG.py:353-413) on MCPGRound, with the reference's per-round prints
    ("value ... entropy ..." and "num_samples_per_second: ..." as defined at :405-411: kept chains per round divided
    by the wall time of metro + sampler + merge).  ``xs_init`` [N, M] / ``vs_init`` [M]: the incumbents the reference
    gets from LocalSearch (:337-346).  The policy is the reference's parameter vector through a sigmoid (Simpler,
    :62-73) trained with Adam: dense torch, out of the hot path.  Returns (best value, best x bool [N], samples/s list).
    ``kept_offset`` / ``total_kept`` / ``group``: this process runs a shard of the kept chains (see MCPGRound); value, best x
    and the policy are those of the whole batch on every rank."""
    import time
    device = data.graph.device
    rnd = MCPGRound(data, xs_init, vs_init, total_mcmc_num, repeat_times, num_ls, kept_offset=kept_offset,
                    total_kept=total_kept, group=group, seed=seed)
    lin = torch.nn.Parameter(torch.zeros(data.num_nodes, device=device))
    opt = torch.optim.Adam([lin], lr=lr)
    xs_prob = torch.full((data.num_nodes,), 0.5, device=device)
    rates = []
    for r in range(num_rounds):
        torch.cuda.synchronize(device)
        t0 = time.time()
        _, best = rnd.step(xs_prob)
        p = xs_prob[None, :]
        entropy = -(p * p.log2() + (1 - p) * (1 - p).log2()).mean(dim=1).mean()
        now_max = float(best.item())                                   # the reference reads it for the print, too
        running = time.time() - t0
        log(f"value {now_max: 9.2f}  entropy {float(entropy): 9.3f}")
        rates.append(rnd.M_total / running)
        log("num_samples_per_second: ", rates[-1])
        for _ in range(sample_epoch_num):
            xs_prob = torch.sigmoid(lin)
            loss = rnd.get_return(xs_prob)
            opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_([lin], 1)
            opt.step()
        xs_prob = torch.sigmoid(lin).detach()
    v, x = rnd.best_solution()
    return v, x, rates

"""Weighted MaxCut sampler of the upstream MCPG package -- drop-in for
rlsolver/methods/MCPG/sampling.py:89-127 (mcpg_sampling_maxcut) and the graph part of its loader
rlsolver/methods/MCPG/dataloader.py:53-103, on HIP tensors.

    metro_sampling -> gauge fix (every chain XORed with its value at the node of largest |weighted degree|, :101-104)
    -> num_ls node-sequential passes  x_i <- [sum_j w_ij v_j + U / 4 < Wdeg_i / 2 + 0.125]  (:108-118; v holds
    -0.5 | 1.5 until a node's first update) -> sum_e w_e (2x_u - 1)(2x_v - 1), best of the repeats per kept chain.

One kernel for everything after the metro step (rls_mcpg_local_search with edge weights: the batched visit
stream with (neighbour, weight) pairs).  Integer edge weights (Gset's +-1 instances, the reference's data files).
"""
from __future__ import annotations

import types
from typing import Optional

import numpy as np
import torch

from .. import ops, ops_mcpg_tsp as mops
from ..graph import build_csr, read_edge_arrays
from .MCPG import _seed_from_torch, build_visit_stream, metro_sampling

TEN = torch.Tensor


def maxcut_dataloader(path, device):
    """dataloader.py:53-103: (data, num_nodes); data carries edge_index [2, E], edge_attr f32 [E, 1], edge_weight_sum,
    weighted_degree (sum of incident weights), sorted_degree_nodes (by |weights| sum, descending)."""
    num_nodes, eu, ev, w = read_edge_arrays(path)
    return make_data(num_nodes, eu, ev, w, device), num_nodes


def make_data(num_nodes: int, eu, ev, w, device, sorted_degree_nodes=None):
    device = torch.device(device)
    eu, ev = np.asarray(eu, dtype=np.int64), np.asarray(ev, dtype=np.int64)
    w = np.asarray(w)
    if np.any(w != np.round(w)):
        raise ValueError("the weighted MCPG sampler takes integer edge weights")
    wi = w.astype(np.int64)
    csr = build_csr((eu, ev, wi), num_nodes=num_nodes, if_bidirectional=False)
    data = types.SimpleNamespace()
    data.num_nodes = num_nodes
    data.edge_index = torch.from_numpy(np.stack([eu, ev])).to(device)
    data.edge_attr = torch.from_numpy(wi.astype(np.float32))[:, None].to(device)
    data.num_edges = int(eu.shape[0])
    data.edge_weight_sum = float(wi.sum())
    wdeg = np.zeros(num_nodes, np.int64)
    np.add.at(wdeg, eu, wi)
    np.add.at(wdeg, ev, wi)
    adeg = np.zeros(num_nodes, np.int64)
    np.add.at(adeg, eu, np.abs(wi))
    np.add.at(adeg, ev, np.abs(wi))
    data.weighted_degree = wdeg.astype(np.float64).tolist()
    data.single_degree = np.bincount(np.concatenate([eu, ev]), minlength=num_nodes).tolist()
    if sorted_degree_nodes is None:   # torch.argsort(descending=True) is not stable; any tie order is a valid reference outcome
        sorted_degree_nodes = torch.argsort(torch.from_numpy(adeg.astype(np.float64)), descending=True, stable=True)
    data.sorted_degree_nodes = torch.as_tensor(sorted_degree_nodes).to(torch.int64)
    data.graph = ops.DeviceGraph(csr, device, use_weights=True)
    order = data.sorted_degree_nodes.cpu().numpy()
    data._order_i32 = data.sorted_degree_nodes.to(device=device, dtype=torch.int32).contiguous()
    data._visit_stream = torch.from_numpy(build_visit_stream(csr, order, weighted=True)).to(device)
    data._edge_w = torch.from_numpy(csr.ew.astype(np.int32)).to(device)       # in the stored (eu, ev) order
    return data


def mcpg_sampling_maxcut(data, start_result: TEN, probs: TEN, num_ls: int, change_times: int, total_mcmc_num: int,
                         device=None, index: Optional[TEN] = None, u: Optional[TEN] = None, uniforms: Optional[TEN] = None):
    """sampling.py:89-127 -> ((edge_weight_sum - best expected) / 2 f32 [M], chains of the best repeats f32 [N, M],
    the metro output f32 [N, C], expected - mean f32 [C]).  ``index`` / ``u`` / ``uniforms`` replace the torch draws of
    the metro step and of the local search (test hooks)."""
    device = start_result.device if device is None else torch.device(device)
    start = metro_sampling(probs, start_result, change_times, device, index=index, u=u)
    C = start.shape[1]
    hub = int(data.sorted_degree_nodes[0])
    xs_loc, expected = mops.mcpg_local_search(data.graph, start, data._order_i32, num_ls, uniforms,
                                              0 if uniforms is not None else _seed_from_torch(), visit_stream=data._visit_stream,
                                              edge_weights=data._edge_w, gauge_node=hub)
    _, vs_good, xs_good = mops.mcpg_pick_best(expected, xs_loc, total_mcmc_num, C // total_mcmc_num, 0)
    vs_good = vs_good + data.edge_weight_sum / 2.0        # pick_best returns (0 - best) / 2; (W - best) / 2 = that + W / 2
    return vs_good, xs_good, start, expected - expected.mean()

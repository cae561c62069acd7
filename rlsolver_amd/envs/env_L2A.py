"""EnvMaxcut, L2A / MCPG flavour -- drop-in for rlsolver/envs/env_L2A.py:24-116
(== env_MCPG.py:24-116, the simulators of env_k_spin.py).

Same constructor keywords, attributes, method names, shapes and dtypes; every method that
touches spins launches a HIP kernel (rlsolver_amd.ops).  Differences, all deliberate:

* ``device`` must be a HIP device: there is no CPU path;
* no per-env index tensors are materialised (the reference caches three int64 [B, E'] tensors);
  ``n0_ids`` / ``n1_ids`` keep their [1, E'] form for callers that read them;
* ``adjacency_bool`` (dense N x N) is built lazily on first access instead of in ``__init__``.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch as th

from .. import ops
from ..graph import MyGraph, build_csr, calc_num_nodes_in_mygraph, load_mygraph2

TEN = th.Tensor


def _seed_from_torch() -> int:
    # consume torch's CPU generator so th.manual_seed() makes kernel RNG reproducible
    return int(th.randint(0, 2 ** 62, (1,), dtype=th.int64).item())


class EnvMaxcut:
    def __init__(self, sim_name: str = 'max_cut', mygraph: MyGraph = (),
                 device=th.device('cpu'), if_bidirectional: bool = False, num_nodes: int = 0):
        self.device = th.device(device)
        if self.device.type != 'cuda':
            raise TypeError(f"rlsolver_amd.EnvMaxcut needs a HIP device (got {self.device}); "
                            "there is no CPU path")
        self.sim_name = sim_name
        self.int_type = th.long
        self.if_maximize = True
        self.if_bidirectional = if_bidirectional

        mygraph = mygraph if mygraph else load_mygraph2(graph_name=sim_name)
        self._mygraph = mygraph
        # the reference sizes the env by distinct endpoints (util.py:35-40); ``num_nodes`` lets a
        # caller that knows the file header keep isolated nodes
        self.num_nodes = num_nodes if num_nodes else calc_num_nodes_in_mygraph(mygraph)
        self.num_edges = len(mygraph)
        csr = build_csr(mygraph, num_nodes=self.num_nodes, if_bidirectional=if_bidirectional)
        self.graph = ops.DeviceGraph(csr, self.device)  # weights are ignored, as in the reference
        self.n0_ids = self.graph.eu.to(th.long)[None, :]
        self.n1_ids = self.graph.ev.to(th.long)[None, :]
        counts = np.bincount(csr.eu, minlength=self.num_nodes) if csr.eu.size else np.zeros(self.num_nodes, np.int64)
        self.n0_num_n1 = th.from_numpy(counts.astype(np.int64)).to(self.device)[None, :]
        self._adjacency_indies = None
        self._adjacency_bool = None
        self.fused_local_search = True   # False: K2 + torch weights/noise/kthvalue + K6 + K5 as separate launches

    # ---- lazily built forms of the reference attributes
    @property
    def adjacency_indies(self):
        if self._adjacency_indies is None:
            erp = self.graph.erowptr.cpu().numpy()
            ev = self.graph.ev.to(th.long)
            self._adjacency_indies = [ev[erp[i]:erp[i + 1]] for i in range(self.num_nodes)]
        return self._adjacency_indies

    @property
    def adjacency_bool(self):
        if self._adjacency_bool is None:
            csr = self.graph.csr  # build_adjacency_bool(mygraph, if_bidirectional=True), env_L2A.py:37
            adj = np.zeros((self.num_nodes, self.num_nodes), dtype=bool)
            adj[np.repeat(np.arange(self.num_nodes), np.diff(csr.rowptr)), csr.col] = True
            self._adjacency_bool = th.from_numpy(adj).to(self.device)
        return self._adjacency_bool

    # ---- objective
    def calculate_obj_values(self, xs: TEN, if_sum: bool = True) -> TEN:
        """env_L2A.py:54-66.  int64 [B] (bool [B, E'] when if_sum=False)."""
        if if_sum:
            return ops.maxcut_obj(self.graph, xs)
        values = ops.maxcut_edge_cut_mask(self.graph, xs)
        if self.if_bidirectional:
            values = values // 2
        return values

    def calculate_obj_values_for_loop(self, xs: TEN, if_sum: bool = True) -> TEN:
        """env_L2A.py:68-80: per-node cut degree (int64), summed if asked; float / 2 when
        bidirectional -- the reference's dtype quirk is kept."""
        values = ops.maxcut_node_cutdeg(self.graph, xs)
        if if_sum:
            values = values.sum(dim=1)
        if self.if_bidirectional:
            values = values.float() / 2
        return values

    def generate_xs_randomly(self, num_sims):
        """env_L2A.py:82-85: Bernoulli(1/2) spins, node 0 := 0 (Philox kernel seeded from torch)."""
        return ops.rand_spins(num_sims, self.num_nodes, _seed_from_torch(), self.device)

    # ---- local search
    def local_search_inplace(self, good_xs: TEN, good_vs: TEN,
                             num_iters: int = 8, num_spin: int = 8, noise_std: float = 0.3,
                             noise: Optional[TEN] = None):
        """env_L2A.py:87-116.  The weights / noise / kthvalue threshold are the reference's own
        [B, N] torch ops (they consume torch's generator the same way); each proposal round
        (clone + masked flip + objective + keep-if-not-worse) is ONE kernel, and the N-iteration
        'addition' loop is ONE sequential O(E) sweep kernel instead of N objective evaluations.

        ``noise`` (f32 [num_iters + 1, B, N]) replaces the randn_like draws -- test hook."""
        compute_vs = good_vs.shape == ()
        if self.fused_local_search and ops.local_search_fusable(self.graph, num_spin, good_xs.shape[0]):
            # pre-pass kernel (weights + whole-batch max/min), then ONE kernel: threshold selection,
            # num_iters proposal rounds, greedy sweep, with the 64-env tile resident in LDS
            ws32, ws_std = ops.maxcut_ls_weights(self.graph, good_xs, 1)   # n0_num_n1 - k * vs_raw, exact in both flavours
            rd_std = ws_std.float() * noise_std
            good_vs = th.empty(good_xs.shape[0], dtype=th.long, device=self.device) if compute_vs else good_vs.long()
            if noise is not None:
                noise = noise.to(device=self.device, dtype=th.float32).contiguous()
            ops.maxcut_local_search(self.graph, good_xs, ws32, rd_std.contiguous(), good_vs, num_iters, num_spin,
                                    noise=noise, seed=0 if noise is not None else _seed_from_torch(),
                                    first_draw_proposes=False, compute_obj=compute_vs)
            return good_xs, good_vs

        vs_raw = self.calculate_obj_values_for_loop(good_xs, if_sum=False)
        ws = self.n0_num_n1 - (2 if self.if_bidirectional else 1) * vs_raw
        ws_std = ws.max(dim=0, keepdim=True)[0] - ws.min(dim=0, keepdim=True)[0]
        rd_std = ws_std.float() * noise_std
        good_vs = vs_raw.sum(dim=1).long() if compute_vs else good_vs.long()
        draw = (lambda i: noise[i]) if noise is not None else (lambda i: th.randn_like(ws, dtype=th.float32))
        spin_rand = ws + draw(0) * rd_std
        thresh = th.kthvalue(spin_rand, k=self.num_nodes - num_spin, dim=1)[0][:, None]

        for it in range(num_iters):
            spin_rand = ws + draw(1 + it) * rd_std
            spin_mask = spin_rand.gt(thresh)
            ops.maxcut_propose_accept(self.graph, good_xs, spin_mask, good_vs)

        ops.maxcut_greedy_sweep(self.graph, good_xs, good_vs)
        return good_xs, good_vs

"""Graph generators of the batched spin-system env -- the surface of rlsolver/methods/ECO_S2V/src/envs/util_envs_PECO.py
(RandomERGraphGenerator :15-57, RandomBAGraphGenerator :60-113, ValidationGraphGenerator / SetGraphGenerator :115-172) for
``SpinSystem(graph_generator=...)``: ``get()`` returns the couplings of all envs, float [num_envs, n_spins, n_spins] on the
device, a fresh draw per call.

The reference builds them from torch ops (the BA generator loops over the nodes in Python: per node a full [B, N, N] row
sum, a multinomial and two scatters); here a draw is ONE kernel (rls_rand_couplings).  Draws come from a counter-based
generator seeded from torch's generator at every call -- the same distributions, not torch's streams.
"""
from __future__ import annotations

from enum import Enum

import torch

from ..ops import _s64, _t
from .env_L2A import _seed_from_torch


class EdgeType(Enum):  # ECO_S2V/src/envs/util_envs.py:11-14
    UNIFORM = 1
    DISCRETE = 2
    RANDOM = 3


class GraphGenerator:
    """util_envs.py:62-81: what the env reads is n_spins, biased and get()."""

    def __init__(self, n_spins, edge_type, biased=False, num_envs=None):
        if biased:
            raise NotImplementedError("biased generators are not part of the MaxCut path")
        self.n_spins, self.edge_type, self.biased, self.num_envs = n_spins, edge_type, biased, num_envs

    def get(self, with_padding=False):
        raise NotImplementedError


class _KernelGenerator(GraphGenerator):
    _kind = None

    def __init__(self, n_spins, edge_type, num_envs, device, dtype=torch.float32, env_offset=0):
        super().__init__(n_spins, EdgeType(edge_type) if not isinstance(edge_type, EdgeType) else edge_type, False, num_envs)
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type != "cuda":
            raise TypeError(f"{type(self).__name__} needs a HIP device (got {self.device}); there is no CPU path")
        self.dtype, self.env_offset = dtype, env_offset
        self._p, self._m = 0.0, 0

    def get(self, with_padding=False, seed=None):
        out = torch.empty((self.num_envs, self.n_spins, self.n_spins), dtype=self.dtype, device=self.device)
        _t.rand_couplings(out, self._kind, float(self._p), int(self._m), self.edge_type.value,
                          _s64(_seed_from_torch() if seed is None else int(seed)), self.env_offset)
        return out


class RandomERGraphGenerator(_KernelGenerator):
    """util_envs_PECO.py:15-57: every pair an edge with probability p_connection, signs by edge_type."""
    _kind = 0

    def __init__(self, n_spins=20, p_connection=0.2, edge_type=EdgeType.DISCRETE, num_envs=8, device="cuda", **kw):
        super().__init__(n_spins, edge_type, num_envs, device, **kw)
        self.p_connection = self._p = p_connection


class RandomBAGraphGenerator(_KernelGenerator):
    """util_envs_PECO.py:60-113: seed clique on m + 1 nodes (self-loops included, as there), m degree-proportional edges per
    later node, signs by edge_type."""
    _kind = 1

    def __init__(self, n_spins=20, m_insertion_edges=4, edge_type=EdgeType.DISCRETE, num_envs=8, device="cuda", **kw):
        super().__init__(n_spins, edge_type, num_envs, device, **kw)
        self.m_insertion_edges = self._m = m_insertion_edges


class ValidationGraphGenerator(GraphGenerator):
    """util_envs_PECO.py:115-136: num_envs fixed networkx graphs -- barabasi_albert_graph(n, 4, seed) / erdos_renyi_graph(n, 0.15,
    seed) with seeds seed, seed + 1, ... -- as unit-weight matrices, zero diagonal; the same batch at every get().  Built on
    the host once (the reference rebuilds it per call)."""

    def __init__(self, device, n_spins=20, edge_type=EdgeType.DISCRETE, num_envs=2 ** 3, seed=None, graph_type="BA"):
        super().__init__(n_spins, edge_type, False, num_envs)
        import networkx as nx
        import numpy as np
        kind = getattr(graph_type, "name", graph_type)
        if kind not in ("BA", "ER"):
            raise NotImplementedError(f"graph_type {graph_type}")
        if seed is None:
            raise ValueError("ValidationGraphGenerator needs a seed (the reference returns uninitialised memory without one)")
        mats = []
        for k in range(num_envs):
            g = nx.barabasi_albert_graph(n_spins, 4, seed=seed + k) if kind == "BA" else nx.erdos_renyi_graph(n_spins, 0.15, seed=seed + k)
            m = nx.to_numpy_array(g).astype("float32")
            np.fill_diagonal(m, 0)
            mats.append(m)
        self.device, self.seed, self.graph_type = torch.device(device), seed, graph_type
        self.adj = torch.from_numpy(np.stack(mats)).to(self.device)

    def get(self, with_padding=False):
        return self.adj


class SetGraphGenerator(GraphGenerator):
    """A fixed batch of matrices [num_envs, n_spins, n_spins] (a tensor, or a list of [n_spins, n_spins]; or ONE [n_spins, n_spins]
    tensor: the shared graph of the inference env) handed back at every
    get() (util_envs_PECO.py:139-171; validation sets and the instance-wise inference batches, inference_PECO.py:84).  The
    edge type is read off the entries as there; biased sets are outside the MaxCut path."""

    def __init__(self, matrices, biases=None, ordered=False, device=None):
        if biases is not None:
            raise NotImplementedError("biased graphs (SpinSystemBiased) are not part of the MaxCut path")
        if isinstance(matrices, (list, tuple)):
            if len({tuple(torch.as_tensor(m).shape) for m in matrices}) != 1:
                raise NotImplementedError("All graphs in SetGraphGenerator must have the same dimension.")
            m = torch.stack([torch.as_tensor(x) for x in matrices])
        else:
            m = torch.as_tensor(matrices)
        if m.dim() not in (2, 3) or m.shape[-1] != m.shape[-2]:
            raise ValueError("matrices must be [num_envs, n_spins, n_spins], or one [n_spins, n_spins] graph for the inference env")
        vals = torch.unique(m)
        if bool(torch.isin(vals, torch.tensor([0, 1], dtype=vals.dtype, device=vals.device)).all()):
            edge_type = EdgeType.UNIFORM
        elif bool(torch.isin(vals, torch.tensor([0, -1, 1], dtype=vals.dtype, device=vals.device)).all()):
            edge_type = EdgeType.DISCRETE
        else:
            edge_type = EdgeType.RANDOM
        # one [N, N] tensor: the single shared graph of envs/inference_network_env.py (inference_PECO.py:84 passes exactly that)
        super().__init__(int(m.shape[-1]), edge_type, False, int(m.shape[0]) if m.dim() == 3 else None)
        self.matrices = m.to(device) if device is not None else m
        self.graphs = self.matrices
        self.ordered = ordered
        if ordered:
            self.i = 0

    def get(self, with_padding=False):
        return self.matrices

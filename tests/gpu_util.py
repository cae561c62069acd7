"""Helpers shared by the -m gpu parity tests (HIP path vs oracle)."""
import numpy as np
import torch

from rlsolver_amd import ops
from rlsolver_amd.graph import build_csr, generate_gnm

DEV = torch.device("cuda:0")


def device_graph(graph_arr, num_nodes, bidir, use_weights=False):
    g = np.asarray(graph_arr, dtype=np.int64).reshape(-1, 3)
    csr = build_csr((g[:, 0].copy(), g[:, 1].copy(), g[:, 2].copy()), num_nodes=num_nodes, if_bidirectional=bool(bidir))
    return ops.DeviceGraph(csr, DEV, use_weights=use_weights)


def to_dev_bool(a):
    return torch.from_numpy(np.ascontiguousarray(a).astype(np.uint8)).to(DEV).view(torch.bool) if a.dtype != bool \
        else torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def gnm_arr(n, m, seed):
    return np.asarray(generate_gnm(n, m, seed), dtype=np.int64)

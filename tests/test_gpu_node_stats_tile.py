"""K2/K3 have three implementations (B >= 2048: bit-sliced lane = node kernel on unweighted graphs of max degree
< 256, lane = env bit-tile kernel otherwise; element-parallel below 2048 envs): they must agree with each other
and with the oracle, incl. ragged N, weights, bidirectional, hubs."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as onp
from rlsolver_amd import ops
from tests.gpu_util import DEV, device_graph, gnm_arr, to_dev_bool

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,m,B", [(333, 2000, 2100), (2000, 19990, 2048), (64, 300, 4096), (1000, 3000, 2049), (100, 384, 2050)])
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("bidir", [0, 1])
def test_tile_and_element_kernels_agree(n, m, B, weighted, bidir):
    graph = gnm_arr(n, m, seed=n)
    rng = np.random.RandomState(B)
    if weighted:
        graph[:, 2] = rng.choice([-3, -1, 1, 2], size=len(graph))
    g = device_graph(graph, n, bidir, use_weights=weighted)
    xs = rng.randint(0, 2, size=(B, n)).astype(np.uint8)
    x = to_dev_bool(xs)
    d_tile = ops.maxcut_delta_all(g, x)                       # B >= 2048 -> tile kernel
    c_tile = ops.maxcut_node_cutdeg(g, x)
    d_el = torch.cat([ops.maxcut_delta_all(g, x[i:i + 1000].contiguous()) for i in range(0, B, 1000)])
    c_el = torch.cat([ops.maxcut_node_cutdeg(g, x[i:i + 1000].contiguous()) for i in range(0, B, 1000)])
    assert torch.equal(d_tile, d_el) and torch.equal(c_tile, c_el)
    sub = rng.choice(B, 6, replace=False)
    assert np.array_equal(d_tile[sub].cpu().numpy(), onp.maxcut_delta_all(xs[sub], graph, n, graph[:, 2] if weighted else None))
    assert np.array_equal(c_tile[sub].cpu().numpy(), onp.maxcut_node_cutdeg(xs[sub], graph, n, bool(bidir)))
    if not weighted:
        # local-search weights ws = stored_deg - mult * cutdeg through the same kernel family
        for mult in (1, 2):
            ws = ops.maxcut_ls_weights(g, x, mult)[0]
            deg = torch.from_numpy(np.bincount(g.csr.eu, minlength=n)).to(DEV)
            assert torch.equal(ws.long(), deg[None, :] - mult * c_tile)


def test_hub_graph_falls_back_to_lane_env_kernel():
    """A star centre of degree 300 (>= 256: byte counters of the bit-sliced kernel would overflow)."""
    n, B = 400, 2048
    graph = np.array([(0, j, 1) for j in range(1, 301)] + [(j, j + 1, 1) for j in range(1, 399)], dtype=np.int64)
    g = device_graph(graph, n, 0)
    xs = np.random.RandomState(1).randint(0, 2, size=(B, n)).astype(np.uint8)
    x = to_dev_bool(xs)
    sub = np.arange(0, B, 300)
    assert np.array_equal(ops.maxcut_delta_all(g, x)[sub].cpu().numpy(), onp.maxcut_delta_all(xs[sub], graph, n, None))
    assert np.array_equal(ops.maxcut_node_cutdeg(g, x)[sub].cpu().numpy(), onp.maxcut_node_cutdeg(xs[sub], graph, n, False))


def test_degree_200_graph_uses_all_eight_counter_planes():
    n, B = 512, 2100
    graph = gnm_arr(n, 40000, seed=5)          # mean degree 156, max ~190
    g = device_graph(graph, n, 0)
    assert 128 <= g.csr.max_degree < 256
    xs = np.random.RandomState(2).randint(0, 2, size=(B, n)).astype(np.uint8)
    x = to_dev_bool(xs)
    sub = np.arange(0, B, 211)
    assert np.array_equal(ops.maxcut_delta_all(g, x)[sub].cpu().numpy(), onp.maxcut_delta_all(xs[sub], graph, n, None))
    assert np.array_equal(ops.maxcut_node_cutdeg(g, x)[sub].cpu().numpy(), onp.maxcut_node_cutdeg(xs[sub], graph, n, False))

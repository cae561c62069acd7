"""K2/K3 have three implementations (bit-sliced lane = node kernel on unweighted graphs, lane = env bit-tile kernel on weighted
ones, element-parallel for batches too small to pay for a tile: ops.node_stats_form says which one a batch takes): they must agree
with each other and with the oracle, incl. ragged N, weights, bidirectional, hubs."""
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
    big = "tile" if weighted else "bits"
    while ops.node_stats_form(g, B, True) != big or ops.node_stats_form(g, B, False) != big:
        B = B * 3 // 2 + 1                                        # (the weighted tile form pays from a few thousand envs on)
    assert B < 40000 and ops.node_stats_form(g, 100, True) == ops.node_stats_form(g, 100, False) == "elem"
    xs = rng.randint(0, 2, size=(B, n)).astype(np.uint8)
    x = to_dev_bool(xs)
    d_tile = ops.maxcut_delta_all(g, x)                       # the bit-sliced / tile kernel
    c_tile = ops.maxcut_node_cutdeg(g, x)
    d_el = torch.cat([ops.maxcut_delta_all(g, x[i:i + 100].contiguous()) for i in range(0, min(B, 1500), 100)])
    c_el = torch.cat([ops.maxcut_node_cutdeg(g, x[i:i + 100].contiguous()) for i in range(0, min(B, 1500), 100)])
    assert torch.equal(d_tile[:1500], d_el) and torch.equal(c_tile[:1500], c_el)
    sub = rng.choice(B, 6, replace=False)
    assert np.array_equal(d_tile[sub].cpu().numpy(), onp.maxcut_delta_all(xs[sub], graph, n, graph[:, 2] if weighted else None))
    assert np.array_equal(c_tile[sub].cpu().numpy(), onp.maxcut_node_cutdeg(xs[sub], graph, n, bool(bidir)))
    if not weighted:
        # local-search weights ws = stored_deg - mult * cutdeg through the same kernel family
        # (int8 where the degrees fit a byte, any wider type on request) and its whole-batch span max_b - min_b per node,
        # folded in by the same kernel; large batch (bit-sliced kernel), small batches (element-parallel kernel), ragged last tile
        deg = torch.from_numpy(np.bincount(g.csr.eu, minlength=n)).to(DEV)
        for mult in (1, 2):
            want = deg[None, :] - mult * c_tile
            for dt in (None, torch.int8, torch.int16, torch.int32):
                if dt == torch.int8 and g.csr.max_degree * max(1, mult - 1) > 127:
                    with pytest.raises(RuntimeError):
                        ops.maxcut_ls_weights(g, x, mult, dtype=dt)
                    continue
                for rows in (B, 1000, 70, 1):
                    ws, span = ops.maxcut_ls_weights(g, x[:rows].contiguous(), mult, dtype=dt)
                    assert ws.dtype == (dt or ops.ls_weight_dtype(g, mult))
                    assert torch.equal(ws.long(), want[:rows])
                    mn, mx = torch.aminmax(want[:rows], dim=0)
                    assert span.dtype == torch.int32 and torch.equal(span.long(), mx - mn), (mult, dt, rows)


def test_ls_weights_hub_graph_int16_and_span():
    """Degrees of 256 .. 511 (16 counter planes, 16-bit fields): ws leaves as int16, the span still comes with it."""
    n, B = 700, 2048 + 5
    graph = np.array([(0, j, 1) for j in range(1, 401)] + [(j, j + 1, 1) for j in range(1, 699)], dtype=np.int64)
    g = device_graph(graph, n, 0)
    assert ops.ls_weight_dtype(g, 2) == torch.int16
    xs = ops.rand_spins(B, n, 11, DEV)
    c = ops.maxcut_node_cutdeg(g, xs)
    deg = torch.from_numpy(np.bincount(g.csr.eu, minlength=n)).to(DEV)
    for mult in (1, 2):
        ws, span = ops.maxcut_ls_weights(g, xs, mult)
        want = deg[None, :] - mult * c
        mn, mx = torch.aminmax(want, dim=0)
        assert ws.dtype == torch.int16 and torch.equal(ws.long(), want) and torch.equal(span.long(), mx - mn)
    # no envs: the span table is still initialised (max - min of nothing: INT32_MIN - INT32_MAX wraps to 1; callers skip B = 0)
    ws0, _ = ops.maxcut_ls_weights(g, xs[:0].contiguous(), 1)
    assert ws0.shape == (0, n)


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


@pytest.mark.parametrize("n,m,B,kind", [(2000, 19990, 4096 + 70, "gnm"), (10000, 9999, 2048, "gnm"), (1000, 0, 4096 + 13, "ba-hub"),
                                         (516, 2000, 4100, "gnm"), (64, 300, 4096, "gnm")])
def test_k3_k2_row_staged_stores_equal_the_per_group_form(n, m, B, kind):
    """Round 6: K3 / K2 of a full 64-env tile through wave-private row staging (two groups' counts parked as bytes, two envs' 128
    nodes per store: csrc/rls_maxcut.hip) -- forced at every size through RLS_NS_ROWS = 1 on the 64-env kernel (RLS_NS_TILE32 = 0)
    and compared with the per-group stores (RLS_NS_ROWS = 0), the launcher's own choice and the oracle: ragged last tile (the
    per-group form takes it), a last group of fewer than 64 nodes, an odd number of groups, hub groups (written by the cooperative
    pass, skipped by the row stores)."""
    from rlsolver_amd import _abi
    from rlsolver_amd.graph import generate_ba
    rng = np.random.RandomState(n + B)
    if kind == "ba-hub":
        graph = np.asarray(generate_ba(n, 4, seed=2), dtype=np.int64)
        star = np.stack([np.zeros(400, dtype=np.int64), np.arange(500, 900), np.ones(400, dtype=np.int64)], axis=1)   # a degree-400+ hub
        graph = np.unique(np.concatenate([graph, star]), axis=0)
        graph = graph[graph[:, 0] != graph[:, 1]]
    else:
        graph = gnm_arr(n, m, seed=n)
    g = device_graph(graph, n, 0)
    xs = rng.randint(0, 2, size=(B, n)).astype(np.uint8)
    x = to_dev_bool(xs)
    got = {}
    try:
        for label, t32, rows in (("auto", None, None), ("groups", 0, 0), ("rows", 0, 1), ("half groups", 1, 0), ("half rows", 1, 1)):
            for k, v in (("RLS_NS_TILE32", t32), ("RLS_NS_ROWS", rows)):
                _abi.tuning_unset(k) if v is None else _abi.tuning_set(k, v)
            got[label] = (ops.maxcut_delta_all(g, x), ops.maxcut_node_cutdeg(g, x))
    finally:
        _abi.tuning_unset("RLS_NS_TILE32")
        _abi.tuning_unset("RLS_NS_ROWS")
    for label in ("groups", "rows", "half groups", "half rows"):
        assert torch.equal(got[label][0], got["auto"][0]) and torch.equal(got[label][1], got["auto"][1]), label
    sub = np.concatenate([rng.choice(B, 5, replace=False), [B - 1]])
    assert np.array_equal(got["rows"][0][sub].cpu().numpy(), onp.maxcut_delta_all(xs[sub], graph, n, None))
    assert np.array_equal(got["rows"][1][sub].cpu().numpy(), onp.maxcut_node_cutdeg(xs[sub], graph, n, False))

"""Known-answer tests on the Gset solution strings the reference ships: the seven ``X_G*``
base-64 strings of rlsolver/methods/util_evaluator.py:258-289 with the cut values claimed beside them (fixture
tests/golden/encoder_base64.npz: ``kat/G<k>/{str, num_nodes, claimed_cut, x}``; consumer envs/env_MCPG.py:360-394).

* With the real file ``data/gset/gset_<k>.txt`` present (the reference ships only a 14-node stub; the files are not
  redistributable here) the decoded solution goes through K1 -- ``EnvMaxcut.calculate_obj_values`` -- and must give the
  claimed cut.  Absent file: SKIPPED with that reason, never passed.
* G49 and G50 need no file: they are Gset's 4-regular toroidal grids (3000 nodes, 6000 edges), and the reference's strings
  themselves show the shape -- the alternating pattern flips phase every 100 (G49: 30 x 100) resp. 120 (G50: 25 x 120)
  nodes.  On the torus with row-major node numbering the strings give exactly the claimed 6000 (both sides even: bipartite,
  every edge cut) and 5880 (25 rows: the wrap-around row pair shares its phase, 120 vertical edges uncut): decoder, graph
  pipeline and K1 / K2 / K3 / the greedy sweep checked against numbers that come from the reference's file."""
import os

import numpy as np
import pytest
import torch

from gpu_util import DEV
from rlsolver_amd.graph import GSET_SIZES, read_graph_header, read_mygraph
from rlsolver_amd.methods.util_evaluator import EncoderBase64

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KATS = (14, 15, 22, 49, 50, 55, 70)


def _decoded(golden, k):
    z = golden("encoder_base64")
    n = int(z[f"kat/G{k}/num_nodes"])
    x = EncoderBase64(encode_len=n).str_to_bool(str(z[f"kat/G{k}/str"]))
    assert np.array_equal(x.numpy().astype(np.uint8), z[f"kat/G{k}/x"])
    return n, x.to(DEV), int(z[f"kat/G{k}/claimed_cut"])


@pytest.mark.parametrize("k", KATS)
def test_real_gset_file_known_answer(golden, k):
    path = os.path.join(ROOT, "data", "gset", f"gset_{k}.txt")
    if not os.path.exists(path):
        pytest.skip(f"data/gset/gset_{k}.txt not supplied (the reference ships no Gset file beyond a 14-node stub)")
    n, x, claimed = _decoded(golden, k)
    if read_graph_header(path) != GSET_SIZES[k]:
        pytest.skip(f"data/gset/gset_{k}.txt is not Gset G{k} (header {read_graph_header(path)}, expected {GSET_SIZES[k]})")
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    for bidir in (False, True):
        env = EnvMaxcut(mygraph=read_mygraph(path), device=DEV, if_bidirectional=bidir, num_nodes=n)
        xs = x[None, :].repeat(65, 1).contiguous()                      # one full tile + a ragged one
        got = env.calculate_obj_values(xs)
        assert got.dtype == torch.int64 and bool((got == claimed).all()), f"G{k}: K1 gives {int(got[0])}, the reference claims {claimed}"
        assert bool((env.calculate_obj_values_for_loop(xs).long() == claimed).all())


def _torus(rows, cols):
    idx = np.arange(rows * cols).reshape(rows, cols)
    right = np.stack([idx, np.roll(idx, -1, axis=1)], axis=-1).reshape(-1, 2)
    down = np.stack([idx, np.roll(idx, -1, axis=0)], axis=-1).reshape(-1, 2)
    e = np.concatenate([right, down])
    return [(int(min(a, b)), int(max(a, b)), 1) for a, b in e]


@pytest.mark.parametrize("k,rows,cols", [(49, 30, 100), (50, 25, 120)])
def test_toroidal_gset_strings_known_answer(golden, k, rows, cols):
    from rlsolver_amd import ops
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    n, x, claimed = _decoded(golden, k)
    assert (n, 2 * n) == GSET_SIZES[k] == (rows * cols, 2 * rows * cols)
    g = _torus(rows, cols)
    assert len(set((a, b) for a, b, _ in g)) == 2 * n
    for bidir in (False, True):
        env = EnvMaxcut(mygraph=g, device=DEV, if_bidirectional=bidir, num_nodes=n)
        xs = x[None, :].repeat(130, 1).contiguous()
        xs[1::2] = ~xs[1::2]                                              # the complement cuts the same edges
        got = env.calculate_obj_values(xs)
        assert bool((got == claimed).all()), f"G{k}: K1 gives {int(got[0])}, the reference claims {claimed}"
        deg_cut = ops.maxcut_node_cutdeg(env.graph, xs)                   # K2: per-node cut degree, each cut edge seen by its stored end(s)
        assert int(deg_cut[0].sum()) == claimed * (2 if bidir else 1)
        if not bidir:
            delta = ops.maxcut_delta_all(env.graph, xs)                   # K3: no single flip improves a claimed optimum
            assert int(delta.max()) <= 0
            vs = got.clone()
            ops.maxcut_greedy_sweep(env.graph, xs, vs)                    # K5 leaves it alone (ties are accepted: cut unchanged)
            assert bool((vs == claimed).all()) and bool((env.calculate_obj_values(xs) == claimed).all())

"""K1, K2, K3, K5, K6 and the local-search weights against the oracle over shapes on both sides of every tile choice, in a process of its own: the tile form is picked per
launch (64-env tiles, half tiles of 32 envs, one env per wave), and the dev knobs RLS_K1_TILE32 / RLS_K5_TILE32 / RLS_K6_TILE32 / RLS_NS_TILE32 -- read once per
process -- force one form at every size.  tests/test_gpu_tile32.py runs this file once per setting."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_c as oc
from oracle import oracle_np as onp
from rlsolver_amd import graph as G, ops
from rlsolver_amd.graph import build_csr
from rlsolver_amd.ops_mcpg_tsp import PackedChains
from rlsolver_amd import _abi; _abi.tuning_from_env()   # RLS_<KNOB> variables -> rls_tuning_set (the library itself reads no environment)

DEV = torch.device("cuda:0")
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
# (nodes, edges, envs): rows of 16-byte multiples and not (the ballot path), one ragged half tile, one and many 128-node chunks,
# sizes past the 64-env tile (20 224) up to the largest half tile (40 448), and one past every tile (the one-env-per-wave forms)
SHAPES = [(64, 200, 31), (130, 500, 97), (2000, 19990, 300), (2048, 8000, 64), (1999, 6000, 70), (800, 4694, 33), (3000, 9000, 129),
          (128, 300, 1), (16, 40, 65), (20240, 30000, 40), (24000, 60000, 100), (30001, 50000, 37), (40448, 60000, 33), (41000, 50000, 5)]
n_k1 = n_k5 = n_k6 = n_ns = 0
for n, m, B in SHAPES:
    g = np.asarray(G.generate_gnm(n, m, int(rng.randint(1 << 30))), dtype=np.int64)
    for bidir in (False, True):
        dg = ops.DeviceGraph(build_csr(g, num_nodes=n, if_bidirectional=bidir), DEV)
        x0 = rng.randint(0, 2, size=(B, n)).astype(bool)
        xd = torch.from_numpy(x0).to(DEV)
        want = onp.maxcut_obj(x0, g, bidir)
        assert np.array_equal(ops.maxcut_obj(dg, xd).cpu().numpy(), want), ("K1 u8", n, m, B, bidir)
        assert np.array_equal(ops.maxcut_obj(dg, xd.float()).cpu().numpy(), want), ("K1 f32", n, m, B, bidir)
        # a view inside a buffer of 0xFF bytes: nothing leaks in
        buf = torch.full((B * n + 64,), 255, dtype=torch.uint8, device=DEV)
        view = buf[16:16 + B * n].view(B, n)
        view.copy_(xd.to(torch.uint8))
        assert np.array_equal(ops.maxcut_obj(dg, view.view(torch.bool)).cpu().numpy(), want), ("K1 view", n, m, B, bidir)
        n_k1 += 3
        # K2 / K3 / the local-search weights (one bit-sliced kernel, three outputs): a batch large enough to take the tile form, a
        # ragged last tile, a hub past 256 neighbours at n = 3000
        Bn = max(B, 4000 // max(1, m // n) + 70) if n <= 3000 else B
        gh = g if n != 3000 else np.concatenate([g, np.asarray([(7, j, 1) for j in range(8, 700) if not ((g[:, 0] == 7) & (g[:, 1] == j)).any()
                                                               and not ((g[:, 1] == 7) & (g[:, 0] == j)).any()], dtype=np.int64)])
        dgh = dg if n != 3000 else ops.DeviceGraph(build_csr(gh, num_nodes=n, if_bidirectional=bidir), DEV)
        xn = rng.randint(0, 2, size=(Bn, n)).astype(bool)
        xnd = torch.from_numpy(xn).to(DEV)
        rows = np.unique(np.concatenate([np.arange(min(Bn, 3)), np.arange(max(0, Bn - 35), Bn), rng.randint(0, Bn, 4)]))
        su, sv = onp.stored_edges(gh, bidir)

        def cutdeg(xr):      # values[:, n0] = sum over n0's STORED neighbours n1 of x[n0] ^ x[n1]  (env_L2A.py:68-76), by scatter-add
            out = np.zeros((xr.shape[0], n), np.int64)
            np.add.at(out.T, su, (xr[:, su] ^ xr[:, sv]).T.astype(np.int64))
            return out

        def delta(xr):       # delta[b, i] = sum over i's neighbours j of (x_i == x_j ? 1 : -1): every edge seen from both ends
            d = np.where(xr[:, gh[:, 0]] == xr[:, gh[:, 1]], 1, -1).T.astype(np.int64)
            out = np.zeros((xr.shape[0], n), np.int64)
            np.add.at(out.T, gh[:, 0], d)
            np.add.at(out.T, gh[:, 1], d)
            return out
        if n <= 130:         # the two restatements above against the literal oracle (slow: flips every node of every env)
            assert np.array_equal(cutdeg(xn[rows]), onp.maxcut_node_cutdeg(xn[rows], gh, n, bidir))
            assert np.array_equal(delta(xn[rows]), onp.maxcut_delta_all(xn[rows], gh, n))
        cd = cutdeg(xn[rows])
        rd = torch.from_numpy(rows).to(DEV)
        assert np.array_equal(ops.maxcut_node_cutdeg(dgh, xnd)[rd].cpu().numpy(), cd), ("K2", n, m, Bn, bidir)
        if not bidir:
            assert np.array_equal(ops.maxcut_delta_all(dgh, xnd)[rd].cpu().numpy(), delta(xn[rows])), ("K3", n, m, Bn)
        ws, mm = ops.maxcut_ls_weights(dgh, xnd, 2, return_minmax=True)
        sdeg = np.bincount(su, minlength=n)
        assert np.array_equal(ws[rd].cpu().numpy().astype(np.int64), sdeg[None, :] - 2 * cd), ("ws", n, m, Bn, bidir)
        if Bn * n <= 40_000_000:
            full = sdeg[None, :] - 2 * cutdeg(xn)
            assert np.array_equal(mm.cpu().numpy(), np.stack([full.min(0), full.max(0)])), ("ws min / max", n, m, Bn, bidir)
        n_ns += 3
        # K5: the greedy sweep (level schedule) against the sequential C oracle, hubs included (a star past 256 neighbours)
        if not bidir:
            gs = g if n != 3000 else np.concatenate([g, np.asarray([(7, j, 1) for j in range(8, 700) if not ((g[:, 0] == 7) & (g[:, 1] == j)).any()
                                                                   and not ((g[:, 1] == 7) & (g[:, 0] == j)).any()], dtype=np.int64)])
            dgs = dg if n != 3000 else ops.DeviceGraph(build_csr(gs, num_nodes=n, if_bidirectional=False), DEV)
            k5 = B if n <= 10000 else min(B, 37)                 # (the sequential C oracle is the slow side past 10^4 nodes)
            xs = xd[:k5].clone()
            vs = ops.maxcut_obj(dgs, xs)
            want_x, want_v = oc.greedy_sweep(x0[:k5].astype(np.uint8).copy(), vs.cpu().numpy().copy(), gs[:, 0], gs[:, 1], False)
            ops.maxcut_greedy_sweep(dgs, xs, vs)
            assert np.array_equal(vs.cpu().numpy(), want_v) and np.array_equal(xs.cpu().numpy().astype(np.uint8), want_x), ("K5", n, m, B)
            n_k5 += 1
        # K6: byte mask and bit-packed mask
        mask = rng.rand(B, n) < 0.03
        mask[0] = False
        x1 = x0 ^ mask
        v1 = onp.maxcut_obj(x1, g, bidir)
        want_x, want_v = x0.copy(), want.copy()
        onp.update_xs_by_vs(want_x, want_v, x1, v1, True)
        for packed in (False, True):
            xs = xd.clone()
            vs = torch.from_numpy(want).to(DEV)
            md = torch.from_numpy(mask).to(DEV)
            if packed:
                if n > 40448:
                    continue                     # past every tile the mask must be bytes
                md = PackedChains.pack(md.t().contiguous()).words
            ops.maxcut_propose_accept(dg, xs, md, vs)
            assert np.array_equal(xs.cpu().numpy(), want_x) and np.array_equal(vs.cpu().numpy(), want_v), ("K6", packed, n, m, B, bidir)
            n_k6 += 1
print(f"tile32_child: {n_k1} K1, {n_k5} K5, {n_k6} K6 and {n_ns} K2 / K3 / weights calls match the oracle "
      f"(RLS_K1_TILE32={os.environ.get('RLS_K1_TILE32', '-')} RLS_K5_TILE32={os.environ.get('RLS_K5_TILE32', '-')} "
      f"RLS_K6_TILE32={os.environ.get('RLS_K6_TILE32', '-')} RLS_NS_TILE32={os.environ.get('RLS_NS_TILE32', '-')})")

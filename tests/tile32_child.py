"""K1 and K6 against the oracle over shapes on both sides of every tile choice, in a process of its own: the tile form is picked per
launch (64-env tiles, half tiles of 32 envs, one env per wave), and the dev knobs RLS_K1_TILE32 / RLS_K6_TILE32 -- read once per
process -- force one form at every size.  tests/test_gpu_tile32.py runs this file once per setting."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as onp
from rlsolver_amd import graph as G, ops
from rlsolver_amd.graph import build_csr
from rlsolver_amd.ops_mcpg_tsp import PackedChains

DEV = torch.device("cuda:0")
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
# (nodes, edges, envs): rows of 16-byte multiples and not (the ballot path), one ragged half tile, one and many 128-node chunks,
# sizes past the 64-env tile (20 224) up to the largest half tile (40 448), and one past every tile (the one-env-per-wave forms)
SHAPES = [(64, 200, 31), (130, 500, 97), (2000, 19990, 300), (2048, 8000, 64), (1999, 6000, 70), (800, 4694, 33), (3000, 9000, 129),
          (128, 300, 1), (16, 40, 65), (20240, 30000, 40), (24000, 60000, 100), (30001, 50000, 37), (40448, 60000, 33), (41000, 50000, 5)]
n_k1 = n_k6 = 0
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
print(f"tile32_child: {n_k1} K1 and {n_k6} K6 calls match the oracle "
      f"(RLS_K1_TILE32={os.environ.get('RLS_K1_TILE32', '-')} RLS_K6_TILE32={os.environ.get('RLS_K6_TILE32', '-')})")

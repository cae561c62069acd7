"""The half-tile forms of K1 / K2 / K3 / K5 / K6 / the local-search weights (csrc/rls_tile32.h: 32 envs per workgroup, 32-bit words -- graphs past the 64-env tile up to
40 448 nodes, and for K1 wherever they measure faster) against the oracle, with the launcher's own choice and with each form forced
at EVERY size through the dev knobs (a fresh process per setting: the knobs are read once).  tests/tile32_child.py is the workload."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.parametrize("knob", ["auto", "0", "1", "narrow16", "narrow8"])
def test_k1_k6_tile_forms_match_oracle(knob):
    """(round 5: "narrow16" / "narrow8" force the 16- / 8-env narrow tiles of K1 / K6 / K5 at every size that fits them --
    RLS_NARROW_TILE = 2 / 3 -- where the launchers otherwise take them for small batches and past the half tile only)"""
    env = dict(os.environ)
    for k in ("RLS_K1_TILE32", "RLS_K5_TILE32", "RLS_K6_TILE32", "RLS_NS_TILE32", "RLS_NARROW_TILE"):
        env.pop(k, None)
        if knob in ("0", "1"):
            env[k] = knob
    if knob.startswith("narrow"):
        env["RLS_NARROW_TILE"] = "2" if knob == "narrow16" else "3"
    r = subprocess.run([sys.executable, os.path.join(HERE, "tile32_child.py"), "7"], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-4000:]
    last = r.stdout.strip().splitlines()[-1]
    assert re.fullmatch(r"tile32_child: \d+ K1, \d+ K5, \d+ K6 and \d+ K2 / K3 / weights calls match the oracle \(.*\)", last), r.stdout[-2000:]

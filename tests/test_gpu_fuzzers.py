"""A seeded, time-bounded slice of every differential fuzzer under tools/fuzz/ as part of `-m gpu`.

The fuzzers are the tools that found the defects no unit test saw (an out-of-bounds tie-coin read, an early return in
the best-merge): random graph shapes (G(n, m), stars and multi-hub graphs, paths, near-complete), ragged and full tiles,
both adjacency forms and batches on both sides of every dispatch threshold, HIP path against the C / numpy oracles, bit
for bit.  Each runs here for RLS_FUZZ_SECONDS (default 5) from a fixed seed -- the configuration sequence is
deterministic, only its length depends on the box; `python tools/fuzz/fuzz_<name>.py 600 <seed>` is the long form."""
import glob
import os
import re
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUZZERS = sorted(os.path.basename(p)[5:-3] for p in glob.glob(os.path.join(ROOT, "tools", "fuzz", "fuzz_*.py")))
SECONDS = os.environ.get("RLS_FUZZ_SECONDS", "5")


def test_all_fuzzers_are_listed():
    assert set(FUZZERS) >= {"gym", "isco", "isco_tsp", "ls", "maxcut", "mcpg", "mcpg_round", "qubo", "rand", "select", "spin", "tsp"}


@pytest.mark.parametrize("name", FUZZERS)
def test_fuzzer_slice(name, capsys, monkeypatch):
    path = os.path.join(ROOT, "tools", "fuzz", f"fuzz_{name}.py")
    monkeypatch.chdir(ROOT)                                   # the scripts put "." on sys.path
    monkeypatch.setattr(sys, "argv", [path, SECONDS, "20261003"])
    runpy.run_path(path, run_name="__main__")                 # an AssertionError names the failing configuration
    out = capsys.readouterr().out
    # the fuzzer's LAST line is its verdict: it ran to the end of its time slice, over at least one configuration.  (A header
    # line that merely mentions "configurations" used to satisfy this test: a fuzzer that stopped early would have passed.)
    last = out.strip().splitlines()[-1] if out.strip() else ""
    m = re.fullmatch(rf"fuzz_{name}: (\d+) random configurations(?: \(.*\))?, no (?:mismatch|violation)", last)
    assert m and int(m.group(1)) >= 1, f"no final verdict line from fuzz_{name}: {out[-500:]!r}"

"""Host-side plumbing that needs no GPU: the tuning table of the C ABI (ABI v11: the production library reads no environment
variable), content-hash staleness of the build, and the seed stream of a re-sharded object."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tuning_table_set_get_unset():
    from rlsolver_amd import _abi
    names = _abi.tuning_names()
    assert "RLS_K1_TILE32" in names and "RLS_STEP_CHASE" in names and len(names) == len(set(names)) >= 25
    _abi.tuning_unset()
    assert all(_abi.tuning_get(n) is None for n in names)
    _abi.tuning_set("RLS_K1_TILE32", 1)
    _abi.tuning_set("K7_WAVES", 16)                      # the prefix is optional
    assert _abi.tuning_get("RLS_K1_TILE32") == 1 and _abi.tuning_get("RLS_K7_WAVES") == 16
    _abi.tuning_set("RLS_K1_TILE32", 0)                  # 0 is a value (forces the other form), not "unset"
    assert _abi.tuning_get("RLS_K1_TILE32") == 0
    _abi.tuning_unset("RLS_K1_TILE32")
    assert _abi.tuning_get("RLS_K1_TILE32") is None and _abi.tuning_get("RLS_K7_WAVES") == 16
    with pytest.raises(_abi.RlsError):
        _abi.tuning_set("RLS_NO_SUCH_KNOB", 1)
    assert _abi.tuning_from_env({"RLS_NS_TILE32": "1", "RLS_OUT": "x"}) == {"RLS_NS_TILE32": 1}
    _abi.tuning_unset()
    assert all(_abi.tuning_get(n) is None for n in names)


def test_production_library_ignores_the_environment():
    """A knob exported in the environment reaches the table only in a -DRLS_DEV build; the shipped one starts empty."""
    code = ("from rlsolver_amd import _abi; import sys; "
            "sys.exit(0 if all(_abi.tuning_get(n) is None for n in _abi.tuning_names()) else 1)")
    env = dict(os.environ, RLS_K1_TILE32="1", RLS_STEP_NTS="0", RLS_K7_WAVES="16")
    assert subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env).returncode == 0


def test_no_getenv_outside_the_dev_block():
    """Every getenv of csrc/ sits between `#ifdef RLS_DEV` and its `#endif` (31 knobs were compiled into the
    production library)."""
    for fn in sorted(os.listdir(os.path.join(ROOT, "rlsolver_amd", "csrc"))):
        p = os.path.join(ROOT, "rlsolver_amd", "csrc", fn)
        if not os.path.isfile(p):
            continue
        depth_dev = 0
        for ln in open(p):
            if re.match(r"\s*#\s*ifdef\s+RLS_DEV\b", ln):
                depth_dev += 1
            elif re.match(r"\s*#\s*endif", ln) and depth_dev:
                depth_dev -= 1
            code = ln.split("//")[0]
            assert "getenv" not in code or depth_dev, f"{fn}: {ln.strip()}"


def test_staleness_is_content_not_mtime(tmp_path):
    from rlsolver_amd import build
    a, b = tmp_path / "a.h", tmp_path / "b.hip"
    a.write_text("x")
    b.write_text("y")
    d0 = build._digest([str(a), str(b)], ["-O3"])
    os.utime(a, (1, 1))                                   # an old mtime, a new mtime: the digest does not care
    os.utime(b, None)
    assert build._digest([str(b), str(a)], ["-O3"]) == d0
    assert build._digest([str(a), str(b)], ["-O3", "-DRLS_DEV"]) != d0
    a.write_text("x ")
    assert build._digest([str(a), str(b)], ["-O3"]) != d0
    # the shipped library matches its sources (the round's build() ran), whatever the mtimes say
    build.build()
    for d in build._deps():
        os.utime(d, None)                                 # "a transport that rewrites mtimes"
    assert not build.is_stale()


def test_reshard_keeps_the_seed_stream():
    """set_shard() rebuilt the SeedStream with calls = 0, so an object with a private seed that had already drawn
    handed out the same kernel seeds again after a re-shard."""
    from rlsolver_amd.seeding import KEEP, Sharded

    class Obj(Sharded):
        pass
    o = Obj()
    o._init_shard(0, seed=1234)
    s1, s2 = o._next_seed(), o._next_seed()
    o.set_shard(4096)
    s3 = o._next_seed()
    assert o.env_offset == 4096 and len({s1, s2, s3}) == 3
    fresh = Obj()
    fresh._init_shard(0, seed=1234)
    assert [fresh._next_seed() for _ in range(3)] == [s1, s2, s3]          # the stream went on where it was
    o.set_shard(0, seed=1234)                                              # the SAME seed: still no restart
    assert o._next_seed() not in (s1, s2, s3)
    o.set_shard(0, seed=99)                                                # a new seed: a new stream
    assert o._seeds.calls == 0 and o._seeds.seed == 99
    o.group = "g"
    o.set_shard(8)
    assert o.group == "g"
    o.set_shard(8, seed=None, group=None)                                  # None is a value: back to torch's generator, no group
    assert o.group is None and o._seeds.seed is None
    assert repr(KEEP) == "KEEP"


def test_isco_scratch_query_is_host_logic():
    """rls_isco_maxcut_scratch_bytes (round 5): 0 while a sample's rows fit LDS (N <= ~15 900), 8 N bytes per sample past that; a
    forced knob asks for the scratch at any size.  Pure host logic: runs without a GPU."""
    import ctypes as C
    from rlsolver_amd import _abi
    lib = _abi.lib()
    g = _abi.RlsGraph()
    for n, want in ((2000, 0), (10000, 0), (15000, 0), (20000, 8 * 20000), (80000, 8 * 80000)):
        g.num_nodes = n
        assert lib.rls_isco_maxcut_scratch_bytes(C.byref(g), 1) == want
        assert lib.rls_isco_maxcut_scratch_bytes(C.byref(g), 37) == 37 * want
    g.num_nodes = 2000
    assert lib.rls_isco_maxcut_scratch_bytes(C.byref(g), 0) == 0 and lib.rls_isco_maxcut_scratch_bytes(None, 5) == 0
    _abi.tuning_set("RLS_ISCO_GLOBAL_ROWS", 1)
    try:
        assert lib.rls_isco_maxcut_scratch_bytes(C.byref(g), 3) == 3 * 8 * 2000
    finally:
        _abi.tuning_unset("RLS_ISCO_GLOBAL_ROWS")
    assert lib.rls_isco_maxcut_scratch_bytes(C.byref(g), 3) == 0

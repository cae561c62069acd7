"""Host-side kernel choices that need no GPU."""
from rlsolver_amd.methods.MCPG_qubo import qubo_prefers_sparse


def test_qubo_kernel_choice_follows_the_measured_crossovers():
    # (n, fill, chains) -> CSR kernel wins; measured on MI355X (tools/sweeps/time_qubo_sparse.py, round 6: the CSR kernel sweeps by
    # levels -- n = 1000 at 2 % fill now wins at 2^13 chains too, 410 vs 883 us; 10 % fill stays dense at every chain count)
    measured = [
        (1000, 0.005, 1 << 13, True), (1000, 0.02, 1 << 13, True), (1000, 0.1, 1 << 13, False),
        (1000, 0.005, 1 << 15, True), (1000, 0.02, 1 << 15, True), (1000, 0.1, 1 << 15, False),
        (2000, 0.005, 1 << 13, True), (2000, 0.02, 1 << 13, True), (2000, 0.1, 1 << 13, False),
        (2000, 0.005, 1 << 15, True), (2000, 0.02, 1 << 15, True), (2000, 0.1, 1 << 15, False),
        (1000, 0.02, 1 << 16, True), (1000, 0.1, 1 << 16, False), (2000, 0.25, 1 << 16, False),
    ]
    for n, fill, C, want in measured:
        assert qubo_prefers_sparse(n, int(n * n * fill), C) == want, (n, fill, C)
    assert not qubo_prefers_sparse(64, 64 * 64, 1 << 20)      # a full matrix never goes sparse
    assert not qubo_prefers_sparse(0, 0, 10)


def test_bench_profiled_reader_finds_the_committed_rows():
    """bench.py's _profiled(): the SQ-counter figures of the config rows come from the newest committed per-kernel table
    (profiles/rNN_kernels.json, a flat list of row dicts since r02) -- ADVICE r5: the reader looked for a 'rows' key and
    silently found nothing."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    k7 = bench._profiled("k_mcpg_local_search_levels", "BA-1e4", "valu_frac")
    assert k7 is not None and 0 < k7["value"] < 1 and k7["source"].startswith("profiles/r")
    ls = bench._profiled("k_maxcut_local_search", "G22", "valu_frac")
    assert ls is not None and 0 < ls["value"] < 1
    assert bench._profiled("no_such_kernel", "G22", "valu_frac") is None
    assert bench.metric_for(22) == bench.METRIC and "G70" in bench.metric_for(70)

import os
import sys

os.environ.setdefault("RLS_RECORD_OPS", "1")     # rlsolver_amd.torch_ops records which ops ran (test_gpu_zz_op_coverage.py)

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # the op-coverage check looks back at everything the session ran: keep it last
    items.sort(key=lambda it: "test_gpu_zz_op_coverage" in it.nodeid)
    config._rls_gpu_files_collected = {it.nodeid.split("::")[0] for it in items if "gpu" in it.keywords}
    # gpu-marked tests are skipped (not failed) where no GPU is visible, so `pytest tests/`
    # without -m still works in the CPU container
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)

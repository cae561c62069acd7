"""Runs LAST in a -m gpu session (tests/conftest.py sorts it there): every device entry point of include/rlsolver_hip.h
must have been executed, through its torch.ops.rlsolver_hip op, by the parity tests that ran before it.  Only meaningful
for a whole-suite run; a subset (-k, one file) skips."""
import glob
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_device_entry_point_was_executed(request):
    from rlsolver_amd import torch_ops
    on_disk = {os.path.relpath(p, ROOT) for p in glob.glob(os.path.join(ROOT, "tests", "test_gpu*.py"))}
    collected = getattr(request.config, "_rls_gpu_files_collected", set())
    if not on_disk <= collected or request.config.getoption("-k"):
        pytest.skip("op coverage is checked on whole-suite runs only")
    rec = torch_ops.ops
    assert hasattr(rec, "called"), "RLS_RECORD_OPS was not set before rlsolver_amd was imported"
    missing = sorted(set(torch_ops.DEVICE_ENTRY_POINTS) - rec.called)
    assert not missing, f"device entry points no GPU test executed: {missing}"

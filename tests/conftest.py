import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # On this image torch cannot initialise HIP once another library in the process has done so ("No HIP GPUs are available"): when GPU
    # tests are selected, bring torch's runtime up before the first of them touches libratilqr_hip.so, whatever the file order.
    if any(it.get_closest_marker("gpu") for it in items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


@pytest.fixture(autouse=True)
def _speculation_forced(monkeypatch):
    """spec_eps is an upper bound: by default a handle runs the sequential line-search rule (E = 1) whatever width it was created with
    (include/ratilqr.h, switch spec_force).  The tests that create handles with spec_eps > 1 are there to hold the SPECULATIVE kernels to
    the oracle, so the suite forces the requested width; the default policy has its own tests (test_gpu_parity.py::test_spec_eps_is_an_upper_bound,
    test_gpu_ce.py::test_config3_full_size_ce_matches_the_oracle), which remove this variable."""
    monkeypatch.setenv("RATILQR_SPEC_FORCE", "1")

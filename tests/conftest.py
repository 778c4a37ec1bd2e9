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

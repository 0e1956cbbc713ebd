import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # EGNN_TEST_POISON=1 python -m pytest tests -m gpu: the WHOLE suite with every uninitialised buffer poisoned -- torch.empty
    # tensors start as NaN (torch's debug fill) and, through EGNN_DEBUG_POISON, so does every device buffer the library allocates
    # itself.  A test that passes normally and fails here reads memory before anything wrote it.
    if os.environ.get("EGNN_TEST_POISON", "0") == "1":
        import torch
        os.environ["EGNN_DEBUG_POISON"] = "1"
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)

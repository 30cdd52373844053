import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def conf_var():
    from bvcodec import config
    return config.load_config(config.DEFAULT_CONFIG)


@pytest.fixture(scope="session")
def conf_fix():
    from bvcodec import config
    return config.load_config(config.DEFAULT_CONFIG_64BIT)


@pytest.fixture(autouse=True)
def _recurrence_health(request):
    """After every GPU test: no persistent recurrence kernel may have given up waiting (bvc_model_status)."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import gpu_common
    for model, *_ in list(gpu_common._CACHE.values()):
        model.check_status()

import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture
def knob():
    """knob(name, value): sets a library switch through pdae_set_knob for the duration of the test (the library reads its environment only once,
    at a knob's first use: tests cannot switch routing by editing os.environ)."""
    from pdae_amd import hip
    saved = {}

    def set_(name, value):
        if name not in saved:
            saved[name] = hip.get_knob(name)
        hip.set_knob(name, int(value))
    yield set_
    for n, v in saved.items():
        hip.set_knob(n, v)


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: d[k] for k in d.files}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rel_err(a, b):
    """max |a-b| / max |b| -- the 'relative fp32' measure used by every parity gate."""
    a = a.detach().double().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(a).double()
    b = b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

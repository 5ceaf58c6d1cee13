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
    # the oracle runs on the host cores; a GPU box exposes every core of the node but grants 16: the default thread count then
    # oversubscribes the OpenMP pool by an order of magnitude (a 1 s oracle step took minutes)
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(n, 16)))


def load_golden(name):
    """npz -> nested dict of torch tensors ('a/b' keys become d['a']['b']; integer sub-keys become lists)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        v = torch.from_numpy(z[k])
        if "/" in k:
            a, b = k.split("/", 1)
            out.setdefault(a, {})[b] = v
        else:
            out[k] = v
    for a, d in list(out.items()):
        if isinstance(d, dict) and all(s.isdigit() for s in d):
            out[a] = [d[str(i)] for i in range(len(d))]
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()

import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    torch.set_num_threads(min(8, os.cpu_count() or 1))


def load_golden(name):
    """.npz fixture -> dict of torch tensors (fp16 arrays are widened to fp32)."""
    out = {}
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        for k in z.files:
            a = z[k]
            t = torch.from_numpy(a.astype(np.float32) if a.dtype == np.float16 else a)
            out[k] = t
    return out


@pytest.fixture(scope='session')
def golden():
    return load_golden

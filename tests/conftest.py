import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in 'fiub' and z[k].ndim > 0 else z[k]) for k in z.files}


@pytest.fixture(scope='session')
def golden():
    return load_golden

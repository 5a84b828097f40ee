import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def rel_err(a, b):
    """max|a-b| / max|b|  -- the 'relative fp32' measure used by every parity test."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


class Opts:
    adopted_datasets = ['alpha_tubulin', 'beta_actin', 'desmoplakin', 'dna',
                        'fibrillarin', 'lamin_b1', 'membrane_caax_63x',
                        'myosin_iib', 'sec61_beta', 'st6gal1', 'tom20', 'zo1']
    gpu_ids = -1
    batch_size_eval = 2


@pytest.fixture
def opts():
    return Opts()


def record(test, **values):
    """Append what a tolerance-bound test actually measured to gpurun_out/test_measurements.jsonl (when that directory
    exists: on the GPU box, merged back by gpurun) -- the evidence the tolerances in the test files are set from."""
    import json
    d = os.path.join(ROOT, 'gpurun_out')
    if not os.path.isdir(d):
        return
    with open(os.path.join(d, 'test_measurements.jsonl'), 'a') as f:
        f.write(json.dumps({'test': test, **{k: (float(v) if not isinstance(v, (str, list, dict)) else v) for k, v in values.items()}}) + '\n')

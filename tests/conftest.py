import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_features(g):
    """feature dict (torch) stored by oracle/make_goldens.py under f_* keys"""
    return {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('f_')}


CALL_CASES = ['uncond_n16_b1_t1000', 'uncond_n32_b2_t500', 'ragged_n50_b2_t1', 'twochain_n32_b1_t777',
              'motif_n40_b2_t300']


@pytest.fixture(scope='session')
def base_weights():
    """The synthetic 'everything live' weights the goldens were made with."""
    import hashlib
    from oracle import genie_oracle as O
    sd = O.synthetic_state_dict(O.BASE_DIMS, seed=0)
    g = load_golden('weights_recipe')
    h = hashlib.sha256()
    for k in sd:
        h.update(sd[k].numpy().tobytes())
    if h.digest() != g['sha256'].tobytes():
        pytest.skip('torch RNG stream differs from the one the goldens were generated with')
    return sd


@pytest.fixture(scope='session')
def base_engine(base_weights):
    from oracle import genie_oracle as O
    from genie2_amd.engine import GenieEngine
    eng = GenieEngine(dict(O.BASE_DIMS), base_weights, 'cuda:0')
    eng._test_weights = base_weights
    yield eng
    eng.close()

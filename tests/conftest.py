import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def fp16_kernels_at_test_sizes(monkeypatch):
    """The reduced-precision modes fall back to the fp32 small-batch kernels below 512 residual rows (the fp16-MFMA kernels
    have no small-batch variants and are slower there: mc_model.hip half_min_rows).  The small test configs sit below that
    size, so the tests lift the limit to exercise the fp16 kernels; test_fp16_modes_use_the_fp32_kernels_at_tiny_batches
    checks the default."""
    monkeypatch.setenv('MC_HALF_MIN_ROWS', '0')

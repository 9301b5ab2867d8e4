import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, so a bare
    ``pytest tests`` works in the CPU container; ``-m gpu`` on the box runs them."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def x3_restore():
    """Tests that force a product form put the library's size rule back afterwards."""
    yield
    from prodsearch_amd import _lib
    _lib.load().ps_gemm_x3_config(0 if os.environ.get('PS_GEMM_X3') == '0' else 1, int(os.environ.get('PS_GEMM_X3_SHAPE', '-1')))


@pytest.hookimpl(wrapper=True)
def pytest_runtest_call(item):
    """PS_DETERMINISTIC=1 over the whole suite (tools/env_matrix.sh): the review transformer's deterministic mode REFUSES the
    configurations it does not cover (pv encoder, PV loss, user / item embeddings: DESIGN.md 5e) — those cases are skipped, not
    failed; tests/test_gpu_determinism_rtm.py checks the refusal itself."""
    try:
        return (yield)
    except RuntimeError as e:
        if os.environ.get('PS_DETERMINISTIC') == '1' and 'deterministic mode supports' in str(e):
            pytest.skip("deterministic mode refuses this configuration")
        raise

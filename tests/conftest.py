import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def attn_shape_invariant():
    """f5hip_set_attention_shape_invariant(1) for the duration of a test: every attention variant then adds a query's terms in one order, so
    a sequence's output does not depend on the shape of the launch it is part of (the default picks the fastest kernel per shape, and the
    SIMD-balanced one associates the sums of a third of its query blocks differently)."""
    from tts_indic_server_f5_amd import _lib
    _lib.check(_lib.lib().f5hip_set_attention_shape_invariant(1), "set_attention_shape_invariant")
    yield
    _lib.check(_lib.lib().f5hip_set_attention_shape_invariant(0), "set_attention_shape_invariant")

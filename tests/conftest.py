import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    from gaia_seg_amd.hip import lib
    return lib.load()


def rel_err(a, b):
    """max |a-b| / max(|b|, tiny)  on CPU float64."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def l2_err(a, b):
    """||a-b||_2 / ||b||_2 on CPU float64.  Used for gradients: a ReLU whose pre-activation is
    within rounding of zero can take a different branch in two correct fp32 implementations, which
    changes single gradient elements by O(1) (measured on the MI355X: 1 flip in 80000 elements of
    one head layer moved the max-norm error of every upstream gradient to 1e-2 while every kernel
    matched torch to 3e-7 on the same device data).  The L2 norm is insensitive to such isolated
    flips; forward quantities (loss, logits, features, BN statistics) are still checked in the
    max norm at the 1e-3 tolerance BASELINE.json states."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

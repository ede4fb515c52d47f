import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def usable_cores():
    """CPU cores this process may actually use: the cgroup quota if there is one (a GPU box gives a
    job 16 of its 256 hardware threads), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle's passes dominate the GPU suite's wall time.  PyTorch sizes its thread pool from
    # the affinity mask (all 256 hardware threads of a GPU box), the box's quota is 16 cores: without
    # this the pool is 16x oversubscribed and throttled.
    import torch
    torch.set_num_threads(usable_cores())


@pytest.fixture(autouse=True)
def _default_stream_after_each_test():
    """A runner makes its high-priority training stream the current stream of the thread
    (core/runner.py TRAIN_PRIORITY): every test starts on the default stream again."""
    yield
    import torch
    if torch.cuda.is_available() and torch.cuda.current_stream() != torch.cuda.default_stream():
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream())


@pytest.fixture(scope="session")
def hip_lib():
    from gaia_seg_amd.hip import lib
    return lib.load()


def rel_err(a, b):
    """max |a-b| / max(|b|, tiny)  on CPU float64."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))

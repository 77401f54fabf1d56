import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _collect_handles_between_gpu_tests(request):
    """Library handles (pvr_policy / pvr_encoder: workspaces, streams) of dead Python objects are destroyed HERE, at a quiet point
    between two tests and after a device sync, not whenever the cyclic collector happens to run in the middle of the next test's
    enqueues.  Only for tests that use the GPU."""
    yield
    if request.node.get_closest_marker('gpu') is not None:
        import gc
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.synchronize()

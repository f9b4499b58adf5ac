import os
import sys

os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')      # kernel arguments in device memory: -2 % per train step (read when the HIP runtime loads, i.e. before torch)
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _restore_params():
    """Params is a process-global class; restore defaults after each test."""
    yield
    from multilingual_text_to_speech_amd.params import reset_defaults
    reset_defaults()

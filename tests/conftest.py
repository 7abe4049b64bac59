import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True, scope="session")
def _library_options_follow_the_environment():
    """The C library reads no environment variables (include/speech_amd.h: sa_set_option); speech_amd._lib applies SA_<NAME>
    variables once, at load.  Tests switch kernel paths with monkeypatch.setenv BETWEEN calls, so here the host re-applies the
    environment whenever it changed since the last library access."""
    from speech_amd import _lib
    _lib.ENV_SYNC = True
    yield
    _lib.ENV_SYNC = False

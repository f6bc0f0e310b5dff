import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def deterministic_stats():
    """Trajectory tests run on the fixed-order batch-norm statistics (functional.CONV_EPILOGUE_STATS = False, the
    GANK_EPILOGUE_STATS=0 option: two identical updates then differ by 0.017 % instead of 0.4 %), so a
    statistical bound is not also asked to absorb the arrival order of the conv epilogues' float atomics."""
    from gan_lib_tensorflow_amd import functional as Fn
    were, Fn.CONV_EPILOGUE_STATS = Fn.CONV_EPILOGUE_STATS, False
    try:
        yield
    finally:
        Fn.CONV_EPILOGUE_STATS = were

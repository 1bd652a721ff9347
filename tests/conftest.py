import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _collect_contexts():
    """Contexts are destroyed when their engine is collected; the process-wide packed-f32 guard of lass_finalize (a bf16 context and
    a context routed to wino32.hip must not be alive together) looks at live contexts, so no test may inherit a lingering one."""
    yield
    import gc
    gc.collect()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def synthetic_sd():
    from lass_amd import synthetic
    return synthetic.make_state_dict()

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A GPU test that hangs (a rank of a self-launched bench.py waiting for a collective its peer never issues, say) must
    fail, not sit on the box until the caller's limit: 300 s per test where pytest-timeout is there (the longest takes 20)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if "gpu" in item.keywords and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(300))


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle_lib import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libcurve25519_ref.so not built (needs /root/reference)")
    return Reference()

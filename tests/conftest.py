import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    """the plain-C CPU restatement (test-only checker)"""
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle.Oracle()


@pytest.fixture(scope="session")
def ref():
    """the genuine reference build; only where oracle/_ref/libbtlref.so exists"""
    from oracle import pyoracle

    if not pyoracle.Ref.available():
        pytest.skip("oracle/_ref/libbtlref.so not present (no /root/reference on this machine)")
    return pyoracle.Ref()


@pytest.fixture(scope="session")
def lib():
    """the product C-ABI library (HIP).  Loading it needs no GPU; calling compute entry points does."""
    from btl_bloomfilter_amd import _lib

    return _lib.load()

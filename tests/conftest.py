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


def require_hbm(nbytes, what=""):
    """The full-size tests are the parity evidence at the benchmark's geometries.  On the device they are written for
    (an MI355X: 288 GB) too little FREE memory means a leak, a scratch-footprint regression or somebody else on the
    card -- a failure to look at, not a reason to turn green by absence.  Only a smaller GPU skips."""
    import gc

    import torch

    gc.collect()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if free >= nbytes:
        return
    msg = "%s needs %.0f GiB of free HBM; %.0f of %.0f GiB are free" % (what or "this test", nbytes / 2**30, free / 2**30, total / 2**30)
    if total >= 250 * 2**30:
        pytest.fail(msg)
    pytest.skip(msg)


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

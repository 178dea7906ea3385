"""The C++ drop-in shims (include/btlbf/*.hpp: BloomFilter, CountingBloomFilter<uint8_t>,
KmerBloomFilter, ntHashIterator, stHashIterator, insertSeq) exercised by a C++ program that replays
the reference's unit-test scenarios through them (tests/cpp/test_shims.cpp)."""
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT


def test_shim_headers_compile_on_cpu():
    """CPU: the shim headers compile and link against the C ABI (no GPU needed to build)"""
    import __graft_entry__ as g
    from btl_bloomfilter_amd import build

    build.build()
    exe = g.build_shim_test()
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_reference_unit_scenarios_through_shims():
    import __graft_entry__ as g
    from btl_bloomfilter_amd import build

    build.build()
    exe = g.build_shim_test()
    r = subprocess.run([exe, GOLDEN], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all shim tests passed" in r.stdout

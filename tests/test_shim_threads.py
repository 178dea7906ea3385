"""The reference's threading contract on the drop-in shims (insert / contains / insertAndCheck callable on one
filter from many threads: BloomFilter.hpp:177,191,206-210; Tests/AdHoc/ParallelFilter.cpp:104-122).
CPU: the shim's own locking under ThreadSanitizer against a test-only stub of the C ABI (tests/cpp/stub_abi.cpp).
GPU: the OpenMP replay inside tests/cpp/test_shims.cpp (tests/test_gpu_cpp_shims.py)."""
import os
import subprocess

from conftest import ROOT


def test_shim_locking_is_tsan_clean(tmp_path):
    exe = str(tmp_path / "test_shims_tsan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "test_shims_tsan.cpp"), os.path.join(ROOT, "tests", "cpp", "stub_abi.cpp"),
           "-lpthread", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, r.stdout + r.stderr
    assert "shim threading test passed" in r.stdout

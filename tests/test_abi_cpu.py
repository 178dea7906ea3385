"""CPU-side checks of the product boundary: the C-ABI library builds for gfx950, loads without a
GPU, exports every symbol include/btlbf.h declares, and refuses to compute without a GPU (no
silent CPU fallback).  No compute entry point is exercised here."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "btlbf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(btlbf_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol(lib):
    from btl_bloomfilter_amd import _lib

    names = declared_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), "libbtlbf.so does not export %s" % n
    assert sorted(_lib.EXPORTS) == names, "python binding table and include/btlbf.h disagree"


def test_library_contains_gfx950_code_object():
    from btl_bloomfilter_amd import _lib

    data = open(_lib.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in data


def test_no_cpu_fallback_without_gpu(lib):
    if lib.btlbf_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.btlbf_create(C.byref(h), 0, 1024, 4, 31, 0, 0)
    assert rc == 5  # BTLBF_EHIP
    assert b"no CPU path" in lib.btlbf_last_error()
    rc = lib.btlbf_hash_seqs(31, 4, None, 0, 0, b"ACGT" * 20, 80, None, None, None, None, 0, 0, None)
    assert rc == 5


def test_argument_errors_are_reported_before_touching_the_gpu(lib):
    h = C.c_void_p()
    # BloomFilter.hpp:391-394: a bit count that is not a multiple of 8 is an error
    assert lib.btlbf_create(C.byref(h), 0, 1001, 4, 31, 0, 0) == 1
    assert b"not a multiple of 8" in lib.btlbf_last_error()
    # missing file -> EIO like vendor/IOUtil.h:14-22
    assert lib.btlbf_load(C.byref(h), 0, b"/nonexistent/x.bf", 0, 0) == 3
    # bad magic line -> EFORMAT like BloomFilter.hpp:123-129
    p = os.path.join(ROOT, "tests", "golden", "cbf_1000_k25_h3_insert.bf")
    assert lib.btlbf_load(C.byref(h), 0, p.encode(), 0, 0) == 4
    assert b"magic string does not match" in lib.btlbf_last_error()


def test_product_never_references_the_oracle():
    bad = []
    for base in ("btl_bloomfilter_amd", "include", "tools"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                    t = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"(from|import)\s+oracle|btl_oracle|libbtlref|pyoracle", t):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_read_grid_plan_invariants(lib):
    """pass A's read grid (planning is host code, no GPU needed): whole reads per tile, whole bytes of the window
    bitmaps per tile, every lane inside the workgroup, the tile image inside the LDS, and only where it pays"""
    import ctypes as C

    out = (C.c_uint32 * 4)()
    used = 0
    for k in (5, 21, 25, 31, 32, 33, 47, 64, 96, 150):
        for L in (k, k + 1, 40, 50, 64, 75, 100, 101, 125, 150, 151, 152, 200, 250, 251, 300, 500, 1000, 5000):
            if L < k:
                continue
            for bins in (64, 512, 1024):
                assert lib.btlbf_plan_read_grid(k, 4, L, bins, out) == 0
                reads, gpr, lpad, cap = list(out)
                if reads == 0:
                    assert (gpr, lpad, cap) == (0, 0, 0)
                    continue
                used += 1
                wins = L - k + 1
                assert gpr == (wins + 7) // 8 and lpad == (L + 7) // 8 * 8
                assert reads * gpr <= 1024                      # one lane per group of 8 window starts
                assert (reads * L) % 8 == 0                     # a tile is whole bytes of the per-window bitmaps
                assert reads * L <= 3 * 1024 * 4 - 16           # three staged words per thread
                assert cap >= reads * lpad + k + 8 and cap % 16 == 0 and cap < 40 * 1024
                # busy lane-windows: at least 5 % more than plain tiles of 8192 window starts
                assert reads * wins / 8192.0 >= 1.05 * wins / L
    assert used > 50
    # the bench's geometry: 150-base reads, k = 31 -> 68 reads of 15 groups: 1020 of 1024 lanes carry 8 k-mers
    assert lib.btlbf_plan_read_grid(31, 4, 150, 512, out) == 0 and list(out)[:3] == [68, 15, 152]
    assert lib.btlbf_plan_read_grid(31, 4, 1000, 512, out) == 0 and list(out) == [0, 0, 0, 0]  # long reads: no gain
    assert lib.btlbf_plan_read_grid(31, 4, 150, 4096, out) != 0  # argument check

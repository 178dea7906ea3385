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
def test_reference_unit_scenarios_through_shims(tmp_path):
    import __graft_entry__ as g
    from btl_bloomfilter_amd import build

    build.build()
    exe = g.build_shim_test()
    out = tmp_path / "parallel.bf"
    swig_out = tmp_path / "swig_test_pl.bf"
    r = subprocess.run([exe, GOLDEN, str(out), str(swig_out)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, OMP_NUM_THREADS="8"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all shim tests passed" in r.stdout
    # the filter built by 8 OpenMP threads calling bloom.insert(*itr) on one BloomFilter (the reference's
    # Tests/AdHoc/ParallelFilter.cpp pattern) is the reference's own filter, byte for byte
    import hashlib

    from conftest import load_golden

    g = load_golden("digests.json")["bf_small"]
    data = out.read_bytes()
    body = data[data.index(b"[HeaderEnd]\n") + len(b"[HeaderEnd]\n"):]
    assert len(body) == g["bits"] // 8 and hashlib.sha256(body).hexdigest() == g["body_sha256"]
    # the file swig/test.pl stores (k = 20 raw k-mers through KmerBloomFilter::insert(const char*)) is the reference's
    t = load_golden("kmer_path.json")["swig_test_pl"]
    assert hashlib.sha256(swig_out.read_bytes()).hexdigest() == t["file_sha256"]


@pytest.mark.gpu
def test_example_tool_build_and_query(tmp_path, oracle):
    """examples/bloom_tool: build a filter from a gzipped FASTQ, store it, load it again and query"""
    import gzip

    import numpy as np

    import __graft_entry__ as g

    g.build_shim_test()
    exe = os.path.join(ROOT, "examples", "bloom_tool")
    rng = np.random.RandomState(3)
    reads = ["".join(rng.choice(list("ACGT"), 100)) for _ in range(300)]
    fq = tmp_path / "r.fq.gz"
    with gzip.open(fq, "wt") as fh:
        for i, s in enumerate(reads):
            fh.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
    out = tmp_path / "r.bf"
    bits, h, k = 1 << 20, 3, 21
    r = subprocess.run([exe, "build", str(fq), str(bits), str(h), str(k), str(out)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "records 300" in r.stdout, r.stdout + r.stderr
    mine = np.zeros(bits // 8, np.uint8)
    for s in reads:
        oracle.bf_insert_seq(mine, bits, h, k, s.encode())
    body = open(out, "rb").read()
    assert body.endswith(mine.tobytes()) and body.startswith(b"[BTLBloomFilter_v1]")
    fa = tmp_path / "q.fa"
    fa.write_text(">a\n%s\n>b\n%s\n" % (reads[0], "".join(rng.choice(list("ACGT"), 500))))
    r = subprocess.run([exe, "query", str(out), str(fa)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    n_all = (100 - k + 1) + (500 - k + 1)
    assert ("k-mers %d " % n_all) in r.stdout and ("found %d " % (100 - k + 1)) in r.stdout, r.stdout


@pytest.mark.gpu
def test_example_sharded_rccl_one_gpu():
    """examples/sharded_rccl.cpp: the routed multi-GPU path driven from C++ over RCCL (all visible GPUs;
    one here, so every block travels GPU -> RCCL -> same GPU).  1.1 M reads make a 2.4 GB block, which
    only arrives whole because the example slices its messages; the program checks popcount and hits."""
    import __graft_entry__ as g

    g.build_shim_test()
    exe = os.path.join(ROOT, "examples", "sharded_rccl")
    r = subprocess.run([exe, "33", "1100000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "gpus " in r.stdout and "clean 132000000  hits 132000000" in r.stdout, r.stdout

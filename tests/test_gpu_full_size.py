"""Parity at the geometries bench.py and tools/config_bench.py time (BASELINE.json configs C2, C3, C5 at
FULL filter size): the partitioned pipeline (pass A hash + radix partition, pass B split, pass C apply /
test in LDS) against the direct kernels, which tests/test_gpu_parity.py pins to the reference's golden
vectors and to the CPU oracle.  Two filters of the full size sit side by side in HBM and are compared
there (btlbf_compare: XOR + popcount on the device); query bitmaps are compared on the device too.

What is matched: BloomFilter::insert / contains (BloomFilter.hpp:185-194,252-262),
CountingBloomFilter::incrementAll / contains (CountingBloomFilter.hpp:165-196), the stHashIterator-fed
filter of config 5 (vendor/stHashIterator.hpp:53)."""
import pytest
from conftest import require_hbm

pytestmark = pytest.mark.gpu

L = 150
C5_SEEDS = ["1110111011101110111011101110111", "1101101101101101011011011011011",
            "1111001111001111111001111001111", "1011101011101011101011101011101"]


@pytest.fixture(scope="module")
def bf():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    torch.zeros(1, device="cuda")
    import btl_bloomfilter_amd as m

    assert m._lib.load().btlbf_device_count() > 0
    return m


def need_hbm(nbytes):
    require_hbm(nbytes, "a full-size parity test")


def spliced_query(bf, reads, n, foreign, seed):
    """the inserted reads with `foreign` reads nobody inserted spliced in at regular intervals, one N"""
    import torch

    q = reads.clone()
    other = bf.synth_reads_device(seed, 0, foreign, L)
    idx = torch.arange(foreign, device="cuda") * (n // foreign)
    q.view(n, L)[idx] = other.view(foreign, L)
    q[L * 7 + 40] = ord("N")
    return q


def batches_of(prof, slot):
    return prof.get(slot, (0, 0))[1]


def test_c2_partitioned_pipeline_at_bench_geometry(bf):
    """2^39 bits, k=31, h=4: 128 KiB segments, 512 x 1024 bins, part_apply_kernel<..., 1024> -- the geometry
    behind the headline number.  2 x 10^7 reads under a scratch cap that forces several batches."""
    import torch

    bits, h, k, n = 1 << 39, 4, 31, 20_000_000
    need_hbm(2 * (bits // 8) + (60 << 30))
    reads = bf.synth_reads_device(42, 0, n, L)
    a, b = bf.BloomFilter(bits, h, k), bf.BloomFilter(bits, h, k)
    a.setInsertMode("direct")
    b.setInsertMode("partitioned", scratch_bytes=24 << 30)
    b.setProfiling(True)
    a.insertSeqs(reads, read_len=L)
    b.insertSeqs(reads, read_len=L)
    torch.cuda.synchronize()
    prof = b.getProfile()
    assert batches_of(prof, "insert_hash") >= 2 and batches_of(prof, "insert_apply") >= 16, prof
    assert "insert_direct" not in prof
    assert b.compare(a) == (0, 0, 0), "partitioned insert differs from the direct kernel at 2^39 bits"
    pop = a.getPop()
    assert pop == b.getPop() and 0.99 * n * 120 * h < pop <= n * 120 * h
    # inserting again changes nothing (idempotence) -- through the partitioned path on the direct filter too
    a.setInsertMode("partitioned", scratch_bytes=24 << 30)
    a.insertSeqs(reads[: 5_000_000 * L], read_len=L)
    assert a.compare(b) == (0, 0, 0)

    # query: ~0.1 % foreign reads spliced in; bitmaps and counts of the two modes must be identical
    q = spliced_query(bf, reads, n, 20_000, 43)
    out = {}
    for mode in ("direct", "partitioned"):
        b.setQueryMode(mode)
        b.getProfile()
        hit, valid, cnt = b.containsSeqs(q, read_len=L, want_valid=True, want_counts=True)
        torch.cuda.synchronize()
        out[mode] = (hit, valid, cnt.tolist(), b.getProfile())
    assert batches_of(out["partitioned"][3], "query_hash") >= 2 and batches_of(out["partitioned"][3], "query_test") >= 16
    assert "query_direct" not in out["partitioned"][3] or batches_of(out["partitioned"][3], "query_direct") == 0
    assert out["direct"][2] == out["partitioned"][2]
    assert bool(torch.equal(out["direct"][1], out["partitioned"][1])), "valid bitmaps differ"
    assert bool(torch.equal(out["direct"][0], out["partitioned"][0])), "hit bitmaps differ"
    clean, hits = out["direct"][2]
    assert clean == n * 120 - 31
    assert n * 120 - 31 - 20_000 * 120 <= hits < clean - 19_000 * 120  # foreign reads miss (a few false positives)
    # all hits: nothing may fail
    b.setQueryMode("partitioned")
    _, _, cnt = b.containsSeqs(reads, read_len=L, want_valid=False, want_counts=True)
    assert cnt.tolist() == [n * 120, n * 120]


def test_c2_geometry_skewed_reads_overflow_rings_late_image_and_regions(bf):
    """The same 512 x 1024-bin geometry fed reads that hammer a few bins: long runs of one base (every window the same
    k-mer: four positions), blocks of copies of one read, a low-complexity repeat.  Pass A's rings overflow, their late
    image fills up and overflows too, regions run over their capacity -- all of that has to land in the array exactly as
    the direct kernel puts it (insert) and to be answered like it (query)."""
    import torch

    bits, h, k, n = 1 << 39, 4, 31, 6_000_000
    need_hbm(2 * (bits // 8) + (40 << 30))
    reads = bf.synth_reads_device(42, 0, n, L).view(n, L)
    reads[1_000_000:1_400_000] = ord("A")                                     # 400 000 reads of poly-A
    reads[2_000_000:2_600_000] = reads[7].clone()                               # 600 000 copies of one read
    acgt = torch.tensor(list(b"ACGT" * 38)[:L], dtype=torch.uint8, device="cuda")
    reads[3_000_000:3_200_000] = acgt                                          # a period-4 repeat
    reads[4_000_000:4_000_500, 75] = ord("N")
    reads = reads.reshape(-1)
    a, b = bf.BloomFilter(bits, h, k), bf.BloomFilter(bits, h, k)
    a.setInsertMode("direct")
    b.setInsertMode("partitioned", scratch_bytes=16 << 30)
    b.setProfiling(True)
    a.insertSeqs(reads, read_len=L)
    b.insertSeqs(reads, read_len=L)
    torch.cuda.synchronize()
    prof = b.getProfile()
    assert batches_of(prof, "insert_hash") >= 1 and "insert_direct" not in prof, prof
    assert b.compare(a) == (0, 0, 0), "skewed reads: partitioned insert differs from the direct kernel"
    assert a.digest() == b.digest() and a.getPop() == b.getPop() > 0
    q = torch.cat([reads[: 1_500_000 * L], bf.synth_reads_device(43, 0, 10_000, L), reads[1_900_000 * L: 2_700_000 * L]])
    out = {}
    for mode in ("direct", "partitioned"):
        b.setQueryMode(mode)
        hit, valid, cnt = b.containsSeqs(q, read_len=L, want_valid=True, want_counts=True)
        torch.cuda.synchronize()
        out[mode] = (hit, valid, cnt.tolist())
    assert out["direct"][2] == out["partitioned"][2]
    assert bool(torch.equal(out["direct"][1], out["partitioned"][1])), "valid bitmaps differ"
    assert bool(torch.equal(out["direct"][0], out["partitioned"][0])), "hit bitmaps differ"


def test_c3_counting_filter_full_size(bf):
    """CountingBloomFilter<uint8_t>, 2^35 counters, k=25, h=3, threshold 2: incrementAll (saturation
    included) and contains through the partitioned pipeline against the direct kernels"""
    import torch

    nbytes, h, k, thr, n = 1 << 35, 3, 25, 2, 10_000_000
    need_hbm(2 * nbytes + (60 << 30))
    reads = bf.synth_reads_device(42, 0, n, L)
    sat = reads[: 3 * L].repeat(300)  # 300 copies of three reads: their counters stop at 255
    buf = torch.cat([reads, reads[: 4_000_000 * L], sat])
    a, b = bf.CountingBloomFilter(nbytes, h, k, thr), bf.CountingBloomFilter(nbytes, h, k, thr)
    a.setInsertMode("direct")
    b.setInsertMode("partitioned", scratch_bytes=16 << 30)
    b.setProfiling(True)
    a.insertSeqs(buf, read_len=L, increment_all=True)
    b.insertSeqs(buf, read_len=L, increment_all=True)
    torch.cuda.synchronize()
    prof = b.getProfile()
    assert batches_of(prof, "insert_hash") >= 2 and batches_of(prof, "insert_apply") >= 16, prof
    assert b.compare(a) == (0, 0, 0), "partitioned incrementAll differs from the direct kernel at 2^35 counters"
    assert a.popCount() == b.popCount() and a.filtered_popcount() == b.filtered_popcount() > 0
    # a counter of the 300-fold reads is saturated
    mn, _ = a.minCountSeqs(reads[:L], read_len=L)
    assert int(mn[: L - k + 1].min()) == 255
    # reads inserted twice pass threshold 2; reads inserted once mostly do not; foreign reads do not
    q = torch.cat([reads[: 5_000_000 * L], bf.synth_reads_device(44, 0, 1000, L)])
    res = {}
    for mode in ("direct", "partitioned"):
        b.setQueryMode(mode)
        hit, valid, cnt = b.containsSeqs(q, read_len=L, want_valid=True, want_counts=True)
        torch.cuda.synchronize()
        res[mode] = (hit, valid, cnt.tolist())
    assert res["direct"][2] == res["partitioned"][2]
    assert bool(torch.equal(res["direct"][0], res["partitioned"][0])) and bool(torch.equal(res["direct"][1], res["partitioned"][1]))
    assert 4_000_000 * (L - k + 1) <= res["direct"][2][1] < 4_100_000 * (L - k + 1)


def test_c5_spaced_seed_filter_full_size(bf):
    """2^37 bits, k=31, the four spaced seeds of SURVEY.md 8d with h2=1 (stHashIterator-fed filter)"""
    import torch

    bits, k, n = 1 << 37, 31, 10_000_000
    need_hbm(2 * (bits // 8) + (40 << 30))
    reads = bf.synth_reads_device(42, 0, n, L)
    reads[L * 1000 + 17] = ord("N")
    a, b = bf.BloomFilter(bits, 4, k), bf.BloomFilter(bits, 4, k)
    for f in (a, b):
        f.setSpacedSeeds(C5_SEEDS, 1)
    a.setInsertMode("direct")
    b.setInsertMode("partitioned", scratch_bytes=12 << 30)
    b.setProfiling(True)
    a.insertSeqs(reads, read_len=L)
    b.insertSeqs(reads, read_len=L)
    torch.cuda.synchronize()
    prof = b.getProfile()
    assert batches_of(prof, "insert_hash") >= 2, prof
    assert b.compare(a) == (0, 0, 0), "partitioned spaced-seed insert differs from the direct kernel at 2^37 bits"
    q = spliced_query(bf, reads, n, 10_000, 45)
    res = {}
    for mode in ("direct", "partitioned"):
        b.setQueryMode(mode)
        hit, valid, cnt = b.containsSeqs(q, read_len=L, want_valid=True, want_counts=True)
        torch.cuda.synchronize()
        res[mode] = (hit, valid, cnt.tolist())
    assert res["direct"][2] == res["partitioned"][2]
    assert bool(torch.equal(res["direct"][0], res["partitioned"][0])) and bool(torch.equal(res["direct"][1], res["partitioned"][1]))
    assert res["direct"][2][1] < res["direct"][2][0]


def test_compare_reports_differences(bf):
    """btlbf_compare itself: against numpy on small filters"""
    import numpy as np

    rng = np.random.RandomState(3)
    x, y = rng.randint(0, 256, 4096).astype(np.uint8), rng.randint(0, 256, 4096).astype(np.uint8)
    a, b = bf.BloomFilter(4096 * 8, 2, 5), bf.BloomFilter(4096 * 8, 2, 5)
    a.upload(x)
    b.upload(y)
    bx, by = np.unpackbits(x), np.unpackbits(y)
    assert a.compare(b) == (int((bx != by).sum()), int((bx > by).sum()), int((bx < by).sum()))
    assert a.compare(a) == (0, 0, 0)
    c, d = bf.CountingBloomFilter(4096, 2, 5, 1), bf.CountingBloomFilter(4096, 2, 5, 1)
    c.upload(x)
    d.upload(y)
    assert c.compare(d) == (int((x != y).sum()), int((x > y).sum()), int((x < y).sum()))

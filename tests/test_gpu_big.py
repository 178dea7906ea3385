"""GPU tests at the sizes the north star names and no fixture can hold: the real geometry of BASELINE config 4 (one
128 GiB shard of a 2^43-bit filter), a 2^41-bit (256 GiB) filter on one GPU, the reference's own ad-hoc size of
48 857 600 000 bits (Tests/AdHoc/ParallelFilter.cpp:141-145).  Two such arrays do not fit side by side, so parity is
established through the device-side digest (btlbf_digest) of the array built by the partitioned / routed pipeline
against the digest of the same array built by the oracle-pinned direct kernels (one atomicOr per probe), plus
popcounts and query bitmaps.  Position semantics: /root/reference/BloomFilter.hpp:190 (hash % size, bit p%8 of byte p/8).
"""
import ctypes as C
import numpy as np
import pytest
from conftest import require_hbm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bf():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    torch.zeros(1, device="cuda")
    import btl_bloomfilter_amd as m

    return m


def _free_gib():
    import gc

    import torch

    gc.collect()
    torch.cuda.empty_cache()
    return torch.cuda.mem_get_info()[0] / 2 ** 30


def test_digest_matches_its_definition_and_combines_over_shards(bf, oracle):
    from btl_bloomfilter_amd.engine import body_digest

    L, k, h = 150, 31, 4
    reads = bf.synth_reads_device(5, 0, 3000, L)
    for bits in (1000, 1 << 20, 3 * (1 << 20) + 64):  # 125 bytes: the last word is padded with zeros
        f = bf.BloomFilter(bits, h, k)
        assert f.digest() == (0, 0)
        f.insertSeqs(reads, read_len=L)
        body = f.download()
        assert f.digest() == body_digest(body) != (0, 0)
        # the digest is position dependent: the same bytes rotated by one word give another value
        if len(body) % 8 == 0:
            assert body_digest(np.roll(body, 8)) != body_digest(body)
    c = bf.CountingBloomFilter(1 << 16, 3, 25, 2)
    c.insertSeqs(reads, read_len=L, increment_all=True)
    assert c.digest() == body_digest(c.download())
    # shards: digests add / xor up to the whole filter's
    bits, W = 1 << 24, 4
    whole = bf.BloomFilter(bits, h, k)
    whole.setInsertMode("direct")
    whole.insertSeqs(reads, read_len=L)
    s = x = 0
    for r in range(W):
        sh = bf.BloomFilter.shard(bits, r, W, h, k)
        sh.insertSeqs(reads, read_len=L)  # a shard keeps the probes inside its bit range
        ds, dx = sh.digest()
        assert (ds, dx) == body_digest(sh.download(), first_word=r * bits // W // 64)
        s = (s + ds) & (2 ** 64 - 1)
        x ^= dx
    assert (s, x) == whole.digest() == body_digest(whole.download())


def test_clear_is_ordered_on_the_callers_stream_and_eager_after_device_ptr(bf):
    """btlbf_clear is lazy; the zeroing must still happen AFTER work queued earlier on the clear's stream, whichever
    stream carries it out, and a filter whose raw pointer has been handed out is cleared eagerly"""
    import torch

    bits, h, k, L = 1 << 33, 4, 31, 150  # 1 GiB: a direct insert of a few million reads takes milliseconds
    reads = bf.synth_reads_device(3, 0, 4_000_000, L)
    f = bf.BloomFilter(bits, h, k)
    f.setInsertMode("direct")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        f.insertSeqs(reads, read_len=L, stream=s1)  # long-running work on s1 ...
        f.clear(stream=s1)                          # ... then the clear, on s1
        # the clear is carried out by the next user of the array, here on ANOTHER stream, with no host sync between
        with torch.cuda.stream(s2):  # (the outputs are allocated and zero-filled on the stream the query runs on)
            _, _, cnt = f.containsSeqs(reads[: 1000 * L], read_len=L, want_valid=False, want_counts=True, stream=s2)
        torch.cuda.synchronize()
        assert cnt.tolist() == [1000 * 120, 0] and f.getPop() == 0
    # raw pointer handed out: from now on a clear is an ordinary memset in stream order, visible through the pointer
    import btl_bloomfilter_amd._lib as _lib

    lib = _lib.load()

    def pop_through(ptr, nbytes):  # popcount of raw device memory: no filter object involved
        out = C.c_uint64()
        _lib.check(lib.btlbf_popcount_bits(C.c_void_p(ptr), nbytes, C.byref(out), 0, None))
        return out.value

    f.insertSeqs(reads, read_len=L)
    torch.cuda.synchronize()
    ptr = f.device_ptr()
    assert ptr and pop_through(ptr, bits // 8) == f.getPop() > 0
    f.insertSeqs(reads, read_len=L, stream=s1)
    f.clear(stream=s1)
    s1.synchronize()
    assert pop_through(ptr, bits // 8) == 0  # no library call on the filter between the clear and this look
    assert f.getPop() == 0


@pytest.mark.parametrize("bits,window", [(3 << 32, False), (5 << 33, False), (7 << 34, True), (3 << 37, False), ((1 << 36) + (1 << 32), False)])
def test_sizes_of_an_odd_multiple_of_2p32_partitioned_equals_direct(bf, bits, window):
    """hash % size (BloomFilter.hpp:190) for sizes m * 2^s, s >= 32 (3 * 2^37 bits and the like): segment counts of no
    power of two, so pass A's bins are a whole number of segments (capi.cpp plan_level0) and positions come from the
    32-bit-quotient form of the reduction -- same reads through the direct kernels and the pipeline, identical arrays
    and answers; `window`: as one shard of two, so that the WINDOW form of pass A runs too."""
    import torch

    require_hbm(2 * (bits // 8) + (8 << 30), "two filters of %d bits" % bits)
    h, k, L, n = 4, 31, 150, 1_500_000
    reads = bf.synth_reads_device(bits % 1009, 0, n, L)
    if window:
        a, b = bf.BloomFilter.shard(bits, 1, 2, h, k), bf.BloomFilter.shard(bits, 1, 2, h, k)
    else:
        a, b = bf.BloomFilter(bits, h, k), bf.BloomFilter(bits, h, k)
    a.setInsertMode("direct")
    b.setInsertMode("partitioned", scratch_bytes=3 << 30)
    a.insertSeqs(reads, read_len=L)
    b.insertSeqs(reads, read_len=L)
    assert a.digest() == b.digest() and a.getPop() == b.getPop() > 0
    if not window:
        want = bits * -np.expm1(-n * (L - k + 1) * h / bits)  # uniform positions: what a wrong modulus would not give
        assert abs(a.getPop() - want) < 1e-3 * want
    q = torch.cat([reads[: 100_000 * L], bf.synth_reads_device(77, 0, 100_000, L)])
    a.setQueryMode("direct")
    b.setQueryMode("partitioned")
    ha, va, ca = a.containsSeqs(q, read_len=L, want_counts=True)
    hb, vb, cb = b.containsSeqs(q, read_len=L, want_counts=True)
    assert torch.equal(ha, hb) and torch.equal(va, vb) and ca.tolist() == cb.tolist()
    if not window:
        assert ca.tolist()[1] >= 100_000 * 120


@pytest.mark.parametrize("kind", ["plain", "spaced", "counting"])
def test_2p18_segments_with_256_level0_bins_and_a_1024_way_split(bf, monkeypatch, kind):
    """calls of 4x10^9 k-mers and more plan a filter of 2^18 segments (2^37 bits, 2^34 counters) as 256 bins x 1024 ways
    instead of 512 x 512 (capi.cpp plan_level0); the tests' batches are smaller than that, so the plan is forced here
    through the tuning knob -- against the direct kernels: plain ntHash, config 5's spaced seeds, incrementAll"""
    import torch

    h, k, L, n = 4, 31, 150, 3_000_000
    require_hbm(2 * (16 << 30) + (16 << 30), "two filters of 16 GiB")
    monkeypatch.setenv("BTLBF_SPLIT_BITS", "10")
    reads = bf.synth_reads_device(11, 0, n, L)
    if kind == "counting":
        a, b = bf.CountingBloomFilter(1 << 34, h, k, 2), bf.CountingBloomFilter(1 << 34, h, k, 2)
    else:
        a, b = bf.BloomFilter(1 << 37, h, k), bf.BloomFilter(1 << 37, h, k)
    if kind == "spaced":
        seeds = ["1110111011101110111011101110111", "1101101101101101011011011011011",
                 "1111001111001111111001111001111", "1011101011101011101011101011101"]
        a.setSpacedSeeds(seeds, 1)
        b.setSpacedSeeds(seeds, 1)
    a.setInsertMode("direct")
    b.setInsertMode("partitioned", scratch_bytes=6 << 30)
    b.setProfiling(True)
    for f in (a, b):
        if kind == "counting":
            f.insertSeqs(reads, read_len=L, increment_all=True)
            f.insertSeqs(reads[: (n // 2) * L], read_len=L, increment_all=True)  # the first half twice: threshold 2
        else:
            f.insertSeqs(reads, read_len=L)
    prof = b.getProfile()
    assert prof["insert_hash"][1] >= 1 and prof["insert_split"][1] >= 8
    assert a.digest() == b.digest()
    if kind == "counting":
        assert a.popCount() == b.popCount() > 0 and a.filtered_popcount() == b.filtered_popcount() > 0
    else:
        assert a.getPop() == b.getPop() > 0
    q = torch.cat([reads[: 200_000 * L], bf.synth_reads_device(12, 0, 200_000, L)])
    a.setQueryMode("direct")
    b.setQueryMode("partitioned")
    ha, va, ca = a.containsSeqs(q, read_len=L, want_counts=True)
    hb, vb, cb = b.containsSeqs(q, read_len=L, want_counts=True)
    assert torch.equal(ha, hb) and torch.equal(va, vb) and ca.tolist() == cb.tolist()
    assert ca.tolist()[1] >= 200_000 * 120


def test_reference_adhoc_size_48857600000_bits_partitioned_equals_direct(bf):
    """Tests/AdHoc/ParallelFilter.cpp:141-145 sizes its filter at 48 857 600 000 bits: not a power of two, more than
    2^32 bytes.  The partitioned pipeline against the direct kernels, side by side (6.1 GB each)."""
    import torch

    bits, h, k, L = 48_857_600_000, 4, 31, 150
    n = 3_000_000
    reads = bf.synth_reads_device(42, 0, n, L)
    a, b = bf.BloomFilter(bits, h, k), bf.BloomFilter(bits, h, k)
    a.setInsertMode("direct")
    b.setInsertMode("partitioned", scratch_bytes=3 << 30)  # several batches
    b.setProfiling(True)
    a.insertSeqs(reads, read_len=L)
    b.insertSeqs(reads, read_len=L)
    prof = b.getProfile()
    assert prof["insert_hash"][1] >= 2 and prof["insert_apply"][1] >= 2
    assert a.compare(b) == (0, 0, 0)
    assert a.digest() == b.digest() and a.getPop() == b.getPop() > 0
    q = torch.cat([reads[: 200_000 * L], bf.synth_reads_device(43, 0, 200_000, L)])
    b.setQueryMode("partitioned")
    a.setQueryMode("direct")
    ha, va, ca = a.containsSeqs(q, read_len=L, want_counts=True)
    hb, vb, cb = b.containsSeqs(q, read_len=L, want_counts=True)
    assert torch.equal(ha, hb) and torch.equal(va, vb) and ca.tolist() == cb.tolist()
    assert ca.tolist()[0] == 400_000 * 120 and ca.tolist()[1] >= 200_000 * 120


def _route_jobs(ops, reads, L, batch, query, hit, W, n_win, spw, fail=None, fail_count=None, cnt2=None):
    """what ShardedBloomFilter._routed_pass does for one rank whose peers are absent: every batch is routed once per
    position window; the block this rank routed to ITSELF is applied (n_blocks = 1)"""
    import torch

    dev = reads.device
    ent_b, cnt_b = ops.route_plan(batch, L)
    send_ent = torch.empty(spw * ent_b, dtype=torch.uint8, device=dev)
    send_cnt = torch.empty(spw * cnt_b, dtype=torch.uint8, device=dev)
    spill = torch.empty(1 << 16, dtype=torch.int64, device=dev)
    spill_count = torch.zeros(1, dtype=torch.int64, device=dev)
    spilled = 0
    for off in range(0, reads.numel(), batch):
        chunk = reads[off: off + batch]
        view = hit[off // 64: off // 64 + (chunk.numel() + 63) // 64] if query else None
        for w in range(n_win):
            spill_count.zero_()
            ops.route(chunk, L, batch, query, send_ent, send_cnt, view, None, cnt2 if (query and w == 0) else None, spill,
                      spill_count, window=w)
            spilled += int(spill_count.item())
            if w == 0:  # shard 0 lives in window 0: its own block is block 0 of the send set
                ops.apply_routed(send_ent[:ent_b], send_cnt[:cnt_b], 1, batch, L, query, fail, fail_count)
    return spilled, (ent_b, cnt_b)


def test_config4_real_geometry_one_shard_of_the_1tib_filter(bf):
    """BASELINE config 4 at its real size: this GPU is shard 0 of 8 of a 2^43-bit filter -- 128 GiB resident, two
    position windows of 2^42 bits, 1024 level-0 bins of 2^32 positions per window (256 per shard), two split
    passes down to 2^20 segments of 128 KiB -- the origin and owner code of the 8-GPU run, fed with this rank's
    own reads.  Checked against the same shard built by the direct WINDOW kernel (digest + popcount), the query
    bitmaps against the direct kernel's, and the shard-local partitioned pipeline (gather mode's path) on top."""
    import torch

    from btl_bloomfilter_amd.sharded import HipShardOps

    require_hbm(200 << 30, "a 128 GiB shard plus scratch")
    K, H, L, W = 31, 4, 150, 8
    n_reads, batch_reads = 1_500_000, 600_000  # three batches (the last one short), two windows each: six jobs
    reads = bf.synth_reads_device(42, 0, n_reads, L)
    foreign = bf.synth_reads_device(43, 0, 20_000, L)
    q = torch.cat([reads[: 100_000 * L], foreign, reads[100_000 * L: 150_000 * L]])
    q[7 * L + 40] = 78  # an N
    dev = reads.device

    ops = HipShardOps(1 << 43, H, K, 0, W, 0)
    n_win, spw = ops.route_windows()
    assert (n_win, spw) == (2, 4)
    bins, regions, cap, gb = ops.route_geometry(batch_reads * L, L)
    assert bins == 256  # level-0 bins per shard: 2^32 positions each
    ops.clear()
    spilled, _ = _route_jobs(ops, reads, L, batch_reads * L, 0, None, W, n_win, spw)
    assert spilled == 0
    lib = ops.L
    d_routed = (C.c_uint64 * 2)()
    pop_routed = C.c_uint64()
    assert lib.btlbf_digest(ops.f, d_routed) == 0 and lib.btlbf_popcount(ops.f, C.byref(pop_routed)) == 0
    # query through the routed path: shard 0's failed positions clear the windows that own them
    hit_r = torch.zeros((q.numel() + 63) // 64, dtype=torch.int64, device=dev)
    fail = torch.empty(4 << 20, dtype=torch.int64, device=dev)
    fail_count = torch.zeros(1, dtype=torch.int64, device=dev)
    cnt2 = torch.zeros(2, dtype=torch.int64, device=dev)
    qbatch = 64 * L * 400
    _route_jobs(ops, q, L, qbatch, 1, hit_r, W, n_win, spw, fail, fail_count, cnt2)
    n_fail = int(fail_count.item())
    assert 0 < n_fail <= fail.numel()
    ops.resolve(q, L, fail[:n_fail], hit_r)
    clean = int(cnt2[0].item())
    assert clean == (q.numel() // L) * (L - K + 1) - K  # the N kills the K windows that contain it
    routed = (int(d_routed[0]), int(d_routed[1]), int(pop_routed.value))
    ops.close()
    del ops
    torch.cuda.empty_cache()

    # the same shard from the direct kernels (atomicOr per probe inside the window; oracle-pinned)
    d = bf.BloomFilter.shard(1 << 43, 0, W, H, K)
    assert d.localBytes() == 1 << 37
    d.setInsertMode("direct")
    d.insertSeqs(reads, read_len=L)
    direct = d.digest() + (d.getPop(),)
    assert routed == direct and direct[2] > 0
    d.setQueryMode("direct")
    hit_d, valid_d, cnt_d = d.containsSeqs(q, read_len=L, want_counts=True)
    assert torch.equal(hit_d, hit_r)
    assert int(cnt_d[0]) == clean
    # every inserted k-mer answers "all of my probes are set"; of the foreign reads' k-mers, those with at least one
    # probe in this shard (1 - (7/8)^4 = 41 %) practically all fail
    hits = int(cnt_d[1])
    assert 150_000 * 120 - K <= hits < 150_000 * 120 + 0.62 * 20_000 * 120
    # the shard-local partitioned pipeline at this size (gather mode: pass A's WINDOW variant, 1024 x 1024 segments)
    d.setQueryMode("partitioned")
    hit_p, _, cnt_p = d.containsSeqs(q, read_len=L, want_counts=True)
    assert torch.equal(hit_p, hit_d) and cnt_p.tolist() == cnt_d.tolist()
    d.clear()
    d.setInsertMode("partitioned", scratch_bytes=6 << 30)
    d.setProfiling(True)
    d.insertSeqs(reads, read_len=L)
    prof = d.getProfile()
    assert prof["insert_hash"][1] >= 1 and prof["insert_split"][1] >= 1
    assert d.digest() + (d.getPop(),) == direct


def test_single_gpu_256_gib_filter_2p41_bits(bf):
    """north star: "HBM-resident bit array sized to 288 GB".  A 2^41-bit filter (256 GiB) leaves a few GB for
    scratch: 2^21 segments -> pass A x two split passes; AUTO steers such a filter to the direct kernels (small
    batches are not worth a 512 GiB sweep), the forced partitioned path must still be exact."""
    import torch

    require_hbm(262 << 30, "a 256 GiB array plus scratch")
    bits, h, k, L = 1 << 41, 4, 31, 150
    reads = bf.synth_reads_device(42, 0, 2_000_000, L)
    f = bf.BloomFilter(bits, h, k)
    f.setInsertMode("direct")
    f.insertSeqs(reads, read_len=L)
    direct = f.digest() + (f.getPop(),)
    assert direct[2] > 0.99 * 2_000_000 * 120 * 4
    f.setQueryMode("direct")
    q = torch.cat([reads[: 50_000 * L], bf.synth_reads_device(43, 0, 5_000, L)])
    hit_d, _, cnt_d = f.containsSeqs(q, read_len=L, want_valid=False, want_counts=True)
    f.setQueryMode("partitioned")
    f.setProfiling(True)
    hit_p, _, cnt_p = f.containsSeqs(q, read_len=L, want_valid=False, want_counts=True)
    prof = f.getProfile()
    assert prof["query_split"][1] >= 2 * prof["query_hash"][1] > 0  # two split levels per group
    assert torch.equal(hit_p, hit_d) and cnt_p.tolist() == cnt_d.tolist()
    f.clear()
    f.setInsertMode("partitioned")
    f.insertSeqs(reads, read_len=L)
    prof = f.getProfile()
    assert prof["insert_hash"][1] >= 1 and prof["insert_split"][1] >= 2
    assert f.digest() + (f.getPop(),) == direct
    # AUTO: the call as a whole is worth a sweep of the array (probes >= 0.95 % of its bytes, capi.cpp
    # kAutoInsertRatio), but with this little scratch one BATCH is not -> the direct kernel
    del q, hit_d, hit_p
    more = bf.synth_reads_device(42, 0, 12_000_000, L)
    f.clear()
    f.setInsertMode("auto", scratch_bytes=3 << 30)
    f.getProfile()
    f.insertSeqs(more, read_len=L)
    prof = f.getProfile()
    assert "insert_hash" not in prof and prof["insert_direct"][1] == 1
    assert f.getPop() > direct[2]


_PLAIN_SCHEDULE_CHILD = r"""
import hashlib, json, sys
import torch
import btl_bloomfilter_amd as m
bits, h, k, L, n = 1 << 36, 4, 31, 150, 2_000_000
reads = torch.cat([m.synth_reads_device(42, 0, n, L), m.synth_reads_device(44, 0, 1000, L)[: 1000 * L - 77]])
f = m.BloomFilter(bits, h, k)
f.setInsertMode("partitioned", scratch_bytes=4 << 30)
f.setProfiling(True)
f.insertSeqs(reads[: n * L], read_len=L)
f.insertSeqs(reads[n * L:])
f.setQueryMode("partitioned")
q = torch.cat([reads[: 100_000 * L], m.synth_reads_device(43, 0, 100_000, L)])
hit, valid, cnt = f.containsSeqs(q, read_len=L, want_counts=True)
print(json.dumps({"digest": list(f.digest()), "pop": f.getPop(), "counts": cnt.tolist(),
                  "hit_sha1": hashlib.sha1(hit.cpu().numpy().tobytes() + valid.cpu().numpy().tobytes()).hexdigest(),
                  "hash_launches": f.getProfile()["insert_hash"][1]}))
"""


def test_plain_and_overlapped_schedules_of_pass_a_build_the_same_filter():
    """BTLBF_PART_OVERLAP=0 selects pass A's plain schedule for the geometries that otherwise take the overlapped one
    (the switch is read once per process, hence two child processes): the same reads -- equal
    length and ragged -- give the same array (digest, popcount) and the same query bitmaps under both."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for tag, val in (("overlapped", "1"), ("plain", "0")):
        env = dict(os.environ, BTLBF_PART_OVERLAP=val, PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", _PLAIN_SCHEDULE_CHILD], env=env, cwd=root, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["plain"] == out["overlapped"]
    assert out["plain"]["pop"] > 0 and out["plain"]["hash_launches"] >= 1
    assert out["plain"]["counts"][0] >= 100_000 * 120


def _sparse_digest(positions):
    """btlbf_digest (include/btlbf.h) of an otherwise empty bit filter with exactly these bits set"""
    words = {}
    for p in positions:
        words[p >> 6] = words.get(p >> 6, 0) | (1 << (p & 63))
    M = (1 << 64) - 1

    def mix(z):
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    s = x = 0
    for i, w in words.items():
        m = mix(i + 1) | 1
        s = (s + w * m) & M
        x ^= mix(w ^ m)
    return s, x


@pytest.mark.parametrize("bits", [(1 << 32) + 64, 48_857_600_000, 3 << 37, 5 * 10 ** 11])
def test_positions_of_filters_beyond_2p32_bits_of_no_power_of_two_size(bf, bits):
    """hash % size (BloomFilter.hpp:190) for sizes whose reduction takes the 32-bit-quotient path of reduce_mod
    (device_utils.hpp: every size above 2^32 that is no power of two): 3000 rows of 64-bit hashes, inserted as rows
    and looked up again; the array must hold exactly the bits Python's `%` says -- checked through the device-side
    digest and the popcount, the filter being far too large to download."""
    require_hbm(bits // 8 + (4 << 30), "a filter of %d bits" % bits)
    rng = np.random.RandomState(bits % 9973)
    h = 3
    hv = rng.randint(0, 2 ** 63, size=(3000, h)).astype(np.uint64) * np.uint64(2) + rng.randint(0, 2, size=(3000, h)).astype(np.uint64)
    hv[0, 0] = np.uint64(2 ** 64 - 1)
    hv[1, 0] = np.uint64(bits)          # -> position 0
    hv[2, 0] = np.uint64(bits - 1)      # -> the last position
    hv[3, 0] = np.uint64((2 ** 64 // bits) * bits - 1)
    f = bf.BloomFilter(bits, h, 20)
    f.insert(hv)
    want = {int(v) % bits for v in hv.ravel().tolist()}
    assert f.getPop() == len(want)
    assert f.digest() == _sparse_digest(want)
    assert f.contains(hv).all()
    other = hv ^ np.uint64(0x5555555555555555)
    hits = f.contains(other)
    expect = np.array([all((int(v) % bits) in want for v in row) for row in other.tolist()])
    assert hits.tolist() == expect.tolist()

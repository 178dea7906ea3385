"""GPU parity tests: the HIP path, called through the C ABI, against the golden vectors of the
genuine reference and against the CPU oracle on seeded random inputs.  Bit-exact everywhere
(integer work)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, require_hbm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bf():
    # torch first, as in every production flow (bench.py, smoke(), the sharded path): its HIP runtime
    # and context are up before the library makes its first call
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    torch.zeros(1, device="cuda")
    import btl_bloomfilter_amd as m

    assert m._lib.load().btlbf_device_count() > 0, "GPU tests need a GPU (and the HIP library)"
    return m


def unhex(lst, cols):
    a = np.array([int(x, 16) for x in lst], dtype=np.uint64)
    return a.reshape(-1, cols) if cols else a


def rand_seq(rng, n, p_bad=0.02):
    s = rng.choice(list(b"ACGTacgt"), n).astype(np.uint8)
    bad = rng.rand(n) < p_bad
    s[bad] = rng.choice(list(b"NnRY-\x00\x01\x07\xff"), bad.sum())
    return s.tobytes()


def emitted(hv, valid_bits, n, k):
    nw = max(n - k + 1, 0)
    v = __import__("btl_bloomfilter_amd").bits_to_bool(valid_bits, n)
    assert not v[nw:].any(), "window beyond the end of the buffer reported clean"
    pos = np.flatnonzero(v)
    return pos, np.asarray(hv)[pos]


# ---------------------------------------------------------------------------------------------
# hash streams
# ---------------------------------------------------------------------------------------------
def test_g1_nthash_golden(bf):
    g = load_golden("hash_vectors.json")["nthash"]
    nk = 0
    for case in g:
        s = case["seq"].encode("latin-1")
        if not s:
            continue
        hv, valid = bf.hash_seqs(s, case["h"], case["k"])
        pos, hh = emitted(hv, valid, len(s), case["k"])
        assert pos.tolist() == case["pos"], (case["seq"][:20], case["k"])
        assert (hh == unhex(case["hashes"], case["h"])).all()
        nk += len(pos)
    assert nk > 10000


def test_g2_sthash_golden(bf):
    for case in load_golden("hash_vectors.json")["sthash"]:
        s = case["seq"].encode("latin-1")
        m = len(case["seeds"]) * case["h2"]
        hv, valid, st = bf.sthash_seqs(s, case["seeds"], case["h2"], case["k"])
        pos, hh = emitted(hv, valid, len(s), case["k"])
        assert pos.tolist() == case["pos"]
        assert (hh == unhex(case["hashes"], m)).all()
        strand = ((np.asarray(st)[pos][:, None] >> np.arange(m, dtype=np.uint64)) & np.uint64(1)).ravel()
        assert strand.tolist() == case["strand"]


@pytest.mark.parametrize("k,h", [(1, 1), (4, 5), (25, 3), (31, 4), (32, 2), (64, 4), (150, 2), (301, 1)])
def test_nthash_random_vs_oracle(bf, oracle, k, h):
    rng = np.random.RandomState(k * 7 + h)
    for n in (1, k - 1, k, k + 1, 2047, 2048, 2049, 2048 + k - 1, 5000, 40000):
        if n <= 0:
            continue
        s = rand_seq(rng, n, p_bad=rng.choice([0, 0.003, 0.05]))
        hv, valid = bf.hash_seqs(s, h, k)
        pos, hh = emitted(hv, valid, n, k)
        op, oh = oracle.nthash_seq(s, h, k)
        assert pos.tolist() == op.tolist(), (n, k)
        assert (hh == oh).all()


@pytest.mark.parametrize("k", [2, 3, 5, 8, 30, 31, 33, 64, 127, 128])
def test_nthash_start_up_pair_table_and_raw_byte_codes(bf, oracle, k):
    # a lane's first window comes from a two-base table that holds A C G T only (seq_core.hpp); windows with one of
    # the other bytes the reference's seed table gives a value (the raw bytes 1 3 4 5 7; U and u share T's code) take the Horner form: both, densely
    # mixed, at even and odd k, against the oracle (reference: nthash.hpp:129-147 NTMC64 initial values)
    rng = np.random.RandomState(900 + k)
    for alpha, p_raw in ((b"\x01\x03\x04\x05\x07Uu", 0.01), (b"\x01\x03\x04\x05\x07Uu", 0.3), (b"\x00\x02\x06Nn", 0.05)):
        n = 30000 + k
        s = rng.choice(list(b"ACGTacgt"), n).astype(np.uint8)
        raw = rng.rand(n) < p_raw
        s[raw] = rng.choice(list(alpha), raw.sum())
        s = s.tobytes()
        hv, valid = bf.hash_seqs(s, 3, k)
        pos, hh = emitted(hv, valid, n, k)
        op, oh = oracle.nthash_seq(s, 3, k)
        assert pos.tolist() == op.tolist(), (k, p_raw)
        assert (hh == oh).all(), (k, p_raw)


def test_layouts_ragged_uniform_and_misaligned_device_pointers(bf, oracle):
    import torch

    rng = np.random.RandomState(11)
    k, h = 31, 4
    lens = [0, 5, 30, 31, 32, 150, 150, 1, 2047, 2048, 4100, 0, 77, 31]
    reads = [rand_seq(rng, n, 0.01) for n in lens]
    buf = b"".join(reads)
    starts = np.cumsum([0] + lens).astype(np.uint64)
    exp_pos, exp_h = [], []
    for st, r in zip(starts, reads):
        p, hv = oracle.nthash_seq(r, h, k)
        exp_pos += (p + st).tolist()
        exp_h.append(hv)
    exp_h = np.concatenate(exp_h)
    # host, ragged
    hv, valid = bf.hash_seqs(buf, h, k, starts=starts)
    pos, hh = emitted(hv, valid, len(buf), k)
    assert pos.tolist() == exp_pos and (hh == exp_h).all()
    # device, ragged, every misalignment of the base pointer
    for mis in (0, 1, 3, 8, 15):
        t = torch.zeros(len(buf) + 32, dtype=torch.uint8, device="cuda")
        t[mis:mis + len(buf)] = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
        ts = torch.from_numpy(starts.astype(np.int64)).cuda()
        hv, valid = bf.hash_seqs(t[mis:mis + len(buf)], h, k, starts=ts)
        torch.cuda.synchronize()
        pos, hh = emitted(hv.cpu().numpy().view(np.uint64), valid.cpu().numpy().view(np.uint64), len(buf), k)
        assert pos.tolist() == exp_pos and (hh == exp_h).all(), mis
    # uniform reads of several lengths (including shorter than k and not a multiple of anything)
    for L in (150, 31, 30, 7, 100, 2048, 3000):
        n_reads = max(3, 9000 // L)
        rs = [rand_seq(rng, L, 0.01) for _ in range(n_reads)]
        exp_pos, exp_h = [], []
        for i, r in enumerate(rs):
            p, hv = oracle.nthash_seq(r, h, k)
            exp_pos += (p + i * L).tolist()
            exp_h.append(hv)
        hv, valid = bf.hash_seqs(b"".join(rs), h, k, read_len=L)
        pos, hh = emitted(hv, valid, L * n_reads, k)
        assert pos.tolist() == exp_pos, L
        assert (hh == np.concatenate(exp_h)).all()


# ---------------------------------------------------------------------------------------------
# bit filter
# ---------------------------------------------------------------------------------------------
def test_g3_bloom_files_whole_bytes(bf, tmp_path):
    for f in load_golden("files.json"):
        if f["kind"] != "bloom":
            continue
        flt = bf.BloomFilter(f["bits"], f["h"], f["k"])
        for s in f["inserted"]:
            flt.insertSeqs(s)
        flt.setnEntry(f["n_entry"])
        flt.settEntry(f["t_entry"])
        p = tmp_path / f["file"]
        flt.storeFilter(p)
        raw = open(os.path.join(GOLDEN, f["file"]), "rb").read()
        assert open(p, "rb").read() == raw, f["file"]
        assert flt.getPop() == f["pop"]
        # the reference's own file loads and answers like the filter that wrote it
        g = bf.BloomFilter(path=os.path.join(GOLDEN, f["file"]))
        assert (g.getFilterSize(), g.getHashNum(), g.getKmerSize()) == (f["bits"], f["h"], f["k"])
        assert (g.getnEntry(), g.gettEntry()) == (f["n_entry"], f["t_entry"])
        assert (g.download() == flt.download()).all()
        for s in f["inserted"]:
            hit, valid, _ = g.containsSeqs(s)
            assert (bf.bits_to_bool(hit, len(s)) == bf.bits_to_bool(valid, len(s))).all()


def test_g5_contains_bitmasks(bf):
    ops = load_golden("filter_ops.json")
    for key in ("contains_65536", "contains_100000"):
        g = ops[key]
        flt = bf.BloomFilter(g["bits"], g["h"], g["k"])
        buf = "".join(g["A"]).encode()
        flt.insertSeqs(buf, read_len=150)
        body = flt.download()
        assert hashlib.sha256(body.tobytes()).hexdigest() == g["body_sha256"]
        assert flt.getPop() == g["pop"]
        for s, r in zip(g["B"], g["result"]):
            hit, valid, cnt = flt.containsSeqs(s, want_counts=True)
            v = bf.bits_to_bool(valid, len(s))
            hbits = bf.bits_to_bool(hit, len(s))
            assert np.flatnonzero(v).tolist() == r["pos"]
            assert hbits[v].astype(int).tolist() == r["hit"]
            assert cnt.tolist() == [len(r["pos"]), sum(r["hit"])]
        # all reads of B in one ragged call
        lens = [len(s) for s in g["B"]]
        starts = np.cumsum([0] + lens).astype(np.uint64)
        hit, valid, _ = flt.containsSeqs("".join(g["B"]).encode(), starts=starts)
        hb, vb = bf.bits_to_bool(hit, starts[-1]), bf.bits_to_bool(valid, starts[-1])
        for st, s, r in zip(starts, g["B"], g["result"]):
            st = int(st)
            assert (np.flatnonzero(vb[st:st + len(s)])).tolist() == r["pos"]
            assert hb[st:st + len(s)][vb[st:st + len(s)]].astype(int).tolist() == r["hit"]


def test_g6_insert_and_check(bf, oracle):
    g = load_golden("filter_ops.json")["insert_and_check"]
    flt = bf.BloomFilter(g["bits"], g["h"], g["k"])
    for s, r in zip(g["stream"], g["result"]):
        _, hv = oracle.nthash_seq(s, g["h"], g["k"])
        assert flt.insertAndCheck(hv, serial=True).tolist() == r
    assert hashlib.sha256(flt.download().tobytes()).hexdigest() == g["body_sha256"]
    # sequence form on a buffer without repeated k-mers: parallel order cannot matter
    flt2 = bf.BloomFilter(g["bits"], g["h"], g["k"])
    s = g["stream"][0]
    hit, valid, _ = flt2.insertAndCheckSeqs(s)
    assert bf.bits_to_bool(hit, len(s))[bf.bits_to_bool(valid, len(s))].astype(int).tolist() == g["result"][0]
    hit, valid, _ = flt2.insertAndCheckSeqs(s)
    assert (bf.bits_to_bool(hit, len(s)) == bf.bits_to_bool(valid, len(s))).all()


def test_g8_modulo_edges(bf):
    for g in load_golden("filter_ops.json")["modulo"]:
        flt = bf.BloomFilter(g["size"], 1, 4)
        flt.insert(unhex(g["hashes"], 1))
        bits = np.flatnonzero(np.unpackbits(flt.download(), bitorder="little"))
        assert bits.tolist() == g["set_bits"], g["size"]
        assert flt.contains(unhex(g["hashes"], 1)).all()


@pytest.mark.parametrize("bits", [64, 1000, 4096, 999992, 1 << 20, (1 << 20) + 8])
def test_hash_rows_random_vs_oracle(bf, oracle, bits):
    rng = np.random.RandomState(bits % 1000)
    h = 5
    hv = rng.randint(0, 2 ** 63, size=(3000, h)).astype(np.uint64) * np.uint64(2) + rng.randint(0, 2, size=(3000, h)).astype(np.uint64)
    flt = bf.BloomFilter(bits, h, 20)
    mine = np.zeros(bits // 8, np.uint8)
    flt.insert(hv[:1500])
    oracle.bf_insert(mine, bits, h, hv[:1500])
    assert (flt.download() == mine).all()
    assert flt.contains(hv).tolist() == oracle.bf_contains(mine, bits, h, hv).tolist()
    assert flt.insertAndCheck(hv[1000:2000], serial=True).tolist() == oracle.bf_insert_and_check(mine, bits, h, hv[1000:2000]).tolist()
    assert (flt.download() == mine).all()
    assert flt.getPop() == oracle.bf_popcount(mine, bits)


@pytest.mark.parametrize("k,h,bits", [(31, 4, 1 << 22), (25, 3, 3000000), (8, 7, 1 << 16), (64, 2, 1 << 20)])
def test_insert_contains_seqs_random_vs_oracle(bf, oracle, k, h, bits):
    rng = np.random.RandomState(k + h)
    a = rand_seq(rng, 30000, 0.004)
    b = a[:9000] + rand_seq(rng, 9000, 0.004)
    flt = bf.BloomFilter(bits, h, k)
    flt.insertSeqs(a)
    mine = np.zeros(bits // 8, np.uint8)
    oracle.bf_insert_seq(mine, bits, h, k, a)
    assert (flt.download() == mine).all()
    hit, valid, cnt = flt.containsSeqs(b, want_counts=True)
    ohit, ovalid = oracle.bf_contains_seq_dense(mine, bits, h, k, b)
    nw = len(b) - k + 1
    assert (bf.bits_to_bool(valid, len(b))[:nw] == ovalid.astype(bool)).all()
    assert (bf.bits_to_bool(hit, len(b))[:nw] == ohit.astype(bool)).all()
    assert cnt.tolist() == [int(ovalid.sum()), int(ohit.sum())]


def test_reference_unit_test_bloom(bf, tmp_path):
    # mirrors Tests/Unit/BloomFilterTests.cpp:69-139 (1e9-bit filter, h=5, k=4, "ACGTAC")
    flt = bf.BloomFilter(1000000000, 5, 4)
    hv, valid = bf.hash_seqs(b"ACGTAC", 5, 4)
    pos, rows = emitted(hv, valid, 6, 4)
    assert pos.tolist() == [0, 1, 2]
    flt.insert(rows)
    assert flt.contains(rows).all()
    p = tmp_path / "u.bf"
    flt.storeFilter(p)
    raw = open(p, "rb").read()
    i = raw.index(b"[HeaderEnd]\n") + 12
    assert len(raw) - i == flt.sizeInBytes() == 125000000
    f2 = bf.BloomFilter(path=p)
    assert f2.contains(rows).all() and f2.getPop() == flt.getPop() <= 15


def test_spaced_seed_filter(bf, oracle):
    seeds = ["1110111011101110111011101110111", "1101101101101101011011011011011",
             "1111001111001111111001111001111", "1011101011101011101011101011101"]
    rng = np.random.RandomState(3)
    s = rand_seq(rng, 20000, 0.002)
    bits = 1 << 22
    flt = bf.BloomFilter(bits, 4, 31)
    flt.setSpacedSeeds(seeds, 1)
    flt.insertSeqs(s)
    pos, hv, st = oracle.sthash_seq(s, seeds, 1, 31)
    mine = np.zeros(bits // 8, np.uint8)
    oracle.bf_insert(mine, bits, 4, hv)
    assert (flt.download() == mine).all()
    hit, valid, cnt = flt.containsSeqs(s, want_counts=True)
    assert cnt.tolist() == [len(pos), len(pos)]


def test_mibf_stage1_bit_vector(bf, oracle):
    """stage 1 of the miBF build (MIBFConstructSupport.hpp:55-87 insertBVColli / insertBV,
    MIBloomFilter.hpp:94-104) sets bit `hash % bv.size()` of an sdsl::bit_vector -- 64-bit words, LSB
    first -- for the stHashIterator hashes of every k-mer and counts the k-mers whose h bits were all
    set already.  On a little-endian machine that word array IS this engine's filter body, so the
    spaced-seed BloomFilter is the drop-in for that stage: body == the word array, and the serial
    insertAndCheck total == insertBVColli's return value."""
    seeds = ["1110111011101110111011101110111", "1101101101101101011011011011011",
             "1111001111001111111001111001111", "1011101011101011101011101011101"]
    rng = np.random.RandomState(11)
    base = rand_seq(rng, 6000, 0.003)
    s = base + base[1000:3000] + rand_seq(rng, 3000, 0.0)  # a repeated stretch: collisions
    k, h = 31, len(seeds)
    size = 64 * 5003  # MIBloomFilter::calcOptimalSize returns a multiple of 64, not a power of two
    pos, hv, _ = oracle.sthash_seq(s, seeds, 1, k)
    words = np.zeros(size // 64, np.uint64)
    colli = 0
    for row in hv:  # the reference's loop, serial order
        c = 0
        for x in row:
            p = int(x) % size
            c += int(words[p >> 6] >> np.uint64(p & 63)) & 1
            words[p >> 6] |= np.uint64(1 << (p & 63))
        colli += c == h
    assert colli > 0
    # (a) whole-buffer insert, parallel order
    flt = bf.BloomFilter(size, h, k)
    flt.setSpacedSeeds(seeds, 1)
    flt.insertSeqs(s)
    assert (flt.download().view(np.uint64) == words).all()
    # (b) insertBVColli: hash rows from the device iterator, applied in buffer order
    rows, valid, _ = bf.sthash_seqs(s, seeds, 1, k)
    vb = np.unpackbits(np.asarray(valid).view(np.uint8), bitorder="little")[: len(s)].astype(bool)
    assert np.flatnonzero(vb).tolist() == pos.tolist()
    f2 = bf.BloomFilter(size, h, k)
    prev = f2.insertAndCheck(np.asarray(rows)[vb], serial=True)
    assert int(np.asarray(prev).sum()) == colli
    assert (f2.download().view(np.uint64) == words).all()


# ---------------------------------------------------------------------------------------------
# counting filter
# ---------------------------------------------------------------------------------------------
def test_g4_counting_files(bf, tmp_path):
    for f in load_golden("files.json"):
        if f["kind"] != "counting":
            continue
        c = bf.CountingBloomFilter(f["bytes"], f["h"], f["k"], f["thr"])
        assert c.size() == f["size"] == c.sizeInBytes()
        for s in f["inserted"][:40]:
            c.insertSeqs(s, increment_all=(f["op"] != "insert"), serial=True)
        rest = f["inserted"][40:]
        if rest:  # the saturation cases repeat one read 300 times: feed the rest as one ragged buffer
            starts = np.cumsum([0] + [len(s) for s in rest]).astype(np.uint64)
            c.insertSeqs("".join(rest).encode(), starts=starts, increment_all=(f["op"] != "insert"), serial=True)
        p = tmp_path / f["file"]
        c.storeFilter(p)
        assert open(p, "rb").read() == open(os.path.join(GOLDEN, f["file"]), "rb").read(), f["file"]
        assert c.popCount() == f["popcount"] and c.filtered_popcount() == f["filtered_popcount"]
        c2 = bf.CountingBloomFilter(path=os.path.join(GOLDEN, f["file"]), countThreshold=f["thr"])
        mins, hits = [], []
        for s in f["inserted"]:
            mn, valid = c2.minCountSeqs(s)
            v = bf.bits_to_bool(valid, len(s))
            mins += mn[v].tolist()
            hit, _, _ = c2.containsSeqs(s)
            hits += bf.bits_to_bool(hit, len(s))[v].astype(int).tolist()
        assert mins == f["min_counts"] and hits == f["contains"]


def test_counting_hash_rows_vs_oracle(bf, oracle):
    rng = np.random.RandomState(5)
    h, thr = 3, 2
    c = bf.CountingBloomFilter(1001, h, 25, thr)
    mine = np.zeros(1008, np.uint8)
    for rnd in range(6):
        hv = rng.randint(0, 2 ** 62, size=(700, h)).astype(np.uint64)
        hv[::7, 1] = hv[::7, 0]
        if rnd % 3 == 0:
            c.incrementMin(hv, serial=True)
            oracle.cbf_increment_min(mine, h, hv)
        elif rnd % 3 == 1:
            c.incrementAll(hv)  # parallel: saturating adds commute
            oracle.cbf_increment_all(mine, h, hv)
        else:
            assert c.insertAndCheck(hv, serial=True).tolist() == oracle.cbf_insert_and_check(mine, h, thr, hv).tolist()
        assert (c.download() == mine).all(), rnd
        mn, ct = oracle.cbf_query(mine, h, thr, hv)
        assert c.minCount(hv).tolist() == mn.tolist() and c.contains(hv).tolist() == ct.tolist()
    assert c.popCount() == oracle.cbf_popcount(mine)
    assert c.filtered_popcount() == oracle.cbf_filtered_popcount(mine, thr)


def test_counting_parallel_increment_min_contract(bf, oracle):
    # parallel incrementMin is not bit-reproducible (neither is the reference
    # under OpenMP); it must stay <= incrementAll counters and never lose a k-mer inserted >= thr times
    d = load_golden("digests.json")["cbf_small_all"]
    reads = oracle.synth_reads(42, 0, 4000, 150)
    cmin = bf.CountingBloomFilter(1 << 20, 3, 25, 2)
    call = bf.CountingBloomFilter(1 << 20, 3, 25, 2)
    for _ in range(2):
        cmin.insertSeqs(reads, read_len=150)
        call.insertSeqs(reads, read_len=150, increment_all=True)
    a, b = cmin.download(), call.download()
    assert (a <= b).all() and a.sum() > 0
    hit, valid, cnt = cmin.containsSeqs(reads, read_len=150, want_counts=True)
    assert cnt[0] == cnt[1] == 4000 * 126
    # serial incrementMin is exact
    ser = bf.CountingBloomFilter(1 << 20, 3, 25, 2)
    mine = np.zeros(1 << 20, np.uint8)
    ser.insertSeqs(reads[:150 * 300], read_len=150, serial=True)
    for r in reads[:150 * 300].reshape(300, 150):
        _, hv = oracle.nthash_seq(r.tobytes(), 3, 25)
        oracle.cbf_increment_min(mine, 3, hv)
    assert (ser.download() == mine).all()
    assert d["op"] == 1


# ---------------------------------------------------------------------------------------------
# G7: digests over synthetic reads (generated on the device)
# ---------------------------------------------------------------------------------------------
def test_synth_reads_device(bf, oracle):
    d = load_golden("digests.json")["synth"]
    r = bf.synth_reads_device(42, 0, 1000, 150).cpu().numpy()
    assert r[:450].tobytes().decode() == d["first3"]
    assert hashlib.sha256(r.tobytes()).hexdigest() == d["sha256_first_1000"]
    assert hashlib.sha256(bf.synth_reads_device(43, 12345, 100, 150).cpu().numpy().tobytes()).hexdigest() == d["sha256_seed43_from_12345"]
    assert hashlib.sha256(bf.synth_reads_device(7, 5, 64, 100).cpu().numpy().tobytes()).hexdigest() == d["sha256_len100"]


@pytest.mark.parametrize("name", ["bf_small", "bf_nonpow2", "bf_medium", "bf_config1"])
def test_g7_bloom_digests(bf, name):
    import torch

    g = load_golden("digests.json")[name]
    flt = bf.BloomFilter(g["bits"], g["h"], g["k"])
    step = 250000
    for first in range(0, g["n_reads"], step):
        n = min(step, g["n_reads"] - first)
        reads = bf.synth_reads_device(g["seed"], first, n, g["read_len"])
        flt.insertSeqs(reads, read_len=g["read_len"])
    torch.cuda.synchronize()
    assert flt.getPop() == g["pop"]
    assert hashlib.sha256(flt.download().tobytes()).hexdigest() == g["body_sha256"]
    for seed, key in ((42, "hits_seed42"), (43, "hits_seed43")):
        q = bf.synth_reads_device(seed, 0, g["n_query"], g["read_len"])
        _, _, cnt = flt.containsSeqs(q, read_len=g["read_len"], want_valid=False, want_counts=True)
        torch.cuda.synchronize()
        assert cnt.cpu().tolist() == [g["n_query"] * (g["read_len"] - g["k"] + 1), g[key]]


@pytest.mark.parametrize("name", ["cbf_small_all", "cbf_medium_all", "cbf_small_min"])
def test_g7_counting_digests(bf, name):
    g = load_golden("digests.json")[name]
    c = bf.CountingBloomFilter(g["bytes"], g["h"], g["k"], g["thr"])
    for n in (g["n_reads"], g["n_reads"] // 2):
        reads = bf.synth_reads_device(g["seed"], 0, n, g["read_len"])
        c.insertSeqs(reads, read_len=g["read_len"], increment_all=(g["op"] == 1), serial=(g["op"] == 0))
    assert c.popCount() == g["popcount"] and c.filtered_popcount() == g["filtered_popcount"]
    assert hashlib.sha256(c.download().tobytes()).hexdigest() == g["body_sha256"]


# ---------------------------------------------------------------------------------------------
# size-independent properties at the BASELINE filter size (2^39 bits = 64 GiB)
# ---------------------------------------------------------------------------------------------
def test_full_size_filter_properties(bf):
    import torch

    free, _ = torch.cuda.mem_get_info()
    bits = 1 << 39
    require_hbm((bits // 8) + (8 << 30), "a 64 GiB filter")
    k, h, L, n = 31, 4, 150, 200000
    flt = bf.BloomFilter(bits, h, k)
    reads = bf.synth_reads_device(42, 0, n, L)
    flt.insertSeqs(reads, read_len=L)
    pop1 = flt.getPop()
    # expected population = number of distinct positions, from the hash-only kernel + numpy
    hv, valid = bf.hash_seqs(reads[: 20000 * L], h, k, read_len=L)
    v = bf.bits_to_bool(valid.cpu().numpy().view(np.uint64), 20000 * L)
    assert v.sum() == 20000 * 120
    sub = bf.BloomFilter(bits, h, k)
    sub.insertSeqs(reads[: 20000 * L], read_len=L)
    pos = hv.cpu().numpy().view(np.uint64)[v] & np.uint64(bits - 1)
    assert sub.getPop() == len(np.unique(pos))
    # idempotence and no false negatives
    flt.insertSeqs(reads, read_len=L)
    assert flt.getPop() == pop1 and pop1 <= n * 120 * h
    _, _, cnt = flt.containsSeqs(reads, read_len=L, want_valid=False, want_counts=True)
    assert cnt.cpu().tolist() == [n * 120, n * 120]
    q = bf.synth_reads_device(43, 0, n, L)
    _, _, cnt = flt.containsSeqs(q, read_len=L, want_valid=False, want_counts=True)
    assert cnt[0].item() == n * 120 and cnt[1].item() < 10


# ---------------------------------------------------------------------------------------------
# partitioned insert (radix partition by segment + LDS apply): same bytes as the direct kernel
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["bf_small", "bf_nonpow2", "bf_medium", "bf_config1"])
def test_partitioned_insert_golden_digests(bf, name):
    g = load_golden("digests.json")[name]  # 1-level (small, nonpow2) and 2-level (medium, config1) geometries
    flt = bf.BloomFilter(g["bits"], g["h"], g["k"])
    flt.setInsertMode("partitioned")
    reads = bf.synth_reads_device(g["seed"], 0, g["n_reads"], g["read_len"])
    flt.insertSeqs(reads, read_len=g["read_len"])
    assert flt.getPop() == g["pop"]
    assert hashlib.sha256(flt.download().tobytes()).hexdigest() == g["body_sha256"]


@pytest.mark.parametrize("bits,k,h", [(1000, 5, 4), (1 << 19, 31, 1), ((1 << 19) + 64, 31, 3), (3 << 29, 25, 5),
                                      (1 << 31, 64, 2), (1 << 28, 25, 8), (1 << 27, 150, 7), (5 << 20, 1, 6)])
def test_partitioned_equals_direct_odd_shapes(bf, oracle, bits, k, h):
    import torch

    rng = np.random.RandomState(bits % 977 + k)
    lens = [int(x) for x in rng.randint(0, 400, 300)] + [5000, 0, k, k - 1]
    reads = [rand_seq(rng, n, 0.01) for n in lens]
    buf = b"".join(reads)
    starts = np.cumsum([0] + lens).astype(np.uint64)
    a, b = bf.BloomFilter(bits, h, k), bf.BloomFilter(bits, h, k)
    a.setInsertMode("direct")
    b.setInsertMode("partitioned")
    # ragged layout on a deliberately misaligned device pointer
    t = torch.zeros(len(buf) + 16, dtype=torch.uint8, device="cuda")
    t[3:3 + len(buf)] = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
    ts = torch.from_numpy(starts.astype(np.int64)).cuda()
    a.insertSeqs(t[3:3 + len(buf)], starts=ts)
    b.insertSeqs(t[3:3 + len(buf)], starts=ts)
    torch.cuda.synchronize()
    assert a.getPop() == b.getPop() > 0
    if bits <= 1 << 24:
        mine = np.zeros(bits // 8, np.uint8)
        for r in reads:
            oracle.bf_insert_seq(mine, bits, h, k, r)
        assert (b.download() == mine).all()
    else:
        assert hashlib.sha256(a.download()).hexdigest() == hashlib.sha256(b.download()).hexdigest()


@pytest.mark.parametrize("bits", [1 << 30, 3 << 29, (5 << 28) + 192, 1_000_000_072])
def test_partitioned_small_scratch_many_batches_and_skew(bf, bits):
    """a scratch cap forces several batches; 4000 copies of one read overflow their bins, which must
    fall back to direct atomics instead of dropping entries.  Sizes of no power of two: level-0 bins are then a whole
    number of segments (capi.cpp plan_level0), and the overflow paths of passes A and B rebuild positions from them"""
    import torch

    h, k, L = 4, 31, 150
    reads = bf.synth_reads_device(42, 0, 60000, L)
    skew = reads[:L].repeat(4000)
    buf = torch.cat([reads, skew, reads[: 1000 * L]])
    a, b = bf.BloomFilter(bits, h, k), bf.BloomFilter(bits, h, k)
    a.setInsertMode("direct")
    b.setInsertMode("partitioned", scratch_bytes=48 << 20)
    a.insertSeqs(buf, read_len=L)
    b.insertSeqs(buf, read_len=L)
    torch.cuda.synchronize()
    assert a.getPop() == b.getPop()
    assert hashlib.sha256(a.download()).hexdigest() == hashlib.sha256(b.download()).hexdigest()
    b.setQueryMode("partitioned")  # (the scratch budget set above holds for queries too)
    a.setQueryMode("direct")
    q = torch.cat([skew[: 2000 * L], bf.synth_reads_device(5, 0, 3000, L), reads[: 3000 * L]])
    ha, va, ca = a.containsSeqs(q, read_len=L, want_counts=True)
    hb, vb, cb = b.containsSeqs(q, read_len=L, want_counts=True)
    assert torch.equal(ha, hb) and torch.equal(va, vb) and ca.tolist() == cb.tolist()


@pytest.mark.parametrize("n_seeds,h2", [(2, 2), (3, 1), (4, 2), (1, 5)])
def test_partitioned_spaced_seed_shapes(bf, n_seeds, h2):
    """seeds x extra hashes per seed in the partitioned pass (hash values indexed statically there)
    against the direct kernel, insert and query"""
    import torch

    seeds = ["1110111011101110111011101110111", "1101101101101101011011011011011",
             "1111001111001111111001111001111", "1011101011101011101011101011101"][:n_seeds]
    bits, L, h = 1 << 30, 150, n_seeds * h2
    reads = bf.synth_reads_device(9, 0, 40000, L)
    reads[1000:1010] = ord("N")
    a, b = bf.BloomFilter(bits, h, 31), bf.BloomFilter(bits, h, 31)
    for f, mode in ((a, "direct"), (b, "partitioned")):
        f.setSpacedSeeds(seeds, h2)
        f.setInsertMode(mode)
        f.setQueryMode(mode)
        f.insertSeqs(reads, read_len=L)
    torch.cuda.synchronize()
    assert a.getPop() == b.getPop() > 0
    assert (a.download() == b.download()).all()
    q = torch.cat([reads[: 5000 * L], bf.synth_reads_device(10, 0, 20, L)])
    ha, _, ca = a.containsSeqs(q, read_len=L, want_valid=False, want_counts=True)
    hb, _, cb = b.containsSeqs(q, read_len=L, want_valid=False, want_counts=True)
    torch.cuda.synchronize()
    assert ca.tolist() == cb.tolist() and bool((ha == hb).all().item())


def test_partitioned_spaced_seeds_and_shard(bf):
    import torch

    seeds = ["1110111011101110111011101110111", "1101101101101101011011011011011",
             "1111001111001111111001111001111", "1011101011101011101011101011101"]
    bits, L = 1 << 30, 150
    reads = bf.synth_reads_device(7, 0, 30000, L)
    a, b = bf.BloomFilter(bits, 4, 31), bf.BloomFilter(bits, 4, 31)
    for f, mode in ((a, "direct"), (b, "partitioned")):
        f.setSpacedSeeds(seeds, 1)
        f.setInsertMode(mode)
        f.insertSeqs(reads, read_len=L)
    torch.cuda.synchronize()
    assert (a.download() == b.download()).all()
    # a shard only keeps the positions it owns, in either mode
    whole = bf.BloomFilter(bits, 4, 31)
    whole.insertSeqs(reads, read_len=L)
    body = whole.download()
    for idx in (0, 3):
        s = bf.BloomFilter.shard(bits, idx, 4, 4, 31)
        s.setInsertMode("partitioned")
        s.insertSeqs(reads, read_len=L)
        n = bits // 8 // 4
        assert (s.download() == body[idx * n:(idx + 1) * n]).all()


# ---------------------------------------------------------------------------------------------
# partitioned contains(): test in LDS + failed-position set; must equal the direct gather kernel
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("bits,miss_reads", [(1 << 30, 0), (1 << 30, 3), (1 << 30, 4000), (3 << 22, 50), (1 << 33, 7),
                                             (1 << 30, 20000), (1 << 30, 39000), (3 << 28, 10000)])
def test_partitioned_query_equals_direct(bf, bits, miss_reads):
    import torch

    k, h, L, n = 31, 4, 150, 40000
    reads = bf.synth_reads_device(42, 0, n, L)
    flt = bf.BloomFilter(bits, h, k)
    flt.insertSeqs(reads, read_len=L)
    # query = inserted reads with a few foreign reads spliced in (all hits / few misses / too many
    # misses for the fail list -> falls back to the direct kernel per batch)
    q = reads.clone()
    if miss_reads:
        other = bf.synth_reads_device(43, 0, miss_reads, L)
        idx = torch.arange(miss_reads, device="cuda") * (n // miss_reads)
        q.view(n, L)[idx] = other.view(miss_reads, L)
    q[12345] = ord("N")  # an unclean window or 31
    out = {}
    for mode in ("direct", "partitioned"):
        flt.setQueryMode(mode)
        hit, valid, cnt = flt.containsSeqs(q, read_len=L, want_valid=True, want_counts=True)
        torch.cuda.synchronize()
        out[mode] = (hit.cpu().numpy(), valid.cpu().numpy(), cnt.cpu().tolist())
    assert (out["direct"][1] == out["partitioned"][1]).all()
    assert (out["direct"][0] == out["partitioned"][0]).all()
    assert out["direct"][2] == out["partitioned"][2]
    assert out["direct"][2][0] == n * 120 - 31
    if miss_reads == 0:
        assert out["partitioned"][2][1] == out["partitioned"][2][0]
    else:
        assert out["partitioned"][2][1] < out["partitioned"][2][0]
    # counts-only call (no bitmaps requested) and auto mode agree too
    flt.setQueryMode("partitioned")
    _, _, cnt = flt.containsSeqs(q, read_len=L, want_valid=False, want_counts=True)
    assert cnt.cpu().tolist() == out["direct"][2]
    # auto mode: 10 % / 50 % / 97 % foreign reads take the split path (reads sampled, warm reads partitioned,
    # cold reads through the gather kernel, bitmaps merged); with and without the valid bitmap, counts only
    flt.setQueryMode("auto")
    flt.setProfiling(True)
    flt.getProfile()
    hit, _, cnt = flt.containsSeqs(q, read_len=L, want_valid=False, want_counts=True)
    assert cnt.cpu().tolist() == out["direct"][2] and (hit.cpu().numpy() == out["direct"][0]).all()
    prof = flt.getProfile()
    if bits == 1 << 30 and 4000 <= miss_reads <= 20000:
        assert "query_hash" in prof and "query_direct" in prof and "query_resolve" in prof, prof  # the split path ran
    hit, valid, cnt = flt.containsSeqs(q, read_len=L, want_valid=True, want_counts=True)
    assert cnt.cpu().tolist() == out["direct"][2]
    assert (hit.cpu().numpy() == out["direct"][0]).all() and (valid.cpu().numpy() == out["direct"][1]).all()
    # a buffer that does not end on a word boundary of the bitmaps
    hit, valid, _ = flt.containsSeqs(q[: (n - 3) * L], read_len=L, want_valid=True, want_counts=False)
    nb = ((n - 3) * L) // 64
    assert (hit.cpu().numpy()[:nb] == out["direct"][0][:nb]).all() and (valid.cpu().numpy()[:nb] == out["direct"][1][:nb]).all()


@pytest.mark.parametrize("L,k,n", [(100, 31, 60000), (151, 25, 40000), (50, 21, 150000), (1100, 31, 8000), (37, 31, 300000)])
def test_split_query_other_read_lengths(bf, L, k, n):
    """AUTO query of a buffer with a third of the reads foreign: the reads are sampled, split into a warm and a cold
    buffer (compact_reads_kernel copies them 4 bytes per lane: L a multiple of 4 or not, reads longer than 1024
    bytes), answered by the partitioned pipeline and the gather kernel, and the bitmaps merged back -- equal to
    the direct kernel on the original buffer."""
    import torch

    h, bits = 4, 1 << 30
    reads = bf.synth_reads_device(42, 0, n, L)
    flt = bf.BloomFilter(bits, h, k)
    flt.insertSeqs(reads, read_len=L)
    q = reads.clone()
    foreign = n // 3
    idx = torch.arange(foreign, device="cuda") * 3 + 1
    q.view(n, L)[idx] = bf.synth_reads_device(43, 0, foreign, L).view(foreign, L)
    q[5 * L + 3] = ord("N")
    flt.setQueryMode("direct")
    hit_d, valid_d, cnt_d = flt.containsSeqs(q, read_len=L, want_valid=True, want_counts=True)
    flt.setQueryMode("auto")
    flt.setProfiling(True)
    flt.getProfile()
    hit_a, valid_a, cnt_a = flt.containsSeqs(q, read_len=L, want_valid=True, want_counts=True)
    torch.cuda.synchronize()
    prof = flt.getProfile()
    assert "query_hash" in prof and "query_direct" in prof and "query_resolve" in prof, prof  # the split path ran
    assert cnt_a.tolist() == cnt_d.tolist()
    assert torch.equal(valid_a, valid_d) and torch.equal(hit_a, hit_d)
    hit_c, _, cnt_c = flt.containsSeqs(q, read_len=L, want_valid=False, want_counts=True)
    assert cnt_c.tolist() == cnt_d.tolist() and torch.equal(hit_c, hit_d)


@pytest.mark.parametrize("L,k", [(150, 31), (100, 31), (151, 25), (50, 33), (250, 33), (152, 21)])
def test_partitioned_many_batches_read_grid_equals_direct(bf, L, k):
    """pass A's read grid (tiles of whole reads) with a scratch cap that forces many batches: batch boundaries
    do not fall on the resolve kernel's tile boundaries, and a later batch's resolve step must not undo an
    earlier one's answers (it did, once: found by tools/fuzz_parity.py); odd L takes the byte-wise staging"""
    import torch

    h, bits, n = 3, 1 << 31, (120000 if L >= 100 else 400000)
    rng = np.random.default_rng(L * 131 + k)
    a = rng.choice(np.frombuffer(b"ACGTacgt", np.uint8), size=(n, L))
    a[rng.random((n, L)) < 0.0005] = ord("N")
    flat = torch.from_numpy(a.reshape(-1).copy()).cuda()
    q = flat.clone()
    idx = np.flatnonzero(rng.random(n) < 0.05)  # 5 % foreign reads: failures in every batch, within the fail list
    q.view(n, L)[torch.from_numpy(idx).cuda()] = torch.from_numpy(
        rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(idx.size, L))).cuda()
    res = []
    for mode in ("direct", "partitioned"):
        f = bf.BloomFilter(bits, h, k)
        f.setInsertMode(mode, scratch_bytes=(64 << 20) if mode == "partitioned" else 0)
        f.setQueryMode(mode)
        f.setProfiling(True)
        f.insertSeqs(flat, read_len=L)
        hit, valid, cnt = f.containsSeqs(q, read_len=L, want_counts=True)
        torch.cuda.synchronize()
        if mode == "partitioned":
            prof = f.getProfile()
            assert prof["insert_hash"][1] >= 2 and prof["query_hash"][1] >= 2  # several batches each
        res.append((f, hit.cpu().numpy(), valid.cpu().numpy(), cnt.tolist()))
    assert res[1][0].compare(res[0][0]) == (0, 0, 0)
    assert (res[0][2] == res[1][2]).all() and (res[0][1] == res[1][1]).all() and res[0][3] == res[1][3]
    assert 0 < res[0][3][1] < res[0][3][0]


@pytest.mark.parametrize("L,k,mis", [(150, 31, 1), (150, 31, 2), (151, 31, 0), (151, 21, 3), (64, 31, 5), (100, 64, 6)])
def test_read_grid_misaligned_buffers_and_odd_read_lengths(bf, L, k, mis):
    """the read grid's staging scatters bytes pairwise only for even L on a 4-byte aligned buffer; every other
    case goes byte by byte -- partitioned == direct (bodies, hit and valid bitmaps) on misaligned device pointers"""
    import torch

    h, bits, n = 4, 1 << 30, 70000
    rng = np.random.default_rng(L + 7 * k + mis)
    a = rng.choice(np.frombuffer(b"ACGTacgt", np.uint8), size=(n, L))
    a[rng.random((n, L)) < 0.001] = ord("N")
    t = torch.zeros(n * L + 16, dtype=torch.uint8, device="cuda")
    t[mis:mis + n * L] = torch.from_numpy(a.reshape(-1).copy()).cuda()
    buf = t[mis:mis + n * L]
    q = torch.zeros(n * L + 16, dtype=torch.uint8, device="cuda")
    q[mis:mis + n * L] = buf
    qv = q[mis:mis + n * L]
    idx = np.flatnonzero(rng.random(n) < 0.02)
    qv.view(n, L)[torch.from_numpy(idx).cuda()] = torch.from_numpy(
        rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(idx.size, L))).cuda()
    res = []
    for mode in ("direct", "partitioned"):
        f = bf.BloomFilter(bits, h, k)
        f.setInsertMode(mode)
        f.setQueryMode(mode)
        f.insertSeqs(buf, read_len=L)
        hit, valid, cnt = f.containsSeqs(qv, read_len=L, want_counts=True)
        torch.cuda.synchronize()
        res.append((f, hit.cpu().numpy(), valid.cpu().numpy(), cnt.tolist()))
    assert res[1][0].compare(res[0][0]) == (0, 0, 0)
    assert (res[0][2] == res[1][2]).all() and (res[0][1] == res[1][1]).all() and res[0][3] == res[1][3]
    assert 0 < res[0][3][1] < res[0][3][0]


# ---------------------------------------------------------------------------------------------
# btlbf_clear is lazy (and a new filter is a cleared filter): the first partitioned insert builds every
# segment from zero in LDS; every other entry point must see the zeros
# ---------------------------------------------------------------------------------------------
def test_lazy_clear_is_invisible(bf, tmp_path):
    import torch

    bits, h, k, L = 1 << 30, 4, 31, 150
    reads = bf.synth_reads_device(42, 0, 60000, L)
    skew = reads[:L].repeat(4000)  # overflows its bins: travels through the fresh batch's spill list
    buf = torch.cat([reads, skew, reads[: 1000 * L]])
    ref = bf.BloomFilter(bits, h, k)
    ref.setInsertMode("direct")
    ref.insertSeqs(buf, read_len=L)
    want = hashlib.sha256(ref.download()).hexdigest()
    # 1. a NEW filter: its first partitioned insert is a fresh one (several batches: only the first is)
    a = bf.BloomFilter(bits, h, k)
    a.setInsertMode("partitioned", scratch_bytes=192 << 20)
    a.setProfiling(True)
    a.insertSeqs(buf, read_len=L)
    assert a.getProfile()["insert_hash"][1] >= 2
    assert hashlib.sha256(a.download()).hexdigest() == want
    # 2. clear, then everything that looks at the array sees zeros
    a.clear()
    assert a.getPop() == 0
    a.insertSeqs(reads[: 10 * L], read_len=L)  # small batch with AUTO off -> partitioned, fresh
    p1 = a.getPop()
    a.clear()
    assert not a.download().any()
    a.clear()
    a.setInsertMode("direct")
    a.insertSeqs(reads[: 10 * L], read_len=L)
    assert a.getPop() == p1
    a.clear()
    _, _, cnt = a.containsSeqs(reads, read_len=L, want_valid=False, want_counts=True)
    assert cnt.tolist() == [60000 * 120, 0]
    a.clear()
    a.storeFilter(tmp_path / "empty.bf")
    raw = open(tmp_path / "empty.bf", "rb").read()
    assert raw.endswith(b"\0" * (bits // 8)) and len(raw) > bits // 8
    a.clear()
    hv, valid = bf.hash_seqs(reads[:L], h, k, read_len=L)
    rows = hv.cpu().numpy().view(np.uint64)[: L - k + 1]
    assert not a.contains(rows).any()
    a.insert(rows)
    assert a.contains(rows).all() and a.compare(ref)[1] == 0  # nothing set that the full filter lacks
    # 3. clear + partitioned again: the cleared state, not the old bits, is what the fresh batch starts from
    a.clear()
    a.setInsertMode("partitioned")
    a.insertSeqs(buf, read_len=L)
    assert hashlib.sha256(a.download()).hexdigest() == want
    a.insertSeqs(buf, read_len=L)  # not fresh any more: ORs into what is there
    assert hashlib.sha256(a.download()).hexdigest() == want
    # 4. counting filter: incrementAll through a fresh batch, saturation included
    c1, c2 = bf.CountingBloomFilter(1 << 27, 3, 25, 2), bf.CountingBloomFilter(1 << 27, 3, 25, 2)
    sat = reads[: 3 * L].repeat(300)
    cb = torch.cat([reads, sat])
    c1.setInsertMode("direct")
    c2.setInsertMode("partitioned")
    for c in (c1, c2):
        c.insertSeqs(cb, read_len=L, increment_all=True)
        c.clear()
        c.insertSeqs(cb, read_len=L, increment_all=True)
    assert c2.compare(c1) == (0, 0, 0) and c1.download().max() == 255


@pytest.mark.parametrize("kind", ["nonpow2", "shard"])
def test_fresh_insert_spill_overflow_is_redone(bf, kind):
    """the first batch into a cleared filter builds its segments from zero and reports what passes A / B could not
    stage as explicit positions; MORE of those than the list holds (here: a short list, because the scratch budget is
    below 2 GiB, and reads that put millions of probes on a few positions) and the batch is redone the ordinary way.
    Non-power-of-two size and a shard (pass A's WINDOW variant): bodies against the direct kernels."""
    import torch

    h, k, L = 4, 31, 150
    reads = bf.synth_reads_device(42, 0, 30000, L)
    units = [b"A", b"C", b"AC", b"AG", b"ACG", b"AAT", b"ACGT", b"AACC"]
    rep = torch.cat([torch.tensor(list((u * L)[:L]), dtype=torch.uint8, device=reads.device).repeat(6000) for u in units])
    buf = torch.cat([reads, rep])
    if kind == "nonpow2":
        mk = lambda: bf.BloomFilter(3 * (1 << 29) + 64 * 7, h, k)  # noqa: E731
    else:
        mk = lambda: bf.BloomFilter.shard(1 << 32, 1, 4, h, k)  # noqa: E731
    a = mk()
    a.setInsertMode("direct")
    a.insertSeqs(buf, read_len=L)
    b = mk()
    b.setInsertMode("partitioned", scratch_bytes=1 << 30)
    b.setProfiling(True)
    b.insertSeqs(buf, read_len=L)  # a new filter: the first batch is a fresh one
    prof = b.getProfile()
    assert prof["insert_hash"][1] >= 2  # the fresh attempt and its ordinary redo (or several batches)
    assert a.compare(b) == (0, 0, 0) and a.getPop() == b.getPop() > 0
    # and once more into the cleared filter, with room for the list this time
    b.clear()
    b.setInsertMode("partitioned")
    b.insertSeqs(buf, read_len=L)
    assert a.compare(b) == (0, 0, 0)


# ---------------------------------------------------------------------------------------------
# the SWIG module's surface (swig/BloomFilter.i): KmerBloomFilter + insertSeq from Python
# ---------------------------------------------------------------------------------------------
def test_swig_surface_kmer_bloom_filter(bf, oracle, tmp_path):
    rng = np.random.RandomState(5)
    k, h, bits = 25, 3, 1 << 16
    seq = rand_seq(rng, 400, 0.02)
    f = bf.KmerBloomFilter(bits, h, k)
    bf.insertSeq(f, seq, h, k)
    mine = np.zeros(bits // 8, np.uint8)
    oracle.bf_insert_seq(mine, bits, h, k, seq)
    assert (f.download() == mine).all()
    pos, hv = oracle.nthash_seq(seq, h, k)
    # contains(kmer string) == contains(hash row) == the iterator's view for k-mers of ACGT/acgt (k % 4 == 1 here;
    # with U, the raw bytes the iterator accepts, or k % 4 == 0 the reference's two paths differ -- see the
    # test_kmer_path_* tests below)
    plain = [int(p) for p in pos if set(seq[int(p):int(p) + k]) <= set(b"ACGTacgt")]
    assert len(plain) > 20
    for p in plain[:5] + [plain[-1]]:
        kmer = seq[p:p + k]
        assert f.contains(kmer) and f.contains(kmer.decode())
    assert f.contains(hv[0].tolist()) is True
    other = bf.KmerBloomFilter(bits, h, k)
    i0, i1 = plain[0], plain[1]
    r0, r1 = int(np.searchsorted(pos, i0)), int(np.searchsorted(pos, i1))
    assert not other.contains(seq[i0:i0 + k])
    other.insert(seq[i0:i0 + k])                          # insert(const char* kmer)
    other.insert(hv[r1].tolist())                         # insert(vector<uint64_t>)
    assert other.contains(hv[r0].tolist()) and other.contains(seq[i1:i1 + k])
    assert other.getPop() <= 2 * h and other.getHashNum() == h and other.getKmerSize() == k
    assert other.getFilterSize() == bits
    path = str(tmp_path / "swig.bf")
    f.storeFilter(path)
    g = bf.KmerBloomFilter(path=path)
    assert g.getPop() == f.getPop() and (g.download() == mine).all()
    with pytest.raises(ValueError):
        bf.insertSeq(f, seq, h + 1, k)


# ---------------------------------------------------------------------------------------------
# the raw-k-mer path: KmerBloomFilter::insert / contains(const char*) = NTC64(kmerSeq, k) + NTE64
# ---------------------------------------------------------------------------------------------
def test_kmer_path_golden_hashes(bf, oracle):
    """every pinned k-mer case -- including k % 4 == 0 ("ub") and U ("u"), where the reference's raw-k-mer
    path disagrees with its own iterator -- comes out of the HIP kernel with the reference's x86-64 value"""
    n = 0
    for case in load_golden("hash_vectors.json")["kmer"]:
        hv, ok = bf.hash_kmers(case["kmer"].encode("latin-1"), case["h"], case["k"])
        assert ok[0] == 1 and (hv[0] == unhex(case["hashes"], 0)).all(), case
        n += 1
    assert n == 37
    n_def = 0
    for case in load_golden("kmer_path.json")["hashes"]:
        hv, ok = bf.hash_kmers(case["kmer"].encode("latin-1"), case["h"], case["k"])
        assert ok[0] == case["defined"], case
        if case["defined"]:
            assert (hv[0] == unhex(case["hashes"], 0)).all(), case
            n_def += 1
        else:
            assert not hv.any()
    assert n_def == 173
    # batches against the oracle: many k-mers per call, every k from 1 to 70 and a few above 256
    rng = np.random.RandomState(8)
    alpha = list(b"ACGT" * 5 + b"acgtUuNn\x01\x03-*")
    for k in list(range(1, 71)) + [255, 256, 257, 300]:
        kms = bytes(rng.choice(alpha, k * 257).tolist())
        hv, ok = bf.hash_kmers(kms, 3, k)
        ehv, eok = oracle.kmer_hashes(kms, k, 3)
        assert (ok == eok).all() and (hv == ehv).all(), k


def test_kmer_path_swig_test_pl_replay(bf, tmp_path):
    """swig/test.pl:8-46 through the drop-in: BloomFilter(1000000000, 5, 20), four insert(kmer) calls (k % 4 == 0),
    contains(kmer) of six k-mers, storeFilter -> the reference's file byte for byte (header, set bits, SHA-256 of
    the 125 MB), load it again and ask the same questions"""
    t = load_golden("kmer_path.json")["swig_test_pl"]
    f = bf.KmerBloomFilter(t["bits"], t["h"], t["k"])
    for km in t["inserted"]:
        f.insert(km)
    assert [int(f.contains(km)) for km in t["queried"]] == t["contains"]
    assert f.containsKmers(t["queried"]).tolist() == t["contains"]
    assert f.getPop() == t["pop"]
    body = f.download()
    assert sorted(int(8 * i + b) for i in np.flatnonzero(body) for b in range(8) if (body[i] >> b) & 1) == t["set_bits"]
    path = str(tmp_path / "BloomFilter.bf")
    f.storeFilter(path)
    raw = open(path, "rb").read()
    assert raw[: len(t["header"])] == t["header"].encode()
    assert hashlib.sha256(raw).hexdigest() == t["file_sha256"]
    g = bf.KmerBloomFilter(path=path)
    assert [int(g.contains(km)) for km in t["queried"]] == t["contains"]
    assert (g.getPop(), g.getHashNum(), g.getKmerSize(), g.getFilterSize()) == (t["pop"], t["h"], t["k"], t["bits"])
    # the iterator path puts the same four k-mers elsewhere when k % 4 == 0 -- as it does in the reference
    it = bf.KmerBloomFilter(t["bits"], t["h"], t["k"])
    for km in t["inserted"]:
        bf.insertSeq(it, km)
    assert it.compare(f)[0] > 0 and not t["same_as_iterator_path"]
    # swig/test.pl:59-84: insertSeq, then contains(kmer) of every 5-mer
    s = load_golden("kmer_path.json")["swig_insert_seq"]
    b = bf.KmerBloomFilter(s["bits"], s["h"], s["k"])
    bf.insertSeq(b, s["seq"], s["h"], s["k"])
    assert hashlib.sha256(b.download().tobytes()).hexdigest() == s["body_sha256"]
    assert [int(b.contains(s["seq"][i:i + s["k"]])) for i in range(len(s["seq"]) - s["k"] + 1)] == s["contains"]


def test_kmer_path_batches_with_undefined_kmers(bf, oracle):
    """batch calls; k-mers without a defined value in the reference are skipped (nothing inserted, contains 0)"""
    rng = np.random.RandomState(9)
    k, h, bits = 22, 3, 1 << 16
    kms = bytes(rng.choice(list(b"ACGT" * 6 + b"UN"), k * 4000).tolist())
    ehv, eok = oracle.kmer_hashes(kms, k, h)
    assert 0 < eok.sum() < len(eok)
    f = bf.KmerBloomFilter(bits, h, k)
    f.insertKmers(np.frombuffer(kms, np.uint8))
    mine = np.zeros(bits // 8, np.uint8)
    oracle.bf_insert(mine, bits, h, ehv[eok == 1])
    assert (f.download() == mine).all()
    q = bytes(rng.choice(list(b"ACGT" * 6 + b"UN"), k * 3000).tolist()) + kms[: 500 * k]
    qhv, qok = oracle.kmer_hashes(q, k, h)
    exp = oracle.bf_contains(mine, bits, h, qhv) * qok
    assert (f.containsKmers(np.frombuffer(q, np.uint8)) == exp).all() and exp[-500:].sum() == eok[:500].sum()


# ---------------------------------------------------------------------------------------------
# counting filter through the partitioned pipeline: incrementAll (exact, saturating) and contains()
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nbytes,k,h", [(1 << 27, 25, 3), (5 << 24, 31, 4), (1 << 30, 25, 3)])
def test_partitioned_counting_increment_all_and_query(bf, nbytes, k, h):
    import torch

    L, thr = 150, 2
    reads = bf.synth_reads_device(21, 0, 60000, L)
    sat = reads[: 3 * L].repeat(300)  # 300 copies of three reads: their counters saturate at 255
    buf = torch.cat([reads, reads[: 20000 * L], sat])
    a, b = bf.CountingBloomFilter(nbytes, h, k, thr), bf.CountingBloomFilter(nbytes, h, k, thr)
    for f, mode in ((a, "direct"), (b, "partitioned")):
        f.setInsertMode(mode)
        f.setQueryMode(mode)
        f.insertSeqs(buf, read_len=L, increment_all=True)
    torch.cuda.synchronize()
    ca, cb = a.download(), b.download()
    assert ca.max() == 255 and (ca == cb).all()
    assert a.popCount() == b.popCount() and a.filtered_popcount() == b.filtered_popcount()
    # reads inserted twice pass the threshold, reads inserted once mostly do not, foreign reads do not
    q = torch.cat([reads[: 25000 * L], bf.synth_reads_device(22, 0, 50, L)])
    ha, va, na = a.containsSeqs(q, read_len=L, want_counts=True)
    hb, vb, nb = b.containsSeqs(q, read_len=L, want_counts=True)
    torch.cuda.synchronize()
    assert na.tolist() == nb.tolist() and bool((ha == hb).all().item()) and bool((va == vb).all().item())
    assert 20000 * (L - k + 1) <= int(na[1]) < int(na[0])
    # the conservative update (`insert`) is not partitioned: it must still take the direct kernel
    c = bf.CountingBloomFilter(nbytes, h, k, thr)
    c.setInsertMode("partitioned")
    c.insertSeqs(reads, read_len=L)
    assert 0 < c.popCount() <= a.popCount()


# ---------------------------------------------------------------------------------------------
# per-read totals of the contains() bitmaps (what read classifiers built on the reference compute)
# ---------------------------------------------------------------------------------------------
def test_count_per_seq_matches_bitmaps(bf, oracle):
    import torch

    rng = np.random.RandomState(8)
    k, h, bits = 21, 3, 1 << 20
    f = bf.BloomFilter(bits, h, k)
    # ragged, host
    lens = [0, 5, 20, 21, 22, 150, 64, 63, 65, 1000, 0, 21]
    reads = [rand_seq(rng, n, 0.02) for n in lens]
    buf = b"".join(reads)
    starts = np.cumsum([0] + lens).astype(np.uint64)
    f.insertSeqs(np.frombuffer(b"".join(reads[::2]), np.uint8).copy(),
                 starts=np.cumsum([0] + lens[::2]).astype(np.uint64))
    hit, valid, _ = f.containsSeqs(np.frombuffer(buf, np.uint8).copy(), starts=starts)
    hits, clean = bf.count_per_seq(hit, valid, len(buf), k, starts=starts)
    hb, vb = bf.bits_to_bool(hit, len(buf)), bf.bits_to_bool(valid, len(buf))
    for i, (a, b) in enumerate(zip(starts[:-1], starts[1:])):
        a, b = int(a), int(b)
        hi = max(b - k + 1, a)
        assert hits[i] == hb[a:hi].sum() and clean[i] == vb[a:hi].sum(), i
        assert clean[i] == len(oracle.nthash_seq(reads[i], h, k)[0])
    assert all(hits[i] == clean[i] for i in range(0, len(lens), 2))  # the inserted reads hit everywhere
    # uniform, device, no valid bitmap: clean = all windows
    L, n = 150, 5000
    dev = bf.synth_reads_device(5, 0, n, L)
    f.insertSeqs(dev[: 1000 * L], read_len=L)
    hit, valid, cnt = f.containsSeqs(dev, read_len=L, want_counts=True)
    hits, clean = bf.count_per_seq(hit, None, dev.numel(), k, read_len=L)
    torch.cuda.synchronize()
    assert int(hits.sum().item()) == int(cnt[1].item()) and bool((clean == L - k + 1).all().item())
    assert bool((hits[:1000] == L - k + 1).all().item()) and int(hits[1000:].max().item()) < L - k + 1


def test_out_of_memory_mid_pass_gives_right_bytes_or_an_error_never_wrong_bytes(bf):
    """HBM runs out while a partitioned pass wants its scratch (somebody else filled the card): the library must either
    carry the operation out another way with the right result (the direct kernels need no scratch) or return an
    error -- and stay usable afterwards.  Also exercised: a query that wants counts only needs a temporary hit bitmap
    (run_query_like): no room for it is an error code, not a crash or a wrong count."""
    import torch

    from btl_bloomfilter_amd._lib import BtlbfError

    bits, h, k, L, n = 1 << 33, 4, 31, 150, 2_000_000
    reads = bf.synth_reads_device(42, 0, n, L)
    a, b = bf.BloomFilter(bits, h, k), bf.BloomFilter(bits, h, k)
    a.setInsertMode("direct")
    a.insertSeqs(reads, read_len=L)
    want = a.digest()
    b.setInsertMode("partitioned")
    b.setQueryMode("partitioned")
    b.setProfiling(True)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    hog = torch.empty(free - (96 << 20), dtype=torch.uint8, device="cuda")  # leaves far less than the scratch needs
    try:
        outcome = "ok"
        try:
            b.insertSeqs(reads, read_len=L)
            torch.cuda.synchronize()
        except BtlbfError as exc:
            outcome = "error: %s" % exc
        if outcome == "ok":
            prof = b.getProfile()
            assert "insert_hash" not in prof and prof.get("insert_direct", (0, 0))[1] >= 1, prof
            assert b.digest() == want
        try:
            _, _, cnt = b.containsSeqs(reads, read_len=L, want_valid=False, want_counts=True)
            torch.cuda.synchronize()
            if outcome == "ok":
                assert cnt.tolist() == [n * (L - k + 1)] * 2
        except BtlbfError as exc:
            assert "memory" in str(exc).lower() or "hipMalloc" in str(exc) or "alloc" in str(exc), str(exc)
    finally:
        del hog
        torch.cuda.empty_cache()
    # with the memory back the same objects work as if nothing had happened
    b.clear()
    b.insertSeqs(reads, read_len=L)
    torch.cuda.synchronize()
    prof = b.getProfile()
    assert prof.get("insert_hash", (0, 0))[1] >= 1, prof
    assert b.digest() == want
    _, _, cnt = b.containsSeqs(reads, read_len=L, want_valid=False, want_counts=True)
    assert cnt.tolist() == [n * (L - k + 1)] * 2


@pytest.mark.parametrize("shape", ["reads_100_200", "tiny_sequences", "tile_edges", "mixed_long"])
def test_partitioned_ragged_layout_start_bitmap(bf, shape):
    """Ragged buffers (btlbf_layout::starts -- what the reference's FASTA loop hands over sequence by sequence,
    Tests/AdHoc/ParallelFilter.cpp:104-122) through pass A's overlapped schedule, which marks the sequence starts of the
    next tile in an LDS bitmap while the current tile is partitioned: many tiles per workgroup, more starts in a tile
    than staging threads (sequences of 1..12 bases, empty ones), starts exactly on and next to tile boundaries
    (tiles of 8192 window starts), long and short sequences mixed, an unaligned base pointer.  Bodies, hit and valid
    bitmaps must equal the direct kernels'."""
    import torch

    bits, h, k = 1 << 32, 4, 31
    rng = np.random.RandomState({"reads_100_200": 1, "tiny_sequences": 2, "tile_edges": 3, "mixed_long": 4}[shape])
    total = 24_000_000  # ~ 2900 tiles: a dozen per workgroup
    if shape == "reads_100_200":
        lens = rng.randint(100, 201, total // 150 + 1000)
    elif shape == "tiny_sequences":
        total = 6_000_000
        lens = np.concatenate([rng.randint(0, 13, 400_000), rng.randint(40, 300, 20_000)])
        rng.shuffle(lens)
    elif shape == "tile_edges":
        # sequence ends placed on multiples of 8192 and one or two bases either side of them, k-1 bases before them ...
        lens = []
        pos = 0
        for i in range(1, total // 8192):
            target = i * 8192 + int(rng.choice([-31, -30, -2, -1, 0, 1, 2, 29, 30, 31]))
            if target - pos > 0:
                mid = pos + (target - pos) // 2
                lens += [mid - pos, target - mid]
                pos = target
        lens = np.array(lens)
    else:
        lens = np.concatenate([rng.randint(31, 120, 100_000), rng.randint(100_000, 900_000, 20), [0, 0, 30, 31, 1]])
        rng.shuffle(lens)
    starts = np.concatenate([[0], np.cumsum(lens)])
    starts = starts[starts <= total]
    if starts[-1] != total:
        starts = np.concatenate([starts, [total]])
    store = torch.zeros(total + 8, dtype=torch.uint8, device="cuda")
    store[3:3 + total] = bf.synth_reads_device(7, 0, total // 100 + 1, 100)[:total]
    seq = store[3:3 + total]  # a base pointer that is no multiple of four
    seq[12345] = ord("N")
    ts = torch.from_numpy(starts.astype(np.int64)).cuda()
    a, b = bf.BloomFilter(bits, h, k), bf.BloomFilter(bits, h, k)
    a.setInsertMode("direct")
    b.setInsertMode("partitioned")
    b.setProfiling(True)
    a.insertSeqs(seq, starts=ts)
    b.insertSeqs(seq, starts=ts)
    torch.cuda.synchronize()
    prof = b.getProfile()
    assert prof.get("insert_hash", (0, 0))[1] >= 1 and "insert_direct" not in prof, prof
    assert b.compare(a) == (0, 0, 0) and a.getPop() == b.getPop() > 0
    # query: the same buffer with some foreign stretches; both modes must answer alike, window by window
    q = seq.clone()
    q[1_000_000:1_020_000] = bf.synth_reads_device(9, 0, 200, 100)
    res = {}
    for mode in ("direct", "partitioned"):
        b.setQueryMode(mode)
        hit, valid, cnt = b.containsSeqs(q, starts=ts, want_valid=True, want_counts=True)
        torch.cuda.synchronize()
        res[mode] = (hit, valid, cnt.tolist())
    assert res["direct"][2] == res["partitioned"][2]
    assert bool(torch.equal(res["direct"][1], res["partitioned"][1])), "valid bitmaps differ"
    assert bool(torch.equal(res["direct"][0], res["partitioned"][0])), "hit bitmaps differ"
    want_clean = int(np.clip(np.diff(starts) - (k - 1), 0, None).sum())
    assert abs(res["direct"][2][0] - want_clean) <= 2 * k  # (the N and the foreign stretch's seams)


@pytest.mark.parametrize("p_raw", [0.0, 0.0005, 0.2])
@pytest.mark.parametrize("bits", [1 << 26, 1 << 27, 3 << 25])
def test_partitioned_spaced_seeds_two_base_rows_and_raw_bytes(bf, oracle, bits, p_raw):
    """Config 5's seeds on filters of at most 256 level-0 bins, where pass A has the LDS for the two-base rows of the
    union list's pairs (seq_core.hpp: one table read per pair and window; A C G T only).  Reads with the raw bytes the
    reference's seed table also takes (1 3 4 5 7) -- rare, so that most groups of windows take the rows and a few are
    done again the one-offset way, and dense -- and lower-case / U / N: against the direct kernels (per-seed walk of the
    single-base table) and, for the first reads, against the oracle's stHashIterator walk."""
    import torch

    seeds = ["1110111011101110111011101110111", "1101101101101101011011011011011",
             "1111001111001111111001111001111", "1011101011101011101011101011101"]
    k, L, n = 31, 150, 40000
    rng = np.random.RandomState(int(bits % 977) + int(p_raw * 1e4))
    s = rng.choice(list(b"ACGTacgtU"), n * L).astype(np.uint8)
    raw = rng.rand(n * L) < p_raw
    s[raw] = rng.choice(list(b"\x01\x03\x04\x05\x07"), int(raw.sum()))
    s[rng.rand(n * L) < 0.0005] = ord("N")
    reads = torch.from_numpy(s).cuda()
    a, b = bf.BloomFilter(bits, 4, k), bf.BloomFilter(bits, 4, k)
    for f, mode in ((a, "direct"), (b, "partitioned")):
        f.setSpacedSeeds(seeds, 1)
        f.setInsertMode(mode)
        f.setQueryMode(mode)
        f.insertSeqs(reads, read_len=L)
    torch.cuda.synchronize()
    assert a.getPop() == b.getPop() > 0
    assert (a.download() == b.download()).all()
    q = torch.cat([reads[: 5000 * L], bf.synth_reads_device(12, 0, 300, L)])
    ha, va, ca = a.containsSeqs(q, read_len=L, want_counts=True)
    hb, vb, cb = b.containsSeqs(q, read_len=L, want_counts=True)
    assert ca.tolist() == cb.tolist() and torch.equal(ha, hb) and torch.equal(va, vb)
    body = b.download()
    host = s.tobytes()
    for r in range(12):
        pos, hv, _ = oracle.sthash_seq(host[r * L: (r + 1) * L], seeds, 1, k)
        pp = (hv % np.uint64(bits)).ravel().astype(np.int64)
        assert ((body[pp >> 3] >> (pp & 7)) & 1).all(), "read %d" % r


@pytest.mark.parametrize("n_seeds,h2,k,L", [(6, 1, 31, 150), (8, 1, 25, 100), (4, 2, 47, 151), (2, 1, 96, 250), (3, 1, 8, 60)])
def test_partitioned_spaced_seed_union_list_shapes(bf, oracle, n_seeds, h2, k, L):
    """Random spaced seeds through pass A's union list of don't-care offsets (seq_core.hpp): more than four hashes per
    k-mer (rounds and groups of two windows), h2 > 1 (the seeds' values picked with a select chain), every mask class of
    odd size (the zero-row filler), offsets every seed leaves out, a seed without don't-cares, and -- k = 96 -- more than
    64 distinct offsets, which pass A does not take (the direct kernels answer, also in "partitioned" mode).  Against
    the direct kernels and, for a slice of the reads, against the oracle's stHashIterator walk (vendor/stHashIterator.hpp:53-104)."""
    import torch

    rng = np.random.RandomState(100 * n_seeds + h2 + k)
    seeds = []
    for j in range(n_seeds):
        s = (rng.random_sample(k) < (0.35 if k >= 96 else 0.75)).astype(int)
        s[k // 2] = 0            # left out by every seed: folded into the common base
        if j == 1:
            s[:] = 1             # a seed that cares about every position
            s[k // 2] = 0
        seeds.append("".join("1" if x else "0" for x in s))
    bits, h = 1 << 29, n_seeds * h2
    reads = bf.synth_reads_device(11, 0, 30000, L)
    reads[700:705] = ord("N")
    a, b = bf.BloomFilter(bits, h, k), bf.BloomFilter(bits, h, k)
    for f, mode in ((a, "direct"), (b, "partitioned")):
        f.setSpacedSeeds(seeds, h2)
        f.setInsertMode(mode)
        f.setQueryMode(mode)
        f.insertSeqs(reads, read_len=L)
    torch.cuda.synchronize()
    assert a.getPop() == b.getPop() > 0
    assert (a.download() == b.download()).all()
    q = torch.cat([reads[: 4000 * L], bf.synth_reads_device(12, 0, 30, L)])
    ha, _, ca = a.containsSeqs(q, read_len=L, want_valid=False, want_counts=True)
    hb, _, cb = b.containsSeqs(q, read_len=L, want_valid=False, want_counts=True)
    torch.cuda.synchronize()
    assert ca.tolist() == cb.tolist() and bool((ha == hb).all().item())
    # the bits the oracle's iterator sets for the first reads are all set
    body = b.download()
    host = bytes(reads[: 20 * L].cpu().numpy())
    for r in range(20):
        pos, hv, _ = oracle.sthash_seq(host[r * L: (r + 1) * L], seeds, h2, k)
        p = (hv % np.uint64(bits)).ravel().astype(np.int64)
        assert ((body[p >> 3] >> (p & 7)) & 1).all(), "read %d" % r

"""The CPU restatement (oracle/btl_oracle.c) against the golden vectors emitted by the genuine
reference (tests/golden/make_golden.py).  CPU only; this is what pins the oracle on machines
without /root/reference."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden


def unhex(lst, cols):
    a = np.array([int(x, 16) for x in lst], dtype=np.uint64)
    return a.reshape(-1, cols) if cols else a


def test_g1_nthash_vectors(oracle):
    g = load_golden("hash_vectors.json")["nthash"]
    assert len(g) > 100
    nk = 0
    for case in g:
        pos, hv = oracle.nthash_seq(case["seq"].encode("latin-1"), case["h"], case["k"])
        assert pos.tolist() == case["pos"], (case["seq"], case["k"])
        assert (hv == unhex(case["hashes"], case["h"])).all()
        nk += len(pos)
    assert nk > 10000


def test_g1_known_answer_from_survey(oracle):
    # "ACGTAC", k=4, h=5
    pos, hv = oracle.nthash_seq(b"ACGTAC", 5, 4)
    assert pos.tolist() == [0, 1, 2]
    assert ["%016x" % x for x in hv[0]] == ["4b21efd76bfc8c8a", "6ab8d13c740e89be", "b5dac11e34491d35",
                                            "00fcb0dfe49bd351", "4c1ea0bee4de43d4"]
    assert "%016x" % hv[1][0] == "62779f381e5f5a2d"
    assert "%016x" % hv[2][0] == "ec40e7b3741c2bdd"


def test_g2_sthash_vectors(oracle):
    g = load_golden("hash_vectors.json")["sthash"]
    for case in g:
        m = len(case["seeds"]) * case["h2"]
        pos, hv, st = oracle.sthash_seq(case["seq"].encode("latin-1"), case["seeds"], case["h2"], case["k"])
        assert pos.tolist() == case["pos"]
        assert (hv == unhex(case["hashes"], m)).all()
        assert st.ravel().tolist() == case["strand"]


def _kmer_bytes(km):
    return km.encode("latin-1")


def test_kmer_path_hash_vectors(oracle):
    """KmerBloomFilter::insert/contains(const char*) hashes a raw k-mer with NTC64(kmerSeq, k)
    (KmerBloomFilter.hpp:47-74, vendor/nthash.hpp:394-439,460-465).  The restatement returns the reference's
    x86-64 values in EVERY pinned case, including the two where that path disagrees with the reference's own
    iterator: k % 4 == 0 (flag "ub": the table walk's shift by 64) and U read as A (flag "u")."""
    n = n_ub = n_u = 0
    for case in load_golden("hash_vectors.json")["kmer"]:
        hv, ok = oracle.kmer_hashes(_kmer_bytes(case["kmer"]), case["k"], case["h"])
        assert ok[0] == 1 and (hv[0] == unhex(case["hashes"], 0)).all(), case
        n += 1
        n_ub += bool(case["ub"])
        n_u += bool(case.get("u"))
        # ... and where the two paths of the reference agree, so do the two restatements
        pos, it = oracle.nthash_seq(_kmer_bytes(case["kmer"]), case["h"], case["k"])
        assert ((it[0] == hv[0]).all()) == (not case["ub"] and not case.get("u")), case
    assert n == 37 and n_ub == 9 and n_u == 10
    g = load_golden("kmer_path.json")["hashes"]
    n_def = 0
    for case in g:
        hv, ok = oracle.kmer_hashes(_kmer_bytes(case["kmer"]), case["k"], case["h"])
        assert ok[0] == case["defined"], case
        if case["defined"]:
            assert (hv[0] == unhex(case["hashes"], 0)).all(), case
            n_def += 1
    assert len(g) == 188 and n_def == 173


def test_kmer_path_swig_test_pl_replay(oracle):
    """swig/test.pl:8-26 builds BloomFilter(1000000000, 5, 20) from four raw k-mers (k % 4 == 0) and stores it:
    the restatement reproduces the file the reference writes (header, set bits, SHA-256 of all 125 MB)."""
    t = load_golden("kmer_path.json")["swig_test_pl"]
    filt = np.zeros(t["bits"] // 8, np.uint8)
    hv, ok = oracle.kmer_hashes(_kmer_bytes("".join(t["inserted"])), t["k"], t["h"])
    assert ok.all()
    oracle.bf_insert(filt, t["bits"], t["h"], hv)
    assert sorted(int(8 * i + b) for i in np.flatnonzero(filt) for b in range(8) if (filt[i] >> b) & 1) == t["set_bits"]
    assert oracle.bf_popcount(filt, t["bits"]) == t["pop"] == 20 and not t["same_as_iterator_path"]
    header = oracle.bf_header(t["bits"], t["h"], t["k"], 0.0, 0, 0)
    assert header == t["header"].encode()
    assert hashlib.sha256(header + filt.tobytes()).hexdigest() == t["file_sha256"]
    qh, _ = oracle.kmer_hashes(_kmer_bytes("".join(t["queried"])), t["k"], t["h"])
    assert oracle.bf_contains(filt, t["bits"], t["h"], qh).tolist() == t["contains"]
    s = load_golden("kmer_path.json")["swig_insert_seq"]
    filt = np.zeros(s["bits"] // 8, np.uint8)
    oracle.bf_insert_seq(filt, s["bits"], s["h"], s["k"], s["seq"].encode())
    assert hashlib.sha256(filt.tobytes()).hexdigest() == s["body_sha256"]
    kms = "".join(s["seq"][i:i + s["k"]] for i in range(len(s["seq"]) - s["k"] + 1))
    qh, _ = oracle.kmer_hashes(_kmer_bytes(kms), s["k"], s["h"])
    assert oracle.bf_contains(filt, s["bits"], s["h"], qh).tolist() == s["contains"]


def _body(path):
    raw = open(path, "rb").read()
    i = raw.index(b"[HeaderEnd]\n") + len(b"[HeaderEnd]\n")
    return raw[:i], raw[i:]


def test_g3_bloom_files_bytes(oracle):
    for f in load_golden("files.json"):
        if f["kind"] != "bloom":
            continue
        raw = open(os.path.join(GOLDEN, f["file"]), "rb").read()
        assert hashlib.sha256(raw).hexdigest() == f["sha256"]
        filt = np.zeros(f["bits"] // 8, np.uint8)
        for s in f["inserted"]:
            oracle.bf_insert_seq(filt, f["bits"], f["h"], f["k"], s)
        mine = oracle.bf_header(f["bits"], f["h"], f["k"], 0.0, f["n_entry"], f["t_entry"]) + filt.tobytes()
        assert mine == raw, f["file"]
        assert oracle.bf_popcount(filt, f["bits"]) == f["pop"]


def test_g4_counting_files_bytes(oracle):
    for f in load_golden("files.json"):
        if f["kind"] != "counting":
            continue
        raw = open(os.path.join(GOLDEN, f["file"]), "rb").read()
        nb = oracle.cbf_round_bytes(f["bytes"])
        assert nb == f["size"]
        c = np.zeros(nb, np.uint8)
        allh = []
        for s in f["inserted"]:
            _, hv = oracle.nthash_seq(s, f["h"], f["k"])
            allh.append(hv)
            if f["op"] == "insert":
                oracle.cbf_increment_min(c, f["h"], hv)
            else:
                oracle.cbf_increment_all(c, f["h"], hv)
        assert oracle.cbf_header(nb, nb, f["h"], f["k"]) + c.tobytes() == raw, f["file"]
        mn, ct = oracle.cbf_query(c, f["h"], f["thr"], np.concatenate(allh))
        assert mn.tolist() == f["min_counts"] and ct.tolist() == f["contains"]
        assert oracle.cbf_popcount(c) == f["popcount"]
        assert oracle.cbf_filtered_popcount(c, f["thr"]) == f["filtered_popcount"]
        if "saturated" in f["file"]:
            assert c.max() == 255


def test_g5_contains_bitmasks(oracle):
    ops = load_golden("filter_ops.json")
    for key in ("contains_65536", "contains_100000"):
        g = ops[key]
        filt = np.zeros(g["bits"] // 8, np.uint8)
        for s in g["A"]:
            oracle.bf_insert_seq(filt, g["bits"], g["h"], g["k"], s)
        assert hashlib.sha256(filt.tobytes()).hexdigest() == g["body_sha256"]
        assert oracle.bf_popcount(filt, g["bits"]) == g["pop"]
        for s, r in zip(g["B"], g["result"]):
            hit, valid = oracle.bf_contains_seq_dense(filt, g["bits"], g["h"], g["k"], s)
            assert np.flatnonzero(valid).tolist() == r["pos"]
            assert hit[valid == 1].tolist() == r["hit"]


def test_g6_insert_and_check(oracle):
    g = load_golden("filter_ops.json")["insert_and_check"]
    filt = np.zeros(g["bits"] // 8, np.uint8)
    for s, r in zip(g["stream"], g["result"]):
        _, hv = oracle.nthash_seq(s, g["h"], g["k"])
        assert oracle.bf_insert_and_check(filt, g["bits"], g["h"], hv).tolist() == r
    assert hashlib.sha256(filt.tobytes()).hexdigest() == g["body_sha256"]
    g = load_golden("filter_ops.json")["cbf_insert_and_check"]
    c = np.zeros(g["bytes"], np.uint8)
    for s, r in zip(g["stream"], g["result"]):
        _, hv = oracle.nthash_seq(s, g["h"], g["k"])
        assert oracle.cbf_insert_and_check(c, g["h"], g["thr"], hv).tolist() == r
    assert hashlib.sha256(c.tobytes()).hexdigest() == g["body_sha256"]


def test_g8_modulo(oracle):
    for g in load_golden("filter_ops.json")["modulo"]:
        filt = np.zeros(g["size"] // 8, np.uint8)
        oracle.bf_insert(filt, g["size"], 1, unhex(g["hashes"], 1))
        assert np.flatnonzero(np.unpackbits(filt, bitorder="little")).tolist() == g["set_bits"]
        assert sorted(set(int(x, 16) % g["size"] for x in g["hashes"])) == g["set_bits"]


def test_g7_synth_and_digests(oracle):
    d = load_golden("digests.json")
    s = d["synth"]
    assert oracle.synth_reads(42, 0, 3, 150).tobytes().decode() == s["first3"]
    assert hashlib.sha256(oracle.synth_reads(42, 0, 1000, 150)).hexdigest() == s["sha256_first_1000"]
    assert hashlib.sha256(oracle.synth_reads(43, 12345, 100, 150)).hexdigest() == s["sha256_seed43_from_12345"]
    assert hashlib.sha256(oracle.synth_reads(7, 5, 64, 100)).hexdigest() == s["sha256_len100"]
    for name in ("bf_small", "bf_nonpow2"):
        g = d[name]
        filt = np.zeros(g["bits"] // 8, np.uint8)
        reads = oracle.synth_reads(g["seed"], 0, g["n_reads"], g["read_len"]).reshape(g["n_reads"], -1)
        for r in reads:
            oracle.bf_insert_seq(filt, g["bits"], g["h"], g["k"], r.tobytes())
        assert oracle.bf_popcount(filt, g["bits"]) == g["pop"]
        assert hashlib.sha256(filt.tobytes()).hexdigest() == g["body_sha256"]
    for name in ("cbf_small_min", "cbf_small_all"):
        g = d[name]
        c = np.zeros(g["bytes"], np.uint8)
        reads = oracle.synth_reads(g["seed"], 0, g["n_reads"], g["read_len"]).reshape(g["n_reads"], -1)
        for n in (g["n_reads"], g["n_reads"] // 2):
            for r in reads[:n]:
                _, hv = oracle.nthash_seq(r.tobytes(), g["h"], g["k"])
                (oracle.cbf_increment_min if g["op"] == 0 else oracle.cbf_increment_all)(c, g["h"], hv)
        assert hashlib.sha256(c.tobytes()).hexdigest() == g["body_sha256"]
        assert oracle.cbf_popcount(c) == g["popcount"]
        assert oracle.cbf_filtered_popcount(c, g["thr"]) == g["filtered_popcount"]


def test_reference_unit_test_cases(oracle):
    # Tests/Unit/BloomFilterTests.cpp:69-95 -- 1e9-bit filter, h=5, k=4, "ACGTAC": no false negatives
    bits = 1000000000
    filt = np.zeros(bits // 8, np.uint8)
    oracle.bf_insert_seq(filt, bits, 5, 4, b"ACGTAC")
    hit, valid = oracle.bf_contains_seq_dense(filt, bits, 5, 4, b"ACGTAC")
    assert valid.tolist() == [1, 1, 1] and hit.tolist() == [1, 1, 1]
    # Tests/Unit/CountingBloomFilterTests.cpp:70-107 -- 100001 bytes -> 100008, h=5, k=8, thr=1
    nb = oracle.cbf_round_bytes(100001)
    assert nb == 100008
    c = np.zeros(nb, np.uint8)
    _, hv = oracle.nthash_seq(b"ACGTACACTGGACTGAGTCT", 5, 8)
    oracle.cbf_increment_min(c, 5, hv)
    mn, ct = oracle.cbf_query(c, 5, 1, hv)
    assert ct.all() and len(ct) == 13

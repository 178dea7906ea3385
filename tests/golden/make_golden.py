#!/usr/bin/env python3
"""Generate tests/golden/* from the GENUINE reference (oracle/_ref/libbtlref.so, i.e. the headers
under /root/reference compiled behind oracle/ref_driver.cpp).  Run in the build container only:

    make -C oracle && python tests/golden/make_golden.py

The outputs are DATA (inputs + expected outputs).  They pin oracle/btl_oracle.c on machines that
have no reference tree (the GPU box) and are the known-answer vectors of the GPU parity tests.
Fixture ids follow SURVEY.md section 8c (G1..G8).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.pyoracle import Ref  # noqa: E402

SEEDS_C5 = [
    "1110111011101110111011101110111",
    "1101101101101101011011011011011",
    "1111001111001111111001111001111",
    "1011101011101011101011101011101",
]


def hx(a):
    return ["%016x" % int(x) for x in np.asarray(a).ravel()]


def rand_seq(rng, n, alphabet="ACGT"):
    return "".join(alphabet[i] for i in rng.randint(0, len(alphabet), n))


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def main():
    ref = Ref()
    rng = np.random.RandomState(20240229)

    # ---------------- G1: ntHashIterator vectors ----------------
    base = rand_seq(rng, 260)
    seqs = [
        "ACGTAC",  # Tests/Unit/BloomFilterTests.cpp:72
        "ACGTACACTGGACTGAGTCT",  # Tests/Unit/CountingBloomFilterTests.cpp:78
        "TAGAATCACCCAAAGA",  # README example
        base,
        base.lower(),
        "".join(c.lower() if i % 3 == 0 else c for i, c in enumerate(base)),
        "NNNN" + base[:100],
        base[:100] + "NNN",
        base[:60] + "N" + base[60:130] + "NN" + base[130:200] + "n" + base[200:],
        base[:80].replace("T", "U") + base[80:160].replace("T", "u"),
        "N" * 50,
        "ACGTNACGTACGTACGGTCA",
        base[:40] + "R" + base[41:90] + "-" + base[91:140] + "*" + base[141:200],
        base[:33] + "\x01\x03\x04\x05\x07" + base[38:120],  # raw bytes seedTab accepts
        "A" * 70,
        "ACGT" * 20,
        "",
        "ACG",
    ]
    g1 = []
    for s in seqs:
        for k in (4, 5, 8, 25, 31, 32, 33, 64, 100):
            for h in (1, 3, 4, 5):
                if (k, h) not in ((4, 5), (5, 4), (8, 5), (25, 3), (31, 4), (32, 1), (33, 3), (64, 4), (100, 1)):
                    continue
                pos, hv = ref.nthash_seq(s.encode("latin-1"), h, k)
                g1.append(dict(seq=s, k=k, h=h, pos=[int(p) for p in pos], hashes=hx(hv)))
    # ---------------- G2: stHashIterator vectors ----------------
    g2 = []
    toy = ["1110111", "1011101"]
    for s in seqs[:14]:
        for seeds, k in ((toy, 7), (SEEDS_C5, 31)):
            for h2 in (1, 2):
                pos, hv, st = ref.sthash_seq(s.encode("latin-1"), seeds, h2, k)
                g2.append(dict(seq=s, k=k, seeds=seeds, h2=h2, pos=[int(p) for p in pos],
                               hashes=hx(hv), strand=[int(x) for x in st.ravel()]))
    # raw-k-mer hashes (KmerBloomFilter path: NTC64(kmer,k) + NTE64)
    gk = []
    for k in (4, 5, 6, 7, 25, 31, 32, 33, 64):
        for t in range(3):
            kmer = rand_seq(rng, k) if t else rand_seq(rng, k).lower()
            # k % 4 == 0: the reference's tetramer path shifts a uint64_t by 64 (rolx/swapxbits033 with
            # remainder 0, nthash.hpp:354-356,388-391,404-406) -- undefined behaviour; recorded but flagged
            gk.append(dict(kmer=kmer, k=k, h=4, hashes=hx(ref.kmer_hashes(kmer, k, 4)), ub=(k % 4 == 0)))
    # k-mers containing U / u: the reference's raw-k-mer path maps U to A (its 2/3/4-mer tables index with
    # convertTab, nthash.hpp:16-86) while its iterator path hashes U like T (seedTab, nthash.hpp:195-228) -- the
    # two paths of the reference disagree with each other; what g++/x86 returns is recorded and flagged
    rng_u = np.random.RandomState(77)  # its own generator: the fixtures that follow keep their random inputs
    for k in (5, 7, 25, 31, 33):
        for t in range(2):
            km = list(rand_seq(rng_u, k))
            for j in rng_u.choice(k, 2, replace=False):
                km[j] = "Uu"[t]
            kmer = "".join(km)
            gk.append(dict(kmer=kmer, k=k, h=4, hashes=hx(ref.kmer_hashes(kmer, k, 4)), ub=False, u=True))
    json.dump(dict(nthash=g1, sthash=g2, kmer=gk), open(os.path.join(HERE, "hash_vectors.json"), "w"))

    # ---------------- G3: tiny .bf files (whole-file bytes) ----------------
    files = []

    def make_bf(name, bits, h, k, seqs_in, n_entry=0, t_entry=0):
        f = ref.bf(bits, h, k)
        for s in seqs_in:
            f.insert_seq(s)
        if n_entry or t_entry:
            f.set_entries(n_entry, t_entry)
        path = os.path.join(HERE, name)
        f.store(path)
        files.append(dict(file=name, kind="bloom", bits=bits, h=h, k=k, inserted=seqs_in,
                          n_entry=n_entry, t_entry=t_entry, pop=f.pop(), sha256=sha(open(path, "rb").read())))
        f.close()

    make_bf("bf_1024_k31_h4.bf", 1024, 4, 31, [base[:150]])
    make_bf("bf_1000_k5_h4_readme.bf", 1000, 4, 5, ["TAGAATCACCCAAAGA"])
    make_bf("bf_1000_k25_h3_entries.bf", 1000, 3, 25, [base[:120], base[100:260]], 17, 123456789012)
    make_bf("bf_4096_k4_h5_unit.bf", 4096, 5, 4, ["ACGTAC"])
    make_bf("bf_64_empty.bf", 64, 1, 1, [])

    # ---------------- G4: tiny counting .bf files ----------------
    def make_cbf(name, nbytes, h, k, thr, seqs_in, op):
        f = ref.cbf(nbytes, h, k, thr)
        allh = []
        for s in seqs_in:
            _, hv = ref.nthash_seq(s, h, k)
            allh.append(hv)
            if op == "insert":
                f.insert(hv)
            else:
                f.increment_all(hv)
        path = os.path.join(HERE, name)
        f.store(path)
        q = np.concatenate(allh) if allh else np.zeros((0, h), np.uint64)
        mn, ct = f.query(q)
        files.append(dict(file=name, kind="counting", bytes=nbytes, size=int(f.size), h=h, k=k, thr=thr,
                          inserted=seqs_in, op=op, popcount=f.popcount(),
                          filtered_popcount=f.filtered_popcount(),
                          min_counts=[int(x) for x in mn], contains=[int(x) for x in ct],
                          sha256=sha(open(path, "rb").read())))
        f.close()

    rep = [base[:150], base[:150], base[50:200], base[:150], "ACGTNNNN" + base[10:90]]
    make_cbf("cbf_1000_k25_h3_insert.bf", 1000, 3, 25, 2, rep, "insert")
    make_cbf("cbf_1000_k25_h3_incall.bf", 1000, 3, 25, 2, rep, "increment_all")
    make_cbf("cbf_100001_k8_h5_unit.bf", 100001, 5, 8, 1, ["ACGTACACTGGACTGAGTCT"], "insert")
    # saturation: the same read 300 times into a tiny filter
    make_cbf("cbf_64_k8_h2_saturated.bf", 64, 2, 8, 3, [base[:40]] * 300, "increment_all")
    make_cbf("cbf_64_k8_h2_saturated_min.bf", 64, 2, 8, 3, [base[:40]] * 300, "insert")
    json.dump(files, open(os.path.join(HERE, "files.json"), "w"), indent=1)

    # ---------------- G5/G6/G8: filter operations on precomputed hashes ----------------
    ops = {}
    A = [rand_seq(rng, 150) for _ in range(40)]
    B = A[:20] + [rand_seq(rng, 150) for _ in range(20)]
    B[3] = B[3][:70] + "N" + B[3][71:]
    for bits in (1 << 16, 100000):
        f = ref.bf(bits, 4, 31)
        for s in A:
            f.insert_seq(s)
        res = []
        for s in B:
            pos, r = f.contains_seq(s)
            res.append(dict(pos=[int(p) for p in pos], hit=[int(x) for x in r]))
        ops["contains_%d" % bits] = dict(bits=bits, h=4, k=31, A=A, B=B, result=res,
                                         body_sha256=sha(f.bytes()), pop=f.pop())
        f.close()
    # G6 insertAndCheck on a stream with repeats
    stream = [A[0], A[1], A[0], A[2], A[1][:100], A[0][20:], A[3]]
    f = ref.bf(1 << 15, 3, 25)
    seqres = []
    for s in stream:
        _, hv = ref.nthash_seq(s, 3, 25)
        seqres.append([int(x) for x in f.insert_and_check(hv)])
    ops["insert_and_check"] = dict(bits=1 << 15, h=3, k=25, stream=stream, result=seqres,
                                   body_sha256=sha(f.bytes()))
    f.close()
    # counting insertAndCheck
    c = ref.cbf(4096, 3, 25, 2)
    cres = []
    for s in stream:
        _, hv = ref.nthash_seq(s, 3, 25)
        cres.append([int(x) for x in c.insert_and_check(hv)])
    ops["cbf_insert_and_check"] = dict(bytes=4096, h=3, k=25, thr=2, stream=stream, result=cres,
                                       body_sha256=sha(c.counters()))
    c.close()
    # G8 modulo edge cases: which bit does hash x land on for size m
    mods = []
    M64 = (1 << 64) - 1
    for m in (8, 64, 1000, 1 << 20, (1 << 20) + 8, 10 ** 9, 999999992):
        hv = [0, 1, 7, 8, m - 1, m, m + 1, 2 * m - 1, M64, M64 - 1, 1 << 63, (1 << 63) + m - 1,
              0x9E3779B97F4A7C15, 0xDEADBEEFCAFEBABE, M64 // 3, M64 // m * m, (M64 // m * m - 1) & M64]
        hv = [x & M64 for x in hv]
        f = ref.bf(m, 1, 4)
        f.insert(np.array(hv, np.uint64))
        body = f.bytes()
        setbits = np.flatnonzero(np.unpackbits(body, bitorder="little"))
        mods.append(dict(size=m, hashes=["%016x" % x for x in hv], set_bits=[int(b) for b in setbits]))
        f.close()
    ops["modulo"] = mods
    json.dump(ops, open(os.path.join(HERE, "filter_ops.json"), "w"))

    # ---------------- G7: medium digests over synthetic reads ----------------
    dig = {}
    first = ref.synth_reads(42, 0, 3, 150).tobytes().decode()
    dig["synth"] = dict(seed=42, read_len=150, first3=first,
                        sha256_first_1000=sha(ref.synth_reads(42, 0, 1000, 150)),
                        sha256_seed43_from_12345=sha(ref.synth_reads(43, 12345, 100, 150)),
                        sha256_len100=sha(ref.synth_reads(7, 5, 64, 100)))
    for name, n_reads, bits, k, h in (("small", 20000, 1 << 24, 31, 4),
                                      ("medium", 200000, 1 << 30, 31, 4),
                                      ("nonpow2", 50000, 100000000 - 64, 25, 3),
                                      ("config1", 1000000, 1 << 33, 31, 4)):
        f = ref.bf(bits, h, k)
        f.insert_synth(42, 0, n_reads, 150)
        body = f.bytes()
        d = dict(n_reads=n_reads, bits=bits, k=k, h=h, seed=42, read_len=150, pop=f.pop(),
                 body_sha256=sha(body),
                 hits_seed42=f.count_synth(42, 0, min(n_reads, 100000), 150),
                 hits_seed43=f.count_synth(43, 0, min(n_reads, 100000), 150),
                 n_query=min(n_reads, 100000))
        dig["bf_" + name] = d
        print(name, d["pop"], d["hits_seed42"], d["hits_seed43"], flush=True)
        f.close()
    for name, n_reads, nbytes, k, h, op in (("cbf_small_min", 20000, 1 << 22, 25, 3, 0),
                                            ("cbf_small_all", 20000, 1 << 22, 25, 3, 1),
                                            ("cbf_medium_all", 200000, 1 << 27, 25, 3, 1)):
        c = ref.cbf(nbytes, h, k, 2)
        c.update_synth(42, 0, n_reads, 150, op)
        c.update_synth(42, 0, n_reads // 2, 150, op)  # second pass over half: exercises threshold 2
        dig[name] = dict(n_reads=n_reads, bytes=nbytes, k=k, h=h, thr=2, op=op, seed=42, read_len=150,
                         body_sha256=sha(c.counters()), popcount=c.popcount(),
                         filtered_popcount=c.filtered_popcount())
        print(name, dig[name]["popcount"], dig[name]["filtered_popcount"], flush=True)
        c.close()
    json.dump(dig, open(os.path.join(HERE, "digests.json"), "w"), indent=1)
    print("golden fixtures written to", HERE)


def kmer_path():
    """kmer_path.json: the raw-k-mer path KmerBloomFilter::insert/contains(const char*) takes
    (KmerBloomFilter.hpp:47-74 -> NTC64(kmerSeq, k), nthash.hpp:394-439,460-465), beyond the cases in
    hash_vectors.json: every k from 1 to 40 and a few large ones (uint8_t offsets wrap above 256), U, bytes that
    are not bases (the uint8_t table index wraps), and the replay of swig/test.pl.  `defined` = 0 marks k-mers
    whose 2-/3-base remainder makes the reference index beyond dimerTab / trimerTab (no value to pin)."""
    ref = Ref()
    rng = np.random.RandomState(424242)
    alpha = "ACGT" * 6 + "acgt" * 2 + "UuNn-*" + "\x01\x03\x04\x05\x07"
    conv = {c: v for v, cs in enumerate(("AaUu", "Cc", "Gg", "Tt")) for c in cs}

    def defined(km, k):
        r = k % 4
        if r in (0, 1):
            return 1
        idx = 0
        for c in km[k - r:]:
            idx = 4 * idx + conv.get(c, 255)
        ridx = 0
        for c in reversed(km[k - r:]):
            ridx = 4 * ridx + (3 - conv[c] if c in conv else 255)
        return int((idx & 255) < 4 ** r and (ridx & 255) < 4 ** r)

    cases = []
    for k in list(range(1, 41)) + [64, 100, 255, 256, 257, 260, 300]:
        for t in range(4):
            if t == 0:
                km = rand_seq(rng, k)
            elif t == 1:
                km = rand_seq(rng, k, "ACGTacgtUu")
            else:
                km = rand_seq(rng, k, alpha)
            d = defined(km, k)
            cases.append(dict(kmer=km, k=k, h=3, defined=d,
                              hashes=hx(ref.kmer_hashes(km.encode("latin-1"), k, 3)) if d else []))
    out = dict(hashes=cases)

    # swig/test.pl:8-26: BloomFilter(1000000000, 5, 20) = KmerBloomFilter, four raw k-mers inserted, six queried
    ins = ["ATCGGGTCATCAACCAATAT", "ATCGGGTCATCAACCAATAC", "ATCGGGTCATCAACCAATAG", "ATCGGGTCATCAACCAATAA"]
    qry = ins + ["ATCGGGTCATCAACCAATTA", "ATCGGGTCATCAACCAATTC"]
    f = ref.bf(1000000000, 5, 20)
    for km in ins:
        f.insert_kmer(km)
    path = os.path.join(HERE, "_swig_test_pl.bf")
    f.store(path)
    raw = open(path, "rb").read()
    os.remove(path)  # 125 MB: pinned by digest, header and set bits instead
    hend = raw.index(b"[HeaderEnd]\n") + len(b"[HeaderEnd]\n")
    body = np.frombuffer(raw, np.uint8, offset=hend)
    nz = np.flatnonzero(body)
    setbits = sorted(int(8 * i + b) for i in nz for b in range(8) if (body[i] >> b) & 1)
    # the same four k-mers through the ITERATOR path land elsewhere (k = 20: k % 4 == 0)
    g = ref.bf(1000000000, 5, 20)
    for km in ins:
        g.insert_seq(km)
    gb = g.bytes()
    out["swig_test_pl"] = dict(bits=1000000000, h=5, k=20, inserted=ins, queried=qry,
                               contains=[int(f.contains_kmer(km)) for km in qry],
                               header=raw[:hend].decode(), file_sha256=sha(raw), body_sha256=sha(body), pop=f.pop(),
                               set_bits=setbits, same_as_iterator_path=bool((gb == body).all()))
    f.close()
    g.close()
    # swig/test.pl:59-84: insertSeq + contains(kmer) of every 5-mer
    s = "TAGAATCACCCAAAGA"
    f = ref.bf(10000, 4, 5)
    f.insert_seq(s)
    out["swig_insert_seq"] = dict(bits=10000, h=4, k=5, seq=s,
                                  contains=[int(f.contains_kmer(s[i:i + 5])) for i in range(len(s) - 4)],
                                  body_sha256=sha(f.bytes()))
    f.close()
    json.dump(out, open(os.path.join(HERE, "kmer_path.json"), "w"))
    print("kmer_path.json:", len(cases), "hash cases,", sum(c["defined"] for c in cases), "defined; test.pl pop",
          out["swig_test_pl"]["pop"], "same as iterator path:", out["swig_test_pl"]["same_as_iterator_path"])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "kmer":
        kmer_path()  # only kmer_path.json (added in round 3; the other fixtures are left as they are)
    else:
        main()
        kmer_path()

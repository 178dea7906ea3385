"""The CPU restatement against the genuine reference build on fresh random inputs.
Runs only where oracle/_ref/libbtlref.so exists (this container; it also travels to the GPU box)."""
import numpy as np
import pytest


def rand_seq(rng, n, p_bad=0.02):
    s = rng.choice(list(b"ACGTacgt"), n).astype(np.uint8)
    bad = rng.rand(n) < p_bad
    s[bad] = rng.choice(list(b"NnRY-\x00\x01\x07\xff"), bad.sum())
    return s.tobytes()


@pytest.mark.parametrize("k,h", [(4, 5), (25, 3), (31, 4), (32, 2), (33, 1), (64, 4), (150, 2)])
def test_nthash_random(oracle, ref, k, h):
    rng = np.random.RandomState(k * 100 + h)
    tot = 0
    for _ in range(60):
        s = rand_seq(rng, int(rng.randint(0, 600)), p_bad=rng.choice([0, 0.005, 0.05]))
        a, b = oracle.nthash_seq(s, h, k), ref.nthash_seq(s, h, k)
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
        tot += len(a[0])
    assert tot > 0


def test_kmer_path_random(oracle, ref):
    """raw k-mers (KmerBloomFilter's NTC64(kmerSeq, k) path): k 1..300, U, lowercase, bytes that are not bases"""
    rng = np.random.RandomState(11)
    alpha = list(b"ACGT" * 5 + b"acgtUuNn\x01\x03-*")
    n_ok = n_skip = 0
    for it in range(6000):
        k = int(rng.randint(1, 300)) if it % 10 == 0 else int(rng.randint(1, 41))
        km = bytes(rng.choice(alpha, k).tolist())
        hv, ok = oracle.kmer_hashes(km, k, 4)
        if ok[0]:  # (the reference reads beyond its 2-/3-mer tables otherwise: nothing to compare)
            assert (hv[0] == ref.kmer_hashes(km, k, 4)).all(), (km, k)
            n_ok += 1
        else:
            assert k % 4 in (2, 3)
            n_skip += 1
    assert n_ok > 4000 and n_skip > 0


@pytest.mark.parametrize("k,h2", [(7, 1), (7, 2), (31, 1), (31, 3)])
def test_sthash_random(oracle, ref, k, h2):
    rng = np.random.RandomState(k + h2)
    seeds = []
    for _ in range(3):
        half = "".join(rng.choice(list("1101"), (k + 1) // 2))
        seeds.append((half + half[::-1][k % 2:])[:k])
    for _ in range(30):
        s = rand_seq(rng, int(rng.randint(0, 400)), p_bad=rng.choice([0, 0.01]))
        a, b = oracle.sthash_seq(s, seeds, h2, k), ref.sthash_seq(s, seeds, h2, k)
        assert all((x == y).all() for x, y in zip(a, b))


@pytest.mark.parametrize("bits", [64, 1000, 1 << 16, 999992])
def test_bit_filter_ops(oracle, ref, bits):
    rng = np.random.RandomState(bits % 9973)
    h = 4
    f = ref.bf(bits, h, 31)
    mine = np.zeros(bits // 8, np.uint8)
    hv = rng.randint(0, 2 ** 63, size=(500, h)).astype(np.uint64) * np.uint64(2) + rng.randint(0, 2, size=(500, h)).astype(np.uint64)
    assert oracle.bf_insert_and_check(mine, bits, h, hv[:200]).tolist() == f.insert_and_check(hv[:200]).tolist()
    oracle.bf_insert(mine, bits, h, hv[200:300])
    f.insert(hv[200:300])
    assert (mine == f.bytes()).all()
    assert oracle.bf_contains(mine, bits, h, hv).tolist() == f.contains(hv).tolist()
    assert oracle.bf_popcount(mine, bits) == f.pop()


def test_counting_ops(oracle, ref):
    rng = np.random.RandomState(5)
    h, thr = 3, 2
    f = ref.cbf(1001, h, 25, thr)
    c = np.zeros(oracle.cbf_round_bytes(1001), np.uint8)
    assert c.size == f.size == 1008
    for rnd in range(6):
        hv = rng.randint(0, 2 ** 62, size=(700, h)).astype(np.uint64)
        hv[::7, 1] = hv[::7, 0]  # duplicate positions inside one k-mer
        if rnd % 3 == 0:
            oracle.cbf_increment_min(c, h, hv); f.insert(hv)
        elif rnd % 3 == 1:
            oracle.cbf_increment_all(c, h, hv); f.increment_all(hv)
        else:
            assert oracle.cbf_insert_and_check(c, h, thr, hv).tolist() == f.insert_and_check(hv).tolist()
        assert (c == f.counters()).all()
        a, b = oracle.cbf_query(c, h, thr, hv), f.query(hv)
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
    assert oracle.cbf_popcount(c) == f.popcount()
    assert oracle.cbf_filtered_popcount(c, thr) == f.filtered_popcount()


def test_headers_and_files(oracle, ref, tmp_path):
    f = ref.bf(4096, 7, 21)
    f.insert_seq(b"ACGTTGCATGCATGCATGCAGTCAGTCGATGCATGCA")
    f.set_entries(12, 34)
    p = str(tmp_path / "a.bf")
    f.store(p)
    raw = open(p, "rb").read()
    assert raw == oracle.bf_header(4096, 7, 21, 0.0, 12, 34) + f.bytes().tobytes()
    c = ref.cbf(123, 2, 9, 1)
    p2 = str(tmp_path / "c.bf")
    c.store(p2)
    assert open(p2, "rb").read() == oracle.cbf_header(128, 128, 2, 9) + bytes(128)


def test_synth_reads(oracle, ref):
    for seed, first, n, L in [(42, 0, 100, 150), (43, 10 ** 9, 10, 150), (1, 3, 7, 33), (9, 0, 5, 64)]:
        assert (oracle.synth_reads(seed, first, n, L) == ref.synth_reads(seed, first, n, L)).all()


def test_port_bench_matches_reference_bench(oracle, ref):
    a = oracle.bench_bf(3000, 150, 31, 4, 1 << 24, 42, 43, threads=2)
    b = ref.bench_bf(3000, 150, 31, 4, 1 << 24, 42, 43, threads=2, skip_pop=0)
    assert a["kmers"] == b["kmers"] == 3000 * 120
    assert a["hits"] == b["hits"] and a["popcount"] == b["popcount"]

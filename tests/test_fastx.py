"""FASTA / FASTQ ingestion: the host-side parser is checked on the CPU against a plain
Python parse of the same files; the file -> filter calls are checked on the GPU against the oracle."""
import gzip
import os
import random

import numpy as np
import pytest

K = 11


def _rand_seq(rng, n, with_n=True):
    s = "".join(rng.choice("ACGT") for _ in range(n))
    if with_n and n > 40:
        p = rng.randrange(n - 5)
        s = s[:p] + "N" * rng.randrange(1, 4) + s[p + 3:]
    if n > 20 and rng.random() < 0.3:
        s = s.lower()
    return s[:n]


def _wrap(s, w):
    return [s[i:i + w] for i in range(0, len(s), w)] or [""]


def make_files(tmp_path, rng):
    """returns {name: (path, sequences in RECORDS mode, sequences in LINES mode)}"""
    out = {}
    # multi-line FASTA, CRLF on some lines, blank lines, a long contig, no trailing newline
    recs = [_rand_seq(rng, n) for n in (5, 70, 1, 333, 12, 2500, 64, 0, 95)]
    lines, per_line = [], []
    for i, s in enumerate(recs):
        lines.append(">contig%d some description" % i)
        if i == 3:
            lines.append("")
        for piece in _wrap(s, 60):
            if piece:
                per_line.append(piece)
            lines.append(piece + ("\r" if i % 2 else ""))
    p = tmp_path / "multi.fa"
    p.write_bytes("\n".join(lines).encode())  # no newline at the end
    out["fasta"] = (p, [s for s in recs if s], per_line)
    # FASTQ, including a zero-length read and '@' / '>' as first quality characters
    reads = [_rand_seq(rng, n, with_n=(n > 60)) for n in (50, 75, 0, 100, 31, 11, 10, 150)]
    fq = []
    for i, s in enumerate(reads):
        q = "".join(rng.choice("@>+IJ#") for _ in s)
        fq += ["@read%d/1" % i, s, "+", q]
    p = tmp_path / "reads.fq"
    p.write_bytes(("\n".join(fq) + "\n").encode())
    out["fastq"] = (p, [s for s in reads if s], [s for s in reads if s])
    # the same FASTQ gzipped
    pz = tmp_path / "reads.fq.gz"
    with gzip.open(pz, "wb") as fh:
        fh.write(p.read_bytes())
    out["fastq_gz"] = (pz, out["fastq"][1], out["fastq"][2])
    # one sequence per line
    plain = [_rand_seq(rng, n) for n in (40, 41, 9, 200)]
    p = tmp_path / "plain.txt"
    p.write_bytes(("\n".join(plain) + "\n").encode())
    out["plain"] = (p, plain, plain)
    return out


def rebuild(batches, k):
    """undo the batching: pieces of a cut sequence overlap by k-1 bases (or are whole)"""
    seqs, pieces = [], []
    for bases, starts in batches:
        assert starts[0] == 0 and starts[-1] == len(bases)
        assert all(a <= b for a, b in zip(starts, starts[1:]))
        for a, b in zip(starts, starts[1:]):
            pieces.append(bases[a:b].decode())
    return pieces


def windows(seqs, k):
    """multiset of all length-k substrings (what the kernels hash before the ACGT test)"""
    from collections import Counter
    c = Counter()
    for s in seqs:
        for i in range(len(s) - k + 1):
            c[s[i:i + k]] += 1
    return c


@pytest.mark.parametrize("batch", [0, 64 + 4 * K + 9, 300, 4096])
def test_parser_matches_python(tmp_path, batch):
    import btl_bloomfilter_amd as m
    rng = random.Random(7)
    files = make_files(tmp_path, rng)
    for name, (path, recs, per_line) in files.items():
        for lines_mode, want in ((False, recs), (True, per_line)):
            got = rebuild(list(m.fastx_batches(path, K, per_line=lines_mode, batch_bytes=batch)), K)
            # every window of every sequence exactly once, whatever the batch size
            assert windows(got, K) == windows(want, K), (name, lines_mode, batch)
            if batch == 0:  # nothing is cut: the sequences come through as they are
                assert [g for g in got if g] == want, (name, lines_mode)


def test_parser_errors(tmp_path):
    import btl_bloomfilter_amd as m
    from btl_bloomfilter_amd._lib import BtlbfError
    with pytest.raises(BtlbfError) as e:
        list(m.fastx_batches(tmp_path / "missing.fa", K))
    assert e.value.code == 3 and "could not be read" in str(e.value)
    p = tmp_path / "x.fa"
    p.write_text(">a\nACGT\n")
    with pytest.raises(BtlbfError):
        list(m.fastx_batches(p, K, batch_bytes=8))  # batch smaller than a window
    p.write_bytes(b"")
    assert list(m.fastx_batches(p, K)) == []


@pytest.mark.gpu
@pytest.mark.parametrize("batch,threads", [(0, 1), (700, 1), (0, 3), (900, 5)])
def test_insert_and_contains_file(tmp_path, oracle, batch, threads, monkeypatch):
    import btl_bloomfilter_amd as m

    # threads > 1: one parser thread per byte range of the (uncompressed) file, whatever its size
    monkeypatch.setenv("BTLBF_FASTX_THREADS", str(threads))
    monkeypatch.setenv("BTLBF_FASTX_MT_MIN_BYTES", "0" if threads > 1 else str(1 << 40))
    rng = random.Random(11)
    files = make_files(tmp_path, rng)
    bits, h, k = 1 << 20, 3, K
    for name, (path, recs, per_line) in files.items():
        for lines_mode, want in ((False, recs), (True, per_line)):
            f = m.BloomFilter(bits, h, k)
            st = f.insertFile(path, per_line=lines_mode, batch_bytes=batch)
            ref = np.zeros(bits // 8, np.uint8)
            n_clean = 0
            for s in want:
                oracle.bf_insert_seq(ref, bits, h, k, s.encode())
                n_clean += len(oracle.nthash_seq(s.encode(), h, k)[0])
            assert np.array_equal(f.download(), ref), (name, lines_mode, batch)
            assert st["n_batches"] >= 1 and st["n_bases"] >= sum(len(s) for s in want)
            # every clean window of the file is found again
            q = f.containsFile(path, per_line=lines_mode, batch_bytes=batch)
            assert q["n_windows"] == n_clean and q["n_hits"] == n_clean, (name, lines_mode, batch, q)
    other = tmp_path / "other.fa"
    other.write_text(">x\n" + _rand_seq(random.Random(99), 5000, with_n=False) + "\n")
    f = m.BloomFilter(bits, h, k)
    f.insertFile(files["fasta"][0])
    q = f.containsFile(other)
    assert q["n_windows"] == 5000 - k + 1 and q["n_hits"] < 20


def test_parser_random_files(tmp_path):
    """seeded random FASTA / FASTQ / plain files (wrap widths, CRLF, blank lines, empty records, missing
    final newline) at random batch sizes: every window of every sequence comes out exactly once"""
    import btl_bloomfilter_amd as m

    rng = random.Random(2024)
    for case in range(120):
        kind = rng.choice(["fasta", "fastq", "plain"])
        k = rng.choice([1, 3, 11, 31])
        crlf = rng.random() < 0.3
        nl = "\r\n" if crlf else "\n"
        seqs = [_rand_seq(rng, rng.choice([0, 1, 2, k - 1, k, k + 1, 40, 90, 400]), with_n=rng.random() < 0.5)
                for _ in range(rng.randrange(1, 12))]
        lines, per_line = [], []
        for i, s in enumerate(seqs):
            if kind == "fasta":
                lines.append(">s%d" % i)
                if rng.random() < 0.2:
                    lines.append("")
                for piece in _wrap(s, rng.choice([7, 60, 1000])):
                    lines.append(piece)
                    if piece:
                        per_line.append(piece)
            elif kind == "fastq":
                lines += ["@q%d" % i, s, "+", "".join(rng.choice("@>+#I") for _ in s)]
                if s:
                    per_line.append(s)
            else:
                if s:  # a blank line is no sequence
                    lines.append(s)
                    per_line.append(s)
        text = nl.join(lines) + (nl if rng.random() < 0.7 else "")
        path = tmp_path / ("f%d" % case)
        path.write_bytes(text.encode())
        recs = [s for s in seqs if s]
        for lines_mode in (False, True):
            want = per_line if (lines_mode and kind == "fasta") else (recs if kind != "plain" else per_line)
            batch = rng.choice([0, 4 * k + 64, 4 * k + 64 + rng.randrange(1, 50), 977, 1 << 14])
            got = rebuild(list(m.fastx_batches(path, k, per_line=lines_mode, batch_bytes=batch)), k)
            assert windows(got, k) == windows(want, k), (case, kind, k, crlf, lines_mode, batch)


def test_ranged_readers_tile_the_file(tmp_path):
    """parallel parsing: readers over byte ranges that tile the file deliver every record exactly once,
    wherever the cuts fall (every cut position of small files, random cuts of larger ones)"""
    import os

    import btl_bloomfilter_amd as m

    rng = random.Random(99)
    files = make_files(tmp_path, rng)
    k = K
    for name, fmt in (("fasta", "fasta"), ("fastq", "fastq"), ("plain", "plain")):
        path, recs, per_line = files[name]
        size = os.path.getsize(path)
        for lines_mode in (False, True):
            want = windows(per_line if lines_mode else recs, k)
            cut_sets = [[c] for c in range(0, size + 1, 1 if size < 1500 else 37)]
            cut_sets += [sorted(rng.sample(range(size), 5)) for _ in range(20)]
            for cuts in cut_sets:
                edges = [0] + cuts + [size]
                got = []
                for a, b in zip(edges[:-1], edges[1:]):
                    if a == b:
                        continue
                    got += rebuild(list(m.fastx_batches(path, k, per_line=lines_mode, batch_bytes=4096,
                                                        byte_range=(a, b), fmt=fmt)), k)
                assert windows(got, k) == want, (name, lines_mode, cuts)

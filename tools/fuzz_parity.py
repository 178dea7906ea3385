#!/usr/bin/env python3
"""Randomised differential run: partitioned pipeline against the direct kernels (both product paths)
over random filter sizes, k, h, layouts, N content, skew and scratch caps.  Not part of the test suite;
run it on a GPU box for as long as you like:   python tools/fuzz_parity.py [seconds] [seed]"""
import sys
import time

import numpy as np
import torch

import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import btl_bloomfilter_amd as m


def rand_reads(rng, n_reads, L, p_bad):
    a = rng.choice(np.frombuffer(b"ACGTacgt", np.uint8), size=(n_reads, L))
    bad = rng.random((n_reads, L)) < p_bad
    a[bad] = rng.choice(np.frombuffer(b"NnR-\x00\x01\x07", np.uint8), size=int(bad.sum()))
    return a


def one_case(rng, it):
    lg = int(rng.integers(10, 34))
    bits = (1 << lg) if rng.random() < 0.6 else int(rng.integers(1 << (lg - 1), 1 << lg)) // 64 * 64 + 64
    k = int(rng.choice([1, 4, 5, 11, 21, 25, 31, 32, 33, 47, 64, 96, 150]))
    h = int(rng.integers(1, 9))
    L = int(rng.choice([k, k + 1, 50, 100, 150, 151, 250, 1000]))
    L = max(L, k)
    n_reads = int(rng.integers(1, max(2, min(200000, 30_000_000 // L))))
    reads = rand_reads(rng, n_reads, L, float(rng.choice([0.0, 0.001, 0.02])))
    if rng.random() < 0.3:  # skew: many copies of a few reads
        reads[rng.integers(0, n_reads, n_reads // 2)] = reads[0]
    counting = rng.random() < 0.25
    seeds, h2 = None, 1
    if not counting and k >= 4 and rng.random() < 0.2:  # spaced seeds (stHashIterator): h = n_seeds * h2
        n_seeds, h2 = int(rng.integers(1, 4)), int(rng.integers(1, 3))
        h = n_seeds * h2
        seeds = []
        for _ in range(n_seeds):
            bits01 = (rng.random(k) < 0.7).astype(int)
            bits01[0] = bits01[-1] = 1
            seeds.append("".join(map(str, bits01)))
    ragged = rng.random() < 0.3
    flat = torch.from_numpy(reads.reshape(-1).copy()).cuda()
    kw = {}
    if ragged:
        cuts = np.sort(rng.choice(np.arange(1, flat.numel()), size=min(n_reads, flat.numel() - 1), replace=False))
        starts = np.concatenate([[0], cuts, [flat.numel()]]).astype(np.int64)
        kw["starts"] = torch.from_numpy(starts).cuda()
    else:
        kw["read_len"] = L
    scratch = int(rng.choice([0, 0, 32 << 20, 256 << 20]))
    thr = int(rng.integers(1, 4))
    q = flat.clone()
    q[torch.from_numpy(rng.integers(0, q.numel(), max(1, q.numel() // 5000))).cuda()] = ord("A")
    if not ragged and rng.random() < 0.5:  # foreign reads among the inserted ones: AUTO may split the query
        frac = float(rng.choice([0.02, 0.1, 0.5, 0.9]))
        idx = np.flatnonzero(rng.random(n_reads) < frac)
        if idx.size:
            q.view(n_reads, L)[torch.from_numpy(idx).cuda()] = torch.from_numpy(rand_reads(rng, idx.size, L, 0.001)).cuda()
    res = []
    stateful = rng.random() < 0.3 and flat.numel() < 12_000_000
    rng_state = int(rng.integers(0, 1 << 30))
    for mode in ("direct", "partitioned", "auto"):
        if counting:
            f = m.CountingBloomFilter(max(bits // 8, 64), h, k, thr)
        else:
            f = m.BloomFilter(bits // 8 * 8, h, k)
            if seeds:
                f.setSpacedSeeds(seeds, h2)
        f.setInsertMode(mode, scratch_bytes=scratch if mode == "partitioned" else 0)
        f.setQueryMode(mode)
        if counting:
            f.insertSeqs(flat, increment_all=True, **kw)
            f.insertSeqs(flat[: flat.numel() // 2 // L * L] if not ragged else flat, increment_all=True,
                         **({"read_len": L} if not ragged else kw))
        else:
            f.insertSeqs(flat, **kw)
        hit, valid, cnt = f.containsSeqs(q, want_counts=True, **kw)
        torch.cuda.synchronize()
        body = f.download().copy()
        extra = None
        if stateful:
            # the same filter goes on: a second (not fresh) insert, a clear (lazy), an insert of the query reads
            # (fresh again), a query of the first reads
            if counting:
                f.insertSeqs(q, increment_all=True, **kw)
            else:
                f.insertSeqs(q, **kw)
            b2 = f.download().copy()
            f.clear()
            if rng_state % 2:
                assert not f.download().any()  # something that must see the cleared array
            if counting:
                f.insertSeqs(q, increment_all=True, **kw)
            else:
                f.insertSeqs(q, **kw)
            h3, v3, c3 = f.containsSeqs(flat, want_counts=True, **kw)
            torch.cuda.synchronize()
            extra = (b2, f.download().copy(), h3.cpu().numpy().copy(), c3.tolist())
        res.append((body, hit.cpu().numpy().copy(), cnt.tolist(), thr, valid.cpu().numpy().copy(), extra))
        f.releaseScratch()
        del f
    a, b, c = res
    same = [bool((a[0] == b[0]).all()), bool((a[1] == b[1]).all()), a[2] == b[2], bool((a[4] == b[4]).all()),
            bool((a[0] == c[0]).all()), bool((a[1] == c[1]).all()), a[2] == c[2], bool((a[4] == c[4]).all())]
    if stateful:
        for o in (b, c):
            same += [bool((a[5][0] == o[5][0]).all()), bool((a[5][1] == o[5][1]).all()),
                     bool((a[5][2] == o[5][2]).all()), a[5][3] == o[5][3]]
    ok = all(same)
    desc = dict(it=it, bits=bits, k=k, h=h, L=L, n_reads=n_reads, counting=bool(counting), ragged=bool(ragged), scratch=scratch,
                spaced=bool(seeds), stateful=bool(stateful))
    if not ok:
        print("MISMATCH (filter, hits, counts, valid) x (partitioned, auto) [+ stateful leg: body2, body3, hits3, counts3 x 2] =",
              same, desc, a[2], b[2], c[2], flush=True)
    return ok, desc


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0, it, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        ok, desc = one_case(rng, it)
        bad += not ok
        it += 1
        if it % 10 == 0:
            print("cases %d  mismatches %d  last %s" % (it, bad, desc), flush=True)
    print("DONE cases %d mismatches %d" % (it, bad), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Rates of the BASELINE.json configurations other than the headline one (C1, C3, C5) -- DESIGN.md
section 5's side table, not the contract bench.   python tools/config_bench.py [n_reads]"""
import json
import sys

import torch

import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import btl_bloomfilter_amd as m

L = 150
SEEDS = ["1110111011101110111011101110111", "1101101101101101011011011011011",
         "1111001111001111111001111001111", "1011101011101011101011101011101"]


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3, r


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    out = {}
    # C1: 10^6 reads, 2^33 bits, k=31, h=4 (the reference's own CPU-runnable case)
    r1 = m.synth_reads_device(42, 0, 1_000_000, L)
    f = m.BloomFilter(1 << 33, 4, 31)
    f.insertSeqs(r1, read_len=L)  # warm-up (scratch allocation)
    f.containsSeqs(r1, read_len=L, want_valid=False, want_counts=True)
    f.clear()
    ti, _ = timed(lambda: f.insertSeqs(r1, read_len=L))
    tq, (_, _, c) = timed(lambda: f.containsSeqs(r1, read_len=L, want_valid=False, want_counts=True))
    km = 1_000_000 * (L - 31 + 1)
    out["C1"] = {"insert_Mkmers_s": km / ti / 1e6, "query_Mkmers_s": km / tq / 1e6, "pop": f.getPop(),
                 "hits": c.tolist()}
    del f
    reads = m.synth_reads_device(42, 0, n, L)
    # C5: 2^37 bits, k=31, 4 spaced seeds x h2=1
    f = m.BloomFilter(1 << 37, 4, 31)
    f.setSpacedSeeds(SEEDS, 1)
    f.insertSeqs(reads, read_len=L)  # warm-up at full size: the first pass pays for the scratch allocation
    f.containsSeqs(reads, read_len=L, want_valid=False, want_counts=True)
    f.clear()
    ti, _ = timed(lambda: f.insertSeqs(reads, read_len=L))
    tq, (_, _, c) = timed(lambda: f.containsSeqs(reads, read_len=L, want_valid=False, want_counts=True))
    km = n * (L - 31 + 1)
    out["C5"] = {"insert_Mkmers_s": km / ti / 1e6, "query_Mkmers_s": km / tq / 1e6, "hits": c.tolist()}
    f.releaseScratch()
    del f
    torch.cuda.empty_cache()
    # C3: counting filter, 2^35 uint8 counters, k=25, h=3, threshold 2; every read set inserted twice
    cb = m.CountingBloomFilter(1 << 35, 3, 25, 2)
    km = n * (L - 25 + 1)
    t1, _ = timed(lambda: cb.insertSeqs(reads, read_len=L))
    t2, _ = timed(lambda: cb.insertSeqs(reads, read_len=L))
    cb.containsSeqs(reads, read_len=L, want_valid=False, want_counts=True)  # first partitioned call: scratch allocation
    tq, (_, _, c) = timed(lambda: cb.containsSeqs(reads, read_len=L, want_valid=False, want_counts=True))
    out["C3"] = {"insert_Mkmers_s": 2 * km / (t1 + t2) / 1e6, "query_Mkmers_s": km / tq / 1e6, "hits": c.tolist(),
                 "kmers": km}
    # the conservative update where the counters of a k-mer differ (reads the filter has not seen, into the filter as
    # it stands: 19 % of the counters are non-zero): only the counters at the minimum take a compare-and-swap
    other = m.synth_reads_device(43, 0, n // 4, L)
    t3, _ = timed(lambda: cb.insertSeqs(other, read_len=L))
    out["C3"]["insert_unseen_reads_into_filled_filter_Mkmers_s"] = (n // 4) * (L - 25 + 1) / t3 / 1e6
    out["C3"]["insert_first_pass_Mkmers_s"] = km / t1 / 1e6
    out["C3"]["insert_second_pass_Mkmers_s"] = km / t2 / 1e6
    del other
    cb.clear()
    cb.insertSeqs(reads, read_len=L, increment_all=True)  # warm-up: this path's own scratch size
    cb.clear()
    ta, _ = timed(lambda: cb.insertSeqs(reads, read_len=L, increment_all=True))
    out["C3"]["increment_all_Mkmers_s"] = km / ta / 1e6
    print(json.dumps(out))


if __name__ == "__main__":
    main()

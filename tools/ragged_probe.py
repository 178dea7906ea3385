"""diagnostic: partitioned insert / query of a RAGGED buffer (sequences of different lengths, btlbf_layout::starts)
against the same bases as uniform 150 bp reads: python tools/ragged_probe.py   (1.5x10^10 bases, 2^39 bits, h = 4)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import btl_bloomfilter_amd as m

n, L = 100_000_000, 150
reads = m.synth_reads_device(42, 0, n, L)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
rng = np.random.default_rng(1)
lens = rng.integers(100, 201, size=n + n // 10)
starts = np.concatenate([[0], np.cumsum(lens)])
starts = starts[starts <= n * L]
if starts[-1] != n * L:
    starts = np.concatenate([starts, [n * L]])
cases = {"uniform 150": dict(read_len=L), "ragged 100..200": dict(starts=torch.from_numpy(starts.astype(np.int64)).cuda()),
         "one sequence": dict()}
for name, kw in cases.items():
    f = m.BloomFilter(1 << 39, 4, 31)
    f.setProfiling(True)
    for rep in range(2):
        ev[0].record()
        f.insertSeqs(reads, **kw)
        ev[1].record()
        _, _, cnt = f.containsSeqs(reads, want_valid=True, want_counts=True, **kw)
        ev[2].record()
        torch.cuda.synchronize()
        prof = f.getProfile()
    ti, tq = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    kmers = cnt.tolist()[0]
    print("%s: %.3e k-mers; insert %.1f ms (%.1f Gk-mers/s), query %.1f ms (%.1f); (launches, ms each) %s" % (
        name, kmers, ti, kmers / ti / 1e6, tq, kmers / tq / 1e6, {k: (v[1], round(v[0] / v[1], 2)) for k, v in prof.items()}), flush=True)
    del f

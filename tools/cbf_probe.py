"""diagnostic: incrementAll / query rates of the counting filter's partitioned pipeline for other sizes and hash counts:
    python tools/cbf_probe.py bytes:h [bytes:h ...]   (5x10^7 reads of 150 bp, k = 31, threshold 1)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import btl_bloomfilter_amd as m

n, L = 50_000_000, 150
reads = m.synth_reads_device(42, 0, n, L)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for spec in sys.argv[1:]:
    size, h = (int(eval(x)) for x in spec.split(":"))
    f = m.CountingBloomFilter(size, h, 31, 1)
    f.setProfiling(True)
    for rep in range(2):
        ev[0].record()
        f.insertSeqs(reads, read_len=L, increment_all=True)
        ev[1].record()
        _, _, cnt = f.containsSeqs(reads, read_len=L, want_valid=False, want_counts=True)
        ev[2].record()
        torch.cuda.synchronize()
        prof = f.getProfile()
    kmers = n * (L - 30)
    ti, tq = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    print("bytes=%s h=%d: incrementAll %.1f ms (%.1f Gk-mers/s), query %.1f ms (%.1f); (launches, ms each) %s" % (
        spec.split(":")[0], h, ti, kmers / ti / 1e6, tq, kmers / tq / 1e6,
        {k: (v[1], round(v[0] / v[1], 2)) for k, v in prof.items()}), flush=True)
    del f

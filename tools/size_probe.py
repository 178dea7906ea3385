"""diagnostic: insert / query rates of the partitioned pipeline for other filter sizes (h = 4, k = 31, 150 bp reads):
    python tools/size_probe.py bits [bits ...]   (10^8 reads; `bits` may be an expression like 2**38 or 5*10**11)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import btl_bloomfilter_amd as m

n, L = 100_000_000, 150
reads = m.synth_reads_device(42, 0, n, L)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for spec in sys.argv[1:]:
    bits = int(eval(spec))
    f = m.BloomFilter(bits, 4, 31)
    f.setProfiling(True)
    for rep in range(2):
        f.clear() if hasattr(f, "clear") else None
        ev[0].record()
        f.insertSeqs(reads, read_len=L)
        ev[1].record()
        _, _, cnt = f.containsSeqs(reads, read_len=L, want_valid=False, want_counts=True)
        ev[2].record()
        torch.cuda.synchronize()
        prof = f.getProfile()
    kmers = n * (L - 30)
    ti, tq = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    print("bits=%s: insert %.1f ms (%.1f Gk-mers/s), query %.1f ms (%.1f), hits %d of %d; (launches, ms each) %s" % (
        spec, ti, kmers / ti / 1e6, tq, kmers / tq / 1e6, cnt.tolist()[1], kmers,
        {k: (v[1], round(v[0] / v[1], 2)) for k, v in prof.items()}), flush=True)
    del f

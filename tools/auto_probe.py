"""diagnostic: where AUTO's choice between the direct kernels and the partitioned pipeline should flip, by batch size:
    python tools/auto_probe.py [log2_bits=39]   (reads of 150 bp into a filter that already holds 10^8 reads)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import btl_bloomfilter_amd as m

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 39
L = 150
f = m.BloomFilter(1 << lg, 4, 31)
base = m.synth_reads_device(42, 0, 100_000_000, L)
f.insertSeqs(base, read_len=L)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for n in (250_000, 500_000, 1_000_000, 2_000_000, 3_000_000, 5_000_000, 10_000_000):
    batch = m.synth_reads_device(7, n, n, L)
    row = []
    for mode in ("direct", "partitioned", "auto"):
        f.setInsertMode(mode)
        f.setQueryMode(mode)
        for rep in range(2):
            ev[0].record()
            f.insertSeqs(batch, read_len=L)
            ev[1].record()
            f.containsSeqs(batch, read_len=L, want_valid=False)
            ev[2].record()
            torch.cuda.synchronize()
        row.append("%s %.1f / %.1f ms" % (mode, ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])))
    print("2^%d bits, %8d reads (%.2f %% of the array's bytes in probes): insert / query  %s" % (
        lg, n, 100.0 * n * 120 * 4 / ((1 << lg) / 8), "   ".join(row)), flush=True)

"""diagnostic: HIP-event time of pass A (insert / query) at the C2 geometry for other k-mer sizes and read lengths:
    python tools/kl_probe.py k:L [k:L ...]   (4.5x10^9 bases of synthetic reads, 2^39-bit filter, h = 4)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import btl_bloomfilter_amd as m

for spec in sys.argv[1:]:
    k, L = (int(x) for x in spec.split(":"))
    n = 4_500_000_000 // L
    reads = m.synth_reads_device(42, 0, n, L)
    f = m.BloomFilter(1 << 39, 4, k)
    f.setInsertMode("partitioned")
    f.setQueryMode("partitioned")
    f.setProfiling(True)
    for rep in range(2):
        f.insertSeqs(reads, read_len=L)
        f.containsSeqs(reads, read_len=L, want_valid=False)
        torch.cuda.synchronize()
        prof = f.getProfile()
    kmers = n * (L - k + 1)
    ih, qh = prof["insert_hash"], prof["query_hash"]
    print("k=%d L=%d: %.2e k-mers; pass A insert %.2f ms (%d launches) = %.2f ms per 1e9 k-mers, query %.2f ms per 1e9" % (
        k, L, kmers, ih[0], ih[1], ih[0] / (kmers / 1e9), qh[0] / (kmers / 1e9)), flush=True)
    del f, reads

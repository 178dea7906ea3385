"""diagnostic: one AUTO query of 10^8 reads with 1 % foreign reads against a 2^39-bit filter (the split path: sampler,
compaction, partitioned + gather kernels, merge); run under `rocprofv3 --kernel-trace --stats` for per-kernel times."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import btl_bloomfilter_amd as m

n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000, 150
f = m.BloomFilter(1 << 39, 4, 31)
reads = m.synth_reads_device(42, 0, n, L)
f.insertSeqs(reads, read_len=L)
foreign = n // 100
q = reads.clone()
idx = torch.arange(foreign, device="cuda") * 100 + 7
q.view(n, L)[idx] = m.synth_reads_device(43, 0, foreign, L).view(foreign, L)
f.setProfiling(True)
for rep in range(2):
    f.getProfile()
    hit, valid, cnt = f.containsSeqs(q, read_len=L, want_valid=True, want_counts=True)
    torch.cuda.synchronize()
    print({k: round(v[0], 2) for k, v in f.getProfile().items()}, cnt.tolist())

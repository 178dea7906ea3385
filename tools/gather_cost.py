"""Per-GPU compute of ShardedBloomFilter's gather mode, measured on one GPU: shard 0 of W (2^39 bits
local) is handed W x n_reads reads -- what every rank hashes per pass at world size W -- and times
insert + contains.  The read gather itself (1 byte per base over xGMI) is not part of this.
usage: python tools/gather_cost.py [W ...]"""
import sys
import time

import torch

import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import btl_bloomfilter_amd as m
from btl_bloomfilter_amd.sharded import HipShardOps

K, H, L, N = 31, 4, 150, 100_000_000
for W in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    ops = HipShardOps((1 << 39) * W, H, K, 0, W, 0)
    reads = m.synth_reads_device(42, 0, N * W, L)
    hit = torch.empty((reads.numel() + 63) // 64, dtype=torch.int64, device="cuda")
    valid = torch.empty_like(hit)
    res = []
    for it in range(3):
        ops.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        own, peers = reads[: N * L], reads[N * L:]  # the two calls of a round: own chunk, then the peers'
        ops.insert_seqs(own, L)
        if W > 1:
            ops.insert_seqs(peers, L)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ops.contains_seqs(own, L, hit, valid)
        if W > 1:
            w0 = N * L // 64
            ops.contains_seqs(peers, L, hit[w0:], valid[w0:])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append((t1 - t0, t2 - t1))
    kmers = N * (L - K + 1)
    ins, qry = min(r[0] for r in res), min(r[1] for r in res)
    own = ops.popcount_bits(hit)
    print("W=%d: insert %.3f s, contains %.3f s per pass of %d x %.1fe9 k-mers -> %.1f Gk-mers/s per GPU of own reads "
          "(insert+query); partial hits %d of %d" % (W, ins, qry, W, kmers / 1e9, 2 * kmers / (ins + qry) / 1e9, own,
                                                      kmers * W), flush=True)
    ops.close()
    del reads, hit, valid
    torch.cuda.empty_cache()

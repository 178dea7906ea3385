"""diagnostic: device-resident hash rows, raw k-mers and insertAndCheck through the direct kernels (2^39-bit filter):
    python tools/rows_probe.py   (random atomics / gathers: the ceilings are tools/microbench.py's)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import btl_bloomfilter_amd as m

ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]


def timed(fn, reps=2):
    for _ in range(reps):
        ev[0].record()
        out = fn()
        ev[1].record()
        torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]), out


f = m.KmerBloomFilter(1 << 39, 4, 31)
n = 500_000_000
rows = torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda")
ms, _ = timed(lambda: f.insert(rows))
print("insert(hash rows, device): %.1f ms for %.1e rows -> %.1f G rows/s (%.1f G probes/s)" % (ms, n, n / ms / 1e6, 4 * n / ms / 1e6))
ms, hit = timed(lambda: f.contains(rows))
print("contains(hash rows): %.1f ms -> %.1f G rows/s; hits %d" % (ms, n / ms / 1e6, int(hit.sum())))
other = torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda")
ms, hit = timed(lambda: f.contains(other))
print("contains(foreign rows): %.1f ms -> %.1f G rows/s; hits %d" % (ms, n / ms / 1e6, int(hit.sum())))
del rows, other
nk = 300_000_000
reads = m.synth_reads_device(42, 0, nk * 31 // 155 + 1, 155)[: nk * 31]
ms, _ = timed(lambda: f.insertKmers(reads))
print("insertKmers(raw k-mers, device): %.1f ms for %.1e k-mers -> %.1f Gk-mers/s" % (ms, nk, nk / ms / 1e6))
ms, hit = timed(lambda: f.containsKmers(reads))
print("containsKmers: %.1f ms -> %.1f Gk-mers/s" % (ms, nk / ms / 1e6))
g = m.BloomFilter(1 << 39, 4, 31)
r2 = m.synth_reads_device(7, 0, 20_000_000, 150)
ms, out = timed(lambda: g.insertAndCheckSeqs(r2, read_len=150, want_valid=False, want_counts=True), reps=1)
print("insertAndCheckSeqs: %.1f ms for %.1e k-mers -> %.1f Gk-mers/s; %s" % (ms, 2.4e9, 2.4e9 / ms / 1e6, out[2].tolist()))

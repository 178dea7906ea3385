#!/usr/bin/env python3
"""Genome-shaped input (a few very long sequences, N runs, soft-masked lower case) through the ragged layout:
direct vs partitioned results and rates.   python tools/genome_probe.py [total_bases]"""
import json
import os as _os
import sys
import sys as _sys

import numpy as np
import torch

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import btl_bloomfilter_amd as m


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3, r


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000_000
    g = torch.Generator(device="cuda").manual_seed(5)
    seq = torch.randint(0, 4, (total,), device="cuda", generator=g, dtype=torch.uint8)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    seq = lut[seq.long()] if total <= 500_000_000 else torch.cat([lut[c.long()] for c in seq.split(250_000_000)])
    # soft-masked stretches and N runs (assembly gaps)
    rng = np.random.default_rng(7)
    for _ in range(200):
        a = int(rng.integers(0, total - 200_000))
        seq[a:a + int(rng.integers(1, 100_000))] |= 0x20
    for _ in range(50):
        a = int(rng.integers(0, total - 200_000))
        seq[a:a + int(rng.integers(1, 50_000))] = ord("N")
    # chromosome-like lengths: a few huge ones, some scaffolds, degenerate ones
    cuts = np.sort(rng.choice(np.arange(1, total), size=40, replace=False))
    starts = np.concatenate([[0], cuts[:20], cuts[20:21], cuts[20:21], cuts[21:], [total]]).astype(np.int64)
    ts = torch.from_numpy(starts).cuda()
    k, h, bits = 31, 4, 1 << 37
    out = {"bases": total, "sequences": len(starts) - 1, "longest": int(np.diff(starts).max())}
    res = {}
    for mode in ("direct", "partitioned"):
        f = m.BloomFilter(bits, h, k)
        f.setInsertMode(mode)
        f.setQueryMode(mode)
        f.insertSeqs(seq, starts=ts)  # warm-up at full size: the first pass pays for the scratch allocation (~1 s)
        f.clear()
        ti, _ = timed(lambda: f.insertSeqs(seq, starts=ts))
        tq, (hit, valid, cnt) = timed(lambda: f.containsSeqs(seq, starts=ts, want_counts=True))
        res[mode] = (f.getPop(), cnt.tolist(), hit, valid)
        out[mode] = {"insert_Mkmers_s": cnt.tolist()[0] / ti / 1e6, "query_Mkmers_s": cnt.tolist()[0] / tq / 1e6,
                     "pop": res[mode][0], "counts": res[mode][1]}
        if mode == "direct":
            ref = f
        else:
            out["bodies_equal"] = f.compare(ref) == (0, 0, 0) if hasattr(f, "compare") else None
    out["hits_equal"] = bool((res["direct"][2] == res["partitioned"][2]).all())
    out["valid_equal"] = bool((res["direct"][3] == res["partitioned"][3]).all())
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py's side_configs table alone (the other BASELINE configurations and the reference's everyday shapes):
    python tools/side_bench.py [name ...]     names: C1 C5 C2_3x2p37_bits C2_ragged C3_incrementAll C3_insert C2"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import btl_bloomfilter_amd as m

only = set(sys.argv[1:]) or None
dev = torch.device("cuda", 0)
reads = m.synth_reads_device(42, 0, bench.N_READS, bench.READ_LEN, device=0)
out = bench.side_configs(m, torch, reads, bench.N_READS, dev, only)
if only and "C2" in only:  # the headline geometry through the same harness
    f = m.BloomFilter(1 << 39, 4, 31)
    f.setProfiling(True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for rep in range(3):
        f.clear()
        ev[0].record()
        f.insertSeqs(reads, read_len=150)
        ev[1].record()
        _, _, c = f.containsSeqs(reads, read_len=150, want_valid=True, want_counts=True)
        ev[2].record()
        torch.cuda.synchronize()
        prof = f.getProfile(reset=True)
    km = bench.N_READS * 120
    out["C2"] = {"insert_Mkmers_s": km / ev[0].elapsed_time(ev[1]) / 1e3, "query_Mkmers_s": km / ev[1].elapsed_time(ev[2]) / 1e3,
                 "kernel_ms_per_launch": {k: round(v[0] / v[1], 2) for k, v in prof.items() if v[1]}, "hits": c.tolist()}
for k, v in out.items():
    print(k, json.dumps({a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items() if a not in ("note", "launches_per_pass")}), flush=True)

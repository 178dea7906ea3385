#!/usr/bin/env python3
"""Per-GPU compute of the routed path at the REAL C4 geometry (BASELINE config 4: 2^43 bits = 1 TiB on 8
GPUs, 1.25e8 reads per GPU), measured on ONE GPU that plays shard 0 of 8 (a 128 GiB shard).

Origin side, exactly as at N = 8: every batch of this GPU's reads is routed once per position window (two
windows of 2^42 positions, 1024 level-0 bins, 32-bit entries, pass A's WINDOW variant).
Owner side: at N = 8 a shard receives one block from each of the 8 origins, together one origin's worth of
probes.  Here the block this GPU routed to itself is replicated 8 times (same volume, same kernels: two split
levels down to 128 KiB segments, OR / test in LDS).  No bytes cross xGMI: this is the compute a rank must
hide the exchange behind, not a scaling measurement.

    python tools/c4_cost.py [reads=125000000] [batch_reads=25000000] > profiles/r02/c4_cost.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import btl_bloomfilter_amd as m
from btl_bloomfilter_amd.sharded import HipShardOps

K, H, L, W = 31, 4, 150, 8
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
batch_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 25_000_000
dev = torch.device("cuda", 0)
ops = HipShardOps(1 << 43, H, K, 0, W, 0)
n_win, spw = ops.route_windows()
reads = m.synth_reads_device(42, 0, n_reads, L)
batch = batch_reads * L
n_batches = -(-reads.numel() // batch)
ent_b, cnt_b = ops.route_plan(batch, L)
bins, regions, cap, gb = ops.route_geometry(batch, L)
send_ent = torch.empty(spw * ent_b, dtype=torch.uint8, device=dev)
send_cnt = torch.empty(spw * cnt_b, dtype=torch.uint8, device=dev)
recv_ent = torch.empty(W * ent_b, dtype=torch.uint8, device=dev)
recv_cnt = torch.empty(W * cnt_b, dtype=torch.uint8, device=dev)
spill = torch.empty(1 << 20, dtype=torch.int64, device=dev)
spill_count = torch.zeros(1, dtype=torch.int64, device=dev)
fail = torch.empty(4 << 20, dtype=torch.int64, device=dev)
fail_count = torch.zeros(1, dtype=torch.int64, device=dev)
hit = torch.zeros((reads.numel() + 63) // 64, dtype=torch.int64, device=dev)
cnt2 = torch.zeros(2, dtype=torch.int64, device=dev)
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
out = {"geometry": {"global_bits": "2^43", "shard_bytes": (1 << 43) // 8 // W, "windows": n_win, "shards_per_window": spw,
                    "level0_bins_per_shard": bins, "regions_per_bin": regions, "bins_per_group": gb,
                    "reads_per_gpu": n_reads, "batches_per_pass": n_batches, "jobs_per_pass": n_batches * n_win,
                    "block_bytes": ent_b}}
for query in (0, 1, 0, 1):  # the first round of each allocates the owner's scratch inside its events: warm-up
    if query == 0:
        ops.clear()
    t_route = t_apply = 0.0
    fail_count.zero_()
    cnt2.zero_()
    spans = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for bi in range(n_batches):
        chunk = reads[bi * batch: (bi + 1) * batch]
        view = hit[bi * batch // 64: bi * batch // 64 + (chunk.numel() + 63) // 64] if query else None
        for w in range(n_win):
            e = [ev() for _ in range(4)]
            e[0].record()
            spill_count.zero_()
            ops.route(chunk, L, batch, query, send_ent, send_cnt, view, None, cnt2 if w == 0 else None, spill, spill_count,
                      window=w)
            e[1].record()
            if w == 0:  # this GPU is shard 0 of window 0: its own block, eight times over (copies not timed)
                recv_ent.view(W, ent_b).copy_(send_ent.view(spw, ent_b)[0].expand(W, ent_b))
                recv_cnt.view(W, cnt_b).copy_(send_cnt.view(spw, cnt_b)[0].expand(W, cnt_b))
            e[2].record()
            if w == 0:
                ops.apply_routed(recv_ent, recv_cnt, W, batch, L, query, fail, fail_count)
            e[3].record()
            spans.append(e)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    for e in spans:
        t_route += e[0].elapsed_time(e[1]) * 1e-3
        t_apply += e[2].elapsed_time(e[3]) * 1e-3
    kmers = n_reads * (L - K + 1)
    out["query" if query else "insert"] = {
        "route_s": t_route, "apply_s": t_apply, "compute_s": t_route + t_apply, "wall_s_incl_block_copies": wall,
        "Gkmers_s_per_gpu": kmers / (t_route + t_apply) / 1e9, "spilled": int(spill_count.item()),
        "failed_positions": int(fail_count.item()) if query else None}
ins, qry = out["insert"], out["query"]
out["step"] = {"compute_s": ins["compute_s"] + qry["compute_s"],
               "Gkmers_s_per_gpu": 2 * n_reads * (L - K + 1) / (ins["compute_s"] + qry["compute_s"]) / 1e9,
               "note": "insert + query of this GPU's reads at the C4 geometry; an N = 8 run adds the xGMI exchange of "
                       "4-byte entries (DESIGN.md section 6), overlapped with this compute"}
print(json.dumps(out, indent=1))

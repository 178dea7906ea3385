"""Shards that are handed ALL reads (the gather mode's per-rank compute, include/btlbf.h "btlbf_create_shard"):
each keeps the probes inside its window; their bodies concatenated must be the single filter's body and
the AND of their contains() bitmaps the single filter's answer.  Randomised geometries (shard counts 2..5,
power-of-two and other shard sizes, k, h, read lengths), direct and partitioned kernels, bit and
counting filters; the single filter is built by the direct kernels, which the oracle tests pin."""
import ctypes as C
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(14))
def test_window_shards_match_single_filter(seed):
    import torch

    import btl_bloomfilter_amd as m
    from btl_bloomfilter_amd import _lib
    from btl_bloomfilter_amd.sharded import HipShardOps

    rng = random.Random(1000 + seed)
    W = rng.choice([2, 3, 4, 5])
    counting = seed % 3 == 2
    h = rng.randint(1, 6)
    k = rng.choice([15, 25, 31, 33, 50])
    L = rng.choice([k, 100, 150, 251])
    per = rng.choice([1 << 20, 3 << 19, 64 * 5 * 977, 1 << 23, 1 << 25])  # positions per shard
    size = per * W
    n_reads = rng.randint(3000, 50000)
    mode = rng.choice([_lib.INSERT_DIRECT, _lib.INSERT_PARTITIONED, _lib.INSERT_AUTO])
    thr = 2 if counting else 0
    reads = m.synth_reads_device(42, 0, n_reads, L)
    again = reads[: (n_reads // 3) * L]  # counting: a third of the reads twice
    q = torch.cat([reads[: min(n_reads, 4000) * L], m.synth_reads_device(43, 0, 500, L)])

    if counting:
        ref = m.CountingBloomFilter(size, h, k, thr)
    else:
        ref = m.BloomFilter(size, h, k)
    ref.setInsertMode("direct")
    ref.setQueryMode("direct")
    if counting:
        ref.insertSeqs(reads, read_len=L, increment_all=True)
        ref.insertSeqs(again, read_len=L, increment_all=True)
    else:
        ref.insertSeqs(reads, read_len=L)
    eh, _, _ = ref.containsSeqs(q, read_len=L, want_valid=False, want_counts=True)
    body = ref.download()

    words = (q.numel() + 63) // 64
    acc = None
    got = []
    for r in range(W):
        ops = HipShardOps(size, h, k, r, W, 0, counting=counting, threshold=thr)
        _lib.check(ops.L.btlbf_set_insert_mode(ops.f, mode, 0))
        _lib.check(ops.L.btlbf_set_query_mode(ops.f, mode))
        ops.insert_seqs(reads, L)
        if counting:
            ops.insert_seqs(again, L)
        hit = torch.zeros(words, dtype=torch.int64, device="cuda")
        valid = torch.zeros(words, dtype=torch.int64, device="cuda")
        ops.contains_seqs(q, L, hit, valid)
        acc = hit if acc is None else acc & hit
        got.append(ops.local_body())
        ops.close()
    torch.cuda.synchronize()
    cfg = dict(W=W, counting=counting, h=h, k=k, L=L, per=per, n_reads=n_reads, mode=mode)
    assert (np.concatenate(got) == body).all(), cfg
    assert bool((acc == eh).all().item()), cfg

"""Helpers for the multi-rank tests: a CPU stand-in for the per-rank compute (built on the test-only
oracle) so that the routing logic of btl_bloomfilter_amd.sharded can run under gloo without a GPU,
and the worker functions spawned per rank."""
import hashlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class OracleShardOps:
    """same interface as sharded.HipShardOps, computed with oracle/btl_oracle.c on CPU tensors"""

    def __init__(self, global_bits, h, k, rank, world):
        from oracle.pyoracle import Oracle

        self.o = Oracle()
        self.bits, self.h, self.k, self.rank, self.world = global_bits, h, k, rank, world
        self.shard_len = global_bits // world
        self.body = np.zeros(self.shard_len // 8, np.uint8)
        self.device = torch.device("cpu")

    def clear(self):
        self.body[:] = 0

    def positions(self, reads, read_len, cap, want_tags):
        buf = reads.numpy()
        n = buf.size
        buckets = np.zeros((self.world, cap), np.int64)
        tags = np.zeros((self.world, cap), np.int64)
        counts = np.zeros(self.world, np.int64)
        valid = np.zeros(n, np.uint8)
        for r in range(n // read_len):
            pos, hv = self.o.nthash_seq(buf[r * read_len:(r + 1) * read_len].tobytes(), self.h, self.k)
            for p, row in zip(pos, hv):
                gp = r * read_len + int(p)
                valid[gp] = 1
                for i, x in enumerate(row):
                    q = int(x) % self.bits
                    own = q // self.shard_len
                    c = counts[own]
                    if c < cap:
                        buckets[own, c] = q - own * self.shard_len
                        tags[own, c] = gp * self.h + i
                    counts[own] += 1
        vb = np.packbits(np.concatenate([valid, np.zeros((-n) % 64, np.uint8)]), bitorder="little").view(np.int64)
        return (torch.from_numpy(buckets), torch.from_numpy(tags) if want_tags else None, torch.from_numpy(counts),
                torch.from_numpy(vb.copy()) if want_tags else None)

    def insert_positions(self, pos):
        p = pos.numpy().astype(np.uint64)
        np.bitwise_or.at(self.body, (p >> np.uint64(3)).astype(np.int64), (1 << (p & np.uint64(7))).astype(np.uint8))

    def test_positions(self, pos):
        p = pos.numpy().astype(np.uint64)
        return torch.from_numpy(((self.body[(p >> np.uint64(3)).astype(np.int64)] >> (p & np.uint64(7)).astype(np.uint8)) & 1).astype(np.uint8))

    def and_answers(self, tags, answers, hit_bits):
        t, a = tags.numpy(), answers.numpy()
        hb = hit_bits.numpy().view(np.uint64)
        for tag in t[a == 0]:
            p = int(tag) // self.h
            hb[p >> 6] &= ~np.uint64(1 << (p & 63))

    def popcount_bits(self, bits):
        return int(np.unpackbits(bits.numpy().view(np.uint8)).sum())

    # gather mode: all reads in, only the probes inside this shard's window count
    def _own(self, buf, read_len):
        lo = self.rank * self.shard_len
        for r in range(buf.size // read_len):
            pos, hv = self.o.nthash_seq(buf[r * read_len:(r + 1) * read_len].tobytes(), self.h, self.k)
            for p, row in zip(pos, hv):
                q = np.array([int(x) % self.bits for x in row], np.int64) - lo
                yield r * read_len + int(p), q[(q >= 0) & (q < self.shard_len)]

    def insert_seqs(self, reads, read_len):
        for _, q in self._own(reads.numpy(), read_len):
            np.bitwise_or.at(self.body, q >> 3, (1 << (q & 7)).astype(np.uint8))

    def contains_seqs(self, reads, read_len, hit_bits, valid_bits):
        n = reads.numel()
        hit = np.zeros(n + (-n) % 64, np.uint8)
        valid = np.zeros_like(hit)
        for gp, q in self._own(reads.numpy(), read_len):
            valid[gp] = 1
            hit[gp] = bool((((self.body[q >> 3] >> (q & 7).astype(np.uint8)) & 1) == 1).all())
        for dst, src in ((hit_bits, hit), (valid_bits, valid)):
            if dst is None:
                continue
            dst[: hit.size // 64] = torch.from_numpy(np.packbits(src, bitorder="little").view(np.int64).copy())

    def local_body(self):
        return self.body.copy()


def _init(rank, world, port, backend="gloo"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend, rank=rank, world_size=world)


def cpu_worker(rank, world, port, outdir, bits, h, k, n_reads, read_len, mode="exchange", uneven=0):
    """gloo + oracle stand-in: validates bucketing / all-to-all / answer routing (mode="gather": the
    read gather, the window-partial answers and their AND at the reads' owner).  uneven: rank r holds
    n_reads - r * uneven reads, so the ranks run out of batches at different times"""
    from oracle.pyoracle import Oracle

    from btl_bloomfilter_amd.sharded import ShardedBloomFilter

    _init(rank, world, port)
    o = Oracle()
    ops = OracleShardOps(bits, h, k, rank, world)
    f = ShardedBloomFilter(bits, h, k, ops=ops, batch_reads=64, mode=mode,
                           batch_bytes_cap=(64 * read_len if mode == "gather" else 0))
    if os.environ.get("BTLBF_TEST_MSG_BYTES"):  # force the sliced exchange (messages above this size)
        f.MSG_BYTES = int(os.environ["BTLBF_TEST_MSG_BYTES"])
    n_mine = n_reads - rank * uneven
    mine = torch.from_numpy(o.synth_reads(42, rank * n_reads, n_mine, read_len))
    f.insert_reads(mine, read_len)
    dist.barrier()
    np.save(os.path.join(outdir, "body%d.npy" % rank), ops.local_body())
    # query: own reads (hits) followed by reads nobody inserted
    q = torch.from_numpy(np.concatenate([o.synth_reads(42, rank * n_reads, n_mine, read_len),
                                         o.synth_reads(43, rank * n_reads, n_mine, read_len)]))
    hit = torch.zeros((q.numel() + 63) // 64, dtype=torch.int64)
    cnt = torch.zeros(2, dtype=torch.int64)
    f.contains_reads(q, read_len, hit, cnt)
    np.save(os.path.join(outdir, "hit%d.npy" % rank), hit.numpy())
    np.save(os.path.join(outdir, "cnt%d.npy" % rank), cnt.numpy())
    dist.barrier()
    dist.destroy_process_group()


def gpu_worker(rank, world, port, outdir, bits, h, k, n_reads, read_len):
    """gloo (host-staged exchange) + the real HIP kernels, all ranks on cuda:0"""
    import btl_bloomfilter_amd as m
    from btl_bloomfilter_amd.sharded import ShardedBloomFilter

    torch.cuda.set_device(0)
    _init(rank, world, port)
    f = ShardedBloomFilter(bits, h, k, device=0, batch_reads=4096, mode="exchange")
    if os.environ.get("BTLBF_TEST_MSG_BYTES"):  # force the sliced exchange (messages above this size)
        f.MSG_BYTES = int(os.environ["BTLBF_TEST_MSG_BYTES"])
    mine = m.synth_reads_device(42, rank * n_reads, n_reads, read_len)
    f.insert_reads(mine, read_len)
    torch.cuda.synchronize()
    dist.barrier()
    np.save(os.path.join(outdir, "body%d.npy" % rank), f.ops.local_body())
    f.store(os.path.join(outdir, "sharded.bf"))
    q = torch.cat([mine, m.synth_reads_device(43, rank * n_reads, n_reads, read_len)])
    hit = torch.zeros((q.numel() + 63) // 64, dtype=torch.int64, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int64)
    f.contains_reads(q, read_len, hit, cnt)
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, "hit%d.npy" % rank), hit.cpu().numpy())
    np.save(os.path.join(outdir, "cnt%d.npy" % rank), cnt.numpy())
    dist.barrier()
    dist.destroy_process_group()


def gpu_worker_routed(rank, world, port, outdir, bits, h, k, n_reads, read_len, route_bins=0, pipeline=None,
                      mode="exchange", window_bits=0, spill_cap=0, skew=0):
    """the routed (partitioned) multi-GPU path -- or, mode="gather", the gather path -- with the real HIP
    kernels, all ranks on cuda:0.  window_bits: BTLBF_ROUTE_WINDOW_BITS (several position windows on a
    small filter); spill_cap: size of the per-job spill lists; skew: copies of one read appended to every
    rank's reads (hot positions that cannot be staged at the origin)"""
    if route_bins:
        os.environ["BTLBF_ROUTE_BINS"] = str(route_bins)
    if window_bits:
        os.environ["BTLBF_ROUTE_WINDOW_BITS"] = str(window_bits)
    import btl_bloomfilter_amd as m
    from btl_bloomfilter_amd.sharded import ShardedBloomFilter

    torch.cuda.set_device(0)
    _init(rank, world, port)
    # several batches; pipeline=True runs the double-buffered schedule of the RCCL path over gloo
    f = ShardedBloomFilter(bits, h, k, device=0, batch_bytes_cap=4 << 20, pipeline=pipeline, mode=mode)
    if os.environ.get("BTLBF_TEST_MSG_BYTES"):  # force the sliced exchange (messages above this size)
        f.MSG_BYTES = int(os.environ["BTLBF_TEST_MSG_BYTES"])
    assert f.mode == mode and (mode == "gather" or f._routed())
    if spill_cap:
        f.SPILL_CAP = spill_cap
    if window_bits:
        assert f.ops.route_windows()[0] == bits >> window_bits
    mine = m.synth_reads_device(42, rank * n_reads, n_reads, read_len)
    hot = m.synth_reads_device(45, 0, 1, read_len).repeat(skew) if skew else None
    if skew:
        mine = torch.cat([mine, hot])
    f.insert_reads(mine, read_len)
    torch.cuda.synchronize()
    dist.barrier()
    np.save(os.path.join(outdir, "body%d.npy" % rank), f.ops.local_body())
    # reference: the whole filter on this GPU, direct kernels
    ref = m.BloomFilter(bits, h, k)
    ref.setInsertMode("direct")
    ref.setQueryMode("direct")
    ref.insertSeqs(m.synth_reads_device(42, 0, world * n_reads, read_len), read_len=read_len)
    if skew:
        ref.insertSeqs(hot[:read_len], read_len=read_len)
    res = {}
    mine_body = f.ops.local_body()
    whole = ref.download()
    per = whole.size // world
    res["body"] = (bool((whole[rank * per:(rank + 1) * per] == mine_body).all()), [0, 0], [0, 0])
    for name, q in (("hits", mine.clone()),
                    ("few_misses", mine.clone()),
                    ("many_misses", torch.cat([mine[: 3000 * read_len], m.synth_reads_device(43, rank * 50000, 50000, read_len)]))):
        if name == "few_misses":
            q.view(-1, read_len)[torch.arange(7, device="cuda") * 1000 + rank] = m.synth_reads_device(44, rank * 7, 7, read_len).view(7, read_len)
        hit = torch.zeros((q.numel() + 63) // 64, dtype=torch.int64, device="cuda")
        cnt = torch.zeros(2, dtype=torch.int64)
        f.contains_reads(q, read_len, hit, cnt)
        eh, _, ec = ref.containsSeqs(q, read_len=read_len, want_valid=False, want_counts=True)
        torch.cuda.synchronize()
        res[name] = (bool((hit == eh).all().item()), cnt.tolist(), ec.cpu().tolist())
    np.save(os.path.join(outdir, "res%d.npy" % rank), np.array([repr(res)]))
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def gpu_worker_counting(rank, world, port, outdir, counters, h, k, thr, n_reads, read_len, mode="exchange", twice=True):
    """sharded CountingBloomFilter (incrementAll + contains) on the routed path or in gather mode, all ranks
    on cuda:0"""
    import btl_bloomfilter_amd as m
    from btl_bloomfilter_amd.sharded import ShardedBloomFilter

    torch.cuda.set_device(0)
    _init(rank, world, port)
    f = ShardedBloomFilter(counters, h, k, device=0, batch_bytes_cap=4 << 20, pipeline=True, counting=True,
                           threshold=thr, mode=mode)
    assert f.mode == mode
    if os.environ.get("BTLBF_TEST_MSG_BYTES"):  # force the sliced exchange (messages above this size)
        f.MSG_BYTES = int(os.environ["BTLBF_TEST_MSG_BYTES"])
    mine = m.synth_reads_device(42, rank * n_reads, n_reads, read_len)
    f.insert_reads(mine, read_len)
    if twice == "golden":
        # the reference fixture inserts reads [0, n) and then the first n/2 again: with two ranks that is all of
        # rank 0's reads a second time and none of rank 1's (which still takes part, with an empty buffer)
        f.insert_reads(mine if rank == 0 else mine[:0], read_len)
    elif twice:
        f.insert_reads(mine[: (n_reads // 2) * read_len], read_len)  # half of the reads a second time
    torch.cuda.synchronize()
    dist.barrier()
    np.save(os.path.join(outdir, "body%d.npy" % rank), f.ops.local_body())
    # reference: the whole counting filter on this GPU, direct kernels, the same multiset of reads
    ref = m.CountingBloomFilter(counters, h, k, thr)
    ref.setInsertMode("direct")
    ref.setQueryMode("direct")
    for r in range(world):
        rr = m.synth_reads_device(42, r * n_reads, n_reads, read_len)
        ref.insertSeqs(rr, read_len=read_len, increment_all=True)
        if twice == "golden":
            if r == 0:
                ref.insertSeqs(rr, read_len=read_len, increment_all=True)
        elif twice:
            ref.insertSeqs(rr[: (n_reads // 2) * read_len], read_len=read_len, increment_all=True)
    # mostly reads inserted twice (pass the threshold), a few inserted once and a few foreign ones
    half = (n_reads // 2) * read_len
    q = torch.cat([mine[: half + 50 * read_len], m.synth_reads_device(43, rank * 20, 20, read_len)])
    hit = torch.zeros((q.numel() + 63) // 64, dtype=torch.int64, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int64)
    if twice is True:
        f.contains_reads(q, read_len, hit, cnt)
        eh, _, ec = ref.containsSeqs(q, read_len=read_len, want_valid=False, want_counts=True)
        torch.cuda.synchronize()
        res = (bool((hit == eh).all().item()), cnt.tolist(), ec.cpu().tolist())
    else:  # the golden run is about the counters
        res = (True, [0, 0], [0, 0])
    np.save(os.path.join(outdir, "res%d.npy" % rank), np.array([repr(res)]))
    f.store(os.path.join(outdir, "sharded.bf"))  # every rank writes its counter range into the one file
    if rank == 0:
        np.save(os.path.join(outdir, "ref.npy"), ref.download())
        ref.storeFilter(os.path.join(outdir, "ref.bf"))
    dist.barrier()
    dist.destroy_process_group()

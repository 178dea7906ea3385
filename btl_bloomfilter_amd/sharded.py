"""Hash-range sharded Bloom filter over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).  The M-bit filter is cut
into `world` contiguous bit ranges; rank g holds bits [g*M/W, (g+1)*M/W) in its HBM.  Probe
positions are `hash % M` exactly as in the single-GPU filter (BloomFilter.hpp:190), so the shard
bodies concatenated in rank order ARE the single-filter body (and the .bf file).

    insert : every rank hashes its own reads (fused ntHash kernel) and buckets the h positions of
             each k-mer by owning shard; ONE all-to-all moves shard-local positions to their owners;
             owners atomicOr them into their bit range.
    query  : all-to-all of positions out, owners test bits, all-to-all of one byte per probe back in
             the same order; the origin ANDs the h answers of each k-mer into the per-window bitmap.

The exchange is the only collective on the data path and it is a real data dependency: the h probes
of one k-mer land on different shards.  Work is cut into batches of reads so that bucket memory
stays bounded; the split sizes of each all-to-all are the bucket fills.

`ops` abstracts the per-rank compute: HipShardOps (the C ABI / HIP kernels) in production; the
tests substitute a CPU stand-in to exercise this routing logic under gloo with world_size 2."""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib


class HipShardOps:
    """per-rank compute through the C ABI (HIP kernels); tensors live on cuda:<device>"""

    def __init__(self, global_bits, hash_num, kmer_size, rank, world, device):
        self.L = _lib.load()
        self.h = hash_num
        self.k = kmer_size
        self.world = world
        self.device_index = device
        self.device = torch.device("cuda", device)
        hnd = C.c_void_p()
        _lib.check(self.L.btlbf_create_shard(C.byref(hnd), _lib.BLOOM, global_bits, rank, world, hash_num,
                                             kmer_size, 0, device))
        self.f = hnd

    def close(self):
        if self.f:
            self.L.btlbf_destroy(self.f)
            self.f = None

    def _sp(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def clear(self):
        _lib.check(self.L.btlbf_clear(self.f, self._sp()))

    def positions(self, reads, read_len, cap, want_tags):
        n = reads.numel()
        lay = _lib.Layout()
        lay.starts, lay.n_seqs, lay.read_len = None, 0, read_len
        buckets = torch.empty((self.world, cap), dtype=torch.int64, device=self.device)
        tags = torch.empty((self.world, cap), dtype=torch.int64, device=self.device) if want_tags else None
        counts = torch.zeros(self.world, dtype=torch.int64, device=self.device)
        valid = torch.zeros((n + 63) // 64, dtype=torch.int64, device=self.device) if want_tags else None
        _lib.check(self.L.btlbf_positions_seqs(
            self.f, C.c_void_p(reads.data_ptr()), n, C.byref(lay), self.world, C.c_void_p(buckets.data_ptr()),
            C.c_void_p(tags.data_ptr()) if want_tags else None, cap, C.c_void_p(counts.data_ptr()),
            C.c_void_p(valid.data_ptr()) if want_tags else None, self._sp()))
        return buckets, tags, counts, valid

    def insert_positions(self, pos):
        _lib.check(self.L.btlbf_insert_positions(self.f, C.c_void_p(pos.data_ptr()), pos.numel(), self._sp()))

    def test_positions(self, pos):
        out = torch.empty(pos.numel(), dtype=torch.uint8, device=self.device)
        _lib.check(self.L.btlbf_test_positions(self.f, C.c_void_p(pos.data_ptr()), pos.numel(),
                                               C.c_void_p(out.data_ptr()), self._sp()))
        return out

    def and_answers(self, tags, answers, hit_bits):
        _lib.check(self.L.btlbf_and_answers(C.c_void_p(tags.data_ptr()), C.c_void_p(answers.data_ptr()),
                                            tags.numel(), self.h, C.c_void_p(hit_bits.data_ptr()),
                                            self.device_index, self._sp()))

    def popcount_bits(self, bits):
        n = bits.numel() * bits.element_size()
        pad = (-n) % 16
        if pad:
            bits = torch.cat([bits.view(torch.uint8), torch.zeros(pad, dtype=torch.uint8, device=bits.device)])
            n += pad
        out = C.c_uint64()
        _lib.check(self.L.btlbf_popcount_bits(C.c_void_p(bits.data_ptr()), n, C.byref(out), self.device_index,
                                              self._sp()))
        return out.value

    def local_body(self):
        import numpy as np

        n = self.L.btlbf_local_bytes(self.f)
        out = np.zeros(n, np.uint8)
        _lib.check(self.L.btlbf_download(self.f, C.c_void_p(out.ctypes.data), 0, n))
        return out

    def store_shard(self, path):
        _lib.check(self.L.btlbf_store_shard(self.f, str(path).encode()))


class ShardedBloomFilter:
    def __init__(self, global_bits, hash_num, kmer_size, device=0, group=None, ops=None, batch_reads=2_000_000,
                 slack=1.25):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if global_bits % (64 * self.world):
            raise ValueError("filter bits must split into multiples of 64 per shard")
        self.bits = global_bits
        self.h = hash_num
        self.k = kmer_size
        self.ops = ops if ops is not None else HipShardOps(global_bits, hash_num, kmer_size, self.rank, self.world,
                                                           device)
        self.batch_reads = batch_reads
        self.slack = slack
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
        # gloo moves host memory: stage device tensors through the CPU (test mode only)
        self.stage_cpu = backend == "gloo"

    # ---- the exchange -----------------------------------------------------------------------
    def _all_to_all(self, send, send_counts, recv_counts):
        """variable-size all-to-all of a flat tensor; counts are python ints per peer"""
        if self.world == 1:
            return send
        dev = send.device
        if self.stage_cpu and send.is_cuda:
            send = send.cpu()
        recv = torch.empty(sum(recv_counts), dtype=send.dtype, device=send.device)
        dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=send_counts,
                               group=self.group)
        return recv.to(dev) if recv.device != dev else recv

    def _exchange_counts(self, counts):
        if self.world == 1:
            return counts
        c = counts.cpu() if (self.stage_cpu and counts.is_cuda) else counts
        out = torch.empty_like(c)
        dist.all_to_all_single(out, c, group=self.group)
        return out

    def _bucketed(self, reads, read_len, want_tags):
        n_kmers = (reads.numel() // read_len) * max(read_len - self.k + 1, 0)
        cap = int(n_kmers * self.h / self.world * self.slack) + 4096
        while True:
            buckets, tags, counts, valid = self.ops.positions(reads, read_len, cap, want_tags)
            cnt = counts.cpu().tolist()  # split sizes must be host integers
            if max(cnt) <= cap:
                break
            cap = max(cnt) + 4096  # skewed batch: retry once with the exact capacity
        send = torch.cat([buckets[s, : cnt[s]] for s in range(self.world)])
        stags = torch.cat([tags[s, : cnt[s]] for s in range(self.world)]) if want_tags else None
        recv_cnt = self._exchange_counts(counts).cpu().tolist()
        return send, stags, cnt, recv_cnt, valid

    def _batches(self, reads, read_len):
        step = self.batch_reads * read_len
        for off in range(0, reads.numel(), step):
            yield off, reads[off: off + step]

    # ---- public -----------------------------------------------------------------------------
    def clear(self):
        self.ops.clear()

    def insert_reads(self, reads, read_len):
        """insertSeq over this rank's uniform-length reads (flat uint8 tensor)"""
        for _, chunk in self._batches(reads, read_len):
            send, _, cnt, recv_cnt, _ = self._bucketed(chunk, read_len, False)
            mine = self._all_to_all(send, cnt, recv_cnt)
            self.ops.insert_positions(mine)

    def contains_reads(self, reads, read_len, hit_bits, counts=None):
        """contains() of every window of this rank's reads -> hit_bits (int64 bitmap over the whole
        buffer, bit p = window at byte p); counts (optional int64[2]) = {clean windows, hits}"""
        n_valid = n_hit = 0
        for off, chunk in self._batches(reads, read_len):
            send, stags, cnt, recv_cnt, valid = self._bucketed(chunk, read_len, True)
            theirs = self._all_to_all(send, cnt, recv_cnt)
            answers = self.ops.test_positions(theirs)
            back = self._all_to_all(answers, recv_cnt, cnt)  # same order as `send`
            words = valid.numel()
            w0 = off // 64  # batches are multiples of 64 bytes as long as batch_reads*read_len % 64 == 0
            if off % 64:
                raise ValueError("batch_reads * read_len must be a multiple of 64")
            view = hit_bits[w0: w0 + words]
            view.copy_(valid)
            self.ops.and_answers(stags, back, view)
            if counts is not None:
                n_valid += self.ops.popcount_bits(valid)
                n_hit += self.ops.popcount_bits(view)
        if counts is not None:
            counts[0] = n_valid
            counts[1] = n_hit
        return hit_bits

    def store(self, path):
        """every rank writes its bit range into the one BTLBloomFilter_v1 file"""
        self.ops.store_shard(path)
        if dist.is_initialized():
            dist.barrier(self.group)

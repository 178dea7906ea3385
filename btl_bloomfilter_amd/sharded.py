"""Hash-range sharded Bloom filter over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).  The M-bit filter is cut
into `world` contiguous bit ranges; rank g holds bits [g*M/W, (g+1)*M/W) in its HBM.  Probe
positions are `hash % M` exactly as in the single-GPU filter (BloomFilter.hpp:190), so the shard
bodies concatenated in rank order ARE the single-filter body (and the .bf file).

Three data paths:
  gather (default for 2..4 ranks; any geometry; bit and counting filters): reads move, probes do not.
    Every rank sends its reads to all peers (1 byte per base), hashes its own chunk while they travel
    and then the peers' chunks, and keeps the probes that fall into its own bit range (pass A's WINDOW
    variant; passes B and C then see the shard's 1/W of the probes).  A query answers "every probe of
    this window inside my range is set" per shard; the partial bitmaps go back to the reads' owners
    and are ANDed.  W times the hashing for 1/8 (W = 2) to 1/12 (W = 4) of the traffic: xGMI is one link per GPU
    pair, so the probe exchange below is link-bound at small W.
  routed (8 ranks; power-of-two geometries, large batches): the global position space is cut into
    512 bins; each rank hashes its reads and radix-partitions the probe positions into those bins in
    LDS (4-byte entries, 128-byte chunks); the block of bins a shard owns is contiguous, so ONE
    fixed-size all-to-all moves it; the owner splits the received bins down to 64 KiB segments and
    ORs / tests them in LDS.  A query sends nothing back but the few positions found clear
    (all-gather of small fail lists); the origin clears the windows that own one of them.
  direct (any geometry; the stand-in ops of the CPU tests):
    insert : every rank buckets the h positions of each k-mer by owning shard; ONE all-to-all moves
             shard-local positions to their owners; owners atomicOr them into their bit range.
    query  : all-to-all of positions out, owners test bits, all-to-all of one byte per probe back in
             the same order; the origin ANDs the h answers of each k-mer into the per-window bitmap.

The exchange (of reads or of probes) is the only collective on the data path and it is a real data
dependency: the h probes of one k-mer land on different shards.  Work is cut into batches of reads so
that buffer memory stays bounded.

`ops` abstracts the per-rank compute: HipShardOps (the C ABI / HIP kernels) in production; the
tests substitute a CPU stand-in to exercise this routing logic under gloo with world_size 2."""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib


class HipShardOps:
    """per-rank compute through the C ABI (HIP kernels); tensors live on cuda:<device>"""

    def __init__(self, global_bits, hash_num, kmer_size, rank, world, device, counting=False, threshold=0):
        """global_bits: bits of the whole filter, or its uint8_t counters when counting"""
        self.L = _lib.load()
        self.h = hash_num
        self.k = kmer_size
        self.world = world
        self.counting = bool(counting)
        self.device_index = device
        self.device = torch.device("cuda", device)
        hnd = C.c_void_p()
        _lib.check(self.L.btlbf_create_shard(C.byref(hnd), _lib.COUNTING8 if counting else _lib.BLOOM, global_bits,
                                             rank, world, hash_num, kmer_size, threshold, device))
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

    # ---- gather mode: the shard is handed every rank's reads and keeps the probes inside its window ----
    def insert_seqs(self, reads, read_len):
        lay = self._lay(read_len)
        _lib.check(self.L.btlbf_insert_seqs(self.f, C.c_void_p(reads.data_ptr()), reads.numel(), C.byref(lay),
                                            _lib.INCREMENT_ALL, _lib.ORDER_PARALLEL, _lib.DEVICE, self._sp()))

    def contains_seqs(self, reads, read_len, hit_bits, valid_bits):
        """hit bit p = window p is clean and every probe of it that falls into this shard is set"""
        lay = self._lay(read_len)
        _lib.check(self.L.btlbf_contains_seqs(self.f, C.c_void_p(reads.data_ptr()), reads.numel(), C.byref(lay),
                                              C.c_void_p(hit_bits.data_ptr()),
                                              C.c_void_p(valid_bits.data_ptr()) if valid_bits is not None else None,
                                              None, _lib.DEVICE, self._sp()))

    # ---- routing on the partitioned pipeline (btlbf_route_* / btlbf_apply_routed) ----
    def _lay(self, read_len):
        lay = _lib.Layout()
        lay.starts, lay.n_seqs, lay.read_len = None, 0, read_len
        return lay

    def route_supported(self, global_bits, world):
        """power-of-two geometry of at least 512 bins x one 64 KiB segment; filters above 2^42 positions are
        routed in windows of 2^42 and need at least one shard per window (btlbf_route_windows)"""
        pow2 = lambda x: x > 0 and (x & (x - 1)) == 0  # noqa: E731
        if not (pow2(global_bits) and pow2(world) and 1 <= self.h <= 8):
            return False
        nw, spw = C.c_uint(), C.c_uint()
        return self.L.btlbf_route_windows(self.f, world, C.byref(nw), C.byref(spw)) == _lib.OK

    def route_windows(self):
        """(position windows, shards per window): window w is owned by shards [w * spw, (w + 1) * spw)"""
        nw, spw = C.c_uint(), C.c_uint()
        _lib.check(self.L.btlbf_route_windows(self.f, self.world, C.byref(nw), C.byref(spw)))
        return nw.value, spw.value

    def route_plan(self, plan_len, read_len):
        e, c = C.c_uint64(), C.c_uint64()
        lay = self._lay(read_len)
        _lib.check(self.L.btlbf_route_plan(self.f, plan_len, C.byref(lay), self.world, C.byref(e), C.byref(c)))
        return e.value, c.value

    def route(self, reads, read_len, plan_len, query, send_ent, send_cnt, hit, valid, counts, spill, spill_count,
              window=0):
        lay = self._lay(read_len)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        _lib.check(self.L.btlbf_route_seqs(self.f, ptr(reads), reads.numel(), C.byref(lay), plan_len, self.world,
                                           int(window), int(query), ptr(send_ent), ptr(send_cnt), ptr(hit), ptr(valid),
                                           ptr(counts), ptr(spill), spill.numel(), ptr(spill_count), self._sp()))

    def apply_routed(self, recv_ent, recv_cnt, n_blocks, plan_len, read_len, query, fail_list, fail_count):
        lay = self._lay(read_len)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        _lib.check(self.L.btlbf_apply_routed(self.f, ptr(recv_ent), ptr(recv_cnt), n_blocks, plan_len, C.byref(lay),
                                             self.world, int(query), ptr(fail_list),
                                             fail_list.numel() if fail_list is not None else 0, ptr(fail_count),
                                             self._sp()))

    def route_geometry(self, plan_len, read_len):
        """(level-0 bins per shard, regions per bin, 128-byte chunks per region, bins per group)"""
        out = (C.c_uint32 * 4)()
        lay = self._lay(read_len)
        _lib.check(self.L.btlbf_route_geometry(self.f, plan_len, C.byref(lay), self.world, self.world, out))
        return tuple(int(v) for v in out)

    def owner_scratch_bytes(self, plan_len, read_len):
        """bytes of scratch apply_routed[_bins] allocates inside the filter for batches of plan_len bytes of reads"""
        out = C.c_uint64()
        lay = self._lay(read_len)
        _lib.check(self.L.btlbf_owner_scratch_bytes(self.f, plan_len, C.byref(lay), self.world, self.world, C.byref(out)))
        return out.value

    def apply_routed_bins(self, recv_ent, recv_cnt, n_blocks, first_bin, n_bins, plan_len, read_len, query, fail_list,
                          fail_count):
        lay = self._lay(read_len)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        _lib.check(self.L.btlbf_apply_routed_bins(self.f, ptr(recv_ent), ptr(recv_cnt), n_blocks, first_bin, n_bins,
                                                  plan_len, C.byref(lay), self.world, int(query), ptr(fail_list),
                                                  fail_list.numel() if fail_list is not None else 0, ptr(fail_count),
                                                  self._sp()))

    def apply_spill(self, pos, query, fail_list, fail_count):
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        _lib.check(self.L.btlbf_apply_spill(self.f, ptr(pos), pos.numel(), int(query), ptr(fail_list),
                                            fail_list.numel() if fail_list is not None else 0, ptr(fail_count),
                                            self._sp()))

    def resolve(self, reads, read_len, fails, hit_bits):
        lay = self._lay(read_len)
        _lib.check(self.L.btlbf_resolve_seqs(self.f, C.c_void_p(reads.data_ptr()), reads.numel(), C.byref(lay),
                                             C.c_void_p(fails.data_ptr()), fails.numel(), C.c_void_p(hit_bits.data_ptr()),
                                             self._sp()))

    def local_body(self):
        import numpy as np

        n = self.L.btlbf_local_bytes(self.f)
        out = np.zeros(n, np.uint8)
        _lib.check(self.L.btlbf_download(self.f, C.c_void_p(out.ctypes.data), 0, n))
        return out

    def store_shard(self, path):
        _lib.check(self.L.btlbf_store_shard(self.f, str(path).encode()))


class ShardedBloomFilter:
    """counting=True: a sharded CountingBloomFilter<uint8_t> of `global_bits` counters -- insert_reads is
    incrementAll (exact, saturating; shard-local at the owners) and contains_reads is
    "minimum >= threshold"; gather mode or the routed path (the conservative update needs the h counters
    of a k-mer, which live on different shards, and is not offered)."""

    def __init__(self, global_bits, hash_num, kmer_size, device=0, group=None, ops=None, batch_reads=2_000_000,
                 slack=1.25, route=True, batch_bytes_cap=0, pipeline=None, counting=False, threshold=0, mode=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if global_bits % (64 * self.world):
            raise ValueError("filter bits must split into multiples of 64 per shard")
        self.bits = global_bits
        self.h = hash_num
        self.k = kmer_size
        self.counting = bool(counting)
        self.ops = ops if ops is not None else HipShardOps(global_bits, hash_num, kmer_size, self.rank, self.world,
                                                           device, counting=counting, threshold=threshold)
        self.batch_reads = batch_reads
        self.slack = slack
        self.route_enabled = route         # use the partitioned routing path when the geometry allows
        # "gather": every rank receives every rank's reads (1 byte per base over xGMI instead of 4 bytes
        # per probe) and applies the probes inside its own window -- W times the hashing, no probe
        # exchange; "exchange": the routed / direct paths below.  auto = gather for 2..4 ranks: xGMI is one
        # link per GPU pair, so the probe exchange is link-bound at small W.
        # "routed" / "direct" pin the exchange flavour: "routed" raises here when the geometry has no routed
        # path instead of quietly falling back to the (much slower) direct position exchange
        mode = mode or os.environ.get("BTLBF_SHARD_MODE") or "auto"
        if mode not in ("auto", "gather", "exchange", "routed", "direct"):
            raise ValueError("mode must be auto, gather, exchange, routed or direct")
        can_gather = hasattr(self.ops, "insert_seqs")
        if mode == "gather" and not can_gather:
            raise ValueError("gather mode: these ops have no whole-buffer insert")
        if mode == "direct":
            self.route_enabled = False
        if mode == "routed" and not self._routed():
            raise ValueError("routed mode: no routed path for a 2^%.1f-position filter on %d shards with these ops "
                             "(needs powers of two, at least 2^29 bits, h <= 8)"
                             % (__import__("math").log2(global_bits), self.world))
        self.mode = "gather" if can_gather and (mode == "gather" or (mode == "auto" and 2 <= self.world <= 4)) \
            else "exchange"
        if self.counting and self.mode != "gather" and not (route and self.ops.route_supported(global_bits, self.world)):
            raise ValueError("sharded counting filters need gather mode or the routed path (power-of-two "
                             "geometry, h <= 8)")
        self.batch_bytes_cap = batch_bytes_cap
        # routed path: keep the exchange of batch i in flight while batch i+1 is routed (two buffer
        # sets).  None = whenever the exchange is asynchronous (RCCL); True forces the same schedule
        # over a synchronous exchange (tests).
        self.pipeline = pipeline
        # BTLBF_FORCE_EXCHANGE=1: run the routed path's exchange even with a single rank (a one-rank RCCL
        # group sends every block to itself) -- lets a one-GPU box exercise the real collective calls
        self.force_exchange = bool(os.environ.get("BTLBF_FORCE_EXCHANGE")) and dist.is_initialized()
        # "2": as in production the rank's own block bypasses the collective (which then carries only
        # zero-length messages in a one-rank group); "1": the own block goes through RCCL too
        self.self_through_rccl = os.environ.get("BTLBF_FORCE_EXCHANGE") == "1"
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
        # gloo moves host memory: stage device tensors through the CPU (test mode only)
        self.stage_cpu = backend == "gloo"

    # ---- the exchange -----------------------------------------------------------------------
    # No single point-to-point message is larger than this.  Measured on this stack (ROCm 7.2 RCCL,
    # examples/sharded_rccl.cpp): a grouped ncclSend/ncclRecv of more than about 1 GiB delivered only
    # half of its bytes (973 MB arrives whole, 2.0 GB and 4.2 GB arrive halved), so large blocks are
    # exchanged as a sequence of grouped all-to-alls over slices.
    MSG_BYTES = 256 << 20

    def _all_to_all(self, send, send_counts, recv_counts):
        """variable-size all-to-all of a flat tensor; counts are python ints per peer"""
        if self.world == 1:
            return send
        dev = send.device
        if self.stage_cpu and send.is_cuda:
            send = send.cpu()
        recv = torch.empty(sum(recv_counts), dtype=send.dtype, device=send.device)
        step = max(1, self.MSG_BYTES // send.element_size())
        rounds = self._max_over_ranks(-(-max(max(send_counts), max(recv_counts), 1) // step))
        if rounds <= 1:
            dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=send_counts,
                                   group=self.group)
        else:
            so = [0] + list(torch.tensor(send_counts).cumsum(0).tolist())
            ro = [0] + list(torch.tensor(recv_counts).cumsum(0).tolist())
            for r in range(rounds):
                ins = [send[so[p] + min(r * step, send_counts[p]): so[p] + min((r + 1) * step, send_counts[p])]
                       for p in range(self.world)]
                outs = [recv[ro[p] + min(r * step, recv_counts[p]): ro[p] + min((r + 1) * step, recv_counts[p])]
                        for p in range(self.world)]
                if self.stage_cpu:  # gloo (tests): the single-tensor form on the packed slices
                    tmp = torch.empty(sum(o.numel() for o in outs), dtype=send.dtype)
                    dist.all_to_all_single(tmp, torch.cat(ins), output_split_sizes=[o.numel() for o in outs],
                                           input_split_sizes=[i.numel() for i in ins], group=self.group)
                    at = 0
                    for o in outs:
                        o.copy_(tmp[at: at + o.numel()])
                        at += o.numel()
                else:
                    dist.all_to_all(outs, ins, group=self.group)
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
        while reads.numel() == 0:  # a rank that has run out of reads still takes part in the exchange
            dev = self.ops.device
            z = torch.zeros(self.world, dtype=torch.int64, device=dev)
            e = torch.empty(0, dtype=torch.int64, device=dev)
            recv_cnt = self._exchange_counts(z).cpu().tolist()
            return e, (e if want_tags else None), [0] * self.world, recv_cnt, \
                (torch.empty(0, dtype=torch.int64, device=dev) if want_tags else None)
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
        """(offset, chunk) pairs; every rank runs the SAME number of iterations -- each one issues collectives --
        with empty chunks once its own reads are exhausted"""
        step = self.batch_reads * read_len
        n = self._max_over_ranks(-(-reads.numel() // step))
        for i in range(n):
            yield i * step, reads[i * step: (i + 1) * step]

    # ---- public -----------------------------------------------------------------------------
    def clear(self):
        self.ops.clear()

    # ---- routed path: 4-byte partitioned entries, fixed-size all-to-all, owners apply in LDS ----
    FAIL_CAP = 4 << 20   # failed positions one owner may report per batch (library limit for resolve)
    SPILL_CAP = 1 << 20

    def _routed(self):
        return (self.route_enabled and hasattr(self.ops, "route_supported")
                and self.ops.route_supported(self.bits, self.world))

    def _max_over_ranks(self, v):
        if self.world == 1:
            return int(v)
        t = torch.tensor([int(v)], dtype=torch.int64)
        if not self.stage_cpu:
            t = t.to(self.ops.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def _fixed_all_to_all(self, send):
        if self.world == 1:
            return send
        if send.dtype == torch.uint8 and send.numel() % (8 * self.world) == 0:
            # exchange 8-byte elements: keeps per-peer element counts far below 2^31
            return self._fixed_all_to_all(send.view(torch.int64)).view(torch.uint8)
        dev = send.device
        if self.stage_cpu and send.is_cuda:
            send = send.cpu()
        recv = torch.empty_like(send)
        for w in self._sliced_all_to_all(recv, send, async_op=False):
            pass
        return recv.to(dev) if recv.device != dev else recv

    def _sliced_all_to_all(self, recv, send, async_op):
        """all-to-all of equal per-peer blocks of two flat tensors; see _rows_all_to_all"""
        per = send.numel() // self.world
        return self._rows_all_to_all(recv.view(self.world, per), send.view(self.world, per), async_op)

    def _rows_all_to_all(self, r2, s2, async_op, first=0, receiving=True):
        """row i of s2 goes to rank first + i (s2 has one row per TARGET rank: all ranks, or the owners of
        one position window), row p of r2 comes from rank p when this rank is `receiving` (2-D tensors whose
        rows are contiguous; the rows of s2 may be slices of larger blocks).  At most MSG_BYTES per message;
        returns the work handles.  With RCCL the row a rank keeps for itself does not go through the
        collective at all: it is one device-to-device copy on the compute stream (BTLBF_FORCE_EXCHANGE=1
        keeps it in, for tests)."""
        W = self.world
        per = s2.shape[1]
        n_targets = s2.shape[0]
        targets = range(first, first + n_targets)
        everyone = n_targets == W and receiving
        step = max(1, self.MSG_BYTES // s2.element_size())
        local_self = not self.stage_cpu and not self.self_through_rccl
        if everyone and per <= step and not local_self and not self.stage_cpu and s2.is_contiguous() \
                and r2.is_contiguous():
            return [dist.all_to_all_single(r2.view(-1), s2.view(-1), group=self.group, async_op=async_op)]
        works = []
        for c0 in range(0, per, step):
            c1 = min(c0 + step, per)
            n = c1 - c0
            if self.stage_cpu:  # gloo (tests): contiguous copies of the slice through the single-tensor form
                ins = [n if p in targets else 0 for p in range(W)]
                outs = [n if receiving else 0 for _ in range(W)]
                tmp = torch.empty(sum(outs), dtype=s2.dtype)
                dist.all_to_all_single(tmp, s2[:, c0:c1].contiguous().cpu().view(-1), output_split_sizes=outs,
                                       input_split_sizes=ins, group=self.group)
                if receiving:
                    r2[:, c0:c1] = tmp.view(W, n).to(r2.device)
            else:
                keep = (lambda p: p == self.rank) if local_self else (lambda p: False)
                works.append(dist.all_to_all(
                    [r2[p, c0:c1] if receiving and not keep(p) else r2[p, c0:c0] for p in range(W)],
                    [s2[p - first, c0:c1] if p in targets and not keep(p) else s2[0, c0:c0] for p in range(W)],
                    group=self.group, async_op=async_op))
        if local_self and receiving and self.rank in targets:
            r2[self.rank].copy_(s2[self.rank - first])
        return works

    def _all_gather_var(self, t, n):
        """all-gather the first n elements of the 1-D int64 tensor t from every rank -> 1-D tensor"""
        if self.world == 1:
            return t[:n]
        dev = t.device
        cnt = torch.tensor([n], dtype=torch.int64, device="cpu" if self.stage_cpu else dev)
        cnts = [torch.zeros_like(cnt) for _ in range(self.world)]
        dist.all_gather(cnts, cnt, group=self.group)
        cnts = [int(c.item()) for c in cnts]
        m = max(cnts)
        if m == 0:
            return t[:0]
        pad = torch.zeros(m, dtype=t.dtype, device="cpu" if self.stage_cpu else dev)
        pad[:n] = t[:n].to(pad.device)
        outs = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(outs, pad, group=self.group)
        return torch.cat([o[:c] for o, c in zip(outs, cnts)]).to(dev)

    def _route_batch_bytes(self, reads, read_len, n_slots=1, recv_sets=0.0, exact=None):
        """bytes of read buffer per batch (same on every rank): n_slots send block sets, the receive
        buffers (recv_sets block sets: two groups of a block set) and two split levels must fit in free HBM"""
        longest = self._max_over_ranks(reads.numel())
        if longest == 0:
            return 0, 0
        if self.ops.device.type == "cuda":
            # what the driver reports as free, plus what torch's caching allocator holds without using it
            # (the block sets of the previous pass): otherwise every pass after the first plans tiny batches
            free = torch.cuda.mem_get_info(self.ops.device)[0]
            free += torch.cuda.memory_reserved(self.ops.device) - torch.cuda.memory_allocated(self.ops.device)
        else:
            free = 1 << 40
        free = -self._max_over_ranks(-free)  # the smallest over ranks: every rank must plan the same batch
        unit = 64 * read_len
        if exact is not None and hasattr(self.ops, "owner_scratch_bytes") and not self.batch_bytes_cap:
            # the buffers of a pass added up exactly (send block sets, receive groups, the owner's scratch inside the
            # filter, spill and fail lists), for the fewest batches that leave a tenth of the free HBM untouched: the
            # rule of thumb below keeps a fifth and more, which at BASELINE config 4 (128 GiB shards) is the difference
            # between five batches per pass and four -- one sweep of the shard per pass
            n_slots_e, spw, W, r_slots, G, exchanging = exact
            lists = n_slots_e * self.SPILL_CAP * 8 + self.FAIL_CAP * 8 + (64 << 20)
            # (the owner's scratch of an earlier pass is still allocated inside the filter -- it only grows -- and will
            # be used again: it is not free, but it is not an additional need either)
            held = getattr(self, "_owner_scratch", 0)
            for n_b in range(1, 4097):
                batch = -(-(-(-longest // n_b)) // unit) * unit
                ent_b, cnt_b = self.ops.route_plan(batch, read_len)
                owner = self.ops.owner_scratch_bytes(batch, read_len)
                total = n_slots_e * spw * (ent_b + cnt_b) + lists + max(owner - held, 0)
                if exchanging:
                    total += r_slots * W * (ent_b // G + cnt_b // G)
                if total <= 0.90 * free or batch <= unit:
                    self._owner_scratch = max(held, owner)
                    return batch, -(-longest // batch)
        # per read byte: h*(L-k+1)/L probes * 4 B per entry, ~1.1x capacity; block sets: send, receive
        # (when there are peers) + the owner's split levels, which hold 1/8 of a batch
        sets = n_slots + recv_sets + 0.4
        per_byte = self.h * max(read_len - self.k + 1, 1) / read_len * 4 * 1.1 * sets
        unit = 64 * read_len
        batch = int(0.78 * free / per_byte) // unit * unit
        batch = max(batch, unit)
        if self.batch_bytes_cap:
            batch = min(batch, self.batch_bytes_cap // unit * unit or unit)
        n_batches = -(-longest // batch)
        batch = -(-(-(-longest // n_batches)) // unit) * unit  # equal batches: no more memory than needed
        return batch, -(-longest // batch)

    # NOTE for the C ABI: ops.route (btlbf_route_seqs) must stay free of per-filter scratch -- it runs on a second
    # stream while btlbf_apply_routed* uses the filter's scratch on the first (capi.cpp: route passes no late list,
    # so pass A takes its plain schedule there; tests/test_sharded.py runs both streams together)
    def _routed_pass(self, reads, read_len, query, hit_bits=None, counts=None):
        """One insert / query pass over this rank's reads on the routed path.  The unit of work is a JOB =
        (batch of reads, position window): route (pass A with the global geometry, the probes inside the
        window) into a send block set; the owners' bins then travel and are applied a GROUP at a time (1/8
        of a shard's level-0 bins: the unit the owner splits and applies anyway), so the receive side needs
        two groups of buffer instead of two block sets and the exchange of group g+1 overlaps the apply of
        group g.  Filters of up to 2^42 positions have one window; a 2^43-bit filter on 8 GPUs has two, each
        owned by 4 consecutive shards, and every batch is routed once per window (only the window's owners
        receive blocks of that job).  With RCCL the next job is routed on a second stream while the groups
        of this one are exchanged and applied.  Entries that could not be staged at their origin (skewed
        input) come back as explicit positions per job; they are collected without limit and exchanged once
        at the end of the pass, like the fail lists of a query."""
        ops, W, dev = self.ops, self.world, self.ops.device
        exchanging = W > 1 or self.force_exchange
        pipelined = exchanging and (not self.stage_cpu if self.pipeline is None else bool(self.pipeline))
        n_slots = 2 if pipelined else 1
        unit = 64 * read_len
        n_win, spw = ops.route_windows() if hasattr(ops, "route_windows") else (1, W)
        my_win = self.rank // spw
        bins, _, _, gb = ops.route_geometry(unit, read_len)  # bins per shard and per group: the same for any length
        G = bins // gb if exchanging else 1
        r_slots = 2 if G > 1 else 1
        batch, n_batches = self._route_batch_bytes(reads, read_len, n_slots / n_win,
                                                   r_slots / G if exchanging else 0.0,
                                                   exact=(n_slots, spw, W, r_slots, G, exchanging))
        if n_batches == 0:
            return True
        n_jobs = n_batches * n_win
        ent_b, cnt_b = ops.route_plan(batch, read_len)
        ge, gc = ent_b // G, cnt_b // G  # bytes of one group inside one block
        assert ge * G == ent_b and gc * G == cnt_b and ge % 8 == 0
        send_ent = [torch.empty(spw * ent_b, dtype=torch.uint8, device=dev) for _ in range(n_slots)]
        send_cnt = [torch.empty(spw * cnt_b, dtype=torch.uint8, device=dev) for _ in range(n_slots)]
        if exchanging:
            recv_ent = [torch.empty(W * ge, dtype=torch.uint8, device=dev) for _ in range(r_slots)]
            recv_cnt = [torch.empty(W * gc, dtype=torch.uint8, device=dev) for _ in range(r_slots)]
        spill = [torch.empty(self.SPILL_CAP, dtype=torch.int64, device=dev) for _ in range(n_slots)]
        spill_count = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(n_slots)]
        # the host learns a job's spill count from a pinned copy made on the ROUTING stream right behind the routing
        # (a .item() on the caller's stream would wait for the exchange and apply work queued there)
        spill_host = [torch.zeros(1, dtype=torch.int64).pin_memory() if dev.type == "cuda" else None
                      for _ in range(n_slots)]
        spilled = []  # explicit positions of the jobs done so far (device tensors)
        fail = torch.empty(self.FAIL_CAP, dtype=torch.int64, device=dev) if query else None
        fail_count = torch.zeros(1, dtype=torch.int64, device=dev) if query else None
        cnt2 = torch.zeros(2, dtype=torch.int64, device=dev) if query else None
        # pipelined: routing runs on its own stream, the exchange + apply of the previous job on the caller's
        cur = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        side = torch.cuda.Stream(dev) if (pipelined and cur is not None) else None
        routed_ev, free_ev = [None] * n_slots, [None] * n_slots
        if side is not None:
            side.wait_stream(cur)

        def job_args(j):
            bi, w = divmod(j, n_win)
            off = bi * batch
            chunk = reads[off: off + batch]  # empty once this rank has run out of reads: it still takes part
            view = None
            if query:
                w0, nw = off // 64, (chunk.numel() + 63) // 64
                view = hit_bits[w0: w0 + nw]
            # every window's pass re-initialises the hit bits of the batch (hit = valid; failures are resolved
            # at the end of the pass) but the clean windows are counted once
            return chunk, view, w, (cnt2 if w == 0 else None)

        def do_route(j, slot, again=False):
            chunk, view, w, c2 = job_args(j)
            spill_count[slot].zero_()
            # (a job routed a second time has already counted its clean windows)
            ops.route(chunk, read_len, batch, query, send_ent[slot], send_cnt[slot], view, None,
                      None if again else c2, spill[slot], spill_count[slot], window=w)

        def route(j):
            slot = j % n_slots
            if side is None:
                do_route(j, slot)
                return
            if free_ev[slot] is not None:
                side.wait_event(free_ev[slot])  # the exchanges that read this block set have completed
            with torch.cuda.stream(side):
                do_route(j, slot)
                spill_host[slot].copy_(spill_count[slot], non_blocking=True)
                routed_ev[slot] = torch.cuda.Event()
                routed_ev[slot].record(side)

        def collect_spill(j, slot):
            """the job's explicit positions; a list that overflowed (heavily skewed input: poly-A reads, one
            read a million times) is routed again into one that is large enough -- nothing of the job has
            left this rank yet"""
            if routed_ev[slot] is not None:
                routed_ev[slot].synchronize()  # the job's routing (and the copy of its count): nothing else
                n = int(spill_host[slot][0])
            else:
                n = int(spill_count[slot].item())
            while n > spill[slot].numel():
                spill[slot] = torch.empty(n + n // 4 + 1024, dtype=torch.int64, device=dev)
                do_route(j, slot, again=True)  # on the caller's stream; the blocks come out equivalent
                n = int(spill_count[slot].item())
            if n:
                spilled.append(spill[slot][:n].clone())

        def group_rows(t, per_block, g, gbytes):
            """rows [spw, group bytes] of group g inside the blocks of a send set, as 8-byte elements if possible"""
            if gbytes % 8 == 0 and per_block % 8 == 0:
                return t.view(torch.int64).view(spw, per_block // 8)[:, g * gbytes // 8: (g + 1) * gbytes // 8]
            return t.view(spw, per_block)[:, g * gbytes: (g + 1) * gbytes]

        def flat_rows(t, gbytes):
            return t.view(torch.int64).view(W, gbytes // 8) if gbytes % 8 == 0 else t.view(W, gbytes)

        def exchange(slot, g, w):
            rs = g % r_slots
            works = []
            for snd, rcv, per_block, gbytes in ((send_ent[slot], recv_ent[rs], ent_b, ge),
                                                (send_cnt[slot], recv_cnt[rs], cnt_b, gc)):
                s2 = group_rows(snd, per_block, g, gbytes)
                r2 = flat_rows(rcv, gbytes) if s2.dtype == torch.int64 else rcv.view(W, gbytes)
                works += self._rows_all_to_all(r2, s2, async_op=not self.stage_cpu, first=w * spw,
                                               receiving=(w == my_win)) or []
            return [x for x in works if x is not None]

        def finish(j):
            slot = j % n_slots
            w = j % n_win
            collect_spill(j, slot)
            if side is not None:
                cur.wait_event(routed_ev[slot])
            if not exchanging:
                ops.apply_routed(send_ent[slot], send_cnt[slot], W, batch, read_len, query, fail, fail_count)
                return
            inflight = {g: exchange(slot, g, w) for g in range(min(r_slots, G))}
            for g in range(G):
                for x in inflight.pop(g):
                    x.wait()  # RCCL: the compute stream waits, the host does not
                rs = g % r_slots
                if w == my_win:
                    ops.apply_routed_bins(recv_ent[rs], recv_cnt[rs], W, g * gb, gb, batch, read_len, query, fail,
                                          fail_count)
                if g + r_slots < G:
                    inflight[g + r_slots] = exchange(slot, g + r_slots, w)  # into the buffer just applied
            if side is not None:
                free_ev[slot] = torch.cuda.Event()
                free_ev[slot].record(cur)

        pending = None
        for j in range(n_jobs):
            route(j)
            if pipelined:
                if pending is not None:
                    finish(pending)
                pending = j
            else:
                finish(j)
        if pending is not None:
            finish(pending)
        if side is not None:
            cur.wait_stream(side)
        # entries that could not be staged at their origin travel as explicit positions (rare)
        mine = torch.cat(spilled) if spilled else torch.empty(0, dtype=torch.int64, device=dev)
        gathered = self._all_gather_var(mine, mine.numel())
        if gathered.numel():
            ops.apply_spill(gathered, query, fail, fail_count)
        if query:
            n_fail = int(fail_count.item())
            if self._max_over_ranks(1 if n_fail > self.FAIL_CAP else 0):
                return False  # miss-heavy: the caller redoes the query on the direct path
            fails = self._all_gather_var(fail, n_fail)
            if fails.numel() > self.FAIL_CAP:
                return False
            if fails.numel() and reads.numel():
                ops.resolve(reads, read_len, fails, hit_bits)
            if counts is not None:
                counts[0] = int(cnt2[0].item())
                counts[1] = ops.popcount_bits(hit_bits[: (reads.numel() + 63) // 64]) if reads.numel() else 0
        return True

    # ---- gather mode ------------------------------------------------------------------------------
    GATHER_BYTES = 48 << 30  # peers' reads held per round, all peers together

    def _gather_start(self, out, piece, peers):
        """out[i] := the piece of rank peers[i].  RCCL: grouped sends/receives of at most MSG_BYTES,
        asynchronous (returns the work handles).  A rank is its own peer only in the one-rank test modes
        (BTLBF_FORCE_EXCHANGE): its piece then goes through RCCL ("1") or is a device copy ("2")."""
        W, n = self.world, piece.numel()
        if not peers:
            return []
        o2 = out.view(len(peers), n)
        idx = {p: i for i, p in enumerate(peers)}
        if self.stage_cpu:  # gloo (tests)
            parts = [torch.empty(n, dtype=piece.dtype) for _ in range(W)]
            dist.all_gather(parts, piece.cpu(), group=self.group)
            for p in peers:
                o2[idx[p]].copy_(parts[p])
            return []
        works = []
        none = piece[0:0]
        local = self.rank in idx and not self.self_through_rccl
        wire = lambda p: p in idx and not (p == self.rank and local)  # noqa: E731  (p's piece arrives over RCCL)
        step = max(1, self.MSG_BYTES // piece.element_size())
        for c0 in range(0, n, step):
            c1 = min(c0 + step, n)
            works.append(dist.all_to_all([o2[idx[p], c0:c1] if wire(p) else none for p in range(W)],
                                         [piece[c0:c1] if (p != self.rank or wire(p)) else none for p in range(W)],
                                         group=self.group, async_op=True))
        if local:
            o2[idx[self.rank]].copy_(piece)
        return works

    def _gather_pass(self, reads, read_len, query, hit_bits=None, counts=None):
        """One insert / query pass in gather mode.  Per round every rank contributes `chunk` bytes of its
        reads (the last one padded with 'N': no k-mers) to all peers and, while they travel, hashes its
        OWN chunk; then the peers' chunks, while the next round's are in flight.  Query: every shard
        answers "all of my probes of this window are set" for every chunk it sees, the partial bitmaps of
        the peers' chunks go back to their owners (1 bit per window and peer) and are ANDed there."""
        ops, W, dev, rank = self.ops, self.world, self.ops.device, self.rank
        longest = self._max_over_ranks(reads.numel())
        if query and counts is not None:
            counts[0] = counts[1] = 0
        if longest == 0:
            return
        # one-rank test modes: the rank is its own (only) peer, so that the collectives carry data
        peers = list(range(W)) if self.force_exchange else [p for p in range(W) if p != rank]
        own_first = not self.force_exchange
        unit = 64 * read_len
        cap = self.batch_bytes_cap or max(self.GATHER_BYTES // max(len(peers), 1), unit)
        rounds = -(-longest // cap)
        chunk = -(-(-(-longest // rounds)) // unit) * unit
        rounds = -(-longest // chunk)
        n_slots = 2 if rounds > 1 else 1
        words = chunk // 64
        gathered = [torch.empty(len(peers) * chunk, dtype=torch.uint8, device=dev) for _ in range(n_slots)] \
            if peers else []
        pad = None  # send buffer for a chunk shorter than `chunk` (a rank's last one)
        if query:
            own_part = torch.empty(words, dtype=torch.int64, device=dev)
            own_valid = torch.empty(words, dtype=torch.int64, device=dev)
            if peers:
                part = torch.empty(len(peers) * words, dtype=torch.int64, device=dev)
                # the peers count their own clean windows; only the one-rank test modes need this rank's here
                valid = None if own_first else torch.empty(len(peers) * words, dtype=torch.int64, device=dev)
                sendb = torch.zeros(W * words, dtype=torch.int64, device=dev)
                back = torch.empty(W * words, dtype=torch.int64, device=dev)
        n_valid = 0

        def start(r):
            nonlocal pad
            if not peers:
                return []
            mine = reads[r * chunk: (r + 1) * chunk]
            if mine.numel() < chunk:
                if pad is None:
                    pad = torch.empty(chunk, dtype=torch.uint8, device=dev)
                pad[: mine.numel()].copy_(mine)
                pad[mine.numel():].fill_(78)  # 'N'
                mine = pad
            return self._gather_start(gathered[r % n_slots], mine, peers)

        works = start(0)
        for r in range(rounds):
            mine = reads[r * chunk: (r + 1) * chunk]
            have = mine.numel()
            hw = (have + 63) // 64
            if own_first and have:  # hashed while the peers' chunks of this round arrive
                if not query:
                    ops.insert_seqs(mine, read_len)
                else:
                    ops.contains_seqs(mine, read_len, own_part, own_valid)
            for w in works:
                w.wait()
            # the next round's send buffer may be `pad`: this round's sends have completed (waited above)
            nxt = start(r + 1) if r + 1 < rounds else []
            if peers:
                buf = gathered[r % n_slots]
                if not query:
                    ops.insert_seqs(buf, read_len)
                else:
                    ops.contains_seqs(buf, read_len, part, valid)
                    # row p of sendb = my answer for rank p's chunk; row p of back = rank p's answer for mine
                    sendb.view(W, words)[peers] = part.view(len(peers), words)
                    if self.stage_cpu:
                        back.copy_(self._fixed_all_to_all(sendb))
                    else:
                        for w in self._sliced_all_to_all(back, sendb, async_op=True):
                            w.wait()
            if query and hw:
                if own_first:
                    acc, val = own_part[:hw].clone(), own_valid
                else:
                    acc, val = None, valid.view(len(peers), words)[peers.index(rank)]
                for p in peers:
                    row = back.view(W, words)[p, :hw]
                    acc = row.clone() if acc is None else acc.bitwise_and_(row)
                hit_bits[r * words: r * words + hw].copy_(acc)
                if counts is not None:
                    n_valid += ops.popcount_bits(val[:hw])
            works = nxt
        if query and counts is not None:
            counts[0] = n_valid
            counts[1] = ops.popcount_bits(hit_bits[: (reads.numel() + 63) // 64]) if reads.numel() else 0

    def insert_reads(self, reads, read_len):
        """insertSeq over this rank's uniform-length reads (flat uint8 tensor)"""
        if self.mode == "gather":
            return self._gather_pass(reads, read_len, 0)
        if self._routed():
            self._routed_pass(reads, read_len, 0)
            return
        if self.counting:
            raise RuntimeError("counting filters have no direct exchange path")
        for _, chunk in self._batches(reads, read_len):
            send, _, cnt, recv_cnt, _ = self._bucketed(chunk, read_len, False)
            mine = self._all_to_all(send, cnt, recv_cnt)
            self.ops.insert_positions(mine)

    def contains_reads(self, reads, read_len, hit_bits, counts=None):
        """contains() of every window of this rank's reads -> hit_bits (int64 bitmap over the whole
        buffer, bit p = window at byte p); counts (optional int64[2]) = {clean windows, hits}"""
        if self.mode == "gather":
            self._gather_pass(reads, read_len, 1, hit_bits, counts)
            return hit_bits
        if self._routed():
            if self._routed_pass(reads, read_len, 1, hit_bits, counts):
                return hit_bits
            # too many failed probes for the fail lists: fall through to the exact answer routing
            if self.counting:
                raise RuntimeError("counting query: more failed probes than the fail lists hold")
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

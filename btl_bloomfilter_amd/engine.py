"""Host-side mirror of the reference's filter classes over the C ABI (include/btlbf.h).

Method names follow the reference (insert / contains / insertAndCheck / storeFilter / getPop ...,
/root/reference/BloomFilter.hpp:46-381, CountingBloomFilter.hpp:27-111) with batch arguments:
sequence buffers (bytes / numpy uint8 / torch uint8 tensors on the GPU) or (n, h) uint64 hash rows.
Buffers that are torch CUDA tensors are passed as device pointers; everything else is host memory
that the library stages itself."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (BLOOM, COUNTING8, DEVICE, HOST, INCREMENT_ALL, INCREMENT_MIN, ORDER_PARALLEL,
                   ORDER_SERIAL, Layout, check)


def _is_torch_cuda(x):
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)


def _stream_ptr(stream, buf=None):
    """hipStream_t for a call.  stream=None with a torch CUDA buffer means torch's CURRENT stream on that
    device (the stream the buffer's producer and the output zero-fills are ordered on), not the legacy
    NULL stream, which torch's non-blocking side streams do not synchronise with."""
    if stream is None:
        if buf is not None and _is_torch_cuda(buf):
            import torch

            return C.c_void_p(torch.cuda.current_stream(buf.device).cuda_stream)
        return None
    return C.c_void_p(int(getattr(stream, "cuda_stream", stream)))


class _Buf:
    """pointer + length + memory space of a caller buffer; keeps the backing object alive"""

    def __init__(self, x, dtype=np.uint8):
        if _is_torch_cuda(x):
            assert x.is_contiguous()
            self.keep = x
            self.ptr = C.c_void_p(x.data_ptr())
            self.nbytes = x.numel() * x.element_size()
            self.mem = DEVICE
        else:
            if isinstance(x, str):
                x = x.encode("latin-1")
            if isinstance(x, (bytes, bytearray, memoryview)):
                x = np.frombuffer(bytes(x), dtype=np.uint8)
            a = np.ascontiguousarray(x, dtype=dtype)
            self.keep = a
            self.ptr = C.c_void_p(a.ctypes.data) if a.size else C.c_void_p(0)
            self.nbytes = a.nbytes
            self.mem = HOST


def _layout(starts, read_len, mem):
    if starts is None and not read_len:
        return None, None
    lay = Layout()
    keep = None
    if starts is not None:
        b = _Buf(starts, np.uint64)
        if b.mem != mem:
            raise ValueError("starts must live in the same memory space as the sequence buffer")
        keep = b
        lay.starts = b.ptr
        lay.n_seqs = b.nbytes // 8 - 1
        lay.read_len = 0
    else:
        lay.starts = None
        lay.n_seqs = 0
        lay.read_len = int(read_len)
    return lay, keep


def bits_to_bool(bits, n):
    """per-window bitmap (uint64 words, bit p&63 of word p>>6) -> bool array of n windows"""
    b = np.ascontiguousarray(bits).view(np.uint8)
    return np.unpackbits(b, bitorder="little")[:n].astype(bool)


def _bitmap(n):
    return np.zeros((n + 63) // 64, np.uint64)


class _Filter:
    kind = BLOOM

    def __init__(self, handle):
        self._h = handle
        self._L = _lib.load()

    # -- lifetime --------------------------------------------------------------------------
    @classmethod
    def _create(cls, size, hash_num, kmer_size, threshold=0, device=0):
        L = _lib.load()
        h = C.c_void_p()
        check(L.btlbf_create(C.byref(h), cls.kind, int(size), int(hash_num), int(kmer_size), int(threshold),
                             int(device)))
        return h

    @classmethod
    def _load(cls, path, threshold=0, device=0):
        L = _lib.load()
        h = C.c_void_p()
        check(L.btlbf_load(C.byref(h), cls.kind, str(path).encode(), int(threshold), int(device)))
        return h

    def close(self):
        if getattr(self, "_h", None):
            self._L.btlbf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- attributes (reference getters) ------------------------------------------------------
    def getHashNum(self):
        return self._L.btlbf_hash_num(self._h)

    def getKmerSize(self):
        return self._L.btlbf_kmer_size(self._h)

    def sizeInBytes(self):
        return self._L.btlbf_size_bytes(self._h)

    def localBytes(self):
        return self._L.btlbf_local_bytes(self._h)

    def device_ptr(self):
        return self._L.btlbf_device_ptr(self._h)

    def header(self):
        n = C.c_size_t()
        buf = C.create_string_buffer(1024)
        check(self._L.btlbf_header(self._h, buf, 1024, C.byref(n)))
        return buf.raw[: n.value]

    def storeFilter(self, path):
        check(self._L.btlbf_store(self._h, str(path).encode()))

    def storeShard(self, path):
        check(self._L.btlbf_store_shard(self._h, str(path).encode()))

    def clear(self, stream=None):
        check(self._L.btlbf_clear(self._h, _stream_ptr(stream)))

    def download(self):
        out = np.zeros(self.localBytes(), np.uint8)
        check(self._L.btlbf_download(self._h, C.c_void_p(out.ctypes.data), 0, out.nbytes))
        return out

    def upload(self, body):
        a = np.ascontiguousarray(body, np.uint8)
        check(self._L.btlbf_upload(self._h, C.c_void_p(a.ctypes.data), 0, a.nbytes))

    def compare(self, other):
        """(positions that differ, positions where self > other, where self < other) against another filter
        of the same geometry, computed in HBM (btlbf_compare); positions are bits or uint8_t counters"""
        out = (C.c_uint64 * 3)()
        check(self._L.btlbf_compare(self._h, other._h, out))
        return tuple(out)

    def digest(self):
        """(sum, xor) digest of the local array computed in HBM (btlbf_digest); shard digests combine by + and ^"""
        out = (C.c_uint64 * 2)()
        check(self._L.btlbf_digest(self._h, out))
        return int(out[0]), int(out[1])

    def setInsertMode(self, mode, scratch_bytes=0):
        """'auto' | 'direct' | 'partitioned' (see btlbf_set_insert_mode)"""
        m = {"auto": 0, "direct": 1, "partitioned": 2}[mode] if isinstance(mode, str) else int(mode)
        check(self._L.btlbf_set_insert_mode(self._h, m, int(scratch_bytes)))

    def releaseScratch(self):
        check(self._L.btlbf_release_scratch(self._h))

    def setProfiling(self, on=True):
        check(self._L.btlbf_set_profiling(self._h, int(bool(on))))

    def getProfile(self, reset=True):
        """{kernel slot name: (milliseconds, launches)} measured with HIP events on the launch stream"""
        ms = (C.c_double * _lib.PROF_SLOTS)()
        calls = (C.c_uint * _lib.PROF_SLOTS)()
        check(self._L.btlbf_get_profile(self._h, ms, calls, int(bool(reset))))
        return {n: (ms[i], calls[i]) for i, n in enumerate(_lib.PROF_NAMES) if calls[i]}

    def setQueryMode(self, mode):
        """'auto' | 'direct' | 'partitioned' (see btlbf_set_query_mode)"""
        m = {"auto": 0, "direct": 1, "partitioned": 2}[mode] if isinstance(mode, str) else int(mode)
        check(self._L.btlbf_set_query_mode(self._h, m))

    # -- FASTA / FASTQ files (btlbf_insert_fastx / btlbf_contains_fastx) -----------------------------
    def insertFile(self, path, per_line=False, batch_bytes=0):
        """Insert every k-mer of a FASTA / FASTQ / one-sequence-per-line file (optionally gzipped).
        per_line=True treats every sequence line as its own sequence (the reference's loadBf);
        the default concatenates the lines of a FASTA record (contigsToBloom).  Returns the stats dict."""
        st = _lib.FastxStats()
        check(self._L.btlbf_insert_fastx(self._h, str(path).encode(), 1 if per_line else 0, int(batch_bytes),
                                         C.byref(st)))
        return st.as_dict()

    def containsFile(self, path, per_line=False, batch_bytes=0):
        """Query every k-mer of a file; stats['n_windows'] clean windows, stats['n_hits'] of them found."""
        st = _lib.FastxStats()
        check(self._L.btlbf_contains_fastx(self._h, str(path).encode(), 1 if per_line else 0, int(batch_bytes),
                                           C.byref(st)))
        return st.as_dict()

    def setSpacedSeeds(self, seeds, h2=1):
        arr = (C.c_char_p * len(seeds))(*[s.encode() if isinstance(s, str) else s for s in seeds])
        check(self._L.btlbf_set_spaced_seeds(self._h, arr, len(seeds), h2))

    # -- sequence buffers ----------------------------------------------------------------------
    def _query(self, fn, seq, starts, read_len, want_valid, want_counts, stream):
        b = _Buf(seq)
        lay, keep = _layout(starts, read_len, b.mem)
        n = b.nbytes
        if b.mem == DEVICE:
            import torch

            hit = torch.zeros((n + 63) // 64, dtype=torch.int64, device=b.keep.device)
            valid = torch.zeros_like(hit) if want_valid else None
            cnt = torch.zeros(2, dtype=torch.int64, device=b.keep.device) if want_counts else None
            ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        else:
            hit = _bitmap(n)
            valid = _bitmap(n) if want_valid else None
            cnt = np.zeros(2, np.uint64) if want_counts else None
            ptr = lambda a: C.c_void_p(a.ctypes.data) if a is not None else None  # noqa: E731
        check(fn(self._h, b.ptr, n, C.byref(lay) if lay else None, ptr(hit), ptr(valid), ptr(cnt), b.mem,
                 _stream_ptr(stream, b.keep)))
        return hit, valid, cnt

    def containsSeqs(self, seq, starts=None, read_len=0, want_valid=True, want_counts=False, stream=None):
        """contains() of every window of a sequence buffer -> (hit_bits, valid_bits, counts)"""
        return self._query(self._L.btlbf_contains_seqs, seq, starts, read_len, want_valid, want_counts, stream)

    def _insert_seqs(self, seq, starts, read_len, op, order, stream):
        b = _Buf(seq)
        lay, keep = _layout(starts, read_len, b.mem)
        check(self._L.btlbf_insert_seqs(self._h, b.ptr, b.nbytes, C.byref(lay) if lay else None, op, order,
                                        b.mem, _stream_ptr(stream, b.keep)))

    # -- hash rows -----------------------------------------------------------------------------
    def _rows(self, hashes):
        h = self.getHashNum()
        if _is_torch_cuda(hashes):
            b = _Buf(hashes)
            return b, b.nbytes // (8 * h)
        a = np.ascontiguousarray(hashes, dtype=np.uint64).reshape(-1, h)
        return _Buf(a, np.uint64), a.shape[0]

    def _rows_out(self, fn, hashes, *extra, stream=None):
        b, n = self._rows(hashes)
        if b.mem == DEVICE:
            import torch

            out = torch.zeros(n, dtype=torch.uint8, device=b.keep.device)
            optr = C.c_void_p(out.data_ptr())
        else:
            out = np.zeros(max(n, 1), np.uint8)
            optr = C.c_void_p(out.ctypes.data)
        check(fn(self._h, b.ptr, n, optr, *extra, b.mem, _stream_ptr(stream, b.keep)))
        return out[:n]


class BloomFilter(_Filter):
    """bit-array filter (reference: /root/reference/BloomFilter.hpp)"""

    kind = BLOOM

    def __init__(self, filterSize=None, hashNum=None, kmerSize=None, path=None, device=0, _handle=None):
        if _handle is not None:
            super().__init__(_handle)
        elif path is not None:
            super().__init__(self._load(path, 0, device))
        else:
            super().__init__(self._create(filterSize, hashNum, kmerSize, 0, device))

    @classmethod
    def shard(cls, global_bits, shard_index, shard_count, hashNum, kmerSize, device=0):
        L = _lib.load()
        h = C.c_void_p()
        check(L.btlbf_create_shard(C.byref(h), BLOOM, global_bits, shard_index, shard_count, hashNum, kmerSize,
                                   0, device))
        return cls(_handle=h)

    def getFilterSize(self):
        return self._L.btlbf_size(self._h)

    def getnEntry(self):
        return self._L.btlbf_get_n_entry(self._h)

    def gettEntry(self):
        return self._L.btlbf_get_t_entry(self._h)

    def setnEntry(self, v):
        self._L.btlbf_set_n_entry(self._h, v)

    def settEntry(self, v):
        self._L.btlbf_set_t_entry(self._h, v)

    def getPop(self):
        out = C.c_uint64()
        check(self._L.btlbf_popcount(self._h, C.byref(out)))
        return out.value

    def getFPR(self):
        return (self.getPop() / self.getFilterSize()) ** self.getHashNum()  # BloomFilter.hpp:346-350

    # batch forms of insert / contains / insertAndCheck over hash rows
    def insert(self, hashes, stream=None):
        b, n = self._rows(hashes)
        check(self._L.btlbf_insert_hashes(self._h, b.ptr, n, 0, ORDER_PARALLEL, b.mem, _stream_ptr(stream, b.keep)))

    def contains(self, hashes, stream=None):
        return self._rows_out(self._L.btlbf_contains_hashes, hashes, stream=stream)

    def insertAndCheck(self, hashes, serial=True, stream=None):
        return self._rows_out(self._L.btlbf_insert_and_check_hashes, hashes,
                              ORDER_SERIAL if serial else ORDER_PARALLEL, stream=stream)

    # insertSeq (BloomFilterUtil.h:10) over whole buffers
    def insertSeqs(self, seq, starts=None, read_len=0, stream=None):
        self._insert_seqs(seq, starts, read_len, 0, ORDER_PARALLEL, stream)

    def insertAndCheckSeqs(self, seq, starts=None, read_len=0, want_valid=True, want_counts=False, stream=None):
        return self._query(self._L.btlbf_insert_and_check_seqs, seq, starts, read_len, want_valid, want_counts,
                           stream)

    def microbench(self, kind, n_access):
        done = C.c_uint64()
        sec = C.c_double()
        check(self._L.btlbf_microbench(self._h, kind, n_access, C.byref(done), C.byref(sec)))
        return done.value, sec.value


class KmerBloomFilter(BloomFilter):
    """The surface the reference's SWIG module exposes as `BloomFilter` (swig/BloomFilter.i:17-59 =
    KmerBloomFilter.hpp:17-75): insert / contains take either a k-mer string or a row of precomputed
    hashes; together with insertSeq() below this replaces both the SWIG/Perl binding and the stale
    Boost.Python module under pythonInterface/.  K-mer strings are hashed on the GPU with the values of the
    reference's raw-k-mer path, NTC64(kmerSeq, k) -- which are not the iterator's for k % 4 == 0 and for
    k-mers with U (btlbf_insert_kmers in include/btlbf.h has the details)."""

    @staticmethod
    def _is_kmer(x):
        return isinstance(x, (str, bytes, bytearray))

    def _kmers(self, x):
        """one k-mer string, or a list of them -> (uint8 buffer of n*k bytes, n)"""
        k = self.getKmerSize()
        items = [x] if self._is_kmer(x) else list(x)
        out = bytearray()
        for it in items:
            b = it.encode("latin-1") if isinstance(it, str) else bytes(it)
            if len(b) < k:
                raise ValueError("k-mer shorter than kmerSize")
            out += b[:k]
        return np.frombuffer(bytes(out), np.uint8), len(items)

    def insertKmers(self, kmers, stream=None):
        """KmerBloomFilter::insert(const char*) for a batch of raw k-mers (strings, or a uint8 buffer of n*k)"""
        buf, n = self._kmers(kmers) if not isinstance(kmers, np.ndarray) and not _is_torch_cuda(kmers) else (
            kmers, None)
        b = _Buf(buf)
        n = b.nbytes // self.getKmerSize() if n is None else n
        check(self._L.btlbf_insert_kmers(self._h, b.ptr, n, 0, ORDER_PARALLEL, b.mem, _stream_ptr(stream, b.keep)))

    def containsKmers(self, kmers, stream=None):
        """KmerBloomFilter::contains(const char*) for a batch of raw k-mers -> uint8 array"""
        buf, n = self._kmers(kmers) if not isinstance(kmers, np.ndarray) and not _is_torch_cuda(kmers) else (
            kmers, None)
        b = _Buf(buf)
        n = b.nbytes // self.getKmerSize() if n is None else n
        if b.mem == DEVICE:
            import torch

            out = torch.zeros(n, dtype=torch.uint8, device=b.keep.device)
            optr = C.c_void_p(out.data_ptr())
        else:
            out = np.zeros(max(n, 1), np.uint8)
            optr = C.c_void_p(out.ctypes.data)
        check(self._L.btlbf_contains_kmers(self._h, b.ptr, n, optr, b.mem, _stream_ptr(stream, b.keep)))
        return out[:n]

    def insert(self, x, stream=None):
        if self._is_kmer(x):
            return self.insertKmers(x, stream=stream)
        return super().insert(x, stream=stream)

    def contains(self, x, stream=None):
        if self._is_kmer(x):
            return bool(self.containsKmers(x, stream=stream)[0])
        r = super().contains(x, stream=stream)
        return bool(r[0]) if np.ndim(x) == 1 else r


def body_digest(body, first_word=0):
    """btlbf_digest restated with numpy over a host copy of a filter body (bytes / uint8 array): the definition in
    include/btlbf.h, for checking a digest against bytes that are at hand"""
    b = np.frombuffer(bytes(body), np.uint8) if not isinstance(body, np.ndarray) else body.view(np.uint8).ravel()
    if b.size % 8:
        b = np.concatenate([b, np.zeros(8 - b.size % 8, np.uint8)])
    w = b.view("<u8")
    nz = np.flatnonzero(w)

    def mix(z):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))

    with np.errstate(over="ignore"):
        m = mix(nz.astype(np.uint64) + np.uint64(first_word + 1)) | np.uint64(1)
        s = int((w[nz] * m).sum(dtype=np.uint64)) if nz.size else 0
        x = int(np.bitwise_xor.reduce(mix(w[nz] ^ m))) if nz.size else 0
    return s, x


def hash_kmers(kmers, h, k, device=0):
    """NTC64(kmerSeq, k) + NTE64 of raw k-mers (KmerBloomFilter's path; bytes of n*k) -> (hashes[n, h], valid[n])"""
    b = _Buf(kmers)
    n = b.nbytes // k
    hv = np.zeros((max(n, 1), h), np.uint64)
    ok = np.zeros(max(n, 1), np.uint8)
    if b.mem != HOST:
        raise ValueError("hash_kmers takes host buffers")
    check(_lib.load().btlbf_hash_kmers(k, h, b.ptr, n, C.c_void_p(hv.ctypes.data), C.c_void_p(ok.ctypes.data), HOST,
                                       device, None))
    return hv[:n], ok[:n]


def insertSeq(bloom, seq, numHashes=None, k=None):
    """insertSeq(KmerBloomFilter&, const string&, numHashes, k) of BloomFilterUtil.h:9-17 /
    swig/BloomFilter.i:59: every k-mer of `seq` in one fused launch.  numHashes / k must be the filter's
    own (the reference hashes with the arguments and inserts with the filter's hash count)."""
    if numHashes is not None and numHashes != bloom.getHashNum() or k is not None and k != bloom.getKmerSize():
        raise ValueError("insertSeq: numHashes / k differ from the filter's")
    s = seq.encode() if isinstance(seq, str) else bytes(seq)
    bloom.insertSeqs(np.frombuffer(s, np.uint8))


class CountingBloomFilter(_Filter):
    """uint8_t counting filter (reference: /root/reference/CountingBloomFilter.hpp, T = uint8_t)"""

    kind = COUNTING8

    def __init__(self, sizeInBytes=None, hashNum=None, kmerSize=None, countThreshold=0, path=None, device=0):
        if path is not None:
            super().__init__(self._load(path, countThreshold, device))
        else:
            super().__init__(self._create(sizeInBytes, hashNum, kmerSize, countThreshold, device))

    def size(self):
        return self._L.btlbf_size(self._h)

    def threshold(self):
        return self._L.btlbf_threshold(self._h)

    def popCount(self):
        out = C.c_uint64()
        check(self._L.btlbf_popcount(self._h, C.byref(out)))
        return out.value

    def filtered_popcount(self):
        out = C.c_uint64()
        check(self._L.btlbf_filtered_popcount(self._h, C.byref(out)))
        return out.value

    def FPR(self):
        return (self.popCount() / self.size()) ** self.getHashNum()

    def filtered_FPR(self):
        return (self.filtered_popcount() / self.size()) ** self.getHashNum()

    def insert(self, hashes, serial=False, stream=None):  # = incrementMin (CountingBloomFilter.hpp:198-204)
        self.incrementMin(hashes, serial, stream)

    def incrementMin(self, hashes, serial=False, stream=None):
        b, n = self._rows(hashes)
        check(self._L.btlbf_insert_hashes(self._h, b.ptr, n, INCREMENT_MIN,
                                          ORDER_SERIAL if serial else ORDER_PARALLEL, b.mem,
                                          _stream_ptr(stream, b.keep)))

    def incrementAll(self, hashes, serial=False, stream=None):
        b, n = self._rows(hashes)
        check(self._L.btlbf_insert_hashes(self._h, b.ptr, n, INCREMENT_ALL,
                                          ORDER_SERIAL if serial else ORDER_PARALLEL, b.mem,
                                          _stream_ptr(stream, b.keep)))

    def minCount(self, hashes, stream=None):
        return self._rows_out(self._L.btlbf_min_count_hashes, hashes, stream=stream)

    def contains(self, hashes, stream=None):
        return self._rows_out(self._L.btlbf_contains_hashes, hashes, stream=stream)

    def insertAndCheck(self, hashes, serial=True, stream=None):
        return self._rows_out(self._L.btlbf_insert_and_check_hashes, hashes,
                              ORDER_SERIAL if serial else ORDER_PARALLEL, stream=stream)

    def insertSeqs(self, seq, starts=None, read_len=0, increment_all=False, serial=False, stream=None):
        self._insert_seqs(seq, starts, read_len, INCREMENT_ALL if increment_all else INCREMENT_MIN,
                          ORDER_SERIAL if serial else ORDER_PARALLEL, stream)

    def minCountSeqs(self, seq, starts=None, read_len=0, stream=None):
        b = _Buf(seq)
        lay, keep = _layout(starts, read_len, b.mem)
        n = b.nbytes
        if b.mem == DEVICE:
            import torch

            mn = torch.zeros(n, dtype=torch.uint8, device=b.keep.device)
            valid = torch.zeros((n + 63) // 64, dtype=torch.int64, device=b.keep.device)
            p1, p2 = C.c_void_p(mn.data_ptr()), C.c_void_p(valid.data_ptr())
        else:
            mn = np.zeros(max(n, 1), np.uint8)
            valid = _bitmap(n)
            p1, p2 = C.c_void_p(mn.ctypes.data), C.c_void_p(valid.ctypes.data)
        check(self._L.btlbf_min_count_seqs(self._h, b.ptr, n, C.byref(lay) if lay else None, p1, p2, b.mem,
                                           _stream_ptr(stream, b.keep)))
        return mn[:n], valid


class RankSupport:
    """sdsl::bit_vector_il<512> + rank_support_il<1> over a bit filter, in HBM (btlbf_rank_*): what the
    reference's miBF uses to turn a set bit into an index of its ID array (MIBloomFilter.hpp:527,801-803)"""

    def __init__(self, bloom):
        self._L = _lib.load()
        h = C.c_void_p()
        check(self._L.btlbf_rank_create(C.byref(h), bloom._h))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.btlbf_rank_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ones(self):
        return self._L.btlbf_rank_ones(self._h)

    def interleaved(self):
        """the interleaved vector as (n_blocks, 9) uint64: column 0 = set bits before the block"""
        out = np.zeros(self._L.btlbf_rank_words(self._h), np.uint64)
        check(self._L.btlbf_rank_download(self._h, C.c_void_p(out.ctypes.data)))
        return out.reshape(-1, 9)

    def rank(self, values, hashes=False):
        """(rank, bit) of positions, or of hash values reduced modulo the filter size (getRankPos)"""
        v = np.ascontiguousarray(values, np.uint64).ravel()
        r = np.zeros(max(v.size, 1), np.uint64)
        b = np.zeros(max(v.size, 1), np.uint8)
        check(self._L.btlbf_rank_query(self._h, C.c_void_p(v.ctypes.data), v.size, int(bool(hashes)),
                                       C.c_void_p(r.ctypes.data), C.c_void_p(b.ctypes.data), HOST, None))
        return r[: v.size], b[: v.size]


def _hash_seqs(seq, k, h, seeds, h2, starts, read_len, device, stream):
    L = _lib.load()
    b = _Buf(seq)
    lay, keep = _layout(starts, read_len, b.mem)
    n = b.nbytes
    sarr = None
    ns = 0
    if seeds is not None:
        ns = len(seeds)
        sarr = (C.c_char_p * ns)(*[s.encode() if isinstance(s, str) else s for s in seeds])
    if b.mem == DEVICE:
        import torch

        hv = torch.zeros((n, h), dtype=torch.int64, device=b.keep.device)
        valid = torch.zeros((n + 63) // 64, dtype=torch.int64, device=b.keep.device)
        st = torch.zeros(n, dtype=torch.int64, device=b.keep.device) if seeds is not None else None
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    else:
        hv = np.zeros((max(n, 1), h), np.uint64)
        valid = _bitmap(n)
        st = np.zeros(max(n, 1), np.uint64) if seeds is not None else None
        ptr = lambda a: C.c_void_p(a.ctypes.data) if a is not None else None  # noqa: E731
    check(L.btlbf_hash_seqs(k, h, sarr, ns, h2, b.ptr, n, C.byref(lay) if lay else None, ptr(hv), ptr(valid),
                            ptr(st), b.mem, device, _stream_ptr(stream, b.keep)))
    return hv[:n], valid, (st[:n] if st is not None else None)


def hash_seqs(seq, h, k, starts=None, read_len=0, device=0, stream=None):
    """ntHashIterator(seq, h, k) over a whole buffer (vendor/ntHashIterator.hpp:38):
    -> (hashes[n, h], valid_bits).  Window p is emitted by the iterator iff its valid bit is set."""
    hv, valid, _ = _hash_seqs(seq, k, h, None, 0, starts, read_len, device, stream)
    return hv, valid


def sthash_seqs(seq, seeds, h2, k, starts=None, read_len=0, device=0, stream=None):
    """stHashIterator(seq, parseSeed(seeds), len(seeds), h2, k) (vendor/stHashIterator.hpp:53):
    -> (hashes[n, len(seeds)*h2], valid_bits, strand_bits[n])"""
    return _hash_seqs(seq, k, len(seeds) * h2, seeds, h2, starts, read_len, device, stream)


def synth_reads_device(seed, first, n_reads, read_len, device=0, stream=None):
    """synthetic reads of SURVEY.md 8d generated in HBM -> torch uint8 tensor (n_reads*read_len)"""
    import torch

    out = torch.empty(n_reads * read_len, dtype=torch.uint8, device="cuda:%d" % device)
    check(_lib.load().btlbf_synth_reads(C.c_void_p(out.data_ptr()), seed, first, n_reads, read_len, device,
                                        _stream_ptr(stream, out)))
    return out


def count_per_seq(hit_bits, valid_bits, n_bytes, k, starts=None, read_len=0, device=0, stream=None):
    """per-sequence (hits, clean windows) from the per-window bitmaps of containsSeqs / insertAndCheckSeqs
    over a buffer of n_bytes bytes (btlbf_count_per_seq): what a read classifier needs per read"""
    hb = _Buf(hit_bits, np.uint64)
    lay, keep = _layout(starts, read_len, hb.mem)
    if lay is None:
        raise ValueError("count_per_seq needs starts or read_len")
    n_seqs = lay.n_seqs if starts is not None else n_bytes // read_len
    vb = _Buf(valid_bits, np.uint64) if valid_bits is not None else None
    if hb.mem == DEVICE:
        import torch

        hits = torch.zeros(n_seqs, dtype=torch.int32, device=hb.keep.device)
        valid = torch.zeros(n_seqs, dtype=torch.int32, device=hb.keep.device)
        p1, p2 = C.c_void_p(hits.data_ptr()), C.c_void_p(valid.data_ptr())
    else:
        hits, valid = np.zeros(n_seqs, np.uint32), np.zeros(n_seqs, np.uint32)
        p1, p2 = C.c_void_p(hits.ctypes.data), C.c_void_p(valid.ctypes.data)
    check(_lib.load().btlbf_count_per_seq(hb.ptr, vb.ptr if vb is not None else None, int(n_bytes), C.byref(lay), int(k),
                                           p1, p2, hb.mem, device, _stream_ptr(stream, hb.keep)))
    return hits, valid


def fastx_batches(path, k, per_line=False, batch_bytes=0, pageable=True, byte_range=None, fmt=None):
    """Iterate over the parser's batches as (bases: bytes, starts: list[int]) -- the host-side reader
    behind insertFile (btlbf_fastx_open / btlbf_fastx_next); needs no GPU.  byte_range=(begin, end) with
    fmt in {"fasta", "fastq", "plain"}: only the records that start in that byte range of an
    uncompressed file (btlbf_fastx_open_range, the unit of parallel parsing)."""
    L = _lib.load()
    r = C.c_void_p()
    flags = (1 if per_line else 0) | (2 if pageable else 0)
    if byte_range is None:
        check(L.btlbf_fastx_open(C.byref(r), str(path).encode(), flags, int(k), int(batch_bytes)))
    else:
        check(L.btlbf_fastx_open_range(C.byref(r), str(path).encode(), flags, int(k), int(batch_bytes),
                                       {"fasta": 1, "fastq": 2, "plain": 3}[fmt], int(byte_range[0]),
                                       int(byte_range[1])))
    try:
        while True:
            b, s = C.c_void_p(), C.c_void_p()
            nb, ns = C.c_uint64(), C.c_uint64()
            check(L.btlbf_fastx_next(r, C.byref(b), C.byref(nb), C.byref(s), C.byref(ns)))
            if ns.value == 0:
                return
            bases = C.string_at(b.value, nb.value) if nb.value else b""
            starts = list((C.c_uint64 * (ns.value + 1)).from_address(s.value))
            yield bases, starts
    finally:
        L.btlbf_fastx_close(r)

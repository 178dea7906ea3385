"""ctypes bindings for the two TEST-ONLY checkers.

* ``Oracle``  -> oracle/libbtl_oracle.so  (plain-C CPU restatement, oracle/btl_oracle.c)
* ``Ref``     -> oracle/_ref/libbtlref.so (genuine reference headers behind oracle/ref_driver.cpp)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libbtl_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libbtlref.so")

u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(verbose=False):
    """(Re)build the checkers: always the C restatement; the reference build only where
    /root/reference exists (oracle/Makefile decides)."""
    out = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if verbose:
        print(out.stdout)


def _bytes(seq):
    return seq if isinstance(seq, (bytes, bytearray)) else seq.encode("latin-1")


def _hashes(a, h):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, h)
    return a, a.shape[0]


class Oracle:
    """numpy-facing wrapper of oracle/btl_oracle.h"""

    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            build()
        L = self.L = C.CDLL(path)
        L.bo_seed.restype = C.c_uint64
        L.bo_seed.argtypes = [C.c_ubyte]
        L.bo_srol.restype = C.c_uint64
        L.bo_srol.argtypes = [C.c_uint64]
        L.bo_sror.restype = C.c_uint64
        L.bo_sror.argtypes = [C.c_uint64]
        L.bo_srol_n.restype = C.c_uint64
        L.bo_srol_n.argtypes = [C.c_uint64, C.c_uint]
        L.bo_extra.restype = C.c_uint64
        L.bo_extra.argtypes = [C.c_uint64, C.c_uint, C.c_uint]
        L.bo_nthash_seq.restype = C.c_size_t
        L.bo_nthash_seq.argtypes = [C.c_char_p, C.c_size_t, C.c_uint, C.c_uint, u64p, u64p, C.c_size_t]
        L.bo_sthash_seq.restype = C.c_size_t
        L.bo_sthash_seq.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p), C.c_uint, C.c_uint,
                                    C.c_uint, u64p, u64p, u8p, C.c_size_t]
        L.bo_kmer_hashes.argtypes = [C.c_char_p, C.c_size_t, C.c_uint, C.c_uint, u64p, u8p]
        L.bo_bf_insert.argtypes = [u8p, C.c_uint64, C.c_uint, u64p, C.c_size_t]
        L.bo_bf_contains.argtypes = [u8p, C.c_uint64, C.c_uint, u64p, C.c_size_t, u8p]
        L.bo_bf_insert_and_check.argtypes = [u8p, C.c_uint64, C.c_uint, u64p, C.c_size_t, u8p]
        L.bo_bf_popcount.restype = C.c_uint64
        L.bo_bf_popcount.argtypes = [u8p, C.c_uint64]
        L.bo_bf_insert_seq.argtypes = [u8p, C.c_uint64, C.c_uint, C.c_uint, C.c_char_p, C.c_size_t]
        L.bo_bf_contains_seq_dense.argtypes = [u8p, C.c_uint64, C.c_uint, C.c_uint, C.c_char_p,
                                               C.c_size_t, u8p, u8p]
        L.bo_cbf_round_bytes.restype = C.c_uint64
        L.bo_cbf_round_bytes.argtypes = [C.c_uint64]
        L.bo_cbf_increment_min.argtypes = [u8p, C.c_uint64, C.c_uint, u64p, C.c_size_t]
        L.bo_cbf_increment_all.argtypes = [u8p, C.c_uint64, C.c_uint, u64p, C.c_size_t]
        L.bo_cbf_insert_and_check.argtypes = [u8p, C.c_uint64, C.c_uint, C.c_uint, u64p, C.c_size_t, u8p]
        L.bo_cbf_query.argtypes = [u8p, C.c_uint64, C.c_uint, C.c_uint, u64p, C.c_size_t, u8p, u8p]
        L.bo_cbf_popcount.restype = C.c_uint64
        L.bo_cbf_popcount.argtypes = [u8p, C.c_uint64]
        L.bo_cbf_filtered_popcount.restype = C.c_uint64
        L.bo_cbf_filtered_popcount.argtypes = [u8p, C.c_uint64, C.c_uint]
        L.bo_bf_header.restype = C.c_int
        L.bo_bf_header.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint, C.c_uint, C.c_double,
                                   C.c_uint64, C.c_uint64]
        L.bo_cbf_header.restype = C.c_int
        L.bo_cbf_header.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint, C.c_uint,
                                    C.c_uint]
        L.bo_synth_reads.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint, u8p]
        L.bo_bench_bf.restype = C.c_int
        L.bo_bench_bf.argtypes = [C.c_uint64, C.c_uint, C.c_uint, C.c_uint, C.c_uint64, C.c_uint64,
                                  C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_double)]

    # -- hashing -------------------------------------------------------------------------
    def nthash_seq(self, seq, h, k):
        s = _bytes(seq)
        cap = max(len(s), 1)
        pos = np.zeros(cap, np.uint64)
        hv = np.zeros(cap * h, np.uint64)
        n = self.L.bo_nthash_seq(s, len(s), h, k, pos, hv, cap)
        return pos[:n].copy(), hv[: n * h].reshape(n, h).copy()

    def sthash_seq(self, seq, seeds, h2, k):
        s = _bytes(seq)
        cap = max(len(s), 1)
        m = len(seeds) * h2
        arr = (C.c_char_p * len(seeds))(*[_bytes(x) for x in seeds])
        pos = np.zeros(cap, np.uint64)
        hv = np.zeros(cap * m, np.uint64)
        st = np.zeros(cap * m, np.uint8)
        n = self.L.bo_sthash_seq(s, len(s), arr, len(seeds), h2, k, pos, hv, st, cap)
        return pos[:n].copy(), hv[: n * m].reshape(n, m).copy(), st[: n * m].reshape(n, m).copy()

    def kmer_hashes(self, kmers, k, h):
        """raw k-mers (bytes, n*k long) -> (hashes [n, h], valid [n]): the KmerBloomFilter path"""
        s = _bytes(kmers)
        n = len(s) // k
        hv = np.zeros(max(n, 1) * h, np.uint64)
        ok = np.zeros(max(n, 1), np.uint8)
        self.L.bo_kmer_hashes(s, n, k, h, hv, ok)
        return hv[: n * h].reshape(n, h).copy(), ok[:n].copy()

    # -- bit filter ----------------------------------------------------------------------
    def bf_insert(self, filt, size_bits, h, hashes):
        a, n = _hashes(hashes, h)
        self.L.bo_bf_insert(filt, size_bits, h, a, n)

    def bf_contains(self, filt, size_bits, h, hashes):
        a, n = _hashes(hashes, h)
        out = np.zeros(n, np.uint8)
        self.L.bo_bf_contains(filt, size_bits, h, a, n, out)
        return out

    def bf_insert_and_check(self, filt, size_bits, h, hashes):
        a, n = _hashes(hashes, h)
        out = np.zeros(n, np.uint8)
        self.L.bo_bf_insert_and_check(filt, size_bits, h, a, n, out)
        return out

    def bf_popcount(self, filt, size_bits):
        return int(self.L.bo_bf_popcount(filt, size_bits))

    def bf_insert_seq(self, filt, size_bits, h, k, seq):
        s = _bytes(seq)
        self.L.bo_bf_insert_seq(filt, size_bits, h, k, s, len(s))

    def bf_contains_seq_dense(self, filt, size_bits, h, k, seq):
        s = _bytes(seq)
        nw = max(len(s) - k + 1, 0)
        hit = np.zeros(max(nw, 1), np.uint8)
        valid = np.zeros(max(nw, 1), np.uint8)
        self.L.bo_bf_contains_seq_dense(filt, size_bits, h, k, s, len(s), hit, valid)
        return hit[:nw], valid[:nw]

    # -- counting filter -----------------------------------------------------------------
    def cbf_round_bytes(self, b):
        return int(self.L.bo_cbf_round_bytes(b))

    def cbf_increment_min(self, c, h, hashes):
        a, n = _hashes(hashes, h)
        self.L.bo_cbf_increment_min(c, c.size, h, a, n)

    def cbf_increment_all(self, c, h, hashes):
        a, n = _hashes(hashes, h)
        self.L.bo_cbf_increment_all(c, c.size, h, a, n)

    def cbf_insert_and_check(self, c, h, thr, hashes):
        a, n = _hashes(hashes, h)
        out = np.zeros(n, np.uint8)
        self.L.bo_cbf_insert_and_check(c, c.size, h, thr, a, n, out)
        return out

    def cbf_query(self, c, h, thr, hashes):
        a, n = _hashes(hashes, h)
        mn = np.zeros(n, np.uint8)
        ct = np.zeros(n, np.uint8)
        self.L.bo_cbf_query(c, c.size, h, thr, a, n, mn, ct)
        return mn, ct

    def cbf_popcount(self, c):
        return int(self.L.bo_cbf_popcount(c, c.size))

    def cbf_filtered_popcount(self, c, thr):
        return int(self.L.bo_cbf_filtered_popcount(c, c.size, thr))

    # -- headers -------------------------------------------------------------------------
    def bf_header(self, size_bits, h, k, dfpr=0.0, n_entry=0, t_entry=0):
        buf = C.create_string_buffer(1024)
        n = self.L.bo_bf_header(buf, 1024, size_bits, h, k, dfpr, n_entry, t_entry)
        return buf.raw[:n]

    def cbf_header(self, size, size_bytes, h, k, bits_per_counter=8):
        buf = C.create_string_buffer(1024)
        n = self.L.bo_cbf_header(buf, 1024, size, size_bytes, h, k, bits_per_counter)
        return buf.raw[:n]

    # -- synthetic reads / timed port ----------------------------------------------------
    def synth_reads(self, seed, first, n, read_len):
        out = np.zeros(n * read_len, np.uint8)
        self.L.bo_synth_reads(seed, first, n, read_len, out)
        return out

    def bench_bf(self, n_reads, read_len, k, h, bits, seed_ins, seed_qry, threads=0, prefault=1):
        out = (C.c_double * 6)()
        rc = self.L.bo_bench_bf(n_reads, read_len, k, h, bits, seed_ins, seed_qry, threads, prefault, out)
        if rc:
            raise RuntimeError("bo_bench_bf rc=%d" % rc)
        return dict(t_insert=out[0], t_query=out[1], hits=int(out[2]), kmers=int(out[3]),
                    threads=int(out[4]), popcount=int(out[5]))


class Ref:
    """wrapper of oracle/ref_driver.cpp (genuine reference).  ``Ref.available()`` is False on
    machines that hold neither /root/reference nor a prebuilt oracle/_ref/libbtlref.so."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self, path=REF_SO):
        L = self.L = C.CDLL(path)
        vp = C.c_void_p
        L.ref_nthash_seq.restype = C.c_size_t
        L.ref_nthash_seq.argtypes = [C.c_char_p, C.c_size_t, C.c_uint, C.c_uint, u64p, u64p, C.c_size_t]
        L.ref_sthash_seq.restype = C.c_size_t
        L.ref_sthash_seq.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p), C.c_uint, C.c_uint,
                                     C.c_uint, u64p, u64p, u8p, C.c_size_t]
        L.ref_kmer_hashes.argtypes = [C.c_char_p, C.c_uint, C.c_uint, u64p]
        L.ref_bf_new.restype = vp
        L.ref_bf_new.argtypes = [C.c_size_t, C.c_uint, C.c_uint]
        L.ref_bf_load.restype = vp
        L.ref_bf_load.argtypes = [C.c_char_p]
        L.ref_bf_free.argtypes = [vp]
        L.ref_bf_bytes.restype = C.POINTER(C.c_uint8)
        L.ref_bf_bytes.argtypes = [vp]
        for f in ("ref_bf_size_bits", "ref_bf_size_bytes", "ref_bf_pop"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [vp]
        for f in ("ref_bf_hash_num", "ref_bf_kmer_size"):
            getattr(L, f).restype = C.c_uint
            getattr(L, f).argtypes = [vp]
        L.ref_bf_fpr.restype = C.c_double
        L.ref_bf_fpr.argtypes = [vp]
        L.ref_bf_set_entries.argtypes = [vp, C.c_uint64, C.c_uint64]
        L.ref_bf_store.argtypes = [vp, C.c_char_p]
        L.ref_bf_insert.argtypes = [vp, u64p, C.c_size_t]
        L.ref_bf_contains.argtypes = [vp, u64p, C.c_size_t, u8p]
        L.ref_bf_insert_and_check.argtypes = [vp, u64p, C.c_size_t, u8p]
        L.ref_bf_insert_seq.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.ref_bf_contains_seq.restype = C.c_size_t
        L.ref_bf_contains_seq.argtypes = [vp, C.c_char_p, C.c_size_t, u64p, u8p, C.c_size_t]
        L.ref_kbf_insert_kmer.argtypes = [vp, C.c_char_p]
        L.ref_kbf_contains_kmer.restype = C.c_int
        L.ref_kbf_contains_kmer.argtypes = [vp, C.c_char_p]
        L.ref_cbf_new.restype = vp
        L.ref_cbf_new.argtypes = [C.c_size_t, C.c_uint, C.c_uint, C.c_uint]
        L.ref_cbf_load.restype = vp
        L.ref_cbf_load.argtypes = [C.c_char_p, C.c_uint]
        L.ref_cbf_free.argtypes = [vp]
        for f in ("ref_cbf_size", "ref_cbf_size_bytes", "ref_cbf_popcount", "ref_cbf_filtered_popcount"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [vp]
        for f in ("ref_cbf_hash_num", "ref_cbf_kmer_size"):
            getattr(L, f).restype = C.c_uint
            getattr(L, f).argtypes = [vp]
        L.ref_cbf_read.argtypes = [vp, u8p]
        L.ref_cbf_store.argtypes = [vp, C.c_char_p]
        L.ref_cbf_update.argtypes = [vp, u64p, C.c_size_t, C.c_int, u8p]
        L.ref_cbf_query.argtypes = [vp, u64p, C.c_size_t, u8p, u8p]
        L.ref_synth_reads.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint, u8p]
        L.ref_bf_insert_synth.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint]
        L.ref_bf_count_synth.restype = C.c_uint64
        L.ref_bf_count_synth.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint]
        L.ref_cbf_update_synth.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint, C.c_int]
        L.ref_bench_bf.restype = C.c_int
        L.ref_bench_bf.argtypes = [C.c_uint64, C.c_uint, C.c_uint, C.c_uint, C.c_uint64, C.c_uint64,
                                   C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.ref_bench_synth_only.restype = C.c_double
        L.ref_bench_synth_only.argtypes = [C.c_uint64, C.c_uint, C.c_uint64, C.c_int]
        L.ref_bench_bf_digest.restype = C.c_int
        L.ref_bench_bf_digest.argtypes = L.ref_bench_bf.argtypes + [C.POINTER(C.c_uint64)]
        L.ref_bf_digest.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.ref_cbf_digest.argtypes = [vp, C.c_uint, C.POINTER(C.c_uint64)]
        L.ref_cbf_increment_all_synth.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint]
        L.ref_cbf_count_synth.restype = C.c_uint64
        L.ref_cbf_count_synth.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint]
        L.ref_bf_spaced_synth.restype = C.c_uint64
        L.ref_bf_spaced_synth.argtypes = [vp, C.POINTER(C.c_char_p), C.c_uint, C.c_uint, C.c_uint64, C.c_uint64,
                                          C.c_uint64, C.c_uint, C.c_int]

    def nthash_seq(self, seq, h, k):
        s = _bytes(seq)
        cap = max(len(s), 1)
        pos = np.zeros(cap, np.uint64)
        hv = np.zeros(cap * h, np.uint64)
        n = self.L.ref_nthash_seq(s, len(s), h, k, pos, hv, cap)
        return pos[:n].copy(), hv[: n * h].reshape(n, h).copy()

    def sthash_seq(self, seq, seeds, h2, k):
        s = _bytes(seq)
        cap = max(len(s), 1)
        m = len(seeds) * h2
        arr = (C.c_char_p * len(seeds))(*[_bytes(x) for x in seeds])
        pos = np.zeros(cap, np.uint64)
        hv = np.zeros(cap * m, np.uint64)
        st = np.zeros(cap * m, np.uint8)
        n = self.L.ref_sthash_seq(s, len(s), arr, len(seeds), h2, k, pos, hv, st, cap)
        return pos[:n].copy(), hv[: n * m].reshape(n, m).copy(), st[: n * m].reshape(n, m).copy()

    def kmer_hashes(self, kmer, k, h):
        out = np.zeros(h, np.uint64)
        self.L.ref_kmer_hashes(_bytes(kmer), k, h, out)
        return out

    class BF:
        def __init__(self, ref, bits=None, h=None, k=None, path=None):
            self.L = ref.L
            self.p = self.L.ref_bf_load(_bytes(path)) if path else self.L.ref_bf_new(bits, h, k)
            self.h = self.L.ref_bf_hash_num(self.p)
            self.k = self.L.ref_bf_kmer_size(self.p)
            self.bits = self.L.ref_bf_size_bits(self.p)

        def close(self):
            if self.p:
                self.L.ref_bf_free(self.p)
                self.p = None

        __del__ = close

        def bytes(self):
            n = self.L.ref_bf_size_bytes(self.p)
            return np.ctypeslib.as_array(self.L.ref_bf_bytes(self.p), shape=(n,)).copy()

        def insert(self, hashes):
            a, n = _hashes(hashes, self.h)
            self.L.ref_bf_insert(self.p, a, n)

        def contains(self, hashes):
            a, n = _hashes(hashes, self.h)
            out = np.zeros(n, np.uint8)
            self.L.ref_bf_contains(self.p, a, n, out)
            return out

        def insert_and_check(self, hashes):
            a, n = _hashes(hashes, self.h)
            out = np.zeros(n, np.uint8)
            self.L.ref_bf_insert_and_check(self.p, a, n, out)
            return out

        def insert_seq(self, seq):
            s = _bytes(seq)
            self.L.ref_bf_insert_seq(self.p, s, len(s))

        def contains_seq(self, seq):
            s = _bytes(seq)
            cap = max(len(s), 1)
            pos = np.zeros(cap, np.uint64)
            res = np.zeros(cap, np.uint8)
            n = self.L.ref_bf_contains_seq(self.p, s, len(s), pos, res, cap)
            return pos[:n].copy(), res[:n].copy()

        def insert_synth(self, seed, first, n, read_len):
            self.L.ref_bf_insert_synth(self.p, seed, first, n, read_len)

        def count_synth(self, seed, first, n, read_len):
            return int(self.L.ref_bf_count_synth(self.p, seed, first, n, read_len))

        def spaced_synth(self, seeds, h2, seed, first, n, read_len, query=False):
            """stHashIterator over synthetic reads: insert (query=False) or count of k-mers found (query=True)"""
            arr = (C.c_char_p * len(seeds))(*[_bytes(x) for x in seeds])
            return int(self.L.ref_bf_spaced_synth(self.p, arr, len(seeds), h2, seed, first, n, read_len, int(query)))

        def digest(self):
            """(sum, xor) of the body where it lies: the definition of btlbf_digest (include/btlbf.h)"""
            out = (C.c_uint64 * 3)()
            self.L.ref_bf_digest(self.p, out)
            self.last_pop = int(out[2])  # set bits, counted in the same sweep
            return int(out[0]), int(out[1])

        def insert_kmer(self, kmer):
            self.L.ref_kbf_insert_kmer(self.p, _bytes(kmer))

        def contains_kmer(self, kmer):
            return bool(self.L.ref_kbf_contains_kmer(self.p, _bytes(kmer)))

        def pop(self):
            return int(self.L.ref_bf_pop(self.p))

        def fpr(self):
            return float(self.L.ref_bf_fpr(self.p))

        def set_entries(self, n, t):
            self.L.ref_bf_set_entries(self.p, n, t)

        def store(self, path):
            self.L.ref_bf_store(self.p, _bytes(path))

    class CBF:
        def __init__(self, ref, nbytes=None, h=None, k=None, thr=1, path=None):
            self.L = ref.L
            self.p = self.L.ref_cbf_load(_bytes(path), thr) if path else self.L.ref_cbf_new(nbytes, h, k, thr)
            self.h = self.L.ref_cbf_hash_num(self.p)
            self.k = self.L.ref_cbf_kmer_size(self.p)
            self.size = self.L.ref_cbf_size(self.p)
            self.size_bytes = self.L.ref_cbf_size_bytes(self.p)

        def close(self):
            if self.p:
                self.L.ref_cbf_free(self.p)
                self.p = None

        __del__ = close

        def counters(self):
            out = np.zeros(self.size, np.uint8)
            self.L.ref_cbf_read(self.p, out)
            return out

        def insert(self, hashes):
            a, n = _hashes(hashes, self.h)
            self.L.ref_cbf_update(self.p, a, n, 0, np.zeros(1, np.uint8))

        def increment_all(self, hashes):
            a, n = _hashes(hashes, self.h)
            self.L.ref_cbf_update(self.p, a, n, 1, np.zeros(1, np.uint8))

        def insert_and_check(self, hashes):
            a, n = _hashes(hashes, self.h)
            out = np.zeros(max(n, 1), np.uint8)
            self.L.ref_cbf_update(self.p, a, n, 2, out)
            return out[:n]

        def query(self, hashes):
            a, n = _hashes(hashes, self.h)
            mn = np.zeros(max(n, 1), np.uint8)
            ct = np.zeros(max(n, 1), np.uint8)
            self.L.ref_cbf_query(self.p, a, n, mn, ct)
            return mn[:n], ct[:n]

        def update_synth(self, seed, first, n, read_len, op):
            self.L.ref_cbf_update_synth(self.p, seed, first, n, read_len, op)

        def increment_all_synth(self, seed, first, n, read_len):
            self.L.ref_cbf_increment_all_synth(self.p, seed, first, n, read_len)

        def count_synth(self, seed, first, n, read_len):
            return int(self.L.ref_cbf_count_synth(self.p, seed, first, n, read_len))

        def digest(self, thr=1):
            """(sum, xor); self.last_counts = (non-zero counters, counters >= thr) from the same sweep"""
            out = (C.c_uint64 * 4)()
            self.L.ref_cbf_digest(self.p, thr, out)
            self.last_counts = (int(out[2]), int(out[3]))
            return int(out[0]), int(out[1])

        def popcount(self):
            return int(self.L.ref_cbf_popcount(self.p))

        def filtered_popcount(self):
            return int(self.L.ref_cbf_filtered_popcount(self.p))

        def store(self, path):
            self.L.ref_cbf_store(self.p, _bytes(path))

    def bf(self, bits=None, h=None, k=None, path=None):
        return Ref.BF(self, bits, h, k, path)

    def cbf(self, nbytes=None, h=None, k=None, thr=1, path=None):
        return Ref.CBF(self, nbytes, h, k, thr, path)

    def synth_reads(self, seed, first, n, read_len):
        out = np.zeros(n * read_len, np.uint8)
        self.L.ref_synth_reads(seed, first, n, read_len, out)
        return out

    def bench_bf(self, n_reads, read_len, k, h, bits, seed_ins, seed_qry, threads=0, prefault=1,
                 skip_pop=1, digest=False):
        out = (C.c_double * 6)()
        dg = (C.c_uint64 * 2)()
        self.L.ref_bench_bf_digest(n_reads, read_len, k, h, bits, seed_ins, seed_qry, threads, prefault,
                                   skip_pop, out, dg if digest else None)
        r = dict(t_insert=out[0], t_query=out[1], hits=int(out[2]), kmers=int(out[3]),
                 threads=int(out[4]), popcount=int(out[5]))
        if digest:
            r["digest"] = (int(dg[0]), int(dg[1]))
        return r

    def bench_synth_only(self, n_reads, read_len, seed, threads=0):
        return float(self.L.ref_bench_synth_only(n_reads, read_len, seed, threads))

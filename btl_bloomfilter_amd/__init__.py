"""MI355X-native k-mer Bloom filter engine -- Python host side.

Thin callers of the C ABI in include/btlbf.h (libbtlbf.so: hand-written HIP for gfx950).  The
classes mirror the reference's public interface for the hot path (BloomFilter /
CountingBloomFilter insert / contains / insertAndCheck, ntHashIterator / stHashIterator hash
streams, BTLBloomFilter_v1 files).  There is no CPU implementation behind them."""
from . import _lib  # noqa: F401
from .engine import (BloomFilter, KmerBloomFilter, insertSeq, CountingBloomFilter, hash_seqs, hash_kmers, sthash_seqs, synth_reads_device,  # noqa: F401
                     bits_to_bool, fastx_batches, count_per_seq, RankSupport)

__all__ = ["BloomFilter", "KmerBloomFilter", "insertSeq", "CountingBloomFilter", "hash_seqs", "hash_kmers", "sthash_seqs", "synth_reads_device",
           "bits_to_bool", "fastx_batches", "count_per_seq", "RankSupport"]

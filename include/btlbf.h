/* include/btlbf.h -- C ABI of the MI355X-native k-mer Bloom filter engine.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  The C++ shims
 * in include/btlbf/ (same class names and methods as the reference's public headers) and the
 * Python host code in btl_bloomfilter_amd/ are thin callers of these entry points; every entry
 * point that computes runs hand-written HIP kernels for gfx950 (btl_bloomfilter_amd/csrc/).
 * There is no CPU fallback: on a machine without a usable GPU the compute calls return
 * BTLBF_EHIP and btlbf_last_error() says why.
 *
 * The reference (bcgsc/btl_bloomfilter, /root/reference) has no FFI layer; its boundary is its
 * C++ header API.  Each entry below cites the reference interface it replaces (file:line).
 *
 * Conventions
 *  - every function returns 0 (BTLBF_OK) or a BTLBF_E* code; btlbf_last_error() (thread-local)
 *    holds a message for the last failure on the calling thread.
 *  - `mem` says where the caller's buffers live: BTLBF_HOST (pageable/pinned host memory; the
 *    library stages through its own device scratch) or BTLBF_DEVICE (HBM pointers, used as-is).
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream; BTLBF_STREAM_PER_THREAD = HIP's
 *    per-thread default stream, which lets BTLBF_HOST calls of different host threads overlap).  Work on one
 *    filter is ordered by the stream; BTLBF_HOST calls synchronise before returning.
 *  - threads: every entry point that takes a filter holds the filter's internal lock for its whole duration
 *    (a filter keeps device scratch between calls), so ONE filter may be driven from many host threads -- the
 *    calls are serialised, the parallelism is inside each batch call -- and different filters run
 *    concurrently.  One exception, for the reference's per-read query loops: a read-only call on a small
 *    BTLBF_HOST buffer (btlbf_contains_hashes, btlbf_contains_seqs through the calling thread's pinned mailbox)
 *    gives the lock back as soon as its kernel is launched and only then waits for it, so such calls from
 *    different threads overlap (pass BTLBF_STREAM_PER_THREAD).  This is what makes the C++ shims safe for the reference's own threading pattern
 *    (OpenMP threads calling insert()/contains() on one filter, Tests/AdHoc/ParallelFilter.cpp:104-122;
 *    the reference itself relies on byte atomics, BloomFilter.hpp:177,191,206-210).  btlbf_destroy must not
 *    race with other calls on the same filter.  On one thread, btlbf_route_seqs may run on a second
 *    stream while btlbf_apply_routed* runs on the first (they share no scratch).
 *  - a "sequence buffer" is `len` bytes of nucleotide text.  Windows (k-mers) are identified by
 *    the byte offset p of their first base.  A window is *clean* iff all k bytes are bases the
 *    reference accepts (seedTab != 0, vendor/nthash.hpp:195-228: A C G T U, either case, and
 *    the raw bytes 1 3 4 5 7) and it does not cross a sequence boundary.  Exactly the clean
 *    windows are the k-mers ntHashIterator emits (vendor/ntHashIterator.hpp:59-86).
 *  - sequence boundaries inside a buffer are described by a btlbf_layout:
 *      starts != NULL : n_seqs+1 ascending byte offsets (starts[0]=0 .. starts[n_seqs]=len)
 *      else read_len>0: back-to-back reads of exactly read_len bytes (len % read_len == 0)
 *      else           : the whole buffer is one sequence
 *    `starts` lives in the same memory space as the sequence buffer.
 *  - per-window results are bitmaps of ceil(len/64) uint64_t words: bit (p & 63) of word p >> 6
 *    belongs to the window starting at byte p; windows that are not clean report 0.
 */
#ifndef BTLBF_H
#define BTLBF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct btlbf_filter btlbf_filter; /* opaque; owns its HBM array (BloomFilter.hpp:381,398) */

enum { BTLBF_HOST = 0, BTLBF_DEVICE = 1 };
enum { BTLBF_BLOOM = 0, BTLBF_COUNTING8 = 1 };
enum {
	BTLBF_OK = 0,
	BTLBF_EINVAL = 1,  /* bad argument (e.g. bit count not a multiple of 8, BloomFilter.hpp:391-394) */
	BTLBF_ENOMEM = 2,
	BTLBF_EIO = 3,     /* file could not be opened/read/written (vendor/IOUtil.h:14-22) */
	BTLBF_EFORMAT = 4, /* bad magic line / missing [HeaderEnd] / missing key (BloomFilter.hpp:123-148) */
	BTLBF_EHIP = 5     /* HIP runtime error, or no GPU */
};
/* counting-filter update flavours (CountingBloomFilter.hpp:135-183) */
enum { BTLBF_INCREMENT_MIN = 0, BTLBF_INCREMENT_ALL = 1 };
/* BTLBF_ORDER_SERIAL applies k-mers one after another in buffer order on a single lane: the only
 * way incrementMin is reproducible (it is order-dependent). */
enum { BTLBF_ORDER_PARALLEL = 0, BTLBF_ORDER_SERIAL = 1 };

/* hipStreamPerThread (hip_runtime_api.h), for callers that do not include the HIP headers */
#define BTLBF_STREAM_PER_THREAD ((void*)2)

typedef struct btlbf_layout {
	const uint64_t* starts; /* n_seqs+1 offsets, or NULL */
	uint64_t n_seqs;        /* used with starts */
	uint32_t read_len;      /* used when starts == NULL; 0 = one sequence */
} btlbf_layout;

const char* btlbf_last_error(void);
int btlbf_device_count(void); /* number of visible GPUs, 0 if none (never fails) */

/* ---- lifetime ------------------------------------------------------------------------------
 * BTLBF_BLOOM:     `size` = number of bits, must be a multiple of 8
 *                  (BloomFilter(size_t,unsigned,unsigned), BloomFilter.hpp:65-76,389-399)
 * BTLBF_COUNTING8: `size` = bytes requested, rounded up to a multiple of 8; one uint8_t counter
 *                  per byte (CountingBloomFilter<uint8_t>(size_t,unsigned,unsigned,unsigned),
 *                  CountingBloomFilter.hpp:31-50).  `threshold` is ignored for BTLBF_BLOOM.
 * The array is allocated in the HBM of `device` and zeroed. */
int btlbf_create(btlbf_filter** out, int kind, uint64_t size, unsigned hash_num, unsigned kmer_size,
                 unsigned threshold, int device);
/* One hash-range shard of a larger filter: positions are computed modulo
 * `global_size` and this object stores [shard_index*global_size/shard_count, +global_size/shard_count).
 * Concatenating the shard bodies in index order is the single-filter body.
 * btlbf_insert_seqs on a shard sets the probes that fall inside its range and drops the others
 * (counting shards: BTLBF_INCREMENT_ALL in parallel order only); btlbf_contains_seqs answers "clean
 * window and every probe inside my range is set" (counting: "... is at least the threshold").  So W
 * shards that are each handed the same reads build the single filter's body, and the AND of their
 * answers is the single filter's contains() ("gather mode" of btl_bloomfilter_amd/sharded.py: moves
 * reads instead of probes between GPUs).  Both take the partitioned pipeline like an unsharded filter. */
int btlbf_create_shard(btlbf_filter** out, int kind, uint64_t global_size, unsigned shard_index,
                       unsigned shard_count, unsigned hash_num, unsigned kmer_size,
                       unsigned threshold, int device);
int btlbf_destroy(btlbf_filter* f); /* ~BloomFilter, BloomFilter.hpp:381 */

/* ---- BTLBloomFilter_v1 / BTLCountingBloomFilter_v1 files ------------------------------------
 * load:  BloomFilter(const string&) BloomFilter.hpp:101-116,118-166;
 *        CountingBloomFilter(const string&, unsigned) CountingBloomFilter.hpp:262-343
 * store: storeFilter BloomFilter.hpp:304-314 / CountingBloomFilter.hpp:331-342; the header bytes
 *        are those the reference writes (key order, tab indent, %#.17g doubles) */
int btlbf_load(btlbf_filter** out, int kind, const char* path, unsigned threshold, int device);
/* header text only (everything up to and including the "[HeaderEnd]" line) -> a zeroed filter of that geometry
 * with nEntry / Entry / dFPR taken over: what the reference's public loadHeader(std::istream&) leaves behind
 * (BloomFilter.hpp:118-166, CountingBloomFilter.hpp:84,282-343); the caller fills the body (btlbf_upload) */
int btlbf_create_from_header(btlbf_filter** out, int kind, const char* header, size_t len, unsigned threshold,
                             int device);
int btlbf_store(btlbf_filter* f, const char* path);
/* header text only (writeHeader BloomFilter.hpp:264-288, storeHeader CountingBloomFilter.hpp:344-368) */
int btlbf_header(const btlbf_filter* f, char* buf, size_t cap, size_t* len);
/* shard-aware store: shard 0 writes header+body at offset 0, shard s writes its body at
 * header_len + s*shard_bytes of the same file (SURVEY.md section 5 "Checkpoint / resume") */
int btlbf_store_shard(btlbf_filter* f, const char* path);

/* ---- attributes (getters of BloomFilter.hpp:325-327,369-379; CountingBloomFilter.hpp:78-82) -- */
int btlbf_kind(const btlbf_filter* f);
uint64_t btlbf_size(const btlbf_filter* f);        /* getFilterSize() bits / size() counters (global) */
uint64_t btlbf_size_bytes(const btlbf_filter* f);  /* sizeInBytes() (global) */
uint64_t btlbf_local_bytes(const btlbf_filter* f); /* bytes held by this object (== size_bytes unless a shard) */
unsigned btlbf_hash_num(const btlbf_filter* f);
unsigned btlbf_kmer_size(const btlbf_filter* f);
unsigned btlbf_threshold(const btlbf_filter* f);
uint64_t btlbf_get_n_entry(const btlbf_filter* f);
uint64_t btlbf_get_t_entry(const btlbf_filter* f);
void btlbf_set_n_entry(btlbf_filter* f, uint64_t v); /* setnEntry BloomFilter.hpp:373 */
void btlbf_set_t_entry(btlbf_filter* f, uint64_t v); /* settEntry BloomFilter.hpp:375 */
double btlbf_get_dfpr(const btlbf_filter* f);         /* m_dFPR: the header's dFPR field (BloomFilter.hpp:83-99,277) */
void btlbf_set_dfpr(btlbf_filter* f, double v);
void* btlbf_device_ptr(const btlbf_filter* f);       /* the HBM array (carries out a pending clear; NULL for NULL) */
int btlbf_device(const btlbf_filter* f);

/* raw array access; offset/nbytes in bytes of the local array.
   btlbf_clear is LAZY (and so is creation: a new filter is a cleared filter): it marks the array as "all zero"
   and records the point of the clear on `stream`.  If the next thing that happens is a partitioned
   btlbf_insert_seqs, its first batch builds every segment from zero in LDS and writes it out -- no memset
   sweep, no read of the old array -- and anything else that looks at the array (direct insert, queries,
   download, store, popcount, digest, compare, shard/rank/row entry points, btlbf_device_ptr) zeroes it first.
   Whichever stream carries the zeroing out first waits for the recorded point, so work queued on `stream`
   before the clear is never overtaken by it.  The one thing a lazy clear cannot serve is a raw pointer kept
   from an earlier btlbf_device_ptr call: once that function has been called on a filter, its clears are eager
   (hipMemsetAsync on `stream`). */
int btlbf_clear(btlbf_filter* f, void* stream);
int btlbf_upload(btlbf_filter* f, const void* host_src, uint64_t offset, uint64_t nbytes);
int btlbf_download(const btlbf_filter* f, void* host_dst, uint64_t offset, uint64_t nbytes);

/* How btlbf_insert_seqs applies a batch to a bit filter.  DIRECT: one atomicOr per probe (random HBM
 * access).  PARTITIONED: probe positions are radix-partitioned by filter segment and ORed into the
 * segment while it sits in LDS (streamed HBM access; needs scratch memory, pays one sweep of the
 * array per batch).  AUTO picks PARTITIONED when the batch is large relative to the filter.  The
 * filter bytes are identical either way (bit OR is order-free).  scratch_bytes caps the scratch
 * allocation (0 = up to 80 % of the free HBM).  Default: BTLBF_INSERT_AUTO, or the environment
 * variable BTLBF_INSERT_MODE=direct|partitioned. */
enum { BTLBF_INSERT_AUTO = 0, BTLBF_INSERT_DIRECT = 1, BTLBF_INSERT_PARTITIONED = 2 };
int btlbf_set_insert_mode(btlbf_filter* f, int mode, uint64_t scratch_bytes);
/* Same choice for btlbf_contains_seqs on a bit filter.  PARTITIONED tests the partitioned positions
 * against each segment in LDS and keeps the few positions found clear; it pays off when nearly
 * every k-mer hits (a batch with many misses is redone by the direct gather kernel).  AUTO samples
 * the batch first.  Results are identical in every mode.  Environment: BTLBF_QUERY_MODE. */
int btlbf_set_query_mode(btlbf_filter* f, int mode);
/* The partitioned paths keep their scratch allocation (up to 80 % of the HBM that was free at first
 * use, or the cap given to btlbf_set_insert_mode) cached in the filter; this returns it. */
int btlbf_release_scratch(btlbf_filter* f);

/* Per-kernel timing for measurement harnesses: when on, every kernel the sequence calls launch is
 * bracketed by HIP events on the launch stream.  btlbf_get_profile synchronises those events and
 * returns accumulated milliseconds and launch counts per slot (arrays of BTLBF_PROF_SLOTS). */
enum {
	BTLBF_PROF_INSERT_DIRECT = 0, /* seq_kernel<OP_BF_INSERT>: fused ntHash + atomicOr */
	BTLBF_PROF_QUERY_DIRECT = 1,  /* seq_kernel<OP_BF_CONTAINS>: fused ntHash + gather */
	BTLBF_PROF_INSERT_HASH = 2,   /* part_hash_kernel (pass A) */
	BTLBF_PROF_INSERT_SPLIT = 3,  /* part_split_kernel (pass B) */
	BTLBF_PROF_INSERT_APPLY = 4,  /* part_apply_kernel (pass C, OR in LDS) */
	BTLBF_PROF_QUERY_HASH = 5,
	BTLBF_PROF_QUERY_SPLIT = 6,
	BTLBF_PROF_QUERY_TEST = 7,    /* part_apply_kernel<QUERY> (pass C, test in LDS) */
	BTLBF_PROF_QUERY_RESOLVE = 8, /* failed-position set + resolve pass, or direct redo of a batch */
	BTLBF_PROF_OTHER = 9,
	BTLBF_PROF_SLOTS = 10
};
int btlbf_set_profiling(btlbf_filter* f, int on);
int btlbf_get_profile(btlbf_filter* f, double* ms, unsigned* calls, int reset);

/* Use spaced-seed hashing (stHashIterator, vendor/stHashIterator.hpp:23-33,53-57) for every
 * sequence-buffer call on this filter: `seeds` are n_seeds strings of length kmer_size, '1' =
 * care; hash_num must equal n_seeds*h2.  Without this call sequences are hashed like
 * ntHashIterator (vendor/ntHashIterator.hpp:38). */
int btlbf_set_spaced_seeds(btlbf_filter* f, const char* const* seeds, unsigned n_seeds, unsigned h2);

/* ---- the hot path: hash every clean window of a sequence buffer and probe the filter --------
 * insert:   insertSeq (BloomFilterUtil.h:9-17) = ntHashIterator + BloomFilter::insert
 *           (BloomFilter.hpp:185-194); counting filters: CountingBloomFilter::insert =
 *           incrementMin (CountingBloomFilter.hpp:198-204) or incrementAll (:165-183) per `op`.
 * contains: ntHashIterator + BloomFilter::contains (BloomFilter.hpp:252-262) /
 *           CountingBloomFilter::contains (CountingBloomFilter.hpp:190-196).
 * insert_and_check: BloomFilter::insertAndCheck (BloomFilter.hpp:200-214) -- bit of window p =
 *           all h bits were already set before this window's own writes.
 * hit_bits / valid_bits may be NULL.  counts (may be NULL) receives {clean windows, hits}; it
 * lives in `mem` space and is overwritten. */
int btlbf_insert_seqs(btlbf_filter* f, const char* seq, uint64_t len, const btlbf_layout* layout,
                      int op, int order, int mem, void* stream);
int btlbf_contains_seqs(btlbf_filter* f, const char* seq, uint64_t len, const btlbf_layout* layout,
                        uint64_t* hit_bits, uint64_t* valid_bits, uint64_t* counts, int mem,
                        void* stream);
int btlbf_insert_and_check_seqs(btlbf_filter* f, const char* seq, uint64_t len,
                                const btlbf_layout* layout, uint64_t* hit_bits, uint64_t* valid_bits,
                                uint64_t* counts, int mem, void* stream);
/* minCount per window (CountingBloomFilter.hpp:53-64): min_out[p], len bytes, 0 for unclean windows */
int btlbf_min_count_seqs(btlbf_filter* f, const char* seq, uint64_t len, const btlbf_layout* layout,
                         uint8_t* min_out, uint64_t* valid_bits, int mem, void* stream);

/* ---- precomputed hashes: n k-mers x hash_num uint64_t, row-major ------------------------------
 * insert(const uint64_t[]) BloomFilter.hpp:185; contains(const uint64_t[]) :252;
 * insertAndCheck(const uint64_t[]) :200; CountingBloomFilter insert/incrementAll/minCount/contains */
int btlbf_insert_hashes(btlbf_filter* f, const uint64_t* hashes, uint64_t n, int op, int order,
                        int mem, void* stream);
int btlbf_contains_hashes(btlbf_filter* f, const uint64_t* hashes, uint64_t n, uint8_t* out, int mem,
                          void* stream);
int btlbf_insert_and_check_hashes(btlbf_filter* f, const uint64_t* hashes, uint64_t n, uint8_t* out,
                                  int order, int mem, void* stream);
int btlbf_min_count_hashes(btlbf_filter* f, const uint64_t* hashes, uint64_t n, uint8_t* min_out,
                           int mem, void* stream);

/* ---- raw k-mers: KmerBloomFilter::insert(const char*) / contains(const char*) --------------------
 * (KmerBloomFilter.hpp:47-74 -> NTC64(kmerSeq, k) vendor/nthash.hpp:394-439,460-465 + NTE64 :537-542)
 * `kmers` = n k-mers of kmer_size bytes each, back to back.  The values are the ones the reference's
 * x86-64 build returns on this path, which differ from the iterator's (NTMC64) for k % 4 == 0 (the table
 * walk's shift by 64, :354-356,388-391,404-406) and for k-mers with U/u (read as A, :16-86, except in a
 * one-base remainder).  Bytes that are not ACGTU/acgtu wrap the reference's uint8_t table index to some
 * other 4-mer, as there; a k-mer whose 2-/3-base remainder then indexes beyond the reference's tables has
 * no defined value and is skipped (insert: nothing, contains: 0, valid: 0).
 * op / order as btlbf_insert_hashes (counting filters); out[i] = contains() of k-mer i. */
int btlbf_insert_kmers(btlbf_filter* f, const char* kmers, uint64_t n, int op, int order, int mem,
                       void* stream);
int btlbf_contains_kmers(btlbf_filter* f, const char* kmers, uint64_t n, uint8_t* out, int mem, void* stream);
/* the hash values alone: hashes[i*hash_num ..) and (optionally) valid[i] for k-mer i */
int btlbf_hash_kmers(unsigned kmer_size, unsigned hash_num, const char* kmers, uint64_t n, uint64_t* hashes,
                     uint8_t* valid, int mem, int device, void* stream);

/* ---- hash streams only (iterator parity) -------------------------------------------------------
 * ntHashIterator (vendor/ntHashIterator.hpp:38,93): hashes[p*h .. p*h+h) for window p (zeros when
 * not clean), valid_bits as above.  With seeds != NULL: stHashIterator(seq, seeds, n_seeds, h2, k)
 * (vendor/stHashIterator.hpp:53,94-103), h = n_seeds*h2 values per window and strand_bits[p] bit j =
 * strandArray()[j] (h <= 64 in that case).  Buffers live in `mem` space. */
int btlbf_hash_seqs(unsigned kmer_size, unsigned hash_num, const char* const* seeds, unsigned n_seeds,
                    unsigned h2, const char* seq, uint64_t len, const btlbf_layout* layout,
                    uint64_t* hashes, uint64_t* valid_bits, uint64_t* strand_bits, int mem,
                    int device, void* stream);

/* ---- statistics --------------------------------------------------------------------------------
 * getPop BloomFilter.hpp:316-323 (set bits); counting: popCount (non-zero counters) :217-228 and
 * filtered_popcount (counters >= threshold) :231-242.  Local array only for shards. */
int btlbf_popcount(btlbf_filter* f, uint64_t* out);
int btlbf_filtered_popcount(btlbf_filter* f, uint64_t* out);

/* Position-dependent digest of the local array, computed in HBM (one streaming read, like btlbf_popcount) -- what a
 * caller of the reference does with sha256sum on a stored .bf body, for arrays too large to keep two of, or to
 * download.  With w_i the i-th little-endian 64-bit word of the WHOLE filter body (zero-padded at the end) and
 * m_i = mix64(i + 1) | 1 (mix64 = the splitmix64 finaliser: z = (z ^ z >> 30) * 0xBF58476D1CE4E5B9;
 * z = (z ^ z >> 27) * 0x94D049BB133111EB; z ^ z >> 31):
 *   out2[0] = sum over the non-zero words of w_i * m_i (mod 2^64),  out2[1] = xor over them of mix64(w_i ^ m_i).
 * A shard reports its own words with their index in the whole filter, so shard digests combine by + and ^ to the
 * digest of the unsharded filter.  Synchronises the device. */
int btlbf_digest(btlbf_filter* f, uint64_t* out2);

/* Position-wise comparison of two filters of the same kind, size (and shard range) on one device, without
 * leaving HBM -- what a caller of the reference does with two filter bodies and memcmp.  Bit filters:
 * out3 = {bits that differ, bits set only in a, bits set only in b}; counting filters: {counters that
 * differ, counters with a > b, counters with a < b}.  Synchronises the device. */
int btlbf_compare(btlbf_filter* a, btlbf_filter* b, uint64_t* out3);

/* ---- rank structure over a bit filter (miBF stage 2) ------------------------------------
 * The reference's multi-index Bloom filter maps a set bit to an index of its ID array with
 * rank(pos) = set bits before pos -- getRankPos(hash) = m_rankSupport(hash % m_bv.size()), MIBloomFilter.hpp:527,
 * and the same expression in every insert / query loop (:324,391,443,461,488,509,522) -- over
 * sdsl::bit_vector_il<512> + sdsl::rank_support_il<1> (MIBloomFilter.hpp:44,133,144,801-803; sdsl-lite is an
 * un-vendored dependency of the reference).  rank_create builds that interleaved vector in HBM from a bit
 * filter (e.g. the spaced-seed filter that is miBF stage 1, MIBFConstructSupport.hpp:75-87): record b of 9
 * uint64_t = { set bits before bit 512*b, the 8 data words of block b }.  rank_query: values are positions,
 * or hash values to be reduced modulo the filter size (values_are_hashes != 0); rank_out[i] = rank,
 * bit_out[i] = the bit itself (either may be NULL).  rank_ones = set bits in total (= entries of the ID
 * array, MIBloomFilter.hpp:573-577); rank_download copies the rank_words() interleaved words to the host. */
typedef struct btlbf_rank btlbf_rank;
int btlbf_rank_create(btlbf_rank** out, btlbf_filter* f);
void btlbf_rank_destroy(btlbf_rank* r);
uint64_t btlbf_rank_ones(const btlbf_rank* r);
uint64_t btlbf_rank_words(const btlbf_rank* r);
int btlbf_rank_download(const btlbf_rank* r, uint64_t* host_dst);
int btlbf_rank_query(const btlbf_rank* r, const uint64_t* values, uint64_t n, int values_are_hashes,
                     uint64_t* rank_out, uint8_t* bit_out, int mem, void* stream);

/* ---- multi-GPU hash-range sharding ------------------------------------------------
 * The M-bit filter is cut into n_shards contiguous bit ranges; shard g (btlbf_create_shard) holds
 * positions [g*M/n, (g+1)*M/n).  Routing is by POSITION, so the concatenated shard bodies are the
 * single-filter body bit for bit.
 * positions_seqs: hash a device-resident sequence buffer with f's parameters and, instead of probing,
 *   append every probe of every clean window to the bucket of the shard that owns its position.
 *   buckets = n_shards regions of bucket_cap uint64_t in one device array; an entry is the position
 *   LOCAL to the owning shard.  tags (same shape, may be NULL) receives the probe id p*hash_num + i
 *   (p = window offset) so that routed answers can be matched up again.  bucket_counts[n_shards]
 *   (device) is zeroed by the call and receives the fill; a count above bucket_cap means entries
 *   were dropped and the caller must retry with a larger capacity.  valid_bits (device, may be
 *   NULL) is the usual per-window bitmap.
 * insert_positions / test_positions act on local positions of this shard (test: one byte 0/1 each).
 * and_answers: origin side of a sharded query -- answers[i] belongs to probe tags[i]; a 0 answer
 *   clears the bit of window tags[i]/hash_num in hit_bits (which starts as a copy of valid_bits). */
int btlbf_positions_seqs(btlbf_filter* f, const char* seq, uint64_t len, const btlbf_layout* layout,
                         unsigned n_shards, uint64_t* buckets, uint64_t* tags, uint64_t bucket_cap,
                         uint64_t* bucket_counts, uint64_t* valid_bits, void* stream);
int btlbf_insert_positions(btlbf_filter* f, const uint64_t* local_pos, uint64_t n, void* stream);
int btlbf_test_positions(btlbf_filter* f, const uint64_t* local_pos, uint64_t n, uint8_t* out,
                         void* stream);
int btlbf_and_answers(const uint64_t* tags, const uint8_t* answers, uint64_t n, unsigned hash_num,
                      uint64_t* hit_bits, int device, void* stream);
/* ---- multi-GPU on the partitioned pipeline (large batches) -----------------------
 * Needs a filter whose GLOBAL size and shard count are powers of two (at least 2^29 bits / 2^26 counters).
 * An entry is the 32-bit offset of a position inside its level-0 bin and an origin stages at most 1024 bins,
 * so one routing pass covers a WINDOW of at most 2^42 positions: a larger filter (BASELINE config 4: 2^43
 * bits on 8 GPUs) has several windows, each owned by n_shards / n_windows consecutive shards, and the same
 * reads are routed once per window (btlbf_route_windows; position semantics BloomFilter.hpp:190).  A window
 * is cut into B level-0 bins (512, or 1024 when a window has more than 2^41 positions); the s-th shard of
 * a window owns its bins [s*B/spw, (s+1)*B/spw).  All pointers are device pointers.
 *  route_windows: how many windows this geometry has, and how many shards own each.
 *  route_plan : byte sizes of ONE origin->owner block for a buffer of `plan_len` bytes.  Every rank
 *               must plan with the same plan_len (e.g. the maximum over ranks) so that blocks have one
 *               size and the exchange is a fixed-size all-to-all.
 *  route_seqs : origin.  Hash the buffer and partition every probe position inside `window` into the window's
 *               bins: send_ent / send_cnt receive shards_per_window consecutive blocks (block i goes to shard
 *               window*shards_per_window + i).
 *               query != 0 additionally writes valid_bits and initialises hit_bits = valid_bits;
 *               counts[0] (optional) += clean windows.  Entries that cannot be staged are appended to
 *               spill_list as global positions.  counts and spill_count ACCUMULATE over calls (zero
 *               them once per pass): no host round trip per batch is needed, so the exchange of one
 *               batch can overlap the hashing of the next.
 *  apply_routed: owner.  recv_ent / recv_cnt hold n_blocks blocks (one per origin, any order); they are
 *               split down to segments and ORed into (query == 0) or tested against (query != 0) this
 *               shard in LDS.  Positions found clear are appended to fail_list as GLOBAL positions
 *               (fail_count must be zeroed by the caller).
 *  route_geometry / apply_routed_bins: the same, group by group.  out4 = {level-0 bins per shard B/n,
 *               regions per bin, chunks (128 bytes) per region, bins per group}: a block is laid out
 *               [bin][region][chunk], so the bins of one group are a contiguous slice of every block
 *               (entries: bins*regions*chunks*128 bytes, counts: bins*regions*4 bytes).  An owner may
 *               receive and apply its bins a group at a time -- recv_ent / recv_cnt then hold, per
 *               origin block, only bins [first_bin, first_bin + n_bins) (whole groups) -- so that the
 *               receive buffers are a fraction of a block set and the exchange of one group overlaps
 *               the apply of the previous one.  n_blocks = 0 in route_geometry means n_shards.
 *  apply_spill : owner.  Insert / test explicit global positions (the gathered spill lists); positions
 *               of other shards are ignored.
 *  resolve_seqs: origin.  Clear the hit bit of every window that owns one of the (gathered) failed
 *               global positions. */
int btlbf_route_plan(btlbf_filter* f, uint64_t plan_len, const btlbf_layout* layout, unsigned n_shards,
                     uint64_t* ent_bytes_per_shard, uint64_t* cnt_bytes_per_shard);
int btlbf_route_windows(btlbf_filter* f, unsigned n_shards, unsigned* n_windows, unsigned* shards_per_window);
int btlbf_route_seqs(btlbf_filter* f, const char* seq, uint64_t len, const btlbf_layout* layout,
                     uint64_t plan_len, unsigned n_shards, unsigned window, int query, void* send_ent,
                     void* send_cnt, uint64_t* hit_bits, uint64_t* valid_bits, uint64_t* counts,
                     uint64_t* spill_list, uint64_t spill_cap, uint64_t* spill_count, void* stream);
int btlbf_apply_routed(btlbf_filter* f, const void* recv_ent, const void* recv_cnt, unsigned n_blocks,
                       uint64_t plan_len, const btlbf_layout* layout, unsigned n_shards, int query,
                       uint64_t* fail_list, uint64_t fail_cap, uint64_t* fail_count, void* stream);
int btlbf_route_geometry(btlbf_filter* f, uint64_t plan_len, const btlbf_layout* layout, unsigned n_shards,
                         unsigned n_blocks, uint32_t* out4);
/* Bytes of partition scratch the OWNER side (btlbf_apply_routed / btlbf_apply_routed_bins) allocates inside the filter
 * for batches planned with `plan_len`, `n_blocks` origin blocks (0: n_shards): lets the caller size its batches from
 * the free HBM exactly instead of by rule of thumb (sharded.py plans BASELINE config 4 -- 1 TiB on 8 GPUs, SURVEY 8e --
 * with four batches per pass instead of five that way).  No device work. */
int btlbf_owner_scratch_bytes(btlbf_filter* f, uint64_t plan_len, const btlbf_layout* layout, unsigned n_shards,
                              unsigned n_blocks, uint64_t* bytes);
int btlbf_apply_routed_bins(btlbf_filter* f, const void* recv_ent, const void* recv_cnt, unsigned n_blocks,
                            unsigned first_bin, unsigned n_bins, uint64_t plan_len, const btlbf_layout* layout,
                            unsigned n_shards, int query, uint64_t* fail_list, uint64_t fail_cap,
                            uint64_t* fail_count, void* stream);
int btlbf_apply_spill(btlbf_filter* f, const uint64_t* global_pos, uint64_t n, int query, uint64_t* fail_list,
                      uint64_t fail_cap, uint64_t* fail_count, void* stream);
int btlbf_resolve_seqs(btlbf_filter* f, const char* seq, uint64_t len, const btlbf_layout* layout,
                       const uint64_t* fail_list, uint64_t n_fail, uint64_t* hit_bits, void* stream);

/* Planning only, no device needed: the READ GRID pass A of the partitioned pipeline uses for fixed-length reads of
   read_len bases with `level0_bins` level-0 bins: out4 = { reads per tile, groups of 8
   window starts per read, bytes a read occupies in the LDS tile, bytes of LDS the tile image takes }; all zero
   when plain tiles are used instead (ragged input, reads too short / too long, no gain). */
int btlbf_plan_read_grid(unsigned kmer_size, unsigned hash_num, unsigned read_len, unsigned level0_bins,
                         uint32_t* out4);

/* number of set bits in a device buffer of nbytes (multiple of 8, e.g. a hit bitmap); synchronises the stream */
int btlbf_popcount_bits(const void* dev_buf, uint64_t nbytes, uint64_t* out, int device, void* stream);

/* ---- support ------------------------------------------------------------------------------------
 * synthetic reads of SURVEY.md 8d written to device memory: n reads x read_len bytes */
int btlbf_synth_reads(char* dev_out, uint64_t seed, uint64_t first_read, uint64_t n_reads,
                      unsigned read_len, int device, void* stream);
/* bare random-access ceilings on the filter's own array (SURVEY.md 8d "measured ceiling"):
 * kind 0 = independent 4-byte loads, 1 = 4-byte atomicOr (sets bits: clear the filter afterwards),
 * at uniformly random 64-byte-aligned offsets; about n_access accesses, the exact number is
 * returned in *n_done; *seconds = kernel time from HIP events */
int btlbf_microbench(btlbf_filter* f, int kind, uint64_t n_access, uint64_t* n_done, double* seconds);

/* per-sequence totals of the per-window bitmaps that btlbf_contains_seqs / btlbf_insert_and_check_seqs
 * produce: hits_out[s] = windows of sequence s whose bit is set in hit_bits, valid_out[s] (optional) =
 * its clean windows (valid_bits == NULL: all of its len-k+1 windows).  This is what read classifiers
 * built on the reference do per read with an ntHashIterator loop and a counter.  Device pointers
 * (mem == BTLBF_DEVICE) or host buffers; layout as for the call that made the bitmaps. */
int btlbf_count_per_seq(const uint64_t* hit_bits, const uint64_t* valid_bits, uint64_t len,
                        const btlbf_layout* layout, unsigned kmer_size, uint32_t* hits_out, uint32_t* valid_out,
                        int mem, int device, void* stream);

/* ---- FASTA / FASTQ ingestion -------------------------------------------------
 * Replaces the reference's toy loaders: Tests/AdHoc/ParallelFilter.cpp:104-122 (loadBf: header line +
 * one sequence line, an ntHashIterator per line) and swig/writeBloom_rolling.cpp:18-59
 * (contigsToBloom: the lines of a FASTA record are concatenated, then insertSeq).
 * Input may be gzip-compressed.  Format by the first byte: '>' FASTA, '@' FASTQ (4-line records),
 * anything else = one sequence per line. */
enum btlbf_fastx_flags {
	BTLBF_FASTX_RECORDS = 0,  /* the wrapped lines of a FASTA record form ONE sequence (contigsToBloom) */
	BTLBF_FASTX_LINES = 1,    /* every sequence line is its own sequence (loadBf) */
	BTLBF_FASTX_PAGEABLE = 2  /* do not pin the host buffers */
};
typedef struct btlbf_fastx btlbf_fastx;

/* streaming parser: batches of at most batch_bytes sequence bytes (0 = 256 MiB) in the ragged layout
 * of btlbf_layout; a sequence longer than a batch is cut with a k-1 base overlap so that every
 * window appears in exactly one batch */
int btlbf_fastx_open(btlbf_fastx** r, const char* path, uint32_t flags, uint32_t k, uint64_t batch_bytes);
/* *bases (n_bases bytes) and *starts (n_seqs+1 offsets) stay valid until the call after the next
 * one; n_seqs == 0 at end of input */
int btlbf_fastx_next(btlbf_fastx* r, const char** bases, uint64_t* n_bases, const uint64_t** starts,
                     uint64_t* n_seqs);
/* one of several readers over the same UNCOMPRESSED file (parallel parsing): delivers the records that
 * START in byte range [begin, end); the ranges of all readers must tile the file.  fmt: 1 = FASTA,
 * 2 = FASTQ (4-line), 3 = one sequence per line.  btlbf_insert_fastx / btlbf_contains_fastx do this
 * internally (BTLBF_FASTX_THREADS, default min(8, cores)). */
int btlbf_fastx_open_range(btlbf_fastx** r, const char* path, uint32_t flags, uint32_t k, uint64_t batch_bytes,
                           int fmt, uint64_t begin, uint64_t end);
uint64_t btlbf_fastx_records(const btlbf_fastx* r); /* records seen so far */
void btlbf_fastx_close(btlbf_fastx* r);

typedef struct btlbf_fastx_stats {
	uint64_t n_records;  /* FASTA / FASTQ records (lines, for plain input) */
	uint64_t n_bases;    /* sequence bytes sent to the GPU (overlaps of cut sequences included) */
	uint64_t n_windows;  /* contains: clean k-mer windows tested */
	uint64_t n_hits;     /* contains: windows found in the filter */
	uint64_t n_batches;
	double seconds_parse; /* host time inside the parser */
	double seconds_total; /* wall time of the call */
} btlbf_fastx_stats;

/* whole file -> filter: pinned double buffers, copies on a copy stream, kernels on a compute stream;
 * the host parses batch i+1 while batch i is copied and hashed.  stats may be NULL. */
int btlbf_insert_fastx(btlbf_filter* f, const char* path, uint32_t flags, uint64_t batch_bytes,
                       btlbf_fastx_stats* stats);
int btlbf_contains_fastx(btlbf_filter* f, const char* path, uint32_t flags, uint64_t batch_bytes,
                         btlbf_fastx_stats* stats);

#ifdef __cplusplus
}
#endif
#endif /* BTLBF_H */

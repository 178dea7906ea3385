/* oracle/btl_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's k-mer Bloom filter hot path:
 *   ntHash arithmetic          /root/reference/vendor/nthash.hpp
 *   ntHashIterator             /root/reference/vendor/ntHashIterator.hpp
 *   stHashIterator             /root/reference/vendor/stHashIterator.hpp
 *   BloomFilter                /root/reference/BloomFilter.hpp
 *   CountingBloomFilter<u8>    /root/reference/CountingBloomFilter.hpp
 *   KmerBloomFilter            /root/reference/KmerBloomFilter.hpp
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py checks every function here against
 * oracle/_ref/libbtlref.so (the genuine reference headers compiled by oracle/Makefile) and
 * tests/test_oracle_golden.py checks it against the tests/golden/ fixtures (vectors emitted by that same
 * reference build via tests/golden/make_golden.py), including the reference's own unit-test
 * cases (Tests/Unit/BloomFilterTests.cpp:69-95, CountingBloomFilterTests.cpp:70-122).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 * The product (btl_bloomfilter_amd/, include/) never links or loads it.
 */
#ifndef BTL_ORACLE_H
#define BTL_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BO_NPOS ((size_t)-1)

/* ---- ntHash arithmetic (nthash.hpp) ---- */
uint64_t bo_seed(unsigned char c);                 /* seedTab, nthash.hpp:195-228 */
uint64_t bo_srol(uint64_t x);                      /* rol1+swapbits033, :350-352,377-380 */
uint64_t bo_sror(uint64_t x);                      /* ror1+swapbits3263, :361-363,383-386 */
uint64_t bo_srol_n(uint64_t x, unsigned s);        /* srol^s == msTab31l|msTab33r, :230-347 */
/* closed form over one window; returns 0 and *loc_n = index of LAST bad char if not clean
 * (NTMC64 base, :667-692) */
int bo_base_hash(const char* kmer, unsigned k, uint64_t* fh, uint64_t* rh, unsigned* loc_n);
void bo_roll(uint64_t* fh, uint64_t* rh, unsigned k, unsigned char out, unsigned char in); /* :442-457 */
void bo_multi(uint64_t b, unsigned k, unsigned h, uint64_t* hv); /* :585-589 */
uint64_t bo_extra(uint64_t b, unsigned k, unsigned i);            /* NTE64 :537-542 */
/* raw k-mer path of KmerBloomFilter: NTF64/NTR64(kmerSeq,k) as the x86-64 reference build behaves
 * (:394-439; U read as A, the shift-by-64 step for k % 4 == 0, uint8_t index wrap); returns 0 where the
 * reference reads beyond its 2-/3-mer tables */
int bo_kmer_base_hash(const char* kmer, unsigned k, uint64_t* fh, uint64_t* rh);
/* NTC64(kmerSeq,k) + NTE64 for n k-mers of k bytes each, back to back (:460-465,537-542) */
void bo_kmer_hashes(const char* kmers, size_t n, unsigned k, unsigned h, uint64_t* hash_out, uint8_t* valid_out);

/* ---- iterators: emit (pos, hashes) for every clean window, in order ---- */
size_t bo_nthash_seq(const char* seq, size_t len, unsigned h, unsigned k,
                     uint64_t* pos_out, uint64_t* hash_out, size_t cap);
/* seeds: nseeds strings of length k, '1' = care (stHashIterator.hpp:23-33) */
size_t bo_sthash_seq(const char* seq, size_t len, const char* const* seeds, unsigned nseeds,
                     unsigned h2, unsigned k, uint64_t* pos_out, uint64_t* hash_out,
                     uint8_t* strand_out, size_t cap);

/* ---- bit filter on a caller-owned byte array (BloomFilter.hpp) ---- */
void bo_bf_insert(uint8_t* filt, uint64_t size_bits, unsigned h, const uint64_t* hashes, size_t n);
void bo_bf_contains(const uint8_t* filt, uint64_t size_bits, unsigned h, const uint64_t* hashes,
                    size_t n, uint8_t* out);
void bo_bf_insert_and_check(uint8_t* filt, uint64_t size_bits, unsigned h, const uint64_t* hashes,
                            size_t n, uint8_t* out);
uint64_t bo_bf_popcount(const uint8_t* filt, uint64_t size_bits);
/* insertSeq (BloomFilterUtil.h:10) */
void bo_bf_insert_seq(uint8_t* filt, uint64_t size_bits, unsigned h, unsigned k,
                      const char* seq, size_t len);
/* dense per-window result: hit[p] = 1 iff window p clean and contained; valid[p] likewise */
void bo_bf_contains_seq_dense(const uint8_t* filt, uint64_t size_bits, unsigned h, unsigned k,
                              const char* seq, size_t len, uint8_t* hit, uint8_t* valid);

/* ---- counting filter, uint8_t counters (CountingBloomFilter.hpp) ---- */
uint64_t bo_cbf_round_bytes(uint64_t bytes);       /* ctor rounding :40-49 */
uint8_t bo_cbf_min(const uint8_t* c, uint64_t size, unsigned h, const uint64_t* hv); /* :53-64 */
void bo_cbf_increment_min(uint8_t* c, uint64_t size, unsigned h, const uint64_t* hashes, size_t n); /* :135-162 */
void bo_cbf_increment_all(uint8_t* c, uint64_t size, unsigned h, const uint64_t* hashes, size_t n); /* :165-183 */
void bo_cbf_insert_and_check(uint8_t* c, uint64_t size, unsigned h, unsigned thr,
                             const uint64_t* hashes, size_t n, uint8_t* out);          /* :206-214 */
void bo_cbf_query(const uint8_t* c, uint64_t size, unsigned h, unsigned thr,
                  const uint64_t* hashes, size_t n, uint8_t* min_out, uint8_t* contains_out);
uint64_t bo_cbf_popcount(const uint8_t* c, uint64_t size);                      /* :217-228 */
uint64_t bo_cbf_filtered_popcount(const uint8_t* c, uint64_t size, unsigned thr); /* :231-242 */

/* ---- .bf headers, exact bytes (BloomFilter.hpp:264-288, CountingBloomFilter.hpp:344-368) ---- */
int bo_bf_header(char* buf, size_t cap, uint64_t size_bits, unsigned h, unsigned k,
                 double dfpr, uint64_t n_entry, uint64_t t_entry);
int bo_cbf_header(char* buf, size_t cap, uint64_t size, uint64_t size_bytes, unsigned h, unsigned k,
                  unsigned bits_per_counter);

/* ---- synthetic reads (repo-defined) and timed CPU port ---- */
void bo_synth_reads(uint64_t seed, uint64_t first, uint64_t n, unsigned read_len, char* out);
int bo_bench_bf(uint64_t n_reads, unsigned read_len, unsigned k, unsigned h, uint64_t bits,
                uint64_t seed_ins, uint64_t seed_qry, int threads, int prefault, double* out);

#ifdef __cplusplus
}
#endif
#endif

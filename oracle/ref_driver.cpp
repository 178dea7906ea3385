// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A thin extern "C" driver around the GENUINE reference headers, compiled from the
// sources where they lie under /root/reference (never copied into this repo) by
// oracle/Makefile into oracle/_ref/libbtlref.so.  It exists for two purposes only:
//   1. pin oracle/btl_oracle.c (the CPU restatement) and generate tests/golden/*
//      (tests/golden/make_golden.py) from the real reference implementation;
//   2. be the "reference"-kind CPU baseline timed by bench.py's cpu_baseline leg.
// Nothing under btl_bloomfilter_amd/ or include/ may link, load or call this.
//
// Every function below is a plain loop over the reference's public API:
//   ntHashIterator  (vendor/ntHashIterator.hpp:38-121)
//   stHashIterator  (vendor/stHashIterator.hpp:23-104)
//   BloomFilter     (BloomFilter.hpp:171-262, 304-323)
//   KmerBloomFilter (KmerBloomFilter.hpp:47-74)
//   insertSeq       (BloomFilterUtil.h:10)
//   CountingBloomFilter<uint8_t> (CountingBloomFilter.hpp:53-214, 217-242, 268-343)
#include "BloomFilterUtil.h" // pulls KmerBloomFilter.hpp, BloomFilter.hpp, ntHashIterator.hpp
#include "CountingBloomFilter.hpp"
#include "vendor/stHashIterator.hpp"

#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// exposes the protected byte array of the reference filter (BloomFilter.hpp:436)
struct RefBF : public KmerBloomFilter
{
	RefBF(size_t bits, unsigned h, unsigned k)
	  : KmerBloomFilter(bits, h, k)
	{}
	explicit RefBF(const std::string& path)
	  : KmerBloomFilter(path)
	{}
	uint8_t* bytes() { return m_filter; }
};

typedef CountingBloomFilter<uint8_t> RefCBF;

// counter-based synthetic read generator (our definition, not the
// reference's -- the reference has no generator).
inline uint64_t
mix64(uint64_t z)
{
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

inline void
synth_read(uint64_t seed, uint64_t r, unsigned read_len, char* out)
{
	const unsigned wpr = (read_len + 31) / 32;
	for (unsigned j = 0; j < read_len; ++j) {
		uint64_t n = r * wpr + j / 32;
		uint64_t w = mix64(seed + (n + 1) * 0x9E3779B97F4A7C15ULL);
		out[j] = "ACGT"[(w >> (2 * (j % 32))) & 3];
	}
}

double
now_s()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch())
	    .count();
}

} // namespace

extern "C"
{

	// ---- hash iterators -------------------------------------------------------------------
	size_t ref_nthash_seq(
	    const char* seq,
	    size_t len,
	    unsigned h,
	    unsigned k,
	    uint64_t* pos_out,
	    uint64_t* hash_out,
	    size_t cap)
	{
		std::string s(seq, len);
		size_t n = 0;
		ntHashIterator it(s, h, k);
		while (it != ntHashIterator::end()) {
			if (n < cap) {
				pos_out[n] = it.pos();
				for (unsigned i = 0; i < h; ++i)
					hash_out[n * h + i] = (*it)[i];
			}
			++n;
			++it;
		}
		return n;
	}

	size_t ref_sthash_seq(
	    const char* seq,
	    size_t len,
	    const char* const* seeds,
	    unsigned nseeds,
	    unsigned h2,
	    unsigned k,
	    uint64_t* pos_out,
	    uint64_t* hash_out,
	    uint8_t* strand_out,
	    size_t cap)
	{
		std::string s(seq, len);
		std::vector<std::string> ss;
		for (unsigned i = 0; i < nseeds; ++i)
			ss.push_back(seeds[i]);
		std::vector<std::vector<unsigned> > parsed = stHashIterator::parseSeed(ss);
		const unsigned m = nseeds * h2;
		size_t n = 0;
		stHashIterator it(s, parsed, nseeds, h2, k);
		while (it != stHashIterator::end()) {
			if (n < cap) {
				pos_out[n] = it.pos();
				for (unsigned i = 0; i < m; ++i) {
					hash_out[n * m + i] = (*it)[i];
					strand_out[n * m + i] = it.strandArray()[i] ? 1 : 0;
				}
			}
			++n;
			++it;
		}
		return n;
	}

	// raw k-mer canonical hash (nthash.hpp:460-465) + extra hashes (nthash.hpp:537-542)
	void ref_kmer_hashes(const char* kmer, unsigned k, unsigned h, uint64_t* out)
	{
		uint64_t b = NTC64(kmer, k);
		out[0] = b;
		for (unsigned i = 1; i < h; ++i)
			out[i] = NTE64(b, k, i);
	}

	// ---- bit filter -----------------------------------------------------------------------
	void* ref_bf_new(size_t bits, unsigned h, unsigned k) { return new RefBF(bits, h, k); }
	void* ref_bf_load(const char* path) { return new RefBF(std::string(path)); }
	void ref_bf_free(void* p) { delete static_cast<RefBF*>(p); }
	uint8_t* ref_bf_bytes(void* p) { return static_cast<RefBF*>(p)->bytes(); }
	uint64_t ref_bf_size_bits(void* p) { return static_cast<RefBF*>(p)->getFilterSize(); }
	uint64_t ref_bf_size_bytes(void* p) { return static_cast<RefBF*>(p)->sizeInBytes(); }
	unsigned ref_bf_hash_num(void* p) { return static_cast<RefBF*>(p)->getHashNum(); }
	unsigned ref_bf_kmer_size(void* p) { return static_cast<RefBF*>(p)->getKmerSize(); }
	uint64_t ref_bf_pop(void* p) { return static_cast<RefBF*>(p)->getPop(); }
	double ref_bf_fpr(void* p) { return static_cast<RefBF*>(p)->getFPR(); }
	void ref_bf_set_entries(void* p, uint64_t n, uint64_t t)
	{
		static_cast<RefBF*>(p)->setnEntry(n);
		static_cast<RefBF*>(p)->settEntry(t);
	}
	void ref_bf_store(void* p, const char* path)
	{
		static_cast<RefBF*>(p)->storeFilter(std::string(path));
	}

	void ref_bf_insert(void* p, const uint64_t* hashes, size_t n)
	{
		RefBF* f = static_cast<RefBF*>(p);
		const unsigned h = f->getHashNum();
		for (size_t i = 0; i < n; ++i)
			f->insert(hashes + i * h);
	}
	void ref_bf_contains(void* p, const uint64_t* hashes, size_t n, uint8_t* out)
	{
		RefBF* f = static_cast<RefBF*>(p);
		const unsigned h = f->getHashNum();
		for (size_t i = 0; i < n; ++i)
			out[i] = f->contains(hashes + i * h) ? 1 : 0;
	}
	void ref_bf_insert_and_check(void* p, const uint64_t* hashes, size_t n, uint8_t* out)
	{
		RefBF* f = static_cast<RefBF*>(p);
		const unsigned h = f->getHashNum();
		for (size_t i = 0; i < n; ++i)
			out[i] = f->insertAndCheck(hashes + i * h) ? 1 : 0;
	}
	void ref_bf_insert_seq(void* p, const char* seq, size_t len)
	{
		RefBF* f = static_cast<RefBF*>(p);
		insertSeq(*f, std::string(seq, len), f->getHashNum(), f->getKmerSize());
	}
	// per valid k-mer: position and contains() result
	size_t
	ref_bf_contains_seq(void* p, const char* seq, size_t len, uint64_t* pos_out, uint8_t* res, size_t cap)
	{
		RefBF* f = static_cast<RefBF*>(p);
		std::string s(seq, len);
		size_t n = 0;
		ntHashIterator it(s, f->getHashNum(), f->getKmerSize());
		while (it != ntHashIterator::end()) {
			if (n < cap) {
				pos_out[n] = it.pos();
				res[n] = f->contains(*it) ? 1 : 0;
			}
			++n;
			++it;
		}
		return n;
	}
	void ref_kbf_insert_kmer(void* p, const char* kmer) { static_cast<RefBF*>(p)->insert(kmer); }
	int ref_kbf_contains_kmer(void* p, const char* kmer)
	{
		return static_cast<RefBF*>(p)->contains(kmer) ? 1 : 0;
	}

	// ---- counting filter (uint8_t) --------------------------------------------------------
	void* ref_cbf_new(size_t bytes, unsigned h, unsigned k, unsigned thr)
	{
		return new RefCBF(bytes, h, k, thr);
	}
	void* ref_cbf_load(const char* path, unsigned thr) { return new RefCBF(std::string(path), thr); }
	void ref_cbf_free(void* p) { delete static_cast<RefCBF*>(p); }
	uint64_t ref_cbf_size(void* p) { return static_cast<RefCBF*>(p)->size(); }
	uint64_t ref_cbf_size_bytes(void* p) { return static_cast<RefCBF*>(p)->sizeInBytes(); }
	unsigned ref_cbf_hash_num(void* p) { return static_cast<RefCBF*>(p)->getHashNum(); }
	unsigned ref_cbf_kmer_size(void* p) { return static_cast<RefCBF*>(p)->getKmerSize(); }
	uint64_t ref_cbf_popcount(void* p) { return static_cast<RefCBF*>(p)->popCount(); }
	uint64_t ref_cbf_filtered_popcount(void* p)
	{
		return static_cast<RefCBF*>(p)->filtered_popcount();
	}
	void ref_cbf_read(void* p, uint8_t* out)
	{
		RefCBF* f = static_cast<RefCBF*>(p);
		for (size_t i = 0; i < f->size(); ++i)
			out[i] = (*f)[i];
	}
	void ref_cbf_store(void* p, const char* path)
	{
		static_cast<RefCBF*>(p)->storeFilter(std::string(path));
	}
	// op: 0 = insert (incrementMin), 1 = incrementAll, 2 = insertAndCheck (out = found)
	void ref_cbf_update(void* p, const uint64_t* hashes, size_t n, int op, uint8_t* out)
	{
		RefCBF* f = static_cast<RefCBF*>(p);
		const unsigned h = f->getHashNum();
		for (size_t i = 0; i < n; ++i) {
			const uint64_t* hv = hashes + i * h;
			if (op == 0)
				f->insert(hv);
			else if (op == 1)
				f->incrementAll(hv);
			else
				out[i] = f->insertAndCheck(hv) ? 1 : 0;
		}
	}
	void
	ref_cbf_query(void* p, const uint64_t* hashes, size_t n, uint8_t* min_out, uint8_t* contains_out)
	{
		RefCBF* f = static_cast<RefCBF*>(p);
		const unsigned h = f->getHashNum();
		for (size_t i = 0; i < n; ++i) {
			const uint64_t* hv = hashes + i * h;
			if (min_out)
				min_out[i] = f->minCount(hv);
			if (contains_out)
				contains_out[i] = f->contains(hv) ? 1 : 0;
		}
	}

	// ---- synthetic reads + timed CPU baseline ---------------------------------------------
	void ref_synth_reads(uint64_t seed, uint64_t first, uint64_t n, unsigned read_len, char* out)
	{
		for (uint64_t r = 0; r < n; ++r)
			synth_read(seed, first + r, read_len, out + r * read_len);
	}


	// insertSeq over synthetic reads [first, first+n), OpenMP over reads (bit OR is commutative,
	// so the bytes do not depend on the thread count)
	void ref_bf_insert_synth(void* p, uint64_t seed, uint64_t first, uint64_t n, unsigned read_len)
	{
		RefBF* f = static_cast<RefBF*>(p);
		const unsigned h = f->getHashNum(), k = f->getKmerSize();
#pragma omp parallel
		{
			std::string s(read_len, 'A');
#pragma omp for schedule(dynamic, 1024)
			for (int64_t r = 0; r < (int64_t)n; ++r) {
				synth_read(seed, first + (uint64_t)r, read_len, &s[0]);
				ntHashIterator it(s, h, k);
				while (it != ntHashIterator::end()) {
					f->insert(*it);
					++it;
				}
			}
		}
	}
	// number of k-mers of synthetic reads [first, first+n) that the filter contains
	uint64_t ref_bf_count_synth(void* p, uint64_t seed, uint64_t first, uint64_t n, unsigned read_len)
	{
		RefBF* f = static_cast<RefBF*>(p);
		const unsigned h = f->getHashNum(), k = f->getKmerSize();
		uint64_t hits = 0;
#pragma omp parallel reduction(+ : hits)
		{
			std::string s(read_len, 'A');
#pragma omp for schedule(dynamic, 1024)
			for (int64_t r = 0; r < (int64_t)n; ++r) {
				synth_read(seed, first + (uint64_t)r, read_len, &s[0]);
				ntHashIterator it(s, h, k);
				while (it != ntHashIterator::end()) {
					hits += f->contains(*it) ? 1 : 0;
					++it;
				}
			}
		}
		return hits;
	}
	// ---- digests of a reference filter body, where it lies (the definition of btlbf_digest, include/btlbf.h) ----
	// With w_i the i-th little-endian 64-bit word of the body (BloomFilter.hpp:436 m_filter, written bit by bit at
	// :185-194; CountingBloomFilter.hpp:102 m_filter, :165-183) and m_i = mix64(i + 1) | 1:
	//   out2[0] = sum over the non-zero words of w_i * m_i (mod 2^64), out2[1] = xor over them of mix64(w_i ^ m_i).
	// Lets a 64 GiB reference filter built on the GPU box's host be compared with the array in HBM without moving
	// either (tests/test_gpu_ref_full_size.py, bench.py cpu_baseline "digest_equal").
	// out3[2] = set bits (what BloomFilter::getPop, BloomFilter.hpp:316-323, returns -- counted here with OpenMP and
	// popcount instead of its single-threaded byte table, which needs a minute for 64 GiB)
	void ref_bf_digest(void* p, uint64_t* out3)
	{
		RefBF* f = static_cast<RefBF*>(p);
		const uint8_t* b = f->bytes();
		const int64_t nb = (int64_t)f->sizeInBytes();
		const int64_t nw = nb / 8;
		uint64_t sum = 0, x = 0, pop = 0;
#pragma omp parallel for schedule(static) reduction(+ : sum, pop) reduction(^ : x)
		for (int64_t i = 0; i < nw; ++i) {
			uint64_t w;
			memcpy(&w, b + 8 * i, 8);
			if (w) {
				const uint64_t m = mix64((uint64_t)i + 1) | 1ULL;
				sum += w * m;
				x ^= mix64(w ^ m);
				pop += (uint64_t)__builtin_popcountll(w);
			}
		}
		if (nb % 8) { // a body that is no whole number of words: zero-padded
			uint64_t w = 0;
			memcpy(&w, b + 8 * nw, (size_t)(nb % 8));
			if (w) {
				const uint64_t m = mix64((uint64_t)nw + 1) | 1ULL;
				sum += w * m;
				x ^= mix64(w ^ m);
				pop += (uint64_t)__builtin_popcountll(w);
			}
		}
		out3[0] = sum;
		out3[1] = x;
		out3[2] = pop;
	}
	// out4[2] = non-zero counters (popCount, CountingBloomFilter.hpp:217-228), out4[3] = counters >= thr
	// (filtered_popcount, :231-242), counted with OpenMP
	void ref_cbf_digest(void* p, unsigned thr, uint64_t* out4)
	{
		RefCBF* f = static_cast<RefCBF*>(p);
		const int64_t n = (int64_t)f->size(); // counters = bytes; a multiple of 8 (CountingBloomFilter.hpp:34-45)
		const int64_t nw = (n + 7) / 8;
		uint64_t sum = 0, x = 0, nz = 0, ge = 0;
#pragma omp parallel for schedule(static) reduction(+ : sum, nz, ge) reduction(^ : x)
		for (int64_t i = 0; i < nw; ++i) {
			uint64_t w = 0;
			for (int64_t j = 0; j < 8 && 8 * i + j < n; ++j) {
				const uint64_t c = (*f)[(size_t)(8 * i + j)];
				w |= c << (8 * j);
				nz += c != 0;
				ge += c >= thr;
			}
			if (w) {
				const uint64_t m = mix64((uint64_t)i + 1) | 1ULL;
				sum += w * m;
				x ^= mix64(w ^ m);
			}
		}
		out4[0] = sum;
		out4[1] = x;
		out4[2] = nz;
		out4[3] = ge;
	}

	// incrementAll over synthetic reads, OpenMP over reads: a saturating CAS loop per counter
	// (CountingBloomFilter.hpp:165-183), so the bytes do not depend on the thread count
	void ref_cbf_increment_all_synth(void* p, uint64_t seed, uint64_t first, uint64_t n, unsigned read_len)
	{
		RefCBF* f = static_cast<RefCBF*>(p);
		const unsigned h = f->getHashNum(), k = f->getKmerSize();
#pragma omp parallel
		{
			std::string s(read_len, 'A');
#pragma omp for schedule(dynamic, 1024)
			for (int64_t r = 0; r < (int64_t)n; ++r) {
				synth_read(seed, first + (uint64_t)r, read_len, &s[0]);
				ntHashIterator it(s, h, k);
				while (it != ntHashIterator::end()) {
					f->incrementAll(*it);
					++it;
				}
			}
		}
	}
	// k-mers of synthetic reads [first, first+n) with minCount >= threshold (CountingBloomFilter.hpp:190-196)
	uint64_t ref_cbf_count_synth(void* p, uint64_t seed, uint64_t first, uint64_t n, unsigned read_len)
	{
		RefCBF* f = static_cast<RefCBF*>(p);
		const unsigned h = f->getHashNum(), k = f->getKmerSize();
		uint64_t hits = 0;
#pragma omp parallel reduction(+ : hits)
		{
			std::string s(read_len, 'A');
#pragma omp for schedule(dynamic, 1024)
			for (int64_t r = 0; r < (int64_t)n; ++r) {
				synth_read(seed, first + (uint64_t)r, read_len, &s[0]);
				ntHashIterator it(s, h, k);
				while (it != ntHashIterator::end()) {
					hits += f->contains(*it) ? 1 : 0;
					++it;
				}
			}
		}
		return hits;
	}

	// spaced seeds (BASELINE config 5): stHashIterator (vendor/stHashIterator.hpp:53-104) over synthetic reads into a
	// BloomFilter constructed with hashNum = nseeds * h2; query != 0: returns the number of k-mers found instead
	uint64_t ref_bf_spaced_synth(
	    void* p,
	    const char* const* seeds,
	    unsigned nseeds,
	    unsigned h2,
	    uint64_t seed,
	    uint64_t first,
	    uint64_t n,
	    unsigned read_len,
	    int query)
	{
		RefBF* f = static_cast<RefBF*>(p);
		const unsigned k = f->getKmerSize();
		std::vector<std::string> ss;
		for (unsigned i = 0; i < nseeds; ++i)
			ss.push_back(seeds[i]);
		const std::vector<std::vector<unsigned> > parsed = stHashIterator::parseSeed(ss);
		uint64_t hits = 0;
#pragma omp parallel reduction(+ : hits)
		{
			std::string s(read_len, 'A');
#pragma omp for schedule(dynamic, 1024)
			for (int64_t r = 0; r < (int64_t)n; ++r) {
				synth_read(seed, first + (uint64_t)r, read_len, &s[0]);
				stHashIterator it(s, parsed, nseeds, h2, k);
				while (it != stHashIterator::end()) {
					if (query)
						hits += f->contains(*it) ? 1 : 0;
					else
						f->insert(*it);
					++it;
				}
			}
		}
		return hits;
	}

	// serial (order-defined) counting-filter update over synthetic reads; op as ref_cbf_update
	void ref_cbf_update_synth(void* p, uint64_t seed, uint64_t first, uint64_t n, unsigned read_len, int op)
	{
		RefCBF* f = static_cast<RefCBF*>(p);
		const unsigned h = f->getHashNum(), k = f->getKmerSize();
		std::string s(read_len, 'A');
		for (uint64_t r = 0; r < n; ++r) {
			synth_read(seed, first + r, read_len, &s[0]);
			ntHashIterator it(s, h, k);
			while (it != ntHashIterator::end()) {
				if (op == 0)
					f->insert(*it);
				else
					f->incrementAll(*it);
				++it;
			}
		}
	}

	// The hot loop of SURVEY.md section 3.1/3.2 over synthetic reads, OpenMP over reads.
	// out[0]=insert seconds, out[1]=query seconds, out[2]=query hits, out[3]=k-mers per pass,
	// out[4]=threads used, out[5]=popcount after insert (0 if skip_pop)
	// digest2 != NULL: also the digest (ref_bf_digest) of the filter the timed insert built
	int ref_bench_bf_digest(
	    uint64_t n_reads,
	    unsigned read_len,
	    unsigned k,
	    unsigned h,
	    uint64_t bits,
	    uint64_t seed_ins,
	    uint64_t seed_qry,
	    int threads,
	    int prefault,
	    int skip_pop,
	    double* out,
	    uint64_t* digest2)
	{
		RefBF f(bits, h, k);
#ifdef _OPENMP
		if (threads > 0)
			omp_set_num_threads(threads);
		int used = omp_get_max_threads();
#else
		int used = 1;
#endif
		if (prefault) { // touch every page so first-touch faults are not billed to insert
			uint8_t* b = f.bytes();
			const int64_t nb = (int64_t)f.sizeInBytes();
#pragma omp parallel for schedule(static)
			for (int64_t i = 0; i < nb; i += 4096)
				((volatile uint8_t*)b)[i] = 0;
		}
		double t0 = now_s();
#pragma omp parallel
		{
			std::string s(read_len, 'A');
#pragma omp for schedule(dynamic, 1024)
			for (int64_t r = 0; r < (int64_t)n_reads; ++r) {
				synth_read(seed_ins, (uint64_t)r, read_len, &s[0]);
				ntHashIterator it(s, h, k);
				while (it != ntHashIterator::end()) {
					f.insert(*it);
					++it;
				}
			}
		}
		double t1 = now_s();
		uint64_t hits = 0, kmers = 0;
#pragma omp parallel reduction(+ : hits, kmers)
		{
			std::string s(read_len, 'A');
#pragma omp for schedule(dynamic, 1024)
			for (int64_t r = 0; r < (int64_t)n_reads; ++r) {
				synth_read(seed_qry, (uint64_t)r, read_len, &s[0]);
				ntHashIterator it(s, h, k);
				while (it != ntHashIterator::end()) {
					hits += f.contains(*it) ? 1 : 0;
					++kmers;
					++it;
				}
			}
		}
		double t2 = now_s();
		out[0] = t1 - t0;
		out[1] = t2 - t1;
		out[2] = (double)hits;
		out[3] = (double)kmers;
		out[4] = (double)used;
		out[5] = skip_pop ? 0.0 : (double)f.getPop();
		if (digest2) {
			uint64_t d3[3];
			ref_bf_digest(&f, d3);
			digest2[0] = d3[0];
			digest2[1] = d3[1];
		}
		return 0;
	}
	int ref_bench_bf(
	    uint64_t n_reads,
	    unsigned read_len,
	    unsigned k,
	    unsigned h,
	    uint64_t bits,
	    uint64_t seed_ins,
	    uint64_t seed_qry,
	    int threads,
	    int prefault,
	    int skip_pop,
	    double* out)
	{
		return ref_bench_bf_digest(n_reads, read_len, k, h, bits, seed_ins, seed_qry, threads, prefault, skip_pop, out, NULL);
	}

	// cost of generating the synthetic reads alone (to subtract from the loop above)
	double ref_bench_synth_only(uint64_t n_reads, unsigned read_len, uint64_t seed, int threads)
	{
#ifdef _OPENMP
		if (threads > 0)
			omp_set_num_threads(threads);
#endif
		uint64_t acc = 0;
		double t0 = now_s();
#pragma omp parallel reduction(+ : acc)
		{
			std::string s(read_len, 'A');
#pragma omp for schedule(dynamic, 1024)
			for (int64_t r = 0; r < (int64_t)n_reads; ++r) {
				synth_read(seed, (uint64_t)r, read_len, &s[0]);
				acc += (unsigned char)s[r % read_len];
			}
		}
		double t1 = now_s();
		return (t1 - t0) + (acc == 1 ? 1e-12 : 0.0);
	}

} // extern "C"

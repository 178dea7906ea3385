// include/btlbf/KmerBloomFilter.hpp -- drop-in for the reference's `KmerBloomFilter`
// (/root/reference/KmerBloomFilter.hpp:17-75): a BloomFilter that also takes raw k-mer strings.
//
// insert(const char*) / contains(const char*) hash the k bytes at `kmer` on the GPU with the values of the
// reference's raw-k-mer path, NTC64(kmerSeq, k) + NTE64 (vendor/nthash.hpp:394-439,460-465,537-542), as its
// x86-64 build returns them -- including where that path disagrees with the reference's own iterator:
// k % 4 == 0 (the table walk's shift by 64, nthash.hpp:354-356,388-391,404-406) and k-mers with U
// (read as A, nthash.hpp:16-86).  A filter filled here through insert(const char*) is byte-identical to the
// reference's (swig/test.pl:11-16 builds one with k = 20), and a reference-built file answers
// contains(const char*) the same way.  See btlbf_insert_kmers in btlbf.h.
#ifndef BTLBF_KMERBLOOMFILTER_HPP
#define BTLBF_KMERBLOOMFILTER_HPP
#include "BloomFilter.hpp"

class KmerBloomFilter : public BloomFilter
{
  public:
	KmerBloomFilter() = default;
	KmerBloomFilter(size_t filterSize, unsigned hashNum, unsigned kmerSize)
	  : BloomFilter(filterSize, hashNum, kmerSize)
	{}
	explicit KmerBloomFilter(const std::string& filterFilePath)
	  : BloomFilter(filterFilePath)
	{}

	using BloomFilter::contains;
	using BloomFilter::insert;

	bool contains(const char* kmer) const // KmerBloomFilter.hpp:47-61
	{
		flush();
		uint8_t hit = 0;
		btlbf_shim::check(btlbf_contains_kmers(m_f, kmer, 1, &hit, BTLBF_HOST, nullptr));
		return hit != 0;
	}

	// write-combined like BloomFilter::insert(hashes): k-mers are queued per thread and pushed as one
	// btlbf_insert_kmers batch
	void insert(const char* kmer) // KmerBloomFilter.hpp:63-74
	{
		{
			Stripe& st = my_stripe();
			std::lock_guard<std::mutex> g(st.mu);
			st.kmers.append(kmer, getKmerSize());
			if (st.kmers.size() >= kFlushKmers * (size_t)getKmerSize()) {
				push_kmers(st.kmers);
				st.kmers.clear();
			}
			st.ver.store(st.ver.load(std::memory_order_relaxed) + 1, std::memory_order_release);
		}
	}

	// batch forms (no counterpart in the reference): n k-mers of kmerSize bytes each, back to back
	void insertKmers(const char* kmers, size_t n)
	{
		flush();
		btlbf_shim::check(btlbf_insert_kmers(m_f, kmers, n, 0, BTLBF_ORDER_PARALLEL, BTLBF_HOST, nullptr));
		touch();
	}
	std::vector<uint8_t> containsKmers(const char* kmers, size_t n) const
	{
		flush();
		std::vector<uint8_t> out(n ? n : 1);
		btlbf_shim::check(btlbf_contains_kmers(m_f, kmers, n, out.data(), BTLBF_HOST, nullptr));
		out.resize(n);
		return out;
	}
};

#endif

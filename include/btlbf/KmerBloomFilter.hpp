// include/btlbf/KmerBloomFilter.hpp -- drop-in for the reference's `KmerBloomFilter`
// (/root/reference/KmerBloomFilter.hpp:17-75): a BloomFilter that also takes raw k-mer strings.
//
// insert(const char*) / contains(const char*) hash the k bytes at `kmer` on the GPU exactly like a
// one-window sequence.  Defined for A/C/G/T in either case.  Two corner cases of the reference's
// tetramer-table path are NOT reproduced (see tests/golden/make_golden.py, "ub"): for k % 4 == 0 it
// shifts a uint64_t by 64 (undefined behaviour, nthash.hpp:354-356,388-391), and it maps 'U' to 'A'
// (nthash.hpp:16-86) while the iterator path maps 'U' to 'T'.  This class always agrees with the
// iterator path, which is what ends up in filters built with insertSeq.
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
		uint64_t hit = 0;
		btlbf_shim::check(btlbf_contains_seqs(m_f, kmer, getKmerSize(), nullptr, &hit, nullptr, nullptr,
		                                      BTLBF_HOST, nullptr));
		return hit & 1u;
	}

	void insert(const char* kmer) // KmerBloomFilter.hpp:63-74
	{
		flush();
		btlbf_shim::check(btlbf_insert_seqs(m_f, kmer, getKmerSize(), nullptr, 0, BTLBF_ORDER_PARALLEL,
		                                    BTLBF_HOST, nullptr));
	}
};

#endif

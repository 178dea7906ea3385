// include/btlbf/BloomFilterUtil.h -- drop-in for /root/reference/BloomFilterUtil.h.
// insertSeq is the reference's canonical hot loop (BloomFilterUtil.h:9-17: ntHashIterator +
// BloomFilter::insert); here it is ONE fused GPU launch (ntHash + atomicOr) when the iterator's
// (hashNum, kmerSize) are the filter's own, and the literal iterator loop otherwise.
#ifndef BTLBF_BLOOMFILTERUTIL_H
#define BTLBF_BLOOMFILTERUTIL_H
#include "KmerBloomFilter.hpp"
#include "ntHashIterator.hpp"

#include <cmath>

inline void
insertSeq(BloomFilter& bloom, const std::string& seq, unsigned hashNum, unsigned kmerSize)
{
	if (hashNum == bloom.getHashNum() && kmerSize == bloom.getKmerSize()) {
		bloom.insertSeq(seq);
		return;
	}
	ntHashIterator itr(seq, hashNum, kmerSize);
	while (itr != itr.end()) {
		bloom.insert(*itr);
		++itr;
	}
}

// BloomFilterUtil.h:28-46
inline double
calcApproxFPR(size_t size, size_t numEntr, unsigned hashFunctNum)
{
	const double h = double(hashFunctNum);
	return std::pow(1.0 - std::pow(1.0 - 1.0 / double(size), double(numEntr) * h), h);
}

inline double
calcRedunancyFPR(size_t size, size_t numEntr, unsigned hashFunctNum)
{
	double total = std::log(calcApproxFPR(size, 1, hashFunctNum));
	for (size_t i = 2; i < numEntr; ++i)
		total = std::log(std::exp(total) + calcApproxFPR(size, i, hashFunctNum));
	return std::exp(total) / double(numEntr);
}

#endif

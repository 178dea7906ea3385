// tests/cpp/test_swig_surface.cpp -- compile-time check that every declaration in bindings/swig/BloomFilter.i
// exists in the shim headers with that signature (what swig would generate calls to); never run.
#include "btlbf/BloomFilterUtil.h"
#include "btlbf/KmerBloomFilter.hpp"
#include "btlbf/ntHashIterator.hpp"

#include <string>
#include <vector>

template<typename M>
static void use(M) {}

int main()
{
	using std::string;
	using std::vector;
	use(static_cast<void (KmerBloomFilter::*)(vector<uint64_t> const&)>(&KmerBloomFilter::insert));
	use(static_cast<void (KmerBloomFilter::*)(const char*)>(&KmerBloomFilter::insert));
	use(static_cast<bool (KmerBloomFilter::*)(vector<uint64_t> const&) const>(&KmerBloomFilter::contains));
	use(static_cast<bool (KmerBloomFilter::*)(const char*) const>(&KmerBloomFilter::contains));
	use(static_cast<void (BloomFilter::*)(const string&) const>(&KmerBloomFilter::storeFilter));
	use(static_cast<uint64_t (BloomFilter::*)() const>(&KmerBloomFilter::getPop));
	use(static_cast<unsigned (BloomFilter::*)() const>(&KmerBloomFilter::getHashNum));
	use(static_cast<unsigned (BloomFilter::*)() const>(&KmerBloomFilter::getKmerSize));
	use(static_cast<uint64_t (BloomFilter::*)() const>(&KmerBloomFilter::getFilterSize));
	use(static_cast<void (BloomFilter::*)(const string&)>(&KmerBloomFilter::insertSeq));
	use(static_cast<void (BloomFilter::*)(const string&, vector<bool>&, vector<bool>&) const>(&KmerBloomFilter::containsSeq));
	if (false) {
		KmerBloomFilter a, b((uint64_t)1024, 4u, 31u), c(string("x.bf"));
		insertSeq(b, string("ACGT"), 4u, 31u); // the call swig generates for `void insertSeq(KmerBloomFilter&, ...)`
		(void)a;
		(void)c;
	}
	return 0;
}

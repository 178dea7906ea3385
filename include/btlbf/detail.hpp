// include/btlbf/detail.hpp -- helpers shared by the drop-in C++ shims over the C ABI (btlbf.h).
//
// The shims keep the reference's error convention: a failed call prints a message on std::cerr
// and terminates the process with exit(1) (reference: BloomFilter.hpp:124-129,144-148,391-394;
// vendor/IOUtil.h:14-22).  Define BTLBF_SHIM_THROW to get std::runtime_error instead.
#ifndef BTLBF_DETAIL_HPP
#define BTLBF_DETAIL_HPP
#include "../btlbf.h"

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace btlbf_shim {

inline int&
default_device()
{
	static int dev = [] {
		const char* e = std::getenv("BTLBF_DEVICE");
		return e ? std::atoi(e) : 0;
	}();
	return dev;
}

inline void
check(int rc)
{
	if (rc == BTLBF_OK)
		return;
#ifdef BTLBF_SHIM_THROW
	throw std::runtime_error(btlbf_last_error());
#else
	std::cerr << btlbf_last_error() << std::endl;
	std::exit(EXIT_FAILURE);
#endif
}

inline bool
bit(const uint64_t* words, uint64_t p)
{
	return (words[p >> 6] >> (p & 63)) & 1u;
}

// ---- look-ahead for the reference's query loop ---------------------------------------------------------------------
//   ntHashIterator itr(seq, h, k);  while (itr != itr.end()) { if (bloom.contains(*itr)) ...; ++itr; }
// (Tests/AdHoc/ParallelFilter.cpp:93-101 with contains in place of insert).  One contains() is one GPU round trip; the
// rows the iterator will hand out next are already in host memory, in the iterator's own buffer.  So an iterator
// announces that buffer to its thread (HashSpan), and BloomFilter::contains(p) with p inside the announced buffer
// answers ALL its rows in one call and keeps the answers (SpanCache) for as long as neither the buffer nor the filter
// changes: one round trip per read instead of one per k-mer, invisible to the caller.
struct HashSpan {
	const uint64_t* base = nullptr;           // rows[0]
	size_t rows = 0;                          // rows of `stride` values each
	unsigned stride = 0;
	uint64_t id = 0;                          // changes with every refill of the buffer
	std::shared_ptr<std::atomic<bool>> alive; // cleared by the iterator's destructor (on whichever thread that runs)
};
inline HashSpan&
tls_span()
{
	static thread_local HashSpan s;
	return s;
}
inline uint64_t
next_span_id()
{
	static std::atomic<uint64_t> c{ 1 };
	return c.fetch_add(1, std::memory_order_relaxed);
}
struct SpanCache {
	const void* filter = nullptr;
	uint64_t version = 0, span_id = 0;
	std::vector<uint8_t> hit; // one byte per row of the span
};
inline SpanCache&
tls_cache()
{
	static thread_local SpanCache c;
	return c;
}

} // namespace btlbf_shim
#endif

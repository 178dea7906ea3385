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
// answers a window of its rows in one call and keeps the answers (SpanCache) for as long as neither the buffer nor the
// filter changes: one round trip per read (per 4096 k-mers of a long sequence) instead of one per k-mer, invisible to
// the caller.
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
// The answers one thread keeps: per FILTER (a loop that asks two filters by turns -- `a.contains(*itr) &&
// b.contains(*itr)` -- must not throw one's answers away for the other's), for a WINDOW of at most kWindow rows ahead of
// the row asked for (an iterator over a chromosome announces up to 2^20 rows: re-querying all of them after every
// mutation made `if (!bf.contains(*itr)) bf.insert(*itr)` quadratic), and with a way out: a span whose answers were
// voided three times in a row after serving one row each goes back to the one-row path for the rest of that span.
struct SpanCache {
	static constexpr size_t kWindow = 4096, kFilters = 4;
	static constexpr unsigned kGiveUp = 3;
	struct Entry {
		const void* filter = nullptr;
		uint64_t version = 0, span_id = 0;
		size_t first = 0;         // row of the span that hit[0] answers
		std::vector<uint8_t> hit; // one byte per row of the window
		unsigned served = 0;      // rows answered since the last refresh
		unsigned streak = 0;      // refreshes in a row that had served at most one row
		uint64_t off_span = 0;    // the span this filter has given up on (0: none)
		uint64_t stamp = 0;
	};
	Entry e[kFilters];
	uint64_t clock = 0;
	Entry& find(const void* filter)
	{
		Entry* lru = &e[0];
		for (Entry& x : e) {
			if (x.filter == filter) {
				x.stamp = ++clock;
				return x;
			}
			if (x.stamp < lru->stamp)
				lru = &x;
		}
		*lru = Entry();
		lru->filter = filter;
		lru->stamp = ++clock;
		return *lru;
	}
};
inline SpanCache&
tls_cache()
{
	static thread_local SpanCache c;
	return c;
}

} // namespace btlbf_shim
#endif

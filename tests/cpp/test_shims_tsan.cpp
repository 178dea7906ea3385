// tests/cpp/test_shims_tsan.cpp -- the CPU side of the drop-in shims under ThreadSanitizer: many threads
// calling insert / contains / insertAndCheck / getPop on ONE BloomFilter and increments on ONE
// CountingBloomFilter, the reference's threading contract (BloomFilter.hpp:177,191,206-210;
// Tests/AdHoc/ParallelFilter.cpp:104-122).  Linked against tests/cpp/stub_abi.cpp (test-only, no GPU):
//   g++ -std=c++17 -O1 -g -fsanitize=thread -Iinclude tests/cpp/test_shims_tsan.cpp tests/cpp/stub_abi.cpp -lpthread
#define BTLBF_SHIM_MINIMAL
#include "btlbf/BloomFilter.hpp"
#include "btlbf/CountingBloomFilter.hpp"
#include "btlbf/KmerBloomFilter.hpp"

#include <atomic>
#include <cstdio>
#include <memory>
#include <thread>
#include <vector>

static uint64_t mix(uint64_t z)
{
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

int main()
{
	const unsigned h = 4, T = 8, per = 80000;  // > kFlushRows rows per stripe: queues fill and flush under load
	KmerBloomFilter bloom(1 << 26, h, 31);
	CountingBloomFilter<uint8_t> cbf(1 << 20, h, 31, 1);
	std::vector<std::thread> th;
	std::vector<unsigned long> miss(T, 0);
	for (unsigned t = 0; t < T; ++t)
		th.emplace_back([&, t] {
			uint64_t row[4];
			for (unsigned i = 0; i < per; ++i) {
				for (unsigned j = 0; j < h; ++j)
					row[j] = mix((uint64_t)t * per * h + (uint64_t)i * h + j + 1);
				bloom.insert(row);
				if (i % 4096 == 0) { // a reader (and an insertAndCheck) among the writers
					miss[t] += !bloom.contains(row);
					(void)bloom.insertAndCheck(row);
					(void)bloom.getPop();
				}
				if (i % 16 == 0) { // the sequence and raw-k-mer queues, filled and flushed from all threads
					const std::string seq(40 + i % 300, "ACGT"[i & 3]);
					bloom.insertSeq(seq);
					bloom.insert(seq.c_str());
				}
				if (i % 8192 == 4096) {
					// the query loop's look-ahead (detail.hpp): 32 rows announced the way an ntHashIterator does,
					// asked one by one while the other threads keep inserting; the first 17 were inserted by this
					// thread a moment ago and must be found
					std::vector<uint64_t> rows(32 * h);
					for (unsigned q = 0; q < 32; ++q)
						for (unsigned j = 0; j < h; ++j)
							rows[q * h + j] = mix((uint64_t)t * per * h + (uint64_t)(i - 16 + q) * h + j + 1);
					auto alive = std::make_shared<std::atomic<bool>>(true);
					btlbf_shim::HashSpan& sp = btlbf_shim::tls_span();
					sp.base = rows.data();
					sp.rows = 32;
					sp.stride = h;
					sp.id = btlbf_shim::next_span_id();
					sp.alive = alive;
					for (unsigned q = 0; q < 32; ++q) {
						const bool hit = bloom.contains(&rows[q * h]);
						if (q <= 16)
							miss[t] += !hit;
					}
					alive->store(false);
				}
				if (i % 64 == 0) {
					cbf.incrementAll(row);
					miss[t] += !cbf.contains(row);
				}
			}
		});
	for (auto& x : th)
		x.join();
	unsigned long bad = 0;
	for (unsigned t = 0; t < T; ++t)
		bad += miss[t];
	// every row ever inserted is found afterwards
	for (unsigned t = 0; t < T; ++t)
		for (unsigned i = 0; i < per; i += 997) {
			uint64_t row[4];
			for (unsigned j = 0; j < h; ++j)
				row[j] = mix((uint64_t)t * per * h + (uint64_t)i * h + j + 1);
			bad += !bloom.contains(row);
		}
	if (bad) {
		std::fprintf(stderr, "%lu rows not found\n", bad);
		return 1;
	}
	std::printf("shim threading test passed\n");
	return 0;
}

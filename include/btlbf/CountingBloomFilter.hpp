// include/btlbf/CountingBloomFilter.hpp -- drop-in for the reference's `CountingBloomFilter<T>`
// (/root/reference/CountingBloomFilter.hpp:26-111) with T = uint8_t, counters resident in HBM.
//
// Same constructors and method names; `U` is any indexable holder of m_hashNum hash values, as in
// the reference.  Per-k-mer calls apply in call order (BTLBF_ORDER_SERIAL on one row), so a
// single-threaded caller gets exactly the reference's counters; the batch members run in parallel
// (incrementAll stays exact; insert = incrementMin is order-dependent in the reference too).
// Threads: per-k-mer calls may come from many threads at once, as in the reference
// (CountingBloomFilter.hpp:117-132); they are serialised by the filter's lock inside the C ABI.
// Not reproduced: loadFilter's resize to sizeInBytes *elements* (CountingBloomFilter.hpp:275), an
// over-allocation that only matters for T wider than a byte.
#ifndef BTLBF_COUNTINGBLOOMFILTER_HPP
#define BTLBF_COUNTINGBLOOMFILTER_HPP
#include "detail.hpp"

#include <algorithm>
#include <cmath>
#include <istream>
#include <string>
#include <vector>

template<typename T>
class CountingBloomFilter
{
	static_assert(sizeof(T) == 1, "the MI355X engine implements 8-bit counters (uint8_t)");

  public:
	CountingBloomFilter() = default;
	CountingBloomFilter(size_t sizeInBytes, unsigned hashNum, unsigned kmerSize, unsigned countThreshold)
	{
		btlbf_shim::check(btlbf_create(&m_f, BTLBF_COUNTING8, sizeInBytes, hashNum, kmerSize, countThreshold,
		                               btlbf_shim::default_device()));
	}
	CountingBloomFilter(const std::string& path, unsigned countThreshold)
	  : m_threshold(countThreshold)
	{
		loadFilter(path);
	}
	~CountingBloomFilter() { btlbf_destroy(m_f); }
	CountingBloomFilter(const CountingBloomFilter&) = delete;
	CountingBloomFilter& operator=(const CountingBloomFilter&) = delete;

	T operator[](size_t i)
	{
		uint8_t v = 0;
		btlbf_shim::check(btlbf_download(m_f, &v, i, 1));
		return (T)v;
	}

	template<typename U>
	T minCount(const U& hashes) const // CountingBloomFilter.hpp:53-64
	{
		uint8_t v = 0;
		btlbf_shim::check(btlbf_min_count_hashes(m_f, row(hashes), 1, &v, BTLBF_HOST, nullptr));
		return (T)v;
	}
	template<typename U>
	bool contains(const U& hashes) const // :190-196
	{
		uint8_t v = 0;
		btlbf_shim::check(btlbf_contains_hashes(m_f, row(hashes), 1, &v, BTLBF_HOST, nullptr));
		return v != 0;
	}
	template<typename U>
	void insert(const U& hashes) // :198-204
	{
		incrementMin(hashes);
	}
	template<typename U>
	bool insertAndCheck(const U& hashes) // :206-214
	{
		uint8_t v = 0;
		btlbf_shim::check(btlbf_insert_and_check_hashes(m_f, row(hashes), 1, &v, BTLBF_ORDER_SERIAL, BTLBF_HOST,
		                                                nullptr));
		return v != 0;
	}
	template<typename U>
	void incrementMin(const U& hashes) // :135-162
	{
		btlbf_shim::check(btlbf_insert_hashes(m_f, row(hashes), 1, BTLBF_INCREMENT_MIN, BTLBF_ORDER_SERIAL,
		                                      BTLBF_HOST, nullptr));
	}
	template<typename U>
	void incrementAll(const U& hashes) // :165-183
	{
		btlbf_shim::check(btlbf_insert_hashes(m_f, row(hashes), 1, BTLBF_INCREMENT_ALL, BTLBF_ORDER_SERIAL,
		                                      BTLBF_HOST, nullptr));
	}

	// ---- batch interface (fast path) ----
	void insertSeq(const std::string& seq, bool incrementAllCounters = false, bool serialOrder = false)
	{
		btlbf_shim::check(btlbf_insert_seqs(m_f, seq.data(), seq.size(), nullptr,
		                                    incrementAllCounters ? BTLBF_INCREMENT_ALL : BTLBF_INCREMENT_MIN,
		                                    serialOrder ? BTLBF_ORDER_SERIAL : BTLBF_ORDER_PARALLEL, BTLBF_HOST,
		                                    nullptr));
	}
	void containsSeq(const std::string& seq, std::vector<bool>& result, std::vector<bool>& valid) const
	{
		const size_t nw = (seq.size() + 63) / 64;
		std::vector<uint64_t> hb(nw ? nw : 1), vb(nw ? nw : 1);
		btlbf_shim::check(btlbf_contains_seqs(m_f, seq.data(), seq.size(), nullptr, hb.data(), vb.data(),
		                                      nullptr, BTLBF_HOST, nullptr));
		result.assign(seq.size(), false);
		valid.assign(seq.size(), false);
		for (size_t p = 0; p < seq.size(); ++p) {
			result[p] = btlbf_shim::bit(hb.data(), p);
			valid[p] = btlbf_shim::bit(vb.data(), p);
		}
	}

	unsigned getKmerSize() const { return btlbf_kmer_size(m_f); }
	unsigned getHashNum() const { return btlbf_hash_num(m_f); }
	unsigned threshold() const { return btlbf_threshold(m_f); }
	size_t size() const { return btlbf_size(m_f); }
	size_t sizeInBytes() const { return btlbf_size_bytes(m_f); }
	size_t popCount() const // :217-228
	{
		uint64_t v = 0;
		btlbf_shim::check(btlbf_popcount(m_f, &v));
		return v;
	}
	size_t filtered_popcount() const // :231-242
	{
		uint64_t v = 0;
		btlbf_shim::check(btlbf_filtered_popcount(m_f, &v));
		return v;
	}
	double FPR() const { return std::pow((double)popCount() / (double)size(), getHashNum()); }
	double filtered_FPR() const { return std::pow((double)filtered_popcount() / (double)size(), getHashNum()); }

	void loadFilter(const std::string& path) // :268-280
	{
		btlbf_destroy(m_f);
		m_f = nullptr;
		btlbf_shim::check(btlbf_load(&m_f, BTLBF_COUNTING8, path.c_str(), m_threshold, btlbf_shim::default_device()));
	}
	// loadHeader(std::istream&), CountingBloomFilter.hpp:84,282-330: the header lines up to "[HeaderEnd]" -> a zeroed
	// filter of that geometry; the stream is left at the body (loadBody)
	void loadHeader(std::istream& file)
	{
		std::string text, line;
		while (std::getline(file, line)) {
			text += line + "\n";
			if (line == "[HeaderEnd]" || text.size() > (1u << 16))
				break;
		}
		btlbf_destroy(m_f);
		m_f = nullptr;
		btlbf_shim::check(btlbf_create_from_header(&m_f, BTLBF_COUNTING8, text.data(), text.size(), m_threshold,
		                                           btlbf_shim::default_device()));
	}
	void loadBody(std::istream& file)
	{
		std::vector<char> chunk(1u << 24);
		const uint64_t total = sizeInBytes();
		for (uint64_t off = 0; off < total; off += chunk.size()) {
			const uint64_t n = std::min<uint64_t>(chunk.size(), total - off);
			file.read(chunk.data(), (std::streamsize)n);
			if ((uint64_t)file.gcount() != n) {
				std::cerr << "error: short read of the filter body" << std::endl;
				std::exit(EXIT_FAILURE);
			}
			btlbf_shim::check(btlbf_upload(m_f, chunk.data(), off, n));
		}
	}
	void storeFilter(const std::string& path) const // :331-342
	{
		std::cerr << "Writing a " << sizeInBytes() << " byte filter to " << path << " on disk.\n";
		btlbf_shim::check(btlbf_store(m_f, path.c_str()));
	}
	void storeHeader(std::ostream& out) const // :344-368
	{
		char buf[1024];
		size_t n = 0;
		btlbf_shim::check(btlbf_header(m_f, buf, sizeof buf, &n));
		out.write(buf, (std::streamsize)n);
	}
	friend std::ostream& operator<<(std::ostream& out, const CountingBloomFilter& bloom) // :371-379
	{
		bloom.storeHeader(out);
		std::vector<char> body(bloom.sizeInBytes());
		btlbf_shim::check(btlbf_download(bloom.m_f, body.data(), 0, body.size()));
		out.write(body.data(), (std::streamsize)body.size());
		return out;
	}

	btlbf_filter* handle() const { return m_f; }

  private:
	// the first m_hashNum values of any indexable holder as a contiguous row.  A value, not a member: the
	// reference's per-k-mer calls may come from many threads at once (CountingBloomFilter.hpp:117-132)
	struct Row {
		std::vector<uint64_t> v;
		operator const uint64_t*() const { return v.data(); }
	};
	template<typename U>
	Row row(const U& hashes) const
	{
		Row r;
		const unsigned h = getHashNum();
		r.v.resize(h);
		for (unsigned i = 0; i < h; ++i)
			r.v[i] = hashes[i];
		return r;
	}

	btlbf_filter* m_f = nullptr;
	unsigned m_threshold = 0;
};

#endif

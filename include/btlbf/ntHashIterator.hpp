// include/btlbf/ntHashIterator.hpp -- drop-in for the reference's `ntHashIterator`
// (/root/reference/vendor/ntHashIterator.hpp:18-151): walks a sequence and yields, for every k-mer
// made only of valid bases, a pointer to its h hash values.
//
// The hash values come from the GPU: the constructor (and every refill) sends a chunk of the
// sequence through btlbf_hash_seqs, which hashes all windows of the chunk at once; operator++ then
// just steps to the next clean window.  Interface kept: ctor (seq, h, k, pos = 0), operator*, ++,
// ==, !=, static end(), pos().  Like the reference the iterator owns a copy of the sequence and
// operator*'s pointer is valid until the next ++.
#ifndef BTLBF_NTHASHITERATOR_HPP
#define BTLBF_NTHASHITERATOR_HPP
#include "detail.hpp"

#include <atomic>
#include <limits>
#include <memory>
#include <string>
#include <vector>

class ntHashIterator
{
  public:
	ntHashIterator()
	  : m_pos(npos())
	{}
	// copies hash for themselves and announce nothing; a moved-from iterator is empty
	ntHashIterator(const ntHashIterator& o)
	  : m_seq(o.m_seq)
	  , m_h(o.m_h)
	  , m_k(o.m_k)
	  , m_pos(o.m_pos)
	  , m_chunk0(o.m_chunk0)
	  , m_chunk_n(o.m_chunk_n)
	  , m_hashes(o.m_hashes)
	  , m_valid(o.m_valid)
	{}
	ntHashIterator& operator=(const ntHashIterator& o)
	{
		if (this != &o) {
			retire();
			m_seq = o.m_seq;
			m_h = o.m_h;
			m_k = o.m_k;
			m_pos = o.m_pos;
			m_chunk0 = o.m_chunk0;
			m_chunk_n = o.m_chunk_n;
			m_hashes = o.m_hashes;
			m_valid = o.m_valid;
		}
		return *this;
	}
	~ntHashIterator() { retire(); }

	ntHashIterator(const std::string& seq, unsigned h, unsigned k, size_t pos = 0)
	  : m_seq(seq)
	  , m_h(h)
	  , m_k(k)
	  , m_pos(pos)
	{
		if (m_k > m_seq.length() || m_pos > m_seq.length() - m_k) {
			m_pos = npos();
			return;
		}
		seek();
	}

	const uint64_t* operator*() const { return &m_hashes[(m_pos - m_chunk0) * m_h]; }
	size_t pos() const { return m_pos; }
	bool operator==(const ntHashIterator& it) const { return m_pos == it.m_pos; }
	bool operator!=(const ntHashIterator& it) const { return !(*this == it); }
	ntHashIterator& operator++()
	{
		++m_pos;
		seek();
		return *this;
	}
	static const ntHashIterator end() { return ntHashIterator(); }

  private:
	static size_t npos() { return std::numeric_limits<std::size_t>::max(); }
	static constexpr size_t kChunk = 1u << 20; // windows hashed per GPU call

	// hash windows [start, start + kChunk) on the GPU
	void fill(size_t start)
	{
		const size_t n_win = m_seq.length() - m_k + 1;
		const size_t cnt = std::min(kChunk, n_win - start);
		const size_t bytes = cnt + m_k - 1;
		m_chunk0 = start;
		m_chunk_n = cnt;
		m_hashes.resize(bytes * m_h);
		m_valid.assign((bytes + 63) / 64, 0);
		retire(); // (the buffer may have moved, its rows are new)
		btlbf_shim::check(btlbf_hash_seqs(m_k, m_h, nullptr, 0, 0, m_seq.data() + start, bytes, nullptr,
		                                  m_hashes.data(), m_valid.data(), nullptr, BTLBF_HOST,
		                                  btlbf_shim::default_device(), BTLBF_STREAM_PER_THREAD));
		// announce the rows to this thread's filters (detail.hpp: the look-ahead of BloomFilter::contains)
		m_alive = std::make_shared<std::atomic<bool>>(true);
		btlbf_shim::HashSpan& sp = btlbf_shim::tls_span();
		sp.base = m_hashes.data();
		sp.rows = cnt;
		sp.stride = m_h;
		sp.id = btlbf_shim::next_span_id();
		sp.alive = m_alive;
	}
	// the announced rows are about to change or to go away
	void retire()
	{
		if (m_alive) {
			m_alive->store(false, std::memory_order_release);
			m_alive.reset();
		}
	}

	// advance m_pos to the first clean window at or after it (ntHashIterator.hpp:59-86)
	void seek()
	{
		const size_t n_win = m_seq.length() - m_k + 1;
		while (m_pos < n_win) {
			if (m_chunk_n == 0 || m_pos < m_chunk0 || m_pos >= m_chunk0 + m_chunk_n)
				fill(m_pos);
			const size_t stop = m_chunk0 + m_chunk_n;
			while (m_pos < stop && !btlbf_shim::bit(m_valid.data(), m_pos - m_chunk0))
				++m_pos;
			if (m_pos < stop)
				return;
		}
		m_pos = npos();
	}

	std::string m_seq;
	unsigned m_h = 0, m_k = 0;
	size_t m_pos;
	size_t m_chunk0 = 0, m_chunk_n = 0;
	std::vector<uint64_t> m_hashes;
	std::vector<uint64_t> m_valid;
	std::shared_ptr<std::atomic<bool>> m_alive;
};

#endif

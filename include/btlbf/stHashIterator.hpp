// include/btlbf/stHashIterator.hpp -- drop-in for the reference's `stHashIterator`
// (/root/reference/vendor/stHashIterator.hpp:18-174): spaced-seed hashes (h seeds x h2 hashes per
// seed) plus a strand flag per hash for every clean k-mer of a sequence, computed on the GPU
// (btlbf_hash_seqs with seeds).  parseSeed keeps the reference's representation: per seed, the
// indices of the characters that are not '1' (stHashIterator.hpp:23-33).
#ifndef BTLBF_STHASHITERATOR_HPP
#define BTLBF_STHASHITERATOR_HPP
#include "detail.hpp"

#include <limits>
#include <memory>
#include <string>
#include <vector>

class stHashIterator
{
  public:
	static std::vector<std::vector<unsigned> > parseSeed(const std::vector<std::string>& seedString)
	{
		std::vector<std::vector<unsigned> > seedSet;
		for (const std::string& s : seedString) {
			std::vector<unsigned> dontCare;
			for (unsigned j = 0; j < s.size(); ++j)
				if (s[j] != '1')
					dontCare.push_back(j);
			seedSet.push_back(dontCare);
		}
		return seedSet;
	}

	stHashIterator()
	  : m_pos(npos())
	{}

	stHashIterator(
	    const std::string& seq,
	    const std::vector<std::vector<unsigned> >& seed,
	    unsigned h,
	    unsigned h2,
	    unsigned k,
	    size_t pos = 0)
	  : m_seq(seq)
	  , m_h(h)
	  , m_h2(h2)
	  , m_k(k)
	  , m_pos(pos)
	{
		// back to mask strings for the C ABI
		for (unsigned j = 0; j < h; ++j) {
			std::string s(k, '1');
			for (unsigned i : seed.at(j))
				if (i < k)
					s[i] = '0';
			m_masks.push_back(s);
		}
		m_strand.reset(new bool[(size_t)h * h2]());
		if (m_k > m_seq.length() || m_pos > m_seq.length() - m_k) {
			m_pos = npos();
			return;
		}
		seek();
	}

	const uint64_t* operator*() const { return &m_hashes[(m_pos - m_chunk0) * m_h * m_h2]; }
	const bool* strandArray() const { return m_strand.get(); }
	size_t pos() const { return m_pos; }
	bool operator==(const stHashIterator& it) const { return m_pos == it.m_pos; }
	bool operator!=(const stHashIterator& it) const { return !(*this == it); }
	stHashIterator& operator++()
	{
		++m_pos;
		seek();
		return *this;
	}
	static const stHashIterator end() { return stHashIterator(); }

  private:
	static size_t npos() { return std::numeric_limits<std::size_t>::max(); }
	static constexpr size_t kChunk = 1u << 18;

	void fill(size_t start)
	{
		const size_t n_win = m_seq.length() - m_k + 1;
		const size_t cnt = std::min(kChunk, n_win - start);
		const size_t bytes = cnt + m_k - 1;
		const unsigned m = m_h * m_h2;
		m_chunk0 = start;
		m_chunk_n = cnt;
		m_hashes.resize(bytes * m);
		m_valid.assign((bytes + 63) / 64, 0);
		m_stn.assign(bytes, 0);
		std::vector<const char*> ptrs;
		for (const std::string& s : m_masks)
			ptrs.push_back(s.c_str());
		btlbf_shim::check(btlbf_hash_seqs(m_k, m, ptrs.data(), m_h, m_h2, m_seq.data() + start, bytes,
		                                  nullptr, m_hashes.data(), m_valid.data(), m_stn.data(), BTLBF_HOST,
		                                  btlbf_shim::default_device(), BTLBF_STREAM_PER_THREAD));
	}

	void seek()
	{
		const size_t n_win = m_seq.length() - m_k + 1;
		while (m_pos < n_win) {
			if (m_chunk_n == 0 || m_pos < m_chunk0 || m_pos >= m_chunk0 + m_chunk_n)
				fill(m_pos);
			const size_t stop = m_chunk0 + m_chunk_n;
			while (m_pos < stop && !btlbf_shim::bit(m_valid.data(), m_pos - m_chunk0))
				++m_pos;
			if (m_pos < stop) {
				const uint64_t s = m_stn[m_pos - m_chunk0];
				for (unsigned i = 0; i < m_h * m_h2; ++i)
					m_strand[i] = (s >> i) & 1u;
				return;
			}
		}
		m_pos = npos();
	}

	std::string m_seq;
	std::vector<std::string> m_masks;
	unsigned m_h = 0, m_h2 = 0, m_k = 0;
	size_t m_pos;
	size_t m_chunk0 = 0, m_chunk_n = 0;
	std::vector<uint64_t> m_hashes, m_valid, m_stn;
	std::shared_ptr<bool[]> m_strand;
};

#endif

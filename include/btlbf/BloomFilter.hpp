// include/btlbf/BloomFilter.hpp -- drop-in for the reference's `BloomFilter` class
// (/root/reference/BloomFilter.hpp:40-446) whose bit array lives in MI355X HBM.
//
// Same class name, constructors and method names as the reference; every call forwards to the C
// ABI (include/btlbf.h), i.e. to the HIP kernels -- nothing is computed on the host.
//   * inserts are write-combined: hash rows (insert(hashes)), whole sequences (insertSeq) and raw k-mers
//     (KmerBloomFilter::insert(const char*)) are queued on the host and pushed to the GPU in one batch call
//     (btlbf_insert_hashes / one ragged btlbf_insert_seqs / btlbf_insert_kmers) when a queue fills or before
//     anything reads the filter.  Bit OR is order-free, so this is invisible to callers (reference:
//     BloomFilter.hpp:171-194, BloomFilterUtil.h:9-17).
//   * threads: as in the reference, insert / insertAndCheck / contains may be called on ONE filter from
//     many threads at once (Tests/AdHoc/ParallelFilter.cpp:104-122 does so under OpenMP; the reference
//     relies on byte atomics, BloomFilter.hpp:177,191,206-210).  Here the queue is striped -- a thread
//     appends to the stripe its id hashes to, under that stripe's mutex -- and the C ABI underneath
//     serialises calls on one filter with an internal lock.  storeFilter / loadFilter while other
//     threads insert is a caller race here as it is there.
//   * insertAndCheck() per k-mer costs one GPU round trip; so does contains(hashes) -- unless the hashes are an
//     ntHashIterator's (the reference's query loop `if (bloom.contains(*itr))`): then the first call answers every
//     row the iterator holds in one round trip and the following ones are read from that answer, for as long as
//     nothing was inserted in between (detail.hpp HashSpan / SpanCache; m_version below).  Use the *Seq/*Seqs
//     batch members (or BloomFilterUtil.h's insertSeq) on the fast path.
//   * not reproduced on purpose: the (expectedElemNum, fpr, ...) constructor, which in the reference
//     deletes an uninitialised pointer (BloomFilter.hpp:83-99 vs :396-397).
#ifndef BTLBF_BLOOMFILTER_HPP
#define BTLBF_BLOOMFILTER_HPP
#include "detail.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <fstream>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

class BloomFilter
{
  public:
	BloomFilter() = default;

	// BloomFilter(size_t filterSize, unsigned hashNum, unsigned kmerSize), BloomFilter.hpp:65-76
	BloomFilter(size_t filterSize, unsigned hashNum, unsigned kmerSize)
	{
		btlbf_shim::check(btlbf_create(&m_f, BTLBF_BLOOM, filterSize, hashNum, kmerSize, 0,
		                               btlbf_shim::default_device()));
	}

	// BloomFilter(size_t expectedElemNum, double fpr, unsigned hashNum, unsigned kmerSize), BloomFilter.hpp:83-99:
	// sized from the expected number of elements and the target FPR (hashNum 0 = the optimum for that FPR).
	// The reference's own version frees an uninitialised pointer on this path; the
	// arithmetic is the reference's (calcOptiHashNum :419, calcOptimalSize :406-413) and dFPR lands in the header.
	BloomFilter(size_t expectedElemNum, double fpr, unsigned hashNum, unsigned kmerSize)
	{
		const unsigned h = hashNum ? hashNum : calcOptiHashNum(fpr);
		btlbf_shim::check(btlbf_create(&m_f, BTLBF_BLOOM, calcOptimalSize(expectedElemNum, fpr, h), h, kmerSize, 0,
		                               btlbf_shim::default_device()));
		btlbf_set_dfpr(m_f, fpr);
	}
	static unsigned calcOptiHashNum(double fpr) { return unsigned(-std::log(fpr) / std::log(2)); }
	// only multiples of 64 (BloomFilter.hpp:401-413)
	static size_t calcOptimalSize(size_t entries, double fpr, unsigned hashNum)
	{
		const size_t v = size_t(-double(entries) * double(hashNum) / std::log(1.0 - std::pow(fpr, 1.0 / double(hashNum))));
		return v + (64 - v % 64);
	}

	// BloomFilter(const string& filterFilePath), BloomFilter.hpp:101-105
	explicit BloomFilter(const std::string& filterFilePath) { loadFilter(filterFilePath); }

	virtual ~BloomFilter() { btlbf_destroy(m_f); }

	void loadFilter(const std::string& filterFilePath) // BloomFilter.hpp:107-116
	{
		btlbf_destroy(m_f);
		m_f = nullptr;
		for (auto& st : m_stripes)
			st.clear();
		btlbf_shim::check(
		    btlbf_load(&m_f, BTLBF_BLOOM, filterFilePath.c_str(), 0, btlbf_shim::default_device()));
		touch();
	}

	// loadHeader(std::istream&), BloomFilter.hpp:118-166: consumes the header lines up to "[HeaderEnd]" and leaves a
	// zeroed filter of the header's geometry (nEntry / Entry / dFPR taken over); the stream is then positioned at
	// the body, which loadBody reads (the reference's loadFilter does the two in turn, :107-116)
	void loadHeader(std::istream& file)
	{
		std::string text, line;
		bool end = false;
		while (std::getline(file, line)) {
			text += line + "\n";
			if (line == "[HeaderEnd]") {
				end = true;
				break;
			}
			if (text.size() > (1u << 16))
				break;
		}
		(void)end; // a missing header end is reported by the C ABI in the reference's words
		btlbf_destroy(m_f);
		m_f = nullptr;
		for (auto& st : m_stripes)
			st.clear();
		btlbf_shim::check(btlbf_create_from_header(&m_f, BTLBF_BLOOM, text.data(), text.size(), 0,
		                                           btlbf_shim::default_device()));
		touch();
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
		touch();
	}
	double getDesiredFPR() const { return btlbf_get_dfpr(m_f); }

	// ---- per-k-mer interface (precomputed hash values, m_hashNum per k-mer) ----
	void insert(const uint64_t precomputed[]) // BloomFilter.hpp:185-194
	{
		const unsigned h = getHashNum();
		{
			Stripe& st = my_stripe();
			std::lock_guard<std::mutex> g(st.mu);
			st.rows.insert(st.rows.end(), precomputed, precomputed + h);
			// pushed under the stripe's lock: a flush() (= any reader) that finds the stripe empty must be able to
			// rely on its rows having reached the filter, not on their being on their way in another thread
			if (st.rows.size() >= kFlushRows * (size_t)h) {
				push(st.rows);
				st.rows.clear();
			}
			st.ver.store(st.ver.load(std::memory_order_relaxed) + 1, std::memory_order_release);
		}
	}
	void insert(std::vector<uint64_t> const& precomputed) // BloomFilter.hpp:171-180 (.at() range check)
	{
		(void)precomputed.at(getHashNum() - 1);
		insert(precomputed.data());
	}

	bool insertAndCheck(const uint64_t precomputed[]) // BloomFilter.hpp:200-214
	{
		flush();
		uint8_t out = 0;
		btlbf_shim::check(btlbf_insert_and_check_hashes(m_f, precomputed, 1, &out, BTLBF_ORDER_SERIAL,
		                                                BTLBF_HOST, nullptr));
		touch();
		return out != 0;
	}
	bool insertAndCheck(std::vector<uint64_t> const& precomputed) // BloomFilter.hpp:220-232
	{
		(void)precomputed.at(getHashNum() - 1);
		return insertAndCheck(precomputed.data());
	}

	bool contains(const uint64_t precomputed[]) const // BloomFilter.hpp:252-262
	{
		if (const uint8_t* hit = lookahead(precomputed))
			return *hit != 0;
		flush();
		uint8_t out = 0;
		btlbf_shim::check(btlbf_contains_hashes(m_f, precomputed, 1, &out, BTLBF_HOST, BTLBF_STREAM_PER_THREAD));
		return out != 0;
	}
	bool contains(std::vector<uint64_t> const& precomputed) const // BloomFilter.hpp:237-247
	{
		(void)precomputed.at(getHashNum() - 1);
		return contains(precomputed.data());
	}

	// ---- batch interface (the fast path; no counterpart in the reference) ----
	// n rows of m_hashNum hashes each
	void insertBatch(const uint64_t* rows, size_t n)
	{
		flush();
		btlbf_shim::check(btlbf_insert_hashes(m_f, rows, n, 0, BTLBF_ORDER_PARALLEL, BTLBF_HOST, nullptr));
		touch();
	}
	std::vector<uint8_t> containsBatch(const uint64_t* rows, size_t n) const
	{
		flush();
		std::vector<uint8_t> out(n ? n : 1);
		btlbf_shim::check(btlbf_contains_hashes(m_f, rows, n, out.data(), BTLBF_HOST, nullptr));
		out.resize(n);
		return out;
	}
	// every k-mer of `seq` (ntHashIterator semantics: windows with non-ACGT bytes are skipped): the loop of
	// BloomFilterUtil.h:9-17.  Queued like insert(hashes): sequences are appended to the calling thread's stripe
	// and reach the GPU as ONE ragged btlbf_insert_seqs batch (offsets in starts[]), not one launch each.
	void insertSeq(const std::string& seq) { insertSeq(seq.data(), seq.size()); }
	void insertSeq(const char* seq, size_t len)
	{
		if (len < getKmerSize())
			return; // no k-mer (ntHashIterator.hpp:61-64)
		{
			Stripe& st = my_stripe();
			std::lock_guard<std::mutex> g(st.mu);
			st.starts.push_back(st.seqs.size());
			st.seqs.append(seq, len);
			if (st.seqs.size() >= kFlushBases) {
				push_seqs(st);
				st.seqs.clear();
				st.starts.clear();
			}
			st.ver.store(st.ver.load(std::memory_order_relaxed) + 1, std::memory_order_release);
		}
	}
	// many sequences in one call (a batch of reads): result as if insertSeq were called on each
	void insertSeqs(const std::vector<std::string>& seqs)
	{
		for (const auto& s : seqs)
			insertSeq(s);
	}
	// back-to-back reads of exactly readLen bytes in one host buffer: one call, one copy, the partitioned pipeline
	// when the batch is large enough (the fast path for callers that hold their reads in memory)
	void insertReads(const char* reads, size_t len, unsigned readLen)
	{
		flush();
		btlbf_layout lay;
		lay.starts = nullptr;
		lay.n_seqs = 0;
		lay.read_len = readLen;
		btlbf_shim::check(btlbf_insert_seqs(m_f, reads, len, &lay, 0, BTLBF_ORDER_PARALLEL, BTLBF_HOST, nullptr));
		touch();
	}
	// how many k-mers of `seq` (of the reads) contains() finds; *clean (optional) = the k-mers there are
	uint64_t countSeq(const std::string& seq, uint64_t* clean = nullptr) const
	{
		flush();
		uint64_t counts[2] = {0, 0};
		btlbf_shim::check(btlbf_contains_seqs(m_f, seq.data(), seq.size(), nullptr, nullptr, nullptr, counts, BTLBF_HOST,
		                                      BTLBF_STREAM_PER_THREAD));
		if (clean)
			*clean = counts[0];
		return counts[1];
	}
	uint64_t countReads(const char* reads, size_t len, unsigned readLen, uint64_t* clean = nullptr) const
	{
		flush();
		btlbf_layout lay;
		lay.starts = nullptr;
		lay.n_seqs = 0;
		lay.read_len = readLen;
		uint64_t counts[2] = {0, 0};
		btlbf_shim::check(btlbf_contains_seqs(m_f, reads, len, &lay, nullptr, nullptr, counts, BTLBF_HOST, nullptr));
		if (clean)
			*clean = counts[0];
		return counts[1];
	}
	// every k-mer of a FASTA / FASTQ / one-sequence-per-line file, gzip or plain (the job of the
	// reference's loaders Tests/AdHoc/ParallelFilter.cpp:104-122 and swig/writeBloom_rolling.cpp:18-59);
	// perLine: every sequence line is its own sequence instead of one sequence per FASTA record
	btlbf_fastx_stats insertFile(const std::string& path, bool perLine = false, uint64_t batchBytes = 0)
	{
		flush();
		btlbf_fastx_stats st;
		btlbf_shim::check(btlbf_insert_fastx(m_f, path.c_str(), perLine ? BTLBF_FASTX_LINES : BTLBF_FASTX_RECORDS,
		                                     batchBytes, &st));
		touch();
		return st;
	}
	// st.n_windows clean k-mers of the file, st.n_hits of them in the filter
	btlbf_fastx_stats containsFile(const std::string& path, bool perLine = false, uint64_t batchBytes = 0) const
	{
		flush();
		btlbf_fastx_stats st;
		btlbf_shim::check(btlbf_contains_fastx(m_f, path.c_str(), perLine ? BTLBF_FASTX_LINES : BTLBF_FASTX_RECORDS,
		                                       batchBytes, &st));
		return st;
	}
	// contains() of every window: result[p] for window start p; valid[p] = window was a clean k-mer
	void containsSeq(const std::string& seq, std::vector<bool>& result, std::vector<bool>& valid) const
	{
		flush();
		const size_t nw = (seq.size() + 63) / 64;
		std::vector<uint64_t> hb(nw ? nw : 1), vb(nw ? nw : 1);
		btlbf_shim::check(btlbf_contains_seqs(m_f, seq.data(), seq.size(), nullptr, hb.data(), vb.data(),
		                                      nullptr, BTLBF_HOST, BTLBF_STREAM_PER_THREAD));
		result.assign(seq.size(), false);
		valid.assign(seq.size(), false);
		for (size_t p = 0; p < seq.size(); ++p) {
			result[p] = btlbf_shim::bit(hb.data(), p);
			valid[p] = btlbf_shim::bit(vb.data(), p);
		}
	}

	// ---- persistence ----
	void storeFilter(const std::string& filterFilePath) const // BloomFilter.hpp:304-314
	{
		flush();
		std::cerr << "Writing a " << sizeInBytes() << " byte filter to " << filterFilePath
		          << " on disk.\n";
		btlbf_shim::check(btlbf_store(m_f, filterFilePath.c_str()));
	}
	void writeHeader(std::ostream& out) const // BloomFilter.hpp:264-288
	{
		char buf[1024];
		size_t n = 0;
		btlbf_shim::check(btlbf_header(m_f, buf, sizeof buf, &n));
		out.write(buf, (std::streamsize)n);
	}
	friend std::ostream& operator<<(std::ostream& out, const BloomFilter& bloom) // BloomFilter.hpp:291-297
	{
		bloom.flush();
		bloom.writeHeader(out);
		std::vector<char> chunk(1u << 24);
		const uint64_t total = bloom.sizeInBytes();
		for (uint64_t off = 0; off < total; off += chunk.size()) {
			const uint64_t n = std::min<uint64_t>(chunk.size(), total - off);
			btlbf_shim::check(btlbf_download(bloom.m_f, chunk.data(), off, n));
			out.write(chunk.data(), (std::streamsize)n);
		}
		return out;
	}

	// ---- statistics and attributes ----
	uint64_t getPop() const // BloomFilter.hpp:316-323
	{
		flush();
		uint64_t v = 0;
		btlbf_shim::check(btlbf_popcount(m_f, &v));
		return v;
	}
	double getFPR() // BloomFilter.hpp:346-350
	{
		m_FPR = std::pow(double(getPop()) / double(getFilterSize()), double(getHashNum()));
		return m_FPR;
	}
	double getFPRPrecompute() const { return m_FPR; }
	double getFPR_numEle() const // BloomFilter.hpp:363-367,423-427
	{
		const double m = double(getFilterSize()), h = double(getHashNum());
		return std::pow(1.0 - std::pow(1.0 - 1.0 / m, double(btlbf_get_n_entry(m_f)) * h), h);
	}
	// false-positive rate of calling a redundant entry unique (BloomFilter.hpp:333-341; host arithmetic
	// over nEntry terms exactly as in the reference)
	double getRedudancyFPR()
	{
		const double m = double(getFilterSize()), h = double(getHashNum());
		const uint64_t n = btlbf_get_n_entry(m_f);
		auto fpr = [&](double e) { return std::pow(1.0 - std::pow(1.0 - 1.0 / m, e * h), h); };
		double total = std::log(fpr(1));
		for (uint64_t i = 2; i < n; ++i)
			total = std::log(std::exp(total) + fpr(double(i)));
		return std::exp(total) / double(n);
	}
	unsigned getHashNum() const { return btlbf_hash_num(m_f); }
	unsigned getKmerSize() const { return btlbf_kmer_size(m_f); }
	uint64_t getFilterSize() const { return btlbf_size(m_f); }
	uint64_t sizeInBytes() const { return btlbf_size_bytes(m_f); }
	uint64_t getnEntry() { return btlbf_get_n_entry(m_f); }
	uint64_t gettEntry() { return btlbf_get_t_entry(m_f); }
	void setnEntry(uint64_t value) { btlbf_set_n_entry(m_f, value); }
	void settEntry(uint64_t value) { btlbf_set_t_entry(m_f, value); }

	// the C-ABI handle, for callers that want the raw batch entry points (what they do with it is not seen here:
	// contains() stops trusting its look-ahead answers from now on)
	btlbf_filter* handle() const
	{
		m_external.store(true, std::memory_order_release);
		flush();
		return m_f;
	}

  protected:
	BloomFilter(const BloomFilter&) = delete; // BloomFilter.hpp:384
	BloomFilter& operator=(const BloomFilter&) = delete;

	struct Stripe {
		std::mutex mu;
		std::atomic<uint64_t> ver{ 0 }; // bumped (under mu) with every row queued here: see version()
		std::vector<uint64_t> rows;   // hash rows, m_hashNum each
		std::string seqs;             // whole sequences, back to back ...
		std::vector<uint64_t> starts; // ... and where each begins (btlbf_layout::starts without its end sentinel)
		std::string kmers;            // raw k-mers, m_kmerSize bytes each (KmerBloomFilter)
		void clear()
		{
			rows.clear();
			seqs.clear();
			starts.clear();
			kmers.clear();
		}
	};
	// A mutation has been queued or applied: look-ahead answers taken before this moment are void.  The version is
	// bumped AFTER the rows are in a queue (or in the filter) and before the mutating member returns: whoever starts
	// a contains() after that return finds a new version, refreshes its answers and flush() hands it the rows.
	// The per-k-mer inserts count on their own stripe (one shared counter bounced between the cores cost the
	// 16-thread insert loop two thirds of its rate); everything else counts here.
	void touch() const { m_version.fetch_add(1, std::memory_order_acq_rel); }
	uint64_t version() const
	{
		uint64_t v = m_version.load(std::memory_order_acquire);
		for (const auto& st : m_stripes)
			v += st.ver.load(std::memory_order_acquire);
		return v;
	}
	// contains(p) for p inside the rows an ntHashIterator of this thread has announced (detail.hpp): the answers for
	// a window of rows from p on are fetched with one call and kept while filter and rows stay the same.
	// nullptr = not applicable.
	const uint8_t* lookahead(const uint64_t* p) const
	{
		const btlbf_shim::HashSpan& sp = btlbf_shim::tls_span();
		if (!sp.base || !sp.alive || m_external.load(std::memory_order_acquire))
			return nullptr;
		const uintptr_t a = reinterpret_cast<uintptr_t>(p), b = reinterpret_cast<uintptr_t>(sp.base);
		const size_t row_bytes = (size_t)sp.stride * sizeof(uint64_t);
		if (a < b || a >= b + sp.rows * row_bytes || (a - b) % row_bytes)
			return nullptr;
		if (!sp.alive->load(std::memory_order_acquire) || sp.stride != getHashNum())
			return nullptr;
		btlbf_shim::SpanCache::Entry& c = btlbf_shim::tls_cache().find(this);
		if (c.off_span == sp.id)
			return nullptr; // given up on this span (see below): the one-row path
		const size_t row = (a - b) / row_bytes;
		const uint64_t v = version();
		if (c.span_id == sp.id && c.version == v && row >= c.first && row - c.first < c.hit.size()) {
			++c.served;
			return &c.hit[row - c.first];
		}
		// refresh: a window of rows from this one on.  A loop that mutates the filter between its contains() calls
		// (`if (!bf.contains(*itr)) bf.insert(*itr)`) voids every window after one answer: three of those in a row and
		// the rest of the span is answered row by row, which is what such a loop costs anyway
		if (c.span_id == sp.id && c.version != v && c.served <= 1) {
			if (++c.streak >= btlbf_shim::SpanCache::kGiveUp) {
				c.off_span = sp.id;
				c.streak = 0;
				return nullptr;
			}
		} else {
			c.streak = 0;
		}
		flush();
		const size_t n = std::min(btlbf_shim::SpanCache::kWindow, sp.rows - row);
		c.hit.resize(n);
		btlbf_shim::check(btlbf_contains_hashes(m_f, sp.base + row * sp.stride, n, c.hit.data(), BTLBF_HOST, BTLBF_STREAM_PER_THREAD));
		c.first = row;
		c.span_id = sp.id;
		c.version = v; // (read before the flush: a mutation that raced with this call voids the answers again)
		c.served = 1;
		return &c.hit[0];
	}
	Stripe& my_stripe() const
	{
		return m_stripes[std::hash<std::thread::id>()(std::this_thread::get_id()) % kStripes];
	}
	void push(const std::vector<uint64_t>& rows) const
	{
		btlbf_shim::check(btlbf_insert_hashes(m_f, rows.data(), rows.size() / getHashNum(), 0, BTLBF_ORDER_PARALLEL,
		                                      BTLBF_HOST, nullptr));
	}
	void push_seqs(Stripe& st) const
	{
		st.starts.push_back(st.seqs.size()); // end sentinel
		btlbf_layout lay;
		lay.starts = st.starts.data();
		lay.n_seqs = st.starts.size() - 1;
		lay.read_len = 0;
		btlbf_shim::check(btlbf_insert_seqs(m_f, st.seqs.data(), st.seqs.size(), &lay, 0, BTLBF_ORDER_PARALLEL,
		                                    BTLBF_HOST, nullptr));
	}
	void push_kmers(const std::string& kmers) const
	{
		btlbf_shim::check(btlbf_insert_kmers(m_f, kmers.data(), kmers.size() / getKmerSize(), 0, BTLBF_ORDER_PARALLEL,
		                                     BTLBF_HOST, nullptr));
	}
	// everything queued by any thread so far reaches the filter
	void flush() const
	{
		for (auto& st : m_stripes) {
			std::lock_guard<std::mutex> g(st.mu);
			if (!st.rows.empty())
				push(st.rows);
			if (!st.seqs.empty())
				push_seqs(st);
			if (!st.kmers.empty())
				push_kmers(st.kmers);
			st.clear();
		}
	}

	static constexpr size_t kFlushRows = 1u << 16;
	static constexpr size_t kFlushBases = 32u << 20; // bases queued per stripe before a push
	static constexpr size_t kFlushKmers = 1u << 16;
	static constexpr size_t kStripes = 16;
	btlbf_filter* m_f = nullptr;
	mutable Stripe m_stripes[kStripes];
	// (starts at a value no other filter object of this process has: an answer kept for a filter that is gone cannot
	// be mistaken for one of a new filter at the same address)
	mutable std::atomic<uint64_t> m_version{ btlbf_shim::next_span_id() << 40 };
	mutable std::atomic<bool> m_external{ false };
	double m_FPR = 0;
};

#endif

// tests/cpp/test_shims.cpp -- the reference's unit-test scenarios (Tests/Unit/BloomFilterTests.cpp:69-139,
// Tests/Unit/CountingBloomFilterTests.cpp:70-244), written against the drop-in C++ shims in
// include/btlbf/.  A user of the reference changes only the include paths.  Needs a GPU.
#include "btlbf/BloomFilter.hpp"
#include "btlbf/BloomFilterUtil.h"
#include "btlbf/CountingBloomFilter.hpp"
#include "btlbf/KmerBloomFilter.hpp"
#include "btlbf/ntHashIterator.hpp"
#include "btlbf/stHashIterator.hpp"

#include <atomic>
#include <omp.h>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

static int g_fail = 0;
#define CHECK(cond)                                                                    \
	do {                                                                               \
		if (!(cond)) {                                                                 \
			std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
			++g_fail;                                                                  \
		}                                                                              \
	} while (0)

static std::string tmp_name(const char* tag)
{
	char buf[256];
	std::snprintf(buf, sizeof buf, "/tmp/btlbf_shim_%s_%d.bf", tag, (int)getpid());
	return buf;
}

static std::string slurp(const std::string& p)
{
	std::ifstream f(p, std::ios::binary);
	std::stringstream ss;
	ss << f.rdbuf();
	return ss.str();
}

// returns offset of the first byte after the "[HeaderEnd]" line, 0 if absent
static size_t header_len(const std::string& path)
{
	std::ifstream f(path);
	std::string line;
	while (std::getline(f, line))
		if (line == "[HeaderEnd]")
			return (size_t)f.tellg();
	return 0;
}

static void bloom_basic()
{
	const size_t filterSize = 1000000000;
	const unsigned numHashes = 5, k = 4;
	const char* seq = "ACGTAC";
	BloomFilter filter(filterSize, numHashes, k);
	ntHashIterator insertIt(seq, numHashes, k);
	unsigned n = 0;
	while (insertIt != insertIt.end()) {
		filter.insert(*insertIt);
		++insertIt;
		++n;
	}
	CHECK(n == 3);
	ntHashIterator queryIt(seq, numHashes, k);
	while (queryIt != queryIt.end()) {
		CHECK(filter.contains(*queryIt));
		++queryIt;
	}
	const std::string fn = tmp_name("bloom");
	filter.storeFilter(fn);
	const size_t hl = header_len(fn);
	CHECK(hl > 0);
	BloomFilter filter2(fn);
	CHECK(slurp(fn).size() - hl == filter2.sizeInBytes());
	CHECK(filter2.getFilterSize() == filterSize && filter2.getHashNum() == numHashes && filter2.getKmerSize() == k);
	ntHashIterator q2(seq, numHashes, k);
	while (q2 != q2.end()) {
		CHECK(filter2.contains(*q2));
		++q2;
	}
	CHECK(filter2.getPop() == filter.getPop());
	CHECK(filter.getPop() <= 15 && filter.getPop() >= 10);
	std::remove(fn.c_str());
}

static void hash_known_answers()
{
	// values observed from the reference
	ntHashIterator it("ACGTAC", 5, 4);
	CHECK(it.pos() == 0);
	const uint64_t exp0[5] = {0x4b21efd76bfc8c8aULL, 0x6ab8d13c740e89beULL, 0xb5dac11e34491d35ULL,
	                          0x00fcb0dfe49bd351ULL, 0x4c1ea0bee4de43d4ULL};
	for (int i = 0; i < 5; ++i)
		CHECK((*it)[i] == exp0[i]);
	++it;
	CHECK(it.pos() == 1 && (*it)[0] == 0x62779f381e5f5a2dULL);
	++it;
	CHECK(it.pos() == 2 && (*it)[0] == 0xec40e7b3741c2bddULL);
	++it;
	CHECK(it == ntHashIterator::end());
	// N handling and a start offset
	ntHashIterator jt("ACGTNACGTAC", 2, 4, 1);
	CHECK(jt.pos() == 5);
	ntHashIterator kt("ACG", 1, 4);
	CHECK(kt == ntHashIterator::end());
	// spaced seeds, toy case of SURVEY.md 8c
	std::vector<std::string> seeds = {"1110111", "1011101"};
	stHashIterator st("ACGTNACGTACGTACGGTCA", stHashIterator::parseSeed(seeds), 2, 2, 7);
	CHECK(st.pos() == 5);
	const uint64_t exps[4] = {0x589d161199065de2ULL, 0xda4f3c83d59781ddULL, 0xb96a9e21deae3f34ULL,
	                          0x5ae90c2e2b2dbc02ULL};
	const bool expst[4] = {true, true, false, false};
	for (int i = 0; i < 4; ++i) {
		CHECK((*st)[i] == exps[i]);
		CHECK(st.strandArray()[i] == expst[i]);
	}
}

static void counting_basic()
{
	const size_t sizeInBytes = 100001;
	const unsigned numHashes = 5, k = 8, threshold = 1;
	const char* seq = "ACGTACACTGGACTGAGTCT";
	CountingBloomFilter<uint8_t> filter(sizeInBytes, numHashes, k, threshold);
	CHECK(filter.sizeInBytes() == 100008 && filter.size() == 100008);
	ntHashIterator insertIt(seq, numHashes, k);
	while (insertIt != insertIt.end()) {
		filter.insert(*insertIt);
		++insertIt;
	}
	ntHashIterator queryIt(seq, numHashes, k);
	unsigned n = 0;
	while (queryIt != queryIt.end()) {
		CHECK(filter.contains(*queryIt));
		CHECK(filter.minCount(*queryIt) >= 1);
		++queryIt;
		++n;
	}
	CHECK(n == 13);
	// a different sequence is (with overwhelming probability) absent
	const std::string other = "TTGACCAGTTACGGATCCAGTAGGCATTAGCCATGATCGGGATACCATGAACTTGACTGA";
	ntHashIterator o(other, numHashes, k);
	while (o != o.end()) {
		CHECK(!filter.contains(*o));
		++o;
	}
	// incrementAll / insertAndCheck / operator[]
	std::vector<uint64_t> hv = {1, 100009, 2, 3, 4}; // positions 1,1,2,3,4 modulo 100008
	filter.incrementAll(hv);
	CHECK(filter[1] >= 2 && filter[2] >= 1);
	CHECK(filter.insertAndCheck(hv));
	const std::string fn = tmp_name("cbf");
	filter.storeFilter(fn);
	const size_t hl = header_len(fn);
	CHECK(hl > 0 && slurp(fn).size() - hl == filter.sizeInBytes());
	CountingBloomFilter<uint8_t> filter2(fn, threshold);
	CHECK(filter2.size() * sizeof(uint8_t) == filter2.sizeInBytes());
	CHECK(filter2.popCount() == filter.popCount() && filter2.filtered_popcount() == filter.filtered_popcount());
	ntHashIterator q2(seq, numHashes, k);
	while (q2 != q2.end()) {
		CHECK(filter2.contains(*q2));
		++q2;
	}
	std::remove(fn.c_str());
}

static void fused_path_equals_iterator_path(const char* golden_dir)
{
	const std::string seq = "GATTACAGATTACANNGATTACACCCGGGTTTAAACGTACGTTGCAAGCTTAGGCTAACGTAGCTAGCTAGGATCCGAT"
	                        "CGATCGGGATATATCGCGCTAGCTAGCATCGATCGTAGCTAGTCGATCGATGCTAGCTAGCTAGCTAGCATGCATGCA";
	const unsigned h = 4, k = 31;
	KmerBloomFilter a(1 << 16, h, k), b(1 << 16, h, k), c(1 << 16, h, k);
	insertSeq(a, seq, h, k); // one fused launch
	ntHashIterator it(seq, h, k);
	while (it != it.end()) { // literal reference loop
		b.insert(*it);
		++it;
	}
	for (size_t p = 0; p + k <= seq.size(); ++p) { // raw k-mer strings
		const std::string km = seq.substr(p, k);
		if (km.find('N') == std::string::npos)
			c.insert(km.c_str());
	}
	const std::string fa = tmp_name("fa"), fb = tmp_name("fb"), fc = tmp_name("fc");
	a.storeFilter(fa);
	b.storeFilter(fb);
	c.storeFilter(fc);
	CHECK(slurp(fa) == slurp(fb));
	CHECK(slurp(fa) == slurp(fc));
	CHECK(a.contains(seq.substr(40, k).c_str()));
	std::vector<bool> res, valid;
	a.containsSeq(seq, res, valid);
	size_t nv = 0;
	for (size_t p = 0; p < seq.size(); ++p) {
		CHECK(res[p] == valid[p]);
		nv += valid[p];
	}
	CHECK(nv > 0 && nv < seq.size() - k + 1); // the N run removes some windows
	// operator<< writes the same bytes as storeFilter
	std::ostringstream os;
	os << a;
	CHECK(os.str() == slurp(fa));
	std::remove(fa.c_str());
	std::remove(fb.c_str());
	std::remove(fc.c_str());
	// a file written by the genuine reference round-trips byte for byte
	if (golden_dir) {
		const std::string g = std::string(golden_dir) + "/bf_1000_k25_h3_entries.bf";
		BloomFilter f(g);
		CHECK(f.getnEntry() == 17 && f.gettEntry() == 123456789012ULL);
		const std::string out = tmp_name("golden");
		f.storeFilter(out);
		CHECK(slurp(out) == slurp(g));
		std::remove(out.c_str());
	}
}

// file ingestion: the two reference loaders' semantics (one sequence per FASTA record / per line)
static void file_loaders_equal_insert_seq()
{
	const std::string s1 = "GATTACAGATTACANNGATTACACCCGGGTTTAAACGTACGTTGCAAGCTTAGGCTAACGTAGCTAGCTAGGATCCGATC";
	const std::string s2 = "ACGTTGCATGCATGCAAACCGGTTACGATCGATCGATTAGCTAGCTAGGCTAGCTA";
	const unsigned h = 3, k = 21;
	const std::string fa = tmp_name("fasta");
	{
		std::ofstream o(fa.c_str());
		o << ">one\n" << s1.substr(0, 40) << "\n" << s1.substr(40) << "\n>two\n" << s2 << "\n";
	}
	BloomFilter rec(1 << 14, h, k), lin(1 << 14, h, k), a(1 << 14, h, k), b(1 << 14, h, k);
	const btlbf_fastx_stats st = rec.insertFile(fa);
	CHECK(st.n_records == 2 && st.n_bases == s1.size() + s2.size());
	lin.insertFile(fa, true);
	insertSeq(a, s1, h, k); // contigsToBloom: lines of a record concatenated
	insertSeq(a, s2, h, k);
	insertSeq(b, s1.substr(0, 40), h, k); // loadBf: every line by itself
	insertSeq(b, s1.substr(40), h, k);
	insertSeq(b, s2, h, k);
	const std::string f1 = tmp_name("r"), f2 = tmp_name("a"), f3 = tmp_name("l"), f4 = tmp_name("b");
	rec.storeFilter(f1);
	a.storeFilter(f2);
	lin.storeFilter(f3);
	b.storeFilter(f4);
	CHECK(slurp(f1) == slurp(f2));
	CHECK(slurp(f3) == slurp(f4));
	CHECK(slurp(f1) != slurp(f3));
	const btlbf_fastx_stats q = rec.containsFile(fa);
	CHECK(q.n_windows > 0 && q.n_hits == q.n_windows);
	std::remove(fa.c_str());
	std::remove(f1.c_str());
	std::remove(f2.c_str());
	std::remove(f3.c_str());
	std::remove(f4.c_str());
}

// the FPR-sizing constructor (BloomFilter.hpp:83-99; it crashes in the reference) and the public
// loadHeader(std::istream&) (BloomFilter.hpp:118-166)
static void sizing_ctor_and_load_header()
{
	BloomFilter f(100000, 0.01, 0, 21);
	CHECK(f.getHashNum() == 6);                      // unsigned(-log(0.01) / log(2))
	CHECK(f.getFilterSize() % 64 == 0 && f.getFilterSize() == BloomFilter::calcOptimalSize(100000, 0.01, 6));
	CHECK(f.getDesiredFPR() == 0.01);
	insertSeq(f, "ACGTTGCATGCATGCAAACCGGTTACGATCGATCGATTAGCTAGCTAGGCTAGCTA", 6, 21);
	f.setnEntry(36);
	const std::string fn = tmp_name("sizing");
	f.storeFilter(fn);
	const std::string raw = slurp(fn);
	CHECK(raw.find("\tdFPR = 0.010000000000000000\n") != std::string::npos); // cpptoml: showpoint, 17 significant digits
	std::ifstream in(fn, std::ios::binary);
	BloomFilter g;
	g.loadHeader(in);
	CHECK(g.getFilterSize() == f.getFilterSize() && g.getHashNum() == 6 && g.getKmerSize() == 21);
	CHECK(g.getnEntry() == 36 && g.getDesiredFPR() == 0.01 && g.getPop() == 0);
	g.loadBody(in);
	CHECK(g.getPop() == f.getPop() && g.getPop() > 0);
	const std::string fn2 = tmp_name("sizing2");
	g.storeFilter(fn2);
	CHECK(slurp(fn2) == raw);
	std::remove(fn.c_str());
	std::remove(fn2.c_str());
	// the counting filter's loadHeader (CountingBloomFilter.hpp:84)
	CountingBloomFilter<uint8_t> c(4096, 3, 21, 1);
	ntHashIterator it("ACGTTGCATGCATGCAAACCGGTTACGATCGATCGATTAGC", 3, 21);
	while (it != it.end()) {
		c.insert(*it);
		++it;
	}
	const std::string fc = tmp_name("cbfhdr");
	c.storeFilter(fc);
	std::ifstream cin2(fc, std::ios::binary);
	CountingBloomFilter<uint8_t> d2(8, 1, 1, 1);
	d2.loadHeader(cin2);
	CHECK(d2.size() == c.size() && d2.getHashNum() == 3 && d2.getKmerSize() == 21 && d2.popCount() == 0);
	d2.loadBody(cin2);
	CHECK(d2.popCount() == c.popCount() && d2.popCount() > 0);
	std::remove(fc.c_str());
}

// synthetic reads of SURVEY.md 8d (the generator behind tests/golden/digests.json)
static std::string synth_read(uint64_t seed, uint64_t r, unsigned len)
{
	auto w = [&](uint64_t n) {
		uint64_t z = seed + (n + 1) * 0x9E3779B97F4A7C15ULL;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
		return z ^ (z >> 31);
	};
	const unsigned wpr = (len + 31) / 32;
	std::string s(len, 'A');
	for (unsigned j = 0; j < len; ++j)
		s[j] = "ACGT"[(w(r * wpr + j / 32) >> (2 * (j % 32))) & 3];
	return s;
}

// The reference's own multi-threaded harness, Tests/AdHoc/ParallelFilter.cpp:104-122: OpenMP threads pull
// sequences from one stream inside a critical section and each runs the ntHashIterator + bloom.insert(*itr)
// loop on ONE shared filter; here also with concurrent contains() while the others insert.  The filter is
// golden fixture bf_small (tests/golden/digests.json): written to `out_path`, the Python wrapper compares
// its body digest with the reference's.
static void parallel_filter_replay(const char* out_path)
{
	const unsigned n_reads = 20000, L = 150, h = 4, k = 31;
	BloomFilter bloom(1 << 24, h, k);
	unsigned next = 0;
	std::atomic<unsigned long> inserted(0), found(0), asked(0);
#pragma omp parallel
	{
		for (;;) {
			std::string seq;
			bool good = false;
#pragma omp critical(uFile)
			{
				if (next < n_reads) {
					seq = synth_read(42, next++, L);
					good = true;
				}
			}
			if (!good)
				break;
			ntHashIterator itr(seq, h, k);
			unsigned long n = 0;
			while (itr != itr.end()) {
				bloom.insert(*itr);
				++itr;
				++n;
			}
			inserted += n;
			if (next % 64 == 0) { // a reader among the writers: this thread's own k-mers must be there
				ntHashIterator q(seq, h, k);
				while (q != q.end()) {
					found += bloom.contains(*q);
					++asked;
					++q;
				}
			}
		}
	}
	CHECK(inserted == (unsigned long)n_reads * (L - k + 1));
	CHECK(asked > 0 && found == asked);
	CHECK(bloom.getPop() == 7310180); // digests.json: bf_small.pop (the reference's count)
	if (out_path)
		bloom.storeFilter(out_path);
	// counting filter: increments from many threads are serialised, none is lost (incrementAll is exact)
	CountingBloomFilter<uint8_t> cbf(1 << 16, 3, 25, 2);
	const std::string s = synth_read(7, 0, 80);
#pragma omp parallel for
	for (int t = 0; t < 64; ++t) {
		ntHashIterator it(s, 3, 25);
		while (it != it.end()) {
			cbf.incrementAll(*it);
			++it;
		}
	}
	ntHashIterator c(s, 3, 25);
	while (c != c.end()) {
		CHECK(cbf.minCount(*c) >= 64);
		++c;
	}
}

// swig/test.pl:8-46 in C++: the SWIG module's BloomFilter is KmerBloomFilter (swig/BloomFilter.i:17-59).  k = 20, so
// insert(const char*) / contains(const char*) go through the reference's k % 4 == 0 table walk; the stored file is
// compared with the reference's by the Python wrapper (tests/golden/kmer_path.json: file_sha256).
static void swig_test_pl_replay(const char* out_path)
{
	KmerBloomFilter filter(1000000000, 5, 20);
	filter.insert("ATCGGGTCATCAACCAATAT");
	filter.insert("ATCGGGTCATCAACCAATAC");
	filter.insert("ATCGGGTCATCAACCAATAG");
	filter.insert("ATCGGGTCATCAACCAATAA");
	CHECK(filter.contains("ATCGGGTCATCAACCAATAT") && filter.contains("ATCGGGTCATCAACCAATAC") &&
	      filter.contains("ATCGGGTCATCAACCAATAG") && filter.contains("ATCGGGTCATCAACCAATAA"));
	CHECK(!filter.contains("ATCGGGTCATCAACCAATTA") && !filter.contains("ATCGGGTCATCAACCAATTC"));
	CHECK(filter.getPop() == 20);
	const std::string fn = out_path ? std::string(out_path) : tmp_name("swig");
	filter.storeFilter(fn);
	KmerBloomFilter filter2(fn);
	CHECK(filter2.contains("ATCGGGTCATCAACCAATAT") && filter2.contains("ATCGGGTCATCAACCAATAA"));
	CHECK(!filter2.contains("ATCGGGTCATCAACCAATTA"));
	CHECK(filter2.getPop() == 20 && filter2.getHashNum() == 5 && filter2.getKmerSize() == 20 &&
	      filter2.getFilterSize() == 1000000000);
	const std::vector<uint8_t> r = filter2.containsKmers("ATCGGGTCATCAACCAATATATCGGGTCATCAACCAATTA", 2);
	CHECK(r.size() == 2 && r[0] == 1 && r[1] == 0);
	if (!out_path)
		std::remove(fn.c_str());
	// swig/test.pl:59-84
	const std::string str = "TAGAATCACCCAAAGA";
	KmerBloomFilter bloom(10000, 4, 5);
	insertSeq(bloom, str, 4, 5);
	for (size_t i = 0; i + 5 <= str.size(); ++i)
		CHECK(bloom.contains(str.substr(i, 5).c_str()));
}

// insertSeq is queued (include/btlbf/BloomFilter.hpp): many short sequences from many threads reach the GPU as a
// few ragged batches and give the filter the per-sequence loop gives (BloomFilterUtil.h:9-17)
static void insert_seq_is_write_combined()
{
	const unsigned n_reads = 5000, L = 100, h = 3, k = 25;
	BloomFilter queued(1 << 22, h, k), loop(1 << 22, h, k);
#pragma omp parallel for
	for (int r = 0; r < (int)n_reads; ++r) {
		std::string s = synth_read(9, (uint64_t)r, L);
		if (r % 7 == 0)
			s[r % L] = 'N';
		if (r % 11 == 0)
			s = s.substr(0, r % 30); // shorter than k now and then, also empty
		insertSeq(queued, s, h, k);
	}
	for (unsigned r = 0; r < n_reads; ++r) {
		std::string s = synth_read(9, r, L);
		if (r % 7 == 0)
			s[r % L] = 'N';
		if (r % 11 == 0)
			s = s.substr(0, r % 30);
		ntHashIterator itr(s, h, k);
		while (itr != itr.end()) {
			loop.insert(*itr);
			++itr;
		}
	}
	CHECK(queued.getPop() == loop.getPop() && queued.getPop() > 0);
	const std::string fa = tmp_name("q"), fb = tmp_name("l");
	queued.storeFilter(fa);
	loop.storeFilter(fb);
	CHECK(slurp(fa) == slurp(fb));
	std::remove(fa.c_str());
	std::remove(fb.c_str());
}

// The reference's query loop `if (bloom.contains(*itr))` through the look-ahead of BloomFilter::contains
// (include/btlbf/detail.hpp): the answers must be the ones a round trip per k-mer gives, whatever happens between the
// calls -- inserts on this thread and on others, a second filter, copies of the iterator, iterators that are gone.
static void contains_lookahead_is_invisible()
{
	const unsigned L = 150, h = 4, k = 31;
	BloomFilter bloom(1 << 26, h, k), other(1 << 26, h, k);
	std::vector<std::string> reads;
	for (unsigned r = 0; r < 400; ++r)
		reads.push_back(synth_read(5, r, L));
	reads[3][70] = 'N';
	for (unsigned r = 0; r < 400; r += 2)
		insertSeq(bloom, reads[r], h, k); // even reads are in, odd ones are not
	insertSeq(other, reads[1], h, k);
	// 1. plain loop: equals the batch answer
	for (unsigned r = 0; r < 8; ++r) {
		std::vector<bool> hit, valid;
		bloom.containsSeq(reads[r], hit, valid);
		ntHashIterator itr(reads[r], h, k);
		size_t seen = 0;
		while (itr != itr.end()) {
			CHECK(valid[itr.pos()]);
			CHECK(bloom.contains(*itr) == hit[itr.pos()]);
			CHECK(bloom.contains(*itr) == (r % 2 == 0)); // (no false positive in a filter this empty)
			++seen;
			++itr;
		}
		CHECK(seen == (r == 3 ? L - k + 1 - k : L - k + 1));
	}
	// 2. two filters asked by turns about the same rows
	{
		ntHashIterator itr(reads[1], h, k);
		while (itr != itr.end()) {
			CHECK(!bloom.contains(*itr));
			CHECK(other.contains(*itr));
			++itr;
		}
	}
	// 3. "insert what is not there yet" (the loop of a de-duplicating loader): every answer reflects the inserts so far
	{
		const std::string twice = reads[5].substr(0, 100) + reads[5].substr(0, 100);
		ntHashIterator itr(twice, h, k);
		size_t fresh = 0, again = 0;
		while (itr != itr.end()) {
			if (!bloom.contains(*itr)) {
				bloom.insert(*itr);
				CHECK(bloom.contains(*itr));
				++fresh;
			} else {
				++again;
			}
			++itr;
		}
		CHECK(fresh == 100 - k + 1 + (k - 1)); // the windows of the first copy + those across the seam
		CHECK(again == 100 - k + 1);             // the second copy was all there
	}
	// 4. an insert from ANOTHER thread that has returned is seen by the next contains on this one
	{
		ntHashIterator itr(reads[7], h, k);
		CHECK(!bloom.contains(*itr)); // answers for all rows of reads[7] are held now
		std::thread t([&] { insertSeq(bloom, reads[7], h, k); });
		t.join();
		while (itr != itr.end()) {
			CHECK(bloom.contains(*itr));
			++itr;
		}
	}
	// 5. copies hash for themselves; rows of an iterator that is gone are just rows
	{
		std::vector<uint64_t> row;
		{
			ntHashIterator itr(reads[9], h, k);
			ntHashIterator cp = itr;
			CHECK(!bloom.contains(*cp) && !bloom.contains(*itr));
			++cp;
			CHECK(!bloom.contains(*cp));
			row.assign(*itr, *itr + h);
			ntHashIterator in(reads[10], h, k);
			CHECK(bloom.contains(*in));
		}
		CHECK(!bloom.contains(row.data()));
		CHECK(!bloom.contains(row));
	}
	// 6. many threads, each with its own iterator, one filter
	std::atomic<unsigned> wrong{ 0 };
#pragma omp parallel for schedule(dynamic, 8)
	for (int r = 20; r < 400; ++r) {
		ntHashIterator itr(reads[(size_t)r], h, k);
		while (itr != itr.end()) {
			if (bloom.contains(*itr) != (r % 2 == 0))
				++wrong;
			++itr;
		}
	}
	CHECK(wrong == 0);
	// 7. a LONG sequence (the iterator announces tens of thousands of rows): the look-ahead fetches a window, not the
	//    whole span, so the de-duplicating loop stays linear -- and two filters asked by turns keep their own answers.
	//    Timed loosely: a refresh of the whole span per k-mer (the first version) needs minutes here.
	{
		std::string chrom;
		for (unsigned r = 0; r < 400; ++r)
			chrom += synth_read(6, r, L); // 60 000 bases, nothing of it in `bloom`
		chrom += chrom.substr(0, 20000);  // ... and its first third once more
		BloomFilter seen(1 << 26, h, k);
		const double t0 = omp_get_wtime();
		ntHashIterator itr(chrom, h, k);
		size_t fresh = 0, again = 0, in_other = 0;
		while (itr != itr.end()) {
			if (!seen.contains(*itr)) {
				seen.insert(*itr);
				++fresh;
			} else {
				++again;
			}
			in_other += bloom.contains(*itr) ? 1 : 0; // a second filter about the same rows, by turns with the first
			++itr;
		}
		const double dt = omp_get_wtime() - t0;
		CHECK(fresh == 60000 - k + 1 + (k - 1)); // every window of the 60 000 bases + those across the seam
		CHECK(again == 20000 - k + 1);
		CHECK(in_other == 0);
		CHECK(dt < 60.0);
		std::printf("dedup loop over %zu bases: %.2f s\n", chrom.size(), dt);
		// the same sequence again: everything is there now, one window of answers per 4096 rows
		const double t1 = omp_get_wtime();
		ntHashIterator again_itr(chrom, h, k);
		size_t hits = 0;
		while (again_itr != again_itr.end()) {
			hits += seen.contains(*again_itr) ? 1 : 0;
			++again_itr;
		}
		CHECK(hits == chrom.size() - k + 1);
		CHECK(omp_get_wtime() - t1 < 10.0);
	}
}

int main(int argc, char** argv)
{
	bloom_basic();
	hash_known_answers();
	counting_basic();
	fused_path_equals_iterator_path(argc > 1 ? argv[1] : nullptr);
	file_loaders_equal_insert_seq();
	sizing_ctor_and_load_header();
	parallel_filter_replay(argc > 2 ? argv[2] : nullptr);
	swig_test_pl_replay(argc > 3 ? argv[3] : nullptr);
	insert_seq_is_write_combined();
	contains_lookahead_is_invisible();
	if (g_fail) {
		std::fprintf(stderr, "%d checks failed\n", g_fail);
		return 1;
	}
	std::printf("all shim tests passed\n");
	return 0;
}

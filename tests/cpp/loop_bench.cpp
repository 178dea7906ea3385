// tests/cpp/loop_bench.cpp -- the reference's own host loops, timed: TEST / MEASUREMENT harness, not product code.
//
// The same source is compiled twice (tests/loop_bench.py):
//   g++ -DLOOP_BENCH_REFERENCE -I/root/reference ...  -> oracle/_ref/loop_bench_ref   (the genuine headers; recipe in
//                                                        oracle/Makefile, built only where the reference tree exists)
//   g++ -Iinclude ... -lbtlbf                         -> tests/cpp/loop_bench_shim    (the drop-in shims over the C ABI)
// and run on the same host cores.  Loops (OpenMP threads over pre-generated synthetic reads):
//   kmer   : ntHashIterator itr(seq, h, k); while (itr != itr.end()) { bloom.insert(*itr); ++itr; }
//            (Tests/AdHoc/ParallelFilter.cpp:93-101 inside its OpenMP loader :104-122)
//   seq    : insertSeq(bloom, seq, h, k)                                   (BloomFilterUtil.h:9-17)
//   batch  : shims only -- all reads handed over as ONE host buffer (bloom.insertReads; no counterpart there)
// then a query pass of the same shape (contains(*itr) per k-mer for `kmer`; containsSeq per read / one buffer for the
// others -- the reference has no batch query, so its `seq` query IS the per-k-mer loop).
// Prints one JSON line: mode, threads, reads, k-mers, seconds and Mk-mers/s of insert and query, hits, popcount.
#ifdef LOOP_BENCH_REFERENCE
#include "BloomFilter.hpp"
#include "BloomFilterUtil.h"
#include "vendor/ntHashIterator.hpp"
#else
#include "btlbf/BloomFilter.hpp"
#include "btlbf/BloomFilterUtil.h"
#include "btlbf/ntHashIterator.hpp"
#endif

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <omp.h>
#include <string>
#include <vector>

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

static double now()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv)
{
	if (argc < 5) {
		std::fprintf(stderr, "usage: %s kmer|seq|batch n_reads log2_bits threads [n_query_reads]\n", argv[0]);
		return 2;
	}
	const std::string mode = argv[1];
	const long n_reads = std::atol(argv[2]);
	const unsigned log2_bits = (unsigned)std::atoi(argv[3]);
	const int threads = std::atoi(argv[4]);
	// the query pass may look at fewer reads (the first n_query): a per-k-mer contains() through the shims is one GPU
	// round trip each, and a rate does not need 10^8 of them
	const long n_query = argc > 5 ? std::min(std::atol(argv[5]), n_reads) : n_reads;
	const unsigned L = 150, h = 4, k = 31;
	omp_set_num_threads(threads);
	std::vector<std::string> reads((size_t)n_reads);
#pragma omp parallel for schedule(static)
	for (long r = 0; r < n_reads; ++r)
		reads[(size_t)r] = synth_read(42, (uint64_t)r, L);

	BloomFilter bloom((size_t)1 << log2_bits, h, k);
	(void)bloom.getPop(); // the array exists and is zero before the clock starts (GPU: allocation + clear)
	unsigned long long hits = 0;
	const unsigned long long kmers = (unsigned long long)n_reads * (L - k + 1);
	double t0 = now();
	if (mode == "kmer") {
#pragma omp parallel for schedule(dynamic, 64)
		for (long r = 0; r < n_reads; ++r) {
			ntHashIterator itr(reads[(size_t)r], h, k);
			while (itr != itr.end()) {
				bloom.insert(*itr);
				++itr;
			}
		}
	} else if (mode == "seq") {
#pragma omp parallel for schedule(dynamic, 64)
		for (long r = 0; r < n_reads; ++r)
			insertSeq(bloom, reads[(size_t)r], h, k);
	}
#ifndef LOOP_BENCH_REFERENCE
	else if (mode == "batch") {
		std::string all;
		all.reserve((size_t)n_reads * L);
		for (const auto& s : reads)
			all += s;
		bloom.insertReads(all.data(), all.size(), L);
	}
#endif
	else {
		std::fprintf(stderr, "unknown mode %s\n", mode.c_str());
		return 2;
	}
	const unsigned long long pop = bloom.getPop(); // (the shims flush their queues here: part of the insert time)
	const double t_ins = now() - t0;
	t0 = now();
	if (mode == "kmer" || mode == "seq"
#ifdef LOOP_BENCH_REFERENCE
	    || true
#endif
	) {
#ifdef LOOP_BENCH_REFERENCE
		const bool per_kmer = true;
#else
		const bool per_kmer = mode == "kmer";
#endif
		if (per_kmer) {
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : hits)
			for (long r = 0; r < n_query; ++r) {
				ntHashIterator itr(reads[(size_t)r], h, k);
				while (itr != itr.end()) {
					hits += bloom.contains(*itr);
					++itr;
				}
			}
		}
#ifndef LOOP_BENCH_REFERENCE
		else {
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : hits)
			for (long r = 0; r < n_query; ++r)
				hits += bloom.countSeq(reads[(size_t)r]);
		}
#endif
	}
#ifndef LOOP_BENCH_REFERENCE
	else {
		std::string all;
		all.reserve((size_t)n_query * L);
		for (long r = 0; r < n_query; ++r)
			all += reads[(size_t)r];
		hits = bloom.countReads(all.data(), all.size(), L);
	}
#endif
	const double t_qry = now() - t0;
	const unsigned long long qkmers = (unsigned long long)n_query * (L - k + 1);
	std::printf("{\"impl\": \"%s\", \"mode\": \"%s\", \"threads\": %d, \"reads\": %ld, \"kmers\": %llu, \"log2_bits\": %u, "
	            "\"insert_s\": %.4f, \"insert_Mkmers_s\": %.2f, \"query_reads\": %ld, \"query_s\": %.4f, "
	            "\"query_Mkmers_s\": %.3f, \"hits\": %llu, \"pop\": %llu}\n",
#ifdef LOOP_BENCH_REFERENCE
	            "reference",
#else
	            "shim",
#endif
	            mode.c_str(), threads, n_reads, kmers, log2_bits, t_ins, kmers / t_ins / 1e6, n_query, t_qry,
	            qkmers / t_qry / 1e6, hits, pop);
	return 0;
}

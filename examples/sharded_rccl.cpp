// examples/sharded_rccl.cpp -- a hash-range sharded Bloom filter over all GPUs of a node from ONE C++
// process: the C ABI (include/btlbf.h) + RCCL, no Python.  It is the C++ twin of
// btl_bloomfilter_amd/sharded.py's routed path: per batch every GPU hashes its own reads and
// radix-partitions the probe positions by owning shard (btlbf_route_seqs), one fixed-size all-to-all
// moves the 4-byte entries (ncclSend / ncclRecv inside a group), the owners apply them in LDS
// (btlbf_apply_routed).  Queries return only the positions found clear (all-gathered fail lists).
//
//   hipcc -std=c++17 -O2 -Iinclude examples/sharded_rccl.cpp -Lbtl_bloomfilter_amd -lbtlbf -lrccl \
//         -Wl,-rpath,$PWD/btl_bloomfilter_amd -o sharded_rccl
//   ./sharded_rccl [log2_bits_per_gpu=33] [reads_per_gpu=2000000]
//
// Synthetic 150 bp reads (SURVEY 8d generator, GPU g owns reads [g*n, (g+1)*n)), k = 31, h = 4.  With
// one visible GPU the exchange degenerates to a copy to self, which is how the tests run it.
#include "btlbf.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK_HIP(x)                                                                  \
	do {                                                                              \
		hipError_t e_ = (x);                                                          \
		if (e_ != hipSuccess) {                                                       \
			std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
			return 1;                                                                 \
		}                                                                             \
	} while (0)
#define CHECK_NCCL(x)                                                                 \
	do {                                                                              \
		ncclResult_t r_ = (x);                                                        \
		if (r_ != ncclSuccess) {                                                      \
			std::fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_));             \
			return 1;                                                                 \
		}                                                                             \
	} while (0)
#define CHECK_BF(x)                                                                   \
	do {                                                                              \
		if ((x) != BTLBF_OK) {                                                        \
			std::fprintf(stderr, "%s: %s\n", #x, btlbf_last_error());                 \
			return 1;                                                                 \
		}                                                                             \
	} while (0)

struct Rank {
	btlbf_filter* f = nullptr;
	hipStream_t s = nullptr;
	char* reads = nullptr;
	void *send_ent = nullptr, *send_cnt = nullptr, *recv_ent = nullptr, *recv_cnt = nullptr;
	uint64_t *spill = nullptr, *spill_count = nullptr, *fail = nullptr, *fail_count = nullptr;
	uint64_t *hit = nullptr, *counts = nullptr;
};

int main(int argc, char** argv)
{
	const unsigned lg = argc > 1 ? (unsigned)std::atoi(argv[1]) : 33;
	const uint64_t n_reads = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 2000000;
	const unsigned k = 31, h = 4, L = 150;
	int W = 0;
	CHECK_HIP(hipGetDeviceCount(&W));
	while (W & (W - 1))
		--W; // a power of two of them
	if (W < 1)
		return 1;
	const uint64_t total_bits = (uint64_t(1) << lg) * W, len = n_reads * L;
	const uint64_t spill_cap = 1 << 20, fail_cap = 4 << 20;
	std::vector<int> devs(W);
	for (int g = 0; g < W; ++g)
		devs[g] = g;
	std::vector<ncclComm_t> comm(W);
	CHECK_NCCL(ncclCommInitAll(comm.data(), W, devs.data()));
	std::vector<Rank> r(W);
	btlbf_layout lay{nullptr, 0, L};
	uint64_t eb = 0, cb = 0;
	for (int g = 0; g < W; ++g) {
		CHECK_HIP(hipSetDevice(g));
		CHECK_HIP(hipStreamCreate(&r[g].s));
		CHECK_BF(btlbf_create_shard(&r[g].f, BTLBF_BLOOM, total_bits, g, W, h, k, 0, g));
		CHECK_BF(btlbf_route_plan(r[g].f, len, &lay, W, &eb, &cb));
		CHECK_HIP(hipMalloc(&r[g].reads, len + 64));
		CHECK_BF(btlbf_synth_reads(r[g].reads, 42, g * n_reads, n_reads, L, g, r[g].s));
		CHECK_HIP(hipMalloc(&r[g].send_ent, W * eb));
		CHECK_HIP(hipMalloc(&r[g].recv_ent, W * eb));
		CHECK_HIP(hipMalloc(&r[g].send_cnt, W * cb));
		CHECK_HIP(hipMalloc(&r[g].recv_cnt, W * cb));
		CHECK_HIP(hipMalloc(&r[g].spill, spill_cap * 8));
		CHECK_HIP(hipMalloc(&r[g].fail, fail_cap * 8));
		CHECK_HIP(hipMalloc(&r[g].spill_count, 8));
		CHECK_HIP(hipMalloc(&r[g].fail_count, 8));
		CHECK_HIP(hipMalloc(&r[g].counts, 16));
		CHECK_HIP(hipMalloc(&r[g].hit, ((len + 63) / 64) * 8));
	}
	// one pass = route on every GPU, exchange, apply on every GPU (the whole buffer is one batch here)
	auto pass = [&](int query) -> int {
		for (int g = 0; g < W; ++g) {
			CHECK_HIP(hipSetDevice(g));
			CHECK_HIP(hipMemsetAsync(r[g].spill_count, 0, 8, r[g].s));
			CHECK_HIP(hipMemsetAsync(r[g].fail_count, 0, 8, r[g].s));
			CHECK_HIP(hipMemsetAsync(r[g].counts, 0, 16, r[g].s));
			// window 0: this example handles filters of up to 2^42 bits (one position window, btlbf_route_windows)
			CHECK_BF(btlbf_route_seqs(r[g].f, r[g].reads, len, &lay, len, W, 0, query, r[g].send_ent, r[g].send_cnt,
			                          query ? r[g].hit : nullptr, nullptr, query ? r[g].counts : nullptr, r[g].spill,
			                          spill_cap, r[g].spill_count, r[g].s));
		}
		// block p of GPU g goes to GPU p; the stream orders it after the routing.  No message is larger
		// than 256 MiB: on this stack a grouped ncclSend/ncclRecv of more than about 1 GiB delivered only
		// half of its bytes (SHARDED_RCCL_DEBUG=1 prints checksums of what was sent and received)
		const uint64_t slice = 256ull << 20;
		for (uint64_t off = 0; off < eb; off += slice) {
			const uint64_t n = eb - off < slice ? eb - off : slice;
			CHECK_NCCL(ncclGroupStart());
			for (int g = 0; g < W; ++g)
				for (int p = 0; p < W; ++p) {
					CHECK_NCCL(ncclSend((char*)r[g].send_ent + p * eb + off, n / 8, ncclUint64, p, comm[g], r[g].s));
					CHECK_NCCL(ncclRecv((char*)r[g].recv_ent + p * eb + off, n / 8, ncclUint64, p, comm[g], r[g].s));
				}
			CHECK_NCCL(ncclGroupEnd());
		}
		CHECK_NCCL(ncclGroupStart());
		for (int g = 0; g < W; ++g)
			for (int p = 0; p < W; ++p) {
				CHECK_NCCL(ncclSend((char*)r[g].send_cnt + p * cb, cb / 8, ncclUint64, p, comm[g], r[g].s));
				CHECK_NCCL(ncclRecv((char*)r[g].recv_cnt + p * cb, cb / 8, ncclUint64, p, comm[g], r[g].s));
			}
		CHECK_NCCL(ncclGroupEnd());
		if (std::getenv("SHARDED_RCCL_DEBUG")) { // checksum of what was sent to and received from peer 0
			for (int g = 0; g < W; ++g) {
				CHECK_HIP(hipSetDevice(g));
				CHECK_HIP(hipStreamSynchronize(r[g].s));
				uint64_t a = 0, b = 0, c = 0, d = 0;
				CHECK_BF(btlbf_popcount_bits(r[0].send_ent == nullptr ? nullptr : (char*)r[g].send_ent, eb / 8 * 8, &a, g, r[g].s));
				CHECK_BF(btlbf_popcount_bits((char*)r[g].recv_ent, eb / 8 * 8, &b, g, r[g].s));
				CHECK_BF(btlbf_popcount_bits((char*)r[g].send_cnt, cb / 8 * 8, &c, g, r[g].s));
				CHECK_BF(btlbf_popcount_bits((char*)r[g].recv_cnt, cb / 8 * 8, &d, g, r[g].s));
				std::fprintf(stderr, "gpu %d: eb %llu cb %llu  ent sent/received bit sums %llu / %llu  cnt %llu / %llu\n", g,
				             (unsigned long long)eb, (unsigned long long)cb, (unsigned long long)a, (unsigned long long)b,
				             (unsigned long long)c, (unsigned long long)d);
			}
		}
		for (int g = 0; g < W; ++g) {
			CHECK_HIP(hipSetDevice(g));
			CHECK_BF(btlbf_apply_routed(r[g].f, r[g].recv_ent, r[g].recv_cnt, W, len, &lay, W, query,
			                            query ? r[g].fail : nullptr, query ? fail_cap : 0,
			                            query ? r[g].fail_count : nullptr, r[g].s));
		}
		// spill / fail lists: tiny or empty; brought to the host, merged, handed to every GPU
		std::vector<uint64_t> spills, fails;
		for (int g = 0; g < W; ++g) {
			CHECK_HIP(hipSetDevice(g));
			CHECK_HIP(hipStreamSynchronize(r[g].s));
			uint64_t ns = 0;
			CHECK_HIP(hipMemcpy(&ns, r[g].spill_count, 8, hipMemcpyDeviceToHost));
			if (ns > spill_cap)
				return 1;
			const size_t o = spills.size();
			spills.resize(o + ns);
			if (ns)
				CHECK_HIP(hipMemcpy(spills.data() + o, r[g].spill, ns * 8, hipMemcpyDeviceToHost));
		}
		for (int g = 0; g < W && !spills.empty(); ++g) {
			CHECK_HIP(hipSetDevice(g));
			CHECK_HIP(hipMemcpy(r[g].spill, spills.data(), spills.size() * 8, hipMemcpyHostToDevice));
			CHECK_BF(btlbf_apply_spill(r[g].f, r[g].spill, spills.size(), query, query ? r[g].fail : nullptr,
			                           query ? fail_cap : 0, query ? r[g].fail_count : nullptr, r[g].s));
			CHECK_HIP(hipStreamSynchronize(r[g].s));
		}
		if (query) {
			for (int g = 0; g < W; ++g) {
				CHECK_HIP(hipSetDevice(g));
				uint64_t nf = 0;
				CHECK_HIP(hipMemcpy(&nf, r[g].fail_count, 8, hipMemcpyDeviceToHost));
				if (nf > fail_cap)
					return 1; // a miss-heavy query: use the direct answer routing (see sharded.py)
				const size_t o = fails.size();
				fails.resize(o + nf);
				if (nf)
					CHECK_HIP(hipMemcpy(fails.data() + o, r[g].fail, nf * 8, hipMemcpyDeviceToHost));
			}
			for (int g = 0; g < W && !fails.empty(); ++g) {
				CHECK_HIP(hipSetDevice(g));
				CHECK_HIP(hipMemcpy(r[g].fail, fails.data(), fails.size() * 8, hipMemcpyHostToDevice));
				CHECK_BF(btlbf_resolve_seqs(r[g].f, r[g].reads, len, &lay, r[g].fail, fails.size(), r[g].hit, r[g].s));
				CHECK_HIP(hipStreamSynchronize(r[g].s));
			}
		}
		return 0;
	};
	hipEvent_t e0, e1, e2;
	CHECK_HIP(hipSetDevice(0));
	CHECK_HIP(hipEventCreate(&e0));
	CHECK_HIP(hipEventCreate(&e1));
	CHECK_HIP(hipEventCreate(&e2));
	for (int g = 0; g < W; ++g) {
		CHECK_HIP(hipSetDevice(g));
		CHECK_HIP(hipStreamSynchronize(r[g].s));
	}
	CHECK_HIP(hipSetDevice(0));
	CHECK_HIP(hipEventRecord(e0, r[0].s));
	if (pass(0))
		return 1;
	CHECK_HIP(hipSetDevice(0));
	CHECK_HIP(hipEventRecord(e1, r[0].s));
	if (pass(1))
		return 1;
	CHECK_HIP(hipSetDevice(0));
	CHECK_HIP(hipEventRecord(e2, r[0].s));
	CHECK_HIP(hipEventSynchronize(e2));
	float ms_i = 0, ms_q = 0;
	CHECK_HIP(hipEventElapsedTime(&ms_i, e0, e1));
	CHECK_HIP(hipEventElapsedTime(&ms_q, e1, e2));
	uint64_t pop = 0, clean = 0, hits = 0;
	for (int g = 0; g < W; ++g) {
		CHECK_HIP(hipSetDevice(g));
		uint64_t p = 0, c[2] = {0, 0}, hb = 0;
		CHECK_BF(btlbf_popcount(r[g].f, &p));
		CHECK_HIP(hipMemcpy(c, r[g].counts, 16, hipMemcpyDeviceToHost));
		CHECK_BF(btlbf_popcount_bits(r[g].hit, ((len + 63) / 64) * 8, &hb, g, r[g].s));
		pop += p;
		clean += c[0];
		hits += hb;
	}
	const double kmers = (double)W * n_reads * (L - k + 1);
	// expected popcount of M bits after n probes at random: M * (1 - exp(-n/M)); a lost block shows here
	const double expect = (double)total_bits * (1.0 - std::exp(-kmers * h / (double)total_bits));
	std::printf("gpus %d  bits %llu  k-mers %.0f  popcount %llu (expected about %.0f)  clean %llu  hits %llu  insert %.1f Mk-mers/s  query %.1f Mk-mers/s\n",
	            W, (unsigned long long)total_bits, kmers, (unsigned long long)pop, expect, (unsigned long long)clean,
	            (unsigned long long)hits, kmers / ms_i / 1e3, kmers / ms_q / 1e3);
	for (int g = 0; g < W; ++g) {
		CHECK_HIP(hipSetDevice(g));
		btlbf_destroy(r[g].f);
		ncclCommDestroy(comm[g]);
	}
	// every inserted k-mer is found again, and the number of set bits is what that many probes give
	return clean == (uint64_t)kmers && hits == clean && std::fabs((double)pop - expect) < 0.002 * expect ? 0 : 2;
}

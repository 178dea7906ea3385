// tests/cpp/stub_abi.cpp -- TEST-ONLY stand-in for the handful of C-ABI entry points the shim's threading
// test touches, so that the CPU side of include/btlbf/*.hpp (striped insert queues, flush, per-call row
// buffers) can run under ThreadSanitizer without a GPU and without the HIP runtime's own threads in the
// picture.  A plain bit / byte array behind one mutex -- the same contract as the real library's per-filter
// lock.  It is linked only into tests/cpp/test_shims_tsan; nothing in the product links or loads it.
#include "../../include/btlbf.h"

#include <cstring>
#include <mutex>
#include <vector>

struct btlbf_filter {
	std::mutex mu;
	int kind;
	uint64_t size;
	unsigned h, k, thr;
	std::vector<uint8_t> data;
};

extern "C" {
const char* btlbf_last_error(void) { return "stub"; }
int btlbf_create(btlbf_filter** out, int kind, uint64_t size, unsigned h, unsigned k, unsigned thr, int)
{
	btlbf_filter* f = new btlbf_filter();
	f->kind = kind;
	f->size = size;
	f->h = h;
	f->k = k;
	f->thr = thr;
	f->data.assign(kind == BTLBF_BLOOM ? size / 8 : size, 0);
	*out = f;
	return BTLBF_OK;
}
int btlbf_destroy(btlbf_filter* f)
{
	delete f;
	return BTLBF_OK;
}
unsigned btlbf_hash_num(const btlbf_filter* f) { return f->h; }
unsigned btlbf_kmer_size(const btlbf_filter* f) { return f->k; }
uint64_t btlbf_size(const btlbf_filter* f) { return f->size; }
uint64_t btlbf_size_bytes(const btlbf_filter* f) { return f->data.size(); }
int btlbf_insert_hashes(btlbf_filter* f, const uint64_t* hs, uint64_t n, int, int, int, void*)
{
	std::lock_guard<std::mutex> g(f->mu);
	for (uint64_t i = 0; i < n * f->h; ++i) {
		const uint64_t p = hs[i] % f->size;
		if (f->kind == BTLBF_BLOOM)
			f->data[p / 8] |= (uint8_t)(1u << (p % 8));
		else if (f->data[p] != 255)
			++f->data[p];
	}
	return BTLBF_OK;
}
int btlbf_contains_hashes(btlbf_filter* f, const uint64_t* hs, uint64_t n, uint8_t* out, int, void*)
{
	std::lock_guard<std::mutex> g(f->mu);
	for (uint64_t r = 0; r < n; ++r) {
		uint8_t all = 1;
		for (unsigned i = 0; i < f->h; ++i) {
			const uint64_t p = hs[r * f->h + i] % f->size;
			all &= f->kind == BTLBF_BLOOM ? (f->data[p / 8] >> (p % 8)) & 1 : f->data[p] >= f->thr;
		}
		out[r] = all;
	}
	return BTLBF_OK;
}
int btlbf_insert_and_check_hashes(btlbf_filter* f, const uint64_t* hs, uint64_t n, uint8_t* out, int, int, void*)
{
	btlbf_contains_hashes(f, hs, n, out, 0, nullptr);
	return btlbf_insert_hashes(f, hs, n, 0, 0, 0, nullptr);
}
int btlbf_min_count_hashes(btlbf_filter* f, const uint64_t* hs, uint64_t n, uint8_t* out, int, void*)
{
	std::lock_guard<std::mutex> g(f->mu);
	for (uint64_t r = 0; r < n; ++r) {
		uint8_t m = 255;
		for (unsigned i = 0; i < f->h; ++i)
			m = std::min(m, f->data[hs[r * f->h + i] % f->size]);
		out[r] = m;
	}
	return BTLBF_OK;
}
// the sequence / raw-k-mer queues of the shims are flushed through these; the stub has no hashing, so a queued
// sequence or k-mer just sets one position derived from its bytes (enough for the locking test)
static void stub_touch(btlbf_filter* f, const char* p, uint64_t n)
{
	uint64_t x = 1469598103934665603ULL;
	for (uint64_t i = 0; i < n; ++i)
		x = (x ^ (uint8_t)p[i]) * 1099511628211ULL;
	const uint64_t pos = x % f->size;
	if (f->kind == BTLBF_BLOOM)
		f->data[pos / 8] |= (uint8_t)(1u << (pos % 8));
	else if (f->data[pos] != 255)
		++f->data[pos];
}
int btlbf_insert_seqs(btlbf_filter* f, const char* seq, uint64_t len, const btlbf_layout* lay, int, int, int, void*)
{
	std::lock_guard<std::mutex> g(f->mu);
	if (lay && lay->starts)
		for (uint64_t i = 0; i < lay->n_seqs; ++i)
			stub_touch(f, seq + lay->starts[i], lay->starts[i + 1] - lay->starts[i]);
	else
		stub_touch(f, seq, len);
	return BTLBF_OK;
}
int btlbf_insert_kmers(btlbf_filter* f, const char* kmers, uint64_t n, int, int, int, void*)
{
	std::lock_guard<std::mutex> g(f->mu);
	for (uint64_t i = 0; i < n; ++i)
		stub_touch(f, kmers + i * f->k, f->k);
	return BTLBF_OK;
}
int btlbf_popcount(btlbf_filter* f, uint64_t* out)
{
	std::lock_guard<std::mutex> g(f->mu);
	uint64_t c = 0;
	for (uint8_t b : f->data)
		c += f->kind == BTLBF_BLOOM ? (uint64_t)__builtin_popcount(b) : (b != 0);
	*out = c;
	return BTLBF_OK;
}
}

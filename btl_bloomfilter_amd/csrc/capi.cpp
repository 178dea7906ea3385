// csrc/capi.cpp -- the C ABI declared in include/btlbf.h (host side; compiled by hipcc).
//
// Owns the HBM-resident filter arrays, builds the hash/modulo parameter blocks, stages host
// buffers, launches the HIP kernels and reads/writes BTLBloomFilter_v1 files byte-for-byte the
// way the reference does (BloomFilter.hpp:107-166,264-314; CountingBloomFilter.hpp:268-368).
// There is deliberately no CPU implementation of any compute entry point in this file.
#include "../../include/btlbf.h"
#include "internal.hpp"
#include "host_internal.hpp"

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <map>
#include <mutex>
#include <string>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

using namespace btlbf;

// -------------------------------------------------------------------------------------------------
// errors
// -------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

// Every error return of this file passes here BEFORE the locals of the failing call are destroyed.  An error that
// comes after kernels were queued would otherwise hand pooled staging buffers (DevBuf) back to the pool -- and to
// the next call -- with that work still pending; so the device is drained first (errors are not a hot path).
static int fail(int code, const char* fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	if (hipDeviceSynchronize() != hipSuccess)
		(void)hipGetLastError();
	return code;
}

int btlbf_set_error(int code, const char* fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return code;
}

#define HIP_TRY(expr)                                                                              \
	do {                                                                                           \
		hipError_t e__ = (expr);                                                                   \
		if (e__ != hipSuccess)                                                                     \
			return fail(BTLBF_EHIP, "%s failed: %s", #expr, hipGetErrorString(e__));               \
	} while (0)

extern "C" const char* btlbf_last_error(void) { return g_err; }

extern "C" int btlbf_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

// -------------------------------------------------------------------------------------------------
// filter object
// -------------------------------------------------------------------------------------------------
struct btlbf_filter {
	// every entry point that takes the filter holds this for its whole duration: a filter keeps device
	// scratch, event lists and a scalar buffer between calls, so concurrent callers are serialised here
	// (recursive: btlbf_store -> btlbf_store_shard, btlbf_apply_routed -> btlbf_apply_routed_bins)
	mutable std::recursive_mutex mu;
	int kind = BTLBF_BLOOM;
	int device = 0;
	uint64_t size = 0;        // global bits / counters
	uint64_t size_bytes = 0;  // global bytes
	uint64_t local_bytes = 0; // bytes held here
	uint64_t alloc_bytes = 0; // local_bytes rounded up to 16 (zero padded)
	unsigned h = 0, k = 0, thr = 0;
	double dfpr = 0.0;
	uint64_t n_entry = 0, t_entry = 0;
	unsigned bits_per_counter = 8;
	unsigned shard_index = 0, shard_count = 1;
	void* d_data = nullptr;
	// btlbf_clear is LAZY: it only sets this.  The next partitioned insert builds every segment from zero in LDS
	// and writes it (no memset of the array, no read sweep for that batch); any other entry point that touches
	// the array zeroes it first (materialize_clear)
	bool lazy_zero = false;
	// the clear is ordered on the caller's stream: clear_ev is recorded there, and whichever stream carries the
	// zeroing out (materialize_clear, or the fresh partitioned insert) waits for it first, so work that was queued
	// before the clear on the caller's stream cannot run after (or beside) the zeroing
	hipEvent_t clear_ev = nullptr;
	bool clear_ev_pending = false;
	// ... and the zeroing itself, once some stream carries it out, is an event too: a call on ANOTHER stream (another
	// host thread's BTLBF_STREAM_PER_THREAD, say) that finds lazy_zero already false must not look at the array
	// while that memset is still in flight -- it waits for zero_ev first (materialize_clear)
	hipEvent_t zero_ev = nullptr;
	hipStream_t zero_stream = nullptr;
	bool zero_ev_pending = false;
	// set once btlbf_device_ptr has handed the raw pointer out: the caller may keep it, so from then on a clear
	// zeroes eagerly on its stream (a lazily cleared array would show stale contents through that pointer)
	bool ptr_exposed = false;
	ModParams mod{};
	HashParams hp{};
	// spaced seeds
	uint64_t* d_pos_tab = nullptr;
	uint16_t* d_dc_idx = nullptr;
	// small device scratch for counters
	unsigned long long* d_scalar = nullptr; // 4 x u64
	// partitioned insert (partition_kernels.hip): mode + cached scratch
	int insert_mode = BTLBF_INSERT_AUTO;
	int query_mode = BTLBF_INSERT_AUTO;
	void* d_part = nullptr;
	uint64_t part_bytes = 0;
	void* d_split = nullptr; // split query: compacted reads + their bitmaps, cached like d_part
	uint64_t split_bytes = 0;
	void* d_flags = nullptr; // split query: cold flags of the reads + their prefix sums (small)
	uint64_t flags_bytes = 0;
	uint64_t part_budget = 0; // 0 = derive from free HBM
	// optional per-kernel timing with HIP events on the launch stream (btlbf_set_profiling)
	bool profiling = false;
	struct Span {
		int slot;
		hipEvent_t e0, e1;
	};
	std::vector<Span> spans;
	double prof_ms[BTLBF_PROF_SLOTS] = {0};
	unsigned prof_calls[BTLBF_PROF_SLOTS] = {0};
};

namespace {

struct FilterLock {
	const btlbf_filter* f;
	explicit FilterLock(const btlbf_filter* f_)
	  : f(f_)
	{
		if (f)
			f->mu.lock();
	}
	~FilterLock()
	{
		if (f)
			f->mu.unlock();
	}
	// give the filter back early: a read-only call that has launched its kernel (on the caller's own stream, with
	// the caller's own buffers) only waits from here on, and other threads' calls may as well run meanwhile
	void release()
	{
		if (f)
			f->mu.unlock();
		f = nullptr;
	}
	FilterLock(const FilterLock&) = delete;
	FilterLock& operator=(const FilterLock&) = delete;
};

// stream s is about to carry out a pending clear: it first waits for the point of the clear on the caller's stream
hipError_t order_after_clear(btlbf_filter* f, hipStream_t s)
{
	if (!f->clear_ev_pending)
		return hipSuccess;
	f->clear_ev_pending = false;
	return hipStreamWaitEvent(s, f->clear_ev, 0);
}

// a pending btlbf_clear takes effect now, on the stream of the operation that needs the array
hipError_t materialize_clear(btlbf_filter* f, hipStream_t s)
{
	if (!f)
		return hipSuccess;
	if (!f->lazy_zero) {
		// an earlier call zeroed the array on ITS stream: a different stream waits for that zeroing (a per-thread
		// stream handle names a different stream in every host thread, so it always waits; waiting for an event of
		// one's own stream costs nothing)
		if (!f->zero_ev_pending)
			return hipSuccess;
		if (s == f->zero_stream && s != hipStreamPerThread)
			return hipSuccess;
		if (hipEventQuery(f->zero_ev) == hipSuccess) {
			f->zero_ev_pending = false;
			return hipSuccess;
		}
		(void)hipGetLastError(); // hipErrorNotReady is not an error
		return hipStreamWaitEvent(s, f->zero_ev, 0);
	}
	hipError_t e = order_after_clear(f, s);
	if (e != hipSuccess)
		return e;
	f->lazy_zero = false;
	e = hipMemsetAsync(f->d_data, 0, f->alloc_bytes, s);
	if (e != hipSuccess)
		return e;
	if (!f->zero_ev && (e = hipEventCreateWithFlags(&f->zero_ev, hipEventDisableTiming)) != hipSuccess)
		return e;
	if ((e = hipEventRecord(f->zero_ev, s)) != hipSuccess)
		return e;
	f->zero_stream = s;
	f->zero_ev_pending = true;
	return hipSuccess;
}

// every launch of a kernel that reads or writes the array directly goes through this check: an entry point that
// forgot MATERIALIZE would otherwise read uninitialised HBM (the fresh partitioned insert is the one legitimate
// user of a lazily cleared array and does not come this way)
#define REQUIRE_MATERIALIZED(f)                                                                               \
	do {                                                                                                      \
		if ((f)->lazy_zero)                                                                                   \
			return fail(BTLBF_EINVAL, "internal error: %s line %d launches on a lazily cleared array", __func__, \
			            __LINE__);                                                                            \
	} while (0)
#define MATERIALIZE(f, s)                                                                              \
	do {                                                                                               \
		hipError_t em__ = materialize_clear(const_cast<btlbf_filter*>(f), static_cast<hipStream_t>(s)); \
		if (em__ != hipSuccess)                                                                        \
			return fail(BTLBF_EHIP, "clearing the filter failed: %s", hipGetErrorString(em__));         \
	} while (0)

// times one kernel launch with a pair of events when profiling is on
struct ProfSpan {
	btlbf_filter* f;
	hipStream_t s;
	int idx = -1;
	ProfSpan(btlbf_filter* f_, int slot, hipStream_t s_)
	  : f(f_)
	  , s(s_)
	{
		if (!f->profiling)
			return;
		btlbf_filter::Span sp{slot, nullptr, nullptr};
		if (hipEventCreate(&sp.e0) != hipSuccess || hipEventCreate(&sp.e1) != hipSuccess)
			return;
		(void)hipEventRecord(sp.e0, s);
		f->spans.push_back(sp);
		idx = (int)f->spans.size() - 1;
	}
	~ProfSpan()
	{
		if (idx >= 0)
			(void)hipEventRecord(f->spans[idx].e1, s);
	}
};

struct DeviceGuard {
	int prev = -1;
	bool ok = true;
	explicit DeviceGuard(int dev)
	{
		if (hipGetDevice(&prev) != hipSuccess) {
			ok = false;
			return;
		}
		if (prev != dev && hipSetDevice(dev) != hipSuccess)
			ok = false;
	}
	~DeviceGuard()
	{
		if (prev >= 0)
			(void)hipSetDevice(prev);
	}
};

// Small device buffers of HOST-mode calls (staged sequences, result buffers, hash rows) come from a pool:
// hipMalloc + hipFree per call cost more than the kernels of a per-read or per-k-mer call (the drop-in shims'
// ntHashIterator, contains(kmer), insertAndCheck make one such call each).  Power-of-two size classes up to
// 64 MiB, per device, at most 512 MiB parked.  A pooled buffer goes back only when the call that used it has
// synchronised its stream (every HOST-mode entry point does before it returns), so no work is pending on it.
struct DevPool {
	static constexpr size_t kMaxClass = 64u << 20, kMaxParked = 512u << 20;
	std::mutex mu;
	std::map<std::pair<int, size_t>, std::vector<void*>> parked;
	size_t parked_bytes = 0;
	static size_t size_class(size_t n)
	{
		size_t c = 4096;
		while (c < n)
			c <<= 1;
		return c;
	}
	void* take(int dev, size_t cls)
	{
		std::lock_guard<std::mutex> g(mu);
		auto it = parked.find({dev, cls});
		if (it == parked.end() || it->second.empty())
			return nullptr;
		void* p = it->second.back();
		it->second.pop_back();
		parked_bytes -= cls;
		return p;
	}
	bool give(int dev, size_t cls, void* p)
	{
		std::lock_guard<std::mutex> g(mu);
		if (parked_bytes + cls > kMaxParked)
			return false;
		parked[{dev, cls}].push_back(p);
		parked_bytes += cls;
		return true;
	}
	// hand everything parked for `dev` (or for every device: dev < 0) back to the runtime: called when a hipMalloc
	// fails -- a filter that nearly fills the HBM must not lose its scratch to parked staging buffers -- and by
	// btlbf_release_scratch.  The caller has the device selected.
	void drain(int dev)
	{
		std::vector<void*> out;
		{
			std::lock_guard<std::mutex> g(mu);
			for (auto& kv : parked) {
				if (dev >= 0 && kv.first.first != dev)
					continue;
				parked_bytes -= kv.first.second * kv.second.size();
				out.insert(out.end(), kv.second.begin(), kv.second.end());
				kv.second.clear();
			}
		}
		for (void* p : out)
			(void)hipFree(p);
	}
};
DevPool& dev_pool()
{
	static DevPool* pool = new DevPool(); // never destroyed: the HIP runtime may be gone by static destruction time
	return *pool;
}

struct DevBuf {
	void* p = nullptr;
	size_t pooled_class = 0;
	int pooled_dev = -1;
	~DevBuf()
	{
		if (!p)
			return;
		if (pooled_class && dev_pool().give(pooled_dev, pooled_class, p))
			return;
		(void)hipFree(p);
	}
	hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
	// for buffers whose user synchronises its stream before this object dies (HOST-mode staging)
	hipError_t alloc_pooled(size_t n)
	{
		if (n > DevPool::kMaxClass || hipGetDevice(&pooled_dev) != hipSuccess)
			return alloc(n);
		const size_t cls = DevPool::size_class(n ? n : 16);
		if ((p = dev_pool().take(pooled_dev, cls)) != nullptr) {
			pooled_class = cls;
			return hipSuccess;
		}
		hipError_t e = hipMalloc(&p, cls);
		if (e != hipSuccess) { // out of memory with buffers parked: give them back and try once more
			(void)hipGetLastError();
			dev_pool().drain(pooled_dev);
			e = hipMalloc(&p, cls);
		}
		if (e == hipSuccess)
			pooled_class = cls;
		return e;
	}
	// an error return with work still queued on the stream: the buffer must not go back to the pool (its next user
	// would share it with that work) -- freed instead, which synchronises
	void unpool() { pooled_class = 0; }
	template <class T>
	T* as()
	{
		return static_cast<T*>(p);
	}
};

// A pinned, GPU-mapped mailbox per host thread for small HOST-mode calls: the kernel reads its input from and
// writes its results to it directly (zero copy), so such a call is one launch and one stream synchronisation
// instead of staging buffers and three or four copies.  Freed when its thread ends.
struct Mailbox {
	static constexpr size_t kBytes = 1u << 20;
	uint8_t* host = nullptr;
	uint8_t* dev = nullptr;
	// a thread that ends gives its pinned megabytes back (a process that starts a thread per task would otherwise pin
	// memory without bound) -- unless the HIP runtime is already shutting down: hipHostFree then fails, harmlessly
	~Mailbox()
	{
		if (host)
			(void)hipHostFree(host);
		host = dev = nullptr;
	}
	bool get()
	{
		if (host)
			return true;
		void* h = nullptr;
		if (hipHostMalloc(&h, kBytes, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
			(void)hipGetLastError();
			return false;
		}
		void* d = nullptr;
		if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
			(void)hipGetLastError();
			(void)hipHostFree(h);
			return false;
		}
		host = static_cast<uint8_t*>(h);
		dev = static_cast<uint8_t*>(d);
		return true;
	}
};
// one per host thread AND device: the mapping is made under the device that is current (the filter's: every caller
// holds a DeviceGuard) and used for filters on that device only
Mailbox& mailbox()
{
	static thread_local std::map<int, Mailbox> boxes;
	int cur = 0;
	if (hipGetDevice(&cur) != hipSuccess)
		cur = 0;
	return boxes[cur];
}

uint64_t srol_n(uint64_t x, unsigned s)
{
	uint64_t lo = x & 0x1FFFFFFFFULL, hi = x >> 33;
	const unsigned a = s % 33, b = s % 31;
	if (a)
		lo = ((lo << a) | (lo >> (33 - a))) & 0x1FFFFFFFFULL;
	if (b)
		hi = ((hi << b) | (hi >> (31 - b))) & 0x7FFFFFFFULL;
	return (hi << 33) | lo;
}

const uint64_t kSeeds[4] = {kSeedA, kSeedC, kSeedG, kSeedT};
// forward / reverse-strand seed of a device base code (internal.hpp: codes 4..7 are raw bytes whose
// "complement" under c & cpOff is the byte itself)
uint64_t fwd_seed(unsigned c) { return kSeeds[c & 3]; }
uint64_t rev_seed(unsigned c) { return c < 4 ? kSeeds[c ^ 3] : kSeeds[c & 3]; }

void fill_hash_params(HashParams& hp, unsigned k, unsigned h)
{
	memset(&hp, 0, sizeof hp);
	hp.k = k;
	hp.h = h;
	hp.kms = (uint64_t)k * kMultiSeed;
	hp.use_pos_tab = k <= 128; // 128*k bytes of LDS; launchers may clear it when LDS is tight
	for (unsigned c = 0; c < kNumCodes; ++c) {
		const uint64_t s = fwd_seed(c), rc = rev_seed(c);
		hp.init_tab[c][0] = s;
		hp.init_tab[c][1] = srol_n(rc, k - 1);
		hp.in_tab[c][0] = s;
		hp.in_tab[c][1] = srol_n(rc, k);
		hp.out_tab[c][0] = srol_n(s, k);
		hp.out_tab[c][1] = rc;
	}
}

// spaced-seed tables on the device; on success the caller owns *d_pos / *d_dc
int build_spaced(HashParams& hp, const char* const* seeds, unsigned n_seeds, unsigned h2,
                 uint64_t** d_pos, uint16_t** d_dc)
{
	const unsigned k = hp.k;
	if (n_seeds == 0 || n_seeds > (unsigned)kMaxSeeds || h2 == 0)
		return fail(BTLBF_EINVAL, "spaced seeds: need 1..%d seeds and h2 >= 1", kMaxSeeds);
	if ((uint64_t)n_seeds * h2 > (uint64_t)kMaxHash)
		return fail(BTLBF_EINVAL, "spaced seeds: n_seeds*h2 = %u exceeds %d", n_seeds * h2, kMaxHash);
	if (k > 1024)
		return fail(BTLBF_EINVAL, "spaced seeds: k = %u > 1024 unsupported", k);
	std::vector<uint16_t> dc;
	for (unsigned j = 0; j < n_seeds; ++j) {
		if (!seeds[j] || strlen(seeds[j]) != k)
			return fail(BTLBF_EINVAL, "spaced seed %u must have exactly k = %u characters", j, k);
		hp.dc_off[j] = (uint32_t)dc.size();
		for (unsigned i = 0; i < k; ++i)
			if (seeds[j][i] != '1') // parseSeed keeps the indices of non-'1' (stHashIterator.hpp:27-29)
				dc.push_back((uint16_t)i);
	}
	hp.dc_off[n_seeds] = (uint32_t)dc.size();
	// the union list (internal.hpp HashParams::dcu): distinct offsets with their seed masks, those of every seed first
	std::vector<uint32_t> dcu;
	{
		std::map<unsigned, uint32_t> mask_of;
		for (unsigned j = 0; j < n_seeds; ++j)
			for (uint32_t d = hp.dc_off[j]; d < hp.dc_off[j + 1]; ++d)
				mask_of[dc[d]] |= 1u << j;
		const uint32_t all = n_seeds >= 32 ? 0xffffffffu : (1u << n_seeds) - 1;
		for (const auto& kv : mask_of)
			if (kv.second == all)
				dcu.push_back(kv.first | (kv.second << 16));
		hp.n_dcu_all = (uint32_t)dcu.size();
		// the others grouped by their mask, every group an even number of entries: the hash stage takes them in pairs
		// with one mask (seq_core.hpp); the filler is the "offset" k, the zero row behind the positional table
		std::map<uint32_t, std::vector<unsigned>> by_mask;
		for (const auto& kv : mask_of)
			if (kv.second != all)
				by_mask[kv.second].push_back(kv.first);
		for (const auto& g : by_mask) {
			for (unsigned off : g.second)
				dcu.push_back(off | (g.first << 16));
			if (g.second.size() % 2)
				dcu.push_back(k | (g.first << 16));
		}
		if (dcu.size() > kMaxDcu) {
			dcu.clear();
			hp.n_dcu_all = 0;
		}
		// two-base rows for the pairs (seq_core.hpp): the hash stage takes two pairs per trip, so an even number of
		// them -- a pair of fillers with an empty mask if need be; at most 16 rows (4 KB of LDS)
		hp.want_pair_rows = hp.n_pair_rows = 0;
		if (!dcu.empty()) {
			size_t pairs = (dcu.size() - hp.n_dcu_all) / 2;
			if (pairs % 2 && dcu.size() + 2 <= kMaxDcu) {
				dcu.push_back(k);
				dcu.push_back(k);
				++pairs;
			}
			if (pairs && pairs % 2 == 0 && pairs <= 16)
				hp.want_pair_rows = (uint32_t)pairs;
		}
		hp.n_dcu = (uint32_t)dcu.size();
	}
	const size_t dc_bytes = (dc.size() * 2 + 15) / 16 * 16;
	*d_pos = nullptr;
	HIP_TRY(hipMalloc((void**)d_dc, dc_bytes + dcu.size() * 4 + 16));
	if (!dc.empty())
		HIP_TRY(hipMemcpy(*d_dc, dc.data(), dc.size() * 2, hipMemcpyHostToDevice));
	if (!dcu.empty())
		HIP_TRY(hipMemcpy(reinterpret_cast<uint8_t*>(*d_dc) + dc_bytes, dcu.data(), dcu.size() * 4, hipMemcpyHostToDevice));
	hp.n_seeds = n_seeds;
	hp.h2 = h2;
	hp.h = n_seeds * h2;
	hp.use_pos_tab = 1; // spaced seeds are masked through the positional table (k <= 1024 checked above)
	hp.dc_idx = *d_dc;
	hp.dcu = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(*d_dc) + dc_bytes);
	return BTLBF_OK;
}

void fill_mod(ModParams& m, uint64_t size, uint64_t lo, uint64_t len)
{
	memset(&m, 0, sizeof m);
	m.size = size;
	m.pow2 = (size & (size - 1)) == 0;
	m.mask = size - 1;
	// floor(2^64 / size) for size >= 2 that is not a power of two == floor((2^64-1)/size)
	m.magic = m.pow2 ? 0 : (~0ULL) / size;
	m.magic32 = (m.magic >> 32) || (size >> 63) ? 0 : (uint32_t)m.magic;
	m.neg_size = 0 - size;
	m.shard_lo = lo;
	m.shard_len = len;
	m.shard_shift = 0xffffffffu;
	if (len && (len & (len - 1)) == 0) {
		unsigned s = 0;
		while ((1ULL << s) < len)
			++s;
		m.shard_shift = s;
	}
}

uint64_t cbf_round_bytes(uint64_t b) // CountingBloomFilter.hpp:40-49
{
	const uint64_t r = b % 8;
	return r ? b + 8 - r : b;
}

// ---- header text ---------------------------------------------------------------------------
std::string toml_double(double v) // cpptoml.h:3477-3494
{
	char buf[64];
	snprintf(buf, sizeof buf, "%#.17g", v);
	std::string s(buf);
	size_t p = s.find("e0");
	if (p != std::string::npos)
		s.replace(p, 2, "e");
	p = s.find("e-0");
	if (p != std::string::npos)
		s.replace(p, 3, "e-");
	return s;
}

// Key order: what libstdc++'s unordered_map yields for the reference's insertion order
// (cpptoml.h:43-52,3332; BloomFilter.hpp:275-281; CountingBloomFilter.hpp:355-359)
std::string header_text(const btlbf_filter* f)
{
	char buf[640];
	if (f->kind == BTLBF_BLOOM) {
		snprintf(buf, sizeof buf,
		         "[BTLBloomFilter_v1]\n\tnEntry = %llu\n\tdFPR = %s\n\tEntry = %llu\n"
		         "\tBloomFilterSizeInBytes = %llu\n\tBloomFilterSize = %llu\n\tHashNum = %u\n"
		         "\tKmerSize = %u\n[HeaderEnd]\n",
		         (unsigned long long)f->n_entry, toml_double(f->dfpr).c_str(),
		         (unsigned long long)f->t_entry, (unsigned long long)f->size_bytes,
		         (unsigned long long)f->size, f->h, f->k);
	} else {
		snprintf(buf, sizeof buf,
		         "[BTLCountingBloomFilter_v1]\n\tBloomFilterSize = %llu\n\tHashNum = %u\n"
		         "\tKmerSize = %u\n\tBloomFilterSizeInBytes = %llu\n\tBitsPerCounter = %u\n"
		         "[HeaderEnd]\n",
		         (unsigned long long)f->size, f->h, f->k, (unsigned long long)f->size_bytes,
		         f->bits_per_counter);
	}
	return buf;
}

std::string trim(const std::string& s)
{
	size_t a = s.find_first_not_of(" \t\r");
	if (a == std::string::npos)
		return "";
	size_t b = s.find_last_not_of(" \t\r");
	return s.substr(a, b - a + 1);
}

struct ParsedHeader {
	bool has[8] = {false};
	uint64_t size = 0, size_bytes = 0, n_entry = 0, t_entry = 0;
	unsigned h = 0, k = 0, bits_per_counter = 8;
	double dfpr = 0;
	size_t header_len = 0;
};

// Order-insensitive reader of the "key = value" lines between the magic line and [HeaderEnd]
// (the reference hands them to a TOML parser, BloomFilter.hpp:118-166).
int parse_header(FILE* fp, int kind, const char* path, ParsedHeader& out)
{
	const char* magic = kind == BTLBF_BLOOM ? "[BTLBloomFilter_v1]" : "[BTLCountingBloomFilter_v1]";
	std::string line;
	auto getline = [&](std::string& l) -> bool {
		l.clear();
		int c;
		bool any = false;
		while ((c = fgetc(fp)) != EOF) {
			any = true;
			out.header_len++;
			if (c == '\n')
				return true;
			l.push_back((char)c);
			if (l.size() > 4096)
				return true;
		}
		return any;
	};
	if (!getline(line) || line != magic)
		return fail(BTLBF_EFORMAT,
		            "%s: magic string does not match (likely version mismatch): got \"%.60s\", want \"%s\"",
		            path, line.c_str(), magic);
	bool end = false;
	while (getline(line)) {
		if (line == "[HeaderEnd]") {
			end = true;
			break;
		}
		const size_t eq = line.find('=');
		if (eq == std::string::npos)
			continue;
		const std::string key = trim(line.substr(0, eq)), val = trim(line.substr(eq + 1));
		if (key == "BloomFilterSize") {
			out.size = strtoull(val.c_str(), nullptr, 10);
			out.has[0] = true;
		} else if (key == "HashNum") {
			out.h = (unsigned)strtoul(val.c_str(), nullptr, 10);
			out.has[1] = true;
		} else if (key == "KmerSize") {
			out.k = (unsigned)strtoul(val.c_str(), nullptr, 10);
			out.has[2] = true;
		} else if (key == "BloomFilterSizeInBytes") {
			out.size_bytes = strtoull(val.c_str(), nullptr, 10);
			out.has[3] = true;
		} else if (key == "dFPR") {
			out.dfpr = strtod(val.c_str(), nullptr);
			out.has[4] = true;
		} else if (key == "nEntry") {
			out.n_entry = strtoull(val.c_str(), nullptr, 10);
			out.has[5] = true;
		} else if (key == "Entry") {
			out.t_entry = strtoull(val.c_str(), nullptr, 10);
			out.has[6] = true;
		} else if (key == "BitsPerCounter") {
			out.bits_per_counter = (unsigned)strtoul(val.c_str(), nullptr, 10);
			out.has[7] = true;
		}
	}
	if (!end)
		return fail(BTLBF_EFORMAT, "%s: pre-built bloom filter does not have the correct header end", path);
	const int need_bloom[] = {0, 1, 2, 3, 4, 5, 6}, need_cnt[] = {0, 1, 2, 3, 7};
	if (kind == BTLBF_BLOOM) {
		for (int i : need_bloom)
			if (!out.has[i])
				return fail(BTLBF_EFORMAT, "%s: header key missing", path);
	} else {
		for (int i : need_cnt)
			if (!out.has[i])
				return fail(BTLBF_EFORMAT, "%s: header key missing", path);
	}
	return BTLBF_OK;
}

int make_filter(btlbf_filter** out, int kind, uint64_t size, uint64_t size_bytes, unsigned shard_index,
                unsigned shard_count, unsigned h, unsigned k, unsigned thr, int device)
{
	if (!out)
		return fail(BTLBF_EINVAL, "null out pointer");
	*out = nullptr;
	if (kind != BTLBF_BLOOM && kind != BTLBF_COUNTING8)
		return fail(BTLBF_EINVAL, "unknown filter kind %d", kind);
	if (h == 0)
		return fail(BTLBF_EINVAL, "hash_num must be >= 1");
	if (k == 0 || k > 32768)
		return fail(BTLBF_EINVAL, "kmer_size must be in 1..32768");
	if (size < 8)
		return fail(BTLBF_EINVAL, "filter size %llu too small", (unsigned long long)size);
	if (shard_count == 0 || shard_index >= shard_count || size % shard_count ||
	    (shard_count > 1 && (size / shard_count) % 64))
		return fail(BTLBF_EINVAL, "shard %u of %u does not split %llu positions into multiples of 64",
		            shard_index, shard_count, (unsigned long long)size);
	if (btlbf_device_count() <= device || device < 0)
		return fail(BTLBF_EHIP, "no GPU %d (visible devices: %d): this library has no CPU path", device,
		            btlbf_device_count());
	DeviceGuard g(device);
	if (!g.ok)
		return fail(BTLBF_EHIP, "cannot select GPU %d", device);
	btlbf_filter* f = new btlbf_filter();
	f->kind = kind;
	f->device = device;
	f->size = size;
	f->size_bytes = size_bytes;
	f->h = h;
	f->k = k;
	f->thr = thr;
	f->shard_index = shard_index;
	f->shard_count = shard_count;
	const uint64_t len = size / shard_count;
	f->local_bytes = kind == BTLBF_BLOOM ? len / 8 : len;
	if (shard_count == 1)
		f->local_bytes = size_bytes;
	f->alloc_bytes = (f->local_bytes + 15) / 16 * 16;
	fill_mod(f->mod, size, (uint64_t)shard_index * len, len);
	fill_hash_params(f->hp, k, h);
	if (const char* m = getenv("BTLBF_INSERT_MODE")) {
		if (!strcmp(m, "direct"))
			f->insert_mode = BTLBF_INSERT_DIRECT;
		else if (!strcmp(m, "partitioned"))
			f->insert_mode = BTLBF_INSERT_PARTITIONED;
	}
	if (const char* m = getenv("BTLBF_QUERY_MODE")) {
		if (!strcmp(m, "direct"))
			f->query_mode = BTLBF_INSERT_DIRECT;
		else if (!strcmp(m, "partitioned"))
			f->query_mode = BTLBF_INSERT_PARTITIONED;
	}
	hipError_t e = hipMalloc(&f->d_data, f->alloc_bytes);
	if (e != hipSuccess) {
		delete f;
		return fail(e == hipErrorOutOfMemory ? BTLBF_ENOMEM : BTLBF_EHIP, "hipMalloc(%llu bytes): %s",
		            (unsigned long long)f->alloc_bytes, hipGetErrorString(e));
	}
	f->lazy_zero = true; // a new filter is a cleared filter: zeroed by whoever touches the array first
	e = hipMalloc((void**)&f->d_scalar, 64);
	if (e == hipSuccess)
		e = hipDeviceSynchronize();
	if (e != hipSuccess) {
		(void)hipFree(f->d_data);
		delete f;
		return fail(BTLBF_EHIP, "filter initialisation: %s", hipGetErrorString(e));
	}
	*out = f;
	return BTLBF_OK;
}

// device-resident view of a caller's sequence buffer (+ layout), staging host memory if needed
struct SeqView {
	DevBuf seq_buf, starts_buf;
	const uint8_t* d_seq = nullptr;
	LayoutParams lay{nullptr, 0, 0};
};

int check_layout(const btlbf_layout* l, uint64_t len)
{
	if (!l)
		return BTLBF_OK;
	if (l->starts == nullptr && l->read_len && len % l->read_len)
		return fail(BTLBF_EINVAL, "len %llu is not a multiple of read_len %u", (unsigned long long)len,
		            l->read_len);
	return BTLBF_OK;
}

int make_view(SeqView& v, const char* seq, uint64_t len, const btlbf_layout* l, int mem, hipStream_t s)
{
	if (len && !seq)
		return fail(BTLBF_EINVAL, "null sequence buffer");
	int rc = check_layout(l, len);
	if (rc)
		return rc;
	if (l) {
		v.lay.n_seqs = l->n_seqs;
		v.lay.read_len = l->starts ? 0 : l->read_len;
	}
	if (mem == BTLBF_DEVICE) {
		v.d_seq = reinterpret_cast<const uint8_t*>(seq);
		if (l && l->starts)
			v.lay.starts = l->starts;
		return BTLBF_OK;
	}
	if (mem != BTLBF_HOST)
		return fail(BTLBF_EINVAL, "mem must be BTLBF_HOST or BTLBF_DEVICE");
	HIP_TRY(v.seq_buf.alloc_pooled(len + 16));
	if (len)
		HIP_TRY(hipMemcpyAsync(v.seq_buf.p, seq, len, hipMemcpyHostToDevice, s));
	v.d_seq = v.seq_buf.as<uint8_t>();
	if (l && l->starts) {
		if (l->starts[0] != 0 || l->starts[l->n_seqs] != len)
			return fail(BTLBF_EINVAL, "starts[0] must be 0 and starts[n_seqs] must equal len");
		HIP_TRY(v.starts_buf.alloc_pooled((l->n_seqs + 1) * 8));
		HIP_TRY(hipMemcpyAsync(v.starts_buf.p, l->starts, (l->n_seqs + 1) * 8, hipMemcpyHostToDevice, s));
		v.lay.starts = v.starts_buf.as<uint64_t>();
	}
	return BTLBF_OK;
}

SeqArgs base_args(const btlbf_filter* f, const SeqView& v, uint64_t len)
{
	SeqArgs a;
	memset(&a, 0, sizeof a);
	a.seq = v.d_seq;
	a.len = len;
	a.layout = v.lay;
	a.filter = f->d_data;
	a.mod = f->mod;
	a.hp = f->hp;
	a.threshold = f->thr;
	return a;
}

uint64_t bitmap_bytes(uint64_t len) { return (len + 63) / 64 * 8; }

} // namespace

// -------------------------------------------------------------------------------------------------
// lifetime
// -------------------------------------------------------------------------------------------------
extern "C" int btlbf_create(btlbf_filter** out, int kind, uint64_t size, unsigned hash_num,
                            unsigned kmer_size, unsigned threshold, int device)
{
	if (kind == BTLBF_BLOOM) {
		if (size % 8 != 0) // BloomFilter.hpp:391-394
			return fail(BTLBF_EINVAL, "ERROR: Filter Size \"%llu\" is not a multiple of 8.",
			            (unsigned long long)size);
		return make_filter(out, kind, size, size / 8, 0, 1, hash_num, kmer_size, 0, device);
	}
	const uint64_t bytes = cbf_round_bytes(size);
	return make_filter(out, kind, bytes, bytes, 0, 1, hash_num, kmer_size, threshold, device);
}

extern "C" int btlbf_create_shard(btlbf_filter** out, int kind, uint64_t global_size,
                                  unsigned shard_index, unsigned shard_count, unsigned hash_num,
                                  unsigned kmer_size, unsigned threshold, int device)
{
	if (kind == BTLBF_BLOOM) {
		if (global_size % 8 != 0)
			return fail(BTLBF_EINVAL, "ERROR: Filter Size \"%llu\" is not a multiple of 8.",
			            (unsigned long long)global_size);
		return make_filter(out, kind, global_size, global_size / 8, shard_index, shard_count, hash_num,
		                   kmer_size, 0, device);
	}
	const uint64_t bytes = cbf_round_bytes(global_size);
	return make_filter(out, kind, bytes, bytes, shard_index, shard_count, hash_num, kmer_size, threshold,
	                   device);
}

extern "C" int btlbf_destroy(btlbf_filter* f)
{
	if (!f)
		return BTLBF_OK;
	DeviceGuard g(f->device);
	if (f->clear_ev)
		(void)hipEventDestroy(f->clear_ev);
	if (f->zero_ev)
		(void)hipEventDestroy(f->zero_ev);
	(void)hipFree(f->d_data);
	(void)hipFree(f->d_scalar);
	(void)hipFree(f->d_pos_tab);
	(void)hipFree(f->d_dc_idx);
	(void)hipFree(f->d_part);
	(void)hipFree(f->d_split);
	(void)hipFree(f->d_flags);
	delete f;
	return BTLBF_OK;
}

extern "C" int btlbf_set_insert_mode(btlbf_filter* f, int mode, uint64_t scratch_bytes)
{
	FilterLock lk__(f);
	if (!f || mode < BTLBF_INSERT_AUTO || mode > BTLBF_INSERT_PARTITIONED)
		return fail(BTLBF_EINVAL, "bad insert mode");
	f->insert_mode = mode;
	f->part_budget = scratch_bytes;
	return BTLBF_OK;
}

extern "C" int btlbf_release_scratch(btlbf_filter* f)
{
	FilterLock lk__(f);
	if (!f)
		return fail(BTLBF_EINVAL, "null filter");
	DeviceGuard g(f->device);
	(void)hipFree(f->d_part); // synchronises with work in flight
	f->d_part = nullptr;
	f->part_bytes = 0;
	(void)hipFree(f->d_split);
	f->d_split = nullptr;
	f->split_bytes = 0;
	(void)hipFree(f->d_flags);
	f->d_flags = nullptr;
	f->flags_bytes = 0;
	dev_pool().drain(f->device); // parked staging buffers of HOST-mode calls (every user has synchronised)
	return BTLBF_OK;
}

extern "C" int btlbf_set_profiling(btlbf_filter* f, int on)
{
	FilterLock lk__(f);
	if (!f)
		return fail(BTLBF_EINVAL, "null filter");
	f->profiling = on != 0;
	return BTLBF_OK;
}

extern "C" int btlbf_get_profile(btlbf_filter* f, double* ms, unsigned* calls, int reset)
{
	FilterLock lk__(f);
	if (!f || !ms || !calls)
		return fail(BTLBF_EINVAL, "null argument");
	DeviceGuard g(f->device);
	for (auto& sp : f->spans) {
		float t = 0;
		if (hipEventSynchronize(sp.e1) == hipSuccess && hipEventElapsedTime(&t, sp.e0, sp.e1) == hipSuccess) {
			f->prof_ms[sp.slot] += t;
			f->prof_calls[sp.slot] += 1;
		}
		(void)hipEventDestroy(sp.e0);
		(void)hipEventDestroy(sp.e1);
	}
	f->spans.clear();
	for (int i = 0; i < BTLBF_PROF_SLOTS; ++i) {
		ms[i] = f->prof_ms[i];
		calls[i] = f->prof_calls[i];
		if (reset) {
			f->prof_ms[i] = 0;
			f->prof_calls[i] = 0;
		}
	}
	return BTLBF_OK;
}

extern "C" int btlbf_set_query_mode(btlbf_filter* f, int mode)
{
	FilterLock lk__(f);
	if (!f || mode < BTLBF_INSERT_AUTO || mode > BTLBF_INSERT_PARTITIONED)
		return fail(BTLBF_EINVAL, "bad query mode");
	f->query_mode = mode;
	return BTLBF_OK;
}

extern "C" int btlbf_set_spaced_seeds(btlbf_filter* f, const char* const* seeds, unsigned n_seeds,
                                      unsigned h2)
{
	FilterLock lk__(f);
	if (!f || !seeds)
		return fail(BTLBF_EINVAL, "null argument");
	if (n_seeds * h2 != f->h)
		return fail(BTLBF_EINVAL, "n_seeds*h2 = %u must equal the filter's hash_num %u", n_seeds * h2, f->h);
	DeviceGuard g(f->device);
	HashParams hp;
	fill_hash_params(hp, f->k, f->h);
	uint64_t* dp = nullptr;
	uint16_t* dd = nullptr;
	int rc = build_spaced(hp, seeds, n_seeds, h2, &dp, &dd);
	if (rc) {
		(void)hipFree(dp);
		(void)hipFree(dd);
		return rc;
	}
	(void)hipFree(f->d_pos_tab);
	(void)hipFree(f->d_dc_idx);
	f->d_pos_tab = dp;
	f->d_dc_idx = dd;
	f->hp = hp;
	return BTLBF_OK;
}

// -------------------------------------------------------------------------------------------------
// attributes
// -------------------------------------------------------------------------------------------------
extern "C" int btlbf_kind(const btlbf_filter* f) { return f->kind; }
extern "C" uint64_t btlbf_size(const btlbf_filter* f) { return f->size; }
extern "C" uint64_t btlbf_size_bytes(const btlbf_filter* f) { return f->size_bytes; }
extern "C" uint64_t btlbf_local_bytes(const btlbf_filter* f) { return f->local_bytes; }
extern "C" unsigned btlbf_hash_num(const btlbf_filter* f) { return f->h; }
extern "C" unsigned btlbf_kmer_size(const btlbf_filter* f) { return f->k; }
extern "C" unsigned btlbf_threshold(const btlbf_filter* f) { return f->thr; }
extern "C" uint64_t btlbf_get_n_entry(const btlbf_filter* f) { return f->n_entry; }
extern "C" uint64_t btlbf_get_t_entry(const btlbf_filter* f) { return f->t_entry; }
extern "C" void btlbf_set_n_entry(btlbf_filter* f, uint64_t v) { f->n_entry = v; }
extern "C" void btlbf_set_t_entry(btlbf_filter* f, uint64_t v) { f->t_entry = v; }
extern "C" void* btlbf_device_ptr(const btlbf_filter* f_)
{
	if (!f_)
		return nullptr;
	btlbf_filter* f = const_cast<btlbf_filter*>(f_);
	FilterLock lk__(f);
	if (f->lazy_zero) { // a caller that looks at the raw array must see a pending clear
		DeviceGuard g(f->device);
		(void)materialize_clear(f, nullptr);
		(void)hipDeviceSynchronize();
	}
	f->ptr_exposed = true; // from now on btlbf_clear zeroes eagerly, on its stream: the pointer may be kept
	return f->d_data;
}
extern "C" int btlbf_device(const btlbf_filter* f) { return f->device; }

extern "C" int btlbf_clear(btlbf_filter* f, void* stream)
{
	FilterLock lk__(f);
	if (!f)
		return fail(BTLBF_EINVAL, "null filter");
	DeviceGuard g(f->device);
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (f->ptr_exposed) { // someone may hold the raw pointer: zero now, in stream order
		f->lazy_zero = false;
		f->clear_ev_pending = false;
		HIP_TRY(hipMemsetAsync(f->d_data, 0, f->alloc_bytes, s));
		if (!f->zero_ev)
			HIP_TRY(hipEventCreateWithFlags(&f->zero_ev, hipEventDisableTiming));
		HIP_TRY(hipEventRecord(f->zero_ev, s));
		f->zero_stream = s;
		f->zero_ev_pending = true;
		return BTLBF_OK;
	}
	// lazy: zeroed by whoever touches the array next (see btlbf_filter::lazy_zero), after this point of `stream`
	if (!f->clear_ev)
		HIP_TRY(hipEventCreateWithFlags(&f->clear_ev, hipEventDisableTiming));
	HIP_TRY(hipEventRecord(f->clear_ev, s));
	f->clear_ev_pending = true;
	f->lazy_zero = true;
	return BTLBF_OK;
}

extern "C" int btlbf_upload(btlbf_filter* f, const void* src, uint64_t offset, uint64_t nbytes)
{
	FilterLock lk__(f);
	if (!f || (!src && nbytes))
		return fail(BTLBF_EINVAL, "null argument");
	if (offset + nbytes > f->local_bytes)
		return fail(BTLBF_EINVAL, "upload range exceeds the filter");
	DeviceGuard g(f->device);
	MATERIALIZE(f, nullptr);
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipMemcpy(static_cast<uint8_t*>(f->d_data) + offset, src, nbytes, hipMemcpyHostToDevice));
	return BTLBF_OK;
}

extern "C" int btlbf_download(const btlbf_filter* f, void* dst, uint64_t offset, uint64_t nbytes)
{
	FilterLock lk__(f);
	if (!f || (!dst && nbytes))
		return fail(BTLBF_EINVAL, "null argument");
	if (offset + nbytes > f->local_bytes)
		return fail(BTLBF_EINVAL, "download range exceeds the filter");
	DeviceGuard g(f->device);
	MATERIALIZE(f, nullptr);
	HIP_TRY(hipDeviceSynchronize()); // DEVICE-mode calls may have run on non-blocking user streams
	HIP_TRY(hipMemcpy(dst, static_cast<const uint8_t*>(f->d_data) + offset, nbytes, hipMemcpyDeviceToHost));
	return BTLBF_OK;
}

// -------------------------------------------------------------------------------------------------
// files
// -------------------------------------------------------------------------------------------------
extern "C" int btlbf_header(const btlbf_filter* f, char* buf, size_t cap, size_t* len)
{
	FilterLock lk__(f);
	if (!f)
		return fail(BTLBF_EINVAL, "null filter");
	const std::string h = header_text(f);
	if (len)
		*len = h.size();
	if (buf) {
		if (cap < h.size())
			return fail(BTLBF_EINVAL, "header buffer too small (%zu < %zu)", cap, h.size());
		memcpy(buf, h.data(), h.size());
	}
	return BTLBF_OK;
}

extern "C" int btlbf_load(btlbf_filter** out, int kind, const char* path, unsigned threshold, int device)
{
	if (!out || !path)
		return fail(BTLBF_EINVAL, "null argument");
	*out = nullptr;
	FILE* fp = fopen(path, "rb");
	if (!fp)
		return fail(BTLBF_EIO, "error: `%s': %s", path, strerror(errno));
	ParsedHeader ph;
	int rc = parse_header(fp, kind, path, ph);
	if (rc) {
		fclose(fp);
		return rc;
	}
	btlbf_filter* f = nullptr;
	if (kind == BTLBF_BLOOM) {
		if (ph.size % 8 != 0) {
			fclose(fp);
			return fail(BTLBF_EINVAL, "ERROR: Filter Size \"%llu\" is not a multiple of 8.",
			            (unsigned long long)ph.size);
		}
		rc = make_filter(&f, kind, ph.size, ph.size / 8, 0, 1, ph.h, ph.k, 0, device);
	} else {
		if (ph.bits_per_counter != 8 || ph.size != ph.size_bytes) {
			fclose(fp);
			return fail(BTLBF_EFORMAT, "%s: only 8-bit counters are supported (BitsPerCounter = %u)", path,
			            ph.bits_per_counter);
		}
		rc = make_filter(&f, kind, ph.size, ph.size_bytes, 0, 1, ph.h, ph.k, threshold, device);
	}
	if (rc) {
		fclose(fp);
		return rc;
	}
	f->dfpr = ph.dfpr;
	f->n_entry = ph.n_entry;
	f->t_entry = ph.t_entry;
	{
		DeviceGuard g0(device);
		if (materialize_clear(f, nullptr) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
			fclose(fp);
			btlbf_destroy(f);
			return fail(BTLBF_EHIP, "clearing the filter failed");
		}
	}
	// body: stream through a pinned bounce buffer
	const size_t chunk = 64u << 20;
	void* bounce = nullptr;
	DeviceGuard g(device);
	if (hipHostMalloc(&bounce, chunk, hipHostMallocDefault) != hipSuccess) {
		fclose(fp);
		btlbf_destroy(f);
		return fail(BTLBF_ENOMEM, "pinned bounce buffer");
	}
	uint64_t done = 0;
	while (done < f->local_bytes) {
		const size_t n = (size_t)std::min<uint64_t>(chunk, f->local_bytes - done);
		if (fread(bounce, 1, n, fp) != n) {
			(void)hipHostFree(bounce);
			fclose(fp);
			btlbf_destroy(f);
			return fail(BTLBF_EIO, "error: `%s': short read of the filter body", path);
		}
		if (hipMemcpy(static_cast<uint8_t*>(f->d_data) + done, bounce, n, hipMemcpyHostToDevice) != hipSuccess) {
			(void)hipHostFree(bounce);
			fclose(fp);
			btlbf_destroy(f);
			return fail(BTLBF_EHIP, "upload of the filter body failed");
		}
		done += n;
	}
	(void)hipHostFree(bounce);
	fclose(fp);
	*out = f;
	return BTLBF_OK;
}

// A zeroed filter with the geometry and the bookkeeping fields of a header text (everything up to and including
// the "[HeaderEnd]" line): what the reference's public loadHeader(std::istream&) leaves behind
// (BloomFilter.hpp:118-166, CountingBloomFilter.hpp:84,282-343) before loadFilter reads the body.
extern "C" int btlbf_create_from_header(btlbf_filter** out, int kind, const char* header, size_t len, unsigned threshold,
                                        int device)
{
	if (!out || !header)
		return fail(BTLBF_EINVAL, "null argument");
	*out = nullptr;
	FILE* fp = fmemopen(const_cast<char*>(header), len, "rb");
	if (!fp)
		return fail(BTLBF_EIO, "fmemopen: %s", strerror(errno));
	ParsedHeader ph;
	int rc = parse_header(fp, kind, "<header>", ph);
	fclose(fp);
	if (rc)
		return rc;
	btlbf_filter* f = nullptr;
	if (kind == BTLBF_BLOOM) {
		if (ph.size % 8 != 0)
			return fail(BTLBF_EINVAL, "ERROR: Filter Size \"%llu\" is not a multiple of 8.", (unsigned long long)ph.size);
		rc = make_filter(&f, kind, ph.size, ph.size / 8, 0, 1, ph.h, ph.k, 0, device);
	} else {
		if (ph.bits_per_counter != 8 || ph.size != ph.size_bytes)
			return fail(BTLBF_EFORMAT, "only 8-bit counters are supported (BitsPerCounter = %u)", ph.bits_per_counter);
		rc = make_filter(&f, kind, ph.size, ph.size_bytes, 0, 1, ph.h, ph.k, threshold, device);
	}
	if (rc)
		return rc;
	f->dfpr = ph.dfpr;
	f->n_entry = ph.n_entry;
	f->t_entry = ph.t_entry;
	*out = f;
	return BTLBF_OK;
}

extern "C" double btlbf_get_dfpr(const btlbf_filter* f) { return f ? f->dfpr : 0.0; }
extern "C" void btlbf_set_dfpr(btlbf_filter* f, double v)
{
	if (f)
		f->dfpr = v;
}

static int write_body(const btlbf_filter* f, int fd, uint64_t file_off, const char* path)
{
	const size_t chunk = 64u << 20;
	void* bounce = nullptr;
	if (hipHostMalloc(&bounce, chunk, hipHostMallocDefault) != hipSuccess)
		return fail(BTLBF_ENOMEM, "pinned bounce buffer");
	uint64_t done = 0;
	int rc = BTLBF_OK;
	while (done < f->local_bytes && rc == BTLBF_OK) {
		const size_t n = (size_t)std::min<uint64_t>(chunk, f->local_bytes - done);
		if (hipMemcpy(bounce, static_cast<const uint8_t*>(f->d_data) + done, n, hipMemcpyDeviceToHost) !=
		    hipSuccess) {
			rc = fail(BTLBF_EHIP, "download of the filter body failed");
			break;
		}
		size_t w = 0;
		while (w < n) {
			ssize_t r = pwrite(fd, static_cast<const char*>(bounce) + w, n - w, (off_t)(file_off + done + w));
			if (r <= 0) {
				rc = fail(BTLBF_EIO, "error: `%s': %s", path, strerror(errno));
				break;
			}
			w += (size_t)r;
		}
		done += n;
	}
	(void)hipHostFree(bounce);
	return rc;
}

extern "C" int btlbf_store_shard(btlbf_filter* f, const char* path)
{
	FilterLock lk__(f);
	if (!f || !path)
		return fail(BTLBF_EINVAL, "null argument");
	DeviceGuard g(f->device);
	MATERIALIZE(f, nullptr);
	HIP_TRY(hipDeviceSynchronize());
	const std::string hdr = header_text(f);
	const int flags = O_WRONLY | O_CREAT | (f->shard_count == 1 ? O_TRUNC : 0);
	const int fd = open(path, flags, 0644);
	if (fd < 0)
		return fail(BTLBF_EIO, "error: `%s': %s", path, strerror(errno));
	int rc = BTLBF_OK;
	if (f->shard_index == 0) {
		if (pwrite(fd, hdr.data(), hdr.size(), 0) != (ssize_t)hdr.size())
			rc = fail(BTLBF_EIO, "error: `%s': %s", path, strerror(errno));
		if (rc == BTLBF_OK && f->shard_count > 1 && ftruncate(fd, (off_t)(hdr.size() + f->size_bytes)) != 0)
			rc = fail(BTLBF_EIO, "error: `%s': %s", path, strerror(errno));
	}
	if (rc == BTLBF_OK)
		rc = write_body(f, fd, hdr.size() + (uint64_t)f->shard_index * f->local_bytes, path);
	if (close(fd) != 0 && rc == BTLBF_OK)
		rc = fail(BTLBF_EIO, "error: `%s': %s", path, strerror(errno));
	return rc;
}

extern "C" int btlbf_store(btlbf_filter* f, const char* path)
{
	FilterLock lk__(f);
	if (f && f->shard_count != 1)
		return fail(BTLBF_EINVAL, "btlbf_store on a shard: use btlbf_store_shard");
	return btlbf_store_shard(f, path);
}

// -------------------------------------------------------------------------------------------------
// the hot path
// -------------------------------------------------------------------------------------------------
namespace {

// defined further down, next to the partitioned insert
int partitioned_contains(btlbf_filter* f, const SeqArgs& base, uint8_t* hit_bits, uint8_t* valid_bits,
                         uint64_t* counts, hipStream_t s, bool* done, bool defer_hit_count = false);
int want_partitioned_query(btlbf_filter* f, const SeqArgs& base, hipStream_t s, bool* yes);
int split_contains(btlbf_filter* f, const SeqArgs& a, int direct_op, hipStream_t s, int* decided);

int seq_precheck(const btlbf_filter* f, uint64_t len)
{
	if (!f)
		return fail(BTLBF_EINVAL, "null filter");
	if (f->hp.n_seeds == 0 && f->h > 64)
		return fail(BTLBF_EINVAL, "hash_num %u > 64 unsupported by the sequence kernels", f->h);
	(void)len;
	return BTLBF_OK;
}

// copy a device bitmap / array back to the caller when the call was BTLBF_HOST
struct OutBuf {
	DevBuf dev;
	void* host = nullptr;
	size_t n = 0;
	void* d = nullptr;
	int prepare(void* user, size_t nbytes, int mem, bool zero, hipStream_t s)
	{
		n = nbytes;
		if (!user)
			return BTLBF_OK;
		if (mem == BTLBF_DEVICE) {
			d = user;
		} else {
			host = user;
			HIP_TRY(dev.alloc_pooled(nbytes));
			d = dev.p;
		}
		if (zero && nbytes)
			HIP_TRY(hipMemsetAsync(d, 0, nbytes, s));
		return BTLBF_OK;
	}
	int finish(hipStream_t s)
	{
		if (host && n)
			HIP_TRY(hipMemcpyAsync(host, d, n, hipMemcpyDeviceToHost, s));
		return BTLBF_OK;
	}
};

int run_query_like(btlbf_filter* f, int op, const char* seq, uint64_t len, const btlbf_layout* layout,
                   uint64_t* hit_bits, uint64_t* valid_bits, uint64_t* counts, uint8_t* min_out, int mem,
                   void* stream, FilterLock* lk = nullptr)
{
	int rc = seq_precheck(f, len);
	if (rc)
		return rc;
	// contains() on a shard answers for the probes inside its window (ShardedBloomFilter's gather mode
	// ANDs the shards' answers); the other query flavours need all h probes of a k-mer
	if (f->shard_count != 1) {
		if (op != OP_BF_CONTAINS && !(op == OP_CBF_QUERY && !min_out))
			return fail(BTLBF_EINVAL, "this query on a shard goes through btlbf_positions_seqs/btlbf_test_positions");
		if (op == OP_BF_CONTAINS)
			op = OP_BF_CONTAINS_WIN;
	}
	DeviceGuard g(f->device);
	hipStream_t s = static_cast<hipStream_t>(stream);
	MATERIALIZE(f, s);
	// a short sequence from host memory (the shims' per-read containsSeq / countSeq): through the calling thread's
	// pinned mailbox with the direct kernel -- one launch, one synchronisation, no staging (such a batch is far
	// below what the partitioned path takes)
	{
		const uint64_t up16 = ~(uint64_t)15, bm = (bitmap_bytes(len) + 15) & up16;
		const uint64_t o_hit = (len + 16 + 15) & up16, o_valid = o_hit + bm, o_cnt = o_valid + bm, o_min = o_cnt + 16,
		               o_end = o_min + ((len + 15) & up16);
		if (mem == BTLBF_HOST && len && len <= 65536 && (!layout || !layout->starts) && o_end <= Mailbox::kBytes &&
		    mailbox().get()) {
			if ((rc = check_layout(layout, len)))
				return rc;
			Mailbox& mb = mailbox();
			memcpy(mb.host, seq, len);
			SeqView v;
			v.d_seq = mb.dev;
			v.lay.read_len = layout ? layout->read_len : 0;
			SeqArgs a = base_args(f, v, len);
			// {clean windows, hits} are counted on the host from the two bitmaps (no atomics into host memory)
			a.hit_bits = hit_bits || counts ? mb.dev + o_hit : nullptr;
			a.valid_bits = valid_bits || counts ? mb.dev + o_valid : nullptr;
			a.min_out = min_out ? mb.dev + o_min : nullptr;
			REQUIRE_MATERIALIZED(f);
			HIP_TRY(launch_seq_op(op, a, s));
			if (lk && op != OP_BF_INSERT_CHECK)
				lk->release(); // a read-only call only waits from here on (its mailbox is the calling thread's own)
			HIP_TRY(hipStreamSynchronize(s));
			if (hit_bits)
				memcpy(hit_bits, mb.host + o_hit, bitmap_bytes(len));
			if (valid_bits)
				memcpy(valid_bits, mb.host + o_valid, bitmap_bytes(len));
			if (counts) {
				counts[0] = counts[1] = 0;
				const uint64_t* hb = reinterpret_cast<const uint64_t*>(mb.host + o_hit);
				const uint64_t* vb = reinterpret_cast<const uint64_t*>(mb.host + o_valid);
				for (uint64_t i = 0; i < bitmap_bytes(len) / 8; ++i) {
					counts[0] += (uint64_t)__builtin_popcountll(vb[i]);
					counts[1] += (uint64_t)__builtin_popcountll(hb[i]);
				}
			}
			if (min_out)
				memcpy(min_out, mb.host + o_min, len);
			return BTLBF_OK;
		}
	}
	SeqView v;
	rc = make_view(v, seq, len, layout, mem, s);
	if (rc)
		return rc;
	OutBuf ob_hit, ob_valid, ob_cnt, ob_min;
	if ((rc = ob_hit.prepare(hit_bits, bitmap_bytes(len), mem, false, s)))
		return rc;
	if ((rc = ob_valid.prepare(valid_bits, bitmap_bytes(len), mem, false, s)))
		return rc;
	if ((rc = ob_cnt.prepare(counts, 16, mem, true, s)))
		return rc;
	if ((rc = ob_min.prepare(min_out, len, mem, false, s)))
		return rc;
	SeqArgs a = base_args(f, v, len);
	a.hit_bits = static_cast<uint8_t*>(ob_hit.d);
	a.valid_bits = static_cast<uint8_t*>(ob_valid.d);
	a.counts = static_cast<uint64_t*>(ob_cnt.d);
	a.min_out = static_cast<uint8_t*>(ob_min.d);
	bool done = false;
	if (op == OP_BF_CONTAINS || op == OP_BF_CONTAINS_WIN || (op == OP_CBF_QUERY && !min_out)) { // minimum counts need the values: direct
		bool yes = false;
		int decided = 0; // split_contains: 0 = not applicable, 1 = direct, 2 = partitioned, 3 = done
		if (op != OP_BF_CONTAINS_WIN && (rc = split_contains(f, a, op, s, &decided)))
			return rc;
		done = decided == 3;
		yes = decided == 2;
		if (decided == 0 && (rc = want_partitioned_query(f, a, s, &yes)))
			return rc;
		if (yes) {
			DevBuf tmp_hit; // the partitioned path needs a hit bitmap to refine even if the caller wants counts only
			uint8_t* hb = a.hit_bits;
			if (!hb) {
				HIP_TRY(tmp_hit.alloc(bitmap_bytes(len) + 16));
				hb = tmp_hit.as<uint8_t>();
			}
			SeqArgs b = a;
			b.hit_bits = nullptr;
			b.valid_bits = nullptr;
			b.counts = nullptr;
			if ((rc = partitioned_contains(f, b, hb, a.valid_bits, a.counts, s, &done)))
				return rc;
			if (done && !a.hit_bits)
				HIP_TRY(hipStreamSynchronize(s)); // tmp_hit is freed on return
		}
	}
	if (!done) {
		REQUIRE_MATERIALIZED(f);
		ProfSpan ps(f, op == OP_BF_CONTAINS || op == OP_BF_CONTAINS_WIN ? BTLBF_PROF_QUERY_DIRECT : BTLBF_PROF_OTHER, s);
		HIP_TRY(launch_seq_op(op, a, s));
	}
	if ((rc = ob_hit.finish(s)) || (rc = ob_valid.finish(s)) || (rc = ob_cnt.finish(s)) ||
	    (rc = ob_min.finish(s)))
		return rc;
	if (mem == BTLBF_HOST)
		HIP_TRY(hipStreamSynchronize(s));
	return BTLBF_OK;
}

} // namespace

namespace {

// partitioned query: room for the failed positions of one batch and their hash set
static constexpr uint64_t kFailCap = 4ull << 20;          // entries
static constexpr uint64_t kFailTableSlots = 2 * kFailCap; // power of two
static constexpr uint64_t kFailBytes = 256 + kFailCap * 8 + kFailTableSlots * 8;
// entries a FRESH insert batch (partitioned_insert) may report as explicit positions instead of staging them
static constexpr uint64_t kFreshSpillCap = 16ull << 20;

// The tail of the partition scratch: a query's fail list + failed-position table, or a fresh insert's spill list.
// Inserts (fresh or not) and queries reserve the same tail, so that alternating between them never changes the
// scratch size (a re-allocation of ~100 GB costs seconds); a caller-imposed budget below 2 GiB gets short lists
// (more than a list holds and the batch is redone the plain way, which is always correct).
struct PartTail {
	uint64_t fail_cap, table_slots, spill_cap, bytes;
};
static PartTail part_tail(uint64_t budget)
{
	PartTail t;
	const bool full = budget >= (2ull << 30);
	t.fail_cap = full ? kFailCap : 256ull << 10;
	t.spill_cap = full ? kFreshSpillCap : 512ull << 10;
	t.table_slots = 2 * t.fail_cap;
	t.bytes = std::max<uint64_t>(256 + t.fail_cap * 8 + t.table_slots * 8, 256 + t.spill_cap * 8);
	return t;
}

// one partition level: bins of 2^shift positions, written as regions of `cap` chunks
struct PartLevel {
	uint32_t bins = 0;    // bins at this level (covering the local array)
	uint32_t P = 0;       // bins per writer block (pass A: all of them; split: the fan-out)
	uint32_t regions = 0; // writers per bin
	uint32_t cap = 0;     // chunks per region
	uint32_t shift = 0;   // log2(positions per bin)
	uint32_t wseg = 0;    // level 0 only: bins of `wseg` segments each instead (plan_level0); `shift` is then unused
	uint32_t alloc_bins = 0; // bins the arrays hold at a time (== bins unless the level is processed in groups)
	uint64_t cnt_bytes = 0, ent_bytes = 0;
	uint32_t* cnt = nullptr;
	uint32_t* ent = nullptr;
	PartOut out() const { return PartOut{P, regions, cap, cnt, ent}; }
	PartIn in() const { return PartIn{1, alloc_bins, regions, cap, cnt, ent}; }
};

struct PartPlan {
	uint32_t seg_shift = 19;
	uint64_t n_seg = 0;
	uint32_t group_bins = 0; // level-0 bins split + applied together (one-split plans); 0 = all at once
	int n_levels = 0; // lv[0] = pass A output (or the exchanged data), lv[1..] = split outputs
	PartLevel lv[3];
	uint64_t tiles_per_batch = 0;
	uint64_t bytes_total = 0;
	// pass A's overlapped schedule keeps the entries that find their ring full in a late image per workgroup and
	// round parity (part_hash_inst.hip): [regions][2][late_cap] words behind the tail of the scratch
	uint32_t* late_buf = nullptr;
	uint32_t late_cap = 0;
};

// chunks a region needs for `mean_entries` expected entries (Poisson: mean + 8 sigma) plus the
// partially filled chunk flushed at kernel end
uint32_t chunks_for(double mean_entries, uint32_t tail_chunks)
{
	const double m = mean_entries + 8.0 * std::sqrt(mean_entries + 1.0) + 32.0;
	return (uint32_t)std::min<double>(4.0e9, std::ceil(m / (double)kChunk)) + tail_chunks;
}

unsigned ceil_log2(uint64_t x)
{
	unsigned b = 0;
	while ((1ull << b) < x)
		++b;
	return b;
}

// mloc = positions held locally; unit_shift = log2(positions per byte): 3 for bits, 0 for uint8_t counters
bool plan_segments(uint64_t mloc, PartPlan& pl, uint32_t unit_shift = 3)
{
	// 64 KiB segments (two pass-C workgroups per CU) as long as they number at most 2^19, else 128 KiB:
	// with more than 512 x 1024 segments pass A would need 1024 level-0 bins, whose 32-entry rings make
	// a quarter of the entries take the late path (measured: pass A 63 -> 54 ms at 512 bins; the
	// read-only pass C loses 0.7 ms per launch with one workgroup per CU)
	const uint32_t small = 16 + unit_shift;
	pl.seg_shift = small;
	if (((mloc + (1ull << small) - 1) >> small) > 512ull * 1024)
		pl.seg_shift = small + 1;
	if (const char* e = getenv("BTLBF_SEG_SHIFT")) { // tuning knob: 19 or 20 (= 64 / 128 KiB segments)
		const int v = atoi(e);
		if (v == 19 || v == 20)
			pl.seg_shift = (uint32_t)v - 3 + unit_shift;
	}
	pl.n_seg = (mloc + (1ull << pl.seg_shift) - 1) >> pl.seg_shift;
	// up to 2^20 segments: pass A x one split pass; up to 2^22 (a 256 GiB bit array and beyond): two split passes
	// (entries stay 32-bit: 1024 level-0 bins of at most 2^32 positions)
	return pl.n_seg <= 4096ull * 1024;
}

// Slices of a split pass: workgroup (bin, slice) of bins_g input bins, each slice taking every `slices`-th input
// region of its bin (at most r_in).  About two workgroups per CU, but a whole number of rounds over the CUs: with
// 48 bins per group (a 3 x 2^37-bit filter) the old rule, ceil(512 / bins), gave 528 workgroups -- two full rounds of
// 256 and a third for 16 of them, and the split pass took 5.8 ms instead of 4.2.  Looked for between half of that rule's
// count and 16 (pass C walks up to 16 regions per segment one by one, kApplyFewRegions) or the rule's count if larger.
uint32_t split_slices(uint32_t bins_g, uint32_t r_in, uint32_t cus)
{
	const uint32_t want = std::max(1u, std::min(r_in, (2 * cus + bins_g - 1) / bins_g)); // the old rule
	const uint32_t hi = std::max(1u, std::min(r_in, std::max(want, 16u)));
	uint32_t best = want;
	double best_cost = 1e30;
	for (uint32_t s = std::max(1u, want / 2); s <= hi; ++s) {
		const uint64_t wg = (uint64_t)bins_g * s;
		const double rounds = (double)((wg + cus - 1) / cus);
		// time ~ rounds x work per workgroup; a slight preference for the grid the rule aimed at
		const double cost = rounds / (double)wg * (1.0 + 0.02 * std::abs((double)s - (double)want) / (double)want);
		if (cost < best_cost) {
			best_cost = cost;
			best = s;
		}
	}
	return best;
}

// append the split levels that take bins of 2^lv[0].shift positions down to segments
bool plan_splits(PartPlan& pl, uint32_t regions_in_total, uint32_t cus = 256)
{
	pl.n_levels = 1;
	uint32_t regions_in = regions_in_total;
	if (pl.lv[0].wseg) { // bins of wseg segments: one split pass, wseg ways, straight to (real) segment numbers
		PartLevel& o = pl.lv[1];
		o.P = pl.lv[0].wseg;
		o.bins = pl.lv[0].bins * o.P;
		o.shift = pl.seg_shift;
		o.regions = split_slices(pl.lv[0].bins, regions_in, cus);
		pl.n_levels = 2;
	} else {
		const uint32_t rb = pl.lv[0].shift - pl.seg_shift;
		if (rb == 0)
			return true;
		if (rb > 20)
			return false;
		const uint32_t fan[2] = {rb <= 10 ? rb : rb - rb / 2, rb <= 10 ? 0 : rb / 2};
		for (int j = 0; j < 2 && fan[j]; ++j) {
			PartLevel& in = pl.lv[pl.n_levels - 1];
			PartLevel& o = pl.lv[pl.n_levels];
			o.P = 1u << fan[j];
			o.bins = in.bins * o.P;
			o.shift = in.shift - fan[j];
			o.regions = split_slices(in.bins, regions_in, cus);
			regions_in = o.regions;
			++pl.n_levels;
		}
	}
	// the split levels and the apply pass run in groups of level-0 bins (split a group all the way
	// down, apply its segments, next group): the arrays of the split levels then hold one group
	// instead of the whole batch, so a batch can be almost twice as large for the same scratch and the
	// filter is swept fewer times
	pl.group_bins = 0;
	if (pl.n_levels >= 2 && pl.lv[0].bins >= 16) {
		pl.group_bins = (pl.lv[0].bins + 7) / 8;
		if (const char* e = getenv("BTLBF_GROUP_BINS")) { // tuning knob: level-0 bins split + applied per launch pair
			const int v = atoi(e);
			if (v >= 1 && (uint32_t)v <= pl.lv[0].bins)
				pl.group_bins = (uint32_t)v;
		}
		uint32_t bins_g = pl.group_bins, r_in = regions_in_total;
		for (int j = 1; j < pl.n_levels; ++j) {
			pl.lv[j].regions = split_slices(bins_g, r_in, cus);
			r_in = pl.lv[j].regions;
			bins_g *= pl.lv[j].P;
		}
	}
	return true;
}

// capacities + byte sizes for `entries` expected entries in the whole batch
void plan_caps(PartPlan& pl, double entries, int first_level)
{
	pl.bytes_total = 0;
	for (int j = first_level; j < pl.n_levels; ++j) {
		PartLevel& l = pl.lv[j];
		// the last level has n_seg useful bins although bins may be rounded up
		const double useful = j == pl.n_levels - 1 ? (double)std::min<uint64_t>(pl.n_seg, l.bins) : (double)l.bins;
		l.cap = chunks_for(entries / (useful * l.regions), 1);
		l.alloc_bins = l.bins;
		if (j >= 1 && pl.group_bins) {
			l.alloc_bins = pl.group_bins;
			for (int i = 1; i <= j; ++i)
				l.alloc_bins *= pl.lv[i].P;
		}
		l.cnt_bytes = ((uint64_t)l.alloc_bins * l.regions * 4 + 255) / 256 * 256;
		l.ent_bytes = (uint64_t)l.alloc_bins * l.regions * l.cap * (kChunk * 4);
		pl.bytes_total += l.cnt_bytes + l.ent_bytes;
	}
}

uint8_t* carve_levels(PartPlan& pl, uint8_t* p, int first_level)
{
	for (int j = first_level; j < pl.n_levels; ++j) {
		pl.lv[j].cnt = reinterpret_cast<uint32_t*>(p);
		p += pl.lv[j].cnt_bytes;
		pl.lv[j].ent = reinterpret_cast<uint32_t*>(p);
		p += pl.lv[j].ent_bytes;
	}
	return p;
}

int ensure_scratch(btlbf_filter* f, uint64_t bytes, bool* ok)
{
	*ok = true;
	if (bytes <= f->part_bytes)
		return BTLBF_OK;
	(void)hipFree(f->d_part);
	f->d_part = nullptr;
	f->part_bytes = 0;
	hipError_t e = hipMalloc(&f->d_part, bytes);
	if (e != hipSuccess) { // parked staging buffers of HOST-mode calls may be what is missing
		(void)hipGetLastError();
		dev_pool().drain(f->device);
		e = hipMalloc(&f->d_part, bytes);
	}
	if (e != hipSuccess) {
		(void)hipGetLastError();
		f->d_part = nullptr;
		*ok = false; // no room for scratch: the caller falls back to the direct kernels
		return BTLBF_OK;
	}
	f->part_bytes = bytes;
	return BTLBF_OK;
}

uint64_t scratch_budget(btlbf_filter* f)
{
	if (f->part_budget)
		return f->part_budget;
	size_t free_b = 0, total_b = 0;
	if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
		return 0;
	return (uint64_t)((double)(free_b + f->part_bytes) * 0.80);
}

unsigned cu_count(int device)
{
	int cus = 256;
	(void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
	return (unsigned)cus;
}

// pass A gives every workgroup (= region) ceil(tiles / regions) tiles: capacities are planned for the
// fullest region, which matters when a batch has only a few tiles per workgroup
uint64_t tiles_for_caps(uint64_t tiles, uint32_t regions)
{
	return regions ? (tiles + regions - 1) / regions * regions : tiles;
}

// expected probes of one full pass-A tile (+1 so that capacities never come out as zero)
double probes_per_tile(const btlbf_filter* f, const PartTiling& tl)
{
	return tl.windows_per_tile * f->hp.h + 1.0;
}

// how level-0 bins map to positions, for the kernels (PartSide::bin_wseg)
void side_bins(PartSide& sd, const PartPlan& pl)
{
	sd.bin_wseg = pl.lv[0].wseg;
	sd.bin_magic = pl.lv[0].wseg ? (uint32_t)(((1ull << 32) + pl.lv[0].wseg - 1) / pl.lv[0].wseg) : 0;
	sd.bin_seg_shift = pl.seg_shift;
	sd.bin_width = pl.lv[0].wseg << pl.seg_shift;
}

// run the split levels lv[1..] over the level-0 data `in0`, then the apply / test pass
// in0 holds level-0 bins [bin_offset, bin_offset + n_bins0) of the local array (bin i of in0 = absolute bin
// bin_offset + i); the whole array by default
int run_levels(btlbf_filter* f, PartPlan& pl, PartIn in0, const PartSide& sd, int query, hipStream_t s,
               uint32_t bin_offset = 0, uint32_t n_bins0 = 0)
{
	const int exact = sd.counting && !query; // counter increments: every entry exactly once
	if (f->lazy_zero && !(sd.fresh && !query)) // only a fresh insert may run on a lazily cleared array
		return fail(BTLBF_EINVAL, "internal error: partition passes on a lazily cleared array");
	const int prof_split = query ? BTLBF_PROF_QUERY_SPLIT : BTLBF_PROF_INSERT_SPLIT;
	const int prof_apply = query ? BTLBF_PROF_QUERY_TEST : BTLBF_PROF_INSERT_APPLY;
	if (n_bins0 == 0)
		n_bins0 = pl.lv[0].bins - bin_offset;
	// group by group: split the group's level-0 bins all the way down, then apply its segments
	// (plans without groups: one group of everything)
	const uint32_t group = pl.n_levels >= 2 && pl.group_bins ? pl.group_bins : n_bins0;
	for (uint32_t b0 = 0; b0 < n_bins0; b0 += group) {
		PartIn in = in0;
		uint32_t first_in = b0, abs_first = bin_offset + b0, n_in = std::min(group, n_bins0 - b0);
		uint32_t in_shift = pl.lv[0].shift;
		for (int j = 1; j < pl.n_levels; ++j) {
			ProfSpan ps(f, prof_split, s);
			HIP_TRY(launch_part_split(f->d_data, in, first_in, abs_first, n_in, pl.lv[j].out(), pl.lv[j].shift,
			                          in_shift, sd, query, exact, s));
			in = pl.lv[j].in();
			first_in = 0;
			abs_first *= pl.lv[j].P;
			n_in *= pl.lv[j].P;
			in_shift = pl.lv[j].shift;
		}
		const uint64_t seg_first = abs_first;
		if (seg_first >= pl.n_seg)
			break;
		const uint64_t n_seg = std::min<uint64_t>(n_in, pl.n_seg - seg_first);
		ProfSpan ps(f, prof_apply, s);
		HIP_TRY(launch_part_apply(f->d_data, f->local_bytes, pl.seg_shift, seg_first, n_seg, in, sd, query, s));
	}
	return BTLBF_OK;
}

// AUTO's break-even between the direct kernels and a sweep of the array, as probes per byte of the local array.
// Measured on MI355X at 2^39 bits (tools/auto_probe.py): the direct insert costs 24.4 ms per 10^6 reads of 150 bp (21 G
// atomics/s), the partitioned one 26.5 ms + 1.9 ms per 10^6 reads -- equal at 0.82 %; the direct query 9.3 ms per 10^6
// reads (all hits: four gathers per k-mer), the partitioned one 17.3 ms + 1.6 per 10^6 -- equal at 1.57 %.  (Round 2's
// rule was 2 % for both: a batch of 2x10^6 reads was inserted in 48.7 ms instead of 30.4.)
constexpr double kAutoInsertRatio = 0.0095, kAutoQueryRatio = 0.0165;
// plan_level0: calls of this many probes or more (4x10^9 k-mers at h = 4) take 256 level-0 bins where 512 are the rule
constexpr double kWideSplitProbes = 1.6e10;

// decide between the direct (atomicOr per probe) and the partitioned insert
// bit filters: insert; counting filters: incrementAll only (the conservative update of `insert` needs
// the minimum over a k-mer's h counters, which live in different segments)
bool want_partitioned(const btlbf_filter* f, uint64_t len, int counting_op = -1)
{
	if (f->insert_mode == BTLBF_INSERT_DIRECT)
		return false;
	if (f->kind == BTLBF_COUNTING8 ? counting_op != BTLBF_INCREMENT_ALL : f->kind != BTLBF_BLOOM)
		return false;
	if (!part_supported(f->hp) || len == 0)
		return false;
	if (f->insert_mode == BTLBF_INSERT_PARTITIONED)
		return true;
	// auto: one sweep of the local array (read + write) must be cheaper than the random atomics it
	// replaces: ~ 2*bytes/5.8e12 s against probes/21e9 s (kAutoInsertRatio); and the batch must be big
	// enough to be worth five launches
	const double probes = (double)len * f->hp.h * ((double)f->mod.shard_len / (double)f->mod.size);
	return probes >= kAutoInsertRatio * (double)f->local_bytes && probes >= 4.0e6;
}

// the segment size and the level-0 bins (pass A's output) of this filter's local array; false = no partitioned path
// probes a call over `len` bases sends to this filter's local array (a shard keeps its window's share) -- or, with a
// scratch budget imposed by the caller, what one batch of that budget holds (about 5.5 bytes of scratch per probe): the
// figure plan_level0's batch-size rule goes by.  (Deterministic on purpose: the split query plans twice and both plans
// must agree; the free-memory budget would not be the same figure twice.)
double call_probes(const btlbf_filter* f, uint64_t len)
{
	const double all = (double)len * f->hp.h * ((double)f->mod.shard_len / (double)f->mod.size);
	return f->part_budget ? std::min(all, (double)f->part_budget / 5.5) : all;
}

// `call_probes`: probes of the whole call (0 = unknown), for the one choice that depends on the batch size
bool plan_level0(const btlbf_filter* f, PartPlan& pl, double call_probes = 0)
{
	if (!plan_segments(f->mod.shard_len, pl, f->kind == BTLBF_COUNTING8 ? 0 : 3))
		return false;
	PartLevel& l0 = pl.lv[0];
	if (pl.n_seg <= 1024) {
		l0.bins = (uint32_t)pl.n_seg;
		l0.shift = pl.seg_shift;
	} else { // split the segment index bits evenly between pass A and pass B
		unsigned b1 = (ceil_log2(pl.n_seg) + 1) / 2; // pass B takes the larger half: pass A gains more from big rings
		// 2^17 < segments <= 2^18 (bit filters of 8 .. 16 GiB): 512 bins and a 512-way split by that rule, or 256 bins
		// and a 1024-way split.  Pass A is 6 % (plain ntHash: 40.2 -> 37.9 ms per 6x10^9 k-mers) to 8.5 % (four spaced
		// seeds: 89.7 -> 82.1) faster on 128-entry rings; the 1024-way split pass costs the same per launch in batches
		// of 6x10^9 k-mers and 0.4 ms more (of 1.8) in batches of 2.4x10^9 (tools/quick_bench.py with BTLBF_SPLIT_BITS,
		// DESIGN.md B.2) -- so for large calls only.
		if (ceil_log2(pl.n_seg) == 18 && call_probes >= kWideSplitProbes)
			b1 = 10;
		if (const char* e = getenv("BTLBF_SPLIT_BITS")) { // tuning knob: segment-index bits left to pass B
			const int v = atoi(e);
			if (v >= 1 && v <= 10 && ceil_log2(pl.n_seg) - v <= 10)
				b1 = (unsigned)v;
		}
		if (ceil_log2(pl.n_seg) > b1 + 10)
			b1 = ceil_log2(pl.n_seg) - 10; // pass A writes at most 1024 bins
		l0.shift = pl.seg_shift + b1;
		l0.bins = (uint32_t)((pl.n_seg + (1ull << b1) - 1) >> b1);
		if (l0.shift > 32)
			return false; // (cannot happen below 2^22 segments)
		// A bin count that is no power of two leaves staging rings of pass A unused while the others take more
		// entries per round than they are sized for: 3 x 2^37 bits gave 384 bins on the 512-ring geometry, a third
		// more entries per ring and round, and pass A took 22.2 ms per 2.4x10^9 k-mers where a filter of 512 bins and the
		// same reduction takes 18.6.  So the bins are made of a whole number of SEGMENTS instead, as many as fill the
		// geometry's rings (768 segments per bin there, 512 bins); pass B then splits wseg ways.  One split level only.
		const uint32_t rings = 1u << ceil_log2(l0.bins);
		static const bool pow2_bins = getenv("BTLBF_POW2_BINS") != nullptr; // diagnostic: the old rule
		if (l0.bins < rings && b1 <= 10 && !pow2_bins) {
			l0.wseg = (uint32_t)((pl.n_seg + rings - 1) / rings);
			l0.bins = (uint32_t)((pl.n_seg + l0.wseg - 1) / l0.wseg);
		}
	}
	l0.P = l0.bins;
	l0.alloc_bins = l0.bins;
	return true;
}

// plan the single-GPU pipeline for a buffer and (re)allocate the scratch;
// *ok = false means "not applicable, use the direct kernel"
int part_prepare(btlbf_filter* f, const SeqArgs& base, PartTail* tail, PartPlan& pl, PartTiling* tiling,
                 uint8_t** extra, bool* ok, int mode, double auto_ratio)
{
	*ok = false;
	if (!plan_level0(f, pl, call_probes(f, base.len)))
		return BTLBF_OK;
	PartLevel& l0 = pl.lv[0];
	l0.regions = part_hash_regions(f->hp, l0.P, cu_count(f->device)); // pass-A workgroups: one or two per CU
	if (!plan_splits(pl, l0.regions, cu_count(f->device)) || !part_hash_fits(f->hp, l0.P))
		return BTLBF_OK;
	*tiling = part_tiling(f->hp, l0.P, base.layout, base.len);
	const uint64_t budget = scratch_budget(f);
	*tail = part_tail(budget);
	// (a caller-imposed budget below 2 GiB keeps its scratch for the entries: pass A then runs its plain schedule)
	pl.late_cap = budget >= (2ull << 30) ? part_late_cap() : 0;
	const uint64_t late_bytes = (uint64_t)l0.regions * 2 * pl.late_cap * sizeof(uint32_t);
	const uint64_t extra_bytes = ((tail->bytes + 255) / 256) * 256 + late_bytes;
	// a shard fed every rank's reads (ShardedBloomFilter's gather mode) keeps only its window's share
	const double ppt = probes_per_tile(f, *tiling) * ((double)f->mod.shard_len / (double)f->mod.size);
	uint64_t tiles = tiling->n_tiles;
	for (int iter = 0; iter < 64; ++iter) {
		plan_caps(pl, (double)tiles_for_caps(tiles, l0.regions) * ppt, 0);
		pl.bytes_total += extra_bytes;
		if (pl.bytes_total <= budget || tiles <= 1)
			break;
		const double ratio = (double)budget / (double)pl.bytes_total;
		const uint64_t nt = (uint64_t)((double)tiles * ratio * 0.95);
		tiles = nt >= tiles ? tiles - 1 : (nt ? nt : 1);
	}
	if (pl.bytes_total > budget)
		return BTLBF_OK;
	// AUTO: a batch that the scratch budget has cut small is not worth a sweep of the array either (the rule
	// want_partitioned applies to the whole call, applied to one batch): a filter that nearly fills the HBM
	// leaves a few GB for scratch, and the direct kernels are then the faster path
	if (mode == BTLBF_INSERT_AUTO && tiles < tiling->n_tiles && (double)tiles * ppt < auto_ratio * (double)f->local_bytes)
		return BTLBF_OK;
	pl.tiles_per_batch = tiles;
	int rc = ensure_scratch(f, pl.bytes_total, ok);
	if (rc || !*ok)
		return rc;
	*extra = carve_levels(pl, static_cast<uint8_t*>(f->d_part), 0);
	pl.late_buf = pl.late_cap ? reinterpret_cast<uint32_t*>(*extra + ((tail->bytes + 255) / 256) * 256) : nullptr;
	return BTLBF_OK;
}


int partitioned_insert(btlbf_filter* f, const SeqArgs& base, hipStream_t s, bool* done)
{
	*done = false;
	// A pending btlbf_clear is carried out by the first batch itself: pass C builds every segment from zero in
	// LDS and writes it -- no memset of the array and no read sweep for that batch.  Pass C would also wipe what
	// the overflow paths of passes A and B write straight into the array, so a fresh batch reports those entries
	// as explicit positions instead (as the multi-GPU routing does) and they are applied after its last pass C;
	// more of them than the list holds (heavily skewed input) and the batch is redone the ordinary way.
	PartPlan pl;
	PartTiling tiling;
	uint8_t* extra = nullptr;
	bool ok = false;
	PartTail tail;
	int rc = part_prepare(f, base, &tail, pl, &tiling, &extra, &ok, f->insert_mode, kAutoInsertRatio);
	if (rc || !ok)
		return rc;
	const uint64_t total_tiles = tiling.n_tiles;
	for (uint64_t t0 = 0; t0 < total_tiles; t0 += pl.tiles_per_batch) {
		SeqArgs a = base;
		a.first_tile = t0;
		a.n_tiles = std::min<uint64_t>(pl.tiles_per_batch, total_tiles - t0);
		for (int attempt = 0; attempt < 2; ++attempt) {
			const bool fresh = f->lazy_zero;
			PartSide sd;
			memset(&sd, 0, sizeof sd);
			sd.counting = f->kind == BTLBF_COUNTING8;
			sd.late_buf = pl.late_buf;
			sd.late_cap = pl.late_cap;
			side_bins(sd, pl);
			if (fresh) {
				HIP_TRY(order_after_clear(f, s)); // this batch IS the clear: after the point it was asked for
				sd.fresh = 1;
				sd.pos_base = f->mod.shard_lo;
				sd.spill_count = reinterpret_cast<unsigned long long*>(extra);
				sd.spill_list = reinterpret_cast<uint64_t*>(extra + 256);
				sd.spill_cap = tail.spill_cap;
				HIP_TRY(hipMemsetAsync(sd.spill_count, 0, 8, s));
			}
			{
				ProfSpan ps(f, BTLBF_PROF_INSERT_HASH, s);
				HIP_TRY(launch_part_hash(a, pl.lv[0].out(), pl.lv[0].shift, sd, 0, s));
			}
			if ((rc = run_levels(f, pl, pl.lv[0].in(), sd, 0, s)))
				return rc;
			if (!fresh)
				break;
			f->lazy_zero = false; // every segment has been written
			unsigned long long n_spill = 0;
			hipError_t e = hipMemcpyAsync(&n_spill, sd.spill_count, 8, hipMemcpyDeviceToHost, s);
			if (e == hipSuccess)
				e = hipStreamSynchronize(s);
			if (e == hipSuccess && n_spill <= tail.spill_cap) {
				PartSide plain;
				memset(&plain, 0, sizeof plain);
				plain.counting = sd.counting;
				e = launch_spill(f->d_data, sd.spill_list, n_spill, f->mod.shard_lo, f->mod.shard_len, 0, plain, s);
				if (e == hipSuccess)
					break;
			}
			if (e != hipSuccess) { // the batch is half applied: back to a defined (empty) state
				f->lazy_zero = true;
				return fail(BTLBF_EHIP, "fresh partitioned insert: %s", hipGetErrorString(e));
			}
			HIP_TRY(hipMemsetAsync(f->d_data, 0, f->alloc_bytes, s)); // start over, the ordinary way
		}
	}
	*done = true;
	return BTLBF_OK;
}

// hit_bits := hit_bits with the windows owning a failed position cleared, for seq tiles
// [first, first+n) of the direct kernels' tiling
int resolve_range(btlbf_filter* f, const SeqArgs& base, uint8_t* hit_bits, const uint64_t* fail_list, uint64_t n_fail,
                  uint64_t* table, uint64_t max_slots, uint64_t first, uint64_t n, hipStream_t s)
{
	SeqArgs d = base;
	d.first_tile = first;
	d.n_tiles = n;
	d.hit_bits = hit_bits;
	d.valid_bits = nullptr;
	d.counts = nullptr;
	// the table is sized to the set (load <= 1/4): a few thousand failed positions make a table that stays in
	// L2, and every probe of every window of the range is looked up in it
	uint64_t slots = 1024;
	while (slots < 4 * n_fail && slots < max_slots)
		slots <<= 1;
	HIP_TRY(hipMemsetAsync(table, 0, slots * 8, s));
	HIP_TRY(launch_failset_build(fail_list, n_fail, table, slots - 1, s));
	d.buckets = table;
	d.bucket_cap = slots - 1;
	HIP_TRY(launch_seq_op(OP_BF_RESOLVE, d, s));
	return BTLBF_OK;
}

// Partitioned contains(): positions are partitioned exactly as for insert and
// TESTED against each segment in LDS; positions found clear go to a (small) fail list.  A batch
// without failures is finished: every clean window hits.  Otherwise the failed positions become a
// cache-resident hash set and one more hashing pass clears the windows that own one of them.  Too
// many failures (a miss-heavy batch) and the batch is redone by the direct gather kernel.
// hit_bits (device) is required; valid_bits and counts are optional.
// base.read_mask (the split query): those reads are left out -- no bits, no counts; defer_hit_count: counts[1] is left
// for the caller, who adds the left-out reads' answers to the bitmap first.
int partitioned_contains(btlbf_filter* f, const SeqArgs& base, uint8_t* hit_bits, uint8_t* valid_bits,
                         uint64_t* counts, hipStream_t s, bool* done, bool defer_hit_count)
{
	*done = false;
	PartPlan pl;
	PartTiling tiling;
	uint8_t* extra = nullptr;
	bool ok = false;
	PartTail tail;
	int rc = part_prepare(f, base, &tail, pl, &tiling, &extra, &ok, f->query_mode, kAutoQueryRatio);
	if (rc || !ok)
		return rc;
	const uint64_t total_tiles = tiling.n_tiles;
	PartSide sd;
	memset(&sd, 0, sizeof sd);
	sd.fail_count = reinterpret_cast<unsigned long long*>(extra);
	sd.fail_list = reinterpret_cast<uint64_t*>(extra + 256);
	sd.fail_cap = tail.fail_cap;
	sd.counting = f->kind == BTLBF_COUNTING8;
	sd.threshold = f->thr;
	sd.late_buf = pl.late_buf;
	sd.late_cap = pl.late_cap;
	side_bins(sd, pl);
	sd.pos_base = f->mod.shard_lo; // the fail set is keyed by global position
	const int direct_op = sd.counting ? OP_CBF_QUERY : f->shard_count != 1 ? OP_BF_CONTAINS_WIN : OP_BF_CONTAINS;
	uint64_t* table = sd.fail_list + tail.fail_cap;
	uint64_t* ctl = reinterpret_cast<uint64_t*>(extra + 64); // two words of stream-side control next to the fail count
	if (counts)
		HIP_TRY(hipMemsetAsync(counts, 0, 16, s));
	const uint64_t seq_tw = (uint64_t)seq_tile_windows();
	const uint64_t seq_tiles_all = (base.len + seq_tw - 1) / seq_tw;
	for (uint64_t t0 = 0; t0 < total_tiles; t0 += pl.tiles_per_batch) {
		SeqArgs a = base;
		a.first_tile = t0;
		a.n_tiles = std::min<uint64_t>(pl.tiles_per_batch, total_tiles - t0);
		a.hit_bits = hit_bits;
		a.valid_bits = valid_bits;
		a.counts = counts; // pass A adds the clean-window count to counts[0]
		HIP_TRY(hipMemsetAsync(sd.fail_count, 0, 8, s));
		{
			ProfSpan ps(f, BTLBF_PROF_QUERY_HASH, s);
			HIP_TRY(launch_part_hash(a, pl.lv[0].out(), pl.lv[0].shift, sd, 1, s));
		}
		if ((rc = run_levels(f, pl, pl.lv[0].in(), sd, 1, s)))
			return rc;
		// redo / refine this batch's window range with the direct kernels: their tiles that overlap the
		// batch's bytes.  A tile more at either end is harmless: a failed position is a bit that IS clear,
		// so clearing any window that owns it is right, and a direct redo computes the true answer.
		// What happens is decided on the device (GATE_*): no failed position -> nothing; up to fail_cap -> they become
		// a hash set and one hashing pass clears the windows that own one; more -> the range is redone by the direct
		// kernel.  All launches are issued, the ones decided against return at once: no host round trip per batch.
		const uint64_t first = t0 * tiling.tile_bytes / seq_tw;
		const uint64_t end_b = std::min<uint64_t>(base.len, (t0 + a.n_tiles) * (uint64_t)tiling.tile_bytes);
		const uint64_t n = std::min<uint64_t>((end_b + seq_tw - 1) / seq_tw, seq_tiles_all) - first;
		ProfSpan ps(f, BTLBF_PROF_QUERY_RESOLVE, s);
		HIP_TRY(launch_failset_auto(sd.fail_list, sd.fail_count, tail.fail_cap, table, tail.table_slots, ctl, s));
		SeqArgs d = base;
		d.first_tile = first;
		d.n_tiles = n;
		d.hit_bits = hit_bits;
		d.valid_bits = nullptr;
		d.counts = nullptr;
		d.gate = ctl;
		d.gate_mode = GATE_REDO;
		REQUIRE_MATERIALIZED(f);
		HIP_TRY(launch_seq_op(direct_op, d, s));
		d.buckets = table;
		d.gate_mode = GATE_RESOLVE;
		HIP_TRY(launch_seq_op(OP_BF_RESOLVE, d, s));
	}
	if (counts && !defer_hit_count) // hits = set bits of the final bitmap
		HIP_TRY(launch_popcount(hit_bits, ((base.len + 63) / 64) * 8, 0, 0,
		                        reinterpret_cast<unsigned long long*>(counts) + 1, s));
	*done = true;
	return BTLBF_OK;
}

// AUTO decision for contains(): large batch, and a sample of tiles says nearly every k-mer hits
int want_partitioned_query(btlbf_filter* f, const SeqArgs& base, hipStream_t s, bool* yes)
{
	*yes = false;
	if (f->query_mode == BTLBF_INSERT_DIRECT)
		return BTLBF_OK;
	if (!part_supported(f->hp) || base.len == 0)
		return BTLBF_OK;
	if (f->query_mode == BTLBF_INSERT_PARTITIONED) {
		*yes = true;
		return BTLBF_OK;
	}
	const double live = (double)base.len * f->hp.h * ((double)f->mod.shard_len / (double)f->mod.size);
	if (live < kAutoQueryRatio * (double)f->local_bytes || live < 4.0e6)
		return BTLBF_OK;
	// sample 64 tiles spread over the buffer with the direct kernel
	const uint64_t tiles = (base.len + seq_tile_windows() - 1) / seq_tile_windows();
	const unsigned n_s = (unsigned)std::min<uint64_t>(64, tiles);
	HIP_TRY(hipMemsetAsync(f->d_scalar, 0, 16, s));
	for (unsigned i = 0; i < n_s; ++i) {
		SeqArgs a = base;
		a.first_tile = (tiles / n_s) * i;
		a.n_tiles = 1;
		a.hit_bits = nullptr;
		a.valid_bits = nullptr;
		a.counts = reinterpret_cast<uint64_t*>(f->d_scalar);
		a.min_out = nullptr;
		HIP_TRY(launch_seq_op(f->kind == BTLBF_COUNTING8 ? OP_CBF_QUERY
		                      : f->shard_count != 1      ? OP_BF_CONTAINS_WIN
		                                                 : OP_BF_CONTAINS,
		                      a, s));
	}
	unsigned long long c[2] = {0, 0};
	HIP_TRY(hipMemcpyAsync(c, f->d_scalar, 16, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	if (c[0] == 0)
		return BTLBF_OK;
	// expected failed probes in the whole call (at most h per missing k-mer) must stay well below
	// what the fail list holds per batch
	const double miss = (double)(c[0] - c[1]) / (double)c[0];
	*yes = miss * live < 0.25 * (double)part_tail(scratch_budget(f)).fail_cap;
	return BTLBF_OK;
}

// contains() over fixed-length reads in AUTO mode (aux_kernels.hip, "split query"): sample every read; if the
// misses are few enough for the fail list the whole buffer goes partitioned (*decided = 2), if hardly anything
// hits it goes to the gather kernel (1); otherwise the reads are compacted into a warm and a cold buffer, the
// warm one takes the partitioned path, the cold one the early-exit gather kernel, and the two bitmaps are
// merged back into the caller's layout (3: a.hit_bits / a.valid_bits / a.counts are complete).
int split_contains(btlbf_filter* f, const SeqArgs& a, int direct_op, hipStream_t s, int* decided)
{
	*decided = 0;
	const uint32_t L = a.layout.starts ? 0 : a.layout.read_len, k = f->hp.k;
	// whole filters only: the sampler probes f->d_data with positions of the whole array (a shard answers for its
	// window through the WINDOW kernels; want_partitioned_query decides for it)
	if (f->shard_count != 1 || f->mod.shard_lo != 0 || f->mod.shard_len != f->mod.size)
		return BTLBF_OK;
	if (f->query_mode != BTLBF_INSERT_AUTO || !L || L < k || L < 8 || f->hp.n_seeds || !part_supported(f->hp))
		return BTLBF_OK;
	const uint64_t n_reads = a.len / L;
	const uint32_t W = L - k + 1;
	const double live = (double)n_reads * W * f->hp.h;
	if (live < kAutoQueryRatio * (double)f->local_bytes || live < 4.0e6 || n_reads >= (1ull << 32))
		return BTLBF_OK; // small batches: the direct kernel (want_partitioned_query agrees)
	const uint64_t n_fw = (n_reads + 63) / 64;
	// temporaries are cached in the filter (grow-only, btlbf_release_scratch returns them): hipMalloc / hipFree
	// of tens of GB cost more than the kernels.  The small one (flags, prefix sums) is needed by every call;
	// the large one (compacted reads, their bitmaps) only once the split path is taken
	auto up = [](uint64_t x) { return (x + 255) / 256 * 256; };
	auto grow = [](void** p, uint64_t* have, uint64_t bytes) -> bool {
		if (bytes <= *have)
			return true;
		(void)hipFree(*p);
		*p = nullptr;
		*have = 0;
		if (hipMalloc(p, bytes) != hipSuccess) {
			(void)hipGetLastError();
			return false;
		}
		*have = bytes;
		return true;
	};
	// (the flags are readable for 256 bytes behind their last word: pass A reads up to 34 words from a tile's first one on)
	const uint64_t sz_flags = up(n_fw * 8 + 256), sz_prefix = up((n_fw + (n_fw + 1023) / 1024 + 1) * 4);
	if (!grow(&f->d_flags, &f->flags_bytes, 256 + sz_flags + sz_prefix))
		return BTLBF_OK; // no room: the plain paths decide (want_partitioned_query)
	unsigned long long* d_ncold = static_cast<unsigned long long*>(f->d_flags);
	uint64_t* d_flags = reinterpret_cast<uint64_t*>(static_cast<uint8_t*>(f->d_flags) + 256);
	uint32_t* d_prefix = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(f->d_flags) + 256 + sz_flags);
	unsigned long long n_cold = 0;
	auto sample = [&](uint32_t stride, uint32_t probes2) -> int {
		HIP_TRY(hipMemsetAsync(d_ncold, 0, 8, s));
		{
			ProfSpan ps(f, BTLBF_PROF_QUERY_RESOLVE, s);
			HIP_TRY(launch_read_sample(a.seq, n_reads, L, stride, f->hp, f->mod, f->d_data, f->kind == BTLBF_COUNTING8,
			                           f->thr, d_flags, reinterpret_cast<uint64_t*>(d_ncold), s, probes2));
		}
		HIP_TRY(hipMemcpyAsync(&n_cold, d_ncold, 8, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		return BTLBF_OK;
	};
	// what the fail list copes with / what is worth a sweep of the array, in reads
	// (the list the partitioned path will really have: a small scratch budget gets a short one, part_tail)
	const double few_cold = 0.25 * (double)part_tail(scratch_budget(f)).fail_cap / ((double)W * f->hp.h);
	auto warm_too_few = [&](double n_warm_reads) {
		const double wl = n_warm_reads * W * f->hp.h;
		return wl < kAutoQueryRatio * (double)f->local_bytes || wl < 4.0e6;
	};
	// 1. an estimate from one read in 64: all-hit and all-miss buffers -- the common cases -- are recognised at
	//    1/64 of the cost of looking at every read
	int rc;
	const uint32_t stride = n_reads >= (1u << 20) ? 64 : 1;
	double cold_frac = 1.0; // estimate from the first look (unknown: assume many)
	if (stride > 1) {
		if ((rc = sample(stride, 0)))
			return rc;
		cold_frac = (double)n_cold / (double)((n_reads + stride - 1) / stride);
		const uint64_t n_s = (n_reads + stride - 1) / stride;
		if (n_cold == 0 && (double)n_reads * 8.0 / (double)n_s < few_cold) { // none in the sample: few overall
			*decided = 2;
			return BTLBF_OK;
		}
		if (warm_too_few((double)(n_s - n_cold) * stride * 1.5)) {
			*decided = 1;
			return BTLBF_OK;
		}
	}
	// 2. every read.  A present read costs the sampler its probes (they all hit, so they are all loaded), and the second
	//    sample only has to keep a foreign read from passing on ONE false-positive window: with few foreign reads, fewer
	//    of its probes do (a read that passes all the same costs a resolve pass, never a wrong answer)
	const uint32_t h = f->hp.h;
	const uint32_t probes2 = cold_frac <= 0.0025 ? (h + 1) / 2 : cold_frac <= 0.025 ? std::max((h + 1) / 2, h - 1) : h;
	if ((rc = sample(1, probes2)))
		return rc;
	const uint64_t n_warm = n_reads - n_cold;
	if ((double)n_cold < few_cold) { // the fail list copes with that many misses
		*decided = 2;
		return BTLBF_OK;
	}
	if (warm_too_few((double)n_warm)) { // not worth a sweep of the array
		*decided = 1;
		return BTLBF_OK;
	}
	// ---- split ----
	const uint64_t warm_len = n_warm * L, cold_len = n_cold * L;
	const bool wv = a.valid_bits != nullptr;
	// Uniform reads that pass A takes through its read grid: the warm reads stay where they are -- pass A leaves the
	// cold ones out by their flags (zero-staged: no entries, no bits, no counts) --, only the COLD reads are gathered for
	// the direct kernel, and their answers are ORed back into the caller's bitmaps.  The first version gathered the warm
	// reads as well (15 GB copied and 15 GB of HBM that the partition scratch then lacked: a third batch) and merged
	// every word of the bitmaps from two sources.
	{
		PartPlan pl0;
		PartGrid g;
		// (up to a quarter of the reads cold: beyond that the lanes pass A spends on zero-staged reads cost more than
		// gathering the warm reads costs -- at one read in two 76 instead of 43 ms of pass A per 10^8 reads)
		if (4 * n_cold <= n_reads && plan_level0(f, pl0, call_probes(f, a.len)) && part_read_grid(f->hp, pl0.lv[0].P, a.layout, &g)) {
			const uint64_t szm[5] = {up(cold_len + 16), up(bitmap_bytes(cold_len) + 16), wv ? up(bitmap_bytes(cold_len) + 16) : 0,
			                         up(n_cold * 4 + 16), a.hit_bits ? 0 : up(bitmap_bytes(a.len) + 16)};
			uint64_t need = 0;
			for (uint64_t v : szm)
				need += v;
			// (a buffer left behind by a call that gathered the warm reads as well -- 19 GB for 10^8 reads -- is given back
			// first: the partition scratch is planned from the free HBM, and with that much less of it the pass would need
			// a third batch, i.e. a third sweep of the array)
			if (f->split_bytes > 4 * need + (1ull << 30)) {
				(void)hipFree(f->d_split);
				f->d_split = nullptr;
				f->split_bytes = 0;
			}
			if (!grow(&f->d_split, &f->split_bytes, need)) {
				*decided = 1;
				return BTLBF_OK;
			}
			uint8_t* q[5];
			{
				uint64_t off = 0;
				for (int i = 0; i < 5; ++i) {
					q[i] = szm[i] ? static_cast<uint8_t*>(f->d_split) + off : nullptr;
					off += szm[i];
				}
			}
			uint8_t *cold_p = q[0], *cold_hit_p = q[1], *cold_valid_p = q[2];
			uint32_t* cold_index = reinterpret_cast<uint32_t*>(q[3]);
			uint8_t* hb = a.hit_bits ? a.hit_bits : q[4];
			{
				ProfSpan ps(f, BTLBF_PROF_QUERY_RESOLVE, s);
				HIP_TRY(launch_flag_prefix(d_flags, n_reads, d_prefix, s));
				HIP_TRY(launch_gather_cold_reads(a.seq, n_reads, L, d_flags, d_prefix, cold_p, cold_index, s));
				HIP_TRY(hipMemsetAsync(cold_hit_p + bitmap_bytes(cold_len), 0, 16, s)); // (the merge reads a word further)
				if (wv)
					HIP_TRY(hipMemsetAsync(cold_valid_p + bitmap_bytes(cold_len), 0, 16, s));
			}
			SeqArgs b = a;
			b.read_mask = reinterpret_cast<const uint32_t*>(d_flags);
			b.hit_bits = b.valid_bits = nullptr;
			b.counts = nullptr;
			bool done_w = false;
			if ((rc = partitioned_contains(f, b, hb, a.valid_bits, a.counts, s, &done_w, true)))
				return rc;
			if (!done_w) { // no room for the partition scratch: the gather kernel answers the whole buffer
				*decided = 1;
				return BTLBF_OK;
			}
			SeqArgs d = a;
			d.seq = cold_p;
			d.len = cold_len;
			d.hit_bits = cold_hit_p;
			d.valid_bits = cold_valid_p;
			d.counts = a.counts; // the direct kernel ADDS its clean windows (and its hits: recounted below)
			{
				ProfSpan ps(f, BTLBF_PROF_QUERY_DIRECT, s);
				REQUIRE_MATERIALIZED(f);
				HIP_TRY(launch_seq_op(direct_op, d, s));
			}
			{
				ProfSpan ps(f, BTLBF_PROF_QUERY_RESOLVE, s);
				HIP_TRY(launch_merge_cold_bitmaps(n_cold, L, cold_index, reinterpret_cast<const uint64_t*>(cold_hit_p),
				                                  reinterpret_cast<const uint64_t*>(cold_valid_p), reinterpret_cast<uint64_t*>(hb),
				                                  reinterpret_cast<uint64_t*>(a.valid_bits), s));
				if (a.counts) { // hits = set bits of the finished bitmap
					HIP_TRY(hipMemsetAsync(a.counts + 1, 0, 8, s));
					HIP_TRY(launch_popcount(hb, bitmap_bytes(a.len), 0, 0, reinterpret_cast<unsigned long long*>(a.counts) + 1, s));
				}
			}
			*decided = 3;
			return BTLBF_OK;
		}
	}
	const uint64_t sz[6] = {up(warm_len + 16), up(cold_len + 16), up(bitmap_bytes(warm_len) + 16),
	                        up(bitmap_bytes(cold_len) + 16), wv ? up(bitmap_bytes(warm_len) + 16) : 0,
	                        wv ? up(bitmap_bytes(cold_len) + 16) : 0};
	// sized for any split of a buffer this long, so that the next call's ratio does not move memory
	const uint64_t worst = up(a.len + 32) + 512 + (wv ? 2 : 1) * (up(bitmap_bytes(a.len) + 32) + 512);
	if (!grow(&f->d_split, &f->split_bytes, worst)) {
		*decided = 1; // no room for the compacted copies: the gather kernel answers any mix
		return BTLBF_OK;
	}
	uint8_t* bufs[6];
	{
		uint64_t off = 0;
		for (int i = 0; i < 6; ++i) {
			bufs[i] = sz[i] ? static_cast<uint8_t*>(f->d_split) + off : nullptr;
			off += sz[i];
		}
		if (off > f->split_bytes)
			return fail(BTLBF_EINVAL, "split query: buffer arithmetic");
	}
	uint8_t *warm_p = bufs[0], *cold_p = bufs[1], *warm_hit_p = bufs[2], *cold_hit_p = bufs[3], *warm_valid_p = bufs[4],
	        *cold_valid_p = bufs[5];
	{
		ProfSpan ps(f, BTLBF_PROF_QUERY_RESOLVE, s);
		HIP_TRY(launch_flag_prefix(d_flags, n_reads, d_prefix, s));
		HIP_TRY(launch_compact_reads(a.seq, n_reads, L, d_flags, d_prefix, warm_p, cold_p, s));
		// the merge reads one word past the last bit of a compacted bitmap
		HIP_TRY(hipMemsetAsync(warm_hit_p + bitmap_bytes(warm_len), 0, 16, s));
		HIP_TRY(hipMemsetAsync(cold_hit_p + bitmap_bytes(cold_len), 0, 16, s));
		if (a.valid_bits) {
			HIP_TRY(hipMemsetAsync(warm_valid_p + bitmap_bytes(warm_len), 0, 16, s));
			HIP_TRY(hipMemsetAsync(cold_valid_p + bitmap_bytes(cold_len), 0, 16, s));
		}
	}
	if (a.counts)
		HIP_TRY(hipMemsetAsync(a.counts, 0, 16, s));
	SeqArgs b = a;
	b.seq = warm_p;
	b.len = warm_len;
	b.hit_bits = b.valid_bits = nullptr;
	b.counts = nullptr;
	bool done_w = false;
	rc = partitioned_contains(f, b, warm_hit_p, warm_valid_p, a.counts, s, &done_w);
	if (rc)
		return rc;
	if (!done_w) { // no room for the partition scratch: the gather kernel does the warm reads too
		b.hit_bits = warm_hit_p;
		b.valid_bits = warm_valid_p;
		b.counts = a.counts;
		ProfSpan ps(f, BTLBF_PROF_QUERY_DIRECT, s);
		HIP_TRY(launch_seq_op(direct_op, b, s));
	}
	SeqArgs d = a;
	d.seq = cold_p;
	d.len = cold_len;
	d.hit_bits = cold_hit_p;
	d.valid_bits = cold_valid_p;
	d.counts = a.counts; // the direct kernel ADDS its clean windows and hits
	{
		ProfSpan ps(f, BTLBF_PROF_QUERY_DIRECT, s);
		HIP_TRY(launch_seq_op(direct_op, d, s));
	}
	if (a.hit_bits || a.valid_bits) {
		ProfSpan ps(f, BTLBF_PROF_QUERY_RESOLVE, s);
		HIP_TRY(launch_merge_split_bitmaps(a.len, L, d_flags, d_prefix, reinterpret_cast<uint64_t*>(warm_hit_p),
		                                   reinterpret_cast<uint64_t*>(cold_hit_p), reinterpret_cast<uint64_t*>(warm_valid_p),
		                                   reinterpret_cast<uint64_t*>(cold_valid_p), reinterpret_cast<uint64_t*>(a.hit_bits),
		                                   reinterpret_cast<uint64_t*>(a.valid_bits), s));
	}
	*decided = 3;
	return BTLBF_OK;
}

// ---- multi-GPU routing (SURVEY.md 8e on the partitioned pipeline) -------------------------------------
// The GLOBAL filter (size = f->mod.size, a power of two) is cut into B = 512 (or 1024) level-0 bins; with W shards
// owner g holds bins [g*1024/W, (g+1)*1024/W).  An origin partitions its probes into those bins (pass
// A, regions = its CU count); the block of one owner is contiguous, so the exchange is a fixed-size
// all-to-all of [B/W bins][regions][cap][kChunk] uint32 plus the entry counts.
struct RoutePlan {
	uint32_t n_windows = 1;        // position windows routed one after the other (see route_plan)
	uint32_t shards_per_window = 1;
	uint32_t window_shift = 0;     // log2(positions per window)
	uint32_t bins = 1024; // level-0 bins over ONE window of the global position space
	uint32_t shift0 = 0;  // log2(positions per level-0 bin)
	uint32_t bins_per_shard = 0;
	uint32_t regions = 0;
	uint32_t cap = 0;
	uint64_t ent_bytes_per_shard = 0, cnt_bytes_per_shard = 0;
};

// An entry is the offset of a position inside its level-0 bin and has 32 bits; pass A stages at most 1024
// bins.  So one routing pass covers at most 2^42 positions: a larger filter (C4: 2^43 bits on 8 GPUs) is
// routed in WINDOWS of 2^42 positions, one pass A per window over the same reads (its WINDOW variant keeps
// the probes inside the window; the hashing is repeated, the partitioning is not).  A window is owned by
// n_shards / n_windows consecutive shards, and only they receive blocks of that window's pass.
int route_plan(const btlbf_filter* f, uint64_t len, const LayoutParams& lay, unsigned n_shards, RoutePlan& rp)
{
	const uint64_t M = f->mod.size;
	if (!f->mod.pow2 || n_shards == 0 || (n_shards & (n_shards - 1)) || n_shards > 1024)
		return fail(BTLBF_EINVAL, "routing needs a filter whose global size and shard count are powers of two");
	if (!part_supported(f->hp) || !part_hash_fits(f->hp, 1024))
		return fail(BTLBF_EINVAL, "routing does not support this hash configuration");
	const unsigned lm = ceil_log2(M);
	unsigned max_window = 42; // BTLBF_ROUTE_WINDOW_BITS exists for tests: small filters then exercise several windows
	if (const char* e = getenv("BTLBF_ROUTE_WINDOW_BITS")) {
		const int v = atoi(e);
		if (v >= 20 && v <= 42)
			max_window = (unsigned)v;
	}
	rp.window_shift = std::min(lm, max_window);
	rp.n_windows = 1u << (lm - rp.window_shift);
	if (rp.n_windows > n_shards)
		return fail(BTLBF_EINVAL, "routing a 2^%u-bit filter needs at least %u shards (windows of 2^%u positions)", lm,
		            rp.n_windows, rp.window_shift);
	rp.shards_per_window = n_shards / rp.n_windows;
	// 512 level-0 bins per window (64-entry LDS rings at the origin: few late entries) as long as an entry
	// fits 32 bits, else 1024.  BTLBF_ROUTE_BINS (power of two) exists for tests: fewer bins make small
	// filters exercise the two-split and 32-bit-entry geometries of a 1 TiB filter on 8 GPUs
	rp.bins = rp.window_shift - 9 <= 32 && rp.shards_per_window <= 512 ? 512 : 1024;
	if (const char* e = getenv("BTLBF_ROUTE_BINS")) {
		const unsigned b = (unsigned)atoi(e);
		if (b >= rp.shards_per_window && b <= 1024 && !(b & (b - 1)))
			rp.bins = b;
	}
	const unsigned lw = rp.window_shift, lb = ceil_log2(rp.bins);
	const unsigned seg_min = f->kind == BTLBF_COUNTING8 ? 16 : 19; // positions in a 64 KiB segment
	if (lw < lb + seg_min || lw - lb > 32 || rp.bins < rp.shards_per_window)
		return fail(BTLBF_EINVAL, "routing supports global filters of at least 2^29 bits (2^26 counters)");
	rp.shift0 = lw - lb;
	rp.bins_per_shard = rp.bins / rp.shards_per_window;
	rp.regions = part_hash_regions(f->hp, rp.bins, cu_count(f->device));
	const PartTiling tl = part_tiling(f->hp, rp.bins, lay, len);
	const double entries = (double)tiles_for_caps(tl.n_tiles, rp.regions) * probes_per_tile(f, tl) / rp.n_windows;
	rp.cap = chunks_for(entries / ((double)rp.bins * rp.regions), 1);
	rp.ent_bytes_per_shard = (uint64_t)rp.bins_per_shard * rp.regions * rp.cap * (kChunk * 4);
	rp.cnt_bytes_per_shard = (uint64_t)rp.bins_per_shard * rp.regions * 4;
	return BTLBF_OK;
}

} // namespace

extern "C" int btlbf_route_plan(btlbf_filter* f, uint64_t len, const btlbf_layout* layout, unsigned n_shards,
                                uint64_t* ent_bytes_per_shard, uint64_t* cnt_bytes_per_shard)
{
	FilterLock lk__(f);
	if (!f || !ent_bytes_per_shard || !cnt_bytes_per_shard)
		return fail(BTLBF_EINVAL, "null argument");
	LayoutParams lay{nullptr, 0, 0};
	if (layout) {
		lay.starts = layout->starts;
		lay.n_seqs = layout->n_seqs;
		lay.read_len = layout->starts ? 0 : layout->read_len;
	}
	RoutePlan rp;
	int rc = route_plan(f, len, lay, n_shards, rp);
	if (rc)
		return rc;
	*ent_bytes_per_shard = rp.ent_bytes_per_shard;
	*cnt_bytes_per_shard = rp.cnt_bytes_per_shard;
	return BTLBF_OK;
}

// planning only (no device needed): the read grid pass A would use for fixed-length reads
extern "C" int btlbf_plan_read_grid(unsigned kmer_size, unsigned hash_num, unsigned read_len, unsigned level0_bins,
                                    uint32_t* out4)
{
	if (!out4 || kmer_size == 0 || hash_num == 0 || level0_bins == 0 || level0_bins > 1024)
		return fail(BTLBF_EINVAL, "btlbf_plan_read_grid: bad argument");
	HashParams hp;
	fill_hash_params(hp, kmer_size, hash_num);
	PartGrid g;
	(void)part_read_grid(hp, level0_bins, LayoutParams{nullptr, 0, read_len}, &g);
	out4[0] = g.reads;
	out4[1] = g.gpr;
	out4[2] = g.lpad;
	out4[3] = g.cap;
	return BTLBF_OK;
}

extern "C" int btlbf_route_windows(btlbf_filter* f, unsigned n_shards, unsigned* n_windows,
                                   unsigned* shards_per_window)
{
	FilterLock lk__(f);
	if (!f || !n_windows || !shards_per_window)
		return fail(BTLBF_EINVAL, "null argument");
	RoutePlan rp;
	int rc = route_plan(f, 1, LayoutParams{nullptr, 0, 0}, n_shards, rp);
	if (rc)
		return rc;
	*n_windows = rp.n_windows;
	*shards_per_window = rp.shards_per_window;
	return BTLBF_OK;
}

extern "C" int btlbf_route_seqs(btlbf_filter* f, const char* seq, uint64_t len, const btlbf_layout* layout,
                                uint64_t plan_len, unsigned n_shards, unsigned window, int query, void* send_ent,
                                void* send_cnt, uint64_t* hit_bits, uint64_t* valid_bits, uint64_t* counts,
                                uint64_t* spill_list, uint64_t spill_cap, uint64_t* spill_count, void* stream)
{
	FilterLock lk__(f);
	int rc = seq_precheck(f, len);
	if (rc)
		return rc;
	if (!send_ent || !send_cnt || !spill_list || !spill_count)
		return fail(BTLBF_EINVAL, "null argument");
	DeviceGuard g(f->device);
	hipStream_t s = static_cast<hipStream_t>(stream);
	SeqView v;
	if ((rc = make_view(v, seq, len, layout, BTLBF_DEVICE, s)))
		return rc;
	RoutePlan rp;
	if ((rc = route_plan(f, plan_len, v.lay, n_shards, rp)))
		return rc;
	if (window >= rp.n_windows)
		return fail(BTLBF_EINVAL, "window %u of %u", window, rp.n_windows);
	SeqArgs a = base_args(f, v, len);
	// positions of the GLOBAL filter, those inside this window (all of them when there is one window)
	fill_mod(a.mod, f->mod.size, (uint64_t)window << rp.window_shift, 1ull << rp.window_shift);
	a.hit_bits = reinterpret_cast<uint8_t*>(hit_bits);
	a.valid_bits = reinterpret_cast<uint8_t*>(valid_bits);
	a.counts = counts;
	a.first_tile = 0;
	a.n_tiles = part_tiling(f->hp, rp.bins, v.lay, len).n_tiles;
	PartOut out{rp.bins, rp.regions, rp.cap, static_cast<uint32_t*>(send_cnt), static_cast<uint32_t*>(send_ent)};
	PartSide sd;
	memset(&sd, 0, sizeof sd);
	sd.spill_list = spill_list;
	sd.spill_count = reinterpret_cast<unsigned long long*>(spill_count);
	sd.spill_cap = spill_cap;
	sd.pos_base = a.mod.shard_lo; // spilled entries travel as global positions
	// spill_count and counts ACCUMULATE over the batches of a pass (the caller zeroes them once): no
	// host round trip per batch, so the exchange of one batch can overlap the hashing of the next
	if (a.n_tiles == 0) { // nothing to hash: still publish empty regions
		HIP_TRY(hipMemsetAsync(send_cnt, 0, (size_t)rp.cnt_bytes_per_shard * rp.shards_per_window, s));
		return BTLBF_OK;
	}
	ProfSpan ps(f, query ? BTLBF_PROF_QUERY_HASH : BTLBF_PROF_INSERT_HASH, s);
	HIP_TRY(launch_part_hash(a, out, rp.shift0, sd, query, s));
	return BTLBF_OK;
}

namespace {

// the owner's plan for blocks routed with `rp`: split levels below the level-0 bins of this shard
int owner_plan(btlbf_filter* f, const RoutePlan& rp, const LayoutParams& lay, uint64_t plan_len, unsigned n_blocks,
               unsigned n_shards, PartPlan& pl)
{
	if (!plan_segments(f->mod.shard_len, pl, f->kind == BTLBF_COUNTING8 ? 0 : 3))
		return fail(BTLBF_EINVAL, "shard too large for the partitioned pipeline");
	pl.lv[0].bins = rp.bins_per_shard;
	pl.lv[0].shift = rp.shift0;
	pl.lv[0].regions = rp.regions * n_blocks;
	if (pl.lv[0].shift < pl.seg_shift || !plan_splits(pl, pl.lv[0].regions, cu_count(f->device)))
		return fail(BTLBF_EINVAL, "unsupported shard geometry");
	// every origin sends about entries/n_shards to this shard; n_blocks origins
	const PartTiling tl = part_tiling(f->hp, rp.bins, lay, plan_len);
	const double entries = (double)tiles_for_caps(tl.n_tiles, rp.regions) * probes_per_tile(f, tl) * n_blocks / n_shards;
	plan_caps(pl, entries, 1);
	return BTLBF_OK;
}

LayoutParams layout_params(const btlbf_layout* layout)
{
	LayoutParams lay{nullptr, 0, 0};
	if (layout) {
		lay.n_seqs = layout->n_seqs;
		lay.read_len = layout->starts ? 0 : layout->read_len;
		lay.starts = layout->starts;
	}
	return lay;
}

} // namespace

extern "C" int btlbf_route_geometry(btlbf_filter* f, uint64_t plan_len, const btlbf_layout* layout, unsigned n_shards,
                                    unsigned n_blocks, uint32_t* out4)
{
	FilterLock lk__(f);
	if (!f || !out4)
		return fail(BTLBF_EINVAL, "null argument");
	const LayoutParams lay = layout_params(layout);
	RoutePlan rp;
	int rc = route_plan(f, plan_len, lay, n_shards, rp);
	if (rc)
		return rc;
	PartPlan pl;
	if ((rc = owner_plan(f, rp, lay, plan_len, n_blocks ? n_blocks : n_shards, n_shards, pl)))
		return rc;
	out4[0] = rp.bins_per_shard;
	out4[1] = rp.regions;
	out4[2] = rp.cap;
	out4[3] = pl.n_levels >= 2 && pl.group_bins ? pl.group_bins : rp.bins_per_shard;
	return BTLBF_OK;
}

extern "C" int btlbf_owner_scratch_bytes(btlbf_filter* f, uint64_t plan_len, const btlbf_layout* layout, unsigned n_shards,
                                         unsigned n_blocks, uint64_t* bytes)
{
	FilterLock lk__(f);
	if (!f || !bytes)
		return fail(BTLBF_EINVAL, "null argument");
	const LayoutParams lay = layout_params(layout);
	RoutePlan rp;
	int rc = route_plan(f, plan_len, lay, n_shards, rp);
	if (rc)
		return rc;
	PartPlan pl;
	if ((rc = owner_plan(f, rp, lay, plan_len, n_blocks ? n_blocks : n_shards, n_shards, pl)))
		return rc;
	*bytes = pl.bytes_total;
	return BTLBF_OK;
}

extern "C" int btlbf_apply_routed_bins(btlbf_filter* f, const void* recv_ent, const void* recv_cnt, unsigned n_blocks,
                                       unsigned first_bin, unsigned n_bins, uint64_t plan_len,
                                       const btlbf_layout* layout, unsigned n_shards, int query, uint64_t* fail_list,
                                       uint64_t fail_cap, uint64_t* fail_count, void* stream)
{
	FilterLock lk__(f);
	if (!f || !recv_ent || !recv_cnt || n_blocks == 0)
		return fail(BTLBF_EINVAL, "null argument");
	if (f->shard_count != n_shards)
		return fail(BTLBF_EINVAL, "filter is shard %u of %u, not of %u", f->shard_index, f->shard_count, n_shards);
	if (query && (!fail_list || !fail_count))
		return fail(BTLBF_EINVAL, "query needs a fail list");
	DeviceGuard g(f->device);
	hipStream_t s = static_cast<hipStream_t>(stream);
	MATERIALIZE(f, s);
	const LayoutParams lay = layout_params(layout);
	RoutePlan rp;
	int rc = route_plan(f, plan_len, lay, n_shards, rp);
	if (rc)
		return rc;
	PartPlan pl;
	if ((rc = owner_plan(f, rp, lay, plan_len, n_blocks, n_shards, pl)))
		return rc;
	const uint32_t group = pl.n_levels >= 2 && pl.group_bins ? pl.group_bins : rp.bins_per_shard;
	if (n_bins == 0 || first_bin + n_bins > rp.bins_per_shard || first_bin % group || (n_bins % group && first_bin + n_bins != rp.bins_per_shard))
		return fail(BTLBF_EINVAL, "bins [%u, +%u) are not whole groups of %u of this shard's %u level-0 bins", first_bin,
		            n_bins, group, rp.bins_per_shard);
	bool ok = false;
	if ((rc = ensure_scratch(f, pl.bytes_total, &ok)))
		return rc;
	if (!ok)
		return fail(BTLBF_ENOMEM, "no room for %llu bytes of partition scratch", (unsigned long long)pl.bytes_total);
	carve_levels(pl, static_cast<uint8_t*>(f->d_part), 1);
	PartSide sd;
	memset(&sd, 0, sizeof sd);
	sd.pos_base = f->mod.shard_lo;
	sd.fail_list = fail_list;
	sd.fail_count = reinterpret_cast<unsigned long long*>(fail_count);
	sd.fail_cap = fail_cap;
	sd.counting = f->kind == BTLBF_COUNTING8; // incrementAll / counter >= threshold at the owner
	sd.threshold = f->thr;
	PartIn in0{n_blocks, n_bins, rp.regions, rp.cap, static_cast<const uint32_t*>(recv_cnt),
	           static_cast<const uint32_t*>(recv_ent)};
	return run_levels(f, pl, in0, sd, query, s, first_bin, n_bins);
}

extern "C" int btlbf_apply_routed(btlbf_filter* f, const void* recv_ent, const void* recv_cnt, unsigned n_blocks,
                                  uint64_t plan_len, const btlbf_layout* layout, unsigned n_shards, int query,
                                  uint64_t* fail_list, uint64_t fail_cap, uint64_t* fail_count, void* stream)
{
	FilterLock lk__(f);
	if (!f)
		return fail(BTLBF_EINVAL, "null argument");
	const LayoutParams lay = layout_params(layout);
	RoutePlan rp;
	int rc = route_plan(f, plan_len, lay, n_shards, rp);
	if (rc)
		return rc;
	return btlbf_apply_routed_bins(f, recv_ent, recv_cnt, n_blocks, 0, rp.bins_per_shard, plan_len, layout, n_shards,
	                               query, fail_list, fail_cap, fail_count, stream);
}

extern "C" int btlbf_apply_spill(btlbf_filter* f, const uint64_t* global_pos, uint64_t n, int query,
                                 uint64_t* fail_list, uint64_t fail_cap, uint64_t* fail_count, void* stream)
{
	FilterLock lk__(f);
	if (!f || (n && !global_pos))
		return fail(BTLBF_EINVAL, "null argument");
	DeviceGuard g(f->device);
	MATERIALIZE(f, stream);
	PartSide sd;
	memset(&sd, 0, sizeof sd);
	sd.fail_list = fail_list;
	sd.fail_count = reinterpret_cast<unsigned long long*>(fail_count);
	sd.fail_cap = fail_cap;
	sd.counting = f->kind == BTLBF_COUNTING8;
	sd.threshold = f->thr;
	HIP_TRY(launch_spill(f->d_data, global_pos, n, f->mod.shard_lo, f->mod.shard_len, query, sd,
	                     static_cast<hipStream_t>(stream)));
	return BTLBF_OK;
}

extern "C" int btlbf_resolve_seqs(btlbf_filter* f, const char* seq, uint64_t len, const btlbf_layout* layout,
                                  const uint64_t* fail_list, uint64_t n_fail, uint64_t* hit_bits, void* stream)
{
	FilterLock lk__(f);
	int rc = seq_precheck(f, len);
	if (rc)
		return rc;
	if (!hit_bits || (n_fail && !fail_list))
		return fail(BTLBF_EINVAL, "null argument");
	if (n_fail == 0 || len == 0)
		return BTLBF_OK;
	if (n_fail > kFailCap)
		return fail(BTLBF_EINVAL, "more than %llu failed positions: use the direct query", (unsigned long long)kFailCap);
	DeviceGuard g(f->device);
	hipStream_t s = static_cast<hipStream_t>(stream);
	SeqView v;
	if ((rc = make_view(v, seq, len, layout, BTLBF_DEVICE, s)))
		return rc;
	bool ok = false;
	if ((rc = ensure_scratch(f, kFailTableSlots * 8, &ok)))
		return rc;
	if (!ok)
		return fail(BTLBF_ENOMEM, "no room for the failed-position set");
	SeqArgs a = base_args(f, v, len);
	fill_mod(a.mod, f->mod.size, 0, f->mod.size); // global positions
	return resolve_range(f, a, reinterpret_cast<uint8_t*>(hit_bits), fail_list, n_fail,
	                     static_cast<uint64_t*>(f->d_part), kFailTableSlots, 0, 0, s);
}

namespace {
} // namespace

extern "C" int btlbf_insert_seqs(btlbf_filter* f, const char* seq, uint64_t len,
                                 const btlbf_layout* layout, int op, int order, int mem, void* stream)
{
	FilterLock lk__(f);
	int rc = seq_precheck(f, len);
	if (rc)
		return rc;
	DeviceGuard g(f->device);
	hipStream_t s = static_cast<hipStream_t>(stream);
	SeqView v;
	rc = make_view(v, seq, len, layout, mem, s);
	if (rc)
		return rc;
	SeqArgs a = base_args(f, v, len);
	int kop;
	if (f->kind == BTLBF_BLOOM) {
		kop = OP_BF_INSERT; // bit OR is order-free: serial order would give the same bytes
		if (want_partitioned(f, len)) {
			bool done = false;
			rc = partitioned_insert(f, a, s, &done);
			if (rc)
				return rc;
			if (done) {
				if (mem == BTLBF_HOST)
					HIP_TRY(hipStreamSynchronize(s));
				return BTLBF_OK;
			}
		}
		MATERIALIZE(f, s);
	} else {
		// a counting shard keeps the increments inside its window; the conservative update needs all h
		// counters of a k-mer, which live on different shards
		if (f->shard_count != 1 && (op != BTLBF_INCREMENT_ALL || order == BTLBF_ORDER_SERIAL))
			return fail(BTLBF_EINVAL, "a counting-filter shard takes incrementAll in parallel order only");
		if (op != BTLBF_INCREMENT_MIN && op != BTLBF_INCREMENT_ALL)
			return fail(BTLBF_EINVAL, "op must be BTLBF_INCREMENT_MIN or BTLBF_INCREMENT_ALL");
		kop = op == BTLBF_INCREMENT_MIN ? OP_CBF_INC_MIN : OP_CBF_INC_ALL;
		if (order != BTLBF_ORDER_SERIAL && want_partitioned(f, len, op)) {
			// incrementAll is order-free up to saturation, which is order-free too: exact in any order
			bool done = false;
			rc = partitioned_insert(f, a, s, &done);
			if (rc)
				return rc;
			if (done) {
				if (mem == BTLBF_HOST)
					HIP_TRY(hipStreamSynchronize(s));
				return BTLBF_OK;
			}
		}
		MATERIALIZE(f, s);
		if (order == BTLBF_ORDER_SERIAL) {
			// hash on all CUs, then apply the rows in buffer order on a single lane
			DevBuf hashes, valid;
			HIP_TRY(hashes.alloc(len * f->h * 8));
			HIP_TRY(valid.alloc(bitmap_bytes(len)));
			a.hashes = hashes.as<uint64_t>();
			a.valid_bits = valid.as<uint8_t>();
			HIP_TRY(launch_seq_op(OP_HASH_ONLY, a, s));
			HIP_TRY(launch_serial_seq_update(a, op == BTLBF_INCREMENT_MIN ? H_CBF_INC_MIN : H_CBF_INC_ALL,
			                                 hashes.as<uint64_t>(), valid.as<uint8_t>(), nullptr, s));
			HIP_TRY(hipStreamSynchronize(s));
			return BTLBF_OK;
		}
	}
	{
		REQUIRE_MATERIALIZED(f);
		ProfSpan ps(f, kop == OP_BF_INSERT ? BTLBF_PROF_INSERT_DIRECT : BTLBF_PROF_OTHER, s);
		HIP_TRY(launch_seq_op(kop, a, s));
	}
	if (mem == BTLBF_HOST)
		HIP_TRY(hipStreamSynchronize(s));
	return BTLBF_OK;
}

extern "C" int btlbf_contains_seqs(btlbf_filter* f, const char* seq, uint64_t len,
                                   const btlbf_layout* layout, uint64_t* hit_bits, uint64_t* valid_bits,
                                   uint64_t* counts, int mem, void* stream)
{
	FilterLock lk__(f);
	if (!f)
		return fail(BTLBF_EINVAL, "null filter");
	return run_query_like(f, f->kind == BTLBF_BLOOM ? OP_BF_CONTAINS : OP_CBF_QUERY, seq, len, layout,
	                      hit_bits, valid_bits, counts, nullptr, mem, stream, &lk__);
}

extern "C" int btlbf_insert_and_check_seqs(btlbf_filter* f, const char* seq, uint64_t len,
                                           const btlbf_layout* layout, uint64_t* hit_bits,
                                           uint64_t* valid_bits, uint64_t* counts, int mem, void* stream)
{
	FilterLock lk__(f);
	if (!f)
		return fail(BTLBF_EINVAL, "null filter");
	if (f->kind != BTLBF_BLOOM)
		return fail(BTLBF_EINVAL, "insert_and_check_seqs: bit filters only (use the hash-row form for counting)");
	return run_query_like(f, OP_BF_INSERT_CHECK, seq, len, layout, hit_bits, valid_bits, counts, nullptr, mem,
	                      stream);
}

extern "C" int btlbf_min_count_seqs(btlbf_filter* f, const char* seq, uint64_t len,
                                    const btlbf_layout* layout, uint8_t* min_out, uint64_t* valid_bits,
                                    int mem, void* stream)
{
	FilterLock lk__(f);
	if (!f)
		return fail(BTLBF_EINVAL, "null filter");
	if (f->kind != BTLBF_COUNTING8)
		return fail(BTLBF_EINVAL, "min_count needs a counting filter");
	return run_query_like(f, OP_CBF_QUERY, seq, len, layout, nullptr, valid_bits, nullptr, min_out, mem, stream);
}

// -------------------------------------------------------------------------------------------------
// precomputed hash rows
// -------------------------------------------------------------------------------------------------
namespace {

int run_hash_rows(btlbf_filter* f, int hop, const uint64_t* hashes, uint64_t n, uint8_t* out, int serial,
                  int mem, void* stream, FilterLock* lk = nullptr)
{
	if (!f)
		return fail(BTLBF_EINVAL, "null filter");
	if (n && !hashes)
		return fail(BTLBF_EINVAL, "null hashes");
	if (f->shard_count != 1 && hop != H_BF_INSERT)
		return fail(BTLBF_EINVAL, "only insert is defined on a single shard");
	DeviceGuard g(f->device);
	hipStream_t s = static_cast<hipStream_t>(stream);
	MATERIALIZE(f, s);
	// a few rows from host memory (the shims' per-k-mer contains / insertAndCheck / minCount): through the calling
	// thread's pinned mailbox -- no staging buffers, no copies, one launch and one synchronisation
	const uint64_t row_bytes = (n * f->h * 8 + 15) & ~(uint64_t)15;
	if (mem == BTLBF_HOST && n && row_bytes + n <= Mailbox::kBytes && mailbox().get()) {
		Mailbox& mb = mailbox();
		memcpy(mb.host, hashes, n * f->h * 8);
		REQUIRE_MATERIALIZED(f);
		HIP_TRY(launch_hash_op(hop, f->d_data, f->mod, f->h, f->thr, reinterpret_cast<const uint64_t*>(mb.dev), n,
		                       out ? mb.dev + row_bytes : nullptr, serial, s));
		if (lk && (hop == H_BF_CONTAINS || hop == H_CBF_CONTAINS || hop == H_CBF_MIN))
			lk->release(); // the mailbox and (with BTLBF_STREAM_PER_THREAD) the stream are the calling thread's own
		HIP_TRY(hipStreamSynchronize(s));
		if (out)
			memcpy(out, mb.host + row_bytes, n);
		return BTLBF_OK;
	}
	DevBuf hb;
	const uint64_t* d_h = hashes;
	if (mem == BTLBF_HOST) {
		HIP_TRY(hb.alloc_pooled(n * f->h * 8));
		if (n)
			HIP_TRY(hipMemcpyAsync(hb.p, hashes, n * f->h * 8, hipMemcpyHostToDevice, s));
		d_h = hb.as<uint64_t>();
	}
	OutBuf ob;
	int rc = ob.prepare(out, n, mem, false, s);
	if (rc)
		return rc;
	REQUIRE_MATERIALIZED(f);
	HIP_TRY(launch_hash_op(hop, f->d_data, f->mod, f->h, f->thr, d_h, n, static_cast<uint8_t*>(ob.d), serial, s));
	if ((rc = ob.finish(s)))
		return rc;
	if (mem == BTLBF_HOST)
		HIP_TRY(hipStreamSynchronize(s));
	return BTLBF_OK;
}

} // namespace

extern "C" int btlbf_insert_hashes(btlbf_filter* f, const uint64_t* hashes, uint64_t n, int op, int order,
                                   int mem, void* stream)
{
	FilterLock lk__(f);
	if (!f)
		return fail(BTLBF_EINVAL, "null filter");
	int hop = H_BF_INSERT;
	if (f->kind == BTLBF_COUNTING8) {
		if (op != BTLBF_INCREMENT_MIN && op != BTLBF_INCREMENT_ALL)
			return fail(BTLBF_EINVAL, "op must be BTLBF_INCREMENT_MIN or BTLBF_INCREMENT_ALL");
		hop = op == BTLBF_INCREMENT_MIN ? H_CBF_INC_MIN : H_CBF_INC_ALL;
	}
	const int serial = f->kind == BTLBF_COUNTING8 && order == BTLBF_ORDER_SERIAL;
	return run_hash_rows(f, hop, hashes, n, nullptr, serial, mem, stream);
}

extern "C" int btlbf_contains_hashes(btlbf_filter* f, const uint64_t* hashes, uint64_t n, uint8_t* out,
                                     int mem, void* stream)
{
	FilterLock lk__(f);
	if (!f || !out)
		return fail(BTLBF_EINVAL, "null argument");
	return run_hash_rows(f, f->kind == BTLBF_BLOOM ? H_BF_CONTAINS : H_CBF_CONTAINS, hashes, n, out, 0, mem,
	                     stream, &lk__);
}

extern "C" int btlbf_insert_and_check_hashes(btlbf_filter* f, const uint64_t* hashes, uint64_t n,
                                             uint8_t* out, int order, int mem, void* stream)
{
	FilterLock lk__(f);
	if (!f || !out)
		return fail(BTLBF_EINVAL, "null argument");
	return run_hash_rows(f, f->kind == BTLBF_BLOOM ? H_BF_INSERT_CHECK : H_CBF_INSERT_CHECK, hashes, n, out,
	                     order == BTLBF_ORDER_SERIAL, mem, stream);
}

extern "C" int btlbf_min_count_hashes(btlbf_filter* f, const uint64_t* hashes, uint64_t n, uint8_t* min_out,
                                      int mem, void* stream)
{
	FilterLock lk__(f);
	if (!f || !min_out)
		return fail(BTLBF_EINVAL, "null argument");
	if (f->kind != BTLBF_COUNTING8)
		return fail(BTLBF_EINVAL, "min_count needs a counting filter");
	return run_hash_rows(f, H_CBF_MIN, hashes, n, min_out, 0, mem, stream);
}

// -------------------------------------------------------------------------------------------------
// raw k-mers: KmerBloomFilter::insert / contains(const char*) (KmerBloomFilter.hpp:47-74)
// -------------------------------------------------------------------------------------------------
namespace {

// n k-mers of k bytes each -> device hash rows + valid bytes (aux_kernels.hip, kmer_rows_kernel)
struct KmerRows {
	DevBuf seq, rows, valid;
	const uint8_t* d_seq = nullptr;
	int prepare(const char* kmers, uint64_t n, unsigned k, unsigned h, uint64_t kms, int mem, hipStream_t s)
	{
		if (n && !kmers)
			return fail(BTLBF_EINVAL, "null kmers");
		d_seq = reinterpret_cast<const uint8_t*>(kmers);
		if (mem == BTLBF_HOST) {
			HIP_TRY(seq.alloc_pooled(n * k));
			if (n)
				HIP_TRY(hipMemcpyAsync(seq.p, kmers, n * k, hipMemcpyHostToDevice, s));
			d_seq = seq.as<uint8_t>();
		}
		HIP_TRY(rows.alloc_pooled(n * h * 8));
		HIP_TRY(valid.alloc_pooled(n));
		HIP_TRY(launch_kmer_rows(d_seq, n, k, h, kms, rows.as<uint64_t>(), valid.as<uint8_t>(), s));
		return BTLBF_OK;
	}
};

int run_kmer_rows(btlbf_filter* f, int hop, const char* kmers, uint64_t n, uint8_t* out, int serial, int mem,
                  void* stream)
{
	if (f->hp.n_seeds)
		return fail(BTLBF_EINVAL, "raw k-mers are hashed with ntHash, not with spaced seeds");
	if (f->shard_count != 1 && hop != H_BF_INSERT)
		return fail(BTLBF_EINVAL, "only insert is defined on a single shard");
	DeviceGuard g(f->device);
	hipStream_t s = static_cast<hipStream_t>(stream);
	MATERIALIZE(f, s);
	KmerRows kr;
	int rc = kr.prepare(kmers, n, f->k, f->h, f->hp.kms, mem, s);
	if (rc)
		return rc;
	OutBuf ob;
	if ((rc = ob.prepare(out, n, mem, false, s)))
		return rc;
	HIP_TRY(launch_hash_op(hop, f->d_data, f->mod, f->h, f->thr, kr.rows.as<uint64_t>(), n, static_cast<uint8_t*>(ob.d),
	                       serial, s, kr.valid.as<uint8_t>()));
	if ((rc = ob.finish(s)))
		return rc;
	HIP_TRY(hipStreamSynchronize(s)); // the temporaries are freed on return
	return BTLBF_OK;
}

} // namespace

extern "C" int btlbf_insert_kmers(btlbf_filter* f, const char* kmers, uint64_t n, int op, int order, int mem,
                                  void* stream)
{
	FilterLock lk__(f);
	if (!f)
		return fail(BTLBF_EINVAL, "null filter");
	int hop = H_BF_INSERT;
	if (f->kind == BTLBF_COUNTING8) {
		if (op != BTLBF_INCREMENT_MIN && op != BTLBF_INCREMENT_ALL)
			return fail(BTLBF_EINVAL, "op must be BTLBF_INCREMENT_MIN or BTLBF_INCREMENT_ALL");
		hop = op == BTLBF_INCREMENT_MIN ? H_CBF_INC_MIN : H_CBF_INC_ALL;
	}
	const int serial = f->kind == BTLBF_COUNTING8 && order == BTLBF_ORDER_SERIAL;
	return run_kmer_rows(f, hop, kmers, n, nullptr, serial, mem, stream);
}

extern "C" int btlbf_contains_kmers(btlbf_filter* f, const char* kmers, uint64_t n, uint8_t* out, int mem,
                                    void* stream)
{
	FilterLock lk__(f);
	if (!f || !out)
		return fail(BTLBF_EINVAL, "null argument");
	return run_kmer_rows(f, f->kind == BTLBF_BLOOM ? H_BF_CONTAINS : H_CBF_CONTAINS, kmers, n, out, 0, mem, stream);
}

extern "C" int btlbf_hash_kmers(unsigned kmer_size, unsigned hash_num, const char* kmers, uint64_t n,
                                uint64_t* hashes, uint8_t* valid, int mem, int device, void* stream)
{
	if (kmer_size == 0 || kmer_size > 32768 || hash_num == 0 || hash_num > 64)
		return fail(BTLBF_EINVAL, "bad kmer_size / hash_num");
	if (!hashes)
		return fail(BTLBF_EINVAL, "null hashes output");
	if (btlbf_device_count() <= device || device < 0)
		return fail(BTLBF_EHIP, "no GPU %d (visible devices: %d): this library has no CPU path", device,
		            btlbf_device_count());
	DeviceGuard g(device);
	hipStream_t s = static_cast<hipStream_t>(stream);
	KmerRows kr;
	int rc = kr.prepare(kmers, n, kmer_size, hash_num, (uint64_t)kmer_size * kMultiSeed, mem, s);
	if (rc)
		return rc;
	const hipMemcpyKind kind = mem == BTLBF_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
	if (n) {
		HIP_TRY(hipMemcpyAsync(hashes, kr.rows.p, n * hash_num * 8, kind, s));
		if (valid)
			HIP_TRY(hipMemcpyAsync(valid, kr.valid.p, n, kind, s));
	}
	HIP_TRY(hipStreamSynchronize(s));
	return BTLBF_OK;
}

// -------------------------------------------------------------------------------------------------
// hash streams only
// -------------------------------------------------------------------------------------------------
extern "C" int btlbf_hash_seqs(unsigned kmer_size, unsigned hash_num, const char* const* seeds,
                               unsigned n_seeds, unsigned h2, const char* seq, uint64_t len,
                               const btlbf_layout* layout, uint64_t* hashes, uint64_t* valid_bits,
                               uint64_t* strand_bits, int mem, int device, void* stream)
{
	if (kmer_size == 0 || kmer_size > 32768 || hash_num == 0)
		return fail(BTLBF_EINVAL, "bad kmer_size / hash_num");
	if (!seeds && hash_num > 64)
		return fail(BTLBF_EINVAL, "hash_num %u > 64 unsupported", hash_num);
	if (btlbf_device_count() <= device || device < 0)
		return fail(BTLBF_EHIP, "no GPU %d (visible devices: %d): this library has no CPU path", device,
		            btlbf_device_count());
	DeviceGuard g(device);
	hipStream_t s = static_cast<hipStream_t>(stream);
	HashParams hp;
	fill_hash_params(hp, kmer_size, hash_num);
	DevBuf pos_owner, dc_owner;
	if (seeds) {
		if (n_seeds * h2 != hash_num)
			return fail(BTLBF_EINVAL, "hash_num must equal n_seeds*h2");
		uint64_t* dp = nullptr;
		uint16_t* dd = nullptr;
		int rc = build_spaced(hp, seeds, n_seeds, h2, &dp, &dd);
		pos_owner.p = dp;
		dc_owner.p = dd;
		if (rc)
			return rc;
	}
	if (!hashes)
		return fail(BTLBF_EINVAL, "null hashes output");
	// small host-memory calls of plain ntHash (the drop-in ntHashIterator makes one per read): through the
	// calling thread's mailbox -- the kernel reads the bases from and writes the hash rows to pinned host memory
	const uint64_t up16 = ~(uint64_t)15;
	const uint64_t o_h = (len + 16 + 15) & up16, o_v = o_h + ((len * hash_num * 8 + 15) & up16),
	               o_end = o_v + ((bitmap_bytes(len) + 15) & up16);
	if (mem == BTLBF_HOST && !seeds && !strand_bits && len && (!layout || !layout->starts) && o_end <= Mailbox::kBytes &&
	    mailbox().get()) {
		int rc = check_layout(layout, len);
		if (rc)
			return rc;
		Mailbox& mb = mailbox();
		memcpy(mb.host, seq, len);
		SeqArgs a;
		memset(&a, 0, sizeof a);
		a.seq = mb.dev;
		a.len = len;
		a.layout.read_len = layout ? layout->read_len : 0;
		a.hp = hp;
		fill_mod(a.mod, 8, 0, 8);
		a.hashes = reinterpret_cast<uint64_t*>(mb.dev + o_h);
		a.valid_bits = valid_bits ? mb.dev + o_v : nullptr;
		HIP_TRY(launch_seq_op(OP_HASH_ONLY, a, s));
		HIP_TRY(hipStreamSynchronize(s));
		memcpy(hashes, mb.host + o_h, len * hash_num * 8);
		if (valid_bits)
			memcpy(valid_bits, mb.host + o_v, bitmap_bytes(len));
		return BTLBF_OK;
	}
	SeqView v;
	int rc = make_view(v, seq, len, layout, mem, s);
	if (rc)
		return rc;
	OutBuf ob_h, ob_v, ob_s;
	if ((rc = ob_h.prepare(hashes, len * hash_num * 8, mem, false, s)))
		return rc;
	if ((rc = ob_v.prepare(valid_bits, bitmap_bytes(len), mem, false, s)))
		return rc;
	if ((rc = ob_s.prepare(strand_bits, len * 8, mem, false, s)))
		return rc;
	SeqArgs a;
	memset(&a, 0, sizeof a);
	a.seq = v.d_seq;
	a.len = len;
	a.layout = v.lay;
	a.hp = hp;
	fill_mod(a.mod, 8, 0, 8);
	a.hashes = static_cast<uint64_t*>(ob_h.d);
	a.valid_bits = static_cast<uint8_t*>(ob_v.d);
	a.strand_bits = static_cast<uint64_t*>(ob_s.d);
	HIP_TRY(launch_seq_op(OP_HASH_ONLY, a, s));
	if ((rc = ob_h.finish(s)) || (rc = ob_v.finish(s)) || (rc = ob_s.finish(s)))
		return rc;
	HIP_TRY(hipStreamSynchronize(s)); // tables are freed on return
	return BTLBF_OK;
}

// -------------------------------------------------------------------------------------------------
// statistics
// -------------------------------------------------------------------------------------------------
static int popcount_mode(btlbf_filter* f, int mode, uint64_t* out)
{
	if (!f || !out)
		return fail(BTLBF_EINVAL, "null argument");
	DeviceGuard g(f->device);
	MATERIALIZE(f, nullptr);
	HIP_TRY(hipDeviceSynchronize()); // DEVICE-mode calls may have run on non-blocking user streams
	HIP_TRY(hipMemset(f->d_scalar, 0, 8));
	HIP_TRY(launch_popcount(f->d_data, f->alloc_bytes, mode, f->thr, f->d_scalar, nullptr));
	unsigned long long v = 0;
	HIP_TRY(hipMemcpy(&v, f->d_scalar, 8, hipMemcpyDeviceToHost));
	if (mode == 2 && f->thr == 0)
		v -= f->alloc_bytes - f->local_bytes; // zero padding also passes ">= 0"
	*out = v;
	return BTLBF_OK;
}

extern "C" int btlbf_popcount(btlbf_filter* f, uint64_t* out)
{
	FilterLock lk__(f);
	return popcount_mode(f, f && f->kind == BTLBF_COUNTING8 ? 1 : 0, out);
}

extern "C" int btlbf_filtered_popcount(btlbf_filter* f, uint64_t* out)
{
	FilterLock lk__(f);
	if (f && f->kind != BTLBF_COUNTING8)
		return fail(BTLBF_EINVAL, "filtered_popcount needs a counting filter");
	return popcount_mode(f, 2, out);
}

extern "C" int btlbf_digest(btlbf_filter* f, uint64_t* out2)
{
	FilterLock lk__(f);
	if (!f || !out2)
		return fail(BTLBF_EINVAL, "null argument");
	// the first local position must start a 64-bit word of the whole array (shards are cut at multiples of 64)
	const uint64_t per_word = f->kind == BTLBF_BLOOM ? 64 : 8;
	if (f->mod.shard_lo % per_word)
		return fail(BTLBF_EINVAL, "digest: the shard does not start on a 64-bit word of the filter");
	DeviceGuard g(f->device);
	MATERIALIZE(f, nullptr);
	HIP_TRY(hipDeviceSynchronize()); // DEVICE-mode calls may have run on non-blocking user streams
	HIP_TRY(hipMemset(f->d_scalar, 0, 16));
	HIP_TRY(launch_digest(f->d_data, f->alloc_bytes, f->mod.shard_lo / per_word, f->d_scalar, nullptr));
	HIP_TRY(hipMemcpy(out2, f->d_scalar, 16, hipMemcpyDeviceToHost));
	return BTLBF_OK;
}

extern "C" int btlbf_compare(btlbf_filter* a, btlbf_filter* b, uint64_t* out3)
{
	if (!a || !b || !out3)
		return fail(BTLBF_EINVAL, "null argument");
	FilterLock lk1__(a < b ? a : b), lk2__(a == b ? nullptr : (a < b ? b : a));
	if (a->kind != b->kind || a->size != b->size || a->local_bytes != b->local_bytes ||
	    a->mod.shard_lo != b->mod.shard_lo || a->device != b->device)
		return fail(BTLBF_EINVAL, "btlbf_compare: the two filters differ in kind, size, shard range or device");
	DeviceGuard g(a->device);
	MATERIALIZE(a, nullptr);
	MATERIALIZE(b, nullptr);
	HIP_TRY(hipDeviceSynchronize()); // whatever streams the two filters were last used on
	DevBuf acc;
	HIP_TRY(acc.alloc(24));
	HIP_TRY(hipMemset(acc.p, 0, 24));
	HIP_TRY(launch_compare(a->d_data, b->d_data, a->alloc_bytes, a->kind == BTLBF_COUNTING8,
	                       acc.as<unsigned long long>(), nullptr));
	unsigned long long v[3] = {0, 0, 0};
	HIP_TRY(hipMemcpy(v, acc.p, 24, hipMemcpyDeviceToHost));
	out3[0] = v[0];
	out3[1] = v[1];
	out3[2] = v[2];
	return BTLBF_OK;
}

// -------------------------------------------------------------------------------------------------
// rank structure (miBF stage 2)
// -------------------------------------------------------------------------------------------------
struct btlbf_rank {
	int device = 0;
	uint64_t n_bits = 0, n_blocks = 0, ones = 0;
	ModParams mod{};
	uint64_t* d_il = nullptr; // n_blocks records of 9 uint64_t
};

extern "C" int btlbf_rank_create(btlbf_rank** out, btlbf_filter* f)
{
	if (!out || !f)
		return fail(BTLBF_EINVAL, "null argument");
	*out = nullptr;
	if (f->kind != BTLBF_BLOOM || f->shard_count != 1)
		return fail(BTLBF_EINVAL, "rank structure: needs a whole bit filter");
	FilterLock lk__(f);
	DeviceGuard g(f->device);
	MATERIALIZE(f, nullptr);
	HIP_TRY(hipDeviceSynchronize());
	btlbf_rank* r = new btlbf_rank();
	r->device = f->device;
	r->n_bits = f->size;
	r->n_blocks = (f->size + 511) / 512;
	fill_mod(r->mod, f->size, 0, f->size);
	DevBuf scratch;
	hipError_t e = hipMalloc((void**)&r->d_il, r->n_blocks * 9 * 8 + 16);
	if (e == hipSuccess)
		e = scratch.alloc((r->n_blocks + (r->n_blocks + 4095) / 4096 + 2) * 8);
	if (e != hipSuccess) {
		(void)hipFree(r->d_il);
		delete r;
		(void)hipGetLastError();
		return fail(BTLBF_ENOMEM, "rank structure: %llu bytes of HBM", (unsigned long long)(r->n_blocks * 72));
	}
	uint64_t* total = scratch.as<uint64_t>() + r->n_blocks + (r->n_blocks + 4095) / 4096;
	e = launch_rank_build(static_cast<const uint64_t*>(f->d_data), f->size, r->d_il, scratch.as<uint64_t>(), total, nullptr);
	if (e == hipSuccess)
		e = hipMemcpy(&r->ones, total, 8, hipMemcpyDeviceToHost);
	if (e != hipSuccess) {
		(void)hipFree(r->d_il);
		delete r;
		return fail(BTLBF_EHIP, "rank structure: %s", hipGetErrorString(e));
	}
	*out = r;
	return BTLBF_OK;
}

extern "C" void btlbf_rank_destroy(btlbf_rank* r)
{
	if (!r)
		return;
	DeviceGuard g(r->device);
	(void)hipFree(r->d_il);
	delete r;
}

extern "C" uint64_t btlbf_rank_ones(const btlbf_rank* r) { return r ? r->ones : 0; }
extern "C" uint64_t btlbf_rank_words(const btlbf_rank* r) { return r ? r->n_blocks * 9 : 0; }

extern "C" int btlbf_rank_download(const btlbf_rank* r, uint64_t* host_dst)
{
	if (!r || !host_dst)
		return fail(BTLBF_EINVAL, "null argument");
	DeviceGuard g(r->device);
	HIP_TRY(hipMemcpy(host_dst, r->d_il, r->n_blocks * 72, hipMemcpyDeviceToHost));
	return BTLBF_OK;
}

extern "C" int btlbf_rank_query(const btlbf_rank* r, const uint64_t* values, uint64_t n, int values_are_hashes,
                                uint64_t* rank_out, uint8_t* bit_out, int mem, void* stream)
{
	if (!r || (n && !values))
		return fail(BTLBF_EINVAL, "null argument");
	if (mem != BTLBF_HOST && mem != BTLBF_DEVICE)
		return fail(BTLBF_EINVAL, "mem must be BTLBF_HOST or BTLBF_DEVICE");
	DeviceGuard g(r->device);
	hipStream_t s = static_cast<hipStream_t>(stream);
	DevBuf din;
	const uint64_t* dv = values;
	if (mem == BTLBF_HOST) {
		HIP_TRY(din.alloc(n * 8));
		if (n)
			HIP_TRY(hipMemcpyAsync(din.p, values, n * 8, hipMemcpyHostToDevice, s));
		dv = din.as<uint64_t>();
	}
	OutBuf o_rank, o_bit;
	int rc;
	if ((rc = o_rank.prepare(rank_out, n * 8, mem, false, s)) || (rc = o_bit.prepare(bit_out, n, mem, false, s)))
		return rc;
	HIP_TRY(launch_rank_query(r->d_il, r->n_bits, dv, n, r->mod, values_are_hashes, static_cast<uint64_t*>(o_rank.d),
	                          static_cast<uint8_t*>(o_bit.d), s));
	if ((rc = o_rank.finish(s)) || (rc = o_bit.finish(s)))
		return rc;
	if (mem == BTLBF_HOST)
		HIP_TRY(hipStreamSynchronize(s));
	return BTLBF_OK;
}

// -------------------------------------------------------------------------------------------------
// multi-GPU helpers
// -------------------------------------------------------------------------------------------------
extern "C" int btlbf_positions_seqs(btlbf_filter* f, const char* seq, uint64_t len,
                                    const btlbf_layout* layout, unsigned n_shards, uint64_t* buckets,
                                    uint64_t* tags, uint64_t bucket_cap, uint64_t* bucket_counts,
                                    uint64_t* valid_bits, void* stream)
{
	FilterLock lk__(f);
	int rc = seq_precheck(f, len);
	if (rc)
		return rc;
	if (!buckets || !bucket_counts || n_shards == 0 || n_shards > 64)
		return fail(BTLBF_EINVAL, "bad bucket arguments");
	if (f->size % n_shards)
		return fail(BTLBF_EINVAL, "size not divisible by n_shards");
	DeviceGuard g(f->device);
	hipStream_t s = static_cast<hipStream_t>(stream);
	SeqView v;
	rc = make_view(v, seq, len, layout, BTLBF_DEVICE, s);
	if (rc)
		return rc;
	SeqArgs a = base_args(f, v, len);
	fill_mod(a.mod, f->size, 0, f->size / n_shards); // owner = position / (size/n_shards)
	a.n_shards = n_shards;
	a.buckets = buckets;
	a.tags = tags;
	a.bucket_cap = bucket_cap;
	a.bucket_counts = reinterpret_cast<unsigned long long*>(bucket_counts);
	a.valid_bits = reinterpret_cast<uint8_t*>(valid_bits);
	HIP_TRY(hipMemsetAsync(bucket_counts, 0, (size_t)n_shards * 8, s));
	HIP_TRY(launch_seq_op(OP_POSITIONS, a, s));
	return BTLBF_OK;
}

extern "C" int btlbf_insert_positions(btlbf_filter* f, const uint64_t* local_pos, uint64_t n, void* stream)
{
	FilterLock lk__(f);
	if (!f || (n && !local_pos))
		return fail(BTLBF_EINVAL, "null argument");
	if (f->kind != BTLBF_BLOOM)
		return fail(BTLBF_EINVAL, "position routing is defined for bit filters");
	DeviceGuard g(f->device);
	MATERIALIZE(f, stream);
	HIP_TRY(launch_positions(0, f->d_data, f->mod, local_pos, n, nullptr, static_cast<hipStream_t>(stream)));
	return BTLBF_OK;
}

extern "C" int btlbf_test_positions(btlbf_filter* f, const uint64_t* local_pos, uint64_t n, uint8_t* out,
                                    void* stream)
{
	FilterLock lk__(f);
	if (!f || (n && (!local_pos || !out)))
		return fail(BTLBF_EINVAL, "null argument");
	if (f->kind != BTLBF_BLOOM)
		return fail(BTLBF_EINVAL, "position routing is defined for bit filters");
	DeviceGuard g(f->device);
	MATERIALIZE(f, stream);
	HIP_TRY(launch_positions(1, f->d_data, f->mod, local_pos, n, out, static_cast<hipStream_t>(stream)));
	return BTLBF_OK;
}

extern "C" int btlbf_and_answers(const uint64_t* tags, const uint8_t* answers, uint64_t n, unsigned hash_num,
                                 uint64_t* hit_bits, int device, void* stream)
{
	if (n && (!tags || !answers || !hit_bits))
		return fail(BTLBF_EINVAL, "null argument");
	DeviceGuard g(device);
	HIP_TRY(launch_and_answers(tags, answers, n, hash_num, hit_bits, static_cast<hipStream_t>(stream)));
	return BTLBF_OK;
}

extern "C" int btlbf_count_per_seq(const uint64_t* hit_bits, const uint64_t* valid_bits, uint64_t len,
                                   const btlbf_layout* layout, unsigned kmer_size, uint32_t* hits_out,
                                   uint32_t* valid_out, int mem, int device, void* stream)
{
	if (!layout || (!layout->starts && !layout->read_len))
		return fail(BTLBF_EINVAL, "count_per_seq needs a layout (starts[] or read_len)");
	if (kmer_size == 0 || (len && (!hit_bits || !hits_out)))
		return fail(BTLBF_EINVAL, "null argument");
	int rc = check_layout(layout, len);
	if (rc)
		return rc;
	const uint64_t n_seqs = layout->starts ? layout->n_seqs : len / layout->read_len;
	if (n_seqs == 0)
		return BTLBF_OK;
	if (mem != BTLBF_HOST && mem != BTLBF_DEVICE)
		return fail(BTLBF_EINVAL, "mem must be BTLBF_HOST or BTLBF_DEVICE");
	DeviceGuard g(device);
	hipStream_t s = static_cast<hipStream_t>(stream);
	const size_t bm = bitmap_bytes(len);
	DevBuf d_hit, d_valid, d_starts, d_ho, d_vo;
	const uint64_t *ph = hit_bits, *pv = valid_bits, *ps = layout->starts;
	uint32_t *po = hits_out, *pvo = valid_out;
	if (mem == BTLBF_HOST) {
		HIP_TRY(d_hit.alloc(bm));
		HIP_TRY(hipMemcpyAsync(d_hit.p, hit_bits, bm, hipMemcpyHostToDevice, s));
		ph = d_hit.as<uint64_t>();
		if (valid_bits) {
			HIP_TRY(d_valid.alloc(bm));
			HIP_TRY(hipMemcpyAsync(d_valid.p, valid_bits, bm, hipMemcpyHostToDevice, s));
			pv = d_valid.as<uint64_t>();
		}
		if (layout->starts) {
			HIP_TRY(d_starts.alloc((n_seqs + 1) * 8));
			HIP_TRY(hipMemcpyAsync(d_starts.p, layout->starts, (n_seqs + 1) * 8, hipMemcpyHostToDevice, s));
			ps = d_starts.as<uint64_t>();
		}
		HIP_TRY(d_ho.alloc(n_seqs * 4));
		po = d_ho.as<uint32_t>();
		if (valid_out) {
			HIP_TRY(d_vo.alloc(n_seqs * 4));
			pvo = d_vo.as<uint32_t>();
		}
	}
	HIP_TRY(launch_count_per_seq(ph, pv, len, ps, n_seqs, layout->starts ? 0 : layout->read_len, kmer_size, po, pvo, s));
	if (mem == BTLBF_HOST) {
		HIP_TRY(hipMemcpyAsync(hits_out, po, n_seqs * 4, hipMemcpyDeviceToHost, s));
		if (valid_out)
			HIP_TRY(hipMemcpyAsync(valid_out, pvo, n_seqs * 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
	}
	return BTLBF_OK;
}

extern "C" int btlbf_popcount_bits(const void* dev_buf, uint64_t nbytes, uint64_t* out, int device,
                                   void* stream)
{
	if (!out || (nbytes && !dev_buf))
		return fail(BTLBF_EINVAL, "null argument");
	if (nbytes % 8)
		return fail(BTLBF_EINVAL, "nbytes must be a multiple of 8");
	DeviceGuard g(device);
	hipStream_t s = static_cast<hipStream_t>(stream);
	DevBuf acc;
	HIP_TRY(acc.alloc(8));
	HIP_TRY(hipMemsetAsync(acc.p, 0, 8, s));
	HIP_TRY(launch_popcount(dev_buf, nbytes, 0, 0, acc.as<unsigned long long>(), s));
	unsigned long long v = 0;
	HIP_TRY(hipMemcpyAsync(&v, acc.p, 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	*out = v;
	return BTLBF_OK;
}

// -------------------------------------------------------------------------------------------------
// support
// -------------------------------------------------------------------------------------------------
extern "C" int btlbf_synth_reads(char* dev_out, uint64_t seed, uint64_t first_read, uint64_t n_reads,
                                 unsigned read_len, int device, void* stream)
{
	if (!dev_out || read_len == 0)
		return fail(BTLBF_EINVAL, "bad argument");
	DeviceGuard g(device);
	HIP_TRY(launch_synth(reinterpret_cast<uint8_t*>(dev_out), seed, first_read, n_reads, read_len,
	                     static_cast<hipStream_t>(stream)));
	return BTLBF_OK;
}

extern "C" int btlbf_microbench(btlbf_filter* f, int kind, uint64_t n_access, uint64_t* n_done,
                                double* seconds)
{
	FilterLock lk__(f);
	if (!f || !seconds || !n_done)
		return fail(BTLBF_EINVAL, "null argument");
	{
		const uint64_t per_round = 2048ull * 256 * 8; // launch_microbench geometry
		uint64_t rounds = n_access / per_round;
		*n_done = (rounds ? rounds : 1) * per_round;
	}
	DeviceGuard g(f->device);
	MATERIALIZE(f, nullptr);
	hipEvent_t e0, e1;
	HIP_TRY(hipEventCreate(&e0));
	HIP_TRY(hipEventCreate(&e1));
	HIP_TRY(hipMemset(f->d_scalar, 0, 8));
	HIP_TRY(hipEventRecord(e0, nullptr));
	HIP_TRY(launch_microbench(f->d_data, f->local_bytes, kind, n_access, f->d_scalar, nullptr));
	HIP_TRY(hipEventRecord(e1, nullptr));
	HIP_TRY(hipEventSynchronize(e1));
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	*seconds = ms * 1e-3;
	return BTLBF_OK;
}

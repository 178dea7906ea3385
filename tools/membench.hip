// tools/membench.hip -- exploratory random-access microbenchmarks for MI355X (not part of the product).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/membench tools/membench.hip
// Usage: membench <log2_bytes> <alloc: 0 hipMalloc, 1 uncached, 2 finegrained> [variants...]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
	do {                                                                               \
		hipError_t e = (x);                                                            \
		if (e != hipSuccess) {                                                         \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                     \
			exit(1);                                                                   \
		}                                                                              \
	} while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

enum Variant {
	V_LOAD = 0,       // plain global_load_dword
	V_LOAD_NT = 1,    // nontemporal
	V_LOAD_AGENT = 2, // sc1
	V_LOAD_SYS = 3,   // sc0 sc1
	V_LOAD_X2 = 4,    // 8-byte plain
	V_LOAD_X4 = 5,    // 16-byte plain
	V_OR_AGENT = 6,   // atomicOr no return, agent scope
	V_OR_WG = 7,      // workgroup scope
	V_OR_SYS = 8,     // system scope
	V_OR64 = 9,       // 64-bit atomicOr agent
	V_OR_RET = 10,    // returning atomicOr
	V_STORE = 11,     // plain 4-byte store (not equivalent, for reference)
	V_LOAD_2PERLINE = 12, // two loads in the same 128-B line (both 64-B halves)
	V_LOAD_ASM_SC0SC1NT = 13,
	V_OR_LOADFIRST = 14, // load then atomicOr only if bit clear
};

template <int V>
__global__ __launch_bounds__(256) void k(uint32_t* data, uint64_t n_lines, uint64_t rounds, unsigned long long* sink)
{
	const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
	uint32_t acc = 0;
	for (uint64_t it = 0; it < rounds; ++it) {
		uint64_t idx[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const uint64_t x = mix64((it * 8 + u) * nthreads + gid + 0x1234567ull);
			idx[u] = __umul64hi(x, n_lines) * 16;
		}
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			uint32_t* p = data + idx[u];
			if (V == V_LOAD) acc ^= *p;
			if (V == V_LOAD_NT) acc ^= __builtin_nontemporal_load(p);
			if (V == V_LOAD_AGENT) acc ^= __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (V == V_LOAD_SYS) acc ^= __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			if (V == V_LOAD_X2) { uint2 v = *reinterpret_cast<uint2*>(p); acc ^= v.x ^ v.y; }
			if (V == V_LOAD_X4) { uint4 v = *reinterpret_cast<uint4*>(p); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
			if (V == V_OR_AGENT) __hip_atomic_fetch_or(p, 1u << (idx[u] >> 4 & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (V == V_OR_WG) __hip_atomic_fetch_or(p, 1u << (idx[u] >> 4 & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			if (V == V_OR_SYS) __hip_atomic_fetch_or(p, 1u << (idx[u] >> 4 & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			if (V == V_OR64) __hip_atomic_fetch_or(reinterpret_cast<unsigned long long*>(p), 1ull << (idx[u] >> 4 & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (V == V_OR_RET) acc ^= __hip_atomic_fetch_or(p, 1u << (idx[u] >> 4 & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (V == V_STORE) *p = (uint32_t)idx[u];
			if (V == V_LOAD_2PERLINE) { uint32_t* q = data + (idx[u] & ~31ull); acc ^= q[0] ^ q[16]; }
			if (V == V_LOAD_ASM_SC0SC1NT) {
				uint32_t v;
				asm volatile("global_load_dword %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
				acc ^= v;
			}
			if (V == V_OR_LOADFIRST) {
				uint32_t m = 1u << (idx[u] >> 4 & 31);
				if (!(*p & m)) __hip_atomic_fetch_or(p, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
	}
	if (acc == 0x9e3779b9u)
		atomicAdd(sink, 1ull);
}

template <int V>
static void run(const char* name, uint32_t* d, uint64_t bytes, unsigned long long* sink, uint64_t n_access)
{
	const unsigned blocks = 2048;
	const uint64_t nthreads = (uint64_t)blocks * 256;
	uint64_t rounds = n_access / (nthreads * 8);
	if (!rounds) rounds = 1;
	const uint64_t n = rounds * nthreads * 8;
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, bytes / 64, (uint64_t)8, sink); // warm
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0));
	hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, bytes / 64, rounds, sink);
	CK(hipEventRecord(e1));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	printf("%-22s %8.2f G/s  (%7.1f GB/s @64B, %7.1f GB/s @128B)  %.1f ms\n", name, n / (ms * 1e6), n * 64 / (ms * 1e6),
	       n * 128 / (ms * 1e6), ms);
	fflush(stdout);
}

int main(int argc, char** argv)
{
	const int lg = argc > 1 ? atoi(argv[1]) : 33;
	const int alloc = argc > 2 ? atoi(argv[2]) : 0;
	const uint64_t bytes = 1ull << lg;
	uint32_t* d = nullptr;
	if (alloc == 0) CK(hipMalloc((void**)&d, bytes));
	if (alloc == 1) CK(hipExtMallocWithFlags((void**)&d, bytes, hipDeviceMallocUncached));
	if (alloc == 2) CK(hipExtMallocWithFlags((void**)&d, bytes, hipDeviceMallocFinegrained));
	CK(hipMemset(d, 0, bytes));
	unsigned long long* sink;
	CK(hipMalloc((void**)&sink, 8));
	CK(hipMemset(sink, 0, 8));
	printf("== %llu bytes (2^%d), alloc mode %d\n", (unsigned long long)bytes, lg, alloc);
	const uint64_t N = 1ull << 31;
	std::vector<int> sel;
	for (int i = 3; i < argc; ++i) sel.push_back(atoi(argv[i]));
	auto want = [&](int v) { if (sel.empty()) return true; for (int s : sel) if (s == v) return true; return false; };
	if (want(0)) run<V_LOAD>("load", d, bytes, sink, N);
	if (want(1)) run<V_LOAD_NT>("load nt", d, bytes, sink, N);
	if (want(2)) run<V_LOAD_AGENT>("load agent(sc1)", d, bytes, sink, N);
	if (want(3)) run<V_LOAD_SYS>("load system(sc0sc1)", d, bytes, sink, N);
	if (want(4)) run<V_LOAD_X2>("load 8B", d, bytes, sink, N);
	if (want(5)) run<V_LOAD_X4>("load 16B", d, bytes, sink, N);
	if (want(12)) run<V_LOAD_2PERLINE>("load 2x per 128B line", d, bytes, sink, N);
	if (want(13)) run<V_LOAD_ASM_SC0SC1NT>("load asm sc0 sc1 nt", d, bytes, sink, N / 8);
	if (want(6)) run<V_OR_AGENT>("atomicOr agent", d, bytes, sink, N);
	if (want(7)) run<V_OR_WG>("atomicOr workgroup", d, bytes, sink, N);
	if (want(8)) run<V_OR_SYS>("atomicOr system", d, bytes, sink, N);
	if (want(9)) run<V_OR64>("atomicOr 64-bit", d, bytes, sink, N);
	if (want(10)) run<V_OR_RET>("atomicOr returning", d, bytes, sink, N);
	if (want(14)) run<V_OR_LOADFIRST>("load, or if clear", d, bytes, sink, N);
	if (want(11)) run<V_STORE>("store 4B", d, bytes, sink, N);
	return 0;
}

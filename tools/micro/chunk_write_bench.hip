// tools/micro/chunk_write_bench.hip -- MEASUREMENT tool, not product code: what does pass B's write pattern cost when a
// 32-entry chunk is 128 (4-byte entries), 96 (16 + 8 bits, planar) or 80 bytes (16 + 4 bits)?
// Every workgroup streams its slice of an input array (64 bytes per thread and round, as pass B reads its entries) and
// writes 512 chunks per round to pseudo-random ones of its 1024 regions, each region filled front to back -- pass B's
// traffic without its LDS work.  Prints GB/s read + written and microseconds per 10^6 chunks for the three sizes.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/chunk_write_bench.hip -o tools/micro/chunk_write_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                         \
	do {                                                                              \
		hipError_t e__ = (x);                                                         \
		if (e__ != hipSuccess) {                                                      \
			std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__));            \
			std::exit(1);                                                             \
		}                                                                             \
	} while (0)

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
	x ^= x >> 16;
	x *= 0x7feb352dU;
	x ^= x >> 15;
	x *= 0x846ca68bU;
	return x ^ (x >> 16);
}

// LAYOUT 0: region (reg, wg) at (reg * n_wg + wg) * cap -- the writers' regions of one bin lie together (what passes A
// and B do: the reader of a bin finds its regions in one piece); LAYOUT 1: (wg * 1024 + reg) * cap -- a writer's 1024
// streams lie together in one area of 1024 * cap * CB bytes (fewer pages under each workgroup's stores)
template <int CB, int LAYOUT = 0>
__global__ __launch_bounds__(1024) void chunk_kernel(const uint4* __restrict__ in, uint8_t* __restrict__ out, uint32_t cap,
                                                     uint32_t rounds, uint32_t n_wg)
{
	__shared__ uint32_t written[1024];
	const uint32_t tid = threadIdx.x, grp = tid >> 3, l4 = tid & 7, wg = blockIdx.x;
	written[tid] = 0;
	__syncthreads();
	const uint4* src = in + (uint64_t)wg * rounds * 4096;
	for (uint32_t r = 0; r < rounds; ++r) {
		uint4 v[4];
#pragma unroll
		for (int u = 0; u < 4; ++u)
			v[u] = src[(uint64_t)r * 4096 + u * 1024 + tid];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const uint32_t reg = mix32((r * 4 + u) * 128 + grp + wg * 7919u) & 1023u;
			uint32_t c = 0;
			if (l4 == 0)
				c = atomicAdd(&written[reg], 1u);
			c = __shfl(c, (int)((tid & 63) & ~7u));
			if (c >= cap)
				continue;
			uint8_t* p = out + ((uint64_t)(LAYOUT ? wg * 1024 + reg : reg * n_wg + wg) * cap + c) * CB;
			if (CB == 128) {
				*reinterpret_cast<uint4*>(p + l4 * 16) = v[u];
			} else if (CB == 96) {
				*reinterpret_cast<uint2*>(p + l4 * 8) = make_uint2(v[u].x ^ v[u].z, v[u].y ^ v[u].w);
				*reinterpret_cast<uint32_t*>(p + 64 + l4 * 4) = v[u].x + v[u].y;
			} else {
				*reinterpret_cast<uint2*>(p + l4 * 8) = make_uint2(v[u].x ^ v[u].z, v[u].y ^ v[u].w);
				*reinterpret_cast<uint16_t*>(p + 64 + l4 * 2) = (uint16_t)(v[u].x + v[u].y);
			}
		}
	}
}

template <int CB, int LAYOUT = 0>
static void run(const uint4* in, uint8_t* out, uint32_t cap, uint32_t rounds, uint32_t n_wg)
{
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	float best = 1e30f;
	for (int it = 0; it < 4; ++it) {
		CK(hipEventRecord(a, 0));
		hipLaunchKernelGGL((chunk_kernel<CB, LAYOUT>), dim3(n_wg), dim3(1024), 0, 0, in, out, cap, rounds, n_wg);
		CK(hipEventRecord(b, 0));
		CK(hipEventSynchronize(b));
		float ms;
		CK(hipEventElapsedTime(&ms, a, b));
		if (it && ms < best)
			best = ms;
	}
	const double chunks = (double)n_wg * rounds * 512, rd = chunks * 128, wr = chunks * CB;
	std::printf("layout %d chunk %3d B: %8.3f ms  read %.1f GB + written %.1f GB -> %.0f GB/s  (%.3f us per 1e6 chunks... %.2f ns/chunk-equiv)\n",
	            LAYOUT, CB, best, rd / 1e9, wr / 1e9, (rd + wr) / best / 1e6, best * 1e3 / (chunks / 1e6), best * 1e6 / chunks);
}

int main(int argc, char** argv)
{
	const uint32_t n_wg = 256, rounds = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 700;
	// every region gets rounds * 512 / 1024 chunks on average; capacity 1.25x of that
	const uint32_t cap = rounds / 2 + rounds / 8 + 8;
	const size_t in_bytes = (size_t)n_wg * rounds * 4096 * 16, out_bytes = (size_t)1024 * n_wg * cap * 128;
	uint4* in;
	uint8_t* out;
	CK(hipMalloc(&in, in_bytes));
	CK(hipMalloc(&out, out_bytes));
	CK(hipMemset(in, 1, in_bytes));
	CK(hipMemset(out, 0, out_bytes));
	std::printf("%u workgroups x %u rounds: %.1f GB in, %.1f GB out buffer\n", n_wg, rounds, in_bytes / 1e9, out_bytes / 1e9);
	run<128>(in, out, cap, rounds, n_wg);
	run<96>(in, out, cap, rounds, n_wg);
	run<80>(in, out, cap, rounds, n_wg);
	run<128>(in, out, cap, rounds, n_wg);
	run<128, 1>(in, out, cap, rounds, n_wg);
	run<128, 0>(in, out, cap, rounds, n_wg);
	run<128, 1>(in, out, cap, rounds, n_wg);
	return 0;
}

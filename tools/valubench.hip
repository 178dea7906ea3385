// tools/valubench.hip -- issue rate of the integer VALU ops the hash kernels lean on (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o tools/valubench tools/valubench.hip && ./tools/valubench
// Each kernel runs ITER iterations of 8 independent chains of one op per lane; the result is the
// number of wave-instructions per cycle per SIMD (1.0 = full rate is 0.25 here: a wave64 op takes
// 4 cycles on a 16-lane SIMD), printed relative to v_add_u32.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define ITER 4096

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t s0, uint32_t s1)
{
	uint32_t a[8];
	uint64_t q[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		a[i] = threadIdx.x * 2654435761u + i * 40503u + s0;
		q[i] = ((uint64_t)a[i] << 32) | (a[i] ^ s1);
	}
	for (int it = 0; it < ITER; ++it) {
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			if (OP == 0)
				asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s0));
			if (OP == 1)
				asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s0));
			if (OP == 2)
				asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s0));
			if (OP == 3)
				asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a[i]), "v"(s0) : "vcc");
			if (OP == 4)
				asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q[i]));
			if (OP == 5)
				asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(s0));
			if (OP == 6)
				asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(s0));
			if (OP == 7)
				asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(s0));
			if (OP == 8)
				asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(a[i]));
			if (OP == 9)
				asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
			if (OP == 10)
				asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(q[i]), "v"(q[(i + 1) & 7]) : "vcc");
			if (OP == 11)
				asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(s0));
		}
	}
	uint32_t r = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i)
		r ^= a[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
	out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int OP>
float run(uint32_t* d, int blocks)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	return ms;
}

int main()
{
	const int blocks = 256 * 8; // 8 workgroups of 4 waves per CU
	uint32_t* d;
	hipMalloc(&d, blocks * 256 * 4);
	const char* names[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_lshlrev_b64", "v_xor_b32",
	                       "v_alignbit_b32", "v_mul_u32_u24", "v_bfe_u32", "v_lshl_add_u64", "v_cmp_lt_u64", "v_and_b32_sdwa"};
	float t[12];
	t[0] = run<0>(d, blocks);
	t[1] = run<1>(d, blocks);
	t[2] = run<2>(d, blocks);
	t[3] = run<3>(d, blocks);
	t[4] = run<4>(d, blocks);
	t[5] = run<5>(d, blocks);
	t[6] = run<6>(d, blocks);
	t[7] = run<7>(d, blocks);
	t[8] = run<8>(d, blocks);
	t[9] = run<9>(d, blocks);
	t[10] = run<10>(d, blocks);
	t[11] = run<11>(d, blocks);
	for (int i = 0; i < 12; ++i)
		printf("%-16s %8.3f ms  cost relative to v_add_u32: %.2f   %.3f wave-instructions per ns per SIMD\n", names[i], t[i],
		       t[i] / t[0], (blocks * 4 / 1024.0) * ITER * 8 / (t[i] * 1e6));
	return 0;
}

// csrc/aux_kernels.hip -- everything around the fused sequence kernel:
//   * probes from precomputed hash rows (BloomFilter::insert/contains/insertAndCheck(const uint64_t[]),
//     BloomFilter.hpp:185-262; CountingBloomFilter minCount/insert/incrementAll, :53-64,135-183),
//     in parallel or in strict serial order (one lane) for order-dependent operations
//   * popcount reductions (getPop BloomFilter.hpp:316-323; popCount/filtered_popcount
//     CountingBloomFilter.hpp:217-242)
//   * synthetic read generator and the random-access microbenchmarks
//   * shard-local position insert/test and the origin-side AND of routed answers
#include "seq_core.hpp"

namespace btlbf {


template <bool POW2>
__device__ __forceinline__ void hash_row_op(int op, void* filter, const ModParams& mod, uint32_t h,
                                            uint32_t threshold, const uint64_t* row, uint8_t* out)
{
	uint32_t* words = static_cast<uint32_t*>(filter);
	switch (op) {
	case H_BF_INSERT:
		for (uint32_t i = 0; i < h; ++i) {
			const uint64_t p = reduce_mod<POW2>(row[i], mod) - mod.shard_lo;
			if (p < mod.shard_len)
				bf_set(words, p);
		}
		break;
	case H_BF_CONTAINS: {
		uint32_t all = 1;
		for (uint32_t i = 0; i < h; ++i) {
			const uint64_t p = reduce_mod<POW2>(row[i], mod);
			all &= (bf_word(words, p) >> (p & 31)) & 1u;
		}
		*out = (uint8_t)all;
		break;
	}
	case H_BF_INSERT_CHECK: {
		uint32_t all = 1;
		for (uint32_t i = 0; i < h; ++i)
			all &= bf_set_fetch(words, reduce_mod<POW2>(row[i], mod));
		*out = (uint8_t)all;
		break;
	}
	case H_CBF_INC_ALL:
		for (uint32_t i = 0; i < h; ++i)
			cbf_inc_sat(words, reduce_mod<POW2>(row[i], mod));
		break;
	case H_CBF_MIN:
	case H_CBF_CONTAINS:
	case H_CBF_INSERT_CHECK:
	case H_CBF_INC_MIN: {
		if (op != H_CBF_INC_MIN) {
			uint32_t mn = 0xffu;
			for (uint32_t i = 0; i < h; ++i) {
				const uint32_t v = cbf_read_fresh(words, reduce_mod<POW2>(row[i], mod));
				mn = v < mn ? v : mn;
			}
			*out = op == H_CBF_MIN ? (uint8_t)mn : (uint8_t)(mn >= threshold);
			if (op != H_CBF_INSERT_CHECK)
				break;
		}
		// incrementMin, CountingBloomFilter.hpp:135-162
		for (;;) {
			uint32_t mn = 0xffu;
			for (uint32_t i = 0; i < h; ++i) {
				const uint32_t v = cbf_read_fresh(words, reduce_mod<POW2>(row[i], mod));
				mn = v < mn ? v : mn;
			}
			if (mn == 0xffu)
				break;
			bool done = false;
			for (uint32_t i = 0; i < h; ++i)
				done |= cbf_cas_byte(words, reduce_mod<POW2>(row[i], mod), mn);
			if (done)
				break;
		}
		break;
	}
	default: break;
	}
}

// one lane per hash row; rows whose row_valid byte is 0 are skipped (out = 0)
template <bool POW2>
__global__ __launch_bounds__(256) void hash_rows_kernel(int op, void* filter, ModParams mod, uint32_t h,
                                                        uint32_t threshold, const uint64_t* hashes,
                                                        uint64_t n, const uint8_t* row_valid, uint8_t* out)
{
	for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n;
	     r += (uint64_t)gridDim.x * blockDim.x) {
		uint8_t o = 0;
		if (!row_valid || row_valid[r])
			hash_row_op<POW2>(op, filter, mod, h, threshold, hashes + r * h, &o);
		if (out)
			out[r] = o;
	}
}

// strict row order on one lane: the only reproducible form of the order-dependent operations
// (incrementMin, insertAndCheck on streams with repeats); valid_bits: one bit per row, row_valid: one byte
template <bool POW2>
__global__ void hash_rows_serial_kernel(int op, void* filter, ModParams mod, uint32_t h,
                                        uint32_t threshold, const uint64_t* hashes, uint64_t n,
                                        const uint8_t* valid_bits, const uint8_t* row_valid, uint8_t* out)
{
	if (blockIdx.x != 0 || threadIdx.x != 0)
		return;
	for (uint64_t r = 0; r < n; ++r) {
		if (valid_bits && !((valid_bits[r >> 3] >> (r & 7)) & 1))
			continue;
		uint8_t o = 0;
		if (!row_valid || row_valid[r])
			hash_row_op<POW2>(op, filter, mod, h, threshold, hashes + r * h, &o);
		if (out)
			out[r] = o;
	}
}

hipError_t launch_hash_op(int op, void* filter, const ModParams& mod, uint32_t h, uint32_t threshold,
                          const uint64_t* hashes, uint64_t n, uint8_t* out, int serial, hipStream_t s,
                          const uint8_t* row_valid)
{
	if (n == 0)
		return hipSuccess;
	if (serial) {
		if (mod.pow2)
			hipLaunchKernelGGL(hash_rows_serial_kernel<true>, dim3(1), dim3(64), 0, s, op, filter, mod, h,
			                   threshold, hashes, n, (const uint8_t*)nullptr, row_valid, out);
		else
			hipLaunchKernelGGL(hash_rows_serial_kernel<false>, dim3(1), dim3(64), 0, s, op, filter, mod, h,
			                   threshold, hashes, n, (const uint8_t*)nullptr, row_valid, out);
		return hipGetLastError();
	}
	uint64_t blocks = (n + 255) / 256;
	if (blocks > 2048)
		blocks = 2048;
	if (mod.pow2)
		hipLaunchKernelGGL(hash_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, op, filter, mod,
		                   h, threshold, hashes, n, row_valid, out);
	else
		hipLaunchKernelGGL(hash_rows_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, op, filter, mod,
		                   h, threshold, hashes, n, row_valid, out);
	return hipGetLastError();
}

// ---- raw k-mers: KmerBloomFilter::insert / contains(const char*) (KmerBloomFilter.hpp:47-74) -----------
// The reference hashes a raw k-mer with NTC64(kmerSeq, k) (vendor/nthash.hpp:394-439,460-465): a walk through
// 4-/3-/2-mer tables indexed by convertTab / RCconvertTab codes -- a Horner chain over base codes, with
// rolx(h,4) + swapxbits033(h,4) = srol^4.  Its values differ from the iterator's (NTMC64) in the cases below, and
// this kernel returns what the reference's x86-64 build returns (tests/golden/hash_vectors.json "kmer"):
//   * U/u read as A (convertTab 0, RCconvertTab 3; nthash.hpp:16-86), except in a one-base remainder
//     (k % 4 == 1), which goes through seedTab and reads U as T (:416-417,432-433);
//   * k % 4 == 0: the forward walk ends with rolx(h, 0) and swapxbits033(h, 0); their shifts by 64 execute as
//     shifts by 0 on x86-64, so h becomes h ^ (y | y << 33) with y = h ^ (h >> 33) (:354-356,388-391,404-406);
//   * a byte that is not a base has code 255 and table indices / offsets are uint8_t: they wrap modulo 256;
//   * a 2- or 3-base remainder whose wrapped index lies beyond dimerTab[16] / trimerTab[64] reads outside the
//     reference's tables (undefined): the k-mer is reported invalid and skipped.
// One lane per k-mer; rows[r*h ..] = the h hash values (NTE64, :537-542), valid[r] = 0 for a skipped k-mer.
__device__ __forceinline__ uint32_t kmer_conv(uint32_t c, bool rc)
{
	switch (c) {
	case 'A': case 'a': case 'U': case 'u': return rc ? 3u : 0u;
	case 'C': case 'c': return rc ? 2u : 1u;
	case 'G': case 'g': return rc ? 1u : 2u;
	case 'T': case 't': return rc ? 0u : 3u;
	default: return 255u;
	}
}
__device__ __forceinline__ uint64_t kmer_seed(uint32_t d)
{
	return d == 0 ? kSeedA : d == 1 ? kSeedC : d == 2 ? kSeedG : kSeedT;
}
// seedTab (vendor/nthash.hpp:195-228), for the one-base remainder
__device__ __forceinline__ uint64_t kmer_seed_tab(uint32_t c)
{
	switch (c) {
	case 'A': case 'a': case 4: case 5: return kSeedA;
	case 'C': case 'c': case 7: return kSeedC;
	case 'G': case 'g': case 3: return kSeedG;
	case 'T': case 't': case 'U': case 'u': case 1: return kSeedT;
	default: return 0;
	}
}
// h <- srol^n(h) ^ table_n[idx]: n Horner steps over the base-4 digits of idx, most significant first
__device__ __forceinline__ uint64_t kmer_mer_step(uint64_t hv, uint32_t idx, int n)
{
	for (int j = n - 1; j >= 0; --j)
		hv = srol1(hv) ^ kmer_seed((idx >> (2 * j)) & 3u);
	return hv;
}

__global__ __launch_bounds__(256) void kmer_rows_kernel(const uint8_t* kmers, uint64_t n, uint32_t k, uint32_t h,
                                                        uint64_t kms, uint64_t* rows, uint8_t* valid)
{
	const uint32_t q = k / 4, r = k % 4;
	for (uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n;
	     row += (uint64_t)gridDim.x * blockDim.x) {
		const uint8_t* s = kmers + row * k;
		uint64_t f = 0, v = 0;
		bool ok = true;
		for (uint32_t i = 0; i < q; ++i) { // NTF64(kmerSeq, k), :394-407
			const uint32_t off = (4 * i) & 255u;
			const uint32_t loc = (64 * kmer_conv(s[off], false) + 16 * kmer_conv(s[off + 1], false) +
			                      4 * kmer_conv(s[off + 2], false) + kmer_conv(s[off + 3], false)) & 255u;
			f = kmer_mer_step(f, loc, 4);
		}
		if (r == 0) {
			const uint64_t y = f ^ (f >> 33);
			f ^= y | (y << 33);
		} else if (r == 1) {
			f = srol1(f) ^ kmer_seed_tab(s[k - 1]);
			v = kmer_seed_tab(s[k - 1] & 7u);
		} else {
			uint32_t lf = 0, lr = 0;
			for (uint32_t j = 0; j < r; ++j) {
				lf = 4 * lf + kmer_conv(s[k - r + j], false);
				lr = 4 * lr + kmer_conv(s[k - 1 - j], true);
			}
			lf &= 255u;
			lr &= 255u;
			ok = lf < (1u << (2 * r)) && lr < (1u << (2 * r));
			f = kmer_mer_step(f, lf, (int)r);
			v = kmer_mer_step(0, lr, (int)r);
		}
		for (uint32_t i = 0; i < q; ++i) { // NTR64(kmerSeq, k), :423-439
			const uint32_t off = (4 * (q - i) - 1) & 255u;
			const uint32_t loc = (64 * kmer_conv(s[off], true) + 16 * kmer_conv(s[off - 1], true) +
			                      4 * kmer_conv(s[off - 2], true) + kmer_conv(s[off - 3], true)) & 255u;
			v = kmer_mer_step(v, loc, 4);
		}
		const uint64_t b = v < f ? v : f; // NTC64, :460-465
		rows[row * h] = ok ? b : 0;
		for (uint32_t i = 1; i < h; ++i)
			rows[row * h + i] = ok ? extra_hash(b, kms, i) : 0;
		valid[row] = ok ? 1 : 0;
	}
}

hipError_t launch_kmer_rows(const uint8_t* kmers, uint64_t n, uint32_t k, uint32_t h, uint64_t kms, uint64_t* rows,
                            uint8_t* valid, hipStream_t s)
{
	if (n == 0)
		return hipSuccess;
	uint64_t blocks = (n + 255) / 256;
	if (blocks > 4096)
		blocks = 4096;
	hipLaunchKernelGGL(kmer_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, kmers, n, k, h, kms, rows, valid);
	return hipGetLastError();
}

// serial-order update over the dense hash rows of a sequence buffer (row p = window p; rows of
// unclean windows are skipped through valid_bits).  op is a HashOp.
hipError_t launch_serial_seq_update(const SeqArgs& a, int op, const uint64_t* hashes,
                                    const uint8_t* valid_bits, uint8_t* out, hipStream_t s)
{
	if (a.len == 0)
		return hipSuccess;
	if (a.mod.pow2)
		hipLaunchKernelGGL(hash_rows_serial_kernel<true>, dim3(1), dim3(64), 0, s, op, a.filter, a.mod,
		                   a.hp.h, a.threshold, hashes, a.len, valid_bits, (const uint8_t*)nullptr, out);
	else
		hipLaunchKernelGGL(hash_rows_serial_kernel<false>, dim3(1), dim3(64), 0, s, op, a.filter, a.mod,
		                   a.hp.h, a.threshold, hashes, a.len, valid_bits, (const uint8_t*)nullptr, out);
	return hipGetLastError();
}

// ---- popcount ----------------------------------------------------------------------------------
// mode 0: set bits; mode 1: non-zero bytes; mode 2: bytes >= threshold.  nbytes must be a multiple
// of 8 (bitmaps are whole uint64_t words; filter arrays are zero-padded to 16 bytes).
__global__ __launch_bounds__(256) void popcount_kernel(const uint2* data, uint64_t n_vec, int mode,
                                                       uint32_t threshold, unsigned long long* out)
{
	unsigned long long acc = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec;
	     i += (uint64_t)gridDim.x * blockDim.x) {
		const uint2 v = data[i];
		const uint32_t w[2] = {v.x, v.y};
#pragma unroll
		for (int q = 0; q < 2; ++q) {
			if (mode == 0) {
				acc += __popc(w[q]);
			} else {
#pragma unroll
				for (int b = 0; b < 4; ++b) {
					const uint32_t c = (w[q] >> (8 * b)) & 0xff;
					acc += mode == 1 ? (c != 0) : (c >= threshold);
				}
			}
		}
	}
	// wave reduce, then one atomic per wave
	for (int o = 32; o > 0; o >>= 1)
		acc += __shfl_xor(acc, o, 64);
	if ((threadIdx.x & 63) == 0 && acc)
		atomicAdd(out, acc);
}

hipError_t launch_popcount(const void* data, uint64_t nbytes, int mode, uint32_t threshold,
                           unsigned long long* out, hipStream_t s)
{
	if (nbytes % 8)
		return hipErrorInvalidValue;
	const uint64_t n_vec = nbytes / 8;
	if (n_vec == 0)
		return hipSuccess;
	uint64_t blocks = (n_vec + 255) / 256;
	if (blocks > 4096)
		blocks = 4096;
	hipLaunchKernelGGL(popcount_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
	                   static_cast<const uint2*>(data), n_vec, mode, threshold, out);
	return hipGetLastError();
}

// ---- digest of an array, computed where it lies (btlbf_digest) -------------------------------------------
// out[0] += sum of w_i * m_i (mod 2^64), out[1] ^= xor of mix64(w_i ^ m_i) over the non-zero 64-bit words w_i, where
// i = word0 + (index of the word in this array) is the word's index in the WHOLE filter and m_i = mix64(i + 1) | 1.
// Zero words contribute nothing, so the digests of shards combine (add / xor) to the digest of the whole filter
// whatever the split.  Streaming like popcount_kernel: 16-byte loads, one pair of atomics per wave.
__global__ __launch_bounds__(256) void digest_kernel(const uint4* data, uint64_t n_vec, uint64_t word0,
                                                     unsigned long long* out)
{
	unsigned long long sum = 0, x = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec;
	     i += (uint64_t)gridDim.x * blockDim.x) {
		typedef uint32_t v4u __attribute__((ext_vector_type(4)));
		const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(data) + i);
		const uint64_t w[2] = {((uint64_t)v.y << 32) | v.x, ((uint64_t)v.w << 32) | v.z};
#pragma unroll
		for (int q = 0; q < 2; ++q) {
			if (w[q]) {
				const uint64_t m = mix64(word0 + 2 * i + q + 1) | 1ull;
				sum += w[q] * m;
				x ^= mix64(w[q] ^ m);
			}
		}
	}
	for (int o = 32; o > 0; o >>= 1) {
		sum += __shfl_xor(sum, o, 64);
		x ^= __shfl_xor(x, o, 64);
	}
	if ((threadIdx.x & 63) == 0) {
		if (sum)
			atomicAdd(out, sum);
		if (x)
			atomicXor(out + 1, x);
	}
}

hipError_t launch_digest(const void* data, uint64_t nbytes, uint64_t word0, unsigned long long* out2, hipStream_t s)
{
	if (nbytes % 16)
		return hipErrorInvalidValue;
	const uint64_t n_vec = nbytes / 16;
	if (n_vec == 0)
		return hipSuccess;
	uint64_t blocks = (n_vec + 255) / 256;
	if (blocks > 8192)
		blocks = 8192;
	hipLaunchKernelGGL(digest_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint4*>(data), n_vec,
	                   word0, out2);
	return hipGetLastError();
}

// ---- compare two arrays of the same geometry (btlbf_compare) ---------------------------------------
// counting == 0: positions are bits; out[0] += bits that differ, out[1] += bits set only in a, out[2] += bits
// set only in b.  counting != 0: positions are uint8_t counters; out[0] += counters that differ, out[1] +=
// counters with a > b, out[2] += counters with a < b.
__global__ __launch_bounds__(256) void compare_kernel(const uint2* a, const uint2* b, uint64_t n_vec, int counting,
                                                      unsigned long long* out)
{
	unsigned long long gt = 0, lt = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec;
	     i += (uint64_t)gridDim.x * blockDim.x) {
		const uint2 va = a[i], vb = b[i];
		const uint32_t wa[2] = {va.x, va.y}, wb[2] = {vb.x, vb.y};
#pragma unroll
		for (int q = 0; q < 2; ++q) {
			if (!counting) {
				gt += __popc(wa[q] & ~wb[q]);
				lt += __popc(wb[q] & ~wa[q]);
			} else if (wa[q] != wb[q]) {
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					const uint32_t ca = (wa[q] >> (8 * c)) & 0xff, cb = (wb[q] >> (8 * c)) & 0xff;
					gt += ca > cb;
					lt += ca < cb;
				}
			}
		}
	}
	for (int o = 32; o > 0; o >>= 1) {
		gt += __shfl_xor(gt, o, 64);
		lt += __shfl_xor(lt, o, 64);
	}
	if ((threadIdx.x & 63) == 0 && (gt | lt)) {
		atomicAdd(out, gt + lt);
		atomicAdd(out + 1, gt);
		atomicAdd(out + 2, lt);
	}
}

hipError_t launch_compare(const void* a, const void* b, uint64_t nbytes, int counting, unsigned long long* out3,
                          hipStream_t s)
{
	if (nbytes % 8)
		return hipErrorInvalidValue;
	const uint64_t n_vec = nbytes / 8;
	if (n_vec == 0)
		return hipSuccess;
	uint64_t blocks = (n_vec + 255) / 256;
	if (blocks > 4096)
		blocks = 4096;
	hipLaunchKernelGGL(compare_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint2*>(a),
	                   static_cast<const uint2*>(b), n_vec, counting, out3);
	return hipGetLastError();
}

// ---- synthetic reads --------------------------------------------------------------
// read r, base j = "ACGT"[(w(r*wpr + j/32) >> 2*(j%32)) & 3], w(n) = mix64(seed + (n+1)*golden)
__global__ __launch_bounds__(256) void synth_kernel(uint8_t* out, uint64_t seed, uint64_t first,
                                                    uint64_t total_bytes, uint32_t read_len)
{
	const uint32_t wpr = (read_len + 31) / 32;
	for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c * 16 < total_bytes;
	     c += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t o = c * 16;
		uint64_t r = o / read_len;
		uint32_t j = (uint32_t)(o - r * read_len);
		uint64_t cur_n = ~0ull, w = 0;
		uint32_t ow[4] = {0, 0, 0, 0};
		for (int b = 0; b < 16; ++b) {
			if (o + b >= total_bytes)
				break;
			const uint64_t n = (first + r) * wpr + (j >> 5);
			if (n != cur_n) {
				cur_n = n;
				w = mix64(seed + (n + 1) * 0x9E3779B97F4A7C15ULL);
			}
			const uint32_t code = (uint32_t)(w >> (2 * (j & 31))) & 3;
			const uint32_t ch = (0x54474341u >> (8 * code)) & 0xff; // "ACGT"
			ow[b >> 2] |= ch << (8 * (b & 3));
			if (++j == read_len) {
				j = 0;
				++r;
			}
		}
		if (o + 16 <= total_bytes) {
			*reinterpret_cast<uint4*>(out + o) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
		} else {
			for (int b = 0; o + b < total_bytes; ++b)
				out[o + b] = (uint8_t)(ow[b >> 2] >> (8 * (b & 3)));
		}
	}
}

hipError_t launch_synth(uint8_t* out, uint64_t seed, uint64_t first, uint64_t n, uint32_t read_len,
                        hipStream_t s)
{
	const uint64_t total = n * read_len;
	if (total == 0)
		return hipSuccess;
	uint64_t blocks = ((total + 15) / 16 + 255) / 256;
	if (blocks > 8192)
		blocks = 8192;
	hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, seed, first, total,
	                   read_len);
	return hipGetLastError();
}

// ---- random-access microbenchmarks ---------------------------------------------------------------
// kind 0: 4-byte loads, kind 1: 4-byte atomicOr; uniformly random 64-byte-aligned offsets; 8
// independent accesses per lane per round so the memory system, not latency, is the limit.
__global__ __launch_bounds__(256) void microbench_kernel(uint32_t* data, uint64_t n_lines, int kind,
                                                         uint64_t rounds, unsigned long long* sink)
{
	const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
	uint32_t acc = 0;
	for (uint64_t it = 0; it < rounds; ++it) {
		uint64_t idx[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const uint64_t x = mix64((it * 8 + u) * nthreads + gid + 0x1234567ull);
			idx[u] = (__umul64hi(x, n_lines)) * 16; // uint32 index of a 64-byte line start
		}
		if (kind == 0) {
#pragma unroll
			for (int u = 0; u < 8; ++u)
				acc ^= data[idx[u]];
		} else {
#pragma unroll
			for (int u = 0; u < 8; ++u)
				atomicOr(data + idx[u], 1u << (idx[u] >> 4 & 31));
		}
	}
	if (kind == 0 && acc == 0x9e3779b9u)
		atomicAdd(sink, 1ull); // keeps the loads alive
}

hipError_t launch_microbench(void* data, uint64_t nbytes, int kind, uint64_t n_access,
                             unsigned long long* sink, hipStream_t s)
{
	const uint64_t n_lines = nbytes / 64;
	if (n_lines == 0)
		return hipErrorInvalidValue;
	const unsigned blocks = 256 * 8;
	const uint64_t nthreads = (uint64_t)blocks * 256;
	uint64_t rounds = n_access / (nthreads * 8);
	if (rounds == 0)
		rounds = 1;
	hipLaunchKernelGGL(microbench_kernel, dim3(blocks), dim3(256), 0, s, static_cast<uint32_t*>(data),
	                   n_lines, kind, rounds, sink);
	return hipGetLastError();
}

// ---- shard-local positions (multi-GPU) -------------------------------------------------------------
__global__ __launch_bounds__(256) void positions_kernel(int test, uint32_t* words, int counting,
                                                        const uint64_t* pos, uint64_t n, uint8_t* out)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
	     i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t p = pos[i];
		if (!test)
			bf_set(words, p);
		else
			out[i] = (uint8_t)((bf_word(words, p) >> (p & 31)) & 1u);
	}
	(void)counting;
}

hipError_t launch_positions(int test, void* filter, const ModParams& mod, const uint64_t* pos,
                            uint64_t n, uint8_t* out, hipStream_t s)
{
	(void)mod;
	if (n == 0)
		return hipSuccess;
	uint64_t blocks = (n + 255) / 256;
	if (blocks > 2048)
		blocks = 2048;
	hipLaunchKernelGGL(positions_kernel, dim3((unsigned)blocks), dim3(256), 0, s, test,
	                   static_cast<uint32_t*>(filter), 0, pos, n, out);
	return hipGetLastError();
}

// origin side of a sharded query: a probe that missed clears its window's bit
__global__ __launch_bounds__(256) void and_answers_kernel(const uint64_t* tags, const uint8_t* answers,
                                                          uint64_t n, uint32_t h, uint64_t* hit_bits)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
	     i += (uint64_t)gridDim.x * blockDim.x) {
		if (!answers[i]) {
			const uint64_t p = tags[i] / h;
			atomicAnd(reinterpret_cast<unsigned long long*>(hit_bits) + (p >> 6),
			          ~(1ull << (p & 63)));
		}
	}
}

hipError_t launch_and_answers(const uint64_t* tags, const uint8_t* answers, uint64_t n, uint32_t h,
                              uint64_t* hit_bits, hipStream_t s)
{
	if (n == 0)
		return hipSuccess;
	uint64_t blocks = (n + 255) / 256;
	if (blocks > 2048)
		blocks = 2048;
	hipLaunchKernelGGL(and_answers_kernel, dim3((unsigned)blocks), dim3(256), 0, s, tags, answers, n, h,
	                   hit_bits);
	return hipGetLastError();
}

// ---- per-sequence totals of a per-window bitmap (what read classifiers consume) --------------------
// set bits of `bits` in window range [lo, hi) (bit p of the bitmap = window starting at byte p)
__device__ __forceinline__ uint32_t popc_range(const uint64_t* bits, uint64_t lo, uint64_t hi)
{
	if (hi <= lo)
		return 0;
	uint32_t n = 0;
	const uint64_t w0 = lo >> 6, w1 = (hi - 1) >> 6;
	for (uint64_t w = w0; w <= w1; ++w) {
		uint64_t x = bits[w];
		if (w == w0)
			x &= ~0ull << (lo & 63);
		if (w == w1 && ((hi & 63) != 0))
			x &= ~0ull >> (64 - (hi & 63));
		n += __popcll(x);
	}
	return n;
}

__global__ __launch_bounds__(256) void count_per_seq_kernel(const uint64_t* hit_bits, const uint64_t* valid_bits,
                                                            uint64_t len, const uint64_t* starts, uint64_t n_seqs,
                                                            uint32_t read_len, uint32_t k, uint32_t* hits_out,
                                                            uint32_t* valid_out)
{
	for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_seqs;
	     s += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t b = starts ? starts[s] : s * read_len;
		uint64_t e = starts ? starts[s + 1] : b + read_len;
		if (e > len)
			e = len;
		const uint64_t hi = e >= b + k ? e - k + 1 : b; // windows of this sequence start in [b, hi)
		hits_out[s] = popc_range(hit_bits, b, hi);
		if (valid_out)
			valid_out[s] = valid_bits ? popc_range(valid_bits, b, hi) : (uint32_t)(hi - b);
	}
}

hipError_t launch_count_per_seq(const uint64_t* hit_bits, const uint64_t* valid_bits, uint64_t len,
                                const uint64_t* starts, uint64_t n_seqs, uint32_t read_len, uint32_t k,
                                uint32_t* hits_out, uint32_t* valid_out, hipStream_t s)
{
	if (n_seqs == 0)
		return hipSuccess;
	uint64_t blocks = (n_seqs + 255) / 256;
	if (blocks > 65536)
		blocks = 65536;
	hipLaunchKernelGGL(count_per_seq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, hit_bits, valid_bits, len, starts,
	                   n_seqs, read_len, k, hits_out, valid_out);
	return hipGetLastError();
}

// ---- split query (fixed-length reads): likely-present and likely-absent reads take different paths ----
// contains() over a buffer whose misses come clustered by read (reads foreign to the indexed genome next to
// reads from it -- what a classifier feeds a filter): the partitioned path is fast when nearly every k-mer
// hits, the early-exit gather kernel when nearly none does.  So every read is SAMPLED (three of its windows,
// full contains() each), reads whose clean samples all miss are "cold", the buffer is compacted into a warm
// and a cold buffer, each goes down its own path, and the two bitmaps are merged back into the caller's
// layout.  Every path computes the exact contains(), so the sampling only steers speed.

// one lane per sampled read (read i * stride).  A read is COLD when most of its clean samples miss (two of
// three: a single false positive or a single SNP does not flip the call).  stride == 1: flags bit (r & 63) of
// word r >> 6 = read r is cold.  *n_cold += cold reads among the sampled ones.
// STAGED (stride == 1, 256 reads fit the LDS): the workgroup first copies its 256 consecutive reads into LDS with
// coalesced 16-byte loads and the lanes take their windows from there.  Straight from global memory every 4-byte
// load of a wave touches 64 different lines (one per read), more lines than the L1 holds between a lane's loads:
// the kernel spent as long on fetching 62 bytes per read as on its 8 probes per read.
template <bool STAGED>
__global__ __launch_bounds__(256) void read_sample_kernel(const uint8_t* seq, uint64_t n_reads, uint32_t L, uint32_t stride,
                                                          const HashParams hp, const ModParams mod, const void* filter,
                                                          uint32_t counting, uint32_t threshold, uint32_t probes2,
                                                          unsigned long long* flags, unsigned long long* n_cold)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t staged[]; // STAGED: 256 * L + 32 bytes
	__shared__ uint64_t tab[kNumCodes][2]; // Horner start-up seeds per base code (HashParams::init_tab)
	__shared__ uint8_t lut[256];           // byte -> code << 4 | flags (base_entry)
	if (threadIdx.x < kNumCodes * 2)
		tab[threadIdx.x >> 1][threadIdx.x & 1] = hp.init_tab[threadIdx.x >> 1][threadIdx.x & 1];
	lut[threadIdx.x] = base_entry(threadIdx.x); // 256 threads
	uint32_t staged_off = 0; // LDS byte offset of this lane's read
	if (STAGED) {
		const uint64_t r_wg = (uint64_t)blockIdx.x * 256;
		const uint64_t n_here = n_reads - r_wg < 256 ? n_reads - r_wg : 256; // (the grid covers n_reads exactly)
		const uintptr_t a0 = reinterpret_cast<uintptr_t>(seq) + r_wg * L;
		const uint32_t mis = (uint32_t)(a0 & 15);
		const uint32_t pieces = (uint32_t)((mis + n_here * L + 15) / 16);
		// (an aligned 16-byte piece that holds at least one byte of the buffer lies inside the buffer's last page)
		const uint4* src = reinterpret_cast<const uint4*>(a0 - mis);
		for (uint32_t p = threadIdx.x; p < pieces; p += 256)
			reinterpret_cast<uint4*>(staged)[p] = src[p];
		staged_off = mis + threadIdx.x * L;
	}
	__syncthreads();
	const uint64_t i_s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	// stride > 1: a pseudo-random read of every block of `stride` reads, so that input with a period (every
	// other read foreign, say) is not sampled in step with it
	const uint64_t r = i_s * stride + (stride > 1 ? mix64(i_s) % stride : 0);
	const uint32_t k = hp.k, W = L - k + 1;
	bool cold = false;
	if (r < n_reads) {
		const uint8_t* rd = seq + r * L;
		// one sample = full contains() of the window at `off`.  The hashing and the probing are separate steps so
		// that the probes of the first TWO samples -- 2h independent loads -- are requested together: one memory
		// latency for both (with one sample after the other the kernel ran at half the gather rate of the chip).
		// four bases of the read at byte offset o (STAGED: two aligned LDS words and a byte align)
		auto load4 = [&](uint32_t o) -> uint32_t {
			uint32_t word;
			if (STAGED) {
				const uint32_t a = staged_off + o;
				const uint32_t* l32 = reinterpret_cast<const uint32_t*>(staged) + (a >> 2);
				word = __builtin_amdgcn_alignbyte(l32[1], l32[0], a & 3);
			} else {
				__builtin_memcpy(&word, rd + o, 4);
			}
			return word;
		};
		auto hash_at = [&](uint32_t off, uint64_t& b) -> bool { // -> clean window; b = canonical base hash
			uint64_t fh = 0, rh = 0;
			uint32_t ok = kBaseValid;
			uint32_t i = 0;
			for (; i + 4 <= k; i += 4) { // four bases per (unaligned) load
				const uint32_t word = load4(off + i);
				// ACGT/acgt in one step (the byte permute of seq_stage_convert), anything else through the LUT
				const uint32_t idx = (word >> 1) & 0x03030303u;
				uint32_t e4;
				if ((word & 0xdfdfdfdfu) == __builtin_amdgcn_perm(0u, 0x47544341u, idx))
					e4 = __builtin_amdgcn_perm(0u, 0x23331303u, idx);
				else
					e4 = (uint32_t)lut[word & 0xff] | ((uint32_t)lut[(word >> 8) & 0xff] << 8) |
					     ((uint32_t)lut[(word >> 16) & 0xff] << 16) | ((uint32_t)lut[word >> 24] << 24);
#pragma unroll
				for (int b4 = 0; b4 < 4; ++b4) {
					const uint32_t e = (e4 >> (8 * b4)) & 0xff;
					ok &= e;
					fh = srol1(fh) ^ tab[e >> kCodeShift][0];
					rh = sror1(rh) ^ tab[e >> kCodeShift][1];
				}
			}
			for (; i < k; ++i) {
				const uint32_t e = lut[STAGED ? staged[staged_off + off + i] : rd[off + i]];
				ok &= e;
				fh = srol1(fh) ^ tab[e >> kCodeShift][0];
				rh = sror1(rh) ^ tab[e >> kCodeShift][1];
			}
			b = rh < fh ? rh : fh;
			return (ok & kBaseValid) != 0;
		};
		// what the filter holds at probe j of base hash b (bit filters: the bit; counting: the counter); unclean
		// windows probe position 0 (harmless) so that the loads stay unconditional
		auto fetch = [&](uint64_t b, uint32_t j) -> uint32_t {
			const uint64_t hv = j ? extra_hash(b, hp.kms, j) : b;
			const uint64_t p = mod.pow2 ? (hv & mod.mask) : reduce_mod<false>(hv, mod);
			if (counting)
				return cbf_read_fresh(static_cast<const uint32_t*>(filter), p);
			return (bf_word(static_cast<const uint32_t*>(filter), p) >> (p & 31)) & 1u;
		};
		auto all_hit = [&](uint64_t b) -> bool {
			bool hit = true;
			for (uint32_t j = 0; j < hp.h; ++j)
				hit &= counting ? fetch(b, j) >= threshold : fetch(b, j) != 0;
			return hit;
		};
		// The majority rule matters for the WARM side: a foreign read that slipped into the warm buffer on one
		// false-positive sample would put failures into the partitioned query and cost it a resolve pass over
		// everything; two false positives in one read do not happen.  The third sample is only looked at when the
		// first two disagree (or one of them was unclean).
		uint64_t b0 = 0, b1 = 0;
		const bool c0 = hash_at(0, b0);
		const bool c1 = W / 2 != 0 ? hash_at(W / 2, b1) : false;
		bool h0 = true, h1 = true;
		// (probes2 <= h probes of the second sample: what a present read costs is its probes -- all of them hit, so all
		// of them are loaded --, and the second sample is only there so that a foreign read does not pass on one
		// false-positive window; how many probes that takes depends on how many foreign reads there are: the caller
		// works it out from its estimate)
		for (uint32_t j = 0; j < hp.h; ++j) { // both samples' probes, interleaved: up to 2h loads before the first use
			const uint32_t v0 = fetch(c0 ? b0 : 0, j);
			h0 &= counting ? v0 >= threshold : v0 != 0;
			if (j < probes2) {
				const uint32_t v1 = fetch(c1 ? b1 : 0, j);
				h1 &= counting ? v1 >= threshold : v1 != 0;
			}
		}
		const uint32_t s0 = c0 ? (h0 ? 2u : 1u) : 0u, s1 = c1 ? (h1 ? 2u : 1u) : 0u; // 0 = unclean, 1 = miss, 2 = hit
		uint32_t clean = (s0 != 0) + (s1 != 0), misses = (s0 == 1) + (s1 == 1);
		if (!(s0 == s1 && s0 != 0) && W - 1 != W / 2) {
			uint64_t b2 = 0;
			const bool c2 = hash_at(W - 1, b2);
			const uint32_t s2 = c2 ? (all_hit(b2) ? 2u : 1u) : 0u;
			clean += s2 != 0;
			misses += s2 == 1;
		}
		cold = clean > 0 && 2 * misses > clean;
	}
	const unsigned long long m = __ballot(cold);
	if ((threadIdx.x & 63) == 0) {
		if (stride == 1 && r < n_reads)
			flags[r >> 6] = m;
		if (m)
			atomicAdd(n_cold, (unsigned long long)__popcll(m));
	}
}

// prefix[w] = cold reads before flag word w.  Three tiny passes: per-chunk sums (1024 words a chunk), a serial
// exclusive scan of the chunk sums by one thread (at most ~16 K chunks for 10^9 reads), then each word's
// prefix from its chunk's base
__global__ __launch_bounds__(1024) void flag_chunk_sum_kernel(const unsigned long long* flags, uint64_t n_words,
                                                             uint32_t* chunk_sum)
{
	__shared__ uint32_t acc;
	if (threadIdx.x == 0)
		acc = 0;
	__syncthreads();
	const uint64_t w = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
	uint32_t c = w < n_words ? (uint32_t)__popcll(flags[w]) : 0;
	c = wave_sum(c);
	if ((threadIdx.x & 63) == 0 && c)
		atomicAdd(&acc, c);
	__syncthreads();
	if (threadIdx.x == 0)
		chunk_sum[blockIdx.x] = acc;
}
__global__ void flag_chunk_scan_kernel(uint32_t* chunk_sum, uint32_t n_chunks)
{
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		uint32_t run = 0;
		for (uint32_t i = 0; i < n_chunks; ++i) {
			const uint32_t v = chunk_sum[i];
			chunk_sum[i] = run;
			run += v;
		}
	}
}
__global__ __launch_bounds__(1024) void flag_prefix_kernel(const unsigned long long* flags, uint64_t n_words,
                                                          const uint32_t* chunk_base, uint32_t* prefix)
{
	__shared__ uint32_t wave_tot[16];
	const uint64_t w = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
	const uint32_t c = w < n_words ? (uint32_t)__popcll(flags[w]) : 0;
	const uint32_t incl = wave_scan_incl(c);
	if ((threadIdx.x & 63) == 63)
		wave_tot[threadIdx.x >> 6] = incl;
	__syncthreads();
	uint32_t base = chunk_base[blockIdx.x];
	for (uint32_t v = 0; v < (threadIdx.x >> 6); ++v)
		base += wave_tot[v];
	if (w < n_words)
		prefix[w] = base + incl - c;
}

__device__ __forceinline__ uint32_t cold_rank(const unsigned long long* flags, const uint32_t* prefix, uint64_t r,
                                              bool* cold)
{
	const unsigned long long m = flags[r >> 6];
	*cold = (m >> (r & 63)) & 1ull;
	return prefix[r >> 6] + (uint32_t)__popcll(m & ((1ull << (r & 63)) - 1));
}

// One thread per 16-byte piece of a READ (ceil(L / 16) pieces per read, the last one short when L is no multiple of
// 16): consecutive lanes copy consecutive pieces of a read to cold_buf[rank * L + ...] or warm_buf[(r - rank) * L + ...],
// so that the loads and the stores of a wave are runs of (unaligned) 16-byte accesses, no piece straddles two reads
// and only the tail of a read is copied in smaller steps.  Measured on one box, 10^8 reads of 150 bytes, 1 % cold:
// 16-byte pieces of the SOURCE (a tenth of them straddle a read boundary and go byte by byte, and their whole wave
// with them) 20.4 ms; 4-byte pieces of a read 17.5 ms; the same with four pieces per thread in flight 23.8 ms.
__global__ __launch_bounds__(256) void compact_reads_kernel(const uint8_t* seq, uint64_t n_reads, uint32_t L, uint32_t ppr,
                                                            uint32_t ppr_inv, const unsigned long long* flags,
                                                            const uint32_t* prefix, uint8_t* warm_buf, uint8_t* cold_buf)
{
	// this workgroup's reads: 256 threads take 256 / ppr whole reads per trip (reads of more than 4096 bytes: one read
	// per trip, the threads loop over its pieces)
	const bool wide = ppr > 256;
	const uint32_t rpt = wide ? 1u : 256u / ppr; // reads per trip
	const uint32_t lr = wide ? 0u : __umulhi(threadIdx.x, ppr_inv), i0 = threadIdx.x - lr * ppr; // read of the trip, piece in it
	for (uint64_t r0 = (uint64_t)blockIdx.x * rpt; r0 < n_reads; r0 += (uint64_t)gridDim.x * rpt) {
		const uint64_t r = r0 + lr;
		if (lr >= rpt || r >= n_reads)
			continue;
		bool cold;
		const uint32_t rk = cold_rank(flags, prefix, r, &cold);
		const uint8_t* src0 = seq + r * L;
		uint8_t* dst0 = cold ? cold_buf + (uint64_t)rk * L : warm_buf + (r - rk) * L;
		for (uint32_t i = i0; i < ppr; i += wide ? 256u : ppr) {
			const uint8_t* src = src0 + 16 * i;
			uint8_t* dst = dst0 + 16 * i;
			if (16 * i + 16 <= L) {
				uint32_t w[4];
				__builtin_memcpy(w, src, 16); // unaligned loads / stores
				__builtin_memcpy(dst, w, 16);
			} else {
				const uint32_t n = L - 16 * i;
				uint32_t j = 0;
				for (; j + 4 <= n; j += 4) {
					uint32_t w;
					__builtin_memcpy(&w, src + j, 4);
					__builtin_memcpy(dst + j, &w, 4);
				}
				for (; j < n; ++j)
					dst[j] = src[j];
			}
		}
	}
}

// bits [pos, pos + n) of a bitmap (n <= 64), little-endian bit order
__device__ __forceinline__ uint64_t bits_at(const uint64_t* bm, uint64_t pos, uint32_t n)
{
	const uint64_t w = pos >> 6;
	const uint32_t sh = (uint32_t)(pos & 63);
	uint64_t v = bm[w] >> sh;
	if (sh + n > 64)
		v |= bm[w + 1] << (64 - sh);
	return n == 64 ? v : v & ((1ull << n) - 1);
}

// one thread per word of the caller's bitmaps: the bits of the reads that overlap it, fetched from the warm /
// cold bitmaps (over the compacted buffers, which are padded by one word)
__global__ __launch_bounds__(256) void merge_split_bitmaps_kernel(uint64_t n_words, uint64_t len, uint32_t L,
                                                                  const unsigned long long* flags, const uint32_t* prefix,
                                                                  const uint64_t* warm_hit, const uint64_t* cold_hit,
                                                                  const uint64_t* warm_valid, const uint64_t* cold_valid,
                                                                  uint64_t* hit_out, uint64_t* valid_out)
{
	const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= n_words)
		return;
	uint64_t b0 = w * 64, b1 = b0 + 64;
	if (b1 > len)
		b1 = len;
	uint64_t hit = 0, valid = 0;
	for (uint64_t r = b0 / L; b0 < b1; ++r) {
		const uint64_t r_end = (r + 1) * L;
		const uint64_t e = r_end < b1 ? r_end : b1;
		const uint32_t n = (uint32_t)(e - b0);
		bool cold;
		const uint32_t rk = cold_rank(flags, prefix, r, &cold);
		const uint64_t src = (cold ? (uint64_t)rk : r - rk) * L + (b0 - r * L);
		const uint32_t sh = (uint32_t)(b0 - w * 64);
		if (hit_out)
			hit |= bits_at(cold ? cold_hit : warm_hit, src, n) << sh;
		if (valid_out)
			valid |= bits_at(cold ? cold_valid : warm_valid, src, n) << sh;
		b0 = e;
	}
	if (hit_out)
		hit_out[w] = hit;
	if (valid_out)
		valid_out[w] = valid;
}

// Only the cold reads are gathered (the split query's masked form: the warm reads stay where they are): one thread per
// read looks at its flag; a cold one copies its read to cold_buf[rank * L ..] in 16-byte steps and notes where it came
// from.  (compact_reads_kernel, one thread per 16-byte piece of EVERY read, spends its time finding out that 99 % of
// the pieces are not wanted.)
__global__ __launch_bounds__(256) void gather_cold_reads_kernel(const uint8_t* seq, uint64_t n_reads, uint32_t L,
                                                                const unsigned long long* flags, const uint32_t* prefix,
                                                                uint8_t* cold_buf, uint32_t* cold_index)
{
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads)
		return;
	bool cold;
	const uint32_t rk = cold_rank(flags, prefix, r, &cold);
	if (!cold)
		return;
	cold_index[rk] = (uint32_t)r;
	const uint8_t* src = seq + r * L;
	uint8_t* dst = cold_buf + (uint64_t)rk * L;
	uint32_t i = 0;
	for (; i + 16 <= L; i += 16) {
		uint32_t w[4];
		__builtin_memcpy(w, src + i, 16); // unaligned loads / stores
		__builtin_memcpy(dst + i, w, 16);
	}
	for (; i + 4 <= L; i += 4) {
		uint32_t w;
		__builtin_memcpy(&w, src + i, 4);
		__builtin_memcpy(dst + i, &w, 4);
	}
	for (; i < L; ++i)
		dst[i] = src[i];
}

hipError_t launch_gather_cold_reads(const uint8_t* seq, uint64_t n_reads, uint32_t L, const uint64_t* flags,
                                    const uint32_t* prefix, uint8_t* cold_buf, uint32_t* cold_index, hipStream_t s)
{
	if (n_reads == 0)
		return hipSuccess;
	hipLaunchKernelGGL(gather_cold_reads_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, s, seq, n_reads, L,
	                   reinterpret_cast<const unsigned long long*>(flags), prefix, cold_buf, cold_index);
	return hipGetLastError();
}

// The cold reads' answers (bitmaps over the gathered cold buffer) go back to where the reads lie in the caller's
// buffer: one thread per cold read ORs its L bits into the caller's bitmaps, whose bits for that read pass A left zero.
__global__ __launch_bounds__(256) void merge_cold_bitmaps_kernel(uint64_t n_cold, uint32_t L, const uint32_t* cold_index,
                                                                 const uint64_t* cold_hit, const uint64_t* cold_valid,
                                                                 unsigned long long* hit_out, unsigned long long* valid_out)
{
	const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= n_cold)
		return;
	uint64_t src = c * L, dst = (uint64_t)cold_index[c] * L;
	for (uint32_t left = L; left;) {
		const uint32_t sh = (uint32_t)(dst & 63), n = left < 64 - sh ? left : 64 - sh;
		if (hit_out) {
			const uint64_t v = bits_at(cold_hit, src, n);
			if (v)
				atomicOr(&hit_out[dst >> 6], (unsigned long long)(v << sh));
		}
		if (valid_out) {
			const uint64_t v = bits_at(cold_valid, src, n);
			if (v)
				atomicOr(&valid_out[dst >> 6], (unsigned long long)(v << sh));
		}
		src += n;
		dst += n;
		left -= n;
	}
}

hipError_t launch_merge_cold_bitmaps(uint64_t n_cold, uint32_t L, const uint32_t* cold_index, const uint64_t* cold_hit,
                                     const uint64_t* cold_valid, uint64_t* hit_out, uint64_t* valid_out, hipStream_t s)
{
	if (n_cold == 0)
		return hipSuccess;
	hipLaunchKernelGGL(merge_cold_bitmaps_kernel, dim3((unsigned)((n_cold + 255) / 256)), dim3(256), 0, s, n_cold, L,
	                   cold_index, cold_hit, cold_valid, reinterpret_cast<unsigned long long*>(hit_out),
	                   reinterpret_cast<unsigned long long*>(valid_out));
	return hipGetLastError();
}

// stride > 1: only every stride-th read is sampled and only *n_cold is produced (a cheap estimate)
hipError_t launch_read_sample(const uint8_t* seq, uint64_t n_reads, uint32_t L, uint32_t stride, const HashParams& hp,
                              const ModParams& mod, const void* filter, int counting, uint32_t threshold, uint64_t* flags,
                              uint64_t* n_cold, hipStream_t s, uint32_t probes2)
{
	if (n_reads == 0 || stride == 0)
		return hipSuccess;
	if (probes2 == 0 || probes2 > hp.h)
		probes2 = hp.h;
	const uint64_t n_s = (n_reads + stride - 1) / stride;
	const size_t staged_bytes = (size_t)256 * L + 32;
	if (stride == 1 && staged_bytes <= 40 * 1024) { // every read looked at, and 256 of them fit the LDS (L <= 159)
		hipLaunchKernelGGL(read_sample_kernel<true>, dim3((unsigned)((n_s + 255) / 256)), dim3(256), staged_bytes, s, seq,
		                   n_reads, L, stride, hp, mod, filter, (uint32_t)counting, threshold, probes2,
		                   reinterpret_cast<unsigned long long*>(flags), reinterpret_cast<unsigned long long*>(n_cold));
		return hipGetLastError();
	}
	hipLaunchKernelGGL(read_sample_kernel<false>, dim3((unsigned)((n_s + 255) / 256)), dim3(256), 0, s, seq, n_reads, L, stride,
	                   hp, mod, filter, (uint32_t)counting, threshold, probes2, reinterpret_cast<unsigned long long*>(flags),
	                   reinterpret_cast<unsigned long long*>(n_cold));
	return hipGetLastError();
}

// prefix must hold ceil(n_reads / 64) uint32 + one uint32 per 1024 of those (scratch behind it)
hipError_t launch_flag_prefix(const uint64_t* flags, uint64_t n_reads, uint32_t* prefix, hipStream_t s)
{
	const uint64_t n_words = (n_reads + 63) / 64;
	if (n_words == 0)
		return hipSuccess;
	const uint32_t n_chunks = (uint32_t)((n_words + 1023) / 1024);
	uint32_t* chunk = prefix + n_words;
	const unsigned long long* fl = reinterpret_cast<const unsigned long long*>(flags);
	hipLaunchKernelGGL(flag_chunk_sum_kernel, dim3(n_chunks), dim3(1024), 0, s, fl, n_words, chunk);
	hipLaunchKernelGGL(flag_chunk_scan_kernel, dim3(1), dim3(64), 0, s, chunk, n_chunks);
	hipLaunchKernelGGL(flag_prefix_kernel, dim3(n_chunks), dim3(1024), 0, s, fl, n_words, chunk, prefix);
	return hipGetLastError();
}

hipError_t launch_compact_reads(const uint8_t* seq, uint64_t n_reads, uint32_t L, const uint64_t* flags,
                                const uint32_t* prefix, uint8_t* warm_buf, uint8_t* cold_buf, hipStream_t s)
{
	if (n_reads == 0)
		return hipSuccess;
	const uint32_t ppr = (L + 15) / 16; // 16-byte pieces per read
	const uint32_t ppr_inv = (uint32_t)(0xffffffffull / ppr) + 1; // floor(x / ppr) = umulhi(x, ppr_inv) for x < 2^16
	const uint32_t rpt = ppr > 256 ? 1u : 256u / ppr; // reads per workgroup and trip
	const uint64_t trips = (n_reads + rpt - 1) / rpt;
	const unsigned blocks = (unsigned)(trips < 65536 ? trips : 65536);
	hipLaunchKernelGGL(compact_reads_kernel, dim3(blocks), dim3(256), 0, s, seq, n_reads, L, ppr, ppr_inv,
	                   reinterpret_cast<const unsigned long long*>(flags), prefix, warm_buf, cold_buf);
	return hipGetLastError();
}

hipError_t launch_merge_split_bitmaps(uint64_t len, uint32_t L, const uint64_t* flags, const uint32_t* prefix,
                                      const uint64_t* warm_hit, const uint64_t* cold_hit, const uint64_t* warm_valid,
                                      const uint64_t* cold_valid, uint64_t* hit_out, uint64_t* valid_out, hipStream_t s)
{
	const uint64_t n_words = (len + 63) / 64;
	if (n_words == 0)
		return hipSuccess;
	hipLaunchKernelGGL(merge_split_bitmaps_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s, n_words, len, L,
	                   reinterpret_cast<const unsigned long long*>(flags), prefix, warm_hit, cold_hit, warm_valid, cold_valid,
	                   hit_out, valid_out);
	return hipGetLastError();
}

// ---- rank structure over a bit filter (miBF stage 2: sdsl::bit_vector_il<512> + rank_support_il<1>) --------
// The reference's multi-index Bloom filter turns a set bit's position into an index of its ID array with
// rank(pos) = number of set bits before pos (MIBloomFilter.hpp:527 getRankPos, :324,391,443,461,488,509,522),
// over sdsl's INTERLEAVED bit vector (MIBloomFilter.hpp:44,133,144,801-803): the bit vector cut into blocks of
// 512 bits, each preceded by one 64-bit word holding the number of set bits before the block, so that one
// rank query touches one 72-byte record.  sdsl-lite is not vendored with the reference (SURVEY.md section 2,
// row 9); this is its published layout: record b = { ones before bit 512*b, data words 8*b .. 8*b+7 }.
static constexpr uint32_t kRankBlockWords = 8; // 512 bits

// pass 1: ones per 512-bit block.  pass 3 (after the scan of `ones`): write the interleaved records.
__global__ __launch_bounds__(256) void rank_block_ones_kernel(const uint64_t* bits, uint64_t n_words, uint64_t n_blocks,
                                                              uint64_t* ones)
{
	for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
		uint32_t c = 0;
#pragma unroll
		for (uint32_t j = 0; j < kRankBlockWords; ++j) {
			const uint64_t w = b * kRankBlockWords + j;
			c += w < n_words ? (uint32_t)__popcll(bits[w]) : 0;
		}
		ones[b] = c;
	}
}

// exclusive prefix sum of n values in place, by ONE workgroup: every thread owns a contiguous slice
__global__ __launch_bounds__(1024) void scan_u64_kernel(uint64_t* v, uint64_t n, uint64_t* total)
{
	__shared__ uint64_t part[1024];
	const uint64_t per = (n + 1023) / 1024;
	const uint64_t lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
	uint64_t sum = 0;
	for (uint64_t i = lo; i < hi; ++i)
		sum += v[i];
	part[threadIdx.x] = sum;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint64_t run = 0;
		for (int t = 0; t < 1024; ++t) {
			const uint64_t x = part[t];
			part[t] = run;
			run += x;
		}
		if (total)
			*total = run;
	}
	__syncthreads();
	uint64_t run = part[threadIdx.x];
	for (uint64_t i = lo; i < hi; ++i) {
		const uint64_t x = v[i];
		v[i] = run;
		run += x;
	}
}

// two-level: chunk sums (4096 blocks a chunk) -> scan of the chunk sums -> per-chunk exclusive scan
__global__ __launch_bounds__(256) void rank_chunk_sum_kernel(const uint64_t* ones, uint64_t n_blocks, uint64_t* chunk)
{
	__shared__ unsigned long long acc;
	if (threadIdx.x == 0)
		acc = 0;
	__syncthreads();
	unsigned long long c = 0;
	for (uint32_t j = threadIdx.x; j < 4096; j += 256) {
		const uint64_t b = (uint64_t)blockIdx.x * 4096 + j;
		c += b < n_blocks ? ones[b] : 0;
	}
	for (int o = 32; o > 0; o >>= 1)
		c += __shfl_xor(c, o, 64);
	if ((threadIdx.x & 63) == 0)
		atomicAdd(&acc, c);
	__syncthreads();
	if (threadIdx.x == 0)
		chunk[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void rank_write_kernel(const uint64_t* bits, uint64_t n_words, uint64_t n_blocks,
                                                         const uint64_t* ones, const uint64_t* chunk_base, uint64_t* il)
{
	// one workgroup per chunk of 4096 blocks; thread t scans blocks [16t, 16t+16) after a workgroup scan
	__shared__ uint64_t part[256];
	const uint64_t b0 = (uint64_t)blockIdx.x * 4096 + threadIdx.x * 16;
	uint64_t mine[16], sum = 0;
#pragma unroll
	for (int j = 0; j < 16; ++j) {
		mine[j] = b0 + j < n_blocks ? ones[b0 + j] : 0;
		sum += mine[j];
	}
	part[threadIdx.x] = sum;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint64_t run = chunk_base[blockIdx.x];
		for (int t = 0; t < 256; ++t) {
			const uint64_t x = part[t];
			part[t] = run;
			run += x;
		}
	}
	__syncthreads();
	uint64_t run = part[threadIdx.x];
	for (int j = 0; j < 16; ++j) {
		const uint64_t b = b0 + j;
		if (b >= n_blocks)
			break;
		uint64_t* rec = il + b * (kRankBlockWords + 1);
		rec[0] = run;
		for (uint32_t q = 0; q < kRankBlockWords; ++q) {
			const uint64_t w = b * kRankBlockWords + q;
			rec[1 + q] = w < n_words ? bits[w] : 0;
		}
		run += mine[j];
	}
}

// rank_out[i] = set bits before position pos(i); bit_out[i] = the bit itself.  pos(i) = in[i] % size when `size`
// is given (in = hash values: getRankPos, MIBloomFilter.hpp:527), else in[i]
__global__ __launch_bounds__(256) void rank_query_kernel(const uint64_t* il, uint64_t n_bits, const uint64_t* in, uint64_t n,
                                                         const ModParams mod, int reduce, uint64_t* rank_out,
                                                         uint8_t* bit_out)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		uint64_t p = in[i];
		if (reduce)
			p = mod.pow2 ? (p & mod.mask) : reduce_mod<false>(p, mod);
		uint64_t r = 0;
		uint32_t bit = 0;
		if (p < n_bits) {
			const uint64_t* rec = il + (p >> 9) * (kRankBlockWords + 1);
			const uint32_t wq = (uint32_t)(p >> 6) & 7, sh = (uint32_t)p & 63;
			r = rec[0];
			for (uint32_t q = 0; q < wq; ++q)
				r += __popcll(rec[1 + q]);
			const uint64_t w = rec[1 + wq];
			r += __popcll(w & ((1ull << sh) - 1));
			bit = (uint32_t)(w >> sh) & 1u;
		}
		if (rank_out)
			rank_out[i] = r;
		if (bit_out)
			bit_out[i] = (uint8_t)bit;
	}
}

// scratch: n_blocks + ceil(n_blocks / 4096) + 1 uint64_t
hipError_t launch_rank_build(const uint64_t* bits, uint64_t n_bits, uint64_t* il, uint64_t* scratch, uint64_t* total_dev,
                             hipStream_t s)
{
	const uint64_t n_words = (n_bits + 63) / 64, n_blocks = (n_bits + 511) / 512;
	if (n_blocks == 0)
		return hipSuccess;
	const uint64_t n_chunks = (n_blocks + 4095) / 4096;
	if (n_chunks > (1ull << 31))
		return hipErrorInvalidValue;
	uint64_t* ones = scratch;
	uint64_t* chunk = scratch + n_blocks;
	uint64_t g = (n_blocks + 255) / 256;
	if (g > 65536)
		g = 65536;
	hipLaunchKernelGGL(rank_block_ones_kernel, dim3((unsigned)g), dim3(256), 0, s, bits, n_words, n_blocks, ones);
	hipLaunchKernelGGL(rank_chunk_sum_kernel, dim3((unsigned)n_chunks), dim3(256), 0, s, ones, n_blocks, chunk);
	hipLaunchKernelGGL(scan_u64_kernel, dim3(1), dim3(1024), 0, s, chunk, n_chunks, total_dev);
	hipLaunchKernelGGL(rank_write_kernel, dim3((unsigned)n_chunks), dim3(256), 0, s, bits, n_words, n_blocks, ones, chunk, il);
	return hipGetLastError();
}

hipError_t launch_rank_query(const uint64_t* il, uint64_t n_bits, const uint64_t* in, uint64_t n, const ModParams& mod,
                             int reduce, uint64_t* rank_out, uint8_t* bit_out, hipStream_t s)
{
	if (n == 0)
		return hipSuccess;
	uint64_t g = (n + 255) / 256;
	if (g > 65536)
		g = 65536;
	hipLaunchKernelGGL(rank_query_kernel, dim3((unsigned)g), dim3(256), 0, s, il, n_bits, in, n, mod, reduce, rank_out, bit_out);
	return hipGetLastError();
}

} // namespace btlbf

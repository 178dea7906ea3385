// csrc/device_utils.hpp -- device-side arithmetic shared by the kernels (gfx950 only).
#pragma once
#include "internal.hpp"

namespace btlbf {

// One step of ntHash's split rotate: the low 33 bits and the high 31 bits of x each rotate left
// by one (reference semantics: rol1 + swapbits033, vendor/nthash.hpp:350-352,377-380).
// Written on the 32-bit halves (hi:lo): lo' = lo<<1 | hi[0];  hi' = hi[30:1]<<2 | hi[31]<<1 | lo[31].
__device__ __forceinline__ uint64_t srol1(uint64_t x)
{
	const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
	const uint32_t nlo = (lo << 1) | (hi & 1u);
	const uint32_t t = __builtin_amdgcn_alignbit(hi, lo, 31); // hi<<1 | lo>>31: all but bit 1 are right
	const uint32_t nhi = (t & ~2u) | ((hi >> 30) & 2u);       // v_bfi
	return ((uint64_t)nhi << 32) | nlo;
}

// Inverse step (ror1 + swapbits3263, vendor/nthash.hpp:361-363,383-386).
// lo' = lo>>1 | hi[0]<<31;  hi' = hi[31:2]<<1 | hi[1]<<31 | lo[0]
__device__ __forceinline__ uint64_t sror1(uint64_t x)
{
	const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
	const uint32_t nlo = __builtin_amdgcn_alignbit(hi, lo, 1); // lo>>1 | hi<<31
	const uint32_t r = __builtin_amdgcn_alignbit(hi, hi, 1);   // hi rotated right: r[30:1] = hi[31:2], r[0] = hi[1]
	const uint32_t ends = (r << 31) | (lo & 1u);               // v_and + v_lshl_or
	const uint32_t nhi = (r & 0x7ffffffeu) | (ends & ~0x7ffffffeu); // v_bfi
	return ((uint64_t)nhi << 32) | nlo;
}

// srol applied s times (used once per workgroup to build the positional seed table)
__device__ __forceinline__ uint64_t srol_n(uint64_t x, uint32_t s)
{
	uint64_t lo = x & 0x1FFFFFFFFULL, hi = x >> 33;
	const uint32_t a = s % 33, b = s % 31;
	if (a)
		lo = ((lo << a) | (lo >> (33 - a))) & 0x1FFFFFFFFULL;
	if (b)
		hi = ((hi << b) | (hi >> (31 - b))) & 0x7FFFFFFFULL;
	return (hi << 33) | lo;
}

// i-th extra hash of a canonical base hash (NTE64, vendor/nthash.hpp:537-542)
__device__ __forceinline__ uint64_t extra_hash(uint64_t b, uint64_t kms, uint32_t i)
{
	uint64_t t = b * ((uint64_t)i ^ kms);
	return t ^ (t >> kMultiShift);
}

// hash % size without a 64-bit divide: q = mulhi(hash, floor(2^64/size)) is the true quotient
// or one less, so a single conditional subtract finishes (size >= 2).
// A filter of 2^32 positions or more -- every filter the partitioned pipeline is worth it for -- has a magic
// below 2^32, and so a quotient below 2^32: mulhi(hash, magic) is then two 32-bit multiplies instead of four, and
// q * size a 32 x 64-bit product (same q, same remainder).  A 400-Gbit filter of no power-of-two size ran pass A at
// 75 ms per 6x10^9 k-mers with the generic form, against 42 for 2^38 / 2^39 bits.
__device__ __forceinline__ bool mod_small_magic(const ModParams& m) { return m.magic32 != 0; }
__device__ __forceinline__ uint64_t reduce_mod_small(uint64_t hash, const ModParams& m)
{
	// Nine instructions: mul_hi + mad (q), mad + mul_lo + add (hash - q * size as hash + q * (2^64 - size)), sub, subb,
	// a 32-bit compare and two selects.  Written on `magic32` and `neg_size`: from `(uint32_t)m.magic` inside a branch on
	// `m.magic >> 32 == 0` the optimiser concludes zext(trunc(magic)) == magic and multiplies by both halves of it (a
	// run-time zero) -- sixteen instructions, five of them 64-bit multiply-adds
	const uint32_t mg = m.magic32, lo = (uint32_t)hash, hi = (uint32_t)(hash >> 32);
	const uint32_t q = (uint32_t)(((uint64_t)hi * mg + __umulhi(lo, mg)) >> 32);
	const uint64_t r0 = hash + (uint64_t)q * (uint32_t)m.neg_size;
	const uint64_t r = (uint64_t)((uint32_t)(r0 >> 32) + q * (uint32_t)(m.neg_size >> 32)) << 32 | (uint32_t)r0;
	const uint64_t r2 = r + m.neg_size; // r - size; r < 2 * size and size < 2^63 (fill_mod): negative iff r < size
	return (int32_t)(r2 >> 32) < 0 ? r : r2;
}
__device__ __forceinline__ uint64_t reduce_mod_big(uint64_t hash, const ModParams& m)
{
	const uint64_t q = __umul64hi(hash, m.magic);
	const uint64_t r = hash - q * m.size;
	return r >= m.size ? r - m.size : r;
}
// (callers with many probes in a row test mod_small_magic() once and call the form they need: with the test inside,
// the compiler keeps it -- a scalar branch and a handful of register moves per probe)
template <bool POW2>
__device__ __forceinline__ uint64_t reduce_mod(uint64_t hash, const ModParams& m)
{
	if (POW2)
		return hash & m.mask;
	return mod_small_magic(m) ? reduce_mod_small(hash, m) : reduce_mod_big(hash, m);
}

// ---- bit filter probes: bit (p%8) of byte p/8 == bit (p%32) of little-endian word p/32 --------
__device__ __forceinline__ void bf_set(uint32_t* words, uint64_t lp)
{
	atomicOr(words + (lp >> 5), 1u << (lp & 31)); // result unused -> no-return global_atomic_or
}
__device__ __forceinline__ uint32_t bf_set_fetch(uint32_t* words, uint64_t lp)
{
	return (atomicOr(words + (lp >> 5), 1u << (lp & 31)) >> (lp & 31)) & 1u;
}
// probe loads are non-temporal: a random word of a multi-GiB array is never reused, and on MI355X
// `nt` gathers run ~12 % above plain ones (54 vs 48 G requests/s, tools/membench.hip)
__device__ __forceinline__ uint32_t bf_word(const uint32_t* words, uint64_t lp)
{
	return __builtin_nontemporal_load(words + (lp >> 5));
}

// ---- uint8_t counters packed four to a word; HBM has no byte atomics, so CAS the word ----------
__device__ __forceinline__ uint32_t agent_load(const uint32_t* p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t cbf_read_fresh(const uint32_t* words, uint64_t lp)
{
	return (agent_load(words + (lp >> 2)) >> ((lp & 3) * 8)) & 0xffu;
}

// saturating +1 (incrementAll's per-counter step, CountingBloomFilter.hpp:171-181)
__device__ __forceinline__ void cbf_inc_sat(uint32_t* words, uint64_t lp)
{
	uint32_t* w = words + (lp >> 2);
	const uint32_t sh = (lp & 3) * 8;
	uint32_t old = agent_load(w);
	for (;;) {
		if (((old >> sh) & 0xffu) == 0xffu)
			return; // "newVal < currentVal" overflow test: leave 255 alone
		uint32_t prev = atomicCAS(w, old, old + (1u << sh));
		if (prev == old)
			return;
		old = prev;
	}
}

// byte CAS expect -> expect+1; returns true when this call made the change
// (the __sync_bool_compare_and_swap of CountingBloomFilter.hpp:152)
__device__ __forceinline__ bool cbf_cas_byte(uint32_t* words, uint64_t lp, uint32_t expect)
{
	uint32_t* w = words + (lp >> 2);
	const uint32_t sh = (lp & 3) * 8;
	uint32_t old = agent_load(w);
	for (;;) {
		if (((old >> sh) & 0xffu) != expect)
			return false;
		uint32_t prev = atomicCAS(w, old, old + (1u << sh));
		if (prev == old)
			return true;
		old = prev;
	}
}

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

// inclusive prefix sum over the 64 lanes of a wave: Hillis-Steele inside each row of 16 lanes with DPP
// row shifts, then the row totals are broadcast down (row_bcast:15 into rows 1 and 3, row_bcast:31
// into rows 2 and 3) -- six VALU adds, no LDS traffic
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v)
{
	v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); // row_shr:1
	v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); // row_shr:2
	v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); // row_shr:4
	v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); // row_shr:8
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
	v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
	return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
	for (int o = 32; o > 0; o >>= 1)
		v += __shfl_xor(v, o, 64);
	return v;
}

} // namespace btlbf

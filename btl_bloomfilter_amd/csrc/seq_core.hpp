// csrc/seq_core.hpp -- the shared front end of every sequence kernel: tile staging and the
// per-lane rolling ntHash.  Used by seq_kernels.hip (fused hash + probe) and partition_kernels.hip
// (fused hash + radix partition).
//
// A tile is NT*kW consecutive window start offsets (NT = workgroup size, kW = 8 windows per lane).
//   seq_setup_tables : one-time LDS tables (byte -> base code LUT, Horner/roll seed tables, spaced-seed tables)
//   seq_stage_tile   : coalesced loads of the tile's bytes -> LDS byte per base
//                      (bits 6:4 code, bit 0 valid, bit 1 valid and not the first base of a sequence)
//   seq_lane_windows : start-up over the lane's first window (four bases per LDS word, one 16-byte
//                      table entry per base), then kW-1 O(1) rolls (one pair-table entry per roll);
//                      calls f(w, clean, hashes) for each of the lane's windows
// Reference semantics reproduced (not its code): ntHashIterator init/next
// (vendor/ntHashIterator.hpp:59-86), NTMC64/NTMSM64 (vendor/nthash.hpp:581-590,667-692,820-878).
#pragma once
#include "device_utils.hpp"

#ifndef BTLBF_SPACED_GRP
#define BTLBF_SPACED_GRP 2
#endif
#ifndef BTLBF_SPACED_CHK
#define BTLBF_SPACED_CHK 2
#endif

namespace btlbf {

static constexpr int kW = 8; // consecutive windows per lane (default)

struct __attribute__((aligned(16))) U64x2 {
	uint64_t x, y;
};

// static LDS shared by the sequence kernels
struct SeqShared {
	U64x2 pair_tab[kNumCodes * kNumCodes]; // [outgoing code * 8 + incoming code]: the roll's XOR terms for both strands
	U64x2 init_tab[kNumCodes];
	uint8_t lut[256];
	unsigned long long cnt_valid;
	unsigned long long cnt_hit;
	uint64_t start_lo; // ragged layout: first starts[] index that can fall inside the tile
};

// ASCII byte -> code | valid (what vendor/nthash.hpp:195-228 accepts: ACGTU acgtu and 1 3 4 5 7)
__device__ __forceinline__ uint8_t base_entry(uint32_t c)
{
	constexpr uint32_t ok = kBaseValid | kBaseGood;
	switch (c) {
	case 'A': case 'a': return (0 << kCodeShift) | ok;
	case 'C': case 'c': return (1 << kCodeShift) | ok;
	case 'G': case 'g': return (2 << kCodeShift) | ok;
	case 'T': case 't': case 'U': case 'u': return (3 << kCodeShift) | ok;
	case 4: case 5: return (4 << kCodeShift) | ok; // raw bytes: forward A C G T, reverse seed = forward seed
	case 7: return (5 << kCodeShift) | ok;
	case 3: return (6 << kCodeShift) | ok;
	case 1: return (7 << kCodeShift) | ok;
	default: return 0;
	}
}

// bytes of dynamic LDS the tile needs (tile bytes + up to 15 bytes of misalignment, rounded to 16)
__host__ __device__ inline uint32_t seq_tile_cap(uint32_t tile_windows, uint32_t k)
{
	return ((tile_windows + k - 1 + 15 + 15) / 16) * 16;
}
// bytes of dynamic LDS for the tables that follow the tile: the positional seed table
// pos_tab[i*8+c] = {srol^(k-1-i)(fwd[c]), srol^i(rev[c])} (when hp.use_pos_tab) and the spaced seeds'
// don't-care index list
// (+ one row of zeros behind the k rows: spaced seeds pad their offset list with the "offset" k, whose terms change
// nothing)
__host__ __device__ inline uint32_t seq_pos_tab_bytes(const HashParams& hp)
{
	return hp.use_pos_tab ? (hp.k + 1) * kNumCodes * 16 : 0;
}
__host__ __device__ inline uint32_t seq_dc_list_bytes(const HashParams& hp)
{
	return hp.n_seeds ? ((hp.dc_off[hp.n_seeds] * 2 + 15) / 16) * 16 : 0;
}
// ... followed by the union list of distinct don't-care offsets (HashParams::dcu), 4 bytes each
// ... and, where a launcher found room (HashParams::n_pair_rows), the two-base rows of the list's pairs, 256 bytes each
__host__ __device__ inline uint32_t seq_union_list_bytes(const HashParams& hp)
{
	return hp.n_seeds ? ((hp.n_dcu * 4 + 15) / 16) * 16 : 0;
}
__host__ __device__ inline uint32_t seq_spaced_bytes(const HashParams& hp)
{
	return seq_pos_tab_bytes(hp) + seq_dc_list_bytes(hp) + seq_union_list_bytes(hp) + (hp.n_seeds ? hp.n_pair_rows * 256 : 0);
}

// one-time table setup; callers __syncthreads() before first use (seq_stage_tile does)
template <int NT, bool SPACED>
__device__ __forceinline__ void seq_setup_tables(SeqShared& sh, const HashParams& hp, uint8_t* spaced_lds)
{
	const uint32_t tid = threadIdx.x;
	for (uint32_t i = tid; i < 256; i += NT)
		sh.lut[i] = base_entry(i);
	if (tid < kNumCodes)
		sh.init_tab[tid] = U64x2{hp.init_tab[tid][0], hp.init_tab[tid][1]};
	for (uint32_t i = tid; i < kNumCodes * kNumCodes; i += NT) {
		const uint32_t co = i / kNumCodes, ci = i % kNumCodes;
		sh.pair_tab[i] = U64x2{hp.in_tab[ci][0] ^ hp.out_tab[co][0], hp.in_tab[ci][1] ^ hp.out_tab[co][1]};
	}
	if (tid == 0) {
		sh.cnt_valid = 0;
		sh.cnt_hit = 0;
	}
	if (hp.use_pos_tab) {
		// positional seed table, built once per workgroup from the four seeds:
		// fwd[c] = init_tab[c][0], rev[c] = out_tab[c][1] (internal.hpp)
		const uint32_t k = hp.k;
		U64x2* pt = reinterpret_cast<U64x2*>(spaced_lds);
		auto term = [&](uint32_t pos, uint32_t c) {
			return U64x2{srol_n(hp.init_tab[c][0], k - 1 - pos), srol_n(hp.out_tab[c][1], pos)};
		};
		if (SPACED) {
			for (uint32_t i = tid; i < (k + 1) * kNumCodes; i += NT) {
				const uint32_t pos = i / kNumCodes, c = i % kNumCodes;
				pt[i] = pos < k ? term(pos, c) : U64x2{0, 0};
			}
		} else {
			// plain ntHash: the table is indexed by PAIRS of bases -- row j (256 bytes) holds, for the 16 pairs of A C G T,
			// the sum of the terms of positions 2j and 2j+1; an odd k ends with one single-base row of all 8 codes.  Same
			// bytes as the single-base table, half the lookups and XORs in a lane's start-up (seq_lane_range); windows
			// with one of the raw-byte codes 4..7 take the Horner form there
			const uint32_t pairs = k / 2;
			for (uint32_t i = tid; i < pairs * 16; i += NT) {
				const uint32_t j = i / 16, c1 = (i / 4) % 4, c2 = i % 4;
				const U64x2 a = term(2 * j, c1), b = term(2 * j + 1, c2);
				pt[i] = U64x2{a.x ^ b.x, a.y ^ b.y};
			}
			if ((k & 1) && tid < kNumCodes)
				pt[pairs * 16 + tid] = term(k - 1, tid);
		}
	}
	if (SPACED) {
		uint16_t* di = reinterpret_cast<uint16_t*>(spaced_lds + seq_pos_tab_bytes(hp));
		const uint32_t ndc = hp.dc_off[hp.n_seeds];
		for (uint32_t i = tid; i < ndc; i += NT)
			di[i] = hp.dc_idx[i];
		uint32_t* du = reinterpret_cast<uint32_t*>(spaced_lds + seq_pos_tab_bytes(hp) + seq_dc_list_bytes(hp));
		for (uint32_t i = tid; i < hp.n_dcu; i += NT)
			du[i] = hp.dcu[i];
		// two-base rows of the list's pairs: row p, entry (c1, c2) = term(offset 1, c1) ^ term(offset 2, c2) for A C G T;
		// the filler "offset" k contributes nothing
		U64x2* pr = reinterpret_cast<U64x2*>(spaced_lds + seq_pos_tab_bytes(hp) + seq_dc_list_bytes(hp) + seq_union_list_bytes(hp));
		for (uint32_t i = tid; i < hp.n_pair_rows * 16; i += NT) {
			const uint32_t p = i / 16, c1 = (i / 4) % 4, c2 = i % 4;
			const uint32_t o1 = hp.dcu[hp.n_dcu_all + 2 * p] & 0xffffu, o2 = hp.dcu[hp.n_dcu_all + 2 * p + 1] & 0xffffu;
			U64x2 t{0, 0};
			if (o1 < hp.k) {
				t.x ^= srol_n(hp.init_tab[c1][0], hp.k - 1 - o1);
				t.y ^= srol_n(hp.out_tab[c1][1], o1);
			}
			if (o2 < hp.k) {
				t.x ^= srol_n(hp.init_tab[c2][0], hp.k - 1 - o2);
				t.y ^= srol_n(hp.out_tab[c2][1], o2);
			}
			pr[i] = t;
		}
	}
}

// Stage the tile that starts at byte offset g0.  Ends with a __syncthreads(); begins with one so
// that the previous tile has been fully consumed.  tile_off = g0 % read_len (uniform layout only).
// Returns the misalignment `mis`: LDS index of window w's first base is w + mis.
// x mod L for x < 2^24, L >= 1, without a divide: inv = floor((2^32-1)/L)
__device__ __forceinline__ uint32_t small_mod(uint32_t x, uint32_t L, uint32_t inv)
{
	uint32_t r = x - __umulhi(x, inv) * L;
	while (r >= L)
		r -= L;
	return r;
}

// The raw words of a tile this thread converts (its first KW/4+1 words; more only for large k).
template <int KW = kW>
struct StageRaw {
	uint32_t w[KW / 4 + 1];
};

// Request this thread's words of the tile that starts at byte offset g0 (zero past the data).  Kept
// apart from the conversion so that a kernel can have the NEXT tile's loads in flight while it works
// on the current one (pass A of the partitioned pipeline).
// span = bytes a tile stages (0: the NT*KW window starts plus the k-1 bases behind the last one)
// tid_in >= 0: the staging is shared by NT threads numbered tid_in = 0 .. NT-1 that need not be the whole workgroup
// (pass A's overlapped schedule: half of the waves stage while the others flush); KW then only sets the words
// per thread (KW / 4 + 1) and `span` must be given.
template <int NT, int KW = kW>
__device__ __forceinline__ void seq_stage_load(StageRaw<KW>& raw, const uint8_t* seq, uint64_t len, uint32_t k,
                                               uint64_t g0, uint32_t span = 0, int32_t tid_in = -1)
{
	constexpr uint32_t kTileW = NT * KW;
	if (span == 0)
		span = kTileW + k - 1;
	const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(seq + g0) & 3);
	uint64_t need = len > g0 ? len - g0 : 0;
	if (need > (uint64_t)span)
		need = span;
	const uint32_t n_words = need ? (mis + (uint32_t)need + 3) / 4 : 0;
	uint32_t tid = tid_in < 0 ? threadIdx.x : (uint32_t)tid_in; // laundered: see seq_stage_convert
	asm volatile("" : "+v"(tid));
#pragma unroll
	for (int a = 0; a < KW / 4 + 1; ++a) {
		const uint32_t j = tid + (uint32_t)a * NT;
		raw.w[a] = 0;
		if (j < n_words)
			// (a streaming load here changes nothing: 38.8-38.9 ms against 38.9-39.0 per pass-A launch)
			raw.w[a] = *reinterpret_cast<const uint32_t*>(seq + g0 - mis + 4ull * j);
	}
}

// Ragged layout: clear the "good" flag of every staged base that opens a sequence inside the tile (g0 .. g0+need).
// All NT threads of the workgroup; contains a barrier (the staged words must all be there before flags are cleared
// in them); the caller provides the barrier between these atomics and the first read of the tile.
template <int NT>
__device__ __forceinline__ void seq_stage_mark_starts(uint8_t* tile, SeqShared& sh, const LayoutParams& lay, uint64_t g0,
                                                      uint64_t need, uint32_t mis)
{
	const uint64_t* starts = lay.starts;
	const uint32_t tid = threadIdx.x;
	// first index s with starts[s] > g0: every boundary strictly inside (g0, g0+need) matters
	if (tid == 0) {
		uint64_t lo = 0, hi = lay.n_seqs + 1;
		while (lo < hi) {
			uint64_t mid = (lo + hi) >> 1;
			if (starts[mid] > g0)
				hi = mid;
			else
				lo = mid + 1;
		}
		sh.start_lo = lo;
	}
	__syncthreads();
	const uint64_t s_lo = sh.start_lo;
	for (uint64_t s = s_lo + tid; s <= lay.n_seqs; s += NT) {
		const uint64_t p = starts[s];
		if (p >= g0 + need)
			break;
		const uint32_t li = (uint32_t)(p - g0) + mis;
		atomicAnd(reinterpret_cast<uint32_t*>(tile) + (li >> 2), ~(kBaseGood << (8 * (li & 3))));
	}
}

// Stage the tile that starts at byte offset g0: every thread converts aligned 4-byte words of the
// read buffer (coalesced loads) through the LUT and writes them to LDS, so all waves share the work.
// Ends with a __syncthreads(); begins with one so that the previous tile has been fully consumed.
// tile_off = g0 % read_len (uniform layout only).  Returns the misalignment `mis` (0..3): the LDS
// index of window w's first base is w + mis.
// LEAD_BARRIER = false: the caller guarantees that nobody still reads the previous tile (pass A of
// the partitioned pipeline: its last partition round ends with a barrier after the last tile read).
// TRAIL_BARRIER = false: the caller provides the barrier between these writes and the first read.
// NOSYNC: no workgroup barrier anywhere inside (the stagers are only some of the workgroup's waves, tid_in as in
// seq_stage_load); the ragged layout's sequence starts are then left to the caller (seq_stage_mark_starts).
template <int NT, int KW = kW, bool LEAD_BARRIER = true, bool TRAIL_BARRIER = true, bool NOSYNC = false>
__device__ __forceinline__ uint32_t seq_stage_convert(const StageRaw<KW>& pre, uint8_t* tile, uint32_t tile_cap,
                                                      SeqShared& sh, const uint8_t* seq, uint64_t len,
                                                      const LayoutParams& lay, uint32_t k, uint64_t g0,
                                                      uint32_t tile_off, uint32_t span = 0, int32_t tid_in = -1,
                                                      const uint32_t* start_bits = nullptr)
{
	static_assert(!NOSYNC || (!LEAD_BARRIER && !TRAIL_BARRIER), "no barriers at all when only some waves stage");
	constexpr uint32_t kTileW = NT * KW;
	if (span == 0)
		span = kTileW + k - 1;
	// the thread index is laundered so that the per-word LDS addresses below are recomputed every tile: left
	// to itself the compiler hoists them out of the caller's tile loop into registers it then has to spill,
	// and in pass A a scratch reload sits behind the previous flush's stores (vector memory retires in order)
	uint32_t tid = tid_in < 0 ? threadIdx.x : (uint32_t)tid_in;
	asm volatile("" : "+v"(tid));
	const uint32_t L = lay.read_len;
	const uint64_t* starts = lay.starts;
	const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(seq + g0) & 3);
	uint64_t need = len > g0 ? len - g0 : 0;
	if (need > (uint64_t)span)
		need = span;
	const uint32_t n_words = (mis + (uint32_t)need + 3) / 4;
	const bool uniform = !starts && L;
	const uint32_t inv = uniform ? 0xffffffffu / L : 0;
	if (LEAD_BARRIER)
		__syncthreads(); // previous tile fully consumed (and tables written, first time round)

	// words past the data are zero-filled so no stale flags survive
	auto convert = [&](uint32_t j, uint32_t raw) {
		// position of this word's first byte relative to g0 (negative for the misaligned head); bytes
		// outside [0, need) are zeroed with one mask per word
		const int32_t rel0 = (int32_t)(4 * j) - (int32_t)mis;
		const int32_t lo = rel0 < 0 ? (-rel0 < 4 ? -rel0 : 4) : 0; // leading bytes before the data
		const int64_t left = (int64_t)need - rel0;                  // bytes of data from this word on
		const int32_t hi = left <= 0 ? 0 : (left < 4 ? (int32_t)left : 4);
		uint32_t keep = hi >= 4 ? 0xffffffffu : ((1u << (8 * hi)) - 1);
		if (lo)
			keep &= lo >= 4 ? 0u : ~((1u << (8 * lo)) - 1);
		// fast path, four bases at once: (c >> 1) & 3 sends A C T G (either case) to 0 1 2 3; a byte permute
		// maps that index back to the letter it must have come from -- the word holds nothing but ACGT/acgt
		// iff the case-folded word equals that -- and a second one to the staged byte (code << 4 | valid |
		// good).  Anything else (N, U, the raw bytes 1 3 4 5 7, padding) takes the LUT.
		const uint32_t idx = (raw >> 1) & 0x03030303u;
		uint32_t o;
		if ((raw & 0xdfdfdfdfu) == __builtin_amdgcn_perm(0u, 0x47544341u, idx))
			o = __builtin_amdgcn_perm(0u, 0x23331303u, idx);
		else
			o = (uint32_t)sh.lut[raw & 0xff] | ((uint32_t)sh.lut[(raw >> 8) & 0xff] << 8) |
			    ((uint32_t)sh.lut[(raw >> 16) & 0xff] << 16) | ((uint32_t)sh.lut[raw >> 24] << 24);
		o &= keep;
		if (uniform) {
			// offset of the word's first byte inside its read; a read of >= 4 bases starts at most once
			// inside a word, at byte (L - r) % L
			const uint32_t r = small_mod(tile_off + 4 * j + 4 * L - mis, L, inv);
			if (L >= 4) {
				const uint32_t bstar = r ? L - r : 0;
				if (bstar < 4)
					o &= ~(kBaseGood << (8 * bstar));
			} else {
				uint32_t rr = r;
#pragma unroll
				for (int b = 0; b < 4; ++b) {
					if (rr == 0)
						o &= ~(kBaseGood << (8 * b));
					rr = (rr + 1 == L) ? 0 : rr + 1;
				}
			}
		}
		if (start_bits) {
			// ragged layout, sequence starts of this tile marked beforehand in an LDS bitmap (one bit per staged byte;
			// pass A's overlapped schedule): the four bits of this word clear the "good" flags of its bytes
			const uint32_t b4 = (start_bits[j >> 3] >> ((j & 7u) * 4u)) & 0xfu;
			o &= ~(((b4 * 0x00204081u) & 0x01010101u) * kBaseGood);
		}
		reinterpret_cast<uint32_t*>(tile)[j] = o;
	};
	// the thread's first words were all requested before any is converted (one memory latency per
	// tile, not one per word -- or none, when the caller asked for them a tile ahead)
	constexpr int kAhead = KW / 4 + 1;
#pragma unroll
	for (int a = 0; a < kAhead; ++a) {
		const uint32_t j = tid + (uint32_t)a * NT;
		if (j < tile_cap / 4)
			convert(j, pre.w[a]);
	}
	for (uint32_t j = tid + kAhead * NT; j < tile_cap / 4; j += NT) { // large k only
		uint32_t w = 0;
		if (j < n_words)
			w = *reinterpret_cast<const uint32_t*>(seq + g0 - mis + 4ull * j);
		convert(j, w);
	}
	if (starts && !NOSYNC)
		seq_stage_mark_starts<NT>(tile, sh, lay, g0, need, mis);
	if (TRAIL_BARRIER)
		__syncthreads();
	return mis;
}

// Read-grid staging (pass A, uniform layout, tiles of whole reads starting at g0): source byte x of the tile
// (read x / L, base x % L) goes to LDS byte (x / L) * lpad + x % L, so every read starts on an 8-byte boundary
// and the pad bytes between reads (zeroed once by the kernel, never written) end every window that would run
// over a read's end.  Same staged byte as seq_stage_convert (code << 4 | valid | good, `good` cleared on a
// read's first base).  The caller provides the barriers.
// skip_bits != nullptr: bit (skip_bit0 + q) of that array (32-bit words, LDS or global) says that read q of the tile
// is to be LEFT OUT (the split query's cold reads, which another kernel answers): its bytes are staged as zeros, so
// none of its windows is clean -- no entries, no bits in the bitmaps, nothing counted.
template <int NT, int KW>
__device__ __forceinline__ void seq_stage_convert_grid(const StageRaw<KW>& pre, uint8_t* tile, SeqShared& sh,
                                                       const uint8_t* seq, uint64_t len, uint32_t L, uint32_t lpad,
                                                       uint32_t tile_bytes, uint64_t g0, int32_t tid_in = -1,
                                                       const uint32_t* skip_bits = nullptr, uint32_t skip_bit0 = 0)
{
	auto skipped = [&](uint32_t q) -> bool {
		const uint32_t b = skip_bit0 + q;
		return (skip_bits[b >> 5] >> (b & 31)) & 1u;
	};
	uint32_t tid = tid_in < 0 ? threadIdx.x : (uint32_t)tid_in; // laundered: see seq_stage_convert
	asm volatile("" : "+v"(tid));
	const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(seq + g0) & 3);
	uint64_t need64 = len > g0 ? len - g0 : 0;
	const uint32_t need = need64 > tile_bytes ? tile_bytes : (uint32_t)need64;
	const uint32_t inv = 0xffffffffu / L;
	const bool halves = mis == 0 && (L & 1) == 0; // every aligned byte pair of the source lies inside one read
#pragma unroll
	for (int a = 0; a < KW / 4 + 1; ++a) {
		const uint32_t j = tid + (uint32_t)a * NT;
		const int32_t rel0 = (int32_t)(4 * j) - (int32_t)mis; // tile offset of the word's first byte
		if (rel0 >= (int32_t)tile_bytes)
			continue;
		const uint32_t raw = pre.w[a];
		const uint32_t idx = (raw >> 1) & 0x03030303u;
		uint32_t o;
		if ((raw & 0xdfdfdfdfu) == __builtin_amdgcn_perm(0u, 0x47544341u, idx))
			o = __builtin_amdgcn_perm(0u, 0x23331303u, idx);
		else
			o = (uint32_t)sh.lut[raw & 0xff] | ((uint32_t)sh.lut[(raw >> 8) & 0xff] << 8) |
			    ((uint32_t)sh.lut[(raw >> 16) & 0xff] << 16) | ((uint32_t)sh.lut[raw >> 24] << 24);
		// bytes past the data are staged as zeros (no stale flags), bytes before the tile are not staged
		const int32_t left = (int32_t)need - rel0;
		if (left < 4)
			o &= left <= 0 ? 0u : (1u << (8 * left)) - 1;
		const uint32_t x0 = rel0 < 0 ? 0u : (uint32_t)rel0;
		uint32_t q = __umulhi(x0, inv), p = x0 - q * L;
		while (p >= L) {
			p -= L;
			++q;
		}
		uint32_t dst = q * lpad + p;
		if (halves) {
			uint32_t h0 = o & 0xffffu, h1 = o >> 16;
			if (p == 0)
				h0 &= ~kBaseGood;
			if (skip_bits && skipped(q))
				h0 = 0;
			*reinterpret_cast<uint16_t*>(tile + dst) = (uint16_t)h0;
			p += 2;
			dst += 2;
			if (p >= L) { // == L: the second pair opens the next read
				p = 0;
				dst += lpad - L;
				h1 &= ~kBaseGood;
				++q;
			}
			if (skip_bits && skipped(q))
				h1 = 0;
			*reinterpret_cast<uint16_t*>(tile + dst) = (uint16_t)h1;
		} else {
#pragma unroll
			for (int b = 0; b < 4; ++b) {
				const int32_t x = rel0 + b;
				if (x < 0 || x >= (int32_t)tile_bytes)
					continue;
				uint32_t e = (o >> (8 * b)) & 0xffu;
				if (p == 0)
					e &= ~kBaseGood;
				if (skip_bits && skipped(q))
					e = 0;
				tile[dst] = (uint8_t)e;
				++p;
				++dst;
				if (p == L) {
					p = 0;
					dst += lpad - L;
					++q;
				}
			}
		}
	}
}

// load + convert in one go (the direct kernels)
template <int NT, int KW = kW, bool LEAD_BARRIER = true, bool TRAIL_BARRIER = true>
__device__ __forceinline__ uint32_t seq_stage_tile(uint8_t* tile, uint32_t tile_cap, SeqShared& sh,
                                                   const uint8_t* seq, uint64_t len, const LayoutParams& lay,
                                                   uint32_t k, uint64_t g0, uint32_t tile_off)
{
	StageRaw<KW> raw;
	seq_stage_load<NT, KW>(raw, seq, len, k, g0);
	return seq_stage_convert<NT, KW, LEAD_BARRIER, TRAIL_BARRIER>(raw, tile, tile_cap, sh, seq, len, lay, k, g0,
	                                                              tile_off);
}

// hash values of one window.  Plain ntHash: hash i is recomputed from the canonical value where it
// is needed (a multiply and a shift), so nothing is indexed dynamically; spaced seeds keep an array.
template <bool SPACED>
struct WinHash {
	uint64_t bcan;
	uint64_t kms;
	uint32_t stn; // strand flags (spaced seeds), bit i = hash i came from the reverse strand
	uint64_t hv[SPACED ? kMaxHash : 1];
	__device__ __forceinline__ uint64_t at(uint32_t i) const
	{
		if (SPACED)
			return hv[SPACED ? i : 0];
		return i ? extra_hash(bcan, kms, i) : bcan;
	}
};

// LDS reads at any byte address (gfx950's DS unit takes unaligned b32/b64 addresses -- but REPLAYS them: a misaligned
// access costs several aligned ones, round-4 finding; the hot paths keep these reads aligned and use them unaligned only
// for callers' misaligned buffers)
__device__ __forceinline__ uint32_t lds_u32(const uint8_t* p)
{
	uint32_t v;
	__builtin_memcpy(&v, p, 4);
	return v;
}
__device__ __forceinline__ uint64_t lds_u64(const uint8_t* p)
{
	uint64_t v;
	__builtin_memcpy(&v, p, 8);
	return v;
}
__device__ __forceinline__ U64x2 tab16(const void* base, uint32_t byte_off)
{
	return *reinterpret_cast<const U64x2*>(static_cast<const uint8_t*>(base) + byte_off);
}

// The rolling state of a lane between two calls of seq_lane_range (the lane's windows may be walked in pieces:
// pass A of the partitioned pipeline hashes windows 0..3 and 4..7 of a tile at different times)
struct LaneState {
	uint64_t fh, rh; // forward / reverse strand hash of the current window
	uint64_t ob, ib; // the lane's outgoing (li0 .. li0+7) and incoming (li0+k .. li0+k+7) staged bases
	uint32_t good;   // "good" bases among the window's k-1 bases behind its first
	uint32_t first_valid;
};

// Walk windows [W0, W1) of the lane's KW consecutive windows (first base at LDS index li0) and call
// f(w, clean, const WinHash<SPACED>&) for each.  W0 == 0 starts the walk (start-up over the first window); a later
// call continues from the state the previous one left in `st`.
// HS > 0: the number of hashes per window (n_seeds * h2) is known at compile time, so the spaced-seed
// hash values stay in registers (statically indexed); HS = 0: any count, array indexed at run time.
template <bool SPACED, int KW, int HS, int W0, int W1, class F>
__device__ __forceinline__ void seq_lane_range(const uint8_t* tile, const SeqShared& sh, const HashParams& hp,
                                               const uint8_t* spaced_lds, uint32_t li0, LaneState& st, F&& f)
{
	static_assert(KW <= 8, "the lane's outgoing / incoming bases are held in one 64-bit register each");
	static_assert(0 <= W0 && W0 < W1 && W1 <= KW, "a piece of the lane's windows");
	const uint32_t k = hp.k;
	const uint8_t* pos_tab = spaced_lds; // 16-byte entries, entry (i, code) at byte i*128 + code*16
	const uint16_t* dc_idx = reinterpret_cast<const uint16_t*>(spaced_lds + seq_pos_tab_bytes(hp));
	const uint8_t* bp = tile + li0;
	if (W0 == 0) {
		uint64_t fh = 0, rh = 0;
		uint32_t good = 0; // "good" bases among the window's k bases (the first base's flag is taken out below)
		uint32_t i = 0;
		if (hp.use_pos_tab && !SPACED) {
			// first window from the PAIR table (seq_setup_tables): four bases per LDS word, one 16-byte entry and two
			// 64-bit XORs per two bases.  The entry of bases (c1, c2) of pair j sits at j * 256 + c1 * 64 + c2 * 16; a
			// staged byte holds its code in bits 6:4, so c1 * 64 + c2 * 16 = (b1 << 2 & 0xc0) | (b2 & 0x30) for the codes
			// 0..3 -- bit 6 of a byte marks the raw-byte codes 4..7, which this table does not hold
			uint32_t raw = 0;
			for (; i + 4 <= k; i += 4) {
				const uint32_t w = lds_u32(bp + i);
				good += __popc(w & (kBaseGood * 0x01010101u));
				raw |= w;
				const uint32_t x = w & 0x30303030u;
				const U64x2 t0 = tab16(pos_tab, i * 128 + (((x << 2) | (x >> 8)) & 0xf0u));
				const U64x2 t1 = tab16(pos_tab, i * 128 + 256 + (((x >> 14) | (x >> 24)) & 0xf0u));
				fh ^= t0.x ^ t1.x;
				rh ^= t0.y ^ t1.y;
			}
			if (i + 2 <= k) {
				const uint32_t e0 = bp[i], e1 = bp[i + 1];
				good += ((e0 / kBaseGood) & 1) + ((e1 / kBaseGood) & 1);
				raw |= e0 | e1;
				const U64x2 tt = tab16(pos_tab, i * 128 + ((e0 & 0x30u) << 2) + (e1 & 0x30u));
				fh ^= tt.x;
				rh ^= tt.y;
				i += 2;
			}
			if (i < k) { // odd k: the last base through the single-base row behind the pairs
				const uint32_t e = bp[i];
				good += (e / kBaseGood) & 1;
				const U64x2 tt = tab16(pos_tab, i * 128 + (e & kCodeOff));
				fh ^= tt.x;
				rh ^= tt.y;
				++i;
			}
			if (raw & 0x40404040u) { // one of the bytes 1 3 4 5 7 among the k bases: the Horner form (rare)
				fh = rh = 0;
				for (uint32_t q = 0; q < k; ++q) {
					const U64x2 tt = tab16(sh.init_tab, bp[q] & kCodeOff);
					fh = srol1(fh) ^ tt.x;
					rh = sror1(rh) ^ tt.y;
				}
			}
		} else if (hp.use_pos_tab) {
			// (spaced seeds keep the single-base table: their don't-care terms are looked up in it)
			for (; i + 4 <= k; i += 4) {
				const uint32_t w = lds_u32(bp + i);
				good += __popc(w & (kBaseGood * 0x01010101u));
#pragma unroll
				for (int b = 0; b < 4; ++b) {
					const U64x2 tt = tab16(pos_tab, (i + b) * (kNumCodes * 16) + ((w >> (8 * b)) & kCodeOff));
					fh ^= tt.x;
					rh ^= tt.y;
				}
			}
			for (; i < k; ++i) {
				const uint32_t e = bp[i];
				good += (e / kBaseGood) & 1;
				const U64x2 tt = tab16(pos_tab, i * (kNumCodes * 16) + (e & kCodeOff));
				fh ^= tt.x;
				rh ^= tt.y;
			}
		} else {
			// Horner form (large k: the positional table would not fit in LDS)
			for (; i + 4 <= k; i += 4) {
				const uint32_t w = lds_u32(bp + i);
				good += __popc(w & (kBaseGood * 0x01010101u));
#pragma unroll
				for (int b = 0; b < 4; ++b) {
					const U64x2 tt = tab16(sh.init_tab, (w >> (8 * b)) & kCodeOff);
					fh = srol1(fh) ^ tt.x;
					rh = sror1(rh) ^ tt.y;
				}
			}
			for (; i < k; ++i) {
				const uint32_t e = bp[i];
				good += (e / kBaseGood) & 1;
				const U64x2 tt = tab16(sh.init_tab, e & kCodeOff);
				fh = srol1(fh) ^ tt.x;
				rh = sror1(rh) ^ tt.y;
			}
		}
		// bases leaving (ob: li0 .. li0+7) and entering (ib: li0+k .. li0+k+7) the lane's windows.  li0 is a multiple of
		// four unless the caller's buffer is misaligned, li0 + k rarely is: three aligned words and two byte-wise funnel
		// shifts (the shift is the same in every lane) instead of one misaligned 8-byte read, which this chip replays
		st.ob = lds_u64(bp);
#ifdef BTLBF_IB_UNALIGNED
		st.ib = lds_u64(bp + k);
#else
		{
			const uint32_t a = li0 + k, sh = a & 3u;
			const uint32_t* w = reinterpret_cast<const uint32_t*>(tile + (a & ~3u));
			const uint32_t d0 = w[0], d1 = w[1], d2 = w[2]; // (inside the image: seq_tile_cap leaves 15 bytes of slack)
			const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh), hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
			st.ib = ((uint64_t)hi << 32) | lo;
		}
#endif
		st.first_valid = (uint32_t)st.ob & kBaseValid;
		st.good = good - (((uint32_t)st.ob / kBaseGood) & 1);
		st.fh = fh;
		st.rh = rh;
	}
	uint64_t fh = st.fh, rh = st.rh;
	const uint64_t ob = st.ob, ib = st.ib;
	uint32_t good = st.good, first_valid = st.first_valid;
	if constexpr (SPACED && HS > 0) {
		// Spaced seeds, one hash per seed, seed count known at compile time (pass A): the windows are walked in groups of
		// GRP.  The term a don't-care position contributes -- pos_tab[offset][base] -- is the same whichever seed leaves
		// that position out, and the GRP windows of a group read GRP consecutive bases at every offset.  So per DISTINCT
		// offset (HashParams::dcu; 23 for BASELINE config 5's four seeds, whose own lists add up to 35): one LDS read of
		// the GRP bases, one table entry per window, and XORs into exactly the seeds of the offset's mask (the mask is
		// uniform: scalar branches).  Offsets left out by EVERY seed go into the common base first.  Against the per-seed
		// walk below (a byte read, a table read and two 64-bit XORs per seed, window and position): a third fewer LDS
		// table reads, an eighth of the byte reads, no address arithmetic per seed.
		// HS > 0 is pass A, whose host side takes spaced seeds only when the union list exists (part_supported); the
		// per-seed walk below is then dead code for this instantiation and costs it no registers
		{
			// (groups of two windows: four need 96 registers for the seeds' running values and the two table buffers,
			// and pass A then spills 80 bytes per lane; two cost the same instructions per window bar the loop control)
			constexpr int GRP = BTLBF_SPACED_GRP;
			static_assert((W1 - W0) % GRP == 0 && W0 % GRP == 0, "whole groups of windows");
			const uint32_t* dcu = reinterpret_cast<const uint32_t*>(spaced_lds + seq_pos_tab_bytes(hp) + seq_dc_list_bytes(hp));
			const uint32_t n_all = hp.n_dcu_all, n_dcu = hp.n_dcu;
			static_assert(kMaxDcu == 64, "the union list is held in one register of a wave");
#pragma unroll
			for (int g0 = W0; g0 < W1; g0 += GRP) {
				uint64_t bf[GRP], br[GRP];
				bool okw[GRP];
#pragma unroll
				for (int q = 0; q < GRP; ++q) {
					const int w = g0 + q;
					if (w > 0) {
						const uint32_t eo = (uint32_t)(ob >> (8 * (w - 1))) & 0xffu;
						const uint32_t ei = (uint32_t)(ib >> (8 * (w - 1))) & 0xffu;
						const uint32_t en = (uint32_t)(ob >> (8 * w)) & 0xffu;
						const U64x2 tt = tab16(sh.pair_tab, ((eo & kCodeOff) * kNumCodes) | (ei & kCodeOff));
						fh = srol1(fh) ^ tt.x;
						rh = sror1(rh ^ tt.y);
						good += (ei / kBaseGood) & 1;
						good -= (en / kBaseGood) & 1;
						first_valid = en & kBaseValid;
					}
					okw[q] = first_valid && good == k - 1;
					bf[q] = fh;
					br[q] = rh;
				}
				// The list sits in ONE register of the wave (lane u holds entry u; kMaxDcu = 64) and an entry is fetched with
				// v_readlane (uniform, no memory access).  Offsets are taken CHK at a time: the CHK reads of the windows' bases
				// go out together, then the CHK x GRP table reads, then the XORs -- two LDS latencies per CHK offsets.  (One
				// offset at a time, base read -> table reads -> XORs, left three dependent LDS round trips per offset with
				// forty instructions to cover them; a two-deep software pipeline did no better: every read of the bases is
				// followed by a wait for ALL LDS operations, the in-flight table reads included.)
				constexpr int CHK = BTLBF_SPACED_CHK;
				const uint32_t my_e = dcu[threadIdx.x & 63u];
				auto entry = [&](uint32_t u) { return u < n_dcu ? (uint32_t)__builtin_amdgcn_readlane((int)my_e, (int)u) : 0u; };
				auto spans = [&](const uint32_t (&e)[CHK], uint32_t (&span)[CHK]) {
#pragma unroll
					for (int c = 0; c < CHK; ++c) {
						const uint32_t off = e[c] & 0xffffu;
						// byte reads: ONE unaligned ds_read_u16 per offset (half of these addresses are odd) made the whole kernel
						// 45 % slower -- 136 instead of 92 ms per 6x10^9 k-mers; misaligned LDS accesses are replayed on this
						// chip, they are not a free convenience
						span[c] = 0;
#pragma unroll
						for (int q = 0; q < GRP; ++q)
							span[c] |= (uint32_t)bp[g0 + off + q] << (8 * q);
					}
				};
				auto lookups = [&](const uint32_t (&e)[CHK], const uint32_t (&span)[CHK], U64x2 (&tt)[CHK][GRP]) {
#pragma unroll
					for (int c = 0; c < CHK; ++c) {
						const uint32_t row = (e[c] & 0xffffu) * (kNumCodes * 16);
#pragma unroll
						for (int q = 0; q < GRP; ++q)
							tt[c][q] = tab16(pos_tab, row + ((span[c] >> (8 * q)) & kCodeOff));
					}
				};
				// offsets left out by every seed: into the common base (entries past the list's end read row 0 and are
				// masked out with a zero mask below; here they are simply not there: n_all is exact)
				uint32_t u = 0;
				for (; u < n_all; u += CHK) {
					uint32_t e[CHK], span[CHK];
					U64x2 tt[CHK][GRP];
#pragma unroll
					for (int c = 0; c < CHK; ++c)
						e[c] = u + c < n_all ? entry(u + c) : 0u;
					spans(e, span);
					lookups(e, span, tt);
#pragma unroll
					for (int c = 0; c < CHK; ++c) {
						if (u + c < n_all) {
#pragma unroll
							for (int q = 0; q < GRP; ++q) {
								bf[q] ^= tt[c][q].x;
								br[q] ^= tt[c][q].y;
							}
						}
					}
				}
				uint64_t af[GRP][HS], ar[GRP][HS];
#pragma unroll
				for (int q = 0; q < GRP; ++q) {
#pragma unroll
					for (int j = 0; j < HS; ++j) {
						af[q][j] = bf[q];
						ar[q][j] = br[q];
					}
				}
				// the rest of the list comes in PAIRS of offsets that the same seeds leave out (the host groups the offsets by
				// their mask and pads a group of odd size with the zero row): the pair's two terms are XORed together first
				// and the sum goes into the seeds of the mask -- one set of scalar branches per pair, and for a mask of n
				// seeds 1 + n XORs of a term instead of 2n
				static_assert(CHK == 2, "offsets are paired by mask");
				// With two-base rows in LDS (hp.n_pair_rows: a launcher found the room) a pair costs ONE table read per window
				// and no XOR of its two terms; two pairs per trip.  The rows hold A C G T only: a group that meets one of the
				// raw-byte codes 4..7 at a listed offset is done again the one-offset way below (`redo`, by its whole wave).
				bool redo = hp.n_pair_rows == 0;
				if (!redo) {
					const uint8_t* prow = spaced_lds + seq_pos_tab_bytes(hp) + seq_dc_list_bytes(hp) + seq_union_list_bytes(hp);
					uint32_t raw = 0;
					for (u = n_all; u < n_dcu; u += 4) {
						uint32_t e[4], b[4][GRP];
#pragma unroll
						for (int c = 0; c < 4; ++c) {
							e[c] = entry(u + c);
							const uint32_t off = e[c] & 0xffffu;
#pragma unroll
							for (int q = 0; q < GRP; ++q) {
								b[c][q] = bp[g0 + off + q];
								// (a filler reads the base behind the window: a raw byte there sends the group the long way
								// round for nothing, which is cheaper than a test per entry)
								raw |= b[c][q];
							}
						}
						U64x2 tt[2][GRP];
#pragma unroll
						for (int p = 0; p < 2; ++p) {
							const uint32_t row = ((u - n_all) / 2 + p) * 256;
#pragma unroll
							for (int q = 0; q < GRP; ++q)
								tt[p][q] = tab16(prow, row + (((b[2 * p][q] << 2) & 0xc0u) | (b[2 * p + 1][q] & 0x30u)));
						}
#pragma unroll
						for (int p = 0; p < 2; ++p) {
							const uint32_t m = e[2 * p] >> 16;
#pragma unroll
							for (int j = 0; j < HS; ++j) {
								if ((m >> j) & 1u) {
#pragma unroll
									for (int q = 0; q < GRP; ++q) {
										af[q][j] ^= tt[p][q].x;
										ar[q][j] ^= tt[p][q].y;
									}
								}
							}
						}
					}
					redo = __any((raw & 0x40u) != 0) != 0; // (the whole wave: the loops below stay uniform)
					if (redo) {
#pragma unroll
						for (int q = 0; q < GRP; ++q) {
#pragma unroll
							for (int j = 0; j < HS; ++j) {
								af[q][j] = bf[q];
								ar[q][j] = br[q];
							}
						}
					}
				}
				for (u = n_all; redo && u < n_dcu; u += 2) {
					uint32_t e[CHK], span[CHK];
					U64x2 tt[CHK][GRP];
					e[0] = entry(u);
					e[1] = entry(u + 1);
					spans(e, span);
					lookups(e, span, tt);
					const uint32_t m = e[0] >> 16;
#pragma unroll
					for (int q = 0; q < GRP; ++q) {
						tt[0][q].x ^= tt[1][q].x;
						tt[0][q].y ^= tt[1][q].y;
					}
#pragma unroll
					for (int j = 0; j < HS; ++j) {
						if ((m >> j) & 1u) {
#pragma unroll
							for (int q = 0; q < GRP; ++q) {
								af[q][j] ^= tt[0][q].x;
								ar[q][j] ^= tt[0][q].y;
							}
						}
					}
				}
#pragma unroll
				for (int q = 0; q < GRP; ++q) {
					WinHash<SPACED> wh;
					wh.kms = hp.kms;
					wh.stn = 0;
					// (not used by the spaced path's consumers, kept meaningful: the canonical hash of the whole window)
					wh.bcan = 0;
					if (hp.h2 == 1) {
#pragma unroll
						for (int j = 0; j < HS; ++j) {
							const bool rev = ar[q][j] < af[q][j];
							wh.hv[SPACED ? j : 0] = rev ? ar[q][j] : af[q][j];
							wh.stn |= (uint32_t)rev << j;
						}
					} else {
						// h2 hashes per seed: hash j*h2 is the seed's own value, hash j*h2 + j2 its j2-th extra hash
						// (nthash.hpp:847-852).  (j, j2) run along as uniform counters; the seed's value is picked from
						// the registers with a select chain (the index is not a compile-time constant here)
						uint32_t j = 0, j2 = 0;
						uint64_t b = 0;
						bool rev = false;
#pragma unroll
						for (int idx = 0; idx < HS; ++idx) {
							if (j2 == 0) {
								uint64_t fs = 0, rs = 0;
#pragma unroll
								for (int jj = 0; jj < HS; ++jj) {
									fs = j == (uint32_t)jj ? af[q][jj] : fs;
									rs = j == (uint32_t)jj ? ar[q][jj] : rs;
								}
								rev = rs < fs;
								b = rev ? rs : fs;
							}
							wh.hv[SPACED ? idx : 0] = j2 ? extra_hash(b, hp.kms, j2) : b;
							wh.stn |= (uint32_t)rev << idx;
							if (++j2 == hp.h2) {
								j2 = 0;
								++j;
							}
						}
					}
					f(g0 + q, okw[q], wh);
				}
			}
			st.fh = fh;
			st.rh = rh;
			st.good = good;
			st.first_valid = first_valid;
			return;
		}
	}
#pragma unroll
	for (int w = W0; w < W1; ++w) {
		if (w > 0) {
			const uint32_t eo = (uint32_t)(ob >> (8 * (w - 1))) & 0xffu;
			const uint32_t ei = (uint32_t)(ib >> (8 * (w - 1))) & 0xffu;
			const uint32_t en = (uint32_t)(ob >> (8 * w)) & 0xffu;
			const U64x2 tt = tab16(sh.pair_tab, ((eo & kCodeOff) * kNumCodes) | (ei & kCodeOff));
			fh = srol1(fh) ^ tt.x;
			rh = sror1(rh ^ tt.y);
			good += (ei / kBaseGood) & 1;
			good -= (en / kBaseGood) & 1;
			first_valid = en & kBaseValid;
		}
		const bool ok = first_valid && good == k - 1;
		WinHash<SPACED> wh;
		wh.kms = hp.kms;
		wh.stn = 0;
		wh.bcan = rh < fh ? rh : fh;
		if (SPACED) {
			const uint32_t h2 = hp.h2;
			auto seed_base = [&](uint32_t j, bool& rev) {
				uint64_t fs = fh, rs = rh;
				for (uint32_t d = hp.dc_off[j]; d < hp.dc_off[j + 1]; ++d) {
					const uint32_t di = dc_idx[d];
					const U64x2 tt = tab16(pos_tab, di * (kNumCodes * 16) + (bp[w + di] & kCodeOff));
					fs ^= tt.x;
					rs ^= tt.y;
				}
				rev = rs < fs;
				return rev ? rs : fs;
			};
			if (HS > 0) {
				// hash idx = seed j, extra hash j2 with idx = j*h2 + j2: (j, j2) are tracked as uniform
				// counters so that idx -- the array index -- is a compile-time constant
				uint32_t j = 0, j2 = 0;
				uint64_t b = 0;
				bool rev = false;
#pragma unroll
				for (int idx = 0; idx < (HS > 0 ? HS : 1); ++idx) {
					if (j2 == 0)
						b = seed_base(j, rev);
					wh.hv[SPACED ? idx : 0] = j2 ? extra_hash(b, hp.kms, j2) : b;
					if (rev)
						wh.stn |= 1u << idx;
					if (++j2 == h2) {
						j2 = 0;
						++j;
					}
				}
			} else {
				for (uint32_t j = 0; j < hp.n_seeds; ++j) {
					bool rev;
					const uint64_t b = seed_base(j, rev);
					wh.hv[SPACED ? j * h2 : 0] = b;
					for (uint32_t j2 = 1; j2 < h2; ++j2)
						wh.hv[SPACED ? j * h2 + j2 : 0] = extra_hash(b, hp.kms, j2);
					if (rev)
						for (uint32_t j2 = 0; j2 < h2; ++j2)
							wh.stn |= 1u << (j * h2 + j2); // h <= 32 here
				}
			}
		}
		f(w, ok, wh);
	}
	st.fh = fh;
	st.rh = rh;
	st.good = good;
	st.first_valid = first_valid;
}

// all KW windows of the lane in one go (the direct kernels; pass A's original schedule)
template <bool SPACED, int KW = kW, int HS = 0, class F>
__device__ __forceinline__ void seq_lane_windows(const uint8_t* tile, const SeqShared& sh, const HashParams& hp,
                                                 const uint8_t* spaced_lds, uint32_t li0, F&& f)
{
	LaneState st;
	seq_lane_range<SPACED, KW, HS, 0, KW>(tile, sh, hp, spaced_lds, li0, st, f);
}

// advance "offset of the tile start inside its read" by one tile (uniform layout)
__device__ __forceinline__ uint32_t seq_next_tile_off(uint32_t tile_off, uint32_t tile_step, uint32_t L)
{
	tile_off += tile_step;
	if (L && tile_off >= L)
		tile_off -= L;
	return tile_off;
}

} // namespace btlbf

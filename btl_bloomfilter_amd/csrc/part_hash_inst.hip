// csrc/part_hash_inst.hip -- pass A of the partitioned pipeline (fused ntHash + radix partition),
// compiled once per hash count: -DBTLBF_PART_H=n defines launch_part_hash_h<n>.  One translation
// unit per n keeps the build parallel (each holds 8 variants of a large kernel).
#include "partition_core.hpp"

#ifndef BTLBF_PART_H
#error "compile with -DBTLBF_PART_H=<1..8>"
#endif

namespace btlbf {

// ---- pass A --------------------------------------------------------------------------------------
// bin = position >> bin_shift ; entry = position & ((1 << bin_shift) - 1); region = blockIdx.x
// (gridDim.x == out.regions).  `position` is local to a.mod's shard window (the whole filter in
// routing mode, where a.mod describes the GLOBAL filter).
// WINDOW: the filter object is one shard of a larger filter and keeps only the positions inside its
// window (a.mod.shard_lo, shard_len); otherwise every probe of a clean window is an entry.
template <int H, bool POW2, bool SPACED, bool QUERY, bool WINDOW>
__global__ __launch_bounds__(kPartThreads) void part_hash_kernel(const SeqArgs a, const PartOut out,
                                                                const uint32_t bin_shift, const PartSide sd)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
	__shared__ SeqShared sh;
	const uint32_t tid = threadIdx.x;
	const uint32_t k = a.hp.k;
	const uint32_t tile_cap = seq_tile_cap(kPartTile, k);
	uint8_t* tile = dyn;
	uint8_t* spaced_lds = dyn + tile_cap;
	const PartLds pl = part_carve(dyn + tile_cap + seq_spaced_bytes(a.hp), out.P);
	seq_setup_tables<kPartThreads, SPACED>(sh, a.hp, spaced_lds);
	part_init<kPartThreads>(pl, out.P);

	uint32_t* words = static_cast<uint32_t*>(a.filter);
	const uint32_t ent_mask = bin_shift >= 32 ? 0xffffffffu : (1u << bin_shift) - 1;
	auto ovf = [&](uint32_t b, uint32_t v) { part_direct<QUERY>(words, sd, ((uint64_t)b << bin_shift) | v); };
	const uint64_t out_bytes = ((a.len + 63) / 64) * 8;
	uint32_t my_valid = 0;

	const uint64_t t_begin = a.first_tile + (uint64_t)blockIdx.x * a.tiles_per_block;
	uint64_t t_end = t_begin + a.tiles_per_block;
	if (t_end > a.first_tile + a.n_tiles)
		t_end = a.first_tile + a.n_tiles;
	const uint32_t L = a.layout.starts ? 0 : a.layout.read_len;
	uint32_t tile_off = 0;
	if (L && t_begin < t_end)
		tile_off = (uint32_t)((t_begin * (uint64_t)kPartTile) % L);
	const uint32_t tile_step = L ? (uint32_t)(kPartTile % L) : 0;

	__syncthreads(); // tables and partition state ready
	STAMP_DECL;
	StageRaw<kPartW> raw;
	if (t_begin < t_end)
		seq_stage_load<kPartThreads, kPartW>(raw, a.seq, a.len, k, t_begin * (uint64_t)kPartTile);
	for (uint64_t t = t_begin; t < t_end; ++t) {
		const uint64_t g0 = t * (uint64_t)kPartTile;
		STAMP(0);
		// this tile's words were requested a whole tile ago; the next tile's are requested now and stay
		// in flight while this one is hashed and partitioned
#ifdef BTLBF_PHASE_STAMPS
		const uint32_t mis = seq_stage_convert<kPartThreads, kPartW, false, false>(raw, tile, tile_cap, sh, a.seq, a.len, a.layout, k, g0, tile_off);
		STAMP(1); // conversion of this thread's words
		__syncthreads();
		STAMP(3); // waiting for the other waves
#else
		const uint32_t mis = seq_stage_convert<kPartThreads, kPartW, false>(raw, tile, tile_cap, sh, a.seq, a.len, a.layout, k, g0, tile_off);
#endif
		if (t + 1 < t_end)
			seq_stage_load<kPartThreads, kPartW>(raw, a.seq, a.len, k, g0 + kPartTile);
		tile_off = seq_next_tile_off(tile_off, tile_step, L);

		// the lane hashes its 8 consecutive windows with ONE start-up; after every 4 windows the
		// 4*H entries collected so far go through a partition round (the rolling state stays in
		// registers across it)
		uint32_t bin[kPartHalf * H], val[kPartHalf * H];
		uint32_t vmask = 0, live = 0;
		seq_lane_windows<SPACED, kPartW, H>(tile, sh, a.hp, spaced_lds, tid * kPartW + mis, [&](int w, bool ok, const WinHash<SPACED>& wh) {
			vmask |= (uint32_t)ok << w;
			const int w4 = w % kPartHalf;
			if (w4 == 0)
				live = 0;
			if (!WINDOW)
				live |= (uint32_t)ok << w4;
#pragma unroll
			for (int i = 0; i < H; ++i) {
				uint64_t p = reduce_mod<POW2>(wh.at(i), a.mod);
				if (WINDOW) {
					p -= a.mod.shard_lo;
					live |= (uint32_t)(ok && p < a.mod.shard_len) << (w4 * H + i);
				}
				bin[w4 * H + i] = (uint32_t)(p >> bin_shift);
				val[w4 * H + i] = (uint32_t)p & ent_mask;
			}
			if (w4 == kPartHalf - 1) {
				STAMP(2);
				part_round<kPartThreads, kPartHalf * H, WINDOW ? 1 : H>(pl, out, 0, blockIdx.x, bin, val, live, ovf STAMP_PASS);
			}
		});
		if (a.valid_bits || a.hit_bits) {
			// one byte of the per-window bitmaps per lane
			static_assert(kPartW == 8, "one bitmap byte per lane");
			const uint64_t ob = (g0 >> 3) + tid;
			if (ob < out_bytes) {
				if (a.valid_bits)
					a.valid_bits[ob] = (uint8_t)vmask;
				if (a.hit_bits)
					a.hit_bits[ob] = (uint8_t)vmask; // a query starts from "every clean window hits"
			}
		}
		my_valid += __popc(vmask);
	}
	part_finish<kPartThreads>(pl, out, 0, blockIdx.x, ovf);
	if (a.counts) {
		const uint32_t wv = wave_sum(my_valid);
		if ((tid & 63) == 0 && wv)
			atomicAdd(reinterpret_cast<unsigned long long*>(a.counts), (unsigned long long)wv);
	}
	STAMP(9);
	STAMP_FLUSH;
}

template <int H, bool Q>
static hipError_t launch_hash_h(const SeqArgs& a, const PartOut& out, uint32_t bin_shift, const PartSide& sd,
                                size_t dyn, hipStream_t s)
{
	const bool pow2 = a.mod.pow2 != 0, spaced = a.hp.n_seeds > 0;
	const bool window = a.mod.shard_lo != 0 || a.mod.shard_len != a.mod.size;
#define BTLBF_PLAUNCH(P, S, W)                                                                                  \
	do {                                                                                                        \
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&part_hash_kernel<H, P, S, Q, W>),       \
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);                \
		if (e != hipSuccess)                                                                                    \
			return e;                                                                                           \
		hipLaunchKernelGGL((part_hash_kernel<H, P, S, Q, W>), dim3(out.regions), dim3(kPartThreads), dyn, s, a, \
		                   out, bin_shift, sd);                                                                 \
	} while (0)
#define BTLBF_PLAUNCH_W(P, S)        \
	do {                             \
		if (window)                  \
			BTLBF_PLAUNCH(P, S, true);  \
		else                         \
			BTLBF_PLAUNCH(P, S, false); \
	} while (0)
	if (pow2 && !spaced)
		BTLBF_PLAUNCH_W(true, false);
	else if (!pow2 && !spaced)
		BTLBF_PLAUNCH_W(false, false);
	else if (pow2 && spaced)
		BTLBF_PLAUNCH_W(true, true);
	else
		BTLBF_PLAUNCH_W(false, true);
#undef BTLBF_PLAUNCH_W
#undef BTLBF_PLAUNCH
	return hipGetLastError();
}


#define BTLBF_CAT2(a, b) a##b
#define BTLBF_CAT(a, b) BTLBF_CAT2(a, b)
hipError_t BTLBF_CAT(launch_part_hash_h, BTLBF_PART_H)(const SeqArgs& a, const PartOut& out, uint32_t bin_shift,
                                                       const PartSide& sd, size_t dyn, int query, hipStream_t s)
{
	return query ? launch_hash_h<BTLBF_PART_H, true>(a, out, bin_shift, sd, dyn, s)
	             : launch_hash_h<BTLBF_PART_H, false>(a, out, bin_shift, sd, dyn, s);
}

#if defined(BTLBF_PHASE_STAMPS) && BTLBF_PART_H == 4
extern "C" void btlbf_debug_stamps(uint64_t* out16)
{
	(void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamp_out), sizeof(uint64_t) * 16);
	uint64_t z[16] = {0};
	(void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_out), z, sizeof z);
}
#endif

} // namespace btlbf

// csrc/part_hash_inst.hip -- pass A of the partitioned pipeline (fused ntHash + radix partition),
// compiled once per hash count: -DBTLBF_PART_H=n defines launch_part_hash_h<n>.  One translation
// unit per n keeps the build parallel (each holds 8 variants of a large kernel).
#define BTLBF_NT_FLUSH 1 // chunk flushes as streaming stores in this translation unit (partition_core.hpp part_round_p2)
#include "partition_core.hpp"
#include <cstdlib>
#include <cstring>

#ifndef BTLBF_PART_H
#error "compile with -DBTLBF_PART_H=<1..8>"
#endif

namespace btlbf {

// A kernel argument read again from the kernarg segment where it is used, instead of being held in scalar registers
// for the whole kernel.  Pass A's tile loop lives at the limit of both register files; what the compiler cannot hold
// it spills -- uniform values via two lane-spill registers and, beyond those, to SCRATCH --, and a scratch reload
// inside the loop waits for every vector-memory operation in flight, the flush stores included.  Arguments that a tile
// needs two or three times (the buffer, its length, the bitmaps) cost a scalar load each time instead: the pointer is
// laundered through an empty asm, so the load cannot be hoisted out of the loop.
template <class T>
__device__ __forceinline__ T kernarg_reload(uint32_t byte_off)
{
	typedef const uint8_t __attribute__((address_space(4))) kbyte;
	kbyte* kp = (kbyte*)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(kp));
	return *(const T __attribute__((address_space(4)))*)(kp + byte_off);
}
// (SeqArgs is the first argument of both pass-A kernels: its fields lie at their offsets in the segment)
#define KARG(field) kernarg_reload<decltype(SeqArgs::field)>((uint32_t)offsetof(SeqArgs, field))
#define KARG_LAYOUT(field) \
	kernarg_reload<decltype(LayoutParams::field)>((uint32_t)(offsetof(SeqArgs, layout) + offsetof(LayoutParams, field)))

// ---- pass A --------------------------------------------------------------------------------------
// bin = position >> bin_shift ; entry = position & ((1 << bin_shift) - 1); region = blockIdx.x
// (gridDim.x == out.regions).  `position` is local to a.mod's shard window (the whole filter in
// routing mode, where a.mod describes the GLOBAL filter).
// WINDOW: the filter object is one shard of a larger filter and keeps only the positions inside its
// window (a.mod.shard_lo, shard_len); otherwise every probe of a clean window is an entry.
// SMALL: the two-workgroups-per-CU geometry (partition_core.hpp): 512 threads, tiles of 4096 windows, 64 KiB
// of rings; otherwise 1024 threads, tiles of 8192 windows, 128 KiB of rings.  a.first_tile / a.n_tiles /
// a.tiles_per_block are in units of THIS kernel's tile.
template <int H, bool POW2, bool SPACED, bool QUERY, bool WINDOW, bool SMALL>
__global__ __launch_bounds__(SMALL ? kPartThreadsS : kPartThreads, SMALL ? 4 : 1) void part_hash_kernel(
    const SeqArgs a, const PartOut out, const uint32_t bin_shift, const PartSide sd)
{
	constexpr int NT = SMALL ? kPartThreadsS : kPartThreads;
	constexpr int kTile = NT * kPartW;
	extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
	__shared__ SeqShared sh;
	const uint32_t tid = threadIdx.x;
	const uint32_t k = a.hp.k;
	// read grid (internal.hpp PartGrid; never with the small geometry): tiles of rg_reads whole reads
	const bool grid = !SMALL && a.rg_reads != 0;
	const uint32_t tile_cap = grid ? a.rg_cap : seq_tile_cap(kTile, k);
	const uint32_t tile_bytes = grid ? a.rg_reads * a.layout.read_len : (uint32_t)kTile; // window starts per tile
	uint8_t* tile = dyn;
	uint8_t* spaced_lds = dyn + tile_cap + a.sb_words * 4; // (the overlapped schedule's start bitmap: not used here)
	uint8_t* part_base = spaced_lds + seq_spaced_bytes(a.hp);
	PartLds pl{};
	PartLdsS ps{};
	if (SMALL) {
		ps = part_carve_s(part_base, out.P);
		part_init_s(ps, out.P);
	} else {
		pl = part_carve(part_base, out.P);
		part_init<kPartThreads>(pl, out.P);
	}
	seq_setup_tables<NT, SPACED>(sh, a.hp, spaced_lds);

	uint32_t* words = static_cast<uint32_t*>(a.filter);
	const uint32_t ent_mask = bin_shift >= 32 ? 0xffffffffu : (1u << bin_shift) - 1;
	// a power-of-two filter taken whole: bin and entry are bit fields of the hash itself (no 64-bit `& mask` first)
	const uint32_t bin_mask = (uint32_t)(a.mod.mask >> bin_shift);
	const uint32_t ent_mask_p2 = ent_mask & (uint32_t)a.mod.mask; // a filter smaller than one bin
	// (a power-of-two filter taken whole never has bins of whole segments, and must not read sd.bin_* here: with
	// part_bin_base for every variant the benchmark's query kernel spilled 16 bytes per lane and ran 44 instead of
	// 41 ms, the ragged insert 55.6 instead of 46.5 -- tools/kres.py after every change to this file)
	auto ovf = [&](uint32_t b, uint32_t v) {
		part_direct<QUERY>(words, sd, POW2 && !WINDOW ? ((uint64_t)b << bin_shift) | v : part_bin_base(sd, b, bin_shift) + v);
	};
	const uint64_t out_bytes = ((a.len + 63) / 64) * 8;
	uint32_t my_valid = 0;

	const uint64_t t_begin = a.first_tile + (uint64_t)blockIdx.x * a.tiles_per_block;
	uint64_t t_end = t_begin + a.tiles_per_block;
	if (t_end > a.first_tile + a.n_tiles)
		t_end = a.first_tile + a.n_tiles;
	const uint32_t L = a.layout.starts ? 0 : a.layout.read_len;
	uint32_t tile_off = 0;
	if (L && t_begin < t_end && !grid)
		tile_off = (uint32_t)((t_begin * (uint64_t)kTile) % L);
	const uint32_t tile_step = L && !grid ? (uint32_t)(kTile % L) : 0;
	// grid: the lane's 8 window starts are starts 8m .. 8m+7 of read rd of the tile (the same in every tile);
	// lanes beyond the grid look at the zeroed guard behind the last read and find no k-mer there
	uint32_t grid_li0 = 0, grid_bit = 0;
	uint32_t* grid_bm = nullptr; // two bitmaps of the tile's window starts, used by turns
	uint32_t grid_bm_words = 0;
	const bool want_bits = a.valid_bits || a.hit_bits;
	if (grid) {
		const uint32_t rd = tid / a.rg_gpr, m = tid - rd * a.rg_gpr;
		const bool in_grid = rd < a.rg_reads;
		grid_li0 = in_grid ? rd * a.rg_lpad + 8 * m : a.rg_reads * a.rg_lpad;
		grid_bit = in_grid ? rd * L + 8 * m : 0xffffffffu;
		grid_bm_words = ((tile_bytes / 8 + 15) / 16) * 4;
		grid_bm = reinterpret_cast<uint32_t*>(tile + tile_cap) - 2 * grid_bm_words;
		for (uint32_t i = tid; i < tile_cap / 4; i += NT) // pads, guard and bitmaps start as zeros
			reinterpret_cast<uint32_t*>(tile)[i] = 0;
	}
	const uint32_t span = grid ? tile_bytes : 0; // bytes seq_stage_load requests (0: kTile + k - 1)

	__syncthreads(); // tables and partition state ready
	STAMP_DECL;
	StageRaw<kPartW> raw;
	if (t_begin < t_end)
		seq_stage_load<NT, kPartW>(raw, a.seq, a.len, k, t_begin * (uint64_t)tile_bytes, span);
	if (SMALL) {
		// the first tile's words land here; every later tile's land inside the partition rounds (below), so
		// the top of the loop never waits on vector memory -- a wait there would sit behind the flush stores
#pragma unroll
		for (int q = 0; q < kPartW / 4 + 1; ++q)
			asm volatile("" : "+v"(raw.w[q]));
	}
	for (uint64_t t = t_begin; t < t_end; ++t) {
		const uint64_t g0 = t * (uint64_t)tile_bytes;
		STAMP(0);
		// this tile's words were requested a whole tile ago; the next tile's are requested now and stay
		// in flight while this one is hashed and partitioned
		uint32_t mis = 0;
		if (grid) {
			// (a read mask is looked up in global memory here: this schedule stages at the top of the tile, in front of
			// the partition rounds' stores)
			seq_stage_convert_grid<NT, kPartW>(raw, tile, sh, a.seq, a.len, L, a.rg_lpad, tile_bytes, g0, -1, a.read_mask,
			                                   (uint32_t)(t * a.rg_reads));
			STAMP(1); // conversion of this thread's words
			__syncthreads();
		} else {
			mis = seq_stage_convert<NT, kPartW, false>(raw, tile, tile_cap, sh, a.seq, a.len, a.layout, k, g0, tile_off);
			STAMP(1);
		}
		STAMP(3); // waiting for the other waves
		if (t + 1 < t_end)
			seq_stage_load<NT, kPartW>(raw, a.seq, a.len, k, g0 + tile_bytes, span);
		tile_off = seq_next_tile_off(tile_off, tile_step, L);

		// the lane hashes its 8 consecutive windows with ONE start-up; after every 4 windows the
		// 4*H entries collected so far go through a partition round (the rolling state stays in
		// registers across it)
		// (more than four hashes per k-mer: rounds of two windows -- 4h entries per lane and round, with their ring
		// slots and the rolling hash state, are more registers than a lane has: h = 8 spilled its way to 27 ms per
		// 10^9 k-mers insert and 47 query, against 6.9 at h = 4)
		constexpr int kWpr = H <= 4 ? kPartHalf : 2; // windows per partition round
		uint32_t bin[kWpr * H], val[kWpr * H];
		uint32_t vmask = 0, live = 0;
		seq_lane_windows<SPACED, kPartW, H>(tile, sh, a.hp, spaced_lds, grid ? grid_li0 : tid * kPartW + mis, [&](int w, bool ok, const WinHash<SPACED>& wh) {
			vmask |= (uint32_t)ok << w;
			if (w == kPartW - 1 && grid && want_bits && grid_bit != 0xffffffffu) {
				// the lane's byte of the window bitmap sits at bit 8m of its read: collected in LDS (this
				// tile's bitmap; the partition round below has the barriers) and written out whole afterwards
				uint32_t* bm = grid_bm + ((t - t_begin) & 1) * grid_bm_words;
				const uint32_t sh5 = grid_bit & 31;
				if (vmask << sh5)
					atomicOr(&bm[grid_bit >> 5], vmask << sh5);
				if (sh5 > 24 && (vmask >> (32 - sh5))) // (never beyond the bitmap: bits past a read's last window are 0)
					atomicOr(&bm[(grid_bit >> 5) + 1], vmask >> (32 - sh5));
			}
			const int w4 = w % kWpr;
			if (w4 == 0)
				live = 0;
			if (!WINDOW)
				live |= (uint32_t)ok << w4;
			// (no power of two: the form of the reduction is picked once per window, not once per probe)
			auto probes = [&](auto&& position) {
				auto place = [&](auto&& bin_of) {
#pragma unroll
					for (int i = 0; i < H; ++i) {
						if (POW2 && !WINDOW) {
							const uint64_t hv = wh.at(i);
							bin[w4 * H + i] = (uint32_t)(hv >> bin_shift) & bin_mask;
							val[w4 * H + i] = (uint32_t)hv & ent_mask_p2;
							continue;
						}
						uint64_t p = position(wh.at(i));
						if (WINDOW) {
							p -= a.mod.shard_lo;
							live |= (uint32_t)(ok && p < a.mod.shard_len) << (w4 * H + i);
						}
						bin_of(p, bin[w4 * H + i], val[w4 * H + i]);
					}
				};
				// (bins of sd.bin_wseg segments where the local array's segment count is no power of two: part_bin_of)
				if (!(POW2 && !WINDOW) && sd.bin_wseg)
					place([&](uint64_t p, uint32_t& b, uint32_t& v) { part_bin_of(sd, p, b, v); });
				else
					place([&](uint64_t p, uint32_t& b, uint32_t& v) {
						b = (uint32_t)(p >> bin_shift);
						v = (uint32_t)p & ent_mask;
					});
			};
			if (POW2)
				probes([&](uint64_t hv) { return hv & a.mod.mask; });
			else if (mod_small_magic(a.mod))
				probes([&](uint64_t hv) { return reduce_mod_small(hv, a.mod); });
			else
				probes([&](uint64_t hv) { return reduce_mod_big(hv, a.mod); });
			if (w4 == kWpr - 1) {
				STAMP(2);
				if (SMALL) {
					// the next tile's words (requested at the top of this tile) are pinned in their registers
					// before each flush issues its stores (in both rounds, so that on no path the compiler
					// still sees them in flight at the top of the next tile, behind the stores)
					auto land = [&]() {
#pragma unroll
						for (int q = 0; q < kPartW / 4 + 1; ++q)
							asm volatile("" : "+v"(raw.w[q]));
					};
					part_round_s<kWpr * H, WINDOW ? 1 : H>(ps, out, blockIdx.x, bin, val, live, ovf, land STAMP_PASS);
				}
				else
					part_round<kPartThreads, kWpr * H, WINDOW ? 1 : H>(pl, out, 0, blockIdx.x, bin, val, live, ovf STAMP_PASS);
			}
		});
		if (want_bits && grid) {
			// this tile's bitmap is complete (two barriers since the last OR); the other one -- written out a
			// tile ago -- is cleared for the next tile
			uint32_t* bm = grid_bm + ((t - t_begin) & 1) * grid_bm_words;
			uint32_t* other = grid_bm + (((t - t_begin) & 1) ^ 1) * grid_bm_words;
			const uint8_t* bm8 = reinterpret_cast<const uint8_t*>(bm);
			const uint64_t ob0 = g0 >> 3; // tiles are whole bytes of the bitmaps
			// the buffer's last tile also writes the (zero) bytes up to the end of the bitmaps' last 64-bit word
			const uint64_t left = out_bytes > ob0 ? out_bytes - ob0 : 0;
			const uint32_t n_out = g0 + tile_bytes >= a.len || left < tile_bytes / 8 ? (uint32_t)left : tile_bytes / 8;
			for (uint32_t i = tid; i < n_out; i += NT) {
				const uint8_t v = i < tile_bytes / 8 ? bm8[i] : (uint8_t)0;
				if (a.valid_bits)
					a.valid_bits[ob0 + i] = v;
				if (a.hit_bits)
					a.hit_bits[ob0 + i] = v; // a query starts from "every clean window hits"
			}
			for (uint32_t i = tid; i < grid_bm_words; i += NT)
				other[i] = 0;
		} else if (want_bits) {
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
	if (SMALL)
		part_finish_s(ps, out, blockIdx.x, ovf);
	else
		part_finish<kPartThreads>(pl, out, 0, blockIdx.x, ovf);
	if (a.counts) {
		const uint32_t wv = wave_sum(my_valid);
		if ((tid & 63) == 0 && wv)
			atomicAdd(reinterpret_cast<unsigned long long*>(a.counts), (unsigned long long)wv);
	}
	STAMP(9);
	STAMP_FLUSH;
}

// ---- pass A, overlapped schedule (1024 threads, at most 64 * kOvOwners level-0 bins) ------------------------------
// The plain schedule above runs every phase of a partition round on all 16 waves at once: while everybody waits for
// LDS atomics the vector units idle, while everybody hashes the LDS idles, and during the flush only the waves that
// own bins work (with its two round barriers left out -- wrong results, a diagnostic build -- the kernel takes 24 %
// less time).  Here the waves have roles.  Waves 0..7 (X) own all the bins; waves 8..15 (Y) own none and use the
// flush phase for work that needs no ring state:
//   round 0 of a tile (windows 0..3):  all: hash, atomics + ring writes | barrier | X: flush   Y: hash windows 4..7
//   round 1 of a tile (windows 4..7):  X: hash; all: atomics + writes   | barrier | X: flush   Y: stage the NEXT tile
// so Y's atomics of round 1 overlap X's hashing, Y's hashing and ALL of the tile staging overlap X's flushes, and
// the top-of-tile staging phase (15 % of the plain schedule) is gone.  Hashing ahead overwrites the registers that
// hold the current round's entries, and phase 3 still needs the "late" ones among them (0.8 % of the entries with 512
// bins, but nearly every wave has one): here phase 1 stores a late entry in the round's LATE IMAGE, a mirror of the
// rings in global memory (part_round_p1_late: addressed by what the atomic returned, no compaction; L2 traffic), and
// the Y waves fetch the late entries back before the round's second barrier and write them into the rings behind it
// -- no entry lives in a register across a barrier.  Same rounds, same ring protocol, same output as the plain
// schedule (bit-identical filters: tests/test_gpu_parity.py, tools/fuzz_parity.py).
static constexpr int kOvOwners = 8;

// AUX: the instantiation that knows the ragged layout's start bitmap and the split query's read mask.  Both cost
// registers in the tile loop whether a launch uses them or not (the kernel sits at its 128 with nothing to spare: with
// them compiled in, the variants for filters of no power-of-two size went from 16 to 60 bytes of scratch per lane and
// from 57.7 to 62.6 ms per launch), so launches that use neither take the instantiation without.
template <int H, bool POW2, bool SPACED, bool QUERY, bool WINDOW, bool AUX>
__global__ __launch_bounds__(kPartThreads, 1) void part_hash_ov_kernel(const SeqArgs a, const PartOut out,
                                                                      const uint32_t bin_shift, const PartSide sd)
{
	constexpr int NT = kPartThreads;
	constexpr int kTile = NT * kPartW;
	constexpr int NY = NT - kOvOwners * 64;                              // staging threads (the Y waves)
	constexpr int kStageKW = ((kPartW / 4 + 1) * (NT / NY) - 1) * 4;     // StageRaw<kStageKW>: 6 words per Y thread
	constexpr int E = kPartHalf * H;
	extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
	__shared__ SeqShared sh;
	__shared__ uint32_t p2_done; // owner waves that have left their late words (fl[]) of a round: kOvOwners per round
	const uint32_t tid = threadIdx.x;
	const uint32_t k = a.hp.k;
	const bool isY = (tid >> 6) >= (uint32_t)kOvOwners; // uniform over a wave
	const int32_t ytid = (int32_t)tid - kOvOwners * 64;
	const bool grid = a.rg_reads != 0; // read grid (internal.hpp PartGrid): tiles of rg_reads whole reads
	const uint32_t tile_cap = grid ? a.rg_cap : seq_tile_cap(kTile, k);
	const uint32_t tile_bytes = grid ? a.rg_reads * a.layout.read_len : (uint32_t)kTile; // window starts per tile
	uint8_t* tile = dyn;
	// ragged layout: one bit per staged byte of the NEXT tile, set where a sequence starts (see `mark_ahead` below)
	uint32_t* const sbm = reinterpret_cast<uint32_t*>(dyn + tile_cap);
	const bool sb = AUX && a.sb_words != 0;
	__shared__ uint32_t sb_cnt; // starts of the tile being marked that lie at or before the start of the one after it
	// split query: the read-mask words that cover the reads of the tile being staged (a tile holds at most 1024 reads,
	// which begin anywhere in a word), fetched by the first staging threads together with the tile's words and handed
	// over through here
	__shared__ uint32_t rmask[36];
	const bool masked = AUX && a.read_mask != nullptr && grid;
	const uint32_t n_mw = masked ? (a.rg_reads + 62) / 32 : 0; // <= 34
	uint8_t* spaced_lds = dyn + tile_cap + a.sb_words * 4;
	const PartLds pl = part_carve(spaced_lds + seq_spaced_bytes(a.hp), out.P);
	part_init<kPartThreads>(pl, out.P);
	seq_setup_tables<NT, SPACED>(sh, a.hp, spaced_lds);
	if (tid == 0) {
		p2_done = 0;
		sb_cnt = 0;
	}
	for (uint32_t i = tid; i < a.sb_words; i += NT)
		sbm[i] = 0;
	uint32_t* const late0 = sd.late_buf + (uint64_t)blockIdx.x * 2 * sd.late_cap; // this workgroup's two late images

	uint32_t* words = static_cast<uint32_t*>(a.filter);
	const uint32_t ent_mask = bin_shift >= 32 ? 0xffffffffu : (1u << bin_shift) - 1;
	// a power-of-two filter taken whole: bin and entry are bit fields of the hash itself (no 64-bit `& mask` first)
	const uint32_t bin_mask = (uint32_t)(a.mod.mask >> bin_shift);
	const uint32_t ent_mask_p2 = ent_mask & (uint32_t)a.mod.mask; // a filter smaller than one bin
	// (a power-of-two filter taken whole never has bins of whole segments, and must not read sd.bin_* here: with
	// part_bin_base for every variant the benchmark's query kernel spilled 16 bytes per lane and ran 44 instead of
	// 41 ms, the ragged insert 55.6 instead of 46.5 -- tools/kres.py after every change to this file)
	auto ovf = [&](uint32_t b, uint32_t v) {
		part_direct<QUERY>(words, sd, POW2 && !WINDOW ? ((uint64_t)b << bin_shift) | v : part_bin_base(sd, b, bin_shift) + v);
	};
	uint32_t my_valid = 0;

	const uint64_t t_begin = a.first_tile + (uint64_t)blockIdx.x * a.tiles_per_block;
	uint64_t t_end = t_begin + a.tiles_per_block;
	if (t_end > a.first_tile + a.n_tiles)
		t_end = a.first_tile + a.n_tiles;
	const uint32_t L = a.layout.starts ? 0 : a.layout.read_len;
	const uint32_t tile_step = L && !grid ? (uint32_t)(kTile % L) : 0;
	uint32_t tile_off = 0; // offset of the NEXT tile to be staged inside its read (plain tiles, uniform reads)
	if (L && t_begin < t_end && !grid)
		tile_off = (uint32_t)((t_begin * (uint64_t)kTile) % L);
	// grid: the lane's 8 window starts are starts 8m .. 8m+7 of read rd of the tile (the same in every tile);
	// lanes beyond the grid look at the zeroed guard behind the last read and find no k-mer there
	uint32_t grid_li0 = 0;
	uint32_t* grid_bm = nullptr; // two bitmaps of the tile's window starts, used by turns
	uint32_t grid_bm_words = 0;
	const bool want_bits = a.valid_bits || a.hit_bits;
	if (grid) {
		const uint32_t rd = tid / a.rg_gpr, m = tid - rd * a.rg_gpr;
		const bool in_grid = rd < a.rg_reads;
		grid_li0 = in_grid ? rd * a.rg_lpad + 8 * m : a.rg_reads * a.rg_lpad;
		grid_bm_words = ((tile_bytes / 8 + 15) / 16) * 4;
		grid_bm = reinterpret_cast<uint32_t*>(tile + tile_cap) - 2 * grid_bm_words;
		for (uint32_t i = tid; i < tile_cap / 4; i += NT) // pads, guard and bitmaps start as zeros
			reinterpret_cast<uint32_t*>(tile)[i] = 0;
	}
	const uint32_t span = grid ? tile_bytes : (uint32_t)kTile + k - 1; // bytes a tile stages

	// staging is the Y waves' job: 6 words per thread, requested at the start of the tile's second round (the Y
	// waves have no hashing to do there) and converted behind that round's first barrier
	StageRaw<kStageKW> raw;
	auto stage_request = [&](uint64_t t) { seq_stage_load<NY, kStageKW>(raw, KARG(seq), KARG(len), k, t * (uint64_t)tile_bytes, span, ytid); };
	auto stage_convert = [&](uint64_t t) {
		const uint64_t g0 = t * (uint64_t)tile_bytes;
		if (grid)
			seq_stage_convert_grid<NY, kStageKW>(raw, tile, sh, KARG(seq), KARG(len), L, a.rg_lpad, tile_bytes, g0, ytid,
			                                     masked ? rmask : nullptr, (uint32_t)(t * a.rg_reads) & 31u);
		else
			seq_stage_convert<NY, kStageKW, false, false, true>(raw, tile, tile_cap, sh, KARG(seq), KARG(len), a.layout, k, g0, tile_off,
			                                                    span, ytid, sb && t != t_begin ? sbm : nullptr);
		tile_off = seq_next_tile_off(tile_off, tile_step, L);
	};
	// Ragged layout with a start bitmap: the staging waves mark the sequence starts of tile T in `sbm` one segment
	// before they convert it (round 1 of tile T - 1: the round's first barrier lies in between), and the conversion
	// clears the "good" flags from the bitmap -- no barrier of its own, no search: the index of the first start behind
	// a tile's first byte runs along (s_base; the marking counts the starts up to the next tile's first byte).  The
	// first plan marked the starts after the staging, between two more barriers per tile, behind a binary search of
	// starts[] by one thread: pass A took 65 ms per 6x10^9 k-mers of 100..200-base sequences instead of 47.
	uint64_t s_base = 0;
	// the first candidate of every staging thread is requested together with the tile's words (before phase 1's stores)
	auto starts_request = [&](uint64_t& p_mine, uint64_t& p_last) {
		const uint64_t s = s_base + (uint32_t)ytid, sl = s_base + (uint32_t)(NY - 1);
		const uint64_t* const starts = KARG_LAYOUT(starts);
		const uint64_t n_seqs = KARG_LAYOUT(n_seqs);
		p_mine = s <= n_seqs ? starts[s] : ~0ull;
		p_last = sl <= n_seqs ? starts[sl] : ~0ull; // (one address per wave)
	};
	auto mark_ahead = [&](uint64_t T, uint64_t p, const uint64_t p_last) {
		const uint64_t g0 = T * (uint64_t)tile_bytes;
		const uint64_t len = KARG(len);
		uint64_t need = len > g0 ? len - g0 : 0;
		if (need > span)
			need = span;
		const uint64_t end = g0 + need, nxt = g0 + tile_bytes;
		const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(KARG(seq) + g0) & 3);
		uint32_t cnt = 0;
		uint64_t s = s_base + (uint32_t)ytid, sl = s_base + (uint32_t)(NY - 1), pl = p_last;
		for (;;) {
			const bool in = p < end;
			if (in && p > g0) {
				const uint32_t li = (uint32_t)(p - g0) + mis;
				atomicOr(&sbm[li >> 5], 1u << (li & 31));
			}
			cnt += (uint32_t)__popcll(__ballot(in && p <= nxt));
			// more than NY starts in one tile (sequences shorter than 16 bytes): every thread looks NY further on
			// (the test is the same in all lanes: the wave stays together for the ballot)
			if (!(pl < end))
				break;
			s += NY;
			sl += NY;
			p = s <= KARG_LAYOUT(n_seqs) ? KARG_LAYOUT(starts)[s] : ~0ull;
			pl = sl <= KARG_LAYOUT(n_seqs) ? KARG_LAYOUT(starts)[sl] : ~0ull;
		}
		if ((ytid & 63) == 0 && cnt)
			atomicAdd(&sb_cnt, cnt);
	};
	// ragged layout: sequence starts inside the freshly staged tile t (all threads; two barriers)
	auto mark_starts = [&](uint64_t t) {
		const uint64_t g0 = t * (uint64_t)tile_bytes;
		uint64_t need = a.len > g0 ? a.len - g0 : 0;
		if (need > span)
			need = span;
		seq_stage_mark_starts<NT>(tile, sh, a.layout, g0, need, (uint32_t)(reinterpret_cast<uintptr_t>(a.seq + g0) & 3));
		__syncthreads();
	};

	if (tid < n_mw && t_begin < t_end)
		rmask[tid] = a.read_mask[((t_begin * a.rg_reads) >> 5) + tid];
	__syncthreads(); // tables and partition state ready (and the grid's zeroed image)
	if (t_begin < t_end) {
		if (isY) {
			stage_request(t_begin);
			stage_convert(t_begin);
		}
		__syncthreads();
		if (a.layout.starts) {
			mark_starts(t_begin);
			if (sb && tid == 0) {
				// where the marking of the second tile begins: the first start behind that tile's first byte
				const uint64_t g1 = (t_begin + 1) * (uint64_t)tile_bytes;
				uint64_t lo = 0, hi = a.layout.n_seqs + 1;
				while (lo < hi) {
					const uint64_t mid = (lo + hi) >> 1;
					if (a.layout.starts[mid] > g1)
						hi = mid;
					else
						lo = mid + 1;
				}
				sh.start_lo = lo; // (read two barriers further on)
			}
		}
	}

	LaneState st;
	uint32_t bin[E], val[E];
	uint32_t vmask = 0, live = 0;
	uint64_t t = t_begin;
	uint32_t li0 = 0;
	auto on_window = [&](int w, bool ok, const WinHash<SPACED>& wh) {
		vmask |= (uint32_t)ok << w;
		if (w == kPartW - 1 && grid && want_bits) {
			// the lane's byte of the window bitmap sits at bit 8m of its read: collected in LDS (this tile's
			// bitmap; complete behind the barrier of the tile's second round) and written out whole afterwards.
			// Its bit index rd * L + 8m is worked out from the lane's LDS offset rd * lpad + 8m each time (a multiply
			// and a multiply-add) instead of being held: one more live register is one more spill in this kernel,
			// and a scratch reload here would wait for the flush stores (vector memory retires in order)
			uint32_t lo = li0;
			asm volatile("" : "+v"(lo));
			const uint32_t rd = __umulhi(lo, a.rg_lpad_inv);
			if (rd < a.rg_reads) {
				const uint32_t gb = lo - rd * (a.rg_lpad - L);
				uint32_t* bm = grid_bm + ((t - t_begin) & 1) * grid_bm_words;
				const uint32_t sh5 = gb & 31;
				if (vmask << sh5)
					atomicOr(&bm[gb >> 5], vmask << sh5);
				if (sh5 > 24 && (vmask >> (32 - sh5))) // (never beyond the bitmap: bits past a read's last window are 0)
					atomicOr(&bm[(gb >> 5) + 1], vmask >> (32 - sh5));
			}
		}
		const int w4 = w % kPartHalf;
		if (w4 == 0)
			live = 0;
		if (!WINDOW)
			live |= (uint32_t)ok << w4;
		// (no power of two: the form of the reduction is picked once per window, not once per probe)
		auto probes = [&](auto&& position) {
			auto place = [&](auto&& bin_of) {
#pragma unroll
				for (int i = 0; i < H; ++i) {
					if (POW2 && !WINDOW) {
						const uint64_t hv = wh.at(i);
						bin[w4 * H + i] = (uint32_t)(hv >> bin_shift) & bin_mask;
						val[w4 * H + i] = (uint32_t)hv & ent_mask_p2;
						continue;
					}
					uint64_t p = position(wh.at(i));
					if (WINDOW) {
						p -= a.mod.shard_lo;
						live |= (uint32_t)(ok && p < a.mod.shard_len) << (w4 * H + i);
					}
					bin_of(p, bin[w4 * H + i], val[w4 * H + i]);
				}
			};
			// (bins of sd.bin_wseg segments where the local array's segment count is no power of two: part_bin_of)
			if (!(POW2 && !WINDOW) && sd.bin_wseg)
				place([&](uint64_t p, uint32_t& b, uint32_t& v) { part_bin_of(sd, p, b, v); });
			else
				place([&](uint64_t p, uint32_t& b, uint32_t& v) {
					b = (uint32_t)(p >> bin_shift);
					v = (uint32_t)p & ent_mask;
				});
		};
		if (POW2)
			probes([&](uint64_t hv) { return hv & a.mod.mask; });
		else if (mod_small_magic(a.mod))
			probes([&](uint64_t hv) { return reduce_mod_small(hv, a.mod); });
		else
			probes([&](uint64_t hv) { return reduce_mod_big(hv, a.mod); });
	};
	auto hash_lo = [&]() { seq_lane_range<SPACED, kPartW, H, 0, kPartHalf>(tile, sh, a.hp, spaced_lds, li0, st, on_window); };
	auto hash_hi = [&]() { seq_lane_range<SPACED, kPartW, H, kPartHalf, kPartW>(tile, sh, a.hp, spaced_lds, li0, st, on_window); };
	// the Y waves' share of phase 3: the round's late entries (partition_core.hpp part_round_p1_late), fetched before
	// the round's second barrier -- as soon as the owners of the bins have said how many there are -- and written into
	// the rings behind it (the slots they go to are being flushed until then).  One bin per Y thread to look at, the
	// entries themselves spread over the lanes of the wave.
	uint32_t late_v = 0, late_dst = 0, late_n = 0;
	auto late_fetch = [&](uint32_t par, uint32_t rounds_done) {
		uint32_t yt = (uint32_t)ytid; // laundered: nothing derived from it is kept (and spilled) across the tile loop
		asm volatile("" : "+v"(yt));
		// the owners write their words first thing behind the round's first barrier; this wave has hashed or staged
		// since, so the wait is there for correctness, not for time.  It is NOT bounded: going on without the owners'
		// words would drop or misplace late entries silently.  It always ends: the waves of a workgroup are co-resident,
		// and every owner wave reaches its atomicAdd on p2_done straight behind that barrier (part_round_p2), without
		// waiting for anything a Y wave does
		const uint32_t target = (rounds_done + 1) * (uint32_t)kOvOwners;
		while (__hip_atomic_load(&p2_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
			__builtin_amdgcn_s_sleep(1);
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		const uint32_t w = yt < out.P ? pl.fl[yt] : 0u;
		uint32_t src;
		late_n = part_late_assign(pl, yt & ~63u, w, 0, src, late_dst);
		if ((yt & 63u) < late_n)
			late_v = part_late_load(late0 + (uint64_t)par * sd.late_cap, src);
	};
	auto late_apply = [&](uint32_t par) {
		uint32_t yt = (uint32_t)ytid;
		asm volatile("" : "+v"(yt));
		if ((yt & 63u) < late_n)
			pl.stage[late_dst] = late_v;
		for (uint32_t skip = 64; skip < late_n; skip += 64) { // more than 64 late entries in the wave's 64 bins: rare
			const uint32_t w = yt < out.P ? pl.fl[yt] : 0u;
			uint32_t src, dst;
			part_late_assign(pl, yt & ~63u, w, skip, src, dst);
			if ((yt & 63u) + skip < late_n)
				pl.stage[dst] = part_late_load(late0 + (uint64_t)par * sd.late_cap, src);
		}
	};

#ifdef BTLBF_PHASE_STAMPS
	// diagnostic build: cycles thread 0 (an X wave) and thread 512 (a Y wave) spend working / waiting in each of
	// the four segments of a tile (slots 0..7: X, 8..15: Y; even = work up to the barrier, odd = wait in it)
	uint64_t ov_acc[12] = {0}, ov_last = __builtin_readcyclecounter();
	const bool ov_me = (tid & 511) == 0;
#define OV_STAMP(i)                                             \
	do {                                                        \
		if (ov_me) {                                            \
			const uint64_t t__ = __builtin_readcyclecounter();  \
			ov_acc[i] += t__ - ov_last;                         \
			ov_last = t__;                                      \
		}                                                       \
	} while (0)
#else
#define OV_STAMP(i)
#endif
	for (; t < t_end; ++t) {
		const uint64_t g0 = t * (uint64_t)tile_bytes;
		li0 = grid ? grid_li0 : tid * kPartW + (uint32_t)(reinterpret_cast<uintptr_t>(KARG(seq) + g0) & 3);
		vmask = 0;
		if (sb && isY) {
			// the start bitmap was read by the conversion of this tile (two barriers ago) and is marked again in this
			// tile's second round (two barriers on)
			uint32_t yt = (uint32_t)ytid;
			asm volatile("" : "+v"(yt));
			for (uint32_t i = yt; i < a.sb_words; i += NY)
				sbm[i] = 0;
			if (yt == 0)
				sb_cnt = 0;
		}
		uint64_t sp_mine = ~0ull, sp_last = ~0ull;
		uint32_t mask_word = 0;
		// (unrolled: with one rolled copy of the round the register allocator spills inside the loop)
#pragma unroll
		for (uint32_t r = 0; r < 2; ++r) {
			// ---- segment a: round 0 = windows 0..3 of everybody; round 1 = windows 4..7, the Y waves' are hashed already
			if (r == 0) {
				hash_lo();
			} else if (!isY) {
				hash_hi();
			} else if (t + 1 < t_end) {
				// the next tile's words are requested BEFORE phase 1: behind it the compiler's wait for "no vector
				// memory operation pending" in front of the loads would also wait for the stores of late entries,
				// and the X waves would find the Y waves still there when they arrive at the barrier
				stage_request(t + 1);
				if ((uint32_t)ytid < n_mw)
					mask_word = KARG(read_mask)[(((t + 1) * a.rg_reads) >> 5) + (uint32_t)ytid];
				if (sb) {
					if (t == t_begin) { // (uniform: kept in scalar registers)
						const uint64_t v = sh.start_lo;
						s_base = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)v) |
						         ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32);
					}
					starts_request(sp_mine, sp_last);
				}
			}
			OV_STAMP(9);
			part_round_p1_late<E, WINDOW ? 1 : H, WINDOW>(pl, bin, val, live, late0 + (uint64_t)r * sd.late_cap, ovf);
			if (sb && r == 1 && isY && t + 1 < t_end)
				mark_ahead(t + 1, sp_mine, sp_last);
			if (r == 1 && isY && t + 1 < t_end && (uint32_t)ytid < n_mw)
				rmask[ytid] = mask_word; // read by the conversion behind the barrier
			OV_STAMP(0);
			__syncthreads(); // (round 1: nobody reads this tile's image any more)
			OV_STAMP(1);
			// ---- segment b: the X waves flush; the Y waves hash ahead (round 0) or stage the next tile (round 1)
			if (!isY) {
				part_round_p2<NT, kOvOwners>(pl, out, 0, blockIdx.x, ovf, &p2_done);
			} else {
				if (r == 0) {
					hash_hi(); // ahead of the X waves: this wave's entries of round 1
				} else if (t + 1 < t_end) {
					stage_convert(t + 1);
					if (sb)
						s_base += __builtin_amdgcn_readfirstlane(sb_cnt); // complete since the barrier; zeroed at the top of the next tile
				}
				late_fetch(r, (uint32_t)(t - t_begin) * 2 + r);
			}
			OV_STAMP(2);
			__syncthreads();
			OV_STAMP(3);
			if (isY)
				late_apply(r);
			OV_STAMP(8);
		}

		uint32_t ltid = tid; // laundered, as above: the addresses of the write-out below are per-tile work
		asm volatile("" : "+v"(ltid));
		if (want_bits && grid) {
			// this tile's bitmap is complete (two barriers since the last OR); the other one -- written out a
			// tile ago -- is cleared for the next tile
			uint32_t* bm = grid_bm + ((t - t_begin) & 1) * grid_bm_words;
			uint32_t* other = grid_bm + (((t - t_begin) & 1) ^ 1) * grid_bm_words;
			const uint8_t* bm8 = reinterpret_cast<const uint8_t*>(bm);
			const uint64_t ob0 = g0 >> 3; // tiles are whole bytes of the bitmaps
			// the buffer's last tile also writes the (zero) bytes up to the end of the bitmaps' last 64-bit word
			const uint64_t len = KARG(len), out_bytes = ((len + 63) / 64) * 8;
			const uint64_t left = out_bytes > ob0 ? out_bytes - ob0 : 0;
			const uint32_t n_out = g0 + tile_bytes >= len || left < tile_bytes / 8 ? (uint32_t)left : tile_bytes / 8;
			uint8_t* const vb = KARG(valid_bits);
			uint8_t* const hb = KARG(hit_bits);
			for (uint32_t i = ltid; i < n_out; i += NT) {
				const uint8_t v = i < tile_bytes / 8 ? bm8[i] : (uint8_t)0;
				if (vb)
					vb[ob0 + i] = v;
				if (hb)
					hb[ob0 + i] = v; // a query starts from "every clean window hits"
			}
			for (uint32_t i = ltid; i < grid_bm_words; i += NT)
				other[i] = 0;
		} else if (want_bits) {
			// one byte of the per-window bitmaps per lane
			static_assert(kPartW == 8, "one bitmap byte per lane");
			const uint64_t ob = (g0 >> 3) + ltid;
			if (ob < ((KARG(len) + 63) / 64) * 8) {
				uint8_t* const vb = KARG(valid_bits);
				uint8_t* const hb = KARG(hit_bits);
				if (vb)
					vb[ob] = (uint8_t)vmask;
				if (hb)
					hb[ob] = (uint8_t)vmask; // a query starts from "every clean window hits"
			}
		}
		my_valid += __popc(vmask);
		if (a.layout.starts && !sb && t + 1 < t_end)
			mark_starts(t + 1);
	}
#ifdef BTLBF_PHASE_STAMPS
	if (ov_me)
		for (int i = 0; i < 12; ++i)
			atomicAdd((unsigned long long*)&g_stamp_out[(isY ? 12 : 0) + i], (unsigned long long)ov_acc[i]);
#endif
#undef OV_STAMP
	part_finish<kPartThreads>(pl, out, 0, blockIdx.x, ovf);
	if (a.counts) {
		const uint32_t wv = wave_sum(my_valid);
		if ((tid & 63) == 0 && wv)
			atomicAdd(reinterpret_cast<unsigned long long*>(a.counts), (unsigned long long)wv);
	}
}

template <int H, bool Q, bool SMALL>
static hipError_t launch_hash_h(const SeqArgs& a, const PartOut& out, uint32_t bin_shift, const PartSide& sd,
                                size_t dyn, hipStream_t s)
{
	const bool pow2 = a.mod.pow2 != 0, spaced = a.hp.n_seeds > 0;
	const bool window = a.mod.shard_lo != 0 || a.mod.shard_len != a.mod.size;
	constexpr int NT = SMALL ? kPartThreadsS : kPartThreads;
	// the overlapped schedule: 1024 threads and bins that eight waves can own (BTLBF_PART_OVERLAP=0: the plain one)
	static const bool ov_off = [] {
		const char* e = getenv("BTLBF_PART_OVERLAP");
		return e && !strcmp(e, "0");
	}();
	// (not for spaced seeds: their hashing keeps h values per window in registers, the overlapped kernel spills with
	// them and ran 16 % slower than the plain one at BASELINE config 5; and not for more than four hashes per k-mer:
	// a round's 4h entries per lane no longer fit the registers beside the hash state -- h = 5: 34.0 / 41.6 ms insert /
	// query per 3.6x10^9 k-mers against 34.3 / 35.1 with the plain kernel, h = 6: 52.2 / 62.2 against 41.7 / 42.6,
	// h = 8: 230 / 257 against 97 / 169; h = 3: 19.5 / 20.3 against 21.3 / 21.8)
	constexpr bool kOvH = H <= 4;
	// (spaced seeds keep the plain schedule: the overlapped kernel runs them, bit-identically, but 9 % slower -- 99.7
	// against 91.3 ms per 6x10^9 k-mers at BASELINE config 5 -- its registers are full without the seeds' running values)
#ifdef BTLBF_OV_SPACED
	constexpr bool kOvSpaced = true;
#else
	constexpr bool kOvSpaced = false;
#endif
	const bool aux = a.sb_words != 0 || (a.read_mask != nullptr && a.rg_reads != 0); // (see part_hash_ov_kernel)
	const bool overlapped = !SMALL && kOvH && (!spaced || kOvSpaced) && out.P <= 64u * kOvOwners && !ov_off && sd.late_buf != nullptr &&
	                        sd.late_cap >= kStageEntries;
#define BTLBF_PLAUNCH(P, S, W)                                                                                      \
	do {                                                                                                            \
		if constexpr (!SMALL && (!S || kOvSpaced) && kOvH) {                                                        \
			if (overlapped && aux) {                                                                                \
				hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&part_hash_ov_kernel<H, P, S, Q, W, true>), \
				                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);            \
				if (e != hipSuccess)                                                                                \
					return e;                                                                                       \
				hipLaunchKernelGGL((part_hash_ov_kernel<H, P, S, Q, W, true>), dim3(out.regions), dim3(NT), dyn, s, a, out, \
				                   bin_shift, sd);                                                                  \
				break;                                                                                              \
			}                                                                                                       \
			if (overlapped) {                                                                                       \
				hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&part_hash_ov_kernel<H, P, S, Q, W, false>), \
				                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);            \
				if (e != hipSuccess)                                                                                \
					return e;                                                                                       \
				hipLaunchKernelGGL((part_hash_ov_kernel<H, P, S, Q, W, false>), dim3(out.regions), dim3(NT), dyn, s, a, out, \
				                   bin_shift, sd);                                                                  \
				break;                                                                                              \
			}                                                                                                       \
		}                                                                                                           \
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&part_hash_kernel<H, P, S, Q, W, SMALL>),    \
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);                    \
		if (e != hipSuccess)                                                                                        \
			return e;                                                                                               \
		hipLaunchKernelGGL((part_hash_kernel<H, P, S, Q, W, SMALL>), dim3(out.regions), dim3(NT), dyn, s, a, out,    \
		                   bin_shift, sd);                                                                          \
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
                                                       const PartSide& sd, size_t dyn, int query, int small,
                                                       hipStream_t s)
{
	if (small)
		return query ? launch_hash_h<BTLBF_PART_H, true, true>(a, out, bin_shift, sd, dyn, s)
		             : launch_hash_h<BTLBF_PART_H, false, true>(a, out, bin_shift, sd, dyn, s);
	return query ? launch_hash_h<BTLBF_PART_H, true, false>(a, out, bin_shift, sd, dyn, s)
	             : launch_hash_h<BTLBF_PART_H, false, false>(a, out, bin_shift, sd, dyn, s);
}

#if defined(BTLBF_PHASE_STAMPS) && BTLBF_PART_H == 4
extern "C" void btlbf_debug_stamps(uint64_t* out16)
{
	(void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamp_out), sizeof(uint64_t) * 32);
	uint64_t z[32] = {0};
	(void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_out), z, sizeof z);
}
#endif

} // namespace btlbf

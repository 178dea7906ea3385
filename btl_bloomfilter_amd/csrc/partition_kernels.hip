// csrc/partition_kernels.hip -- partitioned insert / contains: turn h random HBM requests per k-mer
// into streamed traffic plus LDS atomics.
//
// Why: the direct kernels are pinned to the memory system's random-request rate (~22 G memory-side
// atomics/s, ~50 G gathers/s on MI355X).  Bit OR is order-free, so a batch of
// probe positions may be applied in any order -- in particular grouped by the 64 KiB (or 128 KiB)
// filter SEGMENT they fall into, with that segment held in LDS.  The result is bit-identical.
//
//   pass A  part_hash_kernel   fused ntHash (seq_core.hpp) + radix partition of the positions by
//                              their top bits into <= 1024 level-0 bins
//   pass B  part_split_kernel  splits every bin of one level into <= 1024 sub-bins (run once, or
//                              twice when the data arrives pre-binned from other GPUs)
//   pass C  part_apply_kernel  one workgroup per segment: load the segment into LDS, ds_or (insert)
//                              or test (contains) every entry, store the segment back (insert)
//
// Bins are written as 128-byte CHUNKS (32 uint32 entries; an entry is the position's offset inside
// its bin).  A workgroup stages entries per bin in an LDS ring and writes a chunk only when it is
// full, so every global write is one aligned 128-byte line.  Every workgroup writes into its OWN
// region of every bin (region = (bin, writer)), so chunk slots are handed out from an LDS counter:
// no global atomics in passes A and B.  The number of ENTRIES of each region is published at kernel
// end (the tail chunk is partial).  Entries that do not fit (a region over capacity, or a bin that
// receives more than about two rings' worth inside one round) take the overflow path: applied to
// the filter directly (single GPU) or appended to a spill list of global positions (multi-GPU
// routing) -- never dropped.
#include "seq_core.hpp"

namespace btlbf {

// 1024-thread workgroups (4 waves per SIMD; one workgroup per CU because the staging rings fill the
// LDS) with 4 windows per lane in pass A and 16 entries per lane in pass B: <= 128 VGPRs
static constexpr int kPartThreads = 1024;
static constexpr int kPartW = 8;                        // windows per lane and tile in pass A ...
static constexpr int kPartHalf = 4;                     // ... partitioned in two rounds of 4 (the rings hold one)
static constexpr int kPartTile = kPartThreads * kPartW; // windows per tile of pass A
static constexpr int kApplyThreads = 512;
static constexpr uint32_t kChunk = 32;           // entries per chunk
static constexpr uint32_t kNoBin = 0xffffffffu;  // empty entry slot of a lane (registers only)
static constexpr uint32_t kStageEntries = 32768; // LDS staging: 128 KiB of uint32 entries
static constexpr uint32_t kPartLdsBudget = 160 * 1024 - 1024; // dynamic LDS a workgroup may ask for

// LDS image of the staged partitioner (carved from dynamic LDS by the kernels)
struct PartLds {
	uint32_t* stage;   // [P][SC] ring per bin, SC = kStageEntries / pow2ceil(P)
	uint32_t* pt;      // [P] low 16 bits: entries in the ring (+ offered this round); high 16: ring write position
	uint32_t* fl;      // [P] entries flushed from the bin this round (multiple of 32)
	uint32_t* written; // [P] chunks written to this workgroup's region of the bin
	uint32_t* fout;    // [1024] flush items, a private slice per wave: output chunk index ...
	uint16_t* flist;   // [1024] ... and bin | ring chunk << 10
	uint32_t sc_shift; // log2(SC)
};

__host__ __device__ inline uint32_t part_pow2ceil(uint32_t x)
{
	uint32_t p = 1;
	while (p < x)
		p <<= 1;
	return p;
}

__host__ __device__ inline uint32_t part_lds_bytes(uint32_t P)
{
	return kStageEntries * 4 + 3 * P * 4 + 1024 * 4 + 1024 * 2 + 16;
}

__device__ __forceinline__ PartLds part_carve(uint8_t* base, uint32_t P)
{
	PartLds l;
	l.stage = reinterpret_cast<uint32_t*>(base);
	l.pt = l.stage + kStageEntries;
	l.fl = l.pt + P;
	l.written = l.fl + P;
	l.fout = l.written + P;
	l.flist = reinterpret_cast<uint16_t*>(l.fout + 1024);
	uint32_t pc = part_pow2ceil(P < 32 ? 32 : P), sh = 0;
	while ((kStageEntries >> sh) > pc)
		++sh; // kStageEntries / 2^sh == pc  ->  SC = 2^sh
	l.sc_shift = sh;
	return l;
}

template <int NT>
__device__ __forceinline__ void part_init(const PartLds& l, uint32_t P)
{
	for (uint32_t b = threadIdx.x; b < P; b += NT) {
		l.pt[b] = 0;
		l.fl[b] = 0;
		l.written[b] = 0;
	}
}

#ifdef BTLBF_PHASE_STAMPS
#define STAMP(i)                                               \
	do {                                                       \
		if (threadIdx.x == 0) {                                \
			const uint64_t t__ = __builtin_readcyclecounter(); \
			g_stamp[i] += t__ - g_last;                        \
			g_last = t__;                                      \
		}                                                      \
	} while (0)
static __device__ uint64_t g_stamp_out[16];
#define STAMP_DECL uint64_t g_stamp[16] = {0}, g_last = __builtin_readcyclecounter()
#define STAMP_FLUSH                                                                                  \
	do {                                                                                             \
		if (threadIdx.x == 0)                                                                        \
			for (int i__ = 0; i__ < 16; ++i__)                                                       \
				atomicAdd((unsigned long long*)&g_stamp_out[i__], (unsigned long long)g_stamp[i__]); \
	} while (0)
#define STAMP_ARGS , uint64_t (&g_stamp)[16], uint64_t& g_last
#define STAMP_PASS , g_stamp, g_last
#else
#define STAMP(i)
#define STAMP_DECL
#define STAMP_FLUSH
#define STAMP_ARGS
#define STAMP_PASS
#endif

// One round: every thread contributes E entries (bin[e] == kNoBin marks an empty slot) to the bins
// [0, o.P) of this workgroup's output block; block-local bin b is global bin bin0 + b.
// Region `region` of global bin g is chunks [(g*o.regions + region)*o.cap, +o.cap) of o.ent
// (32-bit chunk indices: a pass's output holds fewer than 2^32 chunks).
// `ovf(bin, val)` takes the entries that cannot be staged.
//
// pt[b] packs (ring write position << 16 | entries in the ring).  Three phases, two barriers:
//  1. ONE returning LDS atomic per entry adds 0x10001: the old value is the entry's ring slot (high
//     half) and how many entries are ahead of it (low half; fewer than SC means it fits and is written
//     now; otherwise it is "late").
//  2. the lane that owns bin b (b = lane index) sees how much arrived, copies the bin's full 32-entry
//     chunks from the ring to this workgroup's region (one aligned 128-byte line each) and already
//     writes the bin's state for the next round: what the late entries will do is determined by
//     their old values alone (those still beyond the ring after the flush overflow and give their
//     slots back), so nothing has to wait for them.
//  3. late entries move into the ring space the flush freed, or overflow.
// No barrier is needed after phase 3: the next round's phase 1 only touches pt (final since phase 2)
// and ring slots behind the late ones; its phase 2 comes after its own barrier.
template <int NT, int E, class OVF>
__device__ __forceinline__ void part_round(const PartLds& l, const PartOut& o, uint32_t bin0, uint32_t region,
                                           const uint32_t (&bin)[E], const uint32_t (&val)[E], OVF&& ovf STAMP_ARGS)
{
	const uint32_t tid = threadIdx.x;
	const uint32_t P = o.P;
	const uint32_t SC = 1u << l.sc_shift, ring = SC - 1;
	uint32_t old[E];
	uint32_t late = 0; // bit e: entry e found no room before this round's flush
	static_assert(E <= 32, "one flag bit per entry");
	// all E atomics of the lane are issued back to back (independent), then consumed
#pragma unroll
	for (int e = 0; e < E; ++e) {
		old[e] = 0xffffu; // "no room": empty slots fall through both tests below
		if (bin[e] != kNoBin)
			old[e] = atomicAdd(&l.pt[bin[e]], 0x10001u);
	}
#pragma unroll
	for (int e = 0; e < E; ++e) {
		if ((old[e] & 0xffffu) < SC)
			l.stage[(bin[e] << l.sc_shift) + ((old[e] >> 16) & ring)] = val[e];
		else if (bin[e] != kNoBin)
			late |= 1u << e;
	}
	__syncthreads();
	STAMP(4);
	{
		// bins are owned by lanes (P <= NT): wave v owns bins [64v, 64v+64) and flushes them itself,
		// through its private slice of the flush list -- no workgroup barrier in between
		const uint32_t b = tid, lane = tid & 63;
		uint32_t nfl = 0, hc = 0, o0 = 0, w0 = 0;
		if (b < P) {
			const uint32_t w = l.pt[b];
			const uint32_t occ = w & 0xffffu;           // ring content + everything offered this round
			const uint32_t avail = occ < SC ? occ : SC; // entries that really sit in the ring
			nfl = avail >> 5;
			const uint32_t f = nfl << 5;
			// state for the next round (see above)
			const uint32_t tot = occ - f;
			const uint32_t nocc = tot < SC ? tot : SC;
			l.pt[b] = (((w >> 16) - (tot - nocc)) << 16) | nocc;
			l.fl[b] = f;
			if (nfl) {
				w0 = l.written[b];
				l.written[b] = w0 + nfl;
				// read position of the ring: both halves of pt grew by the same amount this round
				hc = (((w >> 16) - occ) & ring) >> 5;
				o0 = ((bin0 + b) * o.regions + region) * o.cap;
			}
		}
		// exclusive prefix sum of nfl over the wave -> slots in the wave's slice (2*SC items)
		uint32_t incl = nfl;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t t = __shfl_up(incl, d, 64);
			if (lane >= (uint32_t)d)
				incl += t;
		}
		const uint32_t total = __shfl(incl, 63, 64);
		const uint32_t slice = (tid >> 6) * (2u << l.sc_shift);
		for (uint32_t c = 0; c < nfl; ++c) {
			const uint32_t j = slice + incl - nfl + c;
			l.flist[j] = (uint16_t)(b | (((hc + c) & (ring >> 5)) << 10));
			l.fout[j] = w0 + c < o.cap ? o0 + w0 + c : 0xffffffffu;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		// 8 lanes per chunk, 16 bytes per lane -> one aligned 128-byte line per chunk
		const uint32_t l8 = lane & 7;
		for (uint32_t j = lane >> 3; j < total; j += 8) {
			const uint32_t it = l.flist[slice + j], oc = l.fout[slice + j];
			const uint32_t fb = it & 1023, rc = it >> 10;
			const uint4 v = *reinterpret_cast<const uint4*>(&l.stage[(fb << l.sc_shift) + (rc << 5) + l8 * 4]);
			if (oc != 0xffffffffu) {
				*reinterpret_cast<uint4*>(&o.ent[(uint64_t)oc * kChunk + l8 * 4]) = v;
			} else {
				ovf(fb, v.x);
				ovf(fb, v.y);
				ovf(fb, v.z);
				ovf(fb, v.w);
			}
		}
	}
	__syncthreads();
	STAMP(6);
	// entries that did not fit before the flush: into the freed ring space, else overflow
	if (late) {
#pragma unroll
		for (int e = 0; e < E; ++e) {
			if ((late >> e) & 1) {
				if ((old[e] & 0xffffu) - l.fl[bin[e]] < SC)
					l.stage[(bin[e] << l.sc_shift) + ((old[e] >> 16) & ring)] = val[e];
				else
					ovf(bin[e], val[e]);
			}
		}
	}
	STAMP(7);
}

// flush whatever is staged and publish the ENTRY count of this workgroup's region of every bin
template <int NT, class OVF>
__device__ __forceinline__ void part_finish(const PartLds& l, const PartOut& o, uint32_t bin0, uint32_t region,
                                            OVF&& ovf)
{
	__syncthreads();
	const uint32_t tid = threadIdx.x, lane32 = tid & 31;
	const uint32_t ring = (1u << l.sc_shift) - 1;
	for (uint32_t b = tid >> 5; b < o.P; b += NT / 32) {
		const uint32_t w = l.pt[b], n = w & 0xffffu, hd = ((w >> 16) - n) & ring;
		const uint32_t w0 = l.written[b];
		const uint32_t full = w0 < o.cap ? w0 : o.cap; // chunks of this region that really hold data
		const uint32_t o0 = ((bin0 + b) * o.regions + region) * o.cap;
		uint32_t stored = 0;
		for (uint32_t c = 0; c * kChunk < n; ++c) {
			const uint32_t i = c * kChunk + lane32;
			const uint32_t v = i < n ? l.stage[(b << l.sc_shift) + ((hd + i) & ring)] : 0;
			if (w0 + c < o.cap) {
				o.ent[(uint64_t)(o0 + w0 + c) * kChunk + lane32] = v;
				stored = (c + 1) * kChunk < n ? (c + 1) * kChunk : n;
			} else if (i < n) {
				ovf(b, v);
			}
		}
		if (lane32 == 0)
			o.cnt[(bin0 + b) * o.regions + region] = full * kChunk + stored;
	}
}

// where the overflow entries of the routing passes go (multi-GPU): global positions
__device__ __forceinline__ void part_spill(const PartSide& sd, uint64_t pos)
{
	const unsigned long long i = atomicAdd(sd.spill_count, 1ull);
	if (i < sd.spill_cap)
		sd.spill_list[i] = pos;
}
// positions whose bit was found clear (partitioned contains)
__device__ __forceinline__ void part_report_fail(const PartSide& sd, uint64_t pos)
{
	const unsigned long long i = atomicAdd(sd.fail_count, 1ull);
	if (i < sd.fail_cap)
		sd.fail_list[i] = pos;
}
// overflow of an insert / contains pass: spill list when routing, else straight to the filter
template <bool QUERY>
__device__ __forceinline__ void part_direct(uint32_t* words, const PartSide& sd, uint64_t lp)
{
	if (sd.spill_count)
		part_spill(sd, sd.pos_base + lp);
	else if (!QUERY)
		bf_set(words, lp);
	else if (!((bf_word(words, lp) >> (lp & 31)) & 1u))
		part_report_fail(sd, sd.pos_base + lp);
}

// ---- pass A --------------------------------------------------------------------------------------
// bin = position >> bin_shift ; entry = position & ((1 << bin_shift) - 1); region = blockIdx.x
// (gridDim.x == out.regions).  `position` is local to a.mod's shard window (the whole filter in
// routing mode, where a.mod describes the GLOBAL filter).
template <int H, bool POW2, bool SPACED, bool QUERY>
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

	STAMP_DECL;
	for (uint64_t t = t_begin; t < t_end; ++t) {
		const uint64_t g0 = t * (uint64_t)kPartTile;
		STAMP(0);
		const uint32_t mis = seq_stage_tile<kPartThreads, kPartW>(tile, tile_cap, sh, a.seq, a.len, a.layout, k, g0, tile_off);
		tile_off = seq_next_tile_off(tile_off, tile_step, L);
		STAMP(1);

		// the lane hashes its 8 consecutive windows with ONE start-up; after every 4 windows the
		// 4*H entries collected so far go through a partition round (the rolling state stays in
		// registers across it)
		uint32_t bin[kPartHalf * H], val[kPartHalf * H];
		uint32_t vmask = 0;
		seq_lane_windows<SPACED, kPartW>(tile, sh, a.hp, spaced_lds, tid * kPartW + mis, [&](int w, bool ok, const WinHash<SPACED>& wh) {
			vmask |= (uint32_t)ok << w;
			const int w4 = w % kPartHalf;
#pragma unroll
			for (int i = 0; i < H; ++i) {
				// positions outside this GPU's window (a shard) are dropped; without sharding the
				// window is the whole filter and the test is always true
				const uint64_t p = reduce_mod<POW2>(wh.at(i), a.mod) - a.mod.shard_lo;
				const bool mine = ok && p < a.mod.shard_len;
				bin[w4 * H + i] = mine ? (uint32_t)(p >> bin_shift) : kNoBin;
				val[w4 * H + i] = (uint32_t)p & ent_mask;
			}
			if (w4 == kPartHalf - 1) {
				STAMP(2);
				part_round<kPartThreads, kPartHalf * H>(pl, out, 0, blockIdx.x, bin, val, ovf STAMP_PASS);
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

// region r of input bin b: the data may consist of several origin blocks (multi-GPU exchange)
__device__ __forceinline__ uint32_t part_in_region(const PartIn& in, uint32_t b, uint32_t r)
{
	return ((r / in.regions_per_block) * in.bins_per_block + b) * in.regions_per_block + r % in.regions_per_block;
}

// ---- pass B --------------------------------------------------------------------------------------
// workgroup (b, g): blockIdx.x = b * slices + g.  It consumes input regions g, g+slices, ... of input
// bin b and writes region g of every sub-bin b*P + sub, sub = entry >> sub_shift; the new entry is
// the low sub_shift bits.  in_shift = log2(positions per input bin), for the overflow path.
// Only input bins [first_bin, first_bin + gridDim.x/slices) are processed (a GROUP of bins: the output
// arrays hold one group at a time, which keeps the level-1 scratch small); output bins are numbered
// relative to the group.
template <bool QUERY>
__global__ __launch_bounds__(kPartThreads) void part_split_kernel(void* filter, const PartIn in, const PartOut out,
                                                                 const uint32_t slices, const uint32_t sub_shift,
                                                                 const uint32_t in_shift, const uint32_t first_bin,
                                                                 const PartSide sd)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
	const uint32_t tid = threadIdx.x;
	const uint32_t b = first_bin + blockIdx.x / slices, g = blockIdx.x % slices;
	const PartLds pl = part_carve(dyn, out.P);
	part_init<kPartThreads>(pl, out.P);
	uint32_t* words = static_cast<uint32_t*>(filter);
	const uint32_t sub_mask = (1u << sub_shift) - 1;
	const uint64_t bin_base = (uint64_t)b << in_shift;
	auto ovf = [&](uint32_t sub, uint32_t v) {
		part_direct<QUERY>(words, sd, bin_base | ((uint64_t)sub << sub_shift) | v);
	};
	const uint32_t bin0 = (b - first_bin) * out.P;
	const uint32_t n_regions_in = in.blocks * in.regions_per_block;
	constexpr int kVec = 4; // uint4 loads per thread per round -> 16 entries
	STAMP_DECL;
	__syncthreads();
	for (uint32_t r = g; r < n_regions_in; r += slices) {
		const uint32_t reg = part_in_region(in, b, r);
		uint32_t n = in.cnt[reg];
		if (n > in.cap * kChunk)
			n = in.cap * kChunk;
		const uint4* src = reinterpret_cast<const uint4*>(in.ent + (uint64_t)reg * in.cap * kChunk);
		const uint32_t n_vec = (n + 3) / 4;
		// software pipeline: the next round's loads are in flight while this round is partitioned
		uint4 nxt[kVec];
#pragma unroll
		for (int v = 0; v < kVec; ++v) {
			const uint32_t i = (uint32_t)v * kPartThreads + tid;
			nxt[v] = i < n_vec ? src[i] : make_uint4(0, 0, 0, 0);
		}
		for (uint32_t base = 0; base < n_vec; base += kPartThreads * kVec) {
			uint32_t bin[kVec * 4], val[kVec * 4];
#pragma unroll
			for (int v = 0; v < kVec; ++v) {
				const uint32_t e4[4] = {nxt[v].x, nxt[v].y, nxt[v].z, nxt[v].w};
				const uint32_t i0 = (base + (uint32_t)v * kPartThreads + tid) * 4;
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					bin[v * 4 + c] = i0 + c < n ? e4[c] >> sub_shift : kNoBin;
					val[v * 4 + c] = e4[c] & sub_mask;
				}
				const uint32_t i = base + kPartThreads * kVec + (uint32_t)v * kPartThreads + tid;
				nxt[v] = i < n_vec ? src[i] : make_uint4(0, 0, 0, 0);
			}
			part_round<kPartThreads, kVec * 4>(pl, out, bin0, g, bin, val, ovf STAMP_PASS);
		}
	}
	part_finish<kPartThreads>(pl, out, bin0, g, ovf);
}

// ---- pass C --------------------------------------------------------------------------------------
// one workgroup per segment (= input bin `seg`); pos_base + (seg << seg_shift | entry) is the
// position reported for failed tests
// Input bin blockIdx.x holds the entries of segment seg_first + blockIdx.x (group-relative numbering).
template <bool QUERY>
__global__ __launch_bounds__(kApplyThreads) void part_apply_kernel(uint8_t* filter, uint64_t local_bytes,
                                                                 uint32_t seg_shift, uint32_t seg_first,
                                                                 const PartIn in, const PartSide sd)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
	__shared__ uint32_t any;
	const uint32_t tid = threadIdx.x;
	const uint32_t ibin = blockIdx.x;
	const uint32_t seg = seg_first + blockIdx.x;
	const uint32_t n_regions = in.blocks * in.regions_per_block;
	if (tid == 0)
		any = 0;
	__syncthreads();
	uint32_t mine = 0;
	for (uint32_t r = tid; r < n_regions; r += kApplyThreads)
		mine |= in.cnt[part_in_region(in, ibin, r)];
	if (mine)
		any = 1;
	__syncthreads();
	if (!any)
		return; // untouched segment: no traffic at all
	const uint64_t seg_bytes = 1ull << (seg_shift - 3);
	const uint64_t byte0 = (uint64_t)seg * seg_bytes;
	uint64_t nbytes = local_bytes - byte0;
	if (nbytes > seg_bytes)
		nbytes = seg_bytes;
	const uint32_t n_vec = (uint32_t)((nbytes + 15) / 16); // the allocation is padded to 16 bytes
	uint4* lds4 = reinterpret_cast<uint4*>(dyn);
	uint4* g4 = reinterpret_cast<uint4*>(filter + byte0);
	for (uint32_t i = tid; i < n_vec; i += kApplyThreads)
		lds4[i] = g4[i];
	__syncthreads();
	uint32_t* lds = reinterpret_cast<uint32_t*>(dyn);
	for (uint32_t r = 0; r < n_regions; ++r) {
		const uint32_t reg = part_in_region(in, ibin, r);
		uint32_t n = in.cnt[reg];
		if (n > in.cap * kChunk)
			n = in.cap * kChunk;
		const uint4* e4 = reinterpret_cast<const uint4*>(in.ent + (uint64_t)reg * in.cap * kChunk);
		const uint32_t n_ev = (n + 3) / 4;
		for (uint32_t i = tid; i < n_ev; i += kApplyThreads) {
			const uint4 q = e4[i];
			const uint32_t e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				if (i * 4 + c >= n)
					continue;
				if (!QUERY)
					atomicOr(&lds[e[c] >> 5], 1u << (e[c] & 31));
				else if (!((lds[e[c] >> 5] >> (e[c] & 31)) & 1u))
					part_report_fail(sd, sd.pos_base + (((uint64_t)seg << seg_shift) | e[c]));
			}
		}
	}
	if (QUERY)
		return; // read-only sweep
	__syncthreads();
	for (uint32_t i = tid; i < n_vec; i += kApplyThreads)
		g4[i] = lds4[i];
}

// ---- launchers -----------------------------------------------------------------------------------
#ifdef BTLBF_PHASE_STAMPS
extern "C" void btlbf_debug_stamps(uint64_t* out16)
{
	(void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamp_out), sizeof(uint64_t) * 16);
	uint64_t z[16] = {0};
	(void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_out), z, sizeof z);
}
#endif
int part_tile_windows() { return kPartTile; }

uint32_t part_hash_lds_bytes(const HashParams& hp, uint32_t p0)
{
	return seq_tile_cap(kPartTile, hp.k) + seq_spaced_bytes(hp) + part_lds_bytes(p0);
}

// can pass A run for this hash configuration at all (possibly without the positional table)?
bool part_hash_fits(const HashParams& hp_in, uint32_t p0)
{
	HashParams hp = hp_in;
	if (part_hash_lds_bytes(hp, p0) <= kPartLdsBudget)
		return true;
	if (hp.n_seeds)
		return false;
	hp.use_pos_tab = 0;
	return part_hash_lds_bytes(hp, p0) <= kPartLdsBudget;
}

template <int H, bool Q>
static hipError_t launch_hash_h(const SeqArgs& a, const PartOut& out, uint32_t bin_shift, const PartSide& sd,
                                size_t dyn, hipStream_t s)
{
	const bool pow2 = a.mod.pow2 != 0, spaced = a.hp.n_seeds > 0;
#define BTLBF_PLAUNCH(P, S)                                                                                  \
	do {                                                                                                     \
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&part_hash_kernel<H, P, S, Q>),       \
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);             \
		if (e != hipSuccess)                                                                                 \
			return e;                                                                                        \
		hipLaunchKernelGGL((part_hash_kernel<H, P, S, Q>), dim3(out.regions), dim3(kPartThreads), dyn, s, a, \
		                   out, bin_shift, sd);                                                              \
	} while (0)
	if (pow2 && !spaced)
		BTLBF_PLAUNCH(true, false);
	else if (!pow2 && !spaced)
		BTLBF_PLAUNCH(false, false);
	else if (pow2 && spaced)
		BTLBF_PLAUNCH(true, true);
	else
		BTLBF_PLAUNCH(false, true);
#undef BTLBF_PLAUNCH
	return hipGetLastError();
}

template <bool Q>
static hipError_t launch_hash_q(const SeqArgs& a, const PartOut& out, uint32_t bin_shift, const PartSide& sd,
                                size_t dyn, hipStream_t s)
{
	switch (a.hp.h) {
	case 1: return launch_hash_h<1, Q>(a, out, bin_shift, sd, dyn, s);
	case 2: return launch_hash_h<2, Q>(a, out, bin_shift, sd, dyn, s);
	case 3: return launch_hash_h<3, Q>(a, out, bin_shift, sd, dyn, s);
	case 4: return launch_hash_h<4, Q>(a, out, bin_shift, sd, dyn, s);
	case 5: return launch_hash_h<5, Q>(a, out, bin_shift, sd, dyn, s);
	case 6: return launch_hash_h<6, Q>(a, out, bin_shift, sd, dyn, s);
	case 7: return launch_hash_h<7, Q>(a, out, bin_shift, sd, dyn, s);
	case 8: return launch_hash_h<8, Q>(a, out, bin_shift, sd, dyn, s);
	default: return hipErrorInvalidValue;
	}
}

bool part_supported_h(uint32_t h) { return h >= 1 && h <= 8; }

// pass A over tiles [a.first_tile, +a.n_tiles) (units: kPartTile windows); exactly out.regions
// workgroups are launched (one region each; idle ones still publish empty counts)
hipError_t launch_part_hash(const SeqArgs& a_in, const PartOut& out, uint32_t bin_shift, const PartSide& sd,
                            int query, hipStream_t s)
{
	SeqArgs a = a_in;
	if (a.n_tiles == 0)
		return hipSuccess;
	// LDS is nearly full with the staging rings: drop the positional table (Horner start-up
	// instead) when it does not fit; spaced seeds cannot do without it
	if (a.hp.n_seeds == 0 && part_hash_lds_bytes(a.hp, out.P) > kPartLdsBudget)
		a.hp.use_pos_tab = 0;
	a.tiles_per_block = (a.n_tiles + out.regions - 1) / out.regions;
	const size_t dyn = part_hash_lds_bytes(a.hp, out.P);
	return query ? launch_hash_q<true>(a, out, bin_shift, sd, dyn, s)
	             : launch_hash_q<false>(a, out, bin_shift, sd, dyn, s);
}

hipError_t launch_part_split(void* filter, const PartIn& in, uint32_t first_bin, uint32_t n_in_bins, const PartOut& out,
                             uint32_t sub_shift, uint32_t in_shift, const PartSide& sd, int query, hipStream_t s)
{
	if (n_in_bins == 0)
		return hipSuccess;
	const size_t dyn = part_lds_bytes(out.P);
	const void* fn = query ? reinterpret_cast<const void*>(&part_split_kernel<true>)
	                       : reinterpret_cast<const void*>(&part_split_kernel<false>);
	hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
	if (e != hipSuccess)
		return e;
	const uint32_t slices = out.regions;
	const dim3 grid(n_in_bins * slices);
	if (query)
		hipLaunchKernelGGL(part_split_kernel<true>, grid, dim3(kPartThreads), dyn, s, filter, in, out, slices,
		                   sub_shift, in_shift, first_bin, sd);
	else
		hipLaunchKernelGGL(part_split_kernel<false>, grid, dim3(kPartThreads), dyn, s, filter, in, out, slices,
		                   sub_shift, in_shift, first_bin, sd);
	return hipGetLastError();
}

hipError_t launch_part_apply(void* filter, uint64_t local_bytes, uint32_t seg_shift, uint64_t seg_first, uint64_t n_seg,
                             const PartIn& in, const PartSide& sd, int query, hipStream_t s)
{
	if (n_seg == 0)
		return hipSuccess;
	const size_t dyn = (size_t)1 << (seg_shift - 3);
	const void* fn = query ? reinterpret_cast<const void*>(&part_apply_kernel<true>)
	                       : reinterpret_cast<const void*>(&part_apply_kernel<false>);
	hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
	if (e != hipSuccess)
		return e;
	if (query)
		hipLaunchKernelGGL(part_apply_kernel<true>, dim3((unsigned)n_seg), dim3(kApplyThreads), dyn, s,
		                   static_cast<uint8_t*>(filter), local_bytes, seg_shift, (uint32_t)seg_first, in, sd);
	else
		hipLaunchKernelGGL(part_apply_kernel<false>, dim3((unsigned)n_seg), dim3(kApplyThreads), dyn, s,
		                   static_cast<uint8_t*>(filter), local_bytes, seg_shift, (uint32_t)seg_first, in, sd);
	return hipGetLastError();
}

// ---- failed-position set (partitioned query, resolve step) ----------------------------------------
// open-addressing table of 64-bit keys (position + 1; 0 = empty), linear probing
__global__ __launch_bounds__(256) void failset_build_kernel(const uint64_t* list, uint64_t n,
                                                           unsigned long long* table, uint64_t mask)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
	     i += (uint64_t)gridDim.x * blockDim.x) {
		const unsigned long long key = list[i] + 1;
		uint64_t slot = mix64(key) & mask;
		for (;;) {
			const unsigned long long prev = atomicCAS(&table[slot], 0ull, key);
			if (prev == 0ull || prev == key)
				break;
			slot = (slot + 1) & mask;
		}
	}
}

hipError_t launch_failset_build(const uint64_t* fail_list, uint64_t n, uint64_t* table, uint64_t mask,
                                hipStream_t s)
{
	if (n == 0)
		return hipSuccess;
	uint64_t blocks = (n + 255) / 256;
	if (blocks > 2048)
		blocks = 2048;
	hipLaunchKernelGGL(failset_build_kernel, dim3((unsigned)blocks), dim3(256), 0, s, fail_list, n,
	                   reinterpret_cast<unsigned long long*>(table), mask);
	return hipGetLastError();
}

// insert (test == 0) or test (test == 1) explicit GLOBAL positions (a spill list): positions outside
// [lo, lo+len) are ignored; failed tests are appended to sd's fail list with their global position
__global__ __launch_bounds__(256) void spill_kernel(uint32_t* words, const uint64_t* pos, uint64_t n, uint64_t lo,
                                                   uint64_t len, int test, const PartSide sd)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
	     i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t lp = pos[i] - lo;
		if (lp >= len)
			continue;
		if (!test)
			bf_set(words, lp);
		else if (!((bf_word(words, lp) >> (lp & 31)) & 1u))
			part_report_fail(sd, pos[i]);
	}
}

hipError_t launch_spill(void* filter, const uint64_t* pos, uint64_t n, uint64_t lo, uint64_t len, int test,
                        const PartSide& sd, hipStream_t s)
{
	if (n == 0)
		return hipSuccess;
	uint64_t blocks = (n + 255) / 256;
	if (blocks > 2048)
		blocks = 2048;
	hipLaunchKernelGGL(spill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<uint32_t*>(filter), pos, n,
	                   lo, len, test, sd);
	return hipGetLastError();
}

} // namespace btlbf

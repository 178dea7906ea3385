// csrc/partition_kernels.hip -- partitioned insert: turn h random HBM atomics per k-mer into streamed
// traffic plus LDS atomics.
//
// Why: the direct insert kernel is pinned to the memory system's random-request rate (~22 G
// memory-side atomics/s on MI355X).  Bit OR is order-free, so a batch of probe
// positions may be applied in any order -- in particular grouped by the 64 KiB (or 128 KiB) filter
// SEGMENT they fall into, with that segment held in LDS.  The result is bit-identical.
//
//   pass A  part_hash_kernel   fused ntHash (seq_core.hpp) + radix partition of the local positions
//                              by their top bits into p0 <= 1024 level-0 bins
//   pass B  part_split_kernel  (filters with more than 1024 segments) splits every level-0 bin into
//                              p1 <= 1024 sub-bins = segments
//   pass C  part_apply_kernel  one workgroup per segment: load the segment into LDS, ds_or every
//                              entry, store the segment back
//
// Bins are written as 128-byte CHUNKS (32 uint32 entries).  A workgroup stages entries per bin in an
// LDS ring and writes a chunk only when it is full, so every global write is one aligned 128-byte
// line.  Every workgroup writes into its OWN region of every bin (region = (bin, writer)), so chunk
// slots are handed out from an LDS counter: no global atomics anywhere in passes A and B.  Chunks
// flushed at kernel end are padded with a sentinel.  Entries that do not fit (a region over capacity,
// or a bin that receives more than about two rings' worth inside one round) are applied to the
// filter directly with atomicOr -- never dropped.
#include "seq_core.hpp"

namespace btlbf {

// 1024-thread workgroups (4 waves per SIMD; one workgroup per CU because the staging rings fill the
// LDS) with 4 windows per lane in pass A and 16 entries per lane in pass B: <= 128 VGPRs
static constexpr int kPartThreads = 1024;
static constexpr int kPartW = 4;                        // windows per lane in pass A
static constexpr int kPartTile = kPartThreads * kPartW; // windows per round of pass A
static constexpr int kApplyThreads = 512;
static constexpr uint32_t kChunk = 32;              // entries per chunk
static constexpr uint32_t kSentinel = 0xffffffffu;
static constexpr uint32_t kStageEntries = 32768;    // LDS staging: 128 KiB of uint32 entries
static constexpr uint32_t kPartLdsBudget = 160 * 1024 - 1024; // dynamic LDS a workgroup may ask for

// LDS image of the staged partitioner (carved from dynamic LDS by the kernels)
struct PartLds {
	uint32_t* stage;   // [P][SC] ring per bin, SC = kStageEntries / pow2ceil(P)
	uint32_t* pt;      // [P] low 16 bits: ring write position (mod 2^16); high 16: entries in the ring
	uint32_t* hist;    // [P] entries offered to the bin this round
	uint32_t* fl;      // [P] entries flushed from the bin this round (multiple of 32)
	uint32_t* written; // [P] chunks written to this workgroup's region of the bin
	uint32_t* fout;    // [1024] output chunk index (inside `out`) of each flush item
	uint16_t* flist;   // [1024] flush items: bin | ring chunk << 10
	uint32_t* fcount;
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
	return kStageEntries * 4 + 4 * P * 4 + 1024 * 4 + 1024 * 2 + 16;
}

__device__ __forceinline__ PartLds part_carve(uint8_t* base, uint32_t P)
{
	PartLds l;
	l.stage = reinterpret_cast<uint32_t*>(base);
	l.pt = l.stage + kStageEntries;
	l.hist = l.pt + P;
	l.fl = l.hist + P;
	l.written = l.fl + P;
	l.fout = l.written + P;
	l.flist = reinterpret_cast<uint16_t*>(l.fout + 1024);
	l.fcount = reinterpret_cast<uint32_t*>(l.flist + 1024);
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
		l.hist[b] = 0;
		l.fl[b] = 0;
		l.written[b] = 0;
	}
	if (threadIdx.x == 0)
		*l.fcount = 0;
}

// One round: every thread contributes E entries (bin[e] == kSentinel marks an empty slot).
// Precondition: hist[] all zero, fcount zero, and a barrier since they were written.
// Region r of bin b is chunks [(b*n_regions + r)*cap_chunks, +cap_chunks) of `out` (32-bit chunk
// indices: the scratch holds fewer than 2^32 chunks).  `ovf(bin, val)` must apply the entry to the
// filter directly.
#ifdef BTLBF_PHASE_STAMPS
#define STAMP(i)                                             \
	do {                                                     \
		if (threadIdx.x == 0) {                              \
			const uint64_t t__ = __builtin_readcyclecounter(); \
			g_stamp[i] += t__ - g_last;                      \
			g_last = t__;                                    \
		}                                                    \
	} while (0)
static __device__ uint64_t g_stamp_out[16];
#define STAMP_DECL uint64_t g_stamp[16] = {0}, g_last = __builtin_readcyclecounter()
#define STAMP_FLUSH                                                      \
	do {                                                                 \
		if (threadIdx.x == 0)                                            \
			for (int i__ = 0; i__ < 16; ++i__)                           \
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

template <int NT, int E, class OVF>
__device__ __forceinline__ void part_round(const PartLds& l, uint32_t P, const uint32_t (&bin)[E],
                                           const uint32_t (&val)[E], uint32_t* out, uint32_t n_regions,
                                           uint32_t region, uint32_t cap_chunks, OVF&& ovf STAMP_ARGS)
{
	const uint32_t tid = threadIdx.x;
	const uint32_t SC = 1u << l.sc_shift, ring = SC - 1;
	uint32_t rank[E];
	uint32_t staged = 0; // bit e: entry e found room in the ring before this round's flush
	static_assert(E <= 32, "one flag bit per entry");
#pragma unroll
	for (int e = 0; e < E; ++e)
		rank[e] = bin[e] != kSentinel ? atomicAdd(&l.hist[bin[e]], 1u) : 0;
	__syncthreads();
	STAMP(3);
	// entries that fit behind what the ring already holds are staged now
#pragma unroll
	for (int e = 0; e < E; ++e) {
		if (bin[e] != kSentinel) {
			const uint32_t w = l.pt[bin[e]];
			if ((w >> 16) + rank[e] < SC) {
				l.stage[(bin[e] << l.sc_shift) + ((w + rank[e]) & ring)] = val[e];
				staged |= 1u << e;
			}
		}
	}
	__syncthreads();
	STAMP(4);
	// per bin: full chunks leave; their slots in this workgroup's region come from an LDS counter
	for (uint32_t b = tid; b < P; b += NT) {
		const uint32_t w = l.pt[b], occ = w >> 16;
		const uint32_t tot = occ + l.hist[b];
		const uint32_t avail = tot < SC ? tot : SC;
		const uint32_t nfl = avail >> 5;
		l.fl[b] = nfl << 5;
		if (nfl) {
			const uint32_t base = atomicAdd(l.fcount, nfl);
			const uint32_t w0 = l.written[b];
			const uint32_t hc = ((w - occ) & ring) >> 5; // ring chunk at the read position
			const uint32_t o0 = (b * n_regions + region) * cap_chunks;
			for (uint32_t c = 0; c < nfl; ++c) {
				l.flist[base + c] = (uint16_t)(b | (((hc + c) & (ring >> 5)) << 10));
				l.fout[base + c] = w0 + c < cap_chunks ? o0 + w0 + c : 0xffffffffu;
			}
			l.written[b] = w0 + nfl;
		}
	}
	__syncthreads();
	STAMP(5);
	// flush full chunks: 8 lanes per chunk, 16 bytes per lane -> one aligned 128-byte line
	{
		const uint32_t n = *l.fcount;
		const uint32_t l8 = tid & 7;
		for (uint32_t j = tid >> 3; j < n; j += NT / 8) {
			const uint32_t it = l.flist[j], oc = l.fout[j];
			const uint32_t b = it & 1023, rc = it >> 10;
			const uint4 v = *reinterpret_cast<const uint4*>(&l.stage[(b << l.sc_shift) + (rc << 5) + l8 * 4]);
			if (oc != 0xffffffffu) {
				*reinterpret_cast<uint4*>(&out[(uint64_t)oc * kChunk + l8 * 4]) = v;
			} else {
				ovf(b, v.x);
				ovf(b, v.y);
				ovf(b, v.z);
				ovf(b, v.w);
			}
		}
	}
	__syncthreads();
	STAMP(6);
	// entries that did not fit before the flush: into the freed ring space, else applied directly
	{
#pragma unroll
		for (int e = 0; e < E; ++e) {
			if (bin[e] != kSentinel && !((staged >> e) & 1)) {
				const uint32_t w = l.pt[bin[e]];
				if ((w >> 16) + rank[e] - l.fl[bin[e]] < SC)
					l.stage[(bin[e] << l.sc_shift) + ((w + rank[e]) & ring)] = val[e];
				else
					ovf(bin[e], val[e]);
			}
		}
	}
	__syncthreads();
	STAMP(7);
	for (uint32_t b = tid; b < P; b += NT) {
		const uint32_t w = l.pt[b], occ = w >> 16, f = l.fl[b];
		const uint32_t tot = occ + l.hist[b] - f; // wants to be in the ring after the flush
		const uint32_t nocc = tot < SC ? tot : SC; // entries beyond went to the filter directly
		const uint32_t accepted = nocc + f - occ;
		l.pt[b] = ((w + accepted) & 0xffffu) | (nocc << 16);
		l.hist[b] = 0;
	}
	if (tid == 0)
		*l.fcount = 0;
	STAMP(8);
	// the caller's next barrier orders these writes before the next round
}

// flush whatever is staged (padded with the sentinel) and publish the chunk counts of this region
template <int NT, class OVF>
__device__ __forceinline__ void part_finish(const PartLds& l, uint32_t P, uint32_t* out, uint32_t* counts,
                                            uint32_t n_regions, uint32_t region, uint32_t cap_chunks, OVF&& ovf)
{
	__syncthreads();
	const uint32_t tid = threadIdx.x, lane32 = tid & 31;
	const uint32_t ring = (1u << l.sc_shift) - 1;
	for (uint32_t b = tid >> 5; b < P; b += NT / 32) {
		const uint32_t w = l.pt[b], n = w >> 16, hd = (w - n) & ring;
		uint32_t oc = l.written[b];
		const uint32_t o0 = (b * n_regions + region) * cap_chunks;
		for (uint32_t c = 0; c * kChunk < n; ++c, ++oc) {
			const uint32_t i = c * kChunk + lane32;
			const uint32_t v = i < n ? l.stage[(b << l.sc_shift) + ((hd + i) & ring)] : kSentinel;
			if (oc < cap_chunks)
				out[(uint64_t)(o0 + oc) * kChunk + lane32] = v;
			else if (v != kSentinel)
				ovf(b, v);
		}
		if (lane32 == 0)
			counts[b * n_regions + region] = oc < cap_chunks ? oc : cap_chunks;
	}
}

// ---- pass A --------------------------------------------------------------------------------------
// bin = local_position >> bin_shift ; entry = local_position & ((1 << bin_shift) - 1)
// region = blockIdx.x (gridDim.x == pa.regions0)
// QUERY = true: positions are partitioned for testing, not setting; an entry that cannot be staged
// is tested against the filter right away and reported through the fail list if its bit is clear
__device__ __forceinline__ void part_report_fail(const PartArgs& pa, uint64_t local_pos)
{
	const unsigned long long i = atomicAdd(pa.fail_count, 1ull);
	if (i < pa.fail_cap)
		pa.fail_list[i] = local_pos;
}

template <int H, bool POW2, bool SPACED, bool QUERY>
__global__ __launch_bounds__(kPartThreads) void part_hash_kernel(const SeqArgs a, const PartArgs pa)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
	__shared__ SeqShared sh;
	const uint32_t tid = threadIdx.x;
	const uint32_t k = a.hp.k;
	const uint32_t tile_cap = seq_tile_cap(kPartTile, k);
	uint8_t* tile = dyn;
	uint8_t* spaced_lds = dyn + tile_cap;
	const PartLds pl = part_carve(dyn + tile_cap + seq_spaced_bytes(a.hp), pa.p0);
	seq_setup_tables<kPartThreads, SPACED>(sh, a.hp, spaced_lds);
	part_init<kPartThreads>(pl, pa.p0);

	uint32_t* words = static_cast<uint32_t*>(a.filter);
	const uint32_t bin_shift = pa.bin_shift;
	const uint32_t ent_mask = (uint32_t)((1ull << bin_shift) - 1);
	auto ovf = [&](uint32_t b, uint32_t v) {
		const uint64_t lp = ((uint64_t)b << bin_shift) | v;
		if (!QUERY)
			bf_set(words, lp);
		else if (!((bf_word(words, lp) >> (lp & 31)) & 1u))
			part_report_fail(pa, lp);
	};
	const bool sharded = a.mod.shard_len != a.mod.size;
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

		uint32_t bin[kPartW * H], val[kPartW * H];
		uint32_t vmask = 0;
		seq_lane_windows<SPACED, kPartW>(tile, sh, a.hp, spaced_lds, tid * kPartW + mis, [&](int w, bool ok, const WinHash<SPACED>& wh) {
			vmask |= (uint32_t)ok << w;
#pragma unroll
			for (int i = 0; i < H; ++i) {
				uint64_t p = reduce_mod<POW2>(wh.at(i), a.mod);
				bool mine = ok;
				if (sharded) { // wave-uniform
					p -= a.mod.shard_lo;
					mine = ok && p < a.mod.shard_len;
				}
				bin[w * H + i] = mine ? (uint32_t)(p >> bin_shift) : kSentinel;
				val[w * H + i] = (uint32_t)p & ent_mask;
			}
		});
		if (a.valid_bits || a.hit_bits) {
			// two lanes (4 windows each) make one byte of the per-window bitmaps
			static_assert(kPartW == 4, "nibble packing");
			const uint32_t other = __shfl_xor(vmask, 1, 64);
			const uint64_t ob = (g0 >> 3) + (tid >> 1);
			if (!(tid & 1) && ob < out_bytes) {
				const uint8_t v = (uint8_t)(vmask | (other << 4));
				if (a.valid_bits)
					a.valid_bits[ob] = v;
				if (a.hit_bits)
					a.hit_bits[ob] = v; // a query starts from "every clean window hits"
			}
		}
		my_valid += __popc(vmask);
		STAMP(2);
		part_round<kPartThreads, kPartW * H>(pl, pa.p0, bin, val, pa.out0, pa.regions0, blockIdx.x, pa.cap0, ovf STAMP_PASS);
	}
	part_finish<kPartThreads>(pl, pa.p0, pa.out0, pa.cur0, pa.regions0, blockIdx.x, pa.cap0, ovf);
	if (a.counts) {
		const uint32_t wv = wave_sum(my_valid);
		if ((tid & 63) == 0 && wv)
			atomicAdd(reinterpret_cast<unsigned long long*>(a.counts), (unsigned long long)wv);
	}
	STAMP(9);
	STAMP_FLUSH;
}

// ---- pass B --------------------------------------------------------------------------------------
// workgroup (b0, g): blockIdx.x = b0 * regions1 + g.  It consumes pass-A regions g, g+regions1, ...
// of level-0 bin b0 and writes region g of every sub-bin of b0.
// entry e -> sub-bin e >> seg_shift, new entry e & seg_mask.
template <bool QUERY>
__global__ __launch_bounds__(kPartThreads) void part_split_kernel(void* filter, const PartArgs pa)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
	const uint32_t tid = threadIdx.x;
	const uint32_t b0 = blockIdx.x / pa.regions1, g = blockIdx.x % pa.regions1;
	const PartLds pl = part_carve(dyn, pa.p1);
	part_init<kPartThreads>(pl, pa.p1);
	uint32_t* words = static_cast<uint32_t*>(filter);
	const uint32_t seg_shift = pa.seg_shift;
	const uint32_t seg_mask = (1u << seg_shift) - 1;
	const uint64_t bin_base = (uint64_t)b0 << pa.bin_shift;
	auto ovf = [&](uint32_t sub, uint32_t v) {
		const uint64_t lp = bin_base | ((uint64_t)sub << seg_shift) | v;
		if (!QUERY)
			bf_set(words, lp);
		else if (!((bf_word(words, lp) >> (lp & 31)) & 1u))
			part_report_fail(pa, lp);
	};

	uint32_t* cur = pa.cur1 + (uint64_t)b0 * pa.p1 * pa.regions1;
	uint32_t* out = pa.out1 + (uint64_t)b0 * pa.p1 * pa.regions1 * pa.cap1 * kChunk;
	constexpr int kVec = 4; // uint4 loads per thread per round -> 16 entries
	STAMP_DECL;
	__syncthreads();
	for (uint32_t r = g; r < pa.regions0; r += pa.regions1) {
		const uint64_t reg = (uint64_t)b0 * pa.regions0 + r;
		uint32_t n_chunks = pa.cur0[reg];
		if (n_chunks > pa.cap0)
			n_chunks = pa.cap0;
		const uint4* src = reinterpret_cast<const uint4*>(pa.out0 + reg * pa.cap0 * kChunk);
		const uint32_t n_vec = n_chunks * (kChunk / 4);
		// software pipeline: the next round's loads are in flight while this round is partitioned
		uint4 nxt[kVec];
#pragma unroll
		for (int v = 0; v < kVec; ++v) {
			const uint32_t i = (uint32_t)v * kPartThreads + tid;
			nxt[v] = i < n_vec ? src[i] : make_uint4(kSentinel, kSentinel, kSentinel, kSentinel);
		}
		for (uint32_t base = 0; base < n_vec; base += kPartThreads * kVec) {
			uint32_t bin[kVec * 4], val[kVec * 4];
#pragma unroll
			for (int v = 0; v < kVec; ++v) {
				const uint32_t e4[4] = {nxt[v].x, nxt[v].y, nxt[v].z, nxt[v].w};
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					bin[v * 4 + c] = e4[c] == kSentinel ? kSentinel : e4[c] >> seg_shift;
					val[v * 4 + c] = e4[c] & seg_mask;
				}
				const uint32_t i = base + kPartThreads * kVec + (uint32_t)v * kPartThreads + tid;
				nxt[v] = i < n_vec ? src[i] : make_uint4(kSentinel, kSentinel, kSentinel, kSentinel);
			}
			part_round<kPartThreads, kVec * 4>(pl, pa.p1, bin, val, out, pa.regions1, g, pa.cap1, ovf STAMP_PASS);
			__syncthreads();
		}
	}
	part_finish<kPartThreads>(pl, pa.p1, out, cur, pa.regions1, g, pa.cap1, ovf);
}

// ---- pass C --------------------------------------------------------------------------------------
// one workgroup per segment; the segment's entries sit in `n_regions` regions of `cap` chunks each
template <bool QUERY>
__global__ __launch_bounds__(kApplyThreads) void part_apply_kernel(uint8_t* filter, uint64_t local_bytes,
                                                                 uint32_t seg_shift, const uint32_t* cur,
                                                                 const uint32_t* ent, uint32_t cap,
                                                                 uint32_t n_regions, const PartArgs pa)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
	__shared__ uint32_t any;
	const uint32_t tid = threadIdx.x;
	const uint64_t seg = blockIdx.x;
	if (tid == 0)
		any = 0;
	__syncthreads();
	uint32_t mine = 0;
	for (uint32_t r = tid; r < n_regions; r += kApplyThreads)
		mine |= cur[seg * n_regions + r];
	if (mine)
		any = 1;
	__syncthreads();
	if (!any)
		return; // untouched segment: no traffic at all
	const uint64_t seg_bytes = 1ull << (seg_shift - 3);
	const uint64_t byte0 = seg * seg_bytes;
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
		const uint64_t reg = seg * n_regions + r;
		uint32_t n_chunks = cur[reg];
		if (n_chunks > cap)
			n_chunks = cap;
		const uint4* e4 = reinterpret_cast<const uint4*>(ent + reg * cap * kChunk);
		const uint32_t n_ev = n_chunks * (kChunk / 4);
		for (uint32_t i = tid; i < n_ev; i += kApplyThreads) {
			const uint4 q = e4[i];
			const uint32_t e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				if (e[c] == kSentinel)
					continue;
				if (!QUERY)
					atomicOr(&lds[e[c] >> 5], 1u << (e[c] & 31));
				else if (!((lds[e[c] >> 5] >> (e[c] & 31)) & 1u))
					part_report_fail(pa, (seg << seg_shift) | e[c]);
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
static hipError_t launch_hash_h(const SeqArgs& a, const PartArgs& pa, unsigned blocks, size_t dyn, hipStream_t s)
{
	const bool pow2 = a.mod.pow2 != 0, spaced = a.hp.n_seeds > 0;
#define BTLBF_PLAUNCH(P, S)                                                                                  \
	do {                                                                                                     \
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&part_hash_kernel<H, P, S, Q>),       \
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);             \
		if (e != hipSuccess)                                                                                 \
			return e;                                                                                        \
		hipLaunchKernelGGL((part_hash_kernel<H, P, S, Q>), dim3(blocks), dim3(kPartThreads), dyn, s, a, pa); \
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
static hipError_t launch_hash_q(const SeqArgs& a, const PartArgs& pa, unsigned blocks, size_t dyn, hipStream_t s)
{
	switch (a.hp.h) {
	case 1: return launch_hash_h<1, Q>(a, pa, blocks, dyn, s);
	case 2: return launch_hash_h<2, Q>(a, pa, blocks, dyn, s);
	case 3: return launch_hash_h<3, Q>(a, pa, blocks, dyn, s);
	case 4: return launch_hash_h<4, Q>(a, pa, blocks, dyn, s);
	case 5: return launch_hash_h<5, Q>(a, pa, blocks, dyn, s);
	case 6: return launch_hash_h<6, Q>(a, pa, blocks, dyn, s);
	case 7: return launch_hash_h<7, Q>(a, pa, blocks, dyn, s);
	case 8: return launch_hash_h<8, Q>(a, pa, blocks, dyn, s);
	default: return hipErrorInvalidValue;
	}
}

bool part_supported_h(uint32_t h) { return h >= 1 && h <= 8; }

// pass A over tiles [a.first_tile, +a.n_tiles) (units: kPartTile windows); exactly pa.regions0
// workgroups are launched (one region each; idle ones still publish empty counts)
hipError_t launch_part_hash(const SeqArgs& a_in, const PartArgs& pa, hipStream_t s)
{
	SeqArgs a = a_in;
	if (a.n_tiles == 0)
		return hipSuccess;
	// LDS is nearly full with the staging rings: drop the positional table (Horner start-up
	// instead) when it does not fit; spaced seeds cannot do without it
	if (a.hp.n_seeds == 0 && part_hash_lds_bytes(a.hp, pa.p0) > kPartLdsBudget)
		a.hp.use_pos_tab = 0;
	const unsigned blocks = pa.regions0;
	a.tiles_per_block = (a.n_tiles + blocks - 1) / blocks;
	const size_t dyn = part_hash_lds_bytes(a.hp, pa.p0);
	return pa.fail_count ? launch_hash_q<true>(a, pa, blocks, dyn, s) : launch_hash_q<false>(a, pa, blocks, dyn, s);
}

hipError_t launch_part_split(void* filter, const PartArgs& pa, hipStream_t s)
{
	const size_t dyn = part_lds_bytes(pa.p1);
	const void* fn = pa.fail_count ? reinterpret_cast<const void*>(&part_split_kernel<true>)
	                               : reinterpret_cast<const void*>(&part_split_kernel<false>);
	hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
	if (e != hipSuccess)
		return e;
	if (pa.fail_count)
		hipLaunchKernelGGL(part_split_kernel<true>, dim3(pa.p0 * pa.regions1), dim3(kPartThreads), dyn, s, filter, pa);
	else
		hipLaunchKernelGGL(part_split_kernel<false>, dim3(pa.p0 * pa.regions1), dim3(kPartThreads), dyn, s, filter, pa);
	return hipGetLastError();
}

hipError_t launch_part_apply(void* filter, uint64_t local_bytes, const PartArgs& pa, int test_only, hipStream_t s)
{
	const size_t dyn = (size_t)1 << (pa.seg_shift - 3);
	const void* fn = test_only ? reinterpret_cast<const void*>(&part_apply_kernel<true>)
	                           : reinterpret_cast<const void*>(&part_apply_kernel<false>);
	hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
	if (e != hipSuccess)
		return e;
	const bool two = pa.levels == 2;
	const uint32_t* cur = two ? pa.cur1 : pa.cur0;
	const uint32_t* ent = two ? pa.out1 : pa.out0;
	const uint32_t cap = two ? pa.cap1 : pa.cap0, regions = two ? pa.regions1 : pa.regions0;
	if (test_only)
		hipLaunchKernelGGL(part_apply_kernel<true>, dim3((unsigned)pa.n_seg), dim3(kApplyThreads), dyn, s,
		                   static_cast<uint8_t*>(filter), local_bytes, pa.seg_shift, cur, ent, cap, regions, pa);
	else
		hipLaunchKernelGGL(part_apply_kernel<false>, dim3((unsigned)pa.n_seg), dim3(kApplyThreads), dyn, s,
		                   static_cast<uint8_t*>(filter), local_bytes, pa.seg_shift, cur, ent, cap, regions, pa);
	return hipGetLastError();
}

// ---- failed-position set (partitioned query, resolve step) ----------------------------------------
// open-addressing table of 64-bit keys (position + 1; 0 = empty), linear probing
__global__ __launch_bounds__(256) void failset_build_kernel(const uint64_t* list, uint64_t n,
                                                           unsigned long long* table, uint64_t mask)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
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

hipError_t launch_failset_build(const uint64_t* fail_list, uint64_t n, uint64_t* table, uint64_t mask, hipStream_t s)
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

} // namespace btlbf

// csrc/partition_kernels.hip -- partitioned insert: turn h random HBM atomics per k-mer into streamed
// traffic plus LDS atomics.
//
// Why: the direct insert kernel is pinned to the memory system's random-request rate (~22 G
// memory-side atomics/s on MI355X).  Bit OR is order-free, so a batch of probe
// positions may be applied in any order -- in particular grouped by the 64 KiB (or 128 KiB) filter
// SEGMENT they fall into, with that segment held in LDS.  The result is bit-identical.
//
//   pass A  part_hash_kernel   fused ntHash (seq_core.hpp) + radix partition of the local positions
//                              by their top bits into <= 1024 level-0 bins
//   pass B  part_split_kernel  (filters with more than 1024 segments) splits every level-0 bin into
//                              <= 1024 sub-bins = segments
//   pass C  part_apply_kernel  one workgroup per segment: load the segment into LDS, ds_or every
//                              entry, store the segment back
//
// Bins are arrays of 128-byte CHUNKS (32 uint32 entries).  A workgroup stages entries per bin in LDS
// and writes a chunk only when it is full, so every global write is one aligned 128-byte line and
// costs one atomicAdd on the bin's chunk cursor per 32 entries; chunks flushed at kernel end are
// padded with a sentinel.  Entries that do not fit (a bin over capacity, or more than two chunks of
// one bin inside one round) are applied to the filter directly with atomicOr -- never dropped.
#include "seq_core.hpp"

namespace btlbf {

static constexpr int kPartThreads = 512;
static constexpr int kPartTile = kPartThreads * kW; // windows per round of pass A
static constexpr uint32_t kChunk = 32;              // entries per chunk
static constexpr uint32_t kSentinel = 0xffffffffu;
static constexpr uint32_t kMaxBins = 1024;

// LDS image of the staged partitioner (carved from dynamic LDS by the kernels)
struct PartLds {
	uint32_t* stage; // [P][32]
	uint32_t* fill;  // [P] entries currently staged (0..32)
	uint32_t* hist;  // [P] per-round counts, then totals
	uint16_t* flist; // [P] bins to flush this round
	uint32_t* fcount;
};

__host__ __device__ inline uint32_t part_lds_bytes(uint32_t P)
{
	return P * kChunk * 4 + P * 4 + P * 4 + ((P * 2 + 15) / 16) * 16 + 16;
}

__device__ __forceinline__ PartLds part_carve(uint8_t* base, uint32_t P)
{
	PartLds l;
	l.stage = reinterpret_cast<uint32_t*>(base);
	l.fill = l.stage + P * kChunk;
	l.hist = l.fill + P;
	l.flist = reinterpret_cast<uint16_t*>(l.hist + P);
	l.fcount = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(l.flist) + ((P * 2 + 15) / 16) * 16);
	return l;
}

template <int NT>
__device__ __forceinline__ void part_init(const PartLds& l, uint32_t P)
{
	for (uint32_t b = threadIdx.x; b < P; b += NT) {
		l.fill[b] = 0;
		l.hist[b] = 0;
	}
	if (threadIdx.x == 0)
		*l.fcount = 0;
}

// One round: every thread contributes E entries (bin[e] == kSentinel marks an empty slot).
// Precondition: hist[] all zero, fcount zero, and a barrier since they were written.
// `ovf(bin, val)` must apply the entry to the filter directly.
template <int NT, int E, class OVF>
__device__ __forceinline__ void part_round(const PartLds& l, uint32_t P, const uint32_t (&bin)[E],
                                           const uint32_t (&val)[E], uint32_t* cursors, uint32_t* out,
                                           uint32_t cap_chunks, OVF&& ovf)
{
	const uint32_t tid = threadIdx.x;
	uint32_t idx[E];
	// rank inside the bin for this round
#pragma unroll
	for (int e = 0; e < E; ++e)
		idx[e] = bin[e] != kSentinel ? atomicAdd(&l.hist[bin[e]], 1u) : 0;
	__syncthreads();
	// absolute slot = entries already staged + rank; slots < 32 go into the open chunk
#pragma unroll
	for (int e = 0; e < E; ++e) {
		if (bin[e] != kSentinel) {
			idx[e] += l.fill[bin[e]];
			if (idx[e] < kChunk)
				l.stage[bin[e] * kChunk + idx[e]] = val[e];
		}
	}
	__syncthreads();
	for (uint32_t b = tid; b < P; b += NT) {
		const uint32_t tot = l.fill[b] + l.hist[b];
		l.hist[b] = tot;
		if (tot >= kChunk)
			l.flist[atomicAdd(l.fcount, 1u)] = (uint16_t)b;
	}
	__syncthreads();
	// flush full chunks: one half-wave (32 lanes) per chunk, one aligned 128-byte store
	{
		const uint32_t n = *l.fcount;
		const uint32_t lane32 = tid & 31;
		for (uint32_t j = tid >> 5; j < n; j += NT / 32) {
			const uint32_t b = l.flist[j];
			const uint32_t v = l.stage[b * kChunk + lane32];
			uint32_t chunk = 0;
			if (lane32 == 0)
				chunk = atomicAdd(&cursors[b], 1u);
			chunk = __shfl(chunk, tid & 32, 64);
			if (chunk < cap_chunks)
				out[((uint64_t)b * cap_chunks + chunk) * kChunk + lane32] = v;
			else
				ovf(b, v);
		}
	}
	__syncthreads();
	// second chunk's worth goes into the emptied buffer; anything beyond is applied directly
#pragma unroll
	for (int e = 0; e < E; ++e) {
		if (bin[e] != kSentinel && idx[e] >= kChunk) {
			const uint32_t i2 = idx[e] - kChunk;
			if (i2 < kChunk)
				l.stage[bin[e] * kChunk + i2] = val[e];
			else
				ovf(bin[e], val[e]);
		}
	}
	__syncthreads();
	for (uint32_t b = tid; b < P; b += NT) {
		const uint32_t tot = l.hist[b];
		l.fill[b] = tot >= kChunk ? (tot - kChunk < kChunk ? tot - kChunk : kChunk) : tot;
		l.hist[b] = 0;
	}
	if (tid == 0)
		*l.fcount = 0;
	// the caller's next barrier (tile staging / loop top) orders these writes before the next round
}

// flush whatever is staged, padded with the sentinel
template <int NT, class OVF>
__device__ __forceinline__ void part_finish(const PartLds& l, uint32_t P, uint32_t* cursors, uint32_t* out,
                                            uint32_t cap_chunks, OVF&& ovf)
{
	__syncthreads();
	const uint32_t tid = threadIdx.x, lane32 = tid & 31;
	for (uint32_t b = tid >> 5; b < P; b += NT / 32) {
		const uint32_t n = l.fill[b];
		if (n == 0)
			continue;
		const uint32_t v = lane32 < n ? l.stage[b * kChunk + lane32] : kSentinel;
		uint32_t chunk = 0;
		if (lane32 == 0)
			chunk = atomicAdd(&cursors[b], 1u);
		chunk = __shfl(chunk, tid & 32, 64);
		if (chunk < cap_chunks)
			out[((uint64_t)b * cap_chunks + chunk) * kChunk + lane32] = v;
		else if (v != kSentinel)
			ovf(b, v);
	}
}

// ---- pass A --------------------------------------------------------------------------------------
// bin = local_position >> bin_shift ; entry = local_position & ((1 << bin_shift) - 1)
template <int H, bool POW2, bool SPACED>
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
	const uint32_t ent_mask = (1u << bin_shift) - 1;
	auto ovf = [&](uint32_t b, uint32_t v) { bf_set(words, ((uint64_t)b << bin_shift) | v); };

	const uint64_t t_begin = a.first_tile + (uint64_t)blockIdx.x * a.tiles_per_block;
	uint64_t t_end = t_begin + a.tiles_per_block;
	if (t_end > a.first_tile + a.n_tiles)
		t_end = a.first_tile + a.n_tiles;
	const uint32_t L = a.layout.starts ? 0 : a.layout.read_len;
	uint32_t tile_off = 0;
	if (L && t_begin < t_end)
		tile_off = (uint32_t)((t_begin * (uint64_t)kPartTile) % L);
	const uint32_t tile_step = L ? (uint32_t)(kPartTile % L) : 0;

	for (uint64_t t = t_begin; t < t_end; ++t) {
		const uint64_t g0 = t * (uint64_t)kPartTile;
		const uint32_t mis = seq_stage_tile<kPartThreads>(tile, tile_cap, sh, a.seq, a.len, a.layout, k, g0, tile_off);
		tile_off = seq_next_tile_off(tile_off, tile_step, L);

		uint32_t bin[kW * H], val[kW * H];
		seq_lane_windows<SPACED>(tile, sh, a.hp, spaced_lds, tid * kW + mis, [&](int w, bool ok, const WinHash<SPACED>& wh) {
#pragma unroll
			for (int i = 0; i < H; ++i) {
				const uint64_t p = reduce_mod<POW2>(wh.at(i), a.mod) - a.mod.shard_lo;
				const bool mine = ok && p < a.mod.shard_len;
				bin[w * H + i] = mine ? (uint32_t)(p >> bin_shift) : kSentinel;
				val[w * H + i] = (uint32_t)p & ent_mask;
			}
		});
		part_round<kPartThreads, kW * H>(pl, pa.p0, bin, val, pa.cur0, pa.out0, pa.cap0, ovf);
	}
	part_finish<kPartThreads>(pl, pa.p0, pa.cur0, pa.out0, pa.cap0, ovf);
}

// ---- pass B --------------------------------------------------------------------------------------
// one workgroup per level-0 bin: entry e -> sub-bin e >> seg_shift, new entry e & seg_mask
__global__ __launch_bounds__(kPartThreads) void part_split_kernel(void* filter, const PartArgs pa)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
	const uint32_t tid = threadIdx.x;
	const uint32_t b0 = blockIdx.x;
	const PartLds pl = part_carve(dyn, pa.p1);
	part_init<kPartThreads>(pl, pa.p1);
	uint32_t* words = static_cast<uint32_t*>(filter);
	const uint32_t seg_shift = pa.seg_shift;
	const uint32_t seg_mask = (1u << seg_shift) - 1;
	const uint64_t bin_base = (uint64_t)b0 << pa.bin_shift;
	auto ovf = [&](uint32_t sub, uint32_t v) { bf_set(words, bin_base | ((uint64_t)sub << seg_shift) | v); };

	uint32_t n_chunks = pa.cur0[b0];
	if (n_chunks > pa.cap0)
		n_chunks = pa.cap0;
	const uint4* src = reinterpret_cast<const uint4*>(pa.out0 + (uint64_t)b0 * pa.cap0 * kChunk);
	const uint64_t n_vec = (uint64_t)n_chunks * (kChunk / 4);
	uint32_t* cur = pa.cur1 + (uint64_t)b0 * pa.p1;
	uint32_t* out = pa.out1 + (uint64_t)b0 * pa.p1 * pa.cap1 * kChunk;
	constexpr int kVec = 4; // uint4 loads per thread per round -> 16 entries
	__syncthreads();
	for (uint64_t base = 0; base < n_vec; base += (uint64_t)kPartThreads * kVec) {
		uint32_t bin[kVec * 4], val[kVec * 4];
#pragma unroll
		for (int v = 0; v < kVec; ++v) {
			const uint64_t i = base + (uint64_t)v * kPartThreads + tid;
			uint4 q = make_uint4(kSentinel, kSentinel, kSentinel, kSentinel);
			if (i < n_vec)
				q = src[i];
			const uint32_t e4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				bin[v * 4 + c] = e4[c] == kSentinel ? kSentinel : e4[c] >> seg_shift;
				val[v * 4 + c] = e4[c] & seg_mask;
			}
		}
		part_round<kPartThreads, kVec * 4>(pl, pa.p1, bin, val, cur, out, pa.cap1, ovf);
		__syncthreads();
	}
	part_finish<kPartThreads>(pl, pa.p1, cur, out, pa.cap1, ovf);
}

// ---- pass C --------------------------------------------------------------------------------------
// one workgroup per segment; `cur`/`ent`/`cap` describe the bins that ARE segments (level 1, or
// level 0 for filters with <= 1024 segments)
__global__ __launch_bounds__(kPartThreads) void part_apply_kernel(uint8_t* filter, uint64_t local_bytes,
                                                                 uint32_t seg_shift, const uint32_t* cur,
                                                                 const uint32_t* ent, uint32_t cap)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
	const uint32_t tid = threadIdx.x;
	const uint64_t seg = blockIdx.x;
	uint32_t n_chunks = cur[seg];
	if (n_chunks == 0)
		return; // untouched segment: no traffic at all
	if (n_chunks > cap)
		n_chunks = cap;
	const uint64_t seg_bytes = 1ull << (seg_shift - 3);
	const uint64_t byte0 = seg * seg_bytes;
	uint64_t nbytes = local_bytes - byte0;
	if (nbytes > seg_bytes)
		nbytes = seg_bytes;
	const uint32_t n_vec = (uint32_t)((nbytes + 15) / 16); // the allocation is padded to 16 bytes
	uint4* lds4 = reinterpret_cast<uint4*>(dyn);
	uint4* g4 = reinterpret_cast<uint4*>(filter + byte0);
	for (uint32_t i = tid; i < n_vec; i += kPartThreads)
		lds4[i] = g4[i];
	__syncthreads();
	uint32_t* lds = reinterpret_cast<uint32_t*>(dyn);
	const uint4* e4 = reinterpret_cast<const uint4*>(ent + seg * (uint64_t)cap * kChunk);
	const uint32_t n_ev = n_chunks * (kChunk / 4);
	for (uint32_t i = tid; i < n_ev; i += kPartThreads) {
		const uint4 q = e4[i];
		const uint32_t e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
		for (int c = 0; c < 4; ++c)
			if (e[c] != kSentinel)
				atomicOr(&lds[e[c] >> 5], 1u << (e[c] & 31));
	}
	__syncthreads();
	for (uint32_t i = tid; i < n_vec; i += kPartThreads)
		g4[i] = lds4[i];
}

// ---- launchers -----------------------------------------------------------------------------------
int part_tile_windows() { return kPartTile; }

uint32_t part_hash_lds_bytes(const HashParams& hp, uint32_t p0)
{
	return seq_tile_cap(kPartTile, hp.k) + seq_spaced_bytes(hp) + part_lds_bytes(p0);
}

template <int H>
static hipError_t launch_hash_h(const SeqArgs& a, const PartArgs& pa, unsigned blocks, size_t dyn, hipStream_t s)
{
	const bool pow2 = a.mod.pow2 != 0, spaced = a.hp.n_seeds > 0;
#define BTLBF_PLAUNCH(P, S)                                                                               \
	do {                                                                                                  \
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&part_hash_kernel<H, P, S>),       \
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);          \
		if (e != hipSuccess)                                                                              \
			return e;                                                                                     \
		hipLaunchKernelGGL((part_hash_kernel<H, P, S>), dim3(blocks), dim3(kPartThreads), dyn, s, a, pa); \
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

bool part_supported_h(uint32_t h) { return h >= 1 && h <= 5; }

// pass A over tiles [a.first_tile, +a.n_tiles) (units: kPartTile windows) with `blocks` workgroups
hipError_t launch_part_hash(const SeqArgs& a_in, const PartArgs& pa, unsigned blocks, hipStream_t s)
{
	SeqArgs a = a_in;
	if (a.n_tiles == 0)
		return hipSuccess;
	if (blocks > a.n_tiles)
		blocks = (unsigned)a.n_tiles;
	a.tiles_per_block = (a.n_tiles + blocks - 1) / blocks;
	blocks = (unsigned)((a.n_tiles + a.tiles_per_block - 1) / a.tiles_per_block);
	const size_t dyn = part_hash_lds_bytes(a.hp, pa.p0);
	switch (a.hp.h) {
	case 1: return launch_hash_h<1>(a, pa, blocks, dyn, s);
	case 2: return launch_hash_h<2>(a, pa, blocks, dyn, s);
	case 3: return launch_hash_h<3>(a, pa, blocks, dyn, s);
	case 4: return launch_hash_h<4>(a, pa, blocks, dyn, s);
	case 5: return launch_hash_h<5>(a, pa, blocks, dyn, s);
	default: return hipErrorInvalidValue;
	}
}

hipError_t launch_part_split(void* filter, const PartArgs& pa, hipStream_t s)
{
	const size_t dyn = part_lds_bytes(pa.p1);
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&part_split_kernel),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
	if (e != hipSuccess)
		return e;
	hipLaunchKernelGGL(part_split_kernel, dim3(pa.p0), dim3(kPartThreads), dyn, s, filter, pa);
	return hipGetLastError();
}

hipError_t launch_part_apply(void* filter, uint64_t local_bytes, const PartArgs& pa, hipStream_t s)
{
	const size_t dyn = (size_t)1 << (pa.seg_shift - 3);
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&part_apply_kernel),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
	if (e != hipSuccess)
		return e;
	const bool two = pa.levels == 2;
	hipLaunchKernelGGL(part_apply_kernel, dim3((unsigned)pa.n_seg), dim3(kPartThreads), dyn, s,
	                   static_cast<uint8_t*>(filter), local_bytes, pa.seg_shift, two ? pa.cur1 : pa.cur0,
	                   two ? pa.out1 : pa.out0, two ? pa.cap1 : pa.cap0);
	return hipGetLastError();
}

} // namespace btlbf

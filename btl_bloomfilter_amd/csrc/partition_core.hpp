// csrc/partition_core.hpp -- device code shared by the partition passes (partition_kernels.hip and the
// per-hash-count instantiations of pass A in part_hash_inst.hip): LDS layout of the staged
// partitioner, the partition round, overflow paths.  See partition_kernels.hip for the overview.
#pragma once
#include "seq_core.hpp"

namespace btlbf {

// 1024-thread workgroups (4 waves per SIMD; one workgroup per CU because the staging rings fill the
// LDS) with 8 windows per lane and tile in pass A (two rounds of 4) and 16 entries per lane and round
// in pass B: <= 128 VGPRs
static constexpr int kPartThreads = 1024;
static constexpr int kPartW = 8;                        // windows per lane and tile in pass A ...
static constexpr int kPartHalf = 4;                     // ... partitioned in two rounds of 4 (the rings hold one)
static constexpr int kPartTile = kPartThreads * kPartW; // windows per tile of pass A
static constexpr uint32_t kStageEntries = 32768; // LDS staging: 128 KiB of uint32 entries
static constexpr uint32_t kFlushItems = kStageEntries / kChunk; // chunks the rings can hold = most one round flushes
static constexpr uint32_t kPartLdsBudget = 160 * 1024 - 1664; // dynamic LDS a workgroup may ask for (static: SeqShared + the overlapped schedule's handover words, 1600 bytes)

// LDS image of the staged partitioner (carved from dynamic LDS by the kernels)
struct PartLds {
	uint32_t* stage;   // [P][SC] ring per bin, SC = kStageEntries / pow2ceil(P)
	uint32_t* pt;      // [P] low 16 bits: entries in the ring (+ offered this round); high 16: ring write position
	uint32_t* fl;      // [P] entries flushed from the bin this round (multiple of 32)
	uint32_t* written; // [P] chunks written to this workgroup's region of the bin
	uint16_t* flist;   // [kFlushItems] flush items, a private slice per wave: chunk index in `stage` ...
	uint32_t* fwc;     // [kFlushItems] ... and the chunk's index in the output array (0xffffffff: over capacity)
	uint32_t* dummy;   // [64] one word per lane: where an entry that found its ring full is "written" for now
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
	return kStageEntries * 4 + 3 * P * 4 + kFlushItems * 6 + 64 * 4 + 16;
}

__device__ __forceinline__ PartLds part_carve(uint8_t* base, uint32_t P)
{
	PartLds l;
	l.stage = reinterpret_cast<uint32_t*>(base);
	l.pt = l.stage + kStageEntries;
	l.fl = l.pt + P;
	l.written = l.fl + P;
	l.fwc = l.written + P;
	l.dummy = l.fwc + kFlushItems;
	l.flist = reinterpret_cast<uint16_t*>(l.dummy + 64);
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

// diagnostic build (-DBTLBF_EXP_NOBARRIER, tools/passa_time.py): the two barriers of a partition round are left
// out -- the output is garbage (in bounds), the time is what the round would cost if its phases could overlap freely
#ifdef BTLBF_EXP_NOBARRIER
#define PART_ROUND_BARRIER() __builtin_amdgcn_wave_barrier()
#else
#define PART_ROUND_BARRIER() __syncthreads()
#endif

#ifdef BTLBF_PHASE_STAMPS
#define STAMP(i)                                               \
	do {                                                       \
		if (threadIdx.x == 0) {                                \
			const uint64_t t__ = __builtin_readcyclecounter(); \
			g_stamp[i] += t__ - g_last;                        \
			g_last = t__;                                      \
		}                                                      \
	} while (0)
static __device__ uint64_t g_stamp_out[32];
#define STAMP_DECL uint64_t g_stamp[16] = {0}, g_last = __builtin_readcyclecounter()
#define STAMP_FLUSH                                                                                  \
	do {                                                                                             \
		if (threadIdx.x == 0)                                                                        \
			for (int i__ = 0; i__ < 16; ++i__)                                                       \
				atomicAdd((unsigned long long*)&g_stamp_out[i__], (unsigned long long)g_stamp[i__]); \
	} while (0)
#define STAMP_ARGS , uint64_t (&g_stamp)[16], uint64_t& g_last
#define STAMP_PASS , g_stamp, g_last
// inside part_round_p2 of the overlapped schedule (thread 0): slots 24.. of g_stamp_out
#define P2_STAMP_DECL uint64_t p2_last__ = __builtin_readcyclecounter()
#define P2_STAMP(i)                                                                                          \
	do {                                                                                                     \
		if (OWNERS && threadIdx.x == 0) {                                                                    \
			const uint64_t t__ = __builtin_readcyclecounter();                                               \
			atomicAdd((unsigned long long*)&g_stamp_out[24 + (i)], (unsigned long long)(t__ - p2_last__));   \
			p2_last__ = t__;                                                                                 \
		}                                                                                                    \
	} while (0)
#else
#define P2_STAMP_DECL
#define P2_STAMP(i)
#define STAMP(i)
#define STAMP_DECL
#define STAMP_FLUSH
#define STAMP_ARGS
#define STAMP_PASS
#endif

// One round: every thread contributes up to E entries (in groups of G, see `live`) to the bins
// [0, o.P) of this workgroup's output block; block-local bin b is global bin bin0 + b.
// Region `region` of global bin g is chunks [(g*o.regions + region)*o.cap, +o.cap) of o.ent
// (32-bit chunk indices: a pass's output holds fewer than 2^32 chunks).
// `ovf(bin, val)` takes the entries that cannot be staged.
//
// pt[b] packs (4 * ring write position << 16 | entries in the ring); the position is kept in BYTES of the ring
// (mod 2^16: rings hold at most 2^10 entries), so an entry's slot offset is one masked half-word.  Three phases, two barriers:
//  1. ONE returning LDS atomic per entry adds 0x40001: the old value is the entry's ring slot (high
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
// ALL: every entry of every lane exists (no `live` tests at all: the full rounds of pass B).
// Phase 1 of a round: the atomics and the ring writes of this lane's entries.  old[e] = what the atomic returned
// (0 for an entry that does not exist); returns whether any entry of the lane found its ring full ("late").
template <int E, int G, bool ALL = false>
__device__ __forceinline__ bool part_round_p1(const PartLds& l, const uint32_t (&bin)[E], const uint32_t (&val)[E],
                                              const uint32_t live, uint32_t (&old)[E])
{
	const uint32_t tid = threadIdx.x;
	const uint32_t SC = 1u << l.sc_shift, ring = SC - 1;
	static_assert(E <= 32 && E % G == 0, "one flag bit per entry; whole groups");
	// bit g of `live`: the G entries of group g exist (pass A: the H probes of one clean window).
	// All atomics of the lane are issued back to back (independent), then consumed.
#pragma unroll
	for (int g = 0; g < E / G; ++g) {
		if (ALL || ((live >> g) & 1)) {
#pragma unroll
			for (int e = g * G; e < (g + 1) * G; ++e)
				old[e] = atomicAdd(&l.pt[bin[e]], 0x40001u);
		} else {
#pragma unroll
			for (int e = g * G; e < (g + 1) * G; ++e)
				old[e] = 0;
		}
	}
	// An entry that finds room is written into its ring; one that does not is written to the lane's dummy word
	// (no branch, no exec juggling per entry) and remembered as one bit per LANE: phase 3 looks at old[] again.
	bool any_late = false;
	uint32_t* const dummy = &l.dummy[tid & 63];
	const uint32_t stage_shift = l.sc_shift + 2, ring4 = ring << 2;
#pragma unroll
	for (int g = 0; g < E / G; ++g) {
		if (ALL || ((live >> g) & 1)) {
#pragma unroll
			for (int e = g * G; e < (g + 1) * G; ++e) {
				const bool fits = (old[e] & 0xffffu) < SC;
				const uint32_t slot4 = (old[e] >> 16) & ring4; // byte offset inside the ring
				uint32_t* dst = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(l.stage) + ((bin[e] << stage_shift) + slot4));
				*(fits ? dst : dummy) = val[e];
				any_late |= !fits;
			}
		}
	}
	return any_late;
}

// Phase 1 for a schedule that must not keep the round's entries in registers across its barriers (pass A's
// overlapped schedule): as part_round_p1, but an entry that finds its ring full is stored in the round's LATE IMAGE,
// a mirror of the rings in global memory (a workgroup's own 128 KiB, L2 traffic): row of its bin, slot = how many
// entries were beyond the ring's end ahead of it.  The address comes from what the atomic returned -- no compaction,
// no second look at the entries (0.8 % of them are late with 512 bins, but every wave has one in every round: a
// per-entry "is it late" pass costs as much as the atomics themselves).  A bin whose ring overflowed has its whole
// ring flushed (part_round_p2), so every entry in the image fits afterwards: part_late_fetch / part_late_apply move
// them into the rings around the round's second barrier.  Entries beyond even the image (a round that offers a bin
// more than two rings) are the ones part_round_p2 counts as overflowed: `ovf` takes them here.
template <int E, int G, bool ONE_OVF = false, class OVF>
__device__ __forceinline__ void part_round_p1_late(const PartLds& l, const uint32_t (&bin)[E], const uint32_t (&val)[E],
                                                   const uint32_t live, uint32_t* late, OVF&& ovf)
{
	const uint32_t tid = threadIdx.x;
	const uint32_t SC = 1u << l.sc_shift, ring = SC - 1;
	static_assert(E <= 32 && E % G == 0, "one flag bit per entry; whole groups");
	uint32_t old[E];
#pragma unroll
	for (int g = 0; g < E / G; ++g) {
		if ((live >> g) & 1) {
#pragma unroll
			for (int e = g * G; e < (g + 1) * G; ++e)
				old[e] = atomicAdd(&l.pt[bin[e]], 0x40001u);
		} else {
#pragma unroll
			for (int e = g * G; e < (g + 1) * G; ++e)
				old[e] = 0;
		}
	}
	uint32_t over = 0; // bit e: entry e lies beyond the late image too
	uint32_t* const dummy = &l.dummy[tid & 63];
	const uint32_t stage_shift = l.sc_shift + 2, ring4 = ring << 2;
#pragma unroll
	for (int g = 0; g < E / G; ++g) {
		if ((live >> g) & 1) {
#pragma unroll
			for (int e = g * G; e < (g + 1) * G; ++e) {
				const uint32_t cnt = old[e] & 0xffffu;
				const bool fits = cnt < SC;
				const uint32_t slot4 = (old[e] >> 16) & ring4; // byte offset inside the ring
				uint32_t* dst = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(l.stage) + ((bin[e] << stage_shift) + slot4));
				*(fits ? dst : dummy) = val[e];
				if (!fits) {
					const uint32_t j = cnt - SC;
					if (j < SC)
						late[(bin[e] << l.sc_shift) + j] = val[e];
					else
						over |= 1u << e;
				}
			}
		}
	}
	// This path runs when a round offers one bin more than two rings of entries.  ONE_OVF: one copy of `ovf`, the
	// lane's entry picked with a select chain -- sixteen inlined copies, one per entry index, cost the WINDOW / QUERY
	// variant of pass A 32 spilled registers (gather mode's query: +11 %), while the variants without WINDOW are 3 %
	// faster with the sixteen copies than with the chain (register allocation, not this path's run time).
	if (!ONE_OVF) {
		if (__any(over != 0)) {
#pragma unroll
			for (int e = 0; e < E; ++e)
				if ((over >> e) & 1u)
					ovf(bin[e], val[e]);
		}
		return;
	}
	while (__any(over != 0)) {
		const int pick = over ? __builtin_ctz(over) : -1;
		uint32_t b = 0, v = 0;
#pragma unroll
		for (int e = 0; e < E; ++e) {
			b = pick == e ? bin[e] : b;
			v = pick == e ? val[e] : v;
		}
		if (pick >= 0) {
			ovf(b, v);
			over &= over - 1;
		}
	}
}

// What the owner of a bin leaves in fl[] for the waves that move the late entries (OWNERS > 0 in part_round_p2):
// low half = entries of the bin in the late image, high half = ring slot of the first of them.
__device__ __forceinline__ uint32_t part_late_word(uint32_t n_late, uint32_t slot0) { return n_late | (slot0 << 16); }

// The late entries of the 64 bins [b0, b0 + 64) -- one bin per lane, `w` = the lane's fl[] word (0 for a lane without
// a bin) -- spread evenly over the wave: lane i takes the i-th of them.  Returns the number of late entries of the
// wave; lanes i < that get the image index `src` and the ring index `dst` (both in words) of their entry.
// `skip`: entries taken in earlier calls (a wave with more than 64 late entries calls again).
__device__ __forceinline__ uint32_t part_late_assign(const PartLds& l, uint32_t b0, uint32_t w, uint32_t skip, uint32_t& src,
                                                     uint32_t& dst)
{
	const uint32_t lane = threadIdx.x & 63, ring = (1u << l.sc_shift) - 1;
	const uint32_t incl = wave_scan_incl(w & 0xffffu);
	const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
	const uint32_t i = lane + skip;
	// the lane q whose bin holds entry i: the first with incl[q] > i (binary search over the wave's prefix sums)
	uint32_t q = 0;
#pragma unroll
	for (int step = 32; step > 0; step >>= 1) {
		const uint32_t probe = q + step - 1;
		const uint32_t v = __shfl(incl, (int)probe, 64);
		q += v <= i ? step : 0;
	}
	q = q > 63 ? 63 : q;
	const uint32_t wq = __shfl(w, (int)q, 64), iq = __shfl(incl, (int)q, 64);
	const uint32_t j = i - (iq - (wq & 0xffffu)); // index inside bin b0 + q
	src = ((b0 + q) << l.sc_shift) + j;
	dst = ((b0 + q) << l.sc_shift) + (((wq >> 16) + j) & ring);
	return total;
}

// a late entry (read back through L2: the image is rewritten every other round, L1 may hold an old line)
__device__ __forceinline__ uint32_t part_late_load(const uint32_t* late, uint32_t i)
{
	return __hip_atomic_load(late + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Phase 2 (between the round's two barriers): the lanes that own bins flush the full chunks of their rings and
// write the bins' state for the next round.
// Bins are owned by lanes (P <= NT): wave v owns bins [v*bpw, (v+1)*bpw) and flushes them itself, through its
// private slice of the flush list -- no workgroup barrier in between.  OWNERS == 0: with at least half as many bins
// as threads a wave owns 64 (spreading 512 bins over all 16 waves was measured: the same time and 5 % more
// instructions, because the waves without bins skip this block outright); with FEW bins -- the 64-way split passes
// of a multi-GPU owner -- they are spread over all waves, or one wave would flush everything (owner side of the C4
// geometry: 0.65 -> 0.49 s per pass).  OWNERS > 0: the first OWNERS waves own all bins (P <= 64 * OWNERS) and only
// they need to come here (pass A's overlapped schedule: the other waves hash meanwhile); fl[] then holds
// part_late_word() instead of the flushed count, and each owner wave adds one to *p2_done (LDS) once its words stand.
template <int NT, int OWNERS = 0, class OVF>
__device__ __forceinline__ void part_round_p2(const PartLds& l, const PartOut& o, uint32_t bin0, uint32_t region,
                                              OVF&& ovf, uint32_t* p2_done = nullptr)
{
	const uint32_t tid = threadIdx.x;
	const uint32_t P = o.P;
	const uint32_t SC = 1u << l.sc_shift, ring = SC - 1;
	const uint32_t lane = tid & 63;
	P2_STAMP_DECL;
	const uint32_t bpw = OWNERS ? (P + OWNERS - 1) / OWNERS : (P >= NT / 2 ? 64u : (P + NT / 64 - 1) / (NT / 64));
	const uint32_t b = (tid >> 6) * bpw + lane;
	uint32_t nfl = 0, rd0 = 0, w0 = 0;
	if (lane < bpw && b < P && (!OWNERS || (tid >> 6) < (uint32_t)OWNERS)) {
		const uint32_t w = l.pt[b];
		const uint32_t occ = w & 0xffffu;           // ring content + everything offered this round
		const uint32_t avail = occ < SC ? occ : SC; // entries that really sit in the ring
		nfl = avail >> kChunkShift;
		const uint32_t f = nfl << kChunkShift;
		// state for the next round (see part_round)
		const uint32_t tot = occ - f;
		const uint32_t nocc = tot < SC ? tot : SC;
		l.pt[b] = (((w >> 16) - 4 * (tot - nocc)) << 16) | nocc;
		rd0 = ((w >> 18) - occ) & ring; // read position of the ring: both halves of pt grew alike
		if (OWNERS) {
			// the bin's late entries (part_round_p1_late): those beyond the ring's end, as far as the image holds them;
			// the ring was full, so all of it is flushed now and entry j of the image goes to slot rd0 + j
			const uint32_t n_late = occ > SC ? (occ - SC < SC ? occ - SC : SC) : 0u;
			l.fl[b] = part_late_word(n_late, rd0);
		} else {
			l.fl[b] = f;
		}
		if (nfl) {
			w0 = l.written[b];
			l.written[b] = w0 + nfl;
		}
	}
	if (OWNERS && p2_done) {
		// this wave's fl[] words are written: tell the waves that fetch the late entries before the round's second barrier
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		if (lane == 0)
			atomicAdd(p2_done, 1u);
	}
	P2_STAMP(0);
	// inclusive prefix sum of nfl over the wave (DPP row shifts + row broadcasts, no LDS) -> slots
	// in the wave's slice (64 bins * SC/kChunk items)
	const uint32_t incl = wave_scan_incl(nfl);
	const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
	const uint32_t slice = (tid >> 6) * ((bpw << l.sc_shift) >> kChunkShift);
	// a flush item is (chunk of the staging area, chunk of the output array): both are worked out here, once
	// per chunk, and the part that depends on the bin alone is the same in every round (hoisted)
	const uint32_t cshift = l.sc_shift - kChunkShift; // log2(chunks per ring)
	const uint32_t cbase = ((bin0 + b) * o.regions + region) * o.cap;
	for (uint32_t c = 0; c < nfl; ++c) {
		const uint32_t j = slice + incl - nfl + c;
		l.flist[j] = (uint16_t)((b << cshift) + (((rd0 >> kChunkShift) + c) & (ring >> kChunkShift)));
		l.fwc[j] = w0 + c < o.cap ? cbase + w0 + c : 0xffffffffu;
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	P2_STAMP(1);
	// kChunk/4 lanes per chunk, 16 bytes per lane -> one aligned line per chunk
	constexpr uint32_t kLanesPerChunk = kChunk / 4;
	const uint32_t l4 = lane & (kLanesPerChunk - 1);
	for (uint32_t j = lane / kLanesPerChunk; j < total; j += 64 / kLanesPerChunk) {
		const uint32_t it = l.flist[slice + j], wc = l.fwc[slice + j];
		const uint4 v = *reinterpret_cast<const uint4*>(&l.stage[it * kChunk + l4 * 4]);
		if (wc != 0xffffffffu) {
#ifdef BTLBF_NT_FLUSH // pass A (part_hash_inst.hip defines it): streaming stores for the chunks, so that they do not
			// push the late image and the reads out of the L2 -- pass A 39.35 -> 39.0 ms, query 41.3 -> 40.8; pass B, whose
			// workgroups have nothing else to keep there, is 1 % slower with them and keeps plain stores
			typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
			v4u_t vv = {v.x, v.y, v.z, v.w};
			__builtin_nontemporal_store(vv, reinterpret_cast<v4u_t*>(&o.ent[(uint64_t)wc * kChunk + l4 * 4]));
#else
			*reinterpret_cast<uint4*>(&o.ent[(uint64_t)wc * kChunk + l4 * 4]) = v;
#endif
		} else {
			const uint32_t fb = it >> cshift;
			ovf(fb, v.x);
			ovf(fb, v.y);
			ovf(fb, v.z);
			ovf(fb, v.w);
		}
	}
	P2_STAMP(2);
}

// Phase 3 (after the second barrier): entries that did not fit before the flush go into the freed ring space,
// else overflow.  No barrier is needed behind it (see part_round).
template <int E, class OVF>
__device__ __forceinline__ void part_round_p3(const PartLds& l, const uint32_t (&bin)[E], const uint32_t (&val)[E],
                                              const uint32_t (&old)[E], const bool any_late, OVF&& ovf)
{
	const uint32_t SC = 1u << l.sc_shift, ring = SC - 1;
	if (any_late) {
#pragma unroll
		for (int e = 0; e < E; ++e) {
			const uint32_t occ = old[e] & 0xffffu; // 0 for an entry that does not exist
			if (occ >= SC) {
				// a ring of ONE chunk always flushed that chunk when an entry found it full
				const uint32_t f = l.sc_shift == kChunkShift ? SC : l.fl[bin[e]];
				if (occ - f < SC)
					l.stage[(bin[e] << l.sc_shift) + ((old[e] >> 18) & ring)] = val[e];
				else
					ovf(bin[e], val[e]);
			}
		}
	}
}

template <int NT, int E, int G, bool ALL = false, class OVF>
__device__ __forceinline__ void part_round(const PartLds& l, const PartOut& o, uint32_t bin0, uint32_t region,
                                           const uint32_t (&bin)[E], const uint32_t (&val)[E], const uint32_t live,
                                           OVF&& ovf STAMP_ARGS)
{
	uint32_t old[E];
	const bool any_late = part_round_p1<E, G, ALL>(l, bin, val, live, old);
	PART_ROUND_BARRIER();
	STAMP(4);
	part_round_p2<NT>(l, o, bin0, region, ovf);
	PART_ROUND_BARRIER();
	STAMP(6);
	part_round_p3<E>(l, bin, val, old, any_late, ovf);
	STAMP(7);
}

// flush whatever is staged and publish the ENTRY count of this workgroup's region of every bin
// (exact; the data itself is padded to a multiple of 4 entries, see below)
template <int NT, class OVF>
__device__ __forceinline__ void part_finish(const PartLds& l, const PartOut& o, uint32_t bin0, uint32_t region,
                                            OVF&& ovf)
{
	__syncthreads();
	const uint32_t tid = threadIdx.x, ln = tid & (kChunk - 1);
	const uint32_t ring = (1u << l.sc_shift) - 1;
	for (uint32_t b = tid / kChunk; b < o.P; b += NT / kChunk) {
		const uint32_t w = l.pt[b], n = w & 0xffffu, hd = ((w >> 18) - n) & ring;
		const uint32_t w0 = l.written[b];
		const uint32_t full = w0 < o.cap ? w0 : o.cap; // chunks of this region that really hold data
		const uint64_t o0 = (uint64_t)((bin0 + b) * o.regions + region) * o.cap;
		uint32_t stored = 0;
		for (uint32_t c = 0; c * kChunk < n; ++c) {
			const uint32_t i = c * kChunk + ln;
			// the tail is padded to a whole 16-byte vector with copies of the last entry: readers
			// take whole vectors, and a repeated position changes nothing for OR / test
			const uint32_t src = i < n ? i : n - 1;
			const uint32_t v = i < ((n + 3) & ~3u) ? l.stage[(b << l.sc_shift) + ((hd + src) & ring)] : 0;
			if (w0 + c < o.cap) {
				o.ent[(o0 + w0 + c) * kChunk + ln] = v;
				stored = (c + 1) * kChunk < n ? (c + 1) * kChunk : n;
			} else if (i < n) {
				ovf(b, v);
			}
		}
		if (ln == 0)
			o.cnt[(bin0 + b) * o.regions + region] = full * kChunk + stored;
	}
}

// ---- the small geometry: two 512-thread workgroups per CU (pass A with 257..512 level-0 bins) ----------
// One 1024-thread workgroup per CU runs its phases in lockstep: while every wave waits on LDS atomics the
// SIMDs idle, while every wave hashes the LDS idles, and a barrier stalls the whole CU.  Two independent
// workgroups per CU (80 KiB of LDS each) let the hardware overlap one's hashing with the other's LDS and
// barrier phases.  Each has 64 KiB of rings: 32 entries per bin, flushed in 64-byte chunks of 16 entries
// (a 32-entry ring cannot work with 32-entry chunks: up to 31 left-over entries plus one round's ~16 new
// ones).  Regions are still counted in 128-byte units (PartOut::cap), so readers see no difference.
static constexpr int kPartThreadsS = 512;
static constexpr uint32_t kStageEntriesS = 16384;
static constexpr uint32_t kRingS = 32, kRingShiftS = 5; // entries per bin ring
static constexpr uint32_t kChunkS = 16, kChunkShiftS = 4;
static constexpr uint32_t kPartMinBinsS = 257, kPartMaxBinsS = 512;
static constexpr uint32_t kPartLdsBudgetS = 80 * 1024 - 1536; // dynamic LDS per workgroup at two per CU

struct PartLdsS {
	uint32_t* stage;   // [512][32]
	uint32_t* pt;      // [P] as PartLds::pt
	uint32_t* written; // [P] 16-entry chunks written to this workgroup's region of the bin
	uint16_t* fl;      // [P] entries flushed from the bin this round
};

__host__ __device__ inline uint32_t part_lds_bytes_s(uint32_t P) { return kStageEntriesS * 4 + P * 10 + 16; }

__device__ __forceinline__ PartLdsS part_carve_s(uint8_t* base, uint32_t P)
{
	PartLdsS l;
	l.stage = reinterpret_cast<uint32_t*>(base);
	l.pt = l.stage + kStageEntriesS;
	l.written = l.pt + P;
	l.fl = reinterpret_cast<uint16_t*>(l.written + P);
	return l;
}

__device__ __forceinline__ void part_init_s(const PartLdsS& l, uint32_t P)
{
	for (uint32_t b = threadIdx.x; b < P; b += kPartThreadsS) {
		l.pt[b] = 0;
		l.written[b] = 0;
		l.fl[b] = 0;
	}
}

// part_round for the small geometry (same three phases and the same pt[] protocol).  Lane b owns bin b
// (P <= 512 = threads); a bin flushes at most two chunks per round.  The flush work of a wave's 64 bins is
// compacted WITHOUT an LDS list: owners push their bin's data to lane `rank` with ds_permute (first
// chunks, then second chunks), and four lanes copy each 64-byte chunk.
// `pre_flush()` runs between the first barrier and the chunk stores (pass A pins the arrival of the next
// tile's prefetched words there: vector-memory operations retire in order, so waiting for those loads
// AFTER a flush would wait for the flush's stores as well).
template <int E, int G, class OVF, class PRE>
__device__ __forceinline__ void part_round_s(const PartLdsS& l, const PartOut& o, uint32_t region,
                                             const uint32_t (&bin)[E], const uint32_t (&val)[E], const uint32_t live,
                                             OVF&& ovf, PRE&& pre_flush STAMP_ARGS)
{
	const uint32_t tid = threadIdx.x;
	const uint32_t P = o.P;
	constexpr uint32_t SC = kRingS, ring = kRingS - 1;
	uint32_t old[E];
	uint32_t late = 0;
	static_assert(E <= 32 && E % G == 0, "one flag bit per entry; whole groups");
#pragma unroll
	for (int g = 0; g < E / G; ++g) {
		if ((live >> g) & 1) {
#pragma unroll
			for (int e = g * G; e < (g + 1) * G; ++e)
				old[e] = atomicAdd(&l.pt[bin[e]], 0x10001u);
		}
	}
#pragma unroll
	for (int g = 0; g < E / G; ++g) {
		if ((live >> g) & 1) {
#pragma unroll
			for (int e = g * G; e < (g + 1) * G; ++e) {
				if ((old[e] & 0xffffu) < SC)
					l.stage[(bin[e] << kRingShiftS) + ((old[e] >> 16) & ring)] = val[e];
				else
					late |= 1u << e;
			}
		}
	}
	__syncthreads();
	STAMP(4);
	pre_flush();
	{
		const uint32_t b = tid, lane = tid & 63;
		uint32_t nfl = 0, rd0 = 0, w0 = 0;
		if (b < P) {
			const uint32_t w = l.pt[b];
			const uint32_t occ = w & 0xffffu;
			const uint32_t avail = occ < SC ? occ : SC;
			nfl = avail >> kChunkShiftS; // 0..2
			const uint32_t f = nfl << kChunkShiftS;
			const uint32_t tot = occ - f;
			const uint32_t nocc = tot < SC ? tot : SC;
			l.pt[b] = (((w >> 16) - (tot - nocc)) << 16) | nocc;
			l.fl[b] = (uint16_t)f;
			if (nfl) {
				w0 = l.written[b];
				l.written[b] = w0 + nfl;
				rd0 = ((w >> 16) - occ) & ring; // multiple of 16: chunks leave whole
			}
		}
		// compaction: list 1 = bins flushing at least one chunk, list 2 = bins flushing two
		const uint64_t m1 = __ballot(nfl >= 1), m2 = __ballot(nfl >= 2);
		const uint32_t n1 = __popcll(m1), n2 = __popcll(m2);
		const uint32_t below = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0));
		const uint32_t below2 = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0));
		// item = bin-in-wave | first ring chunk << 8 ; region chunk index
		const uint32_t item = lane | ((rd0 >> kChunkShiftS) << 8);
		// every lane pushes (ds_permute delivers only to active lanes): owners with work to their rank, the
		// others behind the list -- a full permutation, so no two lanes push to the same place
		const uint32_t d1 = nfl >= 1 ? below : n1 + lane - below;
		const uint32_t d2 = nfl >= 2 ? below2 : n2 + lane - below2;
		const uint32_t it1 = (uint32_t)__builtin_amdgcn_ds_permute((int)(d1 << 2), (int)item);
		const uint32_t wc1 = (uint32_t)__builtin_amdgcn_ds_permute((int)(d1 << 2), (int)w0);
		const uint32_t it2 = (uint32_t)__builtin_amdgcn_ds_permute((int)(d2 << 2), (int)(item ^ 0x100u)); // the ring's other chunk
		const uint32_t wc2 = (uint32_t)__builtin_amdgcn_ds_permute((int)(d2 << 2), (int)(w0 + 1));
		// lanes [0, n1) of it1/wc1 and [0, n2) of it2/wc2 now hold the lists
		constexpr uint32_t kLanesPerChunk = kChunkS / 4; // 16 bytes per lane
		const uint32_t l4 = lane & (kLanesPerChunk - 1), q = lane / kLanesPerChunk;
		const uint32_t cap_s = o.cap * (kChunk / kChunkS);
		const uint32_t wave_bin0 = tid & ~63u;
		auto copy_list = [&](uint32_t items, uint32_t wcs, uint32_t n) {
			for (uint32_t j0 = 0; j0 < n; j0 += 64 / kLanesPerChunk) {
				const uint32_t j = j0 + q;
				const uint32_t it = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)items);
				const uint32_t wc = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)wcs);
				if (j < n) {
					const uint32_t fb = wave_bin0 + (it & 63u), rc = (it >> 8) & 1u;
					const uint4 v = *reinterpret_cast<const uint4*>(&l.stage[(fb << kRingShiftS) + (rc << kChunkShiftS) + l4 * 4]);
					if (wc < cap_s) {
						const uint64_t dst = ((uint64_t)(fb * o.regions + region) * cap_s + wc) * kChunkS + l4 * 4;
						*reinterpret_cast<uint4*>(&o.ent[dst]) = v;
					} else {
						ovf(fb, v.x);
						ovf(fb, v.y);
						ovf(fb, v.z);
						ovf(fb, v.w);
					}
				}
			}
		};
		copy_list(it1, wc1, n1);
		copy_list(it2, wc2, n2);
	}
	__syncthreads();
	STAMP(6);
	if (late) {
#pragma unroll
		for (int e = 0; e < E; ++e) {
			if ((late >> e) & 1) {
				if ((old[e] & 0xffffu) - l.fl[bin[e]] < SC)
					l.stage[(bin[e] << kRingShiftS) + ((old[e] >> 16) & ring)] = val[e];
				else
					ovf(bin[e], val[e]);
			}
		}
	}
	STAMP(7);
}

// flush what is staged (at most a ring = 32 entries per bin) and publish the exact entry counts
template <class OVF>
__device__ __forceinline__ void part_finish_s(const PartLdsS& l, const PartOut& o, uint32_t region, OVF&& ovf)
{
	__syncthreads();
	const uint32_t tid = threadIdx.x, ln = tid & (kRingS - 1);
	const uint32_t cap_e = o.cap * kChunk;
	for (uint32_t b = tid / kRingS; b < o.P; b += kPartThreadsS / kRingS) {
		const uint32_t w = l.pt[b], n = w & 0xffffu, hd = ((w >> 16) - n) & (kRingS - 1);
		const uint64_t base_e = (uint64_t)l.written[b] * kChunkS;
		const uint64_t o0 = (uint64_t)(b * o.regions + region) * cap_e;
		// the tail is padded to a whole 16-byte vector with copies of the last entry (see part_finish)
		if (n && ln < ((n + 3) & ~3u)) {
			const uint32_t v = l.stage[(b << kRingShiftS) + ((hd + (ln < n ? ln : n - 1)) & (kRingS - 1))];
			if (base_e + ln < cap_e)
				o.ent[o0 + base_e + ln] = v;
			else if (ln < n)
				ovf(b, v);
		}
		if (ln == 0) {
			const uint64_t full = base_e < cap_e ? base_e : cap_e;
			const uint64_t room = cap_e - full;
			o.cnt[b * o.regions + region] = (uint32_t)(full + (n < room ? n : room));
		}
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
// Level-0 bins of w = sd.bin_wseg segments (w >= 2, no power of two in general): bin = segment / w by a multiply with
// ceil(2^32 / w) -- exact for segment indices below 2^22 (error term (w * magic - 2^32) * segment < 2^10 * 2^22) --,
// entry = position - bin * width, which is below width <= 2^30, so its low 32 bits are all of it.  Four instructions
// (funnel shift, mul_hi, mul_lo, sub) against three for bit fields.
__device__ __forceinline__ void part_bin_of(const PartSide& sd, uint64_t p, uint32_t& bin, uint32_t& val)
{
	bin = __umulhi((uint32_t)(p >> sd.bin_seg_shift), sd.bin_magic);
	val = (uint32_t)p - bin * sd.bin_width;
}
// first position of level-0 bin `bin` (bins of 2^shift positions unless sd.bin_wseg)
__device__ __forceinline__ uint64_t part_bin_base(const PartSide& sd, uint32_t bin, uint32_t shift)
{
	return sd.bin_wseg ? (uint64_t)bin * sd.bin_width : (uint64_t)bin << shift;
}

// overflow of an insert / contains pass: spill list when routing, else straight to the filter
template <bool QUERY>
__device__ __forceinline__ void part_direct(uint32_t* words, const PartSide& sd, uint64_t lp)
{
	if (sd.spill_count) {
		part_spill(sd, sd.pos_base + lp);
	} else if (sd.counting) {
		if (!QUERY)
			cbf_inc_sat(words, lp);
		else if (cbf_read_fresh(words, lp) < sd.threshold)
			part_report_fail(sd, sd.pos_base + lp);
	} else if (!QUERY) {
		bf_set(words, lp);
	} else if (!((bf_word(words, lp) >> (lp & 31)) & 1u)) {
		part_report_fail(sd, sd.pos_base + lp);
	}
}


// pass A launchers, one translation unit per hash count (part_hash_inst.hip, -DBTLBF_PART_H=n)
// small != 0: the two-workgroups-per-CU geometry (512 threads, tiles of 4096 windows)
#define BTLBF_DECL_HASH_LAUNCH(n)                                                                              \
	hipError_t launch_part_hash_h##n(const SeqArgs& a, const PartOut& out, uint32_t bin_shift, const PartSide& sd, \
	                                 size_t dyn, int query, int small, hipStream_t s);
BTLBF_DECL_HASH_LAUNCH(1)
BTLBF_DECL_HASH_LAUNCH(2)
BTLBF_DECL_HASH_LAUNCH(3)
BTLBF_DECL_HASH_LAUNCH(4)
BTLBF_DECL_HASH_LAUNCH(5)
BTLBF_DECL_HASH_LAUNCH(6)
BTLBF_DECL_HASH_LAUNCH(7)
BTLBF_DECL_HASH_LAUNCH(8)
#undef BTLBF_DECL_HASH_LAUNCH

} // namespace btlbf

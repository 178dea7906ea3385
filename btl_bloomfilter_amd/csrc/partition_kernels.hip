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
// end; the tail chunk is partial and padded to a whole 16-byte vector with copies of its last entry
// (harmless for OR / test; the counting passes read exactly the published count).  Entries that do
// not fit (a region over capacity, or a bin that receives more than its ring and the space freed by
// one flush hold inside one round) take the overflow path: applied to the filter directly (single
// GPU) or appended to a spill list of global positions (multi-GPU routing) -- never dropped.
#include "partition_core.hpp"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace btlbf {

// Streamed once: what a kernel reads or writes a single time need not be kept by the caches.  Measured per launch at C2
// (profiles/r04): pass C's segment loads and write-backs as streaming accesses 4.48 -> 4.16 ms (apply) and
// 3.26 -> 3.18 (test); its entry loads 4.48 -> 4.20 (apply) but 3.26 -> 3.31 (test: plain there); pass B's entry loads:
// no change (plain).
typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load4(const uint4* p)
{
	const v4u_t v = __builtin_nontemporal_load(reinterpret_cast<const v4u_t*>(p));
	return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(uint4* p, const uint4& q)
{
	v4u_t v = {q.x, q.y, q.z, q.w};
	__builtin_nontemporal_store(v, reinterpret_cast<v4u_t*>(p));
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
// One GROUP of input bins per launch: bins [first_in, first_in + gridDim.x/slices) of the input arrays;
// the output arrays hold one group at a time (bins numbered relative to the group), which keeps the
// scratch of the split levels small.  Input bin i is bin i + abs_off of its level in absolute terms
// (the input arrays themselves are group-relative from the second split level on).
// EXACT: every entry counts (counter increments): the padded tail of a region is not taken whole.
template <bool QUERY, bool EXACT>
__global__ __launch_bounds__(kPartThreads) void part_split_kernel(void* filter, const PartIn in, const PartOut out,
                                                                 const uint32_t slices, const uint32_t sub_shift,
                                                                 const uint32_t in_shift, const uint32_t first_in,
                                                                 const uint32_t abs_off, const PartSide sd)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
	const uint32_t tid = threadIdx.x;
	const uint32_t b = first_in + blockIdx.x / slices, g = blockIdx.x % slices;
	const PartLds pl = part_carve(dyn, out.P);
	part_init<kPartThreads>(pl, out.P);
	uint32_t* words = static_cast<uint32_t*>(filter);
	const uint32_t sub_mask = (1u << sub_shift) - 1;
	const uint64_t bin_base = part_bin_base(sd, b + abs_off, in_shift);
	auto ovf = [&](uint32_t sub, uint32_t v) {
		part_direct<QUERY>(words, sd, bin_base + (((uint64_t)sub << sub_shift) | v)); // (+: a bin of wseg segments is not aligned)
	};
	const uint32_t bin0 = (b - first_in) * out.P;
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
		if (n_vec == 0)
			continue;
		// software pipeline: the next round's loads are in flight while this round is partitioned.  A vector
		// slot beyond the region re-reads the region's last vector (a plain clamped load: no branch, no
		// zero-fill); only the region's last round can have such slots, and it masks them with `live`.
		const uint32_t last_vec = n_vec - 1;
		auto fetch = [&](uint4 (&dst)[kVec], uint32_t base) {
#pragma unroll
			for (int v = 0; v < kVec; ++v) {
				const uint32_t i = base + (uint32_t)v * kPartThreads + tid;
				dst[v] = src[i < last_vec ? i : last_vec];
			}
		};
		auto round = [&](const uint4 (&cur)[kVec], uint32_t base) {
			uint32_t bin[kVec * 4], val[kVec * 4];
#pragma unroll
			for (int v = 0; v < kVec; ++v) {
				const uint32_t e4[4] = {cur[v].x, cur[v].y, cur[v].z, cur[v].w};
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					bin[v * 4 + c] = e4[c] >> sub_shift;
					val[v * 4 + c] = e4[c] & sub_mask;
				}
			}
			// EXACT: every entry counts, so the padded tail of the last vector is not taken whole
			const bool full = EXACT ? (uint64_t)(base + kPartThreads * kVec) * 4 <= n : base + kPartThreads * kVec <= n_vec;
			if (full) {
				part_round<kPartThreads, kVec * 4, EXACT ? 1 : 4, true>(pl, out, bin0, g, bin, val, 0, ovf STAMP_PASS);
			} else {
				uint32_t live = 0;
#pragma unroll
				for (int v = 0; v < kVec; ++v) {
					const uint32_t vi = base + (uint32_t)v * kPartThreads + tid;
					if (!EXACT)
						live |= (uint32_t)(vi < n_vec) << v;
#pragma unroll
					for (int c = 0; c < 4; ++c)
						if (EXACT)
							live |= (uint32_t)(vi * 4 + c < n) << (v * 4 + c);
				}
				part_round<kPartThreads, kVec * 4, EXACT ? 1 : 4, false>(pl, out, bin0, g, bin, val, live, ovf STAMP_PASS);
			}
		};
		// two register sets take turns (no copies between rounds)
		uint4 va[kVec], vb[kVec];
		fetch(va, 0);
		for (uint32_t base = 0; base < n_vec; base += 2 * kPartThreads * kVec) {
			const uint32_t base1 = base + kPartThreads * kVec;
			if (base1 < n_vec)
				fetch(vb, base1);
			round(va, base);
			if (base1 >= n_vec)
				break;
			if (base1 + kPartThreads * kVec < n_vec)
				fetch(va, base1 + kPartThreads * kVec);
			round(vb, base1);
		}
	}
	part_finish<kPartThreads>(pl, out, bin0, g, ovf);
}

// ---- pass C --------------------------------------------------------------------------------------
// one workgroup per segment (= input bin `seg`); pos_base + (seg << seg_shift | entry) is the
// position reported for failed tests
// Input bin blockIdx.x holds the entries of segment seg_first + blockIdx.x (group-relative numbering).
static constexpr uint32_t kApplyMaxRegions = 1024; // region table kept in LDS (more: plain loop)
static constexpr uint32_t kApplyFewRegions = 16;    // up to here the regions are walked one by one (uniform control flow)

// what pass C does with an entry (= position inside the segment held in LDS)
enum ApplyMode : int {
	APPLY_BIT_OR = 0,   // BloomFilter::insert
	APPLY_BIT_TEST = 1, // BloomFilter::contains: clear bit -> fail list
	APPLY_CNT_INC = 2,  // CountingBloomFilter::incrementAll: saturating +1 (CountingBloomFilter.hpp:171-181)
	APPLY_CNT_TEST = 3  // CountingBloomFilter::contains: counter < threshold -> fail list
};

template <int MODE>
__device__ __forceinline__ void apply_entry(uint32_t* lds, uint32_t e, const PartSide& sd, uint64_t seg_base)
{
	if (MODE == APPLY_BIT_OR) {
		atomicOr(&lds[e >> 5], 1u << (e & 31));
	} else if (MODE == APPLY_BIT_TEST) {
		if (!((lds[e >> 5] >> (e & 31)) & 1u))
			part_report_fail(sd, sd.pos_base + (seg_base | e));
	} else if (MODE == APPLY_CNT_INC) {
		// uint8_t counters, four to an LDS word: CAS the word, leave 255 alone
		uint32_t* w = &lds[e >> 2];
		const uint32_t sh = (e & 3) * 8;
		uint32_t old = *w;
		for (;;) {
			if (((old >> sh) & 0xffu) == 0xffu)
				break;
			const uint32_t prev = atomicCAS(w, old, old + (1u << sh));
			if (prev == old)
				break;
			old = prev;
		}
	} else {
		if (((lds[e >> 2] >> ((e & 3) * 8)) & 0xffu) < sd.threshold)
			part_report_fail(sd, sd.pos_base + (seg_base | e));
	}
}

// Memory-level parallelism is what this kernel lives on: the segment is fetched with 8 independent
// 16-byte loads per thread, and the entries of ALL regions are walked as one virtual array with 4
// independent loads per thread in flight (the first batch is requested before the segment is waited for).
// NT = 512 for 64 KiB segments (two workgroups per CU), 1024 for 128 KiB segments (one per CU).
// Every region is read up to its exact entry count (its last vector may be padded).
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void part_apply_kernel(uint8_t* filter, uint64_t local_bytes,
                                                        uint32_t seg_shift, uint32_t seg_first,
                                                        const PartIn in, const PartSide sd)
{
	constexpr bool QUERY = (MODE & 1) != 0;
	constexpr bool kNtEnt = !QUERY; // streaming entry loads pay for the passes that write the segment back (see nt_load4)
	constexpr uint32_t kUnitShift = MODE >= APPLY_CNT_INC ? 0 : 3; // log2(positions per byte)
	extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
	__shared__ uint32_t any;
	__shared__ uint32_t r_n[kApplyMaxRegions];   // entries in region r of this bin
	__shared__ uint32_t r_reg[kApplyMaxRegions]; // its index into in.cnt / in.ent
	const uint32_t tid = threadIdx.x;
	const uint32_t ibin = blockIdx.x;
	const uint32_t seg = seg_first + blockIdx.x;
	const uint32_t n_regions = in.blocks * in.regions_per_block;
	const uint32_t cap_entries = in.cap * kChunk;
	const bool tabled = n_regions <= kApplyMaxRegions;
	if (tid == 0)
		any = 0;
	__syncthreads();
	uint32_t mine = 0;
	for (uint32_t r = tid; r < n_regions; r += NT) {
		const uint32_t reg = part_in_region(in, ibin, r);
		uint32_t n = in.cnt[reg];
		if (n > cap_entries)
			n = cap_entries;
		mine |= n;
		if (tabled) {
			r_n[r] = n;
			r_reg[r] = reg;
		}
	}
	if (mine)
		any = 1;
	__syncthreads();
	if (!any && !(sd.fresh && !QUERY))
		return; // untouched segment: no traffic at all (a fresh insert still has to write its zeros)
	const uint64_t seg_bytes = 1ull << (seg_shift - kUnitShift);
	const uint64_t byte0 = (uint64_t)seg * seg_bytes;
	const uint64_t seg_base = (uint64_t)seg << seg_shift;
	uint64_t nbytes = local_bytes - byte0;
	if (nbytes > seg_bytes)
		nbytes = seg_bytes;
	const uint32_t n_vec = (uint32_t)((nbytes + 15) / 16); // the allocation is padded to 16 bytes
	uint4* lds4 = reinterpret_cast<uint4*>(dyn);
	uint4* g4 = reinterpret_cast<uint4*>(filter + byte0);
	constexpr int kSegU = 8;
	constexpr int kEntU = 4;
	// kEntU walkers per thread over the virtual concatenation of the regions' vectors
	uint32_t wr[kEntU], wi[kEntU], left[kEntU]; // region, vector inside it, entries from that vector on (0: done)
	uint4 q[kEntU];
#define BTLBF_SETTLE(r, i)                               \
	while ((r) < n_regions) {                            \
		const uint32_t nev__ = (r_n[(r)] + 3) / 4;       \
		if ((i) < nev__)                                 \
			break;                                       \
		(i) -= nev__;                                    \
		++(r);                                           \
	}
#define BTLBF_FETCH(dst, lft, u)                                                                               \
	do {                                                                                                       \
		lft = 0;                                                                                               \
		dst = make_uint4(0, 0, 0, 0);                                                                          \
		if (wr[u] < n_regions) {                                                                               \
			lft = r_n[wr[u]] - wi[u] * 4;                                                                      \
			dst = reinterpret_cast<const uint4*>(in.ent + (uint64_t)r_reg[wr[u]] * cap_entries)[wi[u]];        \
		}                                                                                                      \
	} while (0)
	// FEW regions (the usual case behind a split pass: one region per slice): they are walked one after the
	// other, kFewU vectors per thread and trip, with control flow that is uniform over the workgroup (region
	// sizes are read into scalar registers) -- an address and a load per vector instead of a per-lane search
	// for "which region is my vector in".  The next trip is requested before the current one is applied.
	constexpr int kFewU = 3;
	const bool few = tabled && n_regions <= kApplyFewRegions;
	uint4 fq[kFewU];
	uint32_t fl[kFewU];
	// next vector to request: vector f_off of region f_r.  A trip takes kFewU * NT consecutive vectors and may
	// run over the end of its region into the NEXT one (not further), so the trips stay full whatever the
	// region sizes are.
	uint32_t f_r = 0, f_off = 0;
	auto few_nvec = [&](uint32_t r) -> uint32_t {
		return ((uint32_t)__builtin_amdgcn_readfirstlane((int)r_n[r]) + 3) / 4;
	};
	auto few_skip = [&]() { // -> a region with vectors left, or f_r == n_regions
		while (f_r < n_regions && f_off >= few_nvec(f_r)) {
			++f_r;
			f_off = 0;
		}
	};
	auto few_request = [&](uint4 (&d)[kFewU], uint32_t (&lf)[kFewU]) {
		const uint32_t n0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r_n[f_r]), nv0 = (n0 + 3) / 4;
		const uint32_t reg0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r_reg[f_r]);
		const bool two = f_r + 1 < n_regions;
		const uint32_t n1 = two ? (uint32_t)__builtin_amdgcn_readfirstlane((int)r_n[two ? f_r + 1 : f_r]) : 0u, nv1 = (n1 + 3) / 4;
		const uint32_t reg1 = two ? (uint32_t)__builtin_amdgcn_readfirstlane((int)r_reg[two ? f_r + 1 : f_r]) : reg0;
		const uint4* src0 = reinterpret_cast<const uint4*>(in.ent + (uint64_t)reg0 * cap_entries);
		const uint4* src1 = reinterpret_cast<const uint4*>(in.ent + (uint64_t)reg1 * cap_entries);
#pragma unroll
		for (int u = 0; u < kFewU; ++u) {
			const uint32_t v = f_off + (uint32_t)u * NT + tid; // vector index counted from the start of region f_r
			lf[u] = 0;
			d[u] = make_uint4(0, 0, 0, 0);
			if (v < nv0) {
				d[u] = kNtEnt ? nt_load4(src0 + v) : src0[v];
				lf[u] = n0 - v * 4;
			} else if (v - nv0 < nv1) {
				d[u] = kNtEnt ? nt_load4(src1 + (v - nv0)) : src1[v - nv0];
				lf[u] = n1 - (v - nv0) * 4;
			}
		}
		const uint32_t end = f_off + (uint32_t)(kFewU * NT);
		if (end <= nv0) {
			f_off = end;
		} else if (two && end - nv0 <= nv1) {
			++f_r;
			f_off = end - nv0;
		} else { // the trip ended with region f_r + 1 (or there is none)
			f_r += 2;
			f_off = 0;
		}
		few_skip();
	};
	bool few_have = false;
	if (few) {
		few_skip();
		few_have = f_r < n_regions;
		if (few_have)
			few_request(fq, fl);
	} else if (tabled) {
#pragma unroll
		for (int u = 0; u < kEntU; ++u) {
			wr[u] = 0;
			wi[u] = (uint32_t)u * NT + tid;
			BTLBF_SETTLE(wr[u], wi[u])
			BTLBF_FETCH(q[u], left[u], u);
		}
	}
	const bool fresh = sd.fresh && !QUERY; // the array is known to be zero: nothing to fetch
	for (uint32_t base = 0; base < n_vec; base += NT * kSegU) {
		uint4 v[kSegU];
#pragma unroll
		for (int u = 0; u < kSegU; ++u) {
			const uint32_t i = base + (uint32_t)u * NT + tid;
			v[u] = make_uint4(0, 0, 0, 0);
			if (i < n_vec && !fresh)
				v[u] = nt_load4(g4 + i);
		}
#pragma unroll
		for (int u = 0; u < kSegU; ++u) {
			const uint32_t i = base + (uint32_t)u * NT + tid;
			if (i < n_vec)
				lds4[i] = v[u];
		}
	}
	__syncthreads();
	uint32_t* lds = reinterpret_cast<uint32_t*>(dyn);
	if (few) {
		while (few_have) {
			uint4 nq[kFewU];
			uint32_t nl[kFewU];
			const bool more = f_r < n_regions;
			if (more)
				few_request(nq, nl);
#pragma unroll
			for (int u = 0; u < kFewU; ++u) {
				if (MODE < APPLY_CNT_INC) {
					if (fl[u] > 0) { // whole vectors: see below
						apply_entry<MODE>(lds, fq[u].x, sd, seg_base);
						apply_entry<MODE>(lds, fq[u].y, sd, seg_base);
						apply_entry<MODE>(lds, fq[u].z, sd, seg_base);
						apply_entry<MODE>(lds, fq[u].w, sd, seg_base);
					}
				} else {
					if (fl[u] > 0)
						apply_entry<MODE>(lds, fq[u].x, sd, seg_base);
					if (fl[u] > 1)
						apply_entry<MODE>(lds, fq[u].y, sd, seg_base);
					if (fl[u] > 2)
						apply_entry<MODE>(lds, fq[u].z, sd, seg_base);
					if (fl[u] > 3)
						apply_entry<MODE>(lds, fq[u].w, sd, seg_base);
				}
				if (more) {
					fq[u] = nq[u];
					fl[u] = nl[u];
				}
			}
			few_have = more;
		}
	} else if (tabled) {
		while (left[0]) {
			uint4 nq[kEntU];
			uint32_t nleft[kEntU];
#pragma unroll
			for (int u = 0; u < kEntU; ++u) {
				wi[u] += kEntU * NT;
				BTLBF_SETTLE(wr[u], wi[u])
				BTLBF_FETCH(nq[u], nleft[u], u);
			}
#pragma unroll
			for (int u = 0; u < kEntU; ++u) {
				if (MODE < APPLY_CNT_INC) {
					// bit filter: a region's last vector is padded with copies of its last entry (part_finish),
					// and a repeated position changes nothing for OR / test -- whole vectors, one test
					if (left[u] > 0) {
						apply_entry<MODE>(lds, q[u].x, sd, seg_base);
						apply_entry<MODE>(lds, q[u].y, sd, seg_base);
						apply_entry<MODE>(lds, q[u].z, sd, seg_base);
						apply_entry<MODE>(lds, q[u].w, sd, seg_base);
					}
				} else {
					if (left[u] > 0)
						apply_entry<MODE>(lds, q[u].x, sd, seg_base);
					if (left[u] > 1)
						apply_entry<MODE>(lds, q[u].y, sd, seg_base);
					if (left[u] > 2)
						apply_entry<MODE>(lds, q[u].z, sd, seg_base);
					if (left[u] > 3)
						apply_entry<MODE>(lds, q[u].w, sd, seg_base);
				}
				q[u] = nq[u];
				left[u] = nleft[u];
			}
		}
#undef BTLBF_SETTLE
#undef BTLBF_FETCH
	} else {
		for (uint32_t r = 0; r < n_regions; ++r) {
			const uint32_t reg = part_in_region(in, ibin, r);
			uint32_t n = in.cnt[reg];
			if (n > cap_entries)
				n = cap_entries;
			const uint32_t* e1 = in.ent + (uint64_t)reg * cap_entries;
			for (uint32_t i = tid; i < n; i += NT)
				apply_entry<MODE>(lds, e1[i], sd, seg_base);
		}
	}
	if (QUERY)
		return; // read-only sweep
	__syncthreads();
	for (uint32_t base = 0; base < n_vec; base += NT * kSegU) {
#pragma unroll
		for (int u = 0; u < kSegU; ++u) {
			const uint32_t i = base + (uint32_t)u * NT + tid;
			if (i < n_vec)
				nt_store4(g4 + i, lds4[i]);
		}
	}
}

// ---- launchers -----------------------------------------------------------------------------------
// How pass A's front end is set up for one (hash configuration, level-0 bins): geometry, whether the
// positional seed table is used, and the dynamic LDS all of it needs.
struct PartFront {
	bool small = false;  // two 512-thread workgroups per CU (partition_core.hpp)
	uint32_t nt = kPartThreads;
	uint32_t use_pos_tab = 0;
	uint32_t dyn = 0;
};

// The small geometry is opt-in (BTLBF_PART_GEOM=small): measured on MI355X at C2 it is no faster than one
// 1024-thread workgroup -- pass A is bound by VALU issue, not by the latencies a second workgroup would
// hide.
static bool part_want_small(uint32_t p0)
{
	if (p0 < kPartMinBinsS || p0 > kPartMaxBinsS)
		return false;
	const char* e = getenv("BTLBF_PART_GEOM");
	return e && !strcmp(e, "small");
}

// with the positional seed table if it fits, else without (Horner start-up; spaced seeds cannot do without it)
static bool part_front(const HashParams& hp_in, uint32_t p0, PartFront& fr, uint32_t tile_lds = 0)
{
	for (int geom = part_want_small(p0) ? 1 : 0; geom >= 0; --geom) {
		fr.small = geom == 1;
		fr.nt = fr.small ? kPartThreadsS : kPartThreads;
		const uint32_t rings = fr.small ? part_lds_bytes_s(p0) : part_lds_bytes(p0);
		const uint32_t budget = fr.small ? kPartLdsBudgetS : kPartLdsBudget;
		for (int tab = 1; tab >= 0; --tab) {
			HashParams hp = hp_in;
			if (!tab) {
				if (hp.n_seeds || !hp.use_pos_tab)
					break;
				hp.use_pos_tab = 0;
			}
			const uint32_t dyn = (tile_lds && !fr.small ? tile_lds : seq_tile_cap(fr.nt * kPartW, hp.k)) + seq_spaced_bytes(hp) + rings;
			if (dyn <= budget) {
				fr.use_pos_tab = hp.use_pos_tab;
				fr.dyn = dyn;
				return true;
			}
		}
	}
	return false;
}

// can pass A run for this hash configuration at all?
bool part_hash_fits(const HashParams& hp, uint32_t p0)
{
	PartFront fr;
	return part_front(hp, p0, fr);
}

// workgroups (= regions per bin) pass A wants for `cus` compute units
uint32_t part_hash_regions(const HashParams& hp, uint32_t p0, uint32_t cus)
{
	PartFront fr;
	return part_front(hp, p0, fr) && fr.small ? 2 * cus : cus;
}

// The read grid for (hash configuration, level-0 bins, layout), if it pays: uniform reads, the 1024-thread
// geometry, the tile image fits the LDS next to the rings, and the lanes it keeps busy beat plain tiles by 5 %.
// BTLBF_READ_GRID=0 turns it off (measurements).
bool part_read_grid(const HashParams& hp, uint32_t p0, const LayoutParams& lay, PartGrid* g)
{
	*g = PartGrid();
	const uint32_t L = lay.starts ? 0 : lay.read_len, k = hp.k;
	if (L < 16 || L < k || L > 4096 || part_want_small(p0))
		return false;
	if (const char* e = getenv("BTLBF_READ_GRID"))
		if (!strcmp(e, "0"))
			return false;
	const uint32_t wins = L - k + 1, gpr = (wins + 7) / 8;
	uint32_t step = 8; // reads per tile: a multiple of `step` keeps a tile's window bitmap in whole bytes
	while (step > 1 && ((uint64_t)(step / 2) * L) % 8 == 0)
		step /= 2;
	// every thread stages kPartW/4+1 words of a tile
	const uint32_t max_bytes = (kPartW / 4 + 1) * kPartThreads * 4 - 16;
	uint32_t reads = std::min<uint32_t>(kPartThreads / gpr, max_bytes / L);
	if (const char* e = getenv("BTLBF_GRID_READS")) { // measurements: fewer reads per tile than fit
		const uint32_t v = (uint32_t)atoi(e);
		if (v >= 1 && v < reads)
			reads = v;
	}
	reads -= reads % step;
	if (reads == 0)
		return false;
	const double eff_grid = (double)reads * wins / (kPartThreads * kPartW), eff_plain = (double)wins / L;
	if (eff_grid < 1.05 * eff_plain)
		return false;
	const uint32_t lpad = (L + 7) / 8 * 8;
	const uint32_t bitmap = ((reads * L / 8 + 15) / 16) * 16; // one bit per window start of the tile
	const uint32_t cap = ((reads * lpad + k + 8 + 15) / 16) * 16 + 2 * bitmap;
	PartFront fr, plain;
	if (!part_front(hp, p0, fr, cap) || fr.small || !part_front(hp, p0, plain) || fr.use_pos_tab != plain.use_pos_tab)
		return false; // (not at the price of the positional seed table)
	g->reads = reads;
	g->gpr = gpr;
	g->lpad = lpad;
	g->cap = cap;
	return true;
}

// how pass A cuts a buffer into tiles; all host-side planning is in these units
PartTiling part_tiling(const HashParams& hp, uint32_t p0, const LayoutParams& lay, uint64_t len)
{
	PartFront fr;
	if (!part_front(hp, p0, fr))
		fr = PartFront();
	PartTiling t;
	const uint32_t L = lay.starts ? 0 : lay.read_len;
	PartGrid g;
	if (part_read_grid(hp, p0, lay, &g)) {
		t.tile_bytes = g.reads * L;
		t.n_tiles = (len + t.tile_bytes - 1) / t.tile_bytes;
		t.windows_per_tile = (double)g.reads * (L - hp.k + 1);
		return t;
	}
	t.tile_bytes = fr.nt * kPartW;
	t.n_tiles = (len + t.tile_bytes - 1) / t.tile_bytes;
	t.windows_per_tile = (double)t.tile_bytes * (L ? (L >= hp.k ? (double)(L - hp.k + 1) / L : 0.0) : 1.0);
	return t;
}

static hipError_t launch_hash_any(const SeqArgs& a, const PartOut& out, uint32_t bin_shift, const PartSide& sd,
                                  size_t dyn, int query, int small, hipStream_t s)
{
	switch (a.hp.h) {
	case 1: return launch_part_hash_h1(a, out, bin_shift, sd, dyn, query, small, s);
	case 2: return launch_part_hash_h2(a, out, bin_shift, sd, dyn, query, small, s);
	case 3: return launch_part_hash_h3(a, out, bin_shift, sd, dyn, query, small, s);
	case 4: return launch_part_hash_h4(a, out, bin_shift, sd, dyn, query, small, s);
	case 5: return launch_part_hash_h5(a, out, bin_shift, sd, dyn, query, small, s);
	case 6: return launch_part_hash_h6(a, out, bin_shift, sd, dyn, query, small, s);
	case 7: return launch_part_hash_h7(a, out, bin_shift, sd, dyn, query, small, s);
	case 8: return launch_part_hash_h8(a, out, bin_shift, sd, dyn, query, small, s);
	default: return hipErrorInvalidValue;
	}
}

// 1..8 hashes per k-mer; spaced seeds only with the union list of don't-care offsets (HashParams::dcu: at most 64
// distinct offsets -- the hash stage keeps the list in one register of a wave); anything else stays on the direct kernels
bool part_supported(const HashParams& hp) { return hp.h >= 1 && hp.h <= 8 && (hp.n_seeds == 0 || hp.n_dcu > 0); }

// pass A over tiles [a.first_tile, +a.n_tiles) in the units of part_tiling() for this buffer; exactly
// out.regions workgroups are launched (one region each; idle ones still publish empty counts)
hipError_t launch_part_hash(const SeqArgs& a_in, const PartOut& out, uint32_t bin_shift, const PartSide& sd,
                            int query, hipStream_t s)
{
	SeqArgs a = a_in;
	if (a.n_tiles == 0)
		return hipSuccess;
	if (!part_supported(a.hp))
		return hipErrorInvalidValue;
	PartFront fr;
	PartGrid g;
	const bool grid = part_read_grid(a.hp, out.P, a.layout, &g);
	if (a.read_mask && !grid) // the mask is one bit per read of the grid: the caller planned with other bins than it runs with
		return hipErrorInvalidValue;
	// ragged layout: room for the start bitmap of the overlapped schedule behind the tile image, if the LDS has it
	// without giving up the positional table (part_hash_inst.hip; the plain kernels leave it unused)
	a.sb_words = 0;
	if (a.layout.starts && !grid && !part_want_small(out.P)) {
		PartFront plain;
		const uint32_t words = ((uint32_t)kPartTile + a.hp.k + 2 + 31) / 32, cap = seq_tile_cap(kPartTile, a.hp.k);
		const uint32_t sb_bytes = (words * 4 + 15) / 16 * 16;
		if (part_front(a.hp, out.P, plain) && part_front(a.hp, out.P, fr, cap + sb_bytes) && !fr.small &&
		    fr.use_pos_tab == plain.use_pos_tab)
			a.sb_words = sb_bytes / 4;
	}
	const uint32_t tile_lds = grid ? g.cap : a.sb_words ? seq_tile_cap(kPartTile, a.hp.k) + a.sb_words * 4 : 0;
	a.hp.n_pair_rows = 0; // (everything above was planned without them)
	if (!part_front(a.hp, out.P, fr, tile_lds))
		return hipErrorInvalidValue;
	// spaced seeds: two-base rows for the union list's pairs (seq_core.hpp) where the LDS still has the room -- with 256
	// level-0 bins it has (3 KB fewer per-bin words than with 512), and nothing else gives way for them
	if (a.hp.want_pair_rows && !fr.small) {
		HashParams with = a.hp;
		with.n_pair_rows = a.hp.want_pair_rows;
		PartFront fr2;
		if (part_front(with, out.P, fr2, tile_lds) && !fr2.small && fr2.use_pos_tab == fr.use_pos_tab) {
			a.hp.n_pair_rows = with.n_pair_rows;
			fr = fr2;
		}
	}
	a.rg_reads = g.reads;
	a.rg_gpr = g.gpr;
	a.rg_lpad = g.lpad;
	a.rg_cap = g.cap;
	a.rg_lpad_inv = g.lpad ? 0xffffffffu / g.lpad + 1 : 0;
	a.hp.use_pos_tab = fr.use_pos_tab;
	a.tiles_per_block = (a.n_tiles + out.regions - 1) / out.regions;
	const size_t dyn = fr.dyn;
	return launch_hash_any(a, out, bin_shift, sd, dyn, query, fr.small, s);
}

hipError_t launch_part_split(void* filter, const PartIn& in, uint32_t first_in, uint32_t abs_first, uint32_t n_in_bins,
                             const PartOut& out, uint32_t sub_shift, uint32_t in_shift, const PartSide& sd, int query,
                             int exact, hipStream_t s)
{
	if (n_in_bins == 0)
		return hipSuccess;
	const size_t dyn = part_lds_bytes(out.P);
	const uint32_t slices = out.regions;
	const dim3 grid(n_in_bins * slices);
#define BTLBF_SLAUNCH(Q, E)                                                                                    \
	do {                                                                                                       \
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&part_split_kernel<Q, E>),            \
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);              \
		if (e != hipSuccess)                                                                                   \
			return e;                                                                                          \
		hipLaunchKernelGGL((part_split_kernel<Q, E>), grid, dim3(kPartThreads), dyn, s, filter, in, out, slices, \
		                   sub_shift, in_shift, first_in, abs_first - first_in, sd);                            \
	} while (0)
	if (query && exact)
		BTLBF_SLAUNCH(true, true);
	else if (query)
		BTLBF_SLAUNCH(true, false);
	else if (exact)
		BTLBF_SLAUNCH(false, true);
	else
		BTLBF_SLAUNCH(false, false);
#undef BTLBF_SLAUNCH
	return hipGetLastError();
}

hipError_t launch_part_apply(void* filter, uint64_t local_bytes, uint32_t seg_shift, uint64_t seg_first, uint64_t n_seg,
                             const PartIn& in, const PartSide& sd, int query, hipStream_t s)
{
	if (n_seg == 0)
		return hipSuccess;
	const int mode = (sd.counting ? APPLY_CNT_INC : APPLY_BIT_OR) + (query ? 1 : 0);
	const size_t dyn = (size_t)1 << (seg_shift - (sd.counting ? 0 : 3)); // bytes of one segment
#define BTLBF_ALAUNCH(M, NT)                                                                                   \
	do {                                                                                                       \
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&part_apply_kernel<M, NT>),           \
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);              \
		if (e != hipSuccess)                                                                                   \
			return e;                                                                                          \
		hipLaunchKernelGGL((part_apply_kernel<M, NT>), dim3((unsigned)n_seg), dim3(NT), dyn, s,                \
		                   static_cast<uint8_t*>(filter), local_bytes, seg_shift, (uint32_t)seg_first, in, sd); \
	} while (0)
#define BTLBF_ALAUNCH_M(NT)                      \
	do {                                         \
		switch (mode) {                          \
		case APPLY_BIT_OR:                       \
			BTLBF_ALAUNCH(APPLY_BIT_OR, NT);     \
			break;                               \
		case APPLY_BIT_TEST:                     \
			BTLBF_ALAUNCH(APPLY_BIT_TEST, NT);   \
			break;                               \
		case APPLY_CNT_INC:                      \
			BTLBF_ALAUNCH(APPLY_CNT_INC, NT);    \
			break;                               \
		default:                                 \
			BTLBF_ALAUNCH(APPLY_CNT_TEST, NT);   \
			break;                               \
		}                                        \
	} while (0)
	if (dyn > 64 * 1024) // 128 KiB segments: one workgroup per CU
		BTLBF_ALAUNCH_M(1024);
	else
		BTLBF_ALAUNCH_M(512);
#undef BTLBF_ALAUNCH_M
#undef BTLBF_ALAUNCH
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

// ---- the same, steered from the device (no host round trip between a batch's test pass and its resolve step) ----
__global__ void failset_plan_kernel(const unsigned long long* fail_count, uint64_t fail_cap, uint64_t max_slots, uint64_t* ctl)
{
	const uint64_t n = *fail_count;
	uint64_t slots = 1024; // sized to the set (load <= 1/4): a small table stays in L2 while every probe is looked up in it
	while (slots < 4 * n && slots < max_slots)
		slots <<= 1;
	ctl[0] = slots - 1;
	ctl[1] = (n == 0 ? (uint64_t)GATE_NONE : n > fail_cap ? (uint64_t)GATE_REDO : (uint64_t)GATE_RESOLVE) | (n << 32);
}
__global__ __launch_bounds__(256) void failset_clear_auto_kernel(unsigned long long* table, const uint64_t* ctl)
{
	if ((uint32_t)ctl[1] != GATE_RESOLVE)
		return;
	const uint64_t slots = ctl[0] + 1;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (uint64_t)gridDim.x * blockDim.x)
		table[i] = 0;
}
__global__ __launch_bounds__(256) void failset_build_auto_kernel(const uint64_t* list, unsigned long long* table,
                                                                const uint64_t* ctl)
{
	if ((uint32_t)ctl[1] != GATE_RESOLVE)
		return;
	const uint64_t n = ctl[1] >> 32, mask = ctl[0];
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

hipError_t launch_failset_auto(const uint64_t* fail_list, const unsigned long long* fail_count, uint64_t fail_cap,
                               uint64_t* table, uint64_t max_slots, uint64_t* ctl, hipStream_t s)
{
	hipLaunchKernelGGL(failset_plan_kernel, dim3(1), dim3(1), 0, s, fail_count, fail_cap, max_slots, ctl);
	hipLaunchKernelGGL(failset_clear_auto_kernel, dim3(512), dim3(256), 0, s, reinterpret_cast<unsigned long long*>(table), ctl);
	hipLaunchKernelGGL(failset_build_auto_kernel, dim3(512), dim3(256), 0, s, fail_list,
	                   reinterpret_cast<unsigned long long*>(table), ctl);
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
		if (sd.counting) {
			if (!test)
				cbf_inc_sat(words, lp);
			else if (cbf_read_fresh(words, lp) < sd.threshold)
				part_report_fail(sd, pos[i]);
		} else if (!test) {
			bf_set(words, lp);
		} else if (!((bf_word(words, lp) >> (lp & 31)) & 1u)) {
			part_report_fail(sd, pos[i]);
		}
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

// csrc/seq_kernels.hip -- the hot path: ntHash k-mer stream fused with the filter probes.
//
// One launch walks a sequence buffer of `len` bytes.  A workgroup (256 threads = 4 wavefronts)
// owns a contiguous run of tiles; a tile is kTile = 2048 consecutive window start offsets.
//   1. stage: coalesced 16-byte loads of the tile's kTile+k-1 bytes; each byte goes through a
//      256-entry LDS table to a 2-bit base code (3 bits with the raw-byte aliases) + "valid base" flag and is written back to LDS;
//      sequence starts inside the tile get a "start" flag (read_len arithmetic or the starts[] array).
//   2. hash: each lane owns kW = 8 consecutive windows: Horner start-up of the forward and reverse
//      hashes over the first window (k LDS reads), then 7 O(1) rolls.  A window is clean iff its
//      first base is valid and the k-1 following bases are valid and not sequence starts; that
//      count rolls along with the hash.
//   3. probe: canonical hash -> h hashes -> positions -> atomicOr / load / byte CAS on the
//      HBM-resident array.  All kW*h probes of a lane are issued before the first is consumed.
//   4. results: the 8 windows of a lane are exactly one byte of the per-window bitmaps, so hit /
//      valid bits leave as coalesced byte stores; __ballot + popcount feed the hit counters.
//
// Reference semantics reproduced (not its code): ntHashIterator init/next
// (vendor/ntHashIterator.hpp:59-86), NTMC64/NTMSM64 (vendor/nthash.hpp:581-590,667-692,820-878),
// BloomFilter insert/contains/insertAndCheck (BloomFilter.hpp:185-262), CountingBloomFilter
// minCount/incrementMin/incrementAll/contains (CountingBloomFilter.hpp:53-64,135-196).
#include "device_utils.hpp"

namespace btlbf {

static constexpr int kThreads = 256;
static constexpr int kW = 8;
static constexpr int kTile = kThreads * kW;

int seq_tile_windows() { return kTile; }

struct __attribute__((aligned(16))) U64x2 {
	uint64_t x, y;
};

// static LDS: translation table + hash tables
struct SeqShared {
	U64x2 init_tab[kNumCodes];
	U64x2 in_tab[kNumCodes];
	U64x2 out_tab[kNumCodes];
	uint8_t lut[256];
	unsigned long long cnt_valid;
	unsigned long long cnt_hit;
	uint32_t hist[64];      // OP_POSITIONS: per-shard count inside the workgroup
	uint64_t hist_base[64]; // OP_POSITIONS: reserved base in the global bucket
	uint64_t start_lo;      // ragged layout: first starts[] index that can fall inside the tile
};

// ASCII byte -> code | valid (what vendor/nthash.hpp:195-228 accepts: ACGTU acgtu and 1 3 4 5 7)
__device__ __forceinline__ uint8_t base_entry(uint32_t c)
{
	switch (c) {
	case 'A': case 'a': return 0 | kBaseValid;
	case 'C': case 'c': return 1 | kBaseValid;
	case 'G': case 'g': return 2 | kBaseValid;
	case 'T': case 't': case 'U': case 'u': return 3 | kBaseValid;
	case 4: case 5: return 4 | kBaseValid; // raw bytes: forward A C G T, reverse seed = forward seed
	case 7: return 5 | kBaseValid;
	case 3: return 6 | kBaseValid;
	case 1: return 7 | kBaseValid;
	default: return 0;
	}
}

template <int OP, bool POW2, bool SPACED>
__global__ __launch_bounds__(kThreads) void seq_kernel(const SeqArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
	__shared__ SeqShared sh;

	const uint32_t tid = threadIdx.x;
	const uint32_t k = a.hp.k;
	const uint32_t h = a.hp.h;
	// dynamic LDS carve: [tile bytes | spaced-seed position table | don't-care index list]
	const uint32_t tile_cap = ((kTile + k - 1 + 15 + 15) / 16) * 16; // + up to 15 bytes of misalignment
	uint8_t* tile = dyn;
	const U64x2* pos_tab = reinterpret_cast<const U64x2*>(dyn + tile_cap);
	const uint16_t* dc_idx = reinterpret_cast<const uint16_t*>(dyn + tile_cap + (SPACED ? k * kNumCodes * 16 : 0));

	// ---- one-time table setup ----
	sh.lut[tid] = base_entry(tid);
	if (tid < kNumCodes) {
		sh.init_tab[tid] = U64x2{a.hp.init_tab[tid][0], a.hp.init_tab[tid][1]};
		sh.in_tab[tid] = U64x2{a.hp.in_tab[tid][0], a.hp.in_tab[tid][1]};
		sh.out_tab[tid] = U64x2{a.hp.out_tab[tid][0], a.hp.out_tab[tid][1]};
	}
	if (tid == 0) {
		sh.cnt_valid = 0;
		sh.cnt_hit = 0;
	}
	if (SPACED) {
		uint64_t* pt = reinterpret_cast<uint64_t*>(dyn + tile_cap);
		for (uint32_t i = tid; i < k * kNumCodes * 2; i += kThreads)
			pt[i] = a.hp.pos_tab[i];
		uint16_t* di = reinterpret_cast<uint16_t*>(dyn + tile_cap + k * kNumCodes * 16);
		const uint32_t ndc = a.hp.dc_off[a.hp.n_seeds];
		for (uint32_t i = tid; i < ndc; i += kThreads)
			di[i] = a.hp.dc_idx[i];
	}

	const uint64_t t_begin = (uint64_t)blockIdx.x * a.tiles_per_block;
	uint64_t t_end = t_begin + a.tiles_per_block;
	if (t_end > a.n_tiles)
		t_end = a.n_tiles;
	const uint32_t L = a.layout.read_len;
	const uint64_t* starts = a.layout.starts;
	// offset of the tile's first byte inside its read (uniform layout), kept incrementally
	uint32_t tile_off = 0;
	if (!starts && L && t_begin < t_end)
		tile_off = (uint32_t)((t_begin * (uint64_t)kTile) % L);
	const uint32_t tile_step = L ? (uint32_t)(kTile % L) : 0;
	const uint64_t out_bytes = ((a.len + 63) / 64) * 8;

	uint32_t my_valid = 0, my_hit = 0;

	for (uint64_t t = t_begin; t < t_end; ++t) {
		const uint64_t g0 = t * (uint64_t)kTile;
		const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(a.seq + g0) & 15);
		uint64_t need = a.len - g0;
		if (need > (uint64_t)(kTile + k - 1))
			need = kTile + k - 1;
		const uint32_t n_chunks = (mis + (uint32_t)need + 15) / 16;
		__syncthreads(); // previous tile fully consumed (and tables written, first time round)

		// ---- 1. stage (chunks past the data are zero-filled so no stale flags survive) ----
		for (uint32_t j = tid; j < tile_cap / 16; j += kThreads) {
			uint4 raw = make_uint4(0, 0, 0, 0);
			if (j < n_chunks)
				raw = *reinterpret_cast<const uint4*>(a.seq + g0 - mis + 16ull * j);
			const uint32_t wv[4] = {raw.x, raw.y, raw.z, raw.w};
			// position of this chunk's first byte relative to g0 (negative for the misaligned head)
			const int32_t rel0 = (int32_t)(16 * j) - (int32_t)mis;
			uint32_t r = 0; // (offset within read) of the chunk's first byte, uniform layout only
			if (!starts && L) {
				const uint32_t m = mis % L;
				r = ((tile_off + 16 * j) % L + L - m) % L;
			}
			uint32_t outw[4];
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				uint32_t o = 0;
#pragma unroll
				for (int b = 0; b < 4; ++b) {
					const int32_t rel = rel0 + q * 4 + b;
					uint32_t e = sh.lut[(wv[q] >> (8 * b)) & 0xff];
					if (rel < 0 || (uint64_t)rel >= need)
						e = 0;
					if (!starts && L) {
						if (r == 0)
							e |= kBaseStart;
						r = (r + 1 == L) ? 0 : r + 1;
					}
					o |= e << (8 * b);
				}
				outw[q] = o;
			}
			*reinterpret_cast<uint4*>(tile + 16 * j) = make_uint4(outw[0], outw[1], outw[2], outw[3]);
		}
		if (starts) {
			// first index s with starts[s] > g0: every boundary strictly inside (g0, g0+need) matters
			if (tid == 0) {
				uint64_t lo = 0, hi = a.layout.n_seqs + 1;
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
			for (uint64_t s = s_lo + tid; s <= a.layout.n_seqs; s += kThreads) {
				const uint64_t p = starts[s];
				if (p >= g0 + need)
					break;
				const uint32_t li = (uint32_t)(p - g0) + mis;
				atomicOr(reinterpret_cast<uint32_t*>(tile) + (li >> 2), kBaseStart << (8 * (li & 3)));
			}
		}
		tile_off += tile_step;
		if (L && tile_off >= L)
			tile_off -= L;
		__syncthreads();

		// ---- 2. hash (lane owns windows w0 .. w0+kW-1 of the tile) ----
		const uint32_t li0 = tid * kW + mis;
		uint64_t fh = 0, rh = 0;
		uint32_t cnt = 0; // valid, non-start bases among the k-1 bases after the window's first
		uint32_t first_valid;
		{
			uint32_t e = tile[li0];
			first_valid = (e >> 3) & 1;
			U64x2 tt = sh.init_tab[e & kCodeMask];
			fh = tt.x;
			rh = tt.y;
			for (uint32_t i = 1; i < k; ++i) {
				e = tile[li0 + i];
				tt = sh.init_tab[e & kCodeMask];
				fh = srol1(fh) ^ tt.x;
				rh = sror1(rh) ^ tt.y;
				cnt += ((e & (kBaseValid | kBaseStart)) == kBaseValid);
			}
		}

		uint32_t valid_mask = 0, hit_mask = 0;
		// contains() with <= kPipe hashes keeps every probe of the lane's kW windows in flight at
		// once (kW*h independent loads); larger hash counts take the per-window loop below.
		constexpr int kPipe = 4;
		uint32_t wordbuf[kW][kPipe];
		uint32_t bitbuf[kW][kPipe];
		(void)wordbuf;
		(void)bitbuf;
		const bool pipelined = (OP == OP_BF_CONTAINS) && !SPACED && h <= kPipe;

#pragma unroll
		for (int w = 0; w < kW; ++w) {
			if (w > 0) {
				const uint32_t eo = tile[li0 + w - 1];
				const uint32_t ei = tile[li0 + w - 1 + k];
				const uint32_t en = tile[li0 + w];
				const U64x2 ti = sh.in_tab[ei & kCodeMask];
				const U64x2 to = sh.out_tab[eo & kCodeMask];
				fh = srol1(fh) ^ ti.x ^ to.x;
				rh = sror1(rh ^ ti.y ^ to.y);
				cnt += ((ei & (kBaseValid | kBaseStart)) == kBaseValid);
				cnt -= ((en & (kBaseValid | kBaseStart)) == kBaseValid);
				first_valid = (en >> 3) & 1;
			}
			const bool ok = first_valid && cnt == k - 1;
			valid_mask |= (uint32_t)ok << w;
			const uint64_t gp = g0 + tid * kW + w; // byte offset of this window

			// ---- hash values of this window ----
			// plain ntHash: hash i is recomputed from the canonical value where needed (a multiply
			// and a shift) so nothing is indexed dynamically; spaced seeds keep an array.
			uint64_t hv[SPACED ? kMaxHash : 1];
			uint32_t stn = 0;
			const uint64_t bcan = rh < fh ? rh : fh;
#define HASH_AT(i) (SPACED ? hv[SPACED ? (i) : 0] : ((i) ? extra_hash(bcan, a.hp.kms, (i)) : bcan))
			if (SPACED) {
				const uint32_t h2 = a.hp.h2;
				for (uint32_t j = 0; j < a.hp.n_seeds; ++j) {
					uint64_t fs = fh, rs = rh;
					for (uint32_t d = a.hp.dc_off[j]; d < a.hp.dc_off[j + 1]; ++d) {
						const uint32_t i = dc_idx[d];
						const U64x2 tt = pos_tab[i * kNumCodes + (tile[li0 + w + i] & kCodeMask)];
						fs ^= tt.x;
						rs ^= tt.y;
					}
					const bool s = rs < fs;
					const uint64_t b = s ? rs : fs;
					hv[j * h2] = b;
					for (uint32_t j2 = 1; j2 < h2; ++j2)
						hv[j * h2 + j2] = extra_hash(b, a.hp.kms, j2);
					if (s)
						for (uint32_t j2 = 0; j2 < h2; ++j2)
							stn |= 1u << (j * h2 + j2); // h <= 32 here
				}
			}

			// ---- 3. probe ----
			if (OP == OP_HASH_ONLY) {
				if (gp < a.len) {
					for (uint32_t i = 0; i < h; ++i)
						a.hashes[gp * h + i] = ok ? HASH_AT(i) : 0;
					if (a.strand_bits)
						a.strand_bits[gp] = ok ? stn : 0;
				}
			} else if (OP == OP_BF_INSERT) {
				if (ok) {
					uint32_t* words = static_cast<uint32_t*>(a.filter);
					for (uint32_t i = 0; i < h; ++i) {
						const uint64_t p = reduce_mod<POW2>(HASH_AT(i), a.mod) - a.mod.shard_lo;
						if (p < a.mod.shard_len)
							bf_set(words, p);
					}
				}
			} else if (OP == OP_BF_INSERT_CHECK) {
				if (ok) {
					uint32_t* words = static_cast<uint32_t*>(a.filter);
					uint32_t all = 1;
					for (uint32_t i = 0; i < h; ++i) {
						const uint64_t p = reduce_mod<POW2>(HASH_AT(i), a.mod);
						all &= bf_set_fetch(words, p);
					}
					hit_mask |= all << w;
				}
			} else if (OP == OP_BF_CONTAINS) {
				const uint32_t* words = static_cast<const uint32_t*>(a.filter);
				if (pipelined) {
#pragma unroll
					for (int i = 0; i < kPipe; ++i) {
						if ((uint32_t)i < h) {
							const uint64_t p = reduce_mod<POW2>(HASH_AT(i), a.mod);
							// unclean windows probe word 0 (harmless) so the loop stays branch-free
							wordbuf[w][i] = bf_word(words, ok ? p : 0);
							bitbuf[w][i] = (uint32_t)p & 31;
						}
					}
				} else if (ok) {
					uint32_t all = 1;
					for (uint32_t i = 0; i < h; ++i) {
						const uint64_t p = reduce_mod<POW2>(HASH_AT(i), a.mod);
						all &= (bf_word(words, p) >> (p & 31)) & 1u;
					}
					hit_mask |= all << w;
				}
			} else if (OP == OP_CBF_QUERY) {
				const uint8_t* ctr = static_cast<const uint8_t*>(a.filter);
				uint32_t mn = 0xff;
				if (ok) {
					for (uint32_t i = 0; i < h; ++i) {
						const uint32_t v = ctr[reduce_mod<POW2>(HASH_AT(i), a.mod)];
						mn = v < mn ? v : mn;
					}
					hit_mask |= (uint32_t)(mn >= a.threshold) << w;
				}
				if (a.min_out && gp < a.len)
					a.min_out[gp] = ok ? (uint8_t)mn : 0;
			} else if (OP == OP_CBF_INC_ALL) {
				if (ok) {
					uint32_t* words = static_cast<uint32_t*>(a.filter);
					for (uint32_t i = 0; i < h; ++i)
						cbf_inc_sat(words, reduce_mod<POW2>(HASH_AT(i), a.mod));
				}
			} else if (OP == OP_CBF_INC_MIN) {
				if (ok) {
					uint32_t* words = static_cast<uint32_t*>(a.filter);
					// incrementMin, CountingBloomFilter.hpp:135-162 (positions recomputed, not stored)
					for (;;) {
						uint32_t mn = 0xffu;
						for (uint32_t i = 0; i < h; ++i) {
							const uint32_t v = cbf_read_fresh(words, reduce_mod<POW2>(HASH_AT(i), a.mod));
							mn = v < mn ? v : mn;
						}
						if (mn == 0xffu)
							break;
						bool done = false;
						for (uint32_t i = 0; i < h; ++i)
							done |= cbf_cas_byte(words, reduce_mod<POW2>(HASH_AT(i), a.mod), mn);
						if (done)
							break;
					}
				}
			} else if (OP == OP_POSITIONS) {
				// multi-GPU routing: append every probe position to its owner's bucket.
				// wave-aggregated reservation: one atomicAdd per (wave, shard) per probe slot.
				for (uint32_t i = 0; i < h; ++i) {
					const uint64_t p = reduce_mod<POW2>(HASH_AT(i), a.mod);
					const uint32_t owner =
					    a.mod.shard_shift != 0xffffffffu ? (uint32_t)(p >> a.mod.shard_shift) : (uint32_t)(p / a.mod.shard_len);
					const uint64_t local = p - (uint64_t)owner * a.mod.shard_len;
					for (uint32_t s = 0; s < a.n_shards; ++s) {
						const uint64_t m = __ballot(ok && owner == s);
						if (m == 0)
							continue;
						const uint32_t lane = __lane_id();
						const uint32_t leader = __ffsll((unsigned long long)m) - 1;
						unsigned long long base = 0;
						if (lane == leader)
							base = atomicAdd(a.bucket_counts + s, (unsigned long long)__popcll(m));
						base = __shfl(base, leader, 64);
						if (ok && owner == s) {
							const uint64_t slot = base + __popcll(m & ((1ull << lane) - 1));
							if (slot < a.bucket_cap) {
								a.buckets[(uint64_t)s * a.bucket_cap + slot] = local;
								if (a.tags)
									a.tags[(uint64_t)s * a.bucket_cap + slot] = gp * h + i;
							}
						}
					}
				}
			}
		}

#undef HASH_AT
		if (OP == OP_BF_CONTAINS && pipelined) {
#pragma unroll
			for (int w = 0; w < kW; ++w) {
				uint32_t all = (valid_mask >> w) & 1u;
#pragma unroll
				for (int i = 0; i < kPipe; ++i)
					if ((uint32_t)i < h)
						all &= wordbuf[w][i] >> bitbuf[w][i];
				hit_mask |= (all & 1u) << w;
			}
		}

		// ---- 4. results ----
		const uint64_t ob = (g0 >> 3) + tid;
		if (OP == OP_BF_CONTAINS || OP == OP_BF_INSERT_CHECK || OP == OP_CBF_QUERY) {
			if (a.hit_bits && ob < out_bytes)
				a.hit_bits[ob] = (uint8_t)hit_mask;
		}
		if (a.valid_bits && ob < out_bytes)
			a.valid_bits[ob] = (uint8_t)valid_mask;
		my_valid += __popc(valid_mask);
		my_hit += __popc(hit_mask);
	}

	if (a.counts) {
		const uint32_t wv = wave_sum(my_valid), wh = wave_sum(my_hit);
		if ((tid & 63) == 0) {
			atomicAdd(&sh.cnt_valid, (unsigned long long)wv);
			atomicAdd(&sh.cnt_hit, (unsigned long long)wh);
		}
		__syncthreads();
		if (tid == 0) {
			atomicAdd(reinterpret_cast<unsigned long long*>(a.counts), sh.cnt_valid);
			atomicAdd(reinterpret_cast<unsigned long long*>(a.counts) + 1, sh.cnt_hit);
		}
	}
}

template <int OP>
static hipError_t launch_one(const SeqArgs& a, hipStream_t s, dim3 grid, size_t dyn)
{
	const bool pow2 = a.mod.pow2 != 0;
	const bool spaced = a.hp.n_seeds > 0;
#define BTLBF_LAUNCH(P, S)                                                                          \
	do {                                                                                            \
		if (dyn > 48 * 1024) {                                                                      \
			hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&seq_kernel<OP, P, S>), \
			                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
			if (e != hipSuccess)                                                                    \
				return e;                                                                           \
		}                                                                                           \
		hipLaunchKernelGGL((seq_kernel<OP, P, S>), grid, dim3(kThreads), dyn, s, a);                \
	} while (0)
	if (pow2 && !spaced)
		BTLBF_LAUNCH(true, false);
	else if (!pow2 && !spaced)
		BTLBF_LAUNCH(false, false);
	else if (pow2 && spaced)
		BTLBF_LAUNCH(true, true);
	else
		BTLBF_LAUNCH(false, true);
#undef BTLBF_LAUNCH
	return hipGetLastError();
}

hipError_t launch_seq_op(int op, const SeqArgs& a_in, hipStream_t s)
{
	SeqArgs a = a_in;
	if (a.len == 0)
		return hipSuccess;
	a.n_tiles = (a.len + kTile - 1) / kTile;
	// a few workgroups per CU, each walking a contiguous run of tiles
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) {
		int v = 0;
		if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
			cus = v;
	}
	uint64_t max_blocks = (uint64_t)cus * 8;
	uint64_t blocks = a.n_tiles < max_blocks ? a.n_tiles : max_blocks;
	a.tiles_per_block = (a.n_tiles + blocks - 1) / blocks;
	blocks = (a.n_tiles + a.tiles_per_block - 1) / a.tiles_per_block;
	const uint32_t k = a.hp.k;
	size_t dyn = ((kTile + k - 1 + 15 + 15) / 16) * 16;
	if (a.hp.n_seeds > 0)
		dyn += (size_t)k * kNumCodes * 16 + (((size_t)a.hp.dc_off[a.hp.n_seeds] * 2 + 15) / 16) * 16;
	dim3 grid((unsigned)blocks);
	switch (op) {
	case OP_BF_INSERT: return launch_one<OP_BF_INSERT>(a, s, grid, dyn);
	case OP_BF_CONTAINS: return launch_one<OP_BF_CONTAINS>(a, s, grid, dyn);
	case OP_BF_INSERT_CHECK: return launch_one<OP_BF_INSERT_CHECK>(a, s, grid, dyn);
	case OP_CBF_INC_MIN: return launch_one<OP_CBF_INC_MIN>(a, s, grid, dyn);
	case OP_CBF_INC_ALL: return launch_one<OP_CBF_INC_ALL>(a, s, grid, dyn);
	case OP_CBF_QUERY: return launch_one<OP_CBF_QUERY>(a, s, grid, dyn);
	case OP_HASH_ONLY: return launch_one<OP_HASH_ONLY>(a, s, grid, dyn);
	case OP_POSITIONS: return launch_one<OP_POSITIONS>(a, s, grid, dyn);
	default: return hipErrorInvalidValue;
	}
}

} // namespace btlbf

// csrc/seq_kernels.hip -- the direct hot path: ntHash k-mer stream fused with the filter probes.
//
// One launch walks a sequence buffer of `len` bytes.  A workgroup (256 threads = 4 wavefronts)
// owns a contiguous run of tiles; a tile is 2048 consecutive window start offsets.
//   1. stage + 2. hash: seq_core.hpp (coalesced loads -> LDS base codes; 8 rolling windows per lane)
//   3. probe: canonical hash -> h hashes -> positions -> atomicOr / load / byte CAS on the
//      HBM-resident array.  For contains() all kW*h probes of a lane are issued before the first is
//      consumed.
//   4. results: the 8 windows of a lane are exactly one byte of the per-window bitmaps, so hit /
//      valid bits leave as coalesced byte stores; popcounts feed the hit counters; the multi-GPU
//      bucketing op reserves bucket slots with __ballot wave aggregation.
//
// Reference semantics reproduced (not its code): BloomFilter insert/contains/insertAndCheck
// (BloomFilter.hpp:185-262), CountingBloomFilter minCount/incrementMin/incrementAll/contains
// (CountingBloomFilter.hpp:53-64,135-196).
#include "seq_core.hpp"

namespace btlbf {

static constexpr int kThreads = 256;
static constexpr int kTile = kThreads * kW;

int seq_tile_windows() { return kTile; }

template <int OP, bool POW2, bool SPACED>
__global__ __launch_bounds__(kThreads) void seq_kernel(const SeqArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
	__shared__ SeqShared sh;

	if (a.gate && (uint32_t)a.gate[1] != a.gate_mode)
		return; // a step of the partitioned query that the device decided against (internal.hpp: SeqArgs::gate)
	const uint32_t tid = threadIdx.x;
	const uint32_t k = a.hp.k;
	const uint32_t h = a.hp.h;
	// dynamic LDS carve: [tile bytes | spaced-seed tables]
	const uint32_t tile_cap = seq_tile_cap(kTile, k);
	uint8_t* tile = dyn;
	uint8_t* spaced_lds = dyn + tile_cap;
	seq_setup_tables<kThreads, SPACED>(sh, a.hp, spaced_lds);

	const uint64_t t_begin = a.first_tile + (uint64_t)blockIdx.x * a.tiles_per_block;
	uint64_t t_end = t_begin + a.tiles_per_block;
	if (t_end > a.first_tile + a.n_tiles)
		t_end = a.first_tile + a.n_tiles;
	const uint32_t L = a.layout.starts ? 0 : a.layout.read_len;
	// offset of the tile's first byte inside its read (uniform layout), kept incrementally
	uint32_t tile_off = 0;
	if (L && t_begin < t_end)
		tile_off = (uint32_t)((t_begin * (uint64_t)kTile) % L);
	const uint32_t tile_step = L ? (uint32_t)(kTile % L) : 0;
	const uint64_t out_bytes = ((a.len + 63) / 64) * 8;

	uint32_t my_valid = 0, my_hit = 0;

	for (uint64_t t = t_begin; t < t_end; ++t) {
		const uint64_t g0 = t * (uint64_t)kTile;
		const uint32_t mis = seq_stage_tile<kThreads>(tile, tile_cap, sh, a.seq, a.len, a.layout, k, g0, tile_off);
		tile_off = seq_next_tile_off(tile_off, tile_step, L);

		uint32_t valid_mask = 0, hit_mask = 0;
		// contains() with <= kPipe hashes runs in two waves of independent loads per lane: probe 0
		// of all kW windows first (like the reference's early exit at the first clear bit,
		// BloomFilter.hpp:257-259, a k-mer that misses here costs one request instead of h), then
		// probes 1..h-1 of the windows that are still alive.  Larger hash counts take the per-window path.
		constexpr int kPipe = 4;
		uint32_t wordbuf[kW][kPipe];
		uint32_t bitbuf[kW][kPipe];
		uint64_t canon[kW];
		(void)wordbuf;
		(void)bitbuf;
		(void)canon;
		const bool pipelined = (OP == OP_BF_CONTAINS) && !SPACED && h <= kPipe;
		// incrementMin with <= kPipe hashes: the kW*h counter words of a lane are requested together in the window
		// walk (one memory latency per lane, not one per counter); the update follows the walk (below)
		const bool min_piped = (OP == OP_CBF_INC_MIN) && !SPACED && h <= kPipe;

		seq_lane_windows<SPACED, kW>(tile, sh, a.hp, spaced_lds, tid * kW + mis, [&](int w, bool ok, const WinHash<SPACED>& wh) {
			valid_mask |= (uint32_t)ok << w;
			const uint64_t gp = g0 + tid * kW + w; // byte offset of this window

			if (OP == OP_HASH_ONLY) {
				if (gp < a.len) {
					for (uint32_t i = 0; i < h; ++i)
						a.hashes[gp * h + i] = ok ? wh.at(i) : 0;
					if (a.strand_bits)
						a.strand_bits[gp] = ok ? wh.stn : 0;
				}
			} else if (OP == OP_BF_INSERT) {
				if (ok) {
					uint32_t* words = static_cast<uint32_t*>(a.filter);
					for (uint32_t i = 0; i < h; ++i) {
						const uint64_t p = reduce_mod<POW2>(wh.at(i), a.mod) - a.mod.shard_lo;
						if (p < a.mod.shard_len)
							bf_set(words, p);
					}
				}
			} else if (OP == OP_BF_INSERT_CHECK) {
				if (ok) {
					uint32_t* words = static_cast<uint32_t*>(a.filter);
					uint32_t all = 1;
					for (uint32_t i = 0; i < h; ++i)
						all &= bf_set_fetch(words, reduce_mod<POW2>(wh.at(i), a.mod));
					hit_mask |= all << w;
				}
			} else if (OP == OP_BF_CONTAINS) {
				const uint32_t* words = static_cast<const uint32_t*>(a.filter);
				if (pipelined) {
					const uint64_t p = reduce_mod<POW2>(wh.bcan, a.mod);
					// unclean windows probe word 0 (harmless) so the loop stays branch-free
					wordbuf[w][0] = bf_word(words, ok ? p : 0);
					bitbuf[w][0] = (uint32_t)p & 31;
					canon[w] = wh.bcan;
				} else if (ok) {
					uint32_t all = 1;
					for (uint32_t i = 0; i < h; ++i) {
						const uint64_t p = reduce_mod<POW2>(wh.at(i), a.mod);
						all &= (bf_word(words, p) >> (p & 31)) & 1u;
					}
					hit_mask |= all << w;
				}
			} else if (OP == OP_BF_CONTAINS_WIN) {
				if (ok) {
					const uint32_t* words = static_cast<const uint32_t*>(a.filter);
					uint32_t all = 1;
					for (uint32_t i = 0; i < h; ++i) {
						const uint64_t p = reduce_mod<POW2>(wh.at(i), a.mod) - a.mod.shard_lo;
						if (p < a.mod.shard_len)
							all &= (bf_word(words, p) >> (p & 31)) & 1u;
					}
					hit_mask |= all << w;
				}
			} else if (OP == OP_BF_RESOLVE) {
				// partitioned query, second step: a.buckets is the failed-position hash set
				// (key = position + 1, 0 = empty, linear probing), a.bucket_cap its index mask
				if (ok) {
					const unsigned long long* table = reinterpret_cast<const unsigned long long*>(a.buckets);
					const uint64_t tmask = a.gate ? a.gate[0] : a.bucket_cap;
					uint32_t all = 1;
					for (uint32_t i = 0; i < h; ++i) {
						const unsigned long long key = reduce_mod<POW2>(wh.at(i), a.mod) + 1;
						uint64_t slot = mix64(key) & tmask;
						for (;;) {
							const unsigned long long v = table[slot];
							if (v == key)
								all = 0;
							if (v == key || v == 0ull)
								break;
							slot = (slot + 1) & tmask;
						}
					}
					hit_mask |= all << w;
				}
			} else if (OP == OP_CBF_QUERY) {
				const uint8_t* ctr = static_cast<const uint8_t*>(a.filter);
				uint32_t mn = 0xff;
				if (ok) {
					for (uint32_t i = 0; i < h; ++i) { // a shard sees the counters inside its window only
						const uint64_t p = reduce_mod<POW2>(wh.at(i), a.mod) - a.mod.shard_lo;
						if (p < a.mod.shard_len) {
							const uint32_t v = ctr[p];
							mn = v < mn ? v : mn;
						}
					}
					hit_mask |= (uint32_t)(mn >= a.threshold) << w;
				}
				if (a.min_out && gp < a.len)
					a.min_out[gp] = ok ? (uint8_t)mn : 0;
			} else if (OP == OP_CBF_INC_ALL) {
				if (ok) {
					uint32_t* words = static_cast<uint32_t*>(a.filter);
					for (uint32_t i = 0; i < h; ++i) {
						const uint64_t p = reduce_mod<POW2>(wh.at(i), a.mod) - a.mod.shard_lo;
						if (p < a.mod.shard_len)
							cbf_inc_sat(words, p);
					}
				}
			} else if (OP == OP_CBF_INC_MIN && min_piped) {
				const uint32_t* words = static_cast<const uint32_t*>(a.filter);
				canon[w] = wh.bcan;
#pragma unroll
				for (int i = 0; i < kPipe; ++i) {
					if ((uint32_t)i < h) {
						const uint64_t p = reduce_mod<POW2>(wh.at(i), a.mod);
						wordbuf[w][i] = agent_load(words + (ok ? p >> 2 : 0)); // unclean windows read word 0 (unused)
						bitbuf[w][i] = ((uint32_t)p & 3) * 8;
					}
				}
			} else if (OP == OP_CBF_INC_MIN) {
				if (ok) {
					uint32_t* words = static_cast<uint32_t*>(a.filter);
					// incrementMin, CountingBloomFilter.hpp:135-162 (positions recomputed, not stored)
					for (;;) {
						uint32_t mn = 0xffu;
						for (uint32_t i = 0; i < h; ++i) {
							const uint32_t v = cbf_read_fresh(words, reduce_mod<POW2>(wh.at(i), a.mod));
							mn = v < mn ? v : mn;
						}
						if (mn == 0xffu)
							break;
						bool done = false;
						for (uint32_t i = 0; i < h; ++i)
							done |= cbf_cas_byte(words, reduce_mod<POW2>(wh.at(i), a.mod), mn);
						if (done)
							break;
					}
				}
			} else if (OP == OP_POSITIONS) {
				// multi-GPU routing: append every probe position to its owner's bucket.
				// wave-aggregated reservation: one atomicAdd per (wave, shard) per probe slot.
				for (uint32_t i = 0; i < h; ++i) {
					const uint64_t p = reduce_mod<POW2>(wh.at(i), a.mod);
					const uint32_t owner = a.mod.shard_shift != 0xffffffffu ? (uint32_t)(p >> a.mod.shard_shift)
					                                                        : (uint32_t)(p / a.mod.shard_len);
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
		});

		if (OP == OP_BF_CONTAINS && pipelined) {
			const uint32_t* words = static_cast<const uint32_t*>(a.filter);
			uint32_t alive = 0;
#pragma unroll
			for (int w = 0; w < kW; ++w)
				alive |= ((valid_mask >> w) & (wordbuf[w][0] >> bitbuf[w][0]) & 1u) << w;
			// second wave: the remaining probes of the windows whose first bit was set
#pragma unroll
			for (int w = 0; w < kW; ++w) {
				if ((alive >> w) & 1u) {
#pragma unroll
					for (int i = 1; i < kPipe; ++i) {
						if ((uint32_t)i < h) {
							const uint64_t p = reduce_mod<POW2>(extra_hash(canon[w], a.hp.kms, i), a.mod);
							wordbuf[w][i] = bf_word(words, p);
							bitbuf[w][i] = (uint32_t)p & 31;
						}
					}
				}
			}
#pragma unroll
			for (int w = 0; w < kW; ++w) {
				uint32_t all = (alive >> w) & 1u;
				if (all) {
#pragma unroll
					for (int i = 1; i < kPipe; ++i)
						if ((uint32_t)i < h)
							all &= wordbuf[w][i] >> bitbuf[w][i];
				}
				hit_mask |= (all & 1u) << w;
			}
		}

		if (OP == OP_CBF_INC_MIN && min_piped) {
			// incrementMin, CountingBloomFilter.hpp:135-162, on the words already fetched: the minimum of the k-mer's h
			// counters, then ONE compare-and-swap per counter that was read equal to it (a counter never decreases, so
			// one read above the minimum cannot be at it any more), all of a lane's in flight together, expected
			// value = the word as read.  The reference's byte CAS succeeds iff the byte still holds the minimum; the
			// word CAS here also fails when a NEIGHBOURING counter changed, so a failed one is retried for as long as
			// its own byte still holds the minimum.  No success at all (every minimum counter was raised by someone
			// else meanwhile): start over from fresh reads, as the reference does (:139-160).
			uint32_t* words = static_cast<uint32_t*>(a.filter);
			uint32_t mn[kW];
			uint32_t prev[kW][kPipe];
#pragma unroll
			for (int w = 0; w < kW; ++w) {
				mn[w] = 0xffu;
#pragma unroll
				for (int i = 0; i < kPipe; ++i)
					if ((uint32_t)i < h) {
						const uint32_t v = (wordbuf[w][i] >> bitbuf[w][i]) & 0xffu;
						mn[w] = v < mn[w] ? v : mn[w];
					}
				if (!((valid_mask >> w) & 1u))
					mn[w] = 0xffu; // nothing to do (also the saturated case, :146-149)
			}
#pragma unroll
			for (int w = 0; w < kW; ++w) {
#pragma unroll
				for (int i = 0; i < kPipe; ++i) {
					prev[w][i] = 0;
					if ((uint32_t)i < h && mn[w] != 0xffu && ((wordbuf[w][i] >> bitbuf[w][i]) & 0xffu) == mn[w]) {
						const uint64_t p = reduce_mod<POW2>(i ? extra_hash(canon[w], a.hp.kms, i) : canon[w], a.mod);
						prev[w][i] = atomicCAS(words + (p >> 2), wordbuf[w][i], wordbuf[w][i] + (1u << bitbuf[w][i]));
					}
				}
			}
#pragma unroll
			for (int w = 0; w < kW; ++w) {
				if (mn[w] == 0xffu)
					continue;
				bool done = false;
#pragma unroll
				for (int i = 0; i < kPipe; ++i) {
					if ((uint32_t)i < h && ((wordbuf[w][i] >> bitbuf[w][i]) & 0xffu) == mn[w]) {
						uint32_t expect = wordbuf[w][i], got = prev[w][i];
						while (got != expect && ((got >> bitbuf[w][i]) & 0xffu) == mn[w]) { // a neighbour changed: again
							const uint64_t p = reduce_mod<POW2>(i ? extra_hash(canon[w], a.hp.kms, i) : canon[w], a.mod);
							expect = got;
							got = atomicCAS(words + (p >> 2), expect, expect + (1u << bitbuf[w][i]));
						}
						done |= got == expect;
					}
				}
				while (!done) { // the reference's retry: fresh minimum, byte CAS on every counter
					uint32_t m2 = 0xffu;
					for (uint32_t i = 0; i < h; ++i) {
						const uint32_t v = cbf_read_fresh(words, reduce_mod<POW2>(i ? extra_hash(canon[w], a.hp.kms, i) : canon[w], a.mod));
						m2 = v < m2 ? v : m2;
					}
					if (m2 == 0xffu)
						break;
					for (uint32_t i = 0; i < h; ++i)
						done |= cbf_cas_byte(words, reduce_mod<POW2>(i ? extra_hash(canon[w], a.hp.kms, i) : canon[w], a.mod), m2);
				}
			}
		}

		// ---- results ----
		const uint64_t ob = (g0 >> 3) + tid;
		if (OP == OP_BF_CONTAINS || OP == OP_BF_INSERT_CHECK || OP == OP_CBF_QUERY || OP == OP_BF_RESOLVE ||
		    OP == OP_BF_CONTAINS_WIN) {
			if (a.hit_bits && ob < out_bytes) {
				// the resolve step only ever CLEARS windows: its tile range may reach back into windows an
				// earlier batch of the partitioned query has already resolved (batches of pass-A tiles need
				// not end on this kernel's tile boundaries), and those answers stand
				if (OP == OP_BF_RESOLVE)
					a.hit_bits[ob] &= (uint8_t)hit_mask;
				else
					a.hit_bits[ob] = (uint8_t)hit_mask;
			}
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

// a.first_tile / a.n_tiles select a tile range of the buffer (n_tiles == 0: the whole buffer)
hipError_t launch_seq_op(int op, const SeqArgs& a_in, hipStream_t s)
{
	SeqArgs a = a_in;
	if (a.len == 0)
		return hipSuccess;
	const uint64_t all_tiles = (a.len + kTile - 1) / kTile;
	if (a.n_tiles == 0) {
		a.first_tile = 0;
		a.n_tiles = all_tiles;
	}
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
	size_t dyn = seq_tile_cap(kTile, a.hp.k) + seq_spaced_bytes(a.hp);
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
	case OP_BF_RESOLVE: return launch_one<OP_BF_RESOLVE>(a, s, grid, dyn);
	case OP_BF_CONTAINS_WIN: return launch_one<OP_BF_CONTAINS_WIN>(a, s, grid, dyn);
	default: return hipErrorInvalidValue;
	}
}

} // namespace btlbf

// csrc/internal.hpp -- parameter blocks shared by the HIP kernels and the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace btlbf {

// ntHash constants (values: /root/reference/vendor/nthash.hpp:183-193; the pre-rotated
// msTab31l/msTab33r tables at :230-347 are srol^s of these and are derived, not stored)
static constexpr uint64_t kSeedA = 0x3c8bfbb395c60474ULL;
static constexpr uint64_t kSeedC = 0x3193c18562a02b4cULL;
static constexpr uint64_t kSeedG = 0x20323ed082572324ULL;
static constexpr uint64_t kSeedT = 0x295549f54be24456ULL;
static constexpr uint64_t kMultiSeed = 0x90b45d39fb6da1faULL;
static constexpr unsigned kMultiShift = 27;

// base codes used on the device: A=0 C=1 G=2 T=3 with reverse-strand seed = seed[code ^ 3]; codes
// 4..7 are the raw bytes {4,5} 7 3 1 that the reference's seedTab also accepts
// (vendor/nthash.hpp:196): their forward seed is A C G T and -- because "c & cpOff" maps such a
// byte to itself (nthash.hpp:180,688) -- their reverse-strand seed is the SAME seed.
// per-base byte staged in LDS: bits 6:4 code (byte & kCodeOff is the byte offset of a 16-byte table
// entry), bit 0 = valid base, bit 1 = valid base that is not the first of a sequence (a window is
// clean when its first base is valid and the k-1 bases after it are all "good" in this sense)
static constexpr unsigned kCodeShift = 4u;
static constexpr unsigned kCodeOff = 0x70u;
static constexpr unsigned kNumCodes = 8u;
static constexpr unsigned kBaseValid = 1u;
static constexpr unsigned kBaseGood = 2u;

// partitioned pipeline: entries leave the LDS rings as CHUNKS of kChunk uint32 = one aligned 128-byte line
// (64-byte chunks were measured: fewer late entries, but the HBM write traffic of pass A rose 22 % above
// the bytes stored -- half-line writes are not free on this memory system)
static constexpr uint32_t kChunk = 32;
static constexpr uint32_t kChunkShift = 5;

static constexpr int kMaxHash = 32;  // hash_num supported by the fused kernels
static constexpr int kMaxSeeds = 16; // spaced seeds per filter

// How positions are reduced modulo the filter size and mapped to this object's local array.
struct ModParams {
	uint64_t size;      // global modulus (bits or counters)
	uint64_t magic;     // floor(2^64 / size) for the mulhi reduction (unused when pow2)
	uint64_t mask;      // size-1 when pow2
	uint64_t shard_lo;  // first global position held locally
	uint64_t shard_len; // number of positions held locally (== size unless a shard)
	uint32_t pow2;
	uint32_t shard_shift; // log2(shard_len) if shard_len is a power of two else 0xffffffff
	// sizes above 2^32 of no power of two: magic below 2^32, kept apart from `magic` (and 2^64 - size apart from
	// `size`) so that the optimiser cannot tie them together again -- see reduce_mod_small; 0 otherwise
	uint32_t magic32;
	uint64_t neg_size;
};

// Everything the hash stage needs.  Tables are tiny and copied to LDS by each workgroup.
struct HashParams {
	uint32_t k;
	uint32_t h;        // hashes per window as seen by the filter (n_seeds*h2 when spaced)
	uint32_t n_seeds;  // 0 = plain ntHash (ntHashIterator); >0 = spaced seeds (stHashIterator)
	uint32_t h2;
	uint32_t use_pos_tab; // build pos_tab[i*8+c] = {srol^(k-1-i)(fwd[c]), srol^i(rev[c])} in LDS (k small enough)
	uint32_t pad_;
	uint64_t kms;      // k * multiSeed (nthash.hpp:585-589: multiplier is i ^ (k*multiSeed))
	// per code c: {fwd[c], srol^(k-1)(rev[c])}                 -> Horner start-up of fh / rh
	uint64_t init_tab[8][2];
	// per code c: {fwd[c], srol^k(rev[c])}                    -> roll, incoming base
	uint64_t in_tab[8][2];
	// per code c: {srol^k(fwd[c]), rev[c]}                    -> roll, outgoing base
	uint64_t out_tab[8][2];
	// spaced seeds: dc_idx (device memory, owned by the filter / call) = concatenated don't-care
	// indices, seed j uses [dc_off[j], dc_off[j+1]); they are XORed back out through pos_tab
	const uint16_t* dc_idx;
	uint32_t dc_off[kMaxSeeds + 1];
	// the same don't-care positions as ONE list of distinct offsets, each with the set of seeds that leave it out:
	// dcu[u] = offset | seed mask << 16 (device memory behind dc_idx), the first n_dcu_all of them left out by EVERY
	// seed.  A window's positional-table term for an offset is the same whichever seed asks for it, so the pass-A
	// hash stage fetches it once per offset and XORs it into the seeds of the mask (seq_core.hpp); n_dcu == 0: list
	// not available (more than kMaxDcu = 64 offsets: the kernels keep the list in one register of a wave), the per-seed lists above are walked instead.
	const uint32_t* dcu;
	uint32_t n_dcu, n_dcu_all;
	// The entries behind the first n_dcu_all come in PAIRS with one mask.  want_pair_rows (set with the list; 0 = none):
	// the pairs, an even number of them, can be served by two-base table rows of 256 bytes each -- row[c1][c2] = the sum
	// of the pair's two terms for the bases A C G T -- if a launcher finds the LDS for them; n_pair_rows is the launcher's
	// answer (the kernels' view: rows behind the union list in LDS, or 0 = one table read per offset).
	uint32_t want_pair_rows, n_pair_rows;
};
static constexpr uint32_t kMaxDcu = 64;

// Sequence boundaries inside a buffer (btlbf_layout, device-resident form).
struct LayoutParams {
	const uint64_t* starts; // device pointer or nullptr
	uint64_t n_seqs;
	uint32_t read_len;
};

enum SeqOp : int {
	OP_BF_INSERT = 0,
	OP_BF_CONTAINS = 1,
	OP_BF_INSERT_CHECK = 2,
	OP_CBF_INC_MIN = 3,
	OP_CBF_INC_ALL = 4,
	OP_CBF_QUERY = 5,    // contains + optional min counts
	OP_HASH_ONLY = 6,    // dense hashes / valid / strand output
	OP_POSITIONS = 7,    // bucket positions by owning shard (multi-GPU)
	OP_BF_RESOLVE = 8,   // partitioned query, second step: windows with a probe in the failed-position set miss
	OP_BF_CONTAINS_WIN = 9 // contains() on one shard of a larger filter: only the probes inside its window are tested
};

// operations on precomputed hash rows (aux_kernels.hip)
enum HashOp : int {
	H_BF_INSERT = 0,
	H_BF_CONTAINS = 1,
	H_BF_INSERT_CHECK = 2,
	H_CBF_INC_MIN = 3,
	H_CBF_INC_ALL = 4,
	H_CBF_MIN = 5,      // out = min count
	H_CBF_CONTAINS = 6, // out = min >= threshold
	H_CBF_INSERT_CHECK = 7
};

struct SeqArgs {
	const uint8_t* seq; // device
	uint64_t len;
	LayoutParams layout;
	void* filter;       // device array (uint32 words for bloom, bytes for counting)
	ModParams mod;
	HashParams hp;
	uint32_t threshold;
	// outputs (device, optional)
	uint8_t* hit_bits;    // byte view of the uint64_t bitmap
	uint8_t* valid_bits;
	uint64_t* counts;     // {clean windows, hits}
	uint8_t* min_out;     // per window min count
	uint64_t* hashes;     // OP_HASH_ONLY: len*h
	uint64_t* strand_bits;
	// OP_POSITIONS
	uint32_t n_shards;
	uint64_t* buckets;
	uint64_t* tags;
	uint64_t bucket_cap;
	unsigned long long* bucket_counts;
	// stream-side control (the partitioned query's resolve step, decided on the device): gate[0] = index mask of the
	// failed-position table (OP_BF_RESOLVE takes it from here instead of bucket_cap), low half of gate[1] = what to
	// do; a launch with gate != nullptr returns at once unless that equals gate_mode
	const uint64_t* gate;
	uint32_t gate_mode;
	// tiling: tiles [first_tile, first_tile + n_tiles) of the buffer (n_tiles == 0: all of it)
	uint64_t first_tile;
	uint64_t n_tiles;
	uint64_t tiles_per_block;
	// pass A's read grid (part_read_grid; rg_reads == 0: plain tiles of NT*8 window starts)
	uint32_t rg_reads; // reads per tile
	uint32_t rg_gpr;   // groups of 8 window starts per read
	uint32_t rg_lpad;  // bytes a read occupies in the LDS tile (read_len rounded up to 8)
	uint32_t rg_cap;   // bytes of dynamic LDS the tile image takes (reads + guard + two window bitmaps)
	uint32_t rg_lpad_inv; // floor(2^32 / rg_lpad) + 1: x / rg_lpad == umulhi(x, rg_lpad_inv) for the offsets of a tile
	// ragged layout in pass A's overlapped schedule: words of the LDS bitmap (behind the tile image) in which the
	// staging waves mark the sequence starts of the NEXT tile while the current one is partitioned (0: none -- the
	// starts are marked after the staging, between two more barriers per tile)
	uint32_t sb_words;
	// reads to leave out (the split query, uniform reads through pass A's read grid only): bit r of this device array
	// of 32-bit words = read r of the buffer is answered elsewhere; readable for 160 bytes behind its last word
	const uint32_t* read_mask;
};

// How pass A cuts a buffer into tiles (partition_kernels.hip: part_tiling).  All host-side planning is in
// these tiles.
// Read grid of pass A (uniform layout): a tile is a whole number of reads, every read sits at an 8-byte aligned
// offset of the LDS image, and a lane takes 8 consecutive window starts of ONE read -- no lane spends its
// time on the k-1 window starts at a read's end that hold no k-mer (20 % of them for 150-base reads, k = 31).
struct PartGrid {
	uint32_t reads = 0, gpr = 0, lpad = 0, cap = 0;
};
bool part_read_grid(const HashParams& hp, uint32_t p0, const LayoutParams& lay, PartGrid* g);

struct PartTiling {
	uint32_t tile_bytes;      // window starts per tile = bytes from one tile's start to the next
	uint64_t n_tiles;         // tiles covering the buffer
	double windows_per_tile;  // expected clean windows per full tile
};

// ---- partitioned insert / contains (partition_kernels.hip) -------------------------------------------
// A SEGMENT is 2^seg_shift bits of the local array (64 KiB or 128 KiB: what one workgroup holds in
// LDS).  A partition pass writes BINS; every bin is written by several workgroups, each into its own
// REGION of `cap` 32-entry chunks: region index = bin * regions + writer.
struct PartOut {
	uint32_t P;       // bins one workgroup writes (its output block)
	uint32_t regions; // writers per bin
	uint32_t cap;     // region capacity in chunks
	uint32_t* cnt;    // [bins * regions] ENTRIES per region
	uint32_t* ent;    // [bins * regions][cap][32]
};
// How a pass reads the regions of its input bins.  Data that arrived through the multi-GPU exchange
// consists of `blocks` origin blocks, each laid out [bins_per_block][regions_per_block][cap][32];
// region r of bin b is ((r / regions_per_block) * bins_per_block + b) * regions_per_block + r % regions_per_block.
struct PartIn {
	uint32_t blocks;
	uint32_t bins_per_block;
	uint32_t regions_per_block;
	uint32_t cap;
	const uint32_t* cnt;
	const uint32_t* ent;
};
// side channels of a pass
struct PartSide {
	uint64_t pos_base;              // added to local positions that are reported (fail / spill lists)
	uint64_t* fail_list;            // partitioned contains: positions whose bit was found clear
	unsigned long long* fail_count;
	uint64_t fail_cap;
	uint64_t* spill_list;           // routing (multi-GPU): entries that could not be staged, as global positions
	unsigned long long* spill_count; // nullptr = overflow goes straight to this GPU's array
	uint64_t spill_cap;
	uint32_t counting;              // the array holds uint8_t counters (incrementAll / min >= threshold), not bits
	uint32_t threshold;             // counting query: a probe fails when its counter is below this
	uint32_t fresh;                 // insert into an array known to be all zero (a pending btlbf_clear): pass C builds
	                                // every segment from zero in LDS and writes it -- no read of the old contents,
	                                // untouched segments are written as zeros
	uint32_t late_cap;              // pass A's overlapped schedule: words of one late image (>= part_late_cap()) ...
	uint32_t* late_buf;             // ... of [workgroup][round parity][late_cap]: where entries that found their ring
	                                // full wait for the flush (nullptr: plain schedule)
	// Level-0 bins of bin_wseg SEGMENTS each instead of 2^bin_shift positions (partition_core.hpp part_bin_of): a local
	// array whose segment count is no power of two still fills all 2^n staging rings of pass A evenly.  0 = bins of
	// 2^bin_shift positions.  bin_magic = ceil(2^32 / bin_wseg), bin_seg_shift = log2(positions per segment),
	// bin_width = bin_wseg << bin_seg_shift = positions per bin (at most 2^30).
	uint32_t bin_wseg, bin_magic, bin_seg_shift, bin_width;
};
// a late image mirrors the staging rings of a pass-A workgroup (partition_core.hpp kStageEntries): slot j of bin b's
// row takes the (j+1)-th entry that found the ring of b full this round
static inline uint32_t part_late_cap() { return 32768u; }

// launchers (defined in the .hip files)
PartTiling part_tiling(const HashParams& hp, uint32_t p0, const LayoutParams& lay, uint64_t len);
bool part_supported(const HashParams& hp); // can pass A hash this configuration (1..8 hashes; spaced seeds: the union list)
bool part_hash_fits(const HashParams& hp, uint32_t p0);
uint32_t part_hash_regions(const HashParams& hp, uint32_t p0, uint32_t cus); // pass A workgroups = regions per bin
hipError_t launch_part_hash(const SeqArgs& a, const PartOut& out, uint32_t bin_shift, const PartSide& sd, int query,
                            hipStream_t s);
// exact != 0: every entry counts (counter increments), so readers honour the exact entry count of a
// region instead of taking its padded last vector whole
hipError_t launch_part_split(void* filter, const PartIn& in, uint32_t first_in, uint32_t abs_first, uint32_t n_in_bins,
                             const PartOut& out, uint32_t sub_shift, uint32_t in_shift, const PartSide& sd, int query,
                             int exact, hipStream_t s);
// seg_shift = log2(positions per segment); the array holds bits, or uint8_t counters when sd.counting
hipError_t launch_part_apply(void* filter, uint64_t local_bytes, uint32_t seg_shift, uint64_t seg_first, uint64_t n_seg,
                             const PartIn& in, const PartSide& sd, int query, hipStream_t s);
hipError_t launch_failset_build(const uint64_t* fail_list, uint64_t n, uint64_t* table, uint64_t mask, hipStream_t s);
// the same decided on the device from the fail count of a batch: ctl[0] := table mask, ctl[1] := mode | n << 32 with
// mode 0 = no failed position (nothing to do), 1 = table built (resolve), 2 = more than fail_cap (redo directly)
enum { GATE_NONE = 0, GATE_RESOLVE = 1, GATE_REDO = 2 };
hipError_t launch_failset_auto(const uint64_t* fail_list, const unsigned long long* fail_count, uint64_t fail_cap,
                               uint64_t* table, uint64_t max_slots, uint64_t* ctl, hipStream_t s);
hipError_t launch_spill(void* filter, const uint64_t* pos, uint64_t n, uint64_t lo, uint64_t len, int test,
                        const PartSide& sd, hipStream_t s);
hipError_t launch_seq_op(int op, const SeqArgs& a, hipStream_t s);
hipError_t launch_hash_op(int op, void* filter, const ModParams& mod, uint32_t h, uint32_t threshold,
                          const uint64_t* hashes, uint64_t n, uint8_t* out, int serial, hipStream_t s,
                          const uint8_t* row_valid = nullptr);
// raw k-mers (KmerBloomFilter's NTC64(kmerSeq, k) path) -> hash rows + one valid byte per k-mer
hipError_t launch_kmer_rows(const uint8_t* kmers, uint64_t n, uint32_t k, uint32_t h, uint64_t kms, uint64_t* rows,
                            uint8_t* valid, hipStream_t s);
hipError_t launch_serial_seq_update(const SeqArgs& a, int op, const uint64_t* hashes,
                                    const uint8_t* valid_bits, uint8_t* out, hipStream_t s);
hipError_t launch_popcount(const void* data, uint64_t nbytes, int mode, uint32_t threshold,
                           unsigned long long* out, hipStream_t s);
hipError_t launch_digest(const void* data, uint64_t nbytes, uint64_t word0, unsigned long long* out2, hipStream_t s);
hipError_t launch_compare(const void* a, const void* b, uint64_t nbytes, int counting, unsigned long long* out3,
                          hipStream_t s);
hipError_t launch_synth(uint8_t* out, uint64_t seed, uint64_t first, uint64_t n, uint32_t read_len,
                        hipStream_t s);
hipError_t launch_microbench(void* data, uint64_t nbytes, int kind, uint64_t n_access,
                             unsigned long long* sink, hipStream_t s);
hipError_t launch_positions(int test, void* filter, const ModParams& mod, const uint64_t* pos,
                            uint64_t n, uint8_t* out, hipStream_t s);
hipError_t launch_count_per_seq(const uint64_t* hit_bits, const uint64_t* valid_bits, uint64_t len,
                                const uint64_t* starts, uint64_t n_seqs, uint32_t read_len, uint32_t k,
                                uint32_t* hits_out, uint32_t* valid_out, hipStream_t s);
hipError_t launch_and_answers(const uint64_t* tags, const uint8_t* answers, uint64_t n, uint32_t h,
                              uint64_t* hit_bits, hipStream_t s);

// split query (aux_kernels.hip): per-read sampling, cold-read ranks, compaction, bitmap merge
hipError_t launch_read_sample(const uint8_t* seq, uint64_t n_reads, uint32_t L, uint32_t stride, const HashParams& hp,
                              const ModParams& mod, const void* filter, int counting, uint32_t threshold, uint64_t* flags,
                              uint64_t* n_cold, hipStream_t s, uint32_t probes2 = 0);
hipError_t launch_flag_prefix(const uint64_t* flags, uint64_t n_reads, uint32_t* prefix, hipStream_t s);
hipError_t launch_compact_reads(const uint8_t* seq, uint64_t n_reads, uint32_t L, const uint64_t* flags,
                                const uint32_t* prefix, uint8_t* warm_buf, uint8_t* cold_buf, hipStream_t s);
hipError_t launch_gather_cold_reads(const uint8_t* seq, uint64_t n_reads, uint32_t L, const uint64_t* flags,
                                    const uint32_t* prefix, uint8_t* cold_buf, uint32_t* cold_index, hipStream_t s);
hipError_t launch_merge_cold_bitmaps(uint64_t n_cold, uint32_t L, const uint32_t* cold_index, const uint64_t* cold_hit,
                                     const uint64_t* cold_valid, uint64_t* hit_out, uint64_t* valid_out, hipStream_t s);
hipError_t launch_merge_split_bitmaps(uint64_t len, uint32_t L, const uint64_t* flags, const uint32_t* prefix,
                                      const uint64_t* warm_hit, const uint64_t* cold_hit, const uint64_t* warm_valid,
                                      const uint64_t* cold_valid, uint64_t* hit_out, uint64_t* valid_out, hipStream_t s);

// rank structure over a bit filter (aux_kernels.hip): interleaved 512-bit blocks + rank queries
hipError_t launch_rank_build(const uint64_t* bits, uint64_t n_bits, uint64_t* il, uint64_t* scratch, uint64_t* total_dev,
                             hipStream_t s);
hipError_t launch_rank_query(const uint64_t* il, uint64_t n_bits, const uint64_t* in, uint64_t n, const ModParams& mod,
                             int reduce, uint64_t* rank_out, uint8_t* bit_out, hipStream_t s);

int seq_tile_windows(); // windows per workgroup tile (for host-side sizing)

} // namespace btlbf

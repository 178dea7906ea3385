/* oracle/btl_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See btl_oracle.h.
 *
 * Plain-C restatement of the reference algorithm, written from the arithmetic stated in
 * SURVEY.md section 3.5 and checked against the genuine reference build (oracle/_ref) and
 * tests/golden/.  Each function cites the reference lines it follows.
 */
#include "btl_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * ntHash arithmetic
 * ---------------------------------------------------------------------------------------- */

/* nthash.hpp:189-193 */
#define SEED_A 0x3c8bfbb395c60474ULL
#define SEED_C 0x3193c18562a02b4cULL
#define SEED_G 0x20323ed082572324ULL
#define SEED_T 0x295549f54be24456ULL
/* nthash.hpp:183,186 */
#define MULTI_SEED 0x90b45d39fb6da1faULL
#define MULTI_SHIFT 27

/* seedTab (nthash.hpp:195-228): A C G T U in both cases, plus the raw bytes 1,3,4,5,7 that the
 * "& cpOff" complement trick (nthash.hpp:180) lands on; everything else is 0 (= seedN). */
uint64_t
bo_seed(unsigned char c)
{
	switch (c) {
	case 'A': case 'a': case 4: case 5:
		return SEED_A;
	case 'C': case 'c': case 7:
		return SEED_C;
	case 'G': case 'g': case 3:
		return SEED_G;
	case 'T': case 't': case 'U': case 'u': case 1:
		return SEED_T;
	default:
		return 0;
	}
}

/* rol1 then swapbits033 (nthash.hpp:350-352,377-380): the low 33 bits and the high 31 bits
 * each rotate left by one, independently. */
uint64_t
bo_srol(uint64_t x)
{
	uint64_t lo = x & 0x1FFFFFFFFULL, hi = x >> 33;
	lo = ((lo << 1) | (lo >> 32)) & 0x1FFFFFFFFULL;
	hi = ((hi << 1) | (hi >> 30)) & 0x7FFFFFFFULL;
	return (hi << 33) | lo;
}

/* ror1 then swapbits3263 (nthash.hpp:361-363,383-386): inverse of bo_srol */
uint64_t
bo_sror(uint64_t x)
{
	uint64_t lo = x & 0x1FFFFFFFFULL, hi = x >> 33;
	lo = ((lo >> 1) | (lo << 32)) & 0x1FFFFFFFFULL;
	hi = ((hi >> 1) | (hi << 30)) & 0x7FFFFFFFULL;
	return (hi << 33) | lo;
}

/* srol applied s times; equals msTab31l[c][s%31] | msTab33r[c][s%33] when x = seedTab[c]
 * (nthash.hpp:230-347 hold exactly these pre-rotated values) */
uint64_t
bo_srol_n(uint64_t x, unsigned s)
{
	uint64_t lo = x & 0x1FFFFFFFFULL, hi = x >> 33;
	unsigned a = s % 33, b = s % 31;
	if (a)
		lo = ((lo << a) | (lo >> (33 - a))) & 0x1FFFFFFFFULL;
	if (b)
		hi = ((hi << b) | (hi >> (31 - b))) & 0x7FFFFFFFULL;
	return (hi << 33) | lo;
}

/* NTMC64 base (nthash.hpp:667-692): scans right to left so that the FIRST bad character met is
 * the LAST one in the window; fh = XOR_i srol^{k-1-i}(seed[c_i]), rh = XOR_i srol^{i}(seed[c_i&7]) */
int
bo_base_hash(const char* kmer, unsigned k, uint64_t* fh, uint64_t* rh, unsigned* loc_n)
{
	uint64_t f = 0, r = 0;
	*loc_n = 0;
	for (unsigned i = k; i-- > 0;) {
		if (bo_seed((unsigned char)kmer[i]) == 0) {
			*loc_n = i;
			return 0;
		}
	}
	for (unsigned i = 0; i < k; ++i) {
		unsigned char c = (unsigned char)kmer[i];
		f ^= bo_srol_n(bo_seed(c), k - 1 - i);
		r ^= bo_srol_n(bo_seed(c & 7), i);
	}
	*fh = f;
	*rh = r;
	return 1;
}

/* NTF64/NTR64 rolling (nthash.hpp:442-457) */
void
bo_roll(uint64_t* fh, uint64_t* rh, unsigned k, unsigned char out, unsigned char in)
{
	*fh = bo_srol(*fh) ^ bo_seed(in) ^ bo_srol_n(bo_seed(out), k);
	*rh = bo_sror(*rh ^ bo_srol_n(bo_seed(in & 7), k) ^ bo_seed(out & 7));
}

/* NTE64 (nthash.hpp:537-542): C precedence makes the multiplier i ^ (k * multiSeed) */
uint64_t
bo_extra(uint64_t b, unsigned k, unsigned i)
{
	uint64_t t = b * ((uint64_t)i ^ ((uint64_t)k * MULTI_SEED));
	return t ^ (t >> MULTI_SHIFT);
}

/* multi-hash tail of NTMC64 (nthash.hpp:585-589) */
void
bo_multi(uint64_t b, unsigned k, unsigned h, uint64_t* hv)
{
	hv[0] = b;
	for (unsigned i = 1; i < h; ++i)
		hv[i] = bo_extra(b, k, i);
}

/* ------------------------------------------------------------------------------------------
 * Raw k-mer hashes: NTF64 / NTR64 / NTC64 (kmerSeq, k) of nthash.hpp:394-439,460-465 -- what
 * KmerBloomFilter::insert/contains(const char*) hash with (KmerBloomFilter.hpp:47-74).
 *
 * The reference walks the k-mer through 4-, 3- and 2-mer tables indexed by convertTab /
 * RCconvertTab codes.  tetramerTab[64a+16b+4c+d] = srol^3(s[a]) ^ srol^2(s[b]) ^ srol(s[c]) ^ s[d]
 * (s = the four seeds; likewise trimerTab, dimerTab), and rolx(h,4) + swapxbits033(h,4) = srol^4,
 * so the walk is a Horner chain over base codes.  What differs from the iterator path, restated
 * here as the reference BEHAVES when built with g++ for x86-64 (the build the goldens come from):
 *   - convertTab sends U/u to A and RCconvertTab sends it to T's code, i.e. U reads as A
 *     (nthash.hpp:16-86) -- except in a one-base remainder (k % 4 == 1), which goes through
 *     seedTab and reads U as T (:416-417,432-433);
 *   - k % 4 == 0: the forward walk ends with rolx(h, 0) and swapxbits033(h, 0), whose shifts by
 *     64 execute as shifts by 0 on x86-64: rolx returns h, swapxbits033 returns
 *     h ^ (y | y << 33) with y = h ^ (h >> 33) (:354-356,388-391,404-406); NTR64 has no such step;
 *   - a byte that is not a base has code 255 and the table index is a uint8_t: the index wraps
 *     modulo 256 to some other 4-mer; the offsets into the k-mer are uint8_t too (:401,430) and
 *     wrap for k > 256;
 *   - a remainder of 2 or 3 bases whose wrapped index falls outside dimerTab[16] / trimerTab[64]
 *     reads beyond the table: undefined, no value to restate -> returns 0 (k-mer skipped).
 * ---------------------------------------------------------------------------------------- */
static unsigned
conv_code(unsigned char c) /* convertTab, nthash.hpp:50-83 */
{
	switch (c) {
	case 'A': case 'a': case 'U': case 'u': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': return 3;
	default: return 255;
	}
}
static unsigned
rc_code(unsigned char c) /* RCconvertTab, nthash.hpp:16-49 */
{
	switch (c) {
	case 'A': case 'a': case 'U': case 'u': return 3;
	case 'C': case 'c': return 2;
	case 'G': case 'g': return 1;
	case 'T': case 't': return 0;
	default: return 255;
	}
}
static const uint64_t kmer_seed[4] = {SEED_A, SEED_C, SEED_G, SEED_T};
/* h <- srol^n(h) ^ table_n[idx]: n Horner steps over the base-4 digits of idx, most significant first */
static uint64_t
mer_step(uint64_t h, unsigned idx, unsigned n)
{
	for (unsigned j = n; j-- > 0;)
		h = bo_srol(h) ^ kmer_seed[(idx >> (2 * j)) & 3];
	return h;
}

int
bo_kmer_base_hash(const char* kmer, unsigned k, uint64_t* fh, uint64_t* rh)
{
	const unsigned char* s = (const unsigned char*)kmer;
	const unsigned q = k / 4, r = k % 4;
	uint64_t f = 0, v = 0;
	/* NTF64(kmerSeq, k), nthash.hpp:394-420 */
	for (unsigned i = 0; i < q; ++i) {
		const unsigned off = (4 * i) & 255u; /* uint8_t currOffSet */
		const unsigned loc = (64 * conv_code(s[off]) + 16 * conv_code(s[off + 1]) + 4 * conv_code(s[off + 2]) +
		                      conv_code(s[off + 3])) & 255u;
		f = mer_step(f, loc, 4);
	}
	if (r == 0) {
		const uint64_t y = f ^ (f >> 33);
		f ^= y | (y << 33);
	} else if (r == 3) {
		const unsigned loc = (16 * conv_code(s[k - 3]) + 4 * conv_code(s[k - 2]) + conv_code(s[k - 1])) & 255u;
		if (loc >= 64)
			return 0;
		f = mer_step(f, loc, 3);
	} else if (r == 2) {
		const unsigned loc = (4 * conv_code(s[k - 2]) + conv_code(s[k - 1])) & 255u;
		if (loc >= 16)
			return 0;
		f = mer_step(f, loc, 2);
	} else {
		f = bo_srol(f) ^ bo_seed(s[k - 1]);
	}
	/* NTR64(kmerSeq, k), nthash.hpp:423-439 */
	if (r == 3) {
		const unsigned loc = (16 * rc_code(s[k - 1]) + 4 * rc_code(s[k - 2]) + rc_code(s[k - 3])) & 255u;
		if (loc >= 64)
			return 0;
		v = mer_step(0, loc, 3);
	} else if (r == 2) {
		const unsigned loc = (4 * rc_code(s[k - 1]) + rc_code(s[k - 2])) & 255u;
		if (loc >= 16)
			return 0;
		v = mer_step(0, loc, 2);
	} else if (r == 1) {
		v = bo_seed(s[k - 1] & 7);
	}
	for (unsigned i = 0; i < q; ++i) {
		const unsigned off = (4 * (q - i) - 1) & 255u; /* uint8_t currOffSet */
		const unsigned loc = (64 * rc_code(s[off]) + 16 * rc_code(s[off - 1]) + 4 * rc_code(s[off - 2]) +
		                      rc_code(s[off - 3])) & 255u;
		v = mer_step(v, loc, 4);
	}
	*fh = f;
	*rh = v;
	return 1;
}

/* NTC64(kmerSeq, k) + NTE64 (nthash.hpp:460-465,537-542) for n k-mers laid back to back */
void
bo_kmer_hashes(const char* kmers, size_t n, unsigned k, unsigned h, uint64_t* hash_out, uint8_t* valid_out)
{
	for (size_t i = 0; i < n; ++i) {
		uint64_t fh, rh;
		valid_out[i] = (uint8_t)bo_kmer_base_hash(kmers + i * k, k, &fh, &rh);
		if (valid_out[i])
			bo_multi(rh < fh ? rh : fh, k, h, hash_out + i * h);
		else
			memset(hash_out + i * h, 0, h * sizeof(uint64_t));
	}
}

/* ------------------------------------------------------------------------------------------
 * Iterators
 * ---------------------------------------------------------------------------------------- */

/* ntHashIterator::init (ntHashIterator.hpp:59-70): from pos, skip forward past the last bad
 * character of each dirty window until a clean one is found; BO_NPOS at end. */
static size_t
seek_clean(const char* seq, size_t len, unsigned k, size_t pos, uint64_t* fh, uint64_t* rh)
{
	if (k > len)
		return BO_NPOS;
	unsigned loc_n = 0;
	while (pos < len - k + 1 && !bo_base_hash(seq + pos, k, fh, rh, &loc_n))
		pos += (size_t)loc_n + 1;
	if (pos >= len - k + 1)
		return BO_NPOS;
	return pos;
}

/* ntHashIterator walk: init() :59-70, next() :73-86, operator* :93 */
size_t
bo_nthash_seq(const char* seq, size_t len, unsigned h, unsigned k,
              uint64_t* pos_out, uint64_t* hash_out, size_t cap)
{
	uint64_t fh = 0, rh = 0;
	size_t n = 0;
	size_t pos = seek_clean(seq, len, k, 0, &fh, &rh);
	while (pos != BO_NPOS) {
		if (n < cap) {
			pos_out[n] = pos;
			bo_multi(rh < fh ? rh : fh, k, h, hash_out + n * h);
		}
		++n;
		/* next(): */
		++pos;
		if (pos >= len - k + 1)
			break;
		if (bo_seed((unsigned char)seq[pos + k - 1]) == 0)
			pos = seek_clean(seq, len, k, pos + k, &fh, &rh);
		else
			bo_roll(&fh, &rh, k, (unsigned char)seq[pos - 1], (unsigned char)seq[pos - 1 + k]);
	}
	return n;
}

/* NTMSM64 masking step (nthash.hpp:838-853 and :866-877): for every don't-care position i of
 * seed j, XOR its contribution back out of both strands; hStn = rs < fs; h2-1 extra hashes. */
static void
spaced_hashes(const char* win, unsigned k, const char* const* seeds, unsigned nseeds, unsigned h2,
              uint64_t fh, uint64_t rh, uint64_t* hv, uint8_t* stn)
{
	for (unsigned j = 0; j < nseeds; ++j) {
		uint64_t fs = fh, rs = rh;
		for (unsigned i = 0; i < k; ++i) {
			if (seeds[j][i] != '1') { /* parseSeed: indices of non-'1' (stHashIterator.hpp:27-29) */
				unsigned char c = (unsigned char)win[i];
				fs ^= bo_srol_n(bo_seed(c), k - 1 - i);
				rs ^= bo_srol_n(bo_seed(c & 7), i);
			}
		}
		uint8_t s = rs < fs;
		uint64_t b = s ? rs : fs;
		hv[j * h2] = b;
		stn[j * h2] = s;
		for (unsigned j2 = 1; j2 < h2; ++j2) {
			hv[j * h2 + j2] = bo_extra(b, k, j2);
			stn[j * h2 + j2] = s;
		}
	}
}

/* stHashIterator walk (stHashIterator.hpp:60-87).  The clean-window rule is the same as
 * ntHashIterator's: don't-care positions still have to be valid bases (nthash.hpp:825-829). */
size_t
bo_sthash_seq(const char* seq, size_t len, const char* const* seeds, unsigned nseeds,
              unsigned h2, unsigned k, uint64_t* pos_out, uint64_t* hash_out,
              uint8_t* strand_out, size_t cap)
{
	const unsigned m = nseeds * h2;
	uint64_t fh = 0, rh = 0;
	size_t n = 0;
	size_t pos = seek_clean(seq, len, k, 0, &fh, &rh);
	while (pos != BO_NPOS) {
		if (n < cap) {
			pos_out[n] = pos;
			spaced_hashes(seq + pos, k, seeds, nseeds, h2, fh, rh, hash_out + n * m,
			              strand_out + n * m);
		}
		++n;
		++pos;
		if (pos >= len - k + 1)
			break;
		if (bo_seed((unsigned char)seq[pos + k - 1]) == 0)
			pos = seek_clean(seq, len, k, pos + k, &fh, &rh);
		else
			bo_roll(&fh, &rh, k, (unsigned char)seq[pos - 1], (unsigned char)seq[pos - 1 + k]);
	}
	return n;
}

/* ------------------------------------------------------------------------------------------
 * Bit filter (BloomFilter.hpp): pos = hash % m_size; bit (pos%8), LSB first, of byte pos/8
 * ---------------------------------------------------------------------------------------- */

/* insert(const uint64_t[]) BloomFilter.hpp:185-194 */
void
bo_bf_insert(uint8_t* filt, uint64_t size_bits, unsigned h, const uint64_t* hashes, size_t n)
{
	for (size_t i = 0; i < n; ++i)
		for (unsigned j = 0; j < h; ++j) {
			uint64_t p = hashes[i * h + j] % size_bits;
			filt[p >> 3] |= (uint8_t)(1u << (p & 7));
		}
}

/* contains(const uint64_t[]) BloomFilter.hpp:252-262 */
void
bo_bf_contains(const uint8_t* filt, uint64_t size_bits, unsigned h, const uint64_t* hashes,
               size_t n, uint8_t* out)
{
	for (size_t i = 0; i < n; ++i) {
		uint8_t all = 1;
		for (unsigned j = 0; j < h; ++j) {
			uint64_t p = hashes[i * h + j] % size_bits;
			all &= (filt[p >> 3] >> (p & 7)) & 1;
		}
		out[i] = all;
	}
}

/* insertAndCheck(const uint64_t[]) BloomFilter.hpp:200-214: AND of the PREVIOUS bits */
void
bo_bf_insert_and_check(uint8_t* filt, uint64_t size_bits, unsigned h, const uint64_t* hashes,
                       size_t n, uint8_t* out)
{
	for (size_t i = 0; i < n; ++i) {
		uint8_t all = 1;
		for (unsigned j = 0; j < h; ++j) {
			uint64_t p = hashes[i * h + j] % size_bits;
			all &= (filt[p >> 3] >> (p & 7)) & 1;
			filt[p >> 3] |= (uint8_t)(1u << (p & 7));
		}
		out[i] = all;
	}
}

/* getPop BloomFilter.hpp:316-323 */
uint64_t
bo_bf_popcount(const uint8_t* filt, uint64_t size_bits)
{
	uint64_t n = 0, nb = (size_bits + 7) / 8;
	for (uint64_t i = 0; i < nb; ++i)
		n += (uint64_t)__builtin_popcount(filt[i]);
	return n;
}

/* insertSeq BloomFilterUtil.h:9-17 */
void
bo_bf_insert_seq(uint8_t* filt, uint64_t size_bits, unsigned h, unsigned k,
                 const char* seq, size_t len)
{
	uint64_t fh = 0, rh = 0, hv[64];
	if (h > 64)
		return;
	size_t pos = seek_clean(seq, len, k, 0, &fh, &rh);
	while (pos != BO_NPOS) {
		bo_multi(rh < fh ? rh : fh, k, h, hv);
		bo_bf_insert(filt, size_bits, h, hv, 1);
		++pos;
		if (pos >= len - k + 1)
			break;
		if (bo_seed((unsigned char)seq[pos + k - 1]) == 0)
			pos = seek_clean(seq, len, k, pos + k, &fh, &rh);
		else
			bo_roll(&fh, &rh, k, (unsigned char)seq[pos - 1], (unsigned char)seq[pos - 1 + k]);
	}
}

void
bo_bf_contains_seq_dense(const uint8_t* filt, uint64_t size_bits, unsigned h, unsigned k,
                         const char* seq, size_t len, uint8_t* hit, uint8_t* valid)
{
	uint64_t fh = 0, rh = 0, hv[64];
	if (len >= k) {
		memset(hit, 0, len - k + 1);
		memset(valid, 0, len - k + 1);
	}
	if (h > 64)
		return;
	size_t pos = seek_clean(seq, len, k, 0, &fh, &rh);
	while (pos != BO_NPOS) {
		bo_multi(rh < fh ? rh : fh, k, h, hv);
		valid[pos] = 1;
		bo_bf_contains(filt, size_bits, h, hv, 1, hit + pos);
		++pos;
		if (pos >= len - k + 1)
			break;
		if (bo_seed((unsigned char)seq[pos + k - 1]) == 0)
			pos = seek_clean(seq, len, k, pos + k, &fh, &rh);
		else
			bo_roll(&fh, &rh, k, (unsigned char)seq[pos - 1], (unsigned char)seq[pos - 1 + k]);
	}
}

/* ------------------------------------------------------------------------------------------
 * Counting filter, uint8_t (CountingBloomFilter.hpp)
 * ---------------------------------------------------------------------------------------- */

/* ctor: bytes rounded up to a multiple of 8 (CountingBloomFilter.hpp:40-49) */
uint64_t
bo_cbf_round_bytes(uint64_t bytes)
{
	uint64_t r = bytes % 8;
	return r ? bytes + 8 - r : bytes;
}

/* minCount :53-64 */
uint8_t
bo_cbf_min(const uint8_t* c, uint64_t size, unsigned h, const uint64_t* hv)
{
	uint8_t m = c[hv[0] % size];
	for (unsigned i = 1; i < h; ++i) {
		uint8_t v = c[hv[i] % size];
		if (v < m)
			m = v;
	}
	return m;
}

/* incrementMin :135-162, serial: every counter equal to the minimum becomes min+1; a minimum
 * of 255 is left alone (the "minVal > newVal" wrap test, :146-149) */
void
bo_cbf_increment_min(uint8_t* c, uint64_t size, unsigned h, const uint64_t* hashes, size_t n)
{
	for (size_t i = 0; i < n; ++i) {
		const uint64_t* hv = hashes + i * h;
		uint8_t m = bo_cbf_min(c, size, h, hv);
		if (m == 255)
			continue;
		/* the CAS at :152 is applied position by position: a position that appears twice in
		 * hv is bumped only once because its value no longer equals minVal the second time */
		for (unsigned j = 0; j < h; ++j) {
			uint64_t p = hv[j] % size;
			if (c[p] == m)
				c[p] = (uint8_t)(m + 1);
		}
	}
}

/* incrementAll :165-183: saturating +1 on every position (a repeated position is bumped twice) */
void
bo_cbf_increment_all(uint8_t* c, uint64_t size, unsigned h, const uint64_t* hashes, size_t n)
{
	for (size_t i = 0; i < n; ++i)
		for (unsigned j = 0; j < h; ++j) {
			uint64_t p = hashes[i * h + j] % size;
			if (c[p] != 255)
				c[p]++;
		}
}

/* insertAndCheck :206-214 = contains then incrementMin */
void
bo_cbf_insert_and_check(uint8_t* c, uint64_t size, unsigned h, unsigned thr,
                        const uint64_t* hashes, size_t n, uint8_t* out)
{
	for (size_t i = 0; i < n; ++i) {
		out[i] = bo_cbf_min(c, size, h, hashes + i * h) >= thr;
		bo_cbf_increment_min(c, size, h, hashes + i * h, 1);
	}
}

/* contains :190-196 */
void
bo_cbf_query(const uint8_t* c, uint64_t size, unsigned h, unsigned thr,
             const uint64_t* hashes, size_t n, uint8_t* min_out, uint8_t* contains_out)
{
	for (size_t i = 0; i < n; ++i) {
		uint8_t m = bo_cbf_min(c, size, h, hashes + i * h);
		if (min_out)
			min_out[i] = m;
		if (contains_out)
			contains_out[i] = m >= thr;
	}
}

uint64_t
bo_cbf_popcount(const uint8_t* c, uint64_t size)
{
	uint64_t n = 0;
	for (uint64_t i = 0; i < size; ++i)
		n += c[i] != 0;
	return n;
}

uint64_t
bo_cbf_filtered_popcount(const uint8_t* c, uint64_t size, unsigned thr)
{
	uint64_t n = 0;
	for (uint64_t i = 0; i < size; ++i)
		n += c[i] >= thr;
	return n;
}

/* ------------------------------------------------------------------------------------------
 * .bf headers.  The key order is what libstdc++'s unordered_map yields for these keys in the
 * reference's insertion order (cpptoml.h:43-52,3332); doubles are printed with
 * showpoint + 17 significant digits and "e0"/"e-0" squeezed (cpptoml.h:3477-3494).
 * ---------------------------------------------------------------------------------------- */

static void
toml_double(double v, char* out, size_t cap)
{
	snprintf(out, cap, "%#.17g", v);
	char* p = strstr(out, "e0");
	if (p)
		memmove(p + 1, p + 2, strlen(p + 2) + 1);
	p = strstr(out, "e-0");
	if (p)
		memmove(p + 2, p + 3, strlen(p + 3) + 1);
}

int
bo_bf_header(char* buf, size_t cap, uint64_t size_bits, unsigned h, unsigned k,
             double dfpr, uint64_t n_entry, uint64_t t_entry)
{
	char d[64];
	toml_double(dfpr, d, sizeof d);
	return snprintf(buf, cap,
	                "[BTLBloomFilter_v1]\n"
	                "\tnEntry = %llu\n"
	                "\tdFPR = %s\n"
	                "\tEntry = %llu\n"
	                "\tBloomFilterSizeInBytes = %llu\n"
	                "\tBloomFilterSize = %llu\n"
	                "\tHashNum = %u\n"
	                "\tKmerSize = %u\n"
	                "[HeaderEnd]\n",
	                (unsigned long long)n_entry, d, (unsigned long long)t_entry,
	                (unsigned long long)(size_bits / 8), (unsigned long long)size_bits, h, k);
}

int
bo_cbf_header(char* buf, size_t cap, uint64_t size, uint64_t size_bytes, unsigned h, unsigned k,
              unsigned bits_per_counter)
{
	return snprintf(buf, cap,
	                "[BTLCountingBloomFilter_v1]\n"
	                "\tBloomFilterSize = %llu\n"
	                "\tHashNum = %u\n"
	                "\tKmerSize = %u\n"
	                "\tBloomFilterSizeInBytes = %llu\n"
	                "\tBitsPerCounter = %u\n"
	                "[HeaderEnd]\n",
	                (unsigned long long)size, h, k, (unsigned long long)size_bytes,
	                bits_per_counter);
}

/* ------------------------------------------------------------------------------------------
 * Synthetic reads and the timed CPU port of the hot loop
 * ---------------------------------------------------------------------------------------- */

static inline uint64_t
mix64(uint64_t z)
{
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

static void
synth_read(uint64_t seed, uint64_t r, unsigned read_len, char* out)
{
	const unsigned wpr = (read_len + 31) / 32;
	for (unsigned j = 0; j < read_len; ++j) {
		uint64_t n = r * wpr + j / 32;
		uint64_t w = mix64(seed + (n + 1) * 0x9E3779B97F4A7C15ULL);
		out[j] = "ACGT"[(w >> (2 * (j % 32))) & 3];
	}
}

void
bo_synth_reads(uint64_t seed, uint64_t first, uint64_t n, unsigned read_len, char* out)
{
	for (uint64_t r = 0; r < n; ++r)
		synth_read(seed, first + r, read_len, out + r * read_len);
}

static double
now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* Same loop shape as the reference under OpenMP (Tests/AdHoc/ParallelFilter.cpp:109-121):
 * threads take reads, each walks its read with the rolling iterator and probes with byte
 * atomics (BloomFilter.hpp:191).  out[] as in ref_bench_bf (oracle/ref_driver.cpp). */
int
bo_bench_bf(uint64_t n_reads, unsigned read_len, unsigned k, unsigned h, uint64_t bits,
            uint64_t seed_ins, uint64_t seed_qry, int threads, int prefault, double* out)
{
	if (h > 64 || read_len > 4096 || bits % 8)
		return -1;
	uint8_t* filt = (uint8_t*)calloc(bits / 8, 1);
	if (!filt)
		return -2;
	int used = 1;
#ifdef _OPENMP
	if (threads > 0)
		omp_set_num_threads(threads);
	used = omp_get_max_threads();
#endif
	if (prefault) {
		const int64_t nb = (int64_t)(bits / 8);
#pragma omp parallel for schedule(static)
		for (int64_t i = 0; i < nb; i += 4096)
			((volatile uint8_t*)filt)[i] = 0;
	}
	double t0 = now_s();
#pragma omp parallel
	{
		char s[4096];
		uint64_t hv[64], fh = 0, rh = 0;
#pragma omp for schedule(dynamic, 1024)
		for (int64_t r = 0; r < (int64_t)n_reads; ++r) {
			synth_read(seed_ins, (uint64_t)r, read_len, s);
			size_t pos = seek_clean(s, read_len, k, 0, &fh, &rh);
			while (pos != BO_NPOS) {
				bo_multi(rh < fh ? rh : fh, k, h, hv);
				for (unsigned j = 0; j < h; ++j) {
					uint64_t p = hv[j] % bits;
					__atomic_fetch_or(&filt[p >> 3], (uint8_t)(1u << (p & 7)), __ATOMIC_RELAXED);
				}
				++pos;
				if (pos >= read_len - k + 1)
					break;
				if (bo_seed((unsigned char)s[pos + k - 1]) == 0)
					pos = seek_clean(s, read_len, k, pos + k, &fh, &rh);
				else
					bo_roll(&fh, &rh, k, (unsigned char)s[pos - 1], (unsigned char)s[pos - 1 + k]);
			}
		}
	}
	double t1 = now_s();
	uint64_t hits = 0, kmers = 0;
#pragma omp parallel reduction(+ : hits, kmers)
	{
		char s[4096];
		uint64_t hv[64], fh = 0, rh = 0;
#pragma omp for schedule(dynamic, 1024)
		for (int64_t r = 0; r < (int64_t)n_reads; ++r) {
			synth_read(seed_qry, (uint64_t)r, read_len, s);
			size_t pos = seek_clean(s, read_len, k, 0, &fh, &rh);
			while (pos != BO_NPOS) {
				bo_multi(rh < fh ? rh : fh, k, h, hv);
				uint8_t all = 1;
				for (unsigned j = 0; j < h && all; ++j) {
					uint64_t p = hv[j] % bits;
					all = (filt[p >> 3] >> (p & 7)) & 1;
				}
				hits += all;
				++kmers;
				++pos;
				if (pos >= read_len - k + 1)
					break;
				if (bo_seed((unsigned char)s[pos + k - 1]) == 0)
					pos = seek_clean(s, read_len, k, pos + k, &fh, &rh);
				else
					bo_roll(&fh, &rh, k, (unsigned char)s[pos - 1], (unsigned char)s[pos - 1 + k]);
			}
		}
	}
	double t2 = now_s();
	out[0] = t1 - t0;
	out[1] = t2 - t1;
	out[2] = (double)hits;
	out[3] = (double)kmers;
	out[4] = (double)used;
	out[5] = (double)bo_bf_popcount(filt, bits);
	free(filt);
	return 0;
}

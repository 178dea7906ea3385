// csrc/fastx.cpp -- FASTA / FASTQ ingestion for the batch path.
//
// The reference only has toy loaders: Tests/AdHoc/ParallelFilter.cpp:104-122 (`loadBf`: header line +
// one sequence line per record, one ntHashIterator per line under OpenMP) and
// swig/writeBloom_rolling.cpp:18-59 (`contigsToBloom`: multi-line FASTA, the lines of a record are
// concatenated and handed to insertSeq).  This file is what a real caller needs instead:
//   * btlbf_fastx_open/next/close : a streaming parser (plain or gzip input through zlib) that fills
//     batches of raw sequence bytes + a starts[] offset table (the ragged layout of btlbf_layout);
//     sequences longer than a batch are cut with a k-1 base overlap, so every window appears in
//     exactly one batch
//   * btlbf_insert_fastx / btlbf_contains_fastx : parse into pinned host buffers, copy to HBM on a
//     copy stream and run the fused kernels on a compute stream, double-buffered -- the host parses
//     batch i+1 while batch i is copied and hashed
// Hashing semantics are those of the kernels (a window is skipped unless all k bytes are ACGTU in
// either case -- the reference iterator's rule, vendor/ntHashIterator.hpp:59-86).
#include "host_internal.hpp"

#include <hip/hip_runtime.h>
#include <zlib.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>

namespace {

constexpr size_t kInBuf = 8u << 20; // raw input window

double now_s()
{
	using namespace std::chrono;
	return duration<double>(steady_clock::now().time_since_epoch()).count();
}

} // namespace

struct btlbf_fastx {
	gzFile gz = nullptr;
	char* in = nullptr;
	size_t in_pos = 0, in_len = 0;
	bool eof = false;
	// two output batches: the one handed out by the previous call stays intact during the next parse
	char* bases[2] = {nullptr, nullptr};
	uint64_t* starts[2] = {nullptr, nullptr};
	bool pinned = false;
	uint64_t cap_bases = 0, cap_seqs = 0;
	int cur = 0;
	uint32_t k = 1, flags = 0;
	// parser state, persistent across batches
	enum Fmt { UNKNOWN, FASTA, FASTQ, PLAIN } fmt = UNKNOWN;
	enum St { LINE_START, SKIP_LINE, IN_SEQ } st = LINE_START;
	int fq_line = 0;        // FASTQ: 0 header, 1 sequence, 2 '+', 3 quality
	bool seq_open = false;  // a sequence is being appended (FASTA records span lines)
	uint64_t seq_in_batch = 0; // bases of the open sequence that sit in the current batch
	bool cont = false;      // the previous batch was cut inside a sequence: this one continues it ...
	char* carry = nullptr;  // ... starting with its last k-1 bases again
	uint32_t carry_len = 0;
	uint64_t n_records = 0;
	double seconds_parse = 0;

	~btlbf_fastx()
	{
		if (gz)
			gzclose(gz);
		free(in);
		free(carry);
		for (int i = 0; i < 2; ++i) {
			if (pinned) {
				if (bases[i])
					(void)hipHostFree(bases[i]);
				if (starts[i])
					(void)hipHostFree(starts[i]);
			} else {
				free(bases[i]);
				free(starts[i]);
			}
		}
	}
};

namespace {

bool refill(btlbf_fastx* r)
{
	if (r->eof)
		return false;
	// keep the unread tail (at most a lone '\r' or nothing)
	const size_t tail = r->in_len - r->in_pos;
	if (tail)
		memmove(r->in, r->in + r->in_pos, tail);
	r->in_pos = 0;
	r->in_len = tail;
	const int got = gzread(r->gz, r->in + tail, (unsigned)(kInBuf - tail));
	if (got <= 0) {
		r->eof = true;
		return r->in_len > 0;
	}
	r->in_len += (size_t)got;
	return true;
}

} // namespace

extern "C" int btlbf_fastx_open(btlbf_fastx** out, const char* path, uint32_t flags, uint32_t k, uint64_t batch_bytes)
{
	if (!out || !path || k == 0)
		return btlbf_set_error(BTLBF_EINVAL, "fastx_open: null argument or k == 0");
	if (batch_bytes == 0)
		batch_bytes = 256ull << 20;
	if (batch_bytes < 4ull * k + 64)
		return btlbf_set_error(BTLBF_EINVAL, "fastx_open: batch of %llu bytes is too small for k = %u",
		                       (unsigned long long)batch_bytes, k);
	btlbf_fastx* r = new (std::nothrow) btlbf_fastx;
	if (!r)
		return btlbf_set_error(BTLBF_ENOMEM, "fastx_open: out of memory");
	r->gz = gzopen(path, "rb"); // transparent for uncompressed input
	if (!r->gz) {
		delete r;
		return btlbf_set_error(BTLBF_EIO, "file \"%s\" could not be read.", path);
	}
	(void)gzbuffer(r->gz, 1u << 20);
	r->k = k;
	r->flags = flags;
	r->cap_bases = batch_bytes;
	r->cap_seqs = batch_bytes / 16 + 1024;
	r->in = static_cast<char*>(malloc(kInBuf));
	r->carry = static_cast<char*>(malloc(k));
	bool ok = r->in && r->carry;
	// pinned memory when a GPU is there (async copies); plain memory otherwise (parser-only use)
	int ndev = 0;
	r->pinned = !(flags & BTLBF_FASTX_PAGEABLE) && hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0;
	for (int i = 0; i < 2 && ok; ++i) {
		const size_t bb = r->cap_bases + 64, sb = (r->cap_seqs + 2) * sizeof(uint64_t);
		if (r->pinned) {
			ok = hipHostMalloc(reinterpret_cast<void**>(&r->bases[i]), bb, hipHostMallocDefault) == hipSuccess &&
			     hipHostMalloc(reinterpret_cast<void**>(&r->starts[i]), sb, hipHostMallocDefault) == hipSuccess;
		} else {
			r->bases[i] = static_cast<char*>(malloc(bb));
			r->starts[i] = static_cast<uint64_t*>(malloc(sb));
			ok = r->bases[i] && r->starts[i];
		}
	}
	if (!ok) {
		delete r;
		return btlbf_set_error(BTLBF_ENOMEM, "fastx_open: could not allocate the batch buffers");
	}
	*out = r;
	return BTLBF_OK;
}

extern "C" void btlbf_fastx_close(btlbf_fastx* r) { delete r; }

extern "C" uint64_t btlbf_fastx_records(const btlbf_fastx* r) { return r ? r->n_records : 0; }

// Next batch: *bases (n_bases bytes) and *starts (n_seqs + 1 offsets, starts[n_seqs] == n_bases) stay
// valid until the call after the next one.  n_seqs == 0 at end of input.
extern "C" int btlbf_fastx_next(btlbf_fastx* r, const char** bases, uint64_t* n_bases, const uint64_t** starts,
                                uint64_t* n_seqs)
{
	if (!r || !bases || !n_bases || !starts || !n_seqs)
		return btlbf_set_error(BTLBF_EINVAL, "fastx_next: null argument");
	const double t0 = now_s();
	r->cur ^= 1;
	char* out = r->bases[r->cur];
	uint64_t* st = r->starts[r->cur];
	uint64_t nb = 0, ns = 0;
	const bool per_line = (r->flags & BTLBF_FASTX_LINES) != 0;
	auto open_seq = [&]() {
		st[ns++] = nb;
		r->seq_open = true;
		r->seq_in_batch = 0;
	};
	if (r->cont) {
		// continuation of a sequence cut at the previous batch boundary: its last k-1 bases again
		open_seq();
		memcpy(out, r->carry, r->carry_len);
		nb = r->seq_in_batch = r->carry_len;
		r->cont = false;
	}
	bool full = false;
	auto cut = [&]() {
		const uint64_t c = r->seq_in_batch < r->k - 1 ? r->seq_in_batch : r->k - 1;
		memcpy(r->carry, out + nb - c, c);
		r->carry_len = (uint32_t)c;
		r->cont = true;
	};
	while (!full) {
		if (r->in_pos == r->in_len && !refill(r))
			break;
		char* p = r->in + r->in_pos;
		const size_t avail = r->in_len - r->in_pos;
		if (r->st == btlbf_fastx::LINE_START) {
			const char c = *p;
			if (c == '\n' || c == '\r') {
				// empty line: ignored, except that an empty FASTQ line still is one of the record's four
				// (zero-length reads exist)
				++r->in_pos;
				if (c == '\n' && r->fmt == btlbf_fastx::FASTQ)
					r->fq_line = (r->fq_line + 1) & 3;
				continue;
			}
			if (r->fmt == btlbf_fastx::UNKNOWN)
				r->fmt = c == '>' ? btlbf_fastx::FASTA : c == '@' ? btlbf_fastx::FASTQ : btlbf_fastx::PLAIN;
			bool is_seq;
			if (r->fmt == btlbf_fastx::FASTQ) {
				is_seq = r->fq_line == 1;
				if (r->fq_line == 0) {
					++r->n_records;
					r->seq_open = false;
				}
			} else if (r->fmt == btlbf_fastx::FASTA) {
				is_seq = c != '>' && c != ';';
				if (c == '>') {
					++r->n_records;
					r->seq_open = false;
				}
			} else {
				is_seq = true;
				++r->n_records;
				r->seq_open = false;
			}
			if (is_seq) {
				// FASTA records continue across lines unless the caller wants one sequence per line
				if (!r->seq_open || per_line || r->fmt != btlbf_fastx::FASTA) {
					if (ns >= r->cap_seqs) {
						full = true;
						break;
					}
					open_seq();
				}
				r->st = btlbf_fastx::IN_SEQ;
			} else {
				r->st = btlbf_fastx::SKIP_LINE;
			}
			continue;
		}
		char* nl = static_cast<char*>(memchr(p, '\n', avail));
		size_t n = nl ? (size_t)(nl - p) : avail;
		if (r->st == btlbf_fastx::SKIP_LINE) {
			r->in_pos += n;
		} else {
			// sequence bytes [p, p+n); a '\r' before the newline is dropped, one at the very end of the
			// window waits for the next refill (it may precede a newline)
			size_t take = n;
			if (take && p[take - 1] == '\r') {
				if (nl || r->eof)
					--take;
				else if (take > 1)
					--take, n = take; // leave the '\r' unread
				else {
					if (!refill(r)) { // lone '\r' at end of input
						r->in_pos = r->in_len;
						break;
					}
					continue;
				}
			}
			const uint64_t room = r->cap_bases - nb;
			if (take > room) {
				take = room;
				n = take;
				nl = nullptr;
				full = true;
			}
			memcpy(out + nb, p, take);
			nb += take;
			r->seq_in_batch += take;
			r->in_pos += n;
			if (full) {
				cut(); // mid-line: the next batch restarts this sequence k-1 bases back
				break;
			}
		}
		if (nl) {
			++r->in_pos; // the newline
			r->st = btlbf_fastx::LINE_START;
			if (r->fmt == btlbf_fastx::FASTQ)
				r->fq_line = (r->fq_line + 1) & 3;
			if (nb == r->cap_bases)
				full = true; // exactly full at a line end: nothing to carry unless the record goes on
			if (full && r->seq_open && r->fmt == btlbf_fastx::FASTA && !per_line)
				cut(); // the record may go on in the next line
		}
	}
	if (r->cont && r->eof && r->in_pos == r->in_len)
		r->cont = false; // nothing follows
	st[ns] = nb;
	*bases = out;
	*n_bases = nb;
	*starts = st;
	*n_seqs = ns;
	r->seconds_parse += now_s() - t0;
	return BTLBF_OK;
}

// -------------------------------------------------------------------------------------------------
// file -> filter
// -------------------------------------------------------------------------------------------------
namespace {

#define HIP_TRY_X(expr)                                                                              \
	do {                                                                                             \
		hipError_t e__ = (expr);                                                                     \
		if (e__ != hipSuccess) {                                                                     \
			rc = btlbf_set_error(BTLBF_EHIP, "%s failed: %s", #expr, hipGetErrorString(e__));        \
			goto done;                                                                               \
		}                                                                                            \
	} while (0)

int run_fastx(btlbf_filter* f, const char* path, uint32_t flags, uint64_t batch_bytes, bool query,
              btlbf_fastx_stats* stats)
{
	if (!f || !path)
		return btlbf_set_error(BTLBF_EINVAL, "fastx: null argument");
	const double t0 = now_s();
	btlbf_fastx* r = nullptr;
	int rc = btlbf_fastx_open(&r, path, flags, btlbf_kmer_size(f), batch_bytes);
	if (rc)
		return rc;
	int prev_dev = 0;
	(void)hipGetDevice(&prev_dev);
	hipStream_t copy_s = nullptr, comp_s = nullptr;
	hipEvent_t copied[2] = {nullptr, nullptr}, done_ev[2] = {nullptr, nullptr};
	char* d_bases[2] = {nullptr, nullptr};
	uint64_t* d_starts[2] = {nullptr, nullptr};
	uint64_t* d_counts = nullptr; // [2 slots][2]
	uint64_t* h_counts = nullptr; // pinned mirror
	bool used[2] = {false, false};
	btlbf_fastx_stats st;
	memset(&st, 0, sizeof st);
	auto harvest = [&](int slot) {
		st.n_windows += h_counts[slot * 2 + 0];
		st.n_hits += h_counts[slot * 2 + 1];
	};
	HIP_TRY_X(hipSetDevice(btlbf_device(f)));
	HIP_TRY_X(hipStreamCreateWithFlags(&copy_s, hipStreamNonBlocking));
	HIP_TRY_X(hipStreamCreateWithFlags(&comp_s, hipStreamNonBlocking));
	HIP_TRY_X(hipHostMalloc(reinterpret_cast<void**>(&h_counts), 4 * sizeof(uint64_t), hipHostMallocDefault));
	HIP_TRY_X(hipMalloc(reinterpret_cast<void**>(&d_counts), 4 * sizeof(uint64_t)));
	memset(h_counts, 0, 4 * sizeof(uint64_t));
	for (int i = 0; i < 2; ++i) {
		HIP_TRY_X(hipEventCreateWithFlags(&copied[i], hipEventDisableTiming));
		HIP_TRY_X(hipEventCreateWithFlags(&done_ev[i], hipEventDisableTiming));
		HIP_TRY_X(hipMalloc(reinterpret_cast<void**>(&d_bases[i]), r->cap_bases + 64));
		HIP_TRY_X(hipMalloc(reinterpret_cast<void**>(&d_starts[i]), (r->cap_seqs + 2) * sizeof(uint64_t)));
	}
	for (int slot = 0;; slot ^= 1) {
		if (used[slot]) {
			// the slot's previous batch must be through the kernels before its buffers are reused
			HIP_TRY_X(hipEventSynchronize(done_ev[slot]));
			harvest(slot);
			used[slot] = false;
		}
		const char* hb;
		const uint64_t* hs;
		uint64_t nb, ns;
		if ((rc = btlbf_fastx_next(r, &hb, &nb, &hs, &ns)))
			goto done;
		if (ns == 0)
			break;
		st.n_bases += nb;
		++st.n_batches;
		if (nb < btlbf_kmer_size(f))
			continue; // no window fits
		HIP_TRY_X(hipMemcpyAsync(d_bases[slot], hb, nb, hipMemcpyHostToDevice, copy_s));
		HIP_TRY_X(hipMemcpyAsync(d_starts[slot], hs, (ns + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, copy_s));
		HIP_TRY_X(hipEventRecord(copied[slot], copy_s));
		HIP_TRY_X(hipStreamWaitEvent(comp_s, copied[slot], 0));
		btlbf_layout lay;
		lay.starts = d_starts[slot];
		lay.n_seqs = ns;
		lay.read_len = 0;
		if (query) {
			rc = btlbf_contains_seqs(f, d_bases[slot], nb, &lay, nullptr, nullptr, d_counts + slot * 2, BTLBF_DEVICE,
			                         comp_s);
		} else {
			rc = btlbf_insert_seqs(f, d_bases[slot], nb, &lay, 0, BTLBF_ORDER_PARALLEL, BTLBF_DEVICE, comp_s);
		}
		if (rc)
			goto done;
		if (query)
			HIP_TRY_X(hipMemcpyAsync(h_counts + slot * 2, d_counts + slot * 2, 2 * sizeof(uint64_t),
			                         hipMemcpyDeviceToHost, comp_s));
		HIP_TRY_X(hipEventRecord(done_ev[slot], comp_s));
		used[slot] = true;
	}
	HIP_TRY_X(hipStreamSynchronize(comp_s));
	for (int i = 0; i < 2; ++i)
		if (used[i])
			harvest(i);
done:
	if (comp_s)
		(void)hipStreamSynchronize(comp_s);
	if (copy_s)
		(void)hipStreamSynchronize(copy_s);
	st.n_records = r->n_records;
	st.seconds_parse = r->seconds_parse;
	for (int i = 0; i < 2; ++i) {
		if (d_bases[i])
			(void)hipFree(d_bases[i]);
		if (d_starts[i])
			(void)hipFree(d_starts[i]);
		if (copied[i])
			(void)hipEventDestroy(copied[i]);
		if (done_ev[i])
			(void)hipEventDestroy(done_ev[i]);
	}
	if (d_counts)
		(void)hipFree(d_counts);
	if (h_counts)
		(void)hipHostFree(h_counts);
	if (copy_s)
		(void)hipStreamDestroy(copy_s);
	if (comp_s)
		(void)hipStreamDestroy(comp_s);
	btlbf_fastx_close(r);
	(void)hipSetDevice(prev_dev);
	st.seconds_total = now_s() - t0;
	if (stats)
		*stats = st;
	return rc;
}

} // namespace

extern "C" int btlbf_insert_fastx(btlbf_filter* f, const char* path, uint32_t flags, uint64_t batch_bytes,
                                  btlbf_fastx_stats* stats)
{
	return run_fastx(f, path, flags, batch_bytes, false, stats);
}

extern "C" int btlbf_contains_fastx(btlbf_filter* f, const char* path, uint32_t flags, uint64_t batch_bytes,
                                    btlbf_fastx_stats* stats)
{
	return run_fastx(f, path, flags, batch_bytes, true, stats);
}

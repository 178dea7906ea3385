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

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

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
	// plain (uncompressed) input read by file offset: several readers can then share one file, each
	// parsing the records that START inside its byte range [.., range_end)
	int fd = -1;
	uint64_t read_pos = 0;            // next file offset to read
	uint64_t buf_off = 0;             // file offset of in[0]
	uint64_t range_end = UINT64_MAX;  // stop before the first record that starts at or after this offset
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
		if (fd >= 0)
			close(fd);
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
	r->buf_off += r->in_pos;
	r->in_pos = 0;
	r->in_len = tail;
	long got;
	if (r->fd >= 0) {
		got = (long)pread(r->fd, r->in + tail, kInBuf - tail, (off_t)r->read_pos);
		if (got > 0)
			r->read_pos += (uint64_t)got;
	} else {
		got = gzread(r->gz, r->in + tail, (unsigned)(kInBuf - tail));
	}
	if (got <= 0) {
		r->eof = true;
		return r->in_len > 0;
	}
	r->in_len += (size_t)got;
	return true;
}

} // namespace

namespace {

// first record start at or after file offset `begin` (file_size if there is none): FASTA '>' at a line
// start; 4-line FASTQ: a line starting with '@' whose second successor starts with '+' (a quality line
// may start with '@', but then the line two below it is a sequence, never a '+'); plain: any line start
// pread that is not defeated by a signal: retried on EINTR, short reads are the caller's business
static ssize_t pread_retry(int fd, void* buf, size_t n, off_t off)
{
	for (;;) {
		const ssize_t r = pread(fd, buf, n, off);
		if (r >= 0 || errno != EINTR)
			return r;
	}
}

// offset of the byte behind the first '\n' at or after `from` (file_size if there is none), reading through a
// small window that is the caller's and grows only for a line longer than it
static uint64_t next_line(int fd, uint64_t from, uint64_t file_size, std::vector<char>& w)
{
	for (uint64_t at = from; at < file_size;) {
		const ssize_t m = pread_retry(fd, w.data(), w.size(), (off_t)at);
		if (m <= 0)
			return file_size;
		if (const char* q = static_cast<const char*>(memchr(w.data(), '\n', (size_t)m)))
			return at + (uint64_t)(q - w.data()) + 1;
		at += (uint64_t)m;
		if (w.size() < (1u << 20))
			w.resize(w.size() * 4); // a long line: fewer, larger reads from here on
	}
	return file_size;
}

uint64_t resync(int fd, uint64_t begin, int fmt, uint64_t file_size)
{
	if (begin == 0)
		return 0;
	if (begin >= file_size)
		return file_size;
	std::vector<char> w(4096); // reads are 4 KiB (a read record) unless a line turns out to be longer
	// 1. first line start >= begin
	uint64_t ls = next_line(fd, begin - 1, file_size, w);
	// 2. advance line by line to a record start
	while (ls < file_size) {
		char c = 0;
		if (pread_retry(fd, &c, 1, (off_t)ls) != 1)
			return file_size;
		if (c != '\n' && c != '\r') {
			if (fmt == btlbf_fastx::PLAIN)
				return ls;
			if (fmt == btlbf_fastx::FASTA && c == '>')
				return ls;
			if (fmt == btlbf_fastx::FASTQ && c == '@') {
				// the first byte of the line two below, however long the two lines in between are
				const uint64_t l2 = next_line(fd, ls, file_size, w);
				const uint64_t l3 = l2 < file_size ? next_line(fd, l2, file_size, w) : file_size;
				char c2 = 0;
				if (l3 < file_size && pread_retry(fd, &c2, 1, (off_t)l3) == 1 && c2 == '+')
					return ls;
			}
		}
		ls = next_line(fd, ls, file_size, w); // next line
	}
	return file_size;
}

int open_impl(btlbf_fastx** out, const char* path, uint32_t flags, uint32_t k, uint64_t batch_bytes, bool ranged,
              int fmt, uint64_t begin, uint64_t end)
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
	if (ranged) {
		r->fd = open(path, O_RDONLY);
		struct stat sb;
		if (r->fd < 0 || fstat(r->fd, &sb) != 0) {
			delete r;
			return btlbf_set_error(BTLBF_EIO, "file \"%s\" could not be read.", path);
		}
		r->fmt = static_cast<btlbf_fastx::Fmt>(fmt);
		r->read_pos = r->buf_off = resync(r->fd, begin, fmt, (uint64_t)sb.st_size);
		r->range_end = end;
		if (r->read_pos >= end && end != UINT64_MAX)
			r->eof = true; // no record starts inside this range
	} else {
		r->gz = gzopen(path, "rb"); // transparent for uncompressed input
		if (!r->gz) {
			delete r;
			return btlbf_set_error(BTLBF_EIO, "file \"%s\" could not be read.", path);
		}
		(void)gzbuffer(r->gz, 1u << 20);
	}
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

} // namespace

extern "C" int btlbf_fastx_open(btlbf_fastx** out, const char* path, uint32_t flags, uint32_t k, uint64_t batch_bytes)
{
	return open_impl(out, path, flags, k, batch_bytes, false, 0, 0, UINT64_MAX);
}

// One of several readers over the same UNCOMPRESSED file: this one delivers the records that start in
// byte range [begin, end) (it reads past `end` to finish its last record).  fmt: 1 FASTA, 2 FASTQ, 3 one
// sequence per line.  The ranges of all readers must tile the file; every record is then delivered once.
extern "C" int btlbf_fastx_open_range(btlbf_fastx** out, const char* path, uint32_t flags, uint32_t k,
                                      uint64_t batch_bytes, int fmt, uint64_t begin, uint64_t end)
{
	if (fmt < 1 || fmt > 3 || begin > end)
		return btlbf_set_error(BTLBF_EINVAL, "fastx_open_range: fmt must be 1..3 and begin <= end");
	return open_impl(out, path, flags, k, batch_bytes, true, fmt, begin, end);
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
			if (r->range_end != UINT64_MAX && r->buf_off + r->in_pos >= r->range_end &&
			    (r->fmt == btlbf_fastx::FASTQ ? r->fq_line == 0 : r->fmt == btlbf_fastx::FASTA ? c == '>' : true)) {
				// a record that starts beyond this reader's byte range belongs to the next reader
				r->eof = true;
				r->in_pos = r->in_len;
				break;
			}
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
// Parser threads (one per byte range of an uncompressed file; one in all for gzip input) fill small
// pinned batches; the calling thread appends them to a large batch in HBM (copy stream) and launches
// the kernels (compute stream) whenever one of its two device batches is full.  Small host batches keep
// the pinned memory modest; large device batches let the filter take its partitioned path.
namespace {

#define HIP_TRY_X(expr)                                                                              \
	do {                                                                                             \
		hipError_t e__ = (expr);                                                                     \
		if (e__ != hipSuccess) {                                                                     \
			rc = btlbf_set_error(BTLBF_EHIP, "%s failed: %s", #expr, hipGetErrorString(e__));        \
			goto done;                                                                               \
		}                                                                                            \
	} while (0)

struct HostBatch {
	int reader;
	char* bases;
	uint64_t* starts;
	uint64_t nb, ns;
};

struct Shared {
	std::mutex m;
	std::condition_variable cv_batch, cv_credit;
	std::deque<HostBatch> q;
	std::vector<int> credits; // host buffers a reader may still fill (two each)
	int active = 0;
	int rc = BTLBF_OK;
	bool abort = false;
	char err[256] = "";
};

void parser_thread(Shared* sh, btlbf_fastx* r, int id)
{
	for (;;) {
		{
			std::unique_lock<std::mutex> lk(sh->m);
			sh->cv_credit.wait(lk, [&] { return sh->credits[id] > 0 || sh->abort; });
			if (sh->abort)
				break;
			--sh->credits[id];
		}
		const char* b;
		const uint64_t* st;
		uint64_t nb, ns;
		const int rc = btlbf_fastx_next(r, &b, &nb, &st, &ns);
		if (rc) {
			std::lock_guard<std::mutex> lk(sh->m);
			if (sh->rc == BTLBF_OK) {
				sh->rc = rc;
				snprintf(sh->err, sizeof sh->err, "%s", btlbf_last_error());
			}
			sh->abort = true;
			break;
		}
		if (ns == 0)
			break;
		{
			std::lock_guard<std::mutex> lk(sh->m);
			sh->q.push_back(HostBatch{id, const_cast<char*>(b), const_cast<uint64_t*>(st), nb, ns});
		}
		sh->cv_batch.notify_one();
	}
	{
		std::lock_guard<std::mutex> lk(sh->m);
		--sh->active;
	}
	sh->cv_batch.notify_one();
	sh->cv_credit.notify_all();
}

unsigned parser_threads()
{
	if (const char* e = getenv("BTLBF_FASTX_THREADS")) {
		const int v = atoi(e);
		if (v >= 1 && v <= 64)
			return (unsigned)v;
	}
	unsigned n = std::thread::hardware_concurrency();
	// honour a cgroup CPU quota (containers): cpu.max = "<quota> <period>"
	if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
		long long q = 0, p = 0;
		if (fscanf(f, "%lld %lld", &q, &p) == 2 && q > 0 && p > 0)
			n = std::min<unsigned>(n, (unsigned)((q + p - 1) / p));
		fclose(f);
	}
	return std::max(1u, std::min(8u, n));
}

int run_fastx(btlbf_filter* f, const char* path, uint32_t flags, uint64_t batch_bytes, bool query,
              btlbf_fastx_stats* stats)
{
	if (!f || !path)
		return btlbf_set_error(BTLBF_EINVAL, "fastx: null argument");
	const double t0 = now_s();
	const uint32_t k = btlbf_kmer_size(f);
	// what kind of input: gzip -> one sequential parser; plain -> one parser per byte range
	uint64_t file_size = 0;
	int fmt = 0;
	bool gz = false;
	{
		const int fd = open(path, O_RDONLY);
		struct stat sb;
		if (fd < 0 || fstat(fd, &sb) != 0) {
			if (fd >= 0)
				close(fd);
			return btlbf_set_error(BTLBF_EIO, "file \"%s\" could not be read.", path);
		}
		file_size = (uint64_t)sb.st_size;
		unsigned char head[4096];
		const ssize_t n = pread(fd, head, sizeof head, 0);
		close(fd);
		gz = n >= 2 && head[0] == 0x1f && head[1] == 0x8b;
		for (ssize_t i = 0; i < n && !fmt; ++i)
			if (head[i] != '\n' && head[i] != '\r')
				fmt = head[i] == '>' ? btlbf_fastx::FASTA : head[i] == '@' ? btlbf_fastx::FASTQ : btlbf_fastx::PLAIN;
	}
	uint64_t mt_min = 32ull << 20;
	if (const char* e = getenv("BTLBF_FASTX_MT_MIN_BYTES"))
		mt_min = strtoull(e, nullptr, 10);
	unsigned T = gz || !fmt || file_size < mt_min ? 1 : parser_threads();
	if (batch_bytes == 0)
		batch_bytes = T > 1 ? (64ull << 20) : (256ull << 20);
	// device batch: large enough for the partitioned path of this filter (probes >= 2 % of its bytes)
	uint64_t dev_cap = std::max<uint64_t>(batch_bytes, std::min<uint64_t>(4ull << 30, std::max<uint64_t>(256ull << 20, btlbf_local_bytes(f) / 32)));
	if (file_size && !gz)
		dev_cap = std::max<uint64_t>(batch_bytes, std::min<uint64_t>(dev_cap, file_size + 64));
	const uint64_t dev_seqs = dev_cap / 16 + 4096;

	std::vector<btlbf_fastx*> readers(T, nullptr);
	std::vector<std::thread> threads;
	Shared sh;
	int rc = BTLBF_OK;
	int prev_dev = 0;
	(void)hipGetDevice(&prev_dev);
	hipStream_t copy_s = nullptr, comp_s = nullptr;
	struct Slot {
		char* d_bases = nullptr;
		uint64_t* d_starts = nullptr;
		uint64_t nb = 0, ns = 0;
		hipEvent_t filled = nullptr, done = nullptr;
		bool used = false;
	} slot[2];
	uint64_t* d_counts = nullptr; // [2 slots][2]
	uint64_t* h_counts = nullptr; // pinned: counts mirror [4] + the closing starts entry of each slot [2]
	std::deque<std::pair<hipEvent_t, int>> pending; // copies in flight: their host buffer returns to its reader afterwards
	std::vector<hipEvent_t> ev_pool;
	btlbf_fastx_stats st;
	memset(&st, 0, sizeof st);
	int cur = 0;
	auto harvest = [&](int s_) {
		st.n_windows += h_counts[s_ * 2 + 0];
		st.n_hits += h_counts[s_ * 2 + 1];
	};
	auto give_back = [&](int reader) {
		{
			std::lock_guard<std::mutex> lk(sh.m);
			++sh.credits[reader];
		}
		sh.cv_credit.notify_all();
	};
	auto reap = [&](bool block) -> hipError_t {
		while (!pending.empty()) {
			hipError_t e = block ? hipEventSynchronize(pending.front().first) : hipEventQuery(pending.front().first);
			if (e == hipErrorNotReady)
				return hipSuccess;
			if (e != hipSuccess)
				return e;
			give_back(pending.front().second);
			ev_pool.push_back(pending.front().first);
			pending.pop_front();
			block = false;
		}
		return hipSuccess;
	};
	auto launch = [&](int s_) -> int {
		Slot& sl = slot[s_];
		if (sl.ns == 0)
			return BTLBF_OK;
		int r2 = BTLBF_OK;
		h_counts[4 + s_] = sl.nb; // the closing entry of starts[]
		if (hipMemcpyAsync(sl.d_starts + sl.ns, h_counts + 4 + s_, 8, hipMemcpyHostToDevice, copy_s) != hipSuccess ||
		    hipEventRecord(sl.filled, copy_s) != hipSuccess || hipStreamWaitEvent(comp_s, sl.filled, 0) != hipSuccess)
			return btlbf_set_error(BTLBF_EHIP, "fastx: stream operation failed");
		++st.n_batches;
		if (query && sl.nb < k) // nothing is tested: the slot's count mirror must not keep the previous batch's numbers
			h_counts[s_ * 2 + 0] = h_counts[s_ * 2 + 1] = 0;
		if (sl.nb >= k) {
			btlbf_layout lay;
			lay.starts = sl.d_starts;
			lay.n_seqs = sl.ns;
			lay.read_len = 0;
			if (query) {
				r2 = btlbf_contains_seqs(f, sl.d_bases, sl.nb, &lay, nullptr, nullptr, d_counts + s_ * 2, BTLBF_DEVICE, comp_s);
				if (r2 == BTLBF_OK && hipMemcpyAsync(h_counts + s_ * 2, d_counts + s_ * 2, 16, hipMemcpyDeviceToHost, comp_s) != hipSuccess)
					r2 = btlbf_set_error(BTLBF_EHIP, "fastx: count copy failed");
			} else {
				r2 = btlbf_insert_seqs(f, sl.d_bases, sl.nb, &lay, 0, BTLBF_ORDER_PARALLEL, BTLBF_DEVICE, comp_s);
			}
		}
		if (r2 == BTLBF_OK && hipEventRecord(sl.done, comp_s) != hipSuccess)
			r2 = btlbf_set_error(BTLBF_EHIP, "fastx: event record failed");
		sl.used = true;
		return r2;
	};

	HIP_TRY_X(hipSetDevice(btlbf_device(f)));
	// readers (pinned buffers need the device to be current)
	for (unsigned t = 0; t < T; ++t) {
		if (T == 1)
			rc = btlbf_fastx_open(&readers[t], path, flags, k, batch_bytes);
		else
			rc = btlbf_fastx_open_range(&readers[t], path, flags, k, batch_bytes, fmt, file_size * t / T,
			                            t + 1 == T ? UINT64_MAX : file_size * (t + 1) / T);
		if (rc)
			goto done;
	}
	HIP_TRY_X(hipStreamCreateWithFlags(&copy_s, hipStreamNonBlocking));
	HIP_TRY_X(hipStreamCreateWithFlags(&comp_s, hipStreamNonBlocking));
	HIP_TRY_X(hipHostMalloc(reinterpret_cast<void**>(&h_counts), 6 * sizeof(uint64_t), hipHostMallocDefault));
	HIP_TRY_X(hipMalloc(reinterpret_cast<void**>(&d_counts), 4 * sizeof(uint64_t)));
	memset(h_counts, 0, 6 * sizeof(uint64_t));
	for (int i = 0; i < 2; ++i) {
		HIP_TRY_X(hipEventCreateWithFlags(&slot[i].filled, hipEventDisableTiming));
		HIP_TRY_X(hipEventCreateWithFlags(&slot[i].done, hipEventDisableTiming));
		HIP_TRY_X(hipMalloc(reinterpret_cast<void**>(&slot[i].d_bases), dev_cap + 64));
		HIP_TRY_X(hipMalloc(reinterpret_cast<void**>(&slot[i].d_starts), (dev_seqs + 2) * sizeof(uint64_t)));
	}
	sh.credits.assign(T, 2);
	sh.active = (int)T;
	for (unsigned t = 0; t < T; ++t)
		threads.emplace_back(parser_thread, &sh, readers[t], (int)t);
	for (;;) {
		HostBatch b;
		{
			std::unique_lock<std::mutex> lk(sh.m);
			while (sh.q.empty() && sh.active > 0 && !sh.abort) {
				if (!pending.empty()) {
					// nothing to do but parsers may be waiting for a buffer: finish the oldest copy
					lk.unlock();
					HIP_TRY_X(reap(true));
					lk.lock();
					continue;
				}
				sh.cv_batch.wait(lk);
			}
			if (sh.abort || (sh.q.empty() && sh.active == 0))
				break;
			b = sh.q.front();
			sh.q.pop_front();
		}
		HIP_TRY_X(reap(false));
		st.n_bases += b.nb;
		Slot* sl = &slot[cur];
		if (sl->nb + b.nb > dev_cap || sl->ns + b.ns > dev_seqs) {
			if ((rc = launch(cur)))
				goto done;
			cur ^= 1;
			sl = &slot[cur];
			if (sl->used) {
				HIP_TRY_X(hipEventSynchronize(sl->done));
				harvest(cur);
				sl->used = false;
			}
			sl->nb = sl->ns = 0;
		}
		// append: this batch's offsets become offsets into the device batch
		for (uint64_t i = 0; i < b.ns; ++i)
			b.starts[i] += sl->nb;
		HIP_TRY_X(hipMemcpyAsync(sl->d_bases + sl->nb, b.bases, b.nb, hipMemcpyHostToDevice, copy_s));
		HIP_TRY_X(hipMemcpyAsync(sl->d_starts + sl->ns, b.starts, b.ns * sizeof(uint64_t), hipMemcpyHostToDevice, copy_s));
		hipEvent_t ev;
		if (ev_pool.empty()) {
			HIP_TRY_X(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
		} else {
			ev = ev_pool.back();
			ev_pool.pop_back();
		}
		HIP_TRY_X(hipEventRecord(ev, copy_s));
		pending.emplace_back(ev, b.reader);
		sl->nb += b.nb;
		sl->ns += b.ns;
	}
	if (sh.rc) {
		rc = btlbf_set_error(sh.rc, "%s", sh.err);
		goto done;
	}
	if ((rc = launch(cur)))
		goto done;
	HIP_TRY_X(hipStreamSynchronize(copy_s));
	HIP_TRY_X(hipStreamSynchronize(comp_s));
	for (int i = 0; i < 2; ++i)
		if (slot[i].used)
			harvest(i);
done:
	{
		std::lock_guard<std::mutex> lk(sh.m);
		sh.abort = sh.abort || rc != BTLBF_OK;
		if (rc != BTLBF_OK)
			sh.abort = true;
	}
	if (rc != BTLBF_OK)
		sh.cv_credit.notify_all();
	else {
		std::lock_guard<std::mutex> lk(sh.m);
		sh.abort = true; // normal end: parsers have all finished already
	}
	sh.cv_credit.notify_all();
	for (auto& th : threads)
		if (th.joinable())
			th.join();
	if (comp_s)
		(void)hipStreamSynchronize(comp_s);
	if (copy_s)
		(void)hipStreamSynchronize(copy_s);
	for (auto& pe : pending)
		(void)hipEventDestroy(pe.first);
	for (auto& e : ev_pool)
		(void)hipEventDestroy(e);
	for (auto* r : readers)
		if (r) {
			st.n_records += r->n_records;
			st.seconds_parse += r->seconds_parse;
			btlbf_fastx_close(r);
		}
	st.seconds_parse /= (double)T; // average per parser thread (they run side by side)
	for (int i = 0; i < 2; ++i) {
		if (slot[i].d_bases)
			(void)hipFree(slot[i].d_bases);
		if (slot[i].d_starts)
			(void)hipFree(slot[i].d_starts);
		if (slot[i].filled)
			(void)hipEventDestroy(slot[i].filled);
		if (slot[i].done)
			(void)hipEventDestroy(slot[i].done);
	}
	if (d_counts)
		(void)hipFree(d_counts);
	if (h_counts)
		(void)hipHostFree(h_counts);
	if (copy_s)
		(void)hipStreamDestroy(copy_s);
	if (comp_s)
		(void)hipStreamDestroy(comp_s);
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

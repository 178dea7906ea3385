#!/usr/bin/env python3
"""bench.py -- the driver's benchmark contract for the k-mer Bloom filter hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], "C2"): per GPU a 2^39-bit (64 GiB) BloomFilter, k=31, h=4, and
100 M synthetic 150 bp reads (SURVEY.md 8d generator, seed 42) resident in HBM before timing starts.
One step = clear the filter (lazily: the first insert batch writes every segment from zero instead of a
separate memset + read sweep), insert all reads, query all reads (every k-mer is a hit, the per-k-mer
contains() bitmask is written).
value = (k-mers inserted + k-mers queried) / wall time, whole job.

N > 1 (weak scaling): the filter is N x 64 GiB, hash-range sharded over the ranks
(btl_bloomfilter_amd/sharded.py).  N = 2..4: the reads are all-gathered (1 byte per base over xGMI), every
shard hashes all of them and applies the probes inside its own bit range; query partials are ANDed at
the reads' owner.  N = 8: every rank hashes its own 100 M reads and routes 4-byte partitioned probe
entries to the owning shard with RCCL all-to-alls; a query returns only the positions found clear.

Large batches take the partitioned pipeline: pass A hash + radix partition of the
probe positions, pass B split into 64 KiB segments, pass C OR / test in LDS.  Every kernel launch is
timed with HIP events on the launch stream inside the library (btlbf_set_profiling).

Extra objects on the JSON line: "roofline" for the kernel with the largest share of the timed region,
"kernels" with the same figures for every kernel, "model_8d" = the whole insert / query operation
priced with SURVEY 8d's random-access byte model; "cpu_baseline" = the genuine reference build
(oracle/_ref, kind "reference") or the C port (oracle/, kind "port") timed on this box's host cores
over a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

K, H, READ_LEN = 31, 4, 150
LOG2_BITS = 39
N_READS = 100_000_000
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic bytes per k-mer, SURVEY.md 8d: query h*64 + L/(L-k+1); insert h*128 + L/(L-k+1)
BYTES_QUERY = H * 64 + READ_LEN / (READ_LEN - K + 1)
BYTES_INSERT = H * 128 + READ_LEN / (READ_LEN - K + 1)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("BTLBF_BENCH_READS", N_READS)),
                    help="reads per GPU (default: the C2 workload, 100 M)")
    ap.add_argument("--log2-bits", type=int, default=int(os.environ.get("BTLBF_BENCH_LOG2_BITS", LOG2_BITS)),
                    help="log2 of filter bits per GPU (default 39 = 64 GiB)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true")
    ap.add_argument("--cpu-reads", type=int, default=2_000_000)
    return ap.parse_args()


def mem_available_bytes():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 0


def host_cores():
    """cores this process may really use: the affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(n_reads, log2_bits, gpu_sample=None):
    """gpu_sample = (digest, popcount) of the HIP library's filter after inserting the SAME n_reads reads into a
    cleared filter of 2^log2_bits bits: compared with the digest of the filter the timed reference run builds
    ("digest_equal": the bodies are identical, bit for bit, at the benchmark's own geometry).

    Reference CPU path (ntHashIterator + BloomFilter insert/contains, OpenMP over reads) on bounded samples
    of the bench workload: all host cores and one thread, plus the C port next to the genuine
    reference on BASELINE config 1's filter (2^33 bits) so that a "port" number can be read as a reference
    number.  Falls back to the C port alone when the reference build is not on this machine."""
    from oracle import pyoracle

    cores = host_cores()
    avail = mem_available_bytes()
    lg = log2_bits
    while lg > 30 and (1 << lg) // 8 > 0.55 * avail:
        lg -= 1
    bits = 1 << lg
    pyoracle.build()
    port = pyoracle.Oracle()
    if pyoracle.Ref.available():
        kind, runner = "reference", pyoracle.Ref()
    else:
        kind, runner = "port", port

    def rate(r):
        return {"Mkmers_s": 2 * r["kmers"] / (r["t_insert"] + r["t_query"]) / 1e6,
                "insert_Mkmers_s": r["kmers"] / r["t_insert"] / 1e6, "query_Mkmers_s": r["kmers"] / r["t_query"] / 1e6,
                "threads": r["threads"], "kmers": r["kmers"], "query_hits": r["hits"]}

    t0 = time.time()
    want_digest = kind == "reference" and gpu_sample is not None and lg == log2_bits
    if want_digest:
        allc = runner.bench_bf(n_reads, READ_LEN, K, H, bits, 42, 42, threads=cores, prefault=1, skip_pop=1, digest=True)
    else:
        allc = runner.bench_bf(n_reads, READ_LEN, K, H, bits, 42, 42, threads=cores, prefault=1)
    n1 = max(n_reads // 16, 50_000)
    one = runner.bench_bf(n1, READ_LEN, K, H, bits, 42, 42, threads=1, prefault=0)  # pages already touched
    wall = time.time() - t0
    out = {
        "value": rate(allc)["Mkmers_s"], "unit": "Mk-mers/s", "cores": allc["threads"], "kind": kind,
        "cpu_model": cpu_model(),
        "insert_Mkmers_s": rate(allc)["insert_Mkmers_s"], "query_Mkmers_s": rate(allc)["query_Mkmers_s"],
        "one_thread": dict(rate(one), reads=n1),
        "sample": "%d synthetic 150 bp reads (seed 42) inserted then queried (all hits), k=31 h=4, 2^%d-bit filter "
                  "(%s), pages pre-touched, OpenMP over reads, all %d cores; then %d reads on one thread; %.0f s wall "
                  "incl. prefault" % (n_reads, lg, "same size as the GPU run" if lg == log2_bits else "scaled to host RAM",
                                      allc["threads"], n1, wall),
        "query_hits": allc["hits"], "kmers": allc["kmers"],
    }
    if want_digest:
        out["digest_equal"] = tuple(allc["digest"]) == tuple(gpu_sample[0])
        out["digest"] = {"reference": ["%016x" % v for v in allc["digest"]], "gpu": ["%016x" % v for v in gpu_sample[0]],
                         "gpu_popcount": gpu_sample[1],
                         "note": "btlbf_digest (include/btlbf.h) of the 2^%d-bit body after inserting the sample's %d "
                                 "reads: the reference's m_filter on the host (oracle/ref_driver.cpp ref_bf_digest) "
                                 "against the array in HBM" % (lg, n_reads)}
    else:
        out["digest_equal"] = None
    # calibration of the port against the genuine reference (same machine, same sample, C1's 2^33-bit filter)
    if kind == "reference":
        try:
            nc = max(n_reads // 4, 100_000)
            r_ref = runner.bench_bf(nc, READ_LEN, K, H, 1 << 33, 42, 42, threads=cores, prefault=1)
            r_port = port.bench_bf(nc, READ_LEN, K, H, 1 << 33, 42, 42, threads=cores, prefault=1)
            out["port_vs_reference_c1"] = {
                "reads": nc, "log2_bits": 33, "threads": r_ref["threads"],
                "reference_Mkmers_s": rate(r_ref)["Mkmers_s"], "port_Mkmers_s": rate(r_port)["Mkmers_s"],
                "ratio_port_over_reference": rate(r_port)["Mkmers_s"] / rate(r_ref)["Mkmers_s"]}
        except Exception as exc:  # never lose the baseline over its calibration
            out["port_vs_reference_c1"] = {"error": repr(exc)}
    return out


C5_SEEDS = ["1110111011101110111011101110111", "1101101101101101011011011011011",
            "1111001111001111111001111001111", "1011101011101011101011101011101"]


def side_configs(m, torch, reads, n_reads, dev, only=None):
    """The other BASELINE configurations and the reference's everyday shapes (ragged sequences as its FASTA loader
    produces them, Tests/AdHoc/ParallelFilter.cpp:104-122; a filter size from calcOptimalSize, BloomFilter.hpp:413-421,
    i.e. no power of two), each run on THIS box after the timed region: one warm-up pass (scratch allocation), then
    `reps` timed insert + query passes over the same 10^8 resident reads.  Not part of `value`."""
    out = {}
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

    def run(name, make, insert, query, kmers, reps=2, note=None, extra=None):
        if only and name not in only:
            return
        try:
            f = make()
            f.setProfiling(True)
            insert(f)
            query(f)
            torch.cuda.synchronize()
            f.getProfile(reset=True)
            ti = tq = 0.0
            cnt = None
            for _ in range(reps):
                f.clear()
                e = [ev() for _ in range(3)]
                e[0].record()
                insert(f)
                e[1].record()
                cnt = query(f)
                e[2].record()
                torch.cuda.synchronize()
                ti += e[0].elapsed_time(e[1]) * 1e-3
                tq += e[1].elapsed_time(e[2]) * 1e-3
            prof = f.getProfile(reset=True)
            c = cnt.tolist()
            out[name] = {"insert_Mkmers_s": kmers * reps / ti / 1e6, "query_Mkmers_s": kmers * reps / tq / 1e6,
                         "insert_ms": ti / reps * 1e3, "query_ms": tq / reps * 1e3, "kmers": c[0], "hits": c[1],
                         "kernel_ms_per_launch": {k_: round(v[0] / v[1], 2) for k_, v in prof.items() if v[1]},
                         "launches_per_pass": {k_: v[1] // reps for k_, v in prof.items() if v[1]}}
            if note:
                out[name]["note"] = note
            if extra:
                out[name].update(extra(f))
            f.releaseScratch()
            del f
        except Exception as exc:  # a side table must never cost the bench line
            out[name] = {"error": repr(exc)}
        torch.cuda.empty_cache()

    L = READ_LEN
    q = lambda f, **kw: f.containsSeqs(reads, want_valid=True, want_counts=True, **kw)[2]  # noqa: E731
    # C1: the reference's own CPU-runnable case
    r1 = reads[: 1_000_000 * L]
    run("C1", lambda: m.BloomFilter(1 << 33, H, K), lambda f: f.insertSeqs(r1, read_len=L),
        lambda f: f.containsSeqs(r1, read_len=L, want_valid=True, want_counts=True)[2], 1_000_000 * (L - K + 1), reps=4,
        note="10^6 reads, 2^33 bits, k=31, h=4", extra=lambda f: {"popcount": f.getPop()})
    # C5: spaced seeds
    def make_c5():
        f = m.BloomFilter(1 << 37, 4, K)
        f.setSpacedSeeds(C5_SEEDS, 1)
        return f
    run("C5", make_c5, lambda f: f.insertSeqs(reads, read_len=L), lambda f: q(f, read_len=L), n_reads * (L - K + 1),
        note="10^8 reads, 2^37 bits, k=31, 4 spaced seeds x h2=1 (stHashIterator)")
    # a filter of no power-of-two size
    run("C2_3x2p37_bits", lambda: m.BloomFilter(3 << 37, H, K), lambda f: f.insertSeqs(reads, read_len=L),
        lambda f: q(f, read_len=L), n_reads * (L - K + 1), note="C2's reads, 3*2^37 bits (48 GiB): hash % size by multiplication")
    # ragged sequences (lengths 100..200) over the same bases
    try:
        if only and "C2_ragged" not in only:
            raise StopIteration
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        lens = torch.randint(100, 201, (n_reads + n_reads // 8,), device=dev, generator=g, dtype=torch.int64)
        st = torch.cumsum(lens, 0)
        st = st[st < n_reads * L]
        starts = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), st,
                            torch.tensor([n_reads * L], dtype=torch.int64, device=dev)])
        n_seq = starts.numel() - 1
        kmers_r = int((torch.clamp(starts[1:] - starts[:-1] - (K - 1), min=0)).sum().item())
        del lens, st
        run("C2_ragged", lambda: m.BloomFilter(1 << LOG2_BITS, H, K), lambda f: f.insertSeqs(reads, starts=starts),
            lambda f: q(f, starts=starts), kmers_r,
            note="C2's bases cut into %d sequences of 100..200 bases (btlbf_layout::starts), 2^39 bits" % n_seq)
        del starts
    except StopIteration:
        pass
    except Exception as exc:
        out["C2_ragged"] = {"error": repr(exc)}
    # C3: counting filter; incrementAll + contains through the partitioned pipeline, then the reference's default
    # insert (incrementMin, CountingBloomFilter.hpp:135-162,198-204) on the direct kernel
    kc, hc = 25, 3
    km3 = n_reads * (L - kc + 1)
    run("C3_incrementAll", lambda: m.CountingBloomFilter(1 << 35, hc, kc, 2),
        lambda f: (f.insertSeqs(reads, read_len=L, increment_all=True), f.insertSeqs(reads, read_len=L, increment_all=True)),
        lambda f: q(f, read_len=L), 2 * km3,
        note="10^8 reads inserted twice (incrementAll), 2^35 uint8 counters, k=25, h=3, threshold 2; insert rate counts "
             "both insertions, query_Mkmers_s is per %d k-mers" % km3)
    if "query_Mkmers_s" in out.get("C3_incrementAll", {}):
        out["C3_incrementAll"]["query_Mkmers_s"] /= 2  # one query pass per timed rep, over km3 k-mers
    try:
        if only and "C3_insert" not in only:
            raise StopIteration
        cb = m.CountingBloomFilter(1 << 35, hc, kc, 2)
        t = []
        for _ in range(2):
            e0, e1 = ev(), ev()
            e0.record()
            cb.insertSeqs(reads, read_len=L)
            e1.record()
            torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1) * 1e-3)
        _, _, c = cb.containsSeqs(reads, read_len=L, want_valid=False, want_counts=True)
        out["C3_insert"] = {"insert_Mkmers_s": 2 * km3 / sum(t) / 1e6, "first_pass_Mkmers_s": km3 / t[0] / 1e6,
                            "second_pass_Mkmers_s": km3 / t[1] / 1e6, "kmers": c.tolist()[0], "hits_threshold_2": c.tolist()[1],
                            "note": "CountingBloomFilter::insert = conservative update (incrementMin), parallel mode: "
                                    "every read set inserted twice; all k-mers must pass threshold 2 afterwards"}
        del cb
    except StopIteration:
        pass
    except Exception as exc:
        out["C3_insert"] = {"error": repr(exc)}
    torch.cuda.empty_cache()
    return out


# algorithmic HBM bytes one launch of each kernel moves: per k-mer figures with
# h = 4, L = 150, k = 31, plus the per-launch sweep of the filter array for pass C
def kernel_bytes(slot, kmers_per_launch, filter_bytes, sweep_frac=1.0, fresh_frac=0.0):
    seq = READ_LEN / (READ_LEN - K + 1)  # bytes of read buffer per k-mer
    per_kmer = {
        "insert_direct": BYTES_INSERT,        # h*128 + seq
        "query_direct": BYTES_QUERY,          # h*64 + seq
        "insert_hash": seq + 4 * H,           # read bases, write one 4-byte entry per probe
        "query_hash": seq + 4 * H + 0.125 * seq,  # + the valid/hit bitmaps
        "insert_split": 8 * H,                # read + write every entry
        "query_split": 8 * H,
        "insert_apply": 4 * H,                # read every entry (+ sweep below)
        "query_test": 4 * H,
        "query_resolve": 0.0,
    }.get(slot, 0.0)
    # pass C reads (and, inserting, writes back) each filter segment once per batch; a batch is applied
    # in several launches (groups of segments), each sweeping sweep_frac of the filter.  The first batch after
    # a clear builds its segments from zero in LDS and only WRITES them (btlbf_clear is lazy):
    # fresh_frac of the insert launches move the array once instead of twice
    sweep = {"insert_apply": (2.0 - fresh_frac) * filter_bytes, "query_test": 1.0 * filter_bytes}.get(slot, 0.0)
    return per_kmer * kmers_per_launch + sweep * sweep_frac


KERNEL_NAMES = {
    "insert_direct": "seq_kernel<OP_BF_INSERT> (fused ntHash + atomicOr)",
    "query_direct": "seq_kernel<OP_BF_CONTAINS> (fused ntHash + gather)",
    "insert_hash": "part_hash_ov_kernel (pass A: fused ntHash + radix partition, overlapped schedule, insert)",
    "query_hash": "part_hash_ov_kernel<QUERY> (pass A: fused ntHash + radix partition, overlapped schedule, query)",
    "insert_split": "part_split_kernel (pass B, insert)",
    "query_split": "part_split_kernel<QUERY> (pass B, query)",
    "insert_apply": "part_apply_kernel (pass C: OR entries into the segment in LDS)",
    "query_test": "part_apply_kernel<QUERY> (pass C: test entries against the segment in LDS)",
    "query_resolve": "failed-position set + resolve pass",
}


def load_traffic():
    """per-launch HBM bytes from the rocprofv3 PMC runs committed under profiles/ (or None)"""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p))
        except ValueError:
            return None
    return None


def traffic_for(traffic, slot, kmers_per_launch):
    """measured HBM bytes per launch, only when the PMC run used the same launch size as this run"""
    t = traffic.get(slot)
    if not t or abs(t.get("kmers_per_launch", 0) - kmers_per_launch) > 0.01 * kmers_per_launch:
        return None
    return t.get("bytes_per_launch")


def launcher_command(n_gpus, argv, port=None):
    """the command the driver itself uses for N > 1: one rank per GPU on this node, rendezvous on 127.0.0.1"""
    port = port or int(os.environ.get("MASTER_PORT", 29500 + os.getpid() % 2000))
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # plain `python bench.py --gpus N`: start one rank per GPU under torch.distributed.run as a CHILD
            # process (nothing has touched the GPU yet in this one), relay its output and leave with its code
            cmd = launcher_command(args.gpus, sys.argv[1:])
            if os.environ.get("BTLBF_BENCH_LAUNCH_DRYRUN"):  # tests: show the command instead of running it
                print(json.dumps(cmd))
                sys.exit(0)
            sys.exit(subprocess.call(cmd))
        args.gpus = world

    import torch

    import btl_bloomfilter_amd as m
    from btl_bloomfilter_amd import _lib

    # rehearsal mode (tests only): all ranks on cuda:0 with the gloo backend, to run the N > 1 code path
    # of this file on a one-GPU box; RCCL itself needs one GPU per rank
    rehearsal = bool(os.environ.get("BTLBF_BENCH_REHEARSAL"))
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib = _lib.load()
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    elif os.environ.get("BTLBF_FORCE_EXCHANGE") and os.environ.get("BTLBF_BENCH_FORCE_SHARDED"):
        # one GPU, but with the exchange of the multi-GPU path really performed: a one-rank RCCL group
        # sends every block to itself (diagnostic: the whole N > 1 code path short of xGMI)
        import torch.distributed as dist1

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist1.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    # N = 8 runs BASELINE config 4 itself: a 2^43-bit (1 TiB) filter hash-sharded over the 8 GPUs (128 GiB
    # shards, two position windows on the routed path) and 10^9 reads = 1.25e8 per GPU -- unless the caller
    # set sizes explicitly
    c4 = world == 8 and args.reads == N_READS and args.log2_bits == LOG2_BITS
    if c4:
        args.reads, args.log2_bits = 125_000_000, 40
    n_reads = args.reads
    bits_per_gpu = 1 << args.log2_bits
    kmers = n_reads * (READ_LEN - K + 1)
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)

    # ---- resident inputs: synthetic reads of this rank (rank r owns reads [r*n, (r+1)*n)) ----
    reads = m.synth_reads_device(42, rank * n_reads, n_reads, READ_LEN, device=local_rank)
    lay = _lib.Layout()
    lay.starts, lay.n_seqs, lay.read_len = None, 0, READ_LEN
    n_bytes = reads.numel()
    hit_bits = torch.zeros((n_bytes + 63) // 64, dtype=torch.int64, device=dev)
    counts = torch.zeros(2, dtype=torch.int64, device=dev)
    # second query set (reads nobody inserted: ~all misses), reported next to the headline, not part of it
    reads_miss = None
    if world == 1 and not os.environ.get("BTLBF_BENCH_FORCE_SHARDED") and not os.environ.get("BTLBF_BENCH_NO_MISS"):
        reads_miss = m.synth_reads_device(43, 0, n_reads, READ_LEN, device=local_rank)

    force_sharded = bool(os.environ.get("BTLBF_BENCH_FORCE_SHARDED"))  # exercise the multi-GPU code path on one GPU
    single = world == 1 and not force_sharded
    if single:
        flt = m.BloomFilter(bits_per_gpu, H, K, device=local_rank)
        flt.setProfiling(True)

        def do_insert():
            _lib.check(lib.btlbf_insert_seqs(flt._h, C.c_void_p(reads.data_ptr()), n_bytes, C.byref(lay), 0, 0,
                                             _lib.DEVICE, sp))

        def do_query():
            _lib.check(lib.btlbf_contains_seqs(flt._h, C.c_void_p(reads.data_ptr()), n_bytes, C.byref(lay),
                                               C.c_void_p(hit_bits.data_ptr()), None,
                                               C.c_void_p(counts.data_ptr()), _lib.DEVICE, sp))

        def do_clear():
            _lib.check(lib.btlbf_clear(flt._h, sp))
    else:
        from btl_bloomfilter_amd.sharded import ShardedBloomFilter

        # 8 ranks: the routed path, and never a silent fall-back to the direct position exchange
        flt = ShardedBloomFilter(bits_per_gpu * world, H, K, device=local_rank,
                                 mode=os.environ.get("BTLBF_SHARD_MODE") or ("routed" if world >= 8 else None))

        def do_insert():
            flt.insert_reads(reads, READ_LEN)

        def do_query():
            flt.contains_reads(reads, READ_LEN, hit_bits, counts)

        def do_clear():
            flt.clear()

    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    t_ins, t_qry = [], []

    def step(timed):
        e = [ev() for _ in range(4)]
        do_clear()
        e[0].record(stream)
        do_insert()
        e[1].record(stream)
        e[2].record(stream)
        do_query()
        e[3].record(stream)
        return e if timed else None

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    if single:
        flt.getProfile(reset=True)  # drop the warm-up launches
    t0 = time.perf_counter()
    events = [step(True) for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    for e in events:
        t_ins.append(e[0].elapsed_time(e[1]) * 1e-3)
        t_qry.append(e[2].elapsed_time(e[3]) * 1e-3)
    if dist is not None:
        rdev = "cpu" if rehearsal else dev  # gloo reduces host tensors
        t = torch.tensor([elapsed], device=rdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = counts.to(rdev)
        dist.all_reduce(c)
        tot_counts = c.tolist()
    else:
        tot_counts = counts.tolist()

    if rank == 0:
        total_kmers = kmers * world
        assert tot_counts[0] == total_kmers, "query saw %d clean k-mers, expected %d" % (tot_counts[0], total_kmers)
        assert tot_counts[1] == total_kmers, "false negatives: %d hits of %d" % (tot_counts[1], total_kmers)
        value = 2 * total_kmers * args.steps / elapsed / 1e6
        ins = sum(t_ins) / len(t_ins)
        qry = sum(t_qry) / len(t_qry)
        traffic = load_traffic() or {}
        out = {
            "metric": "Mk-mers/s insert+query, k=31 h=4, 64 GiB filter",
            "value": value, "unit": "Mk-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("C4: 2^43-bit (1 TiB) BloomFilter hash-sharded over 8 GPUs (2^40-bit shards), k=31, "
                                    "h=4, 10^9 synthetic 150 bp reads (%d per GPU) inserted then queried (all hits), "
                                    "reads resident in HBM" % n_reads) if c4 else
                                   ("C2: per GPU 2^%d-bit BloomFilter, k=31, h=4, %d synthetic 150 bp reads "
                                    "inserted then queried (all hits), reads resident in HBM"
                                    % (args.log2_bits, n_reads)),
                       "collective_world_size": (dist.get_world_size() if dist is not None else 1),
                       "collective_backend": (dist.get_backend() if dist is not None else "none"),
                       "filter_bits_total": bits_per_gpu * world, "kmers_per_pass": total_kmers,
                       "parallelism": "1 GPU" if single else
                                      ("hash-range shards x%d, reads all-gathered over RCCL, every shard hashes all "
                                       "reads and keeps its window's probes" % world if flt.mode == "gather" else
                                       "hash-range shards x%d, partitioned routing, RCCL all-to-all of 4-byte entries"
                                       % world)},
            "insert_Mkmers_s": total_kmers / ins / 1e6, "query_Mkmers_s": total_kmers / qry / 1e6,
        }
        if single:
            prof = flt.getProfile(reset=True)
            filter_bytes = bits_per_gpu // 8
            kernels = {}
            for slot, (ms, calls) in prof.items():
                if slot == "other" or calls == 0:
                    continue
                per_launch = kmers * args.steps / calls  # k-mers one launch processes
                batches = prof.get(slot.split("_")[0] + "_hash", (0, 0))[1]  # one pass-A launch per batch
                # every timed step clears the filter first: one batch per step is a fresh one
                fresh = args.steps / batches if batches else 0.0
                nbytes = kernel_bytes(slot, per_launch, filter_bytes, batches / calls if batches else 1.0, fresh)
                avg_s = ms / calls * 1e-3
                kernels[slot] = {"kernel": KERNEL_NAMES.get(slot, slot), "launches": calls, "avg_launch_ms": ms / calls,
                                 "share_of_timed_region": ms * 1e-3 / elapsed, "kmers_per_launch": per_launch,
                                 "bytes_per_launch": nbytes, "achieved": nbytes / avg_s / 1e9 if avg_s else None,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": nbytes / avg_s / 1e9 / HBM_PEAK_GBS if avg_s else None,
                                 "traffic": traffic_for(traffic, slot, per_launch)}
            dom = max(kernels, key=lambda s_: kernels[s_]["share_of_timed_region"])
            d = kernels[dom]
            out["roofline"] = {"kernel": d["kernel"], "bound": "hbm", "achieved": d["achieved"], "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": d["frac"], "traffic": d["traffic"],
                               "traffic_source": (traffic.get("source", "profiles/traffic.json") + " -- read from the "
                                                  "committed PMC summary, NOT measured in this run")
                               if d["traffic"] is not None else None,
                               "bytes_per_launch": d["bytes_per_launch"], "kmers_per_launch": d["kmers_per_launch"],
                               "launch_ms": d["avg_launch_ms"], "share_of_timed_region": d["share_of_timed_region"],
                               "note": "dominant kernel by HIP-event time; its algorithmic bytes are the streamed "
                                       "bytes of DESIGN.md section 4, not SURVEY 8d's random-sector model"}
            out["kernels"] = kernels
            a_ins = kmers * BYTES_INSERT / ins / 1e9
            a_qry = kmers * BYTES_QUERY / qry / 1e9
            out["model_8d"] = {
                "note": "whole operations priced with SURVEY 8d's random-access model (one 64-B sector per probe "
                        "read, two per probe insert); > 1 means the partitioned pipeline moved fewer bytes than "
                        "that model assumes, not that HBM ran above its peak",
                "insert": {"bytes_per_kmer": BYTES_INSERT, "ms": ins * 1e3, "achieved": a_ins, "frac": a_ins / HBM_PEAK_GBS},
                "query": {"bytes_per_kmer": BYTES_QUERY, "ms": qry * 1e3, "achieved": a_qry, "frac": a_qry / HBM_PEAK_GBS}}
            if reads_miss is not None:
                e0, e1 = ev(), ev()
                e0.record(stream)
                _lib.check(lib.btlbf_contains_seqs(flt._h, C.c_void_p(reads_miss.data_ptr()), n_bytes, C.byref(lay),
                                                   C.c_void_p(hit_bits.data_ptr()), None,
                                                   C.c_void_p(counts.data_ptr()), _lib.DEVICE, sp))
                e1.record(stream)
                torch.cuda.synchronize()
                out["query_all_miss"] = {"Mkmers_s": kmers / (e0.elapsed_time(e1) * 1e-3) / 1e6,
                                         "false_positives": int(counts[1].item()), "kmers": int(counts[0].item()),
                                         "note": "seed-43 reads against the filter of the timed run; AUTO samples the "
                                                 "batch, sees misses and keeps the direct early-exit gather kernel"}
                del reads_miss
                # mixed hit rates (reads foreign to the filter spliced in at regular intervals): AUTO samples
                # every read and splits the buffer; reported next to the headline, not in it
                mixed = {}
                for label, period in (("50", 2), ("90", 10), ("99", 100), ("99.9", 1000)):
                    try:
                        q = m.synth_reads_device(43, 0, n_reads, READ_LEN, device=local_rank)
                        foreign = (torch.arange(n_reads, device=dev) % period == 0).view(-1, 1)
                        torch.where(foreign, q.view(n_reads, READ_LEN), reads.view(n_reads, READ_LEN),
                                    out=q.view(n_reads, READ_LEN))
                        n_foreign = int(foreign.sum().item())
                        del foreign
                        for rep in range(2):  # the first call sizes the cached buffers of the split path
                            flt.getProfile(reset=True)
                            torch.cuda.synchronize()
                            e0, e1 = ev(), ev()
                            e0.record(stream)
                            _lib.check(lib.btlbf_contains_seqs(flt._h, C.c_void_p(q.data_ptr()), n_bytes, C.byref(lay),
                                                               C.c_void_p(hit_bits.data_ptr()), None,
                                                               C.c_void_p(counts.data_ptr()), _lib.DEVICE, sp))
                            e1.record(stream)
                            torch.cuda.synchronize()
                        prof_q = flt.getProfile(reset=True)
                        mixed[label] = {"Mkmers_s": kmers / (e0.elapsed_time(e1) * 1e-3) / 1e6,
                                        "ms": e0.elapsed_time(e1), "foreign_reads": n_foreign,
                                        "kmers": int(counts[0].item()), "hits": int(counts[1].item()),
                                        "kernel_ms": {k_: round(v[0], 2) for k_, v in prof_q.items()}}
                        del q
                    except Exception as exc:  # a side table must never cost the bench line
                        mixed[label] = {"error": repr(exc)}
                out["query_mixed_hit_rates"] = dict(mixed, note="percent of reads that were inserted; the rest are "
                                                    "seed-43 reads; per-window bitmask written as in the headline")
            # the measured random-access ceilings of this GPU on the same array (bare kernels:
            # independent 4-byte loads / 4-byte atomicOr at uniformly random 64-byte-aligned offsets)
            try:
                ng, tg = flt.microbench(0, 1 << 31)
                na, ta = flt.microbench(1, 1 << 30)
                out["random_access_ceiling"] = {
                    "gather_Gprobes_s": ng / tg / 1e9, "gather_GB_s_of_64B_sectors": ng / tg * 64 / 1e9,
                    "atomic_or_Gprobes_s": na / ta / 1e9,
                    "note": "what one probe per random 64-B sector can reach here; the direct kernels sit on these "
                            "ceilings, the partitioned pipeline exists to get off them (DESIGN.md section 5)"}
            except Exception as exc:
                out["random_access_ceiling"] = {"error": repr(exc)}
            # the filter the CPU baseline's sample builds, built here by the HIP library: digest for "digest_equal"
            gpu_sample = None
            if not args.no_cpu_baseline and args.cpu_reads <= n_reads:
                try:
                    do_clear()
                    sub = reads[: args.cpu_reads * READ_LEN]
                    _lib.check(lib.btlbf_insert_seqs(flt._h, C.c_void_p(sub.data_ptr()), sub.numel(), C.byref(lay), 0, 0,
                                                     _lib.DEVICE, sp))
                    torch.cuda.synchronize()
                    gpu_sample = (flt.digest(), flt.getPop())
                except Exception as exc:
                    out["gpu_sample_error"] = repr(exc)
            del hit_bits
            flt.releaseScratch()
            del flt
            torch.cuda.empty_cache()
            if not args.no_side_configs and n_reads == N_READS:
                try:
                    out["side_configs"] = side_configs(m, torch, reads, n_reads, dev)
                except Exception as exc:
                    out["side_configs"] = {"error": repr(exc)}
            if not args.no_cpu_baseline:
                del reads
                try:
                    out["cpu_baseline"] = cpu_baseline(args.cpu_reads, args.log2_bits, gpu_sample)
                except Exception as exc:  # the baseline is a report, never a reason to lose the bench line
                    out["cpu_baseline"] = {"error": repr(exc)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The reference's own host loops through the drop-in shims, next to the genuine reference on the same host cores
(VERDICT r02 item 5).  Not a pytest: a measurement script for the GPU box.

    python tests/loop_bench.py [n_reads=1000000] [log2_bits=33] > gpurun_out/r03/loop_bench.json

Both programs are tests/cpp/loop_bench.cpp: oracle/_ref/loop_bench_ref is built from the reference's headers where
they lie (oracle/Makefile, this container only; the binary travels to the GPU box), tests/cpp/loop_bench_shim from
include/btlbf/*.hpp over libbtlbf.so (__graft_entry__.build_shim_test).  C1 size by default: 10^6 synthetic 150 bp
reads, k = 31, h = 4, 2^33-bit filter; golden popcount 466 832 676 (tests/golden/digests.json: bf_config1)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "loop_bench_ref")
SHIM = os.path.join(ROOT, "tests", "cpp", "loop_bench_shim")


def run(exe, mode, n_reads, log2_bits, threads, n_query=None):
    cmd = [exe, mode, str(n_reads), str(log2_bits), str(threads)] + ([str(n_query)] if n_query else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    if r.returncode != 0:
        return {"error": (r.stdout + r.stderr)[-400:], "mode": mode, "threads": threads}
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    log2_bits = int(sys.argv[2]) if len(sys.argv) > 2 else 33
    cores = len(os.sched_getaffinity(0))
    try:  # the cgroup CPU quota, as bench.py's cpu_baseline counts cores
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    if not os.path.exists(SHIM):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g

        g.build_shim_test()
    out = {"n_reads": n_reads, "log2_bits": log2_bits, "cores": cores, "runs": []}
    for threads in sorted({1, cores}):
        if os.path.exists(REF):
            for mode in ("kmer", "seq"):
                # one thread: a tenth of the reads is enough for a rate (and keeps the script short)
                out["runs"].append(run(REF, mode, n_reads if threads > 1 else max(n_reads // 10, 1000), log2_bits, threads))
                print(json.dumps(out["runs"][-1]), file=sys.stderr, flush=True)
        for mode in ("kmer", "seq", "batch"):
            n = n_reads if (threads > 1 or mode != "kmer") else max(n_reads // 10, 1000)
            # contains(*itr) per k-mer (one GPU round trip per read since the look-ahead of BloomFilter::contains)
            # and countSeq per read through the shims: a sample of the reads is enough for a rate
            nq = {"kmer": 20000 * threads, "seq": 20000 * threads}.get(mode)
            out["runs"].append(run(SHIM, mode, n, log2_bits, threads, nq))
            print(json.dumps(out["runs"][-1]), file=sys.stderr, flush=True)
    full = [r for r in out["runs"] if r.get("reads") == n_reads]
    pops = {r["pop"] for r in full}
    out["all_full_runs_same_popcount"] = len(pops) == 1
    out["popcount"] = sorted(pops)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes of bench.py into profiles/<round>/ and profiles/traffic.json.

    python tools/pmc_summary.py --round r01 --kmers 12000000000 \
        --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write [--tag partitioned]

Each of --fetch / --write is the -d directory of ONE `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE
--kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline` run (counters are collected
in separate passes, MI355X_MICROARCH.md HBM section).  Per kernel the launches of the insert and of the
query pass are averaged; bytes per launch = 2 x FETCH_SIZE KiB (gfx950 counts 64 B per read request and
under-counts the 16-B-per-lane coalesced streams these kernels read by half) + WRITE_SIZE KiB.
"""
import argparse
import collections
import csv
import glob
import json
import os
import sqlite3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel-name fragments -> bench.py profile slots (the QUERY template argument separates the passes)
def slot_of(name):
    if "part_hash_kernel" in name or "part_hash_ov_kernel" in name:  # plain / overlapped schedule of pass A
        return "query_hash" if is_query(name) else "insert_hash"
    if "part_split_kernel" in name:
        return "query_split" if is_query(name) else "insert_split"
    if "part_apply_kernel" in name:
        return "query_test" if is_query(name) else "insert_apply"
    if "seq_kernel" in name:
        return "seq_kernel"
    return None


def is_query(name):
    # QUERY is the 4th template argument of part_hash[_ov]_kernel<H, POW2, SPACED, QUERY, WINDOW[, SMALL]> and the first
    # of part_split_kernel<QUERY> / part_apply_kernel<QUERY, NT>
    args = name[name.find("<") + 1:name.find(">(")] if "<" in name else ""
    parts = [a.strip() for a in args.split(",")]
    q = parts[3] if ("part_hash_kernel" in name or "part_hash_ov_kernel" in name) and len(parts) > 3 else parts[0]
    return q in ("true", "(bool)1", "1")


def collect(d, counter):
    per = collections.defaultdict(list)
    # rocprofv3's default output is a rocpd SQLite database; --output-format csv gives the CSV below
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        con = sqlite3.connect(f)
        for name, value in con.execute("select kernel_name, value from counters_collection where counter_name = ? "
                                       "order by dispatch_id", (counter,)):
            s = slot_of(name)
            if s:
                per[s].append(float(value))
        con.close()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                s = slot_of(row["Kernel_Name"])
                if s:
                    per[s].append(float(row["Counter_Value"]))
    return per


def export_stats(d, out):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        con = sqlite3.connect(f)
        rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        mm = dict((n, (lo, hi)) for n, lo, hi in
                  con.execute("select name, min(duration), max(duration) from kernels group by name"))
        con.close()
        with open(out, "w", newline="") as fh:
            w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for n, calls, tot, avg, pct in rows:
                lo, hi = mm.get(n, (None, None))
                w.writerow([n, calls, int(tot * 1e3) if tot < 1e9 else int(tot), round(avg * 1e3), round(pct, 4), lo, hi])
        return
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as src, open(out, "w") as dst:
            dst.write(src.read())
        return


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r01")
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--kmers", type=float, required=True, help="k-mers one insert (or query) pass processes")
    ap.add_argument("--tag", default="partitioned")
    ap.add_argument("--stats", help="-d directory of a `rocprofv3 --kernel-trace --stats` run: its per-kernel "
                                    "summary is written to profiles/<round>/bench_<tag>_kernel_stats.csv")
    a = ap.parse_args()
    if a.stats:
        export_stats(a.stats, os.path.join(ROOT, "profiles", a.round, "bench_%s_kernel_stats.csv" % a.tag))
    fetch = collect(a.fetch, "FETCH_SIZE")
    write = collect(a.write, "WRITE_SIZE")
    out_dir = os.path.join(ROOT, "profiles", a.round)
    os.makedirs(out_dir, exist_ok=True)
    traffic = {
        "source": "profiles/%s/pmc_%s_summary.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of "
                  "`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline`)" % (a.round, a.tag),
        "note": "bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): gfx950's FETCH_SIZE counts 64 B per "
                "read request and under-counts wide coalesced 16-B-per-lane streams by half "
                "(MI355X_MICROARCH.md HBM section); every read of the partition kernels is such a stream. "
                "WRITE_SIZE is exact for 16-B-per-lane stores.",
    }
    rows = []
    for slot in sorted(set(fetch) | set(write)):
        f, w = fetch.get(slot, []), write.get(slot, [])
        n = max(len(f), len(w))
        fk = sum(f) / len(f) if f else 0.0
        wk = sum(w) / len(w) if w else 0.0
        rows.append((slot, "FETCH_SIZE", len(f), fk))
        rows.append((slot, "WRITE_SIZE", len(w), wk))
        if slot == "seq_kernel":
            continue
        traffic[slot] = {"launches": n, "kmers_per_launch": a.kmers / n if n else None,
                         "fetch_KiB_raw": fk, "write_KiB": wk, "bytes_per_launch": (2 * fk + wk) * 1024}
    with open(os.path.join(out_dir, "pmc_%s_summary.csv" % a.tag), "w") as fh:
        fh.write("# rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace -- python3 bench.py --steps 1 --warmup 0 "
                 "--no-cpu-baseline\n# values per launch, KiB, averaged over the launches of one pass\n")
        fh.write("kernel,counter,launches,KiB_per_launch\n")
        for r in rows:
            fh.write("%s,%s,%d,%.0f\n" % r)
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)
        fh.write("\n")
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel averages of the SQ counters in rocprofv3 --pmc outputs (CSV or rocpd database).

    python tools/sq_counters.py <dir> [<dir> ...]
"""
import collections
import csv
import glob
import os
import sqlite3
import sys


def rows(d):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        con = sqlite3.connect(f)
        for n, c, v in con.execute("select kernel_name, counter_name, value from counters_collection"):
            yield n, c, float(v)
        con.close()


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[1:]:
        for n, c, v in rows(d):
            if "btlbf::" in n:
                acc[n.split("(")[0][:90]][c].append(v)
    for n in sorted(acc):
        print(n)
        for c in sorted(acc[n]):
            v = acc[n][c]
            print("    %-28s %16.0f  (%d launches)" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Print the headline numbers and the per-kernel table of a bench.py JSON line (file argument or stdin)."""
import json
import sys


def main():
    src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
    lines = [l for l in src.read().strip().splitlines() if l.startswith("{")]
    if not lines:
        sys.exit("no JSON line found")
    d = json.loads(lines[-1])
    print("value %.0f %s  ms/step %.1f  insert %.0f  query %.0f" % (
        d["value"], d["unit"], d["ms_per_step"], d.get("insert_Mkmers_s", 0), d.get("query_Mkmers_s", 0)))
    for k, v in d.get("kernels", {}).items():
        print("  %-14s launches %3d  avg %8.2f ms  frac %.3f  share %.3f  traffic/alg %s" % (
            k, v["launches"], v["avg_launch_ms"], v["frac"] or 0, v["share_of_timed_region"],
            "%.3f" % (v["traffic"] / v["bytes_per_launch"]) if v.get("traffic") else "-"))
    if "query_all_miss" in d:
        print("  all-miss query %.0f Mk-mers/s" % d["query_all_miss"]["Mkmers_s"])
    if "cpu_baseline" in d:
        print("  cpu baseline %.1f %s on %d cores (%s)" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"],
                                                         d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"]))


if __name__ == "__main__":
    main()

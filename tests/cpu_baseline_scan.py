"""Thread scaling of the genuine reference build on this host (checker-side diagnostic for bench.py's
cpu_baseline; lives under tests/ because it runs oracle/_ref)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
from oracle.pyoracle import Ref
r = Ref()
print("affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
for t in (8, 16, 32, 64):
    t0=time.time()
    x = r.bench_bf(1000000, 150, 31, 4, 1<<36, 42, 42, threads=t, prefault=1)
    print(t, "threads: insert %.1f query %.1f Mk-mers/s, wall %.1f" % (x["kmers"]/x["t_insert"]/1e6, x["kmers"]/x["t_query"]/1e6, time.time()-t0), flush=True)

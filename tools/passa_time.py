"""diagnostic: HIP-event time of pass A (part_hash_kernel) for one library build.

    BTLBF_LIB=btl_bloomfilter_amd/libbtlbf_<tag>.so [BTLBF_PART_GEOM=large|small] python tools/passa_time.py [reads]

BTLBF_PART_OVERLAP=0 times the plain schedule instead of the overlapped one.  A diagnostic build
(BTLBF_BUILD_TAG=nobar BTLBF_CXXFLAGS=-DBTLBF_EXP_NOBARRIER) leaves out the round barriers of the plain schedule; its
filters are wrong on purpose, only the time of pass A means anything (profiles/r03/passA_overlap.md)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from btl_bloomfilter_amd import _lib

path = os.environ.get("BTLBF_LIB")
if path:
    _lib.LIB_PATH = os.path.abspath(path)
    _lib.load(_lib.LIB_PATH)
import torch

import btl_bloomfilter_amd as m

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
L = int(os.environ.get("READ_LEN", 150))
f = m.BloomFilter(1 << 39, 4, 31)
f.setInsertMode("partitioned")
f.setProfiling(True)
reads = m.synth_reads_device(42, 0, n, L)
f.setQueryMode("partitioned")
for rep in range(3):
    f.insertSeqs(reads, read_len=L)
    f.containsSeqs(reads, read_len=L, want_valid=False)
    torch.cuda.synchronize()
    prof = f.getProfile()
ms, calls = prof["insert_hash"]
print("   per launch ms: " + "  ".join("%s %.3f" % (k, v[0] / v[1]) for k, v in prof.items() if v[1]))
print("%-40s geom=%-6s pass A %.2f ms per launch (%d launches, %.2f ms per 1e9 k-mers)" % (
    os.path.basename(path or "libbtlbf.so"), os.environ.get("BTLBF_PART_GEOM", "auto"), ms / calls, calls,
    ms / (n * (L - 30) / 1e9)))

"""diagnostic: where an X wave and a Y wave of pass A's overlapped schedule spend their cycles (library built with
BTLBF_BUILD_TAG=stamps BTLBF_CXXFLAGS=-DBTLBF_PHASE_STAMPS; tools/stamps_ov.py [reads])"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from btl_bloomfilter_amd import _lib
_lib.LIB_PATH = os.path.join(os.getcwd(), "btl_bloomfilter_amd", "libbtlbf_stamps.so")
lib = _lib.load(_lib.LIB_PATH)
import torch
import btl_bloomfilter_amd as m
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
f = m.BloomFilter(1 << 39, 4, 31); f.setInsertMode("partitioned")
reads = m.synth_reads_device(42, 0, n, 150)
out = (C.c_uint64 * 32)()
raw = C.CDLL(_lib.LIB_PATH)
f.insertSeqs(reads, read_len=150); torch.cuda.synchronize()
raw.btlbf_debug_stamps(out)
f.insertSeqs(reads, read_len=150); torch.cuda.synchronize()
raw.btlbf_debug_stamps(out)
v = list(out)
names = ["phase 1 (atomics, ring writes, late stores)", "  wait barrier 1", "segment b: X flush | Y hash ahead / stage next tile",
         "  wait barrier 2", "-", "-", "-", "-", "Y: late entries into the rings (phase 3)", "hash (round 0: all; round 1: X) | Y: stage request",
         "-", "-"]
for role, off in (("X (wave 0)", 0), ("Y (wave 8)", 12)):
    tot = sum(v[off:off + 12]) or 1
    print(role)
    for i, nme in enumerate(names):
        print("  %-45s %6.2f %%" % (nme, 100.0 * v[off + i] / tot))

tot = sum(v[0:12]) or 1
print("inside the X wave's flush (part_round_p2), share of the X wave's tile loop")
for i, nme in enumerate(["ring state, late words, signal", "prefix sum + flush items", "flush loop (LDS chunk -> 128-byte store)"]):
    print("  %-45s %6.2f %%" % (nme, 100.0 * v[24 + i] / tot))

"""diagnostic: per-phase cycle shares of pass A / pass B (library built with -DBTLBF_PHASE_STAMPS)"""
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
from btl_bloomfilter_amd import _lib
_lib.LIB_PATH = os.path.join(os.getcwd(), "btl_bloomfilter_amd", "libbtlbf_stamps.so")
lib = _lib.load(_lib.LIB_PATH)
import torch
import btl_bloomfilter_amd as m
f = m.BloomFilter(1 << 39, 4, 31); f.setInsertMode("partitioned")
reads = m.synth_reads_device(42, 0, 30_000_000, 150)
out = (C.c_uint64 * 16)()
raw = C.CDLL(_lib.LIB_PATH)
import os
os.environ.setdefault("X", "1")
f.insertSeqs(reads, read_len=150); torch.cuda.synchronize()
raw.btlbf_debug_stamps(out)
f.insertSeqs(reads, read_len=150); torch.cuda.synchronize()
raw.btlbf_debug_stamps(out)
v = list(out); tot = sum(v)
names = ["loop+bitmap", "tile staging: own loads + conversion", "hash (start-up + rolls)", "tile staging: barrier wait", "phase 1: atomics + ring write (+barrier)", "-", "phase 2: flush (+barrier)", "phase 3: late entries", "-", "finish"]
for n, x in zip(names, v): print("%-20s %6.2f %%  %d" % (n, 100.0 * x / tot, x))

"""diagnostic: HIP-event time of pass A (insert / query) at the C2 geometry for other hash counts:
    [BTLBF_PART_OVERLAP=0] python tools/h_probe.py 3 4 5 6 8   (3x10^7 reads of 150 bp, 2^39-bit filter)"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import btl_bloomfilter_amd as m
n, L = 30_000_000, 150
reads = m.synth_reads_device(42, 0, n, L)
for h in (int(x) for x in sys.argv[1:]):
    f = m.BloomFilter(1 << 39, h, 31)
    f.setInsertMode("partitioned"); f.setQueryMode("partitioned"); f.setProfiling(True)
    for rep in range(2):
        f.insertSeqs(reads, read_len=L)
        f.containsSeqs(reads, read_len=L, want_valid=False)
        torch.cuda.synchronize()
        prof = f.getProfile()
    print("h=%d overlap=%s  insert_hash %.2f ms x%d  query_hash %.2f ms x%d" % (h, os.environ.get("BTLBF_PART_OVERLAP", "1"), prof["insert_hash"][0] / prof["insert_hash"][1], prof["insert_hash"][1], prof["query_hash"][0] / prof["query_hash"][1], prof["query_hash"][1]), flush=True)
    del f

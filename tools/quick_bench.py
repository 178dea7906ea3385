"""Quick timing of the fused insert / contains kernels on synthetic reads (not the contract bench)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import btl_bloomfilter_amd as m


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 39
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    k, h, L = 31, 4, 150
    mode = sys.argv[3] if len(sys.argv) > 3 else "auto"
    bits = int(eval(os.environ["QB_BITS"])) if os.environ.get("QB_BITS") else 1 << lg  # e.g. QB_BITS=3*2**37
    f = m.BloomFilter(bits, h, k)
    f.setInsertMode(mode)
    f.setQueryMode(sys.argv[4] if len(sys.argv) > 4 else "auto")
    if os.environ.get("QB_SPACED"):  # BASELINE config 5: four spaced seeds x h2 = 1
        f.setSpacedSeeds(["1110111011101110111011101110111", "1101101101101101011011011011011",
                          "1111001111001111111001111001111", "1011101011101011101011101011101"], 1)
    if os.environ.get("QB_PROFILE"):  # per-kernel HIP-event times of the last repetition
        f.setProfiling(True)
    hit_only = len(sys.argv) > 5 and sys.argv[5] == "hitonly"  # skip the all-miss query (profiling runs)
    reads = m.synth_reads_device(42, 0, n_reads, L)
    q = reads if hit_only else m.synth_reads_device(43, 0, n_reads, L)
    lay = dict(read_len=L)
    if os.environ.get("QB_RAGGED"):  # the same bases cut into sequences of 100..200 bases (btlbf_layout::starts)
        g = torch.Generator(device="cuda")
        g.manual_seed(1)
        lens = torch.randint(100, 201, (n_reads + n_reads // 8,), device="cuda", generator=g, dtype=torch.int64)
        st_ = torch.cumsum(lens, 0)
        st_ = st_[st_ < n_reads * L]
        lay = dict(starts=torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), st_,
                                     torch.tensor([n_reads * L], dtype=torch.int64, device="cuda")]))
    torch.cuda.synchronize()
    kmers = n_reads * (L - k + 1)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    st = torch.cuda.current_stream()
    for rep in range(3):
        f.clear(stream=st)
        ev[0].record()
        f.insertSeqs(reads, stream=st, **lay)
        ev[1].record()
        _, _, c1 = f.containsSeqs(reads, want_valid=False, want_counts=True, stream=st, **lay)
        ev[2].record()
        _, _, c2 = f.containsSeqs(q, want_valid=False, want_counts=True, stream=st, **lay)
        ev[3].record()
        torch.cuda.synchronize()
        ti, th, tm = ev[0].elapsed_time(ev[1]) / 1e3, ev[1].elapsed_time(ev[2]) / 1e3, ev[2].elapsed_time(ev[3]) / 1e3
        print(json.dumps({"mode": mode, "log2_bits": lg, "reads": n_reads, "insert_Gkmers_s": kmers / ti / 1e9,
                          "query_hit_Gkmers_s": kmers / th / 1e9, "query_miss_Gkmers_s": kmers / tm / 1e9,
                          "hits": c1.tolist(), "miss": c2.tolist()}), flush=True)
        if os.environ.get("QB_PROFILE"):
            prof = f.getProfile(reset=True)
            if rep == 2:
                print(json.dumps({a: round(b[0] / b[1], 2) for a, b in prof.items() if b[1]}), flush=True)


if __name__ == "__main__":
    main()

"""Random-access ceilings of the HBM system under the filter's own access pattern:
bare 4-byte gathers and 4-byte atomicOr at random 64-byte-aligned offsets, per filter size."""
import json
import sys
import time

import torch

import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import btl_bloomfilter_amd as m


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [30, 33, 36, 39]
    for lg in sizes:
        f = m.BloomFilter(1 << lg, 4, 31)
        row = {"log2_bits": lg, "bytes": (1 << lg) // 8}
        for kind, name in ((0, "gather"), (1, "atomic_or")):
            f.microbench(kind, 1 << 28)  # warm
            n, sec = f.microbench(kind, 1 << 32)
            row[name] = {"accesses": n, "seconds": sec, "G_per_s": n / sec / 1e9, "GBps_64B": n * 64 / sec / 1e9}
        f.clear()
        torch.cuda.synchronize()
        print(json.dumps(row), flush=True)
        f.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Ingestion rates (not the contract bench): FASTQ file -> filter, and host buffer -> filter.

    python tools/fastx_bench.py [n_reads] [log2_bits]

Writes a synthetic 4-line FASTQ (150 bp reads of the SURVEY 8d generator) under $TMPDIR, then times
  * btlbf_insert_fastx / btlbf_contains_fastx on it (parse + pinned double buffer + PCIe + kernels),
  * the same reads handed over as ONE pageable host buffer (BTLBF_HOST: hipMemcpy inside the call),
  * the same reads resident in HBM (the bench.py condition), for scale.
"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import btl_bloomfilter_amd as m


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
    lg = int(sys.argv[2]) if len(sys.argv) > 2 else 36
    k, h, L = 31, 4, 150
    kmers = n_reads * (L - k + 1)
    reads_d = m.synth_reads_device(42, 0, n_reads, L)
    reads = reads_d.cpu().numpy().reshape(n_reads, L)
    # "@r%09d\n" + seq + "\n+\n" + qual + "\n"
    hdr = np.frombuffer(b"@r000000000\n", np.uint8)
    rec = np.empty((n_reads, len(hdr) + L + 3 + L + 1), np.uint8)
    rec[:, :len(hdr)] = hdr
    idx = np.arange(n_reads)
    for d in range(9):
        rec[:, 2 + 8 - d] = 48 + (idx // 10 ** d) % 10
    o = len(hdr)
    rec[:, o:o + L] = reads
    rec[:, o + L:o + L + 3] = np.frombuffer(b"\n+\n", np.uint8)
    rec[:, o + L + 3:o + 2 * L + 3] = ord("I")
    rec[:, -1] = 10
    path = os.path.join(os.environ.get("TMPDIR", tempfile.gettempdir()), "btlbf_synth.fq")
    rec.tofile(path)
    file_bytes = rec.nbytes
    del rec
    out = {"reads": n_reads, "kmers": kmers, "log2_bits": lg, "fastq_bytes": file_bytes}

    f = m.BloomFilter(1 << lg, h, k)
    for batch in (0, 256 << 20):  # 0 = the library's default (64 MiB per parser thread, 256 MiB with one)
        f.clear()
        torch.cuda.synchronize()
        st = f.insertFile(path, batch_bytes=batch)
        q = f.containsFile(path, batch_bytes=batch)
        assert q["n_windows"] == kmers and q["n_hits"] == kmers, q
        out["file_batch_%s" % ("default" if not batch else "%dMiB" % (batch >> 20))] = {
            "insert_Mkmers_s": kmers / st["seconds_total"] / 1e6, "insert_file_GB_s": file_bytes / st["seconds_total"] / 1e9,
            "insert_parse_s": st["seconds_parse"], "insert_total_s": st["seconds_total"],
            "contains_Mkmers_s": kmers / q["seconds_total"] / 1e6, "contains_parse_s": q["seconds_parse"],
            "contains_total_s": q["seconds_total"], "batches": st["n_batches"]}
    pop_file = f.getPop()

    # one pageable host buffer through the C ABI (BTLBF_HOST)
    f.clear()
    flat = np.ascontiguousarray(reads.reshape(-1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f.insertSeqs(flat, read_len=L)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out["host_buffer_pageable"] = {"insert_Mkmers_s": kmers / (t1 - t0) / 1e6, "GB_s": flat.nbytes / (t1 - t0) / 1e9}
    assert f.getPop() == pop_file

    # resident in HBM
    f.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f.insertSeqs(reads_d, read_len=L)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out["resident"] = {"insert_Mkmers_s": kmers / (t1 - t0) / 1e6}
    assert f.getPop() == pop_file
    os.unlink(path)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

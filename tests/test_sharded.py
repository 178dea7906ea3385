"""Multi-rank path (hash-range shards + all-to-all routing).

CPU: world_size 2 under gloo with an oracle-backed stand-in for the per-rank kernels -- checks that
the routing produces the single-filter body and the single-filter contains() answers.
GPU: the same with the real HIP kernels (both ranks on the one GPU of the test box, exchange staged
through the host by gloo), checked against the golden digest of the reference."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden
from shard_helpers import cpu_worker, free_port, gpu_worker, gpu_worker_counting, gpu_worker_routed, sha


def expected(oracle, bits, h, k, world, n_reads, L):
    body = np.zeros(bits // 8, np.uint8)
    for r in oracle.synth_reads(42, 0, world * n_reads, L).reshape(-1, L):
        oracle.bf_insert_seq(body, bits, h, k, r.tobytes())
    return body


def check_queries(oracle, body, bits, h, k, world, n_reads, L, outdir):
    for rank in range(world):
        q = np.concatenate([oracle.synth_reads(42, rank * n_reads, n_reads, L), oracle.synth_reads(43, rank * n_reads, n_reads, L)])
        eh, ev = [], []
        for r in q.reshape(-1, L):
            a, b = oracle.bf_contains_seq_dense(body, bits, h, k, r.tobytes())
            eh.append(np.concatenate([a, np.zeros(k - 1, np.uint8)]))
            ev.append(np.concatenate([b, np.zeros(k - 1, np.uint8)]))
        eh, ev = np.concatenate(eh), np.concatenate(ev)
        hit = np.unpackbits(np.load(os.path.join(outdir, "hit%d.npy" % rank)).view(np.uint8), bitorder="little")[: q.size]
        assert (hit == eh).all(), rank
        cnt = np.load(os.path.join(outdir, "cnt%d.npy" % rank))
        assert cnt.tolist() == [int(ev.sum()), int(eh.sum())]
        assert eh[: n_reads * L].sum() == n_reads * (L - k + 1)  # own reads: no false negatives


@pytest.mark.parametrize("msg_bytes", [None, 4096])
def test_sharded_routing_gloo_cpu(oracle, tmp_path, msg_bytes, monkeypatch):
    # msg_bytes: the largest single message of the exchange -- small enough here that every all-to-all
    # is cut into several rounds of slices (what protects real runs from > 1 GiB messages)
    if msg_bytes:
        monkeypatch.setenv("BTLBF_TEST_MSG_BYTES", str(msg_bytes))
    bits, h, k, world, n_reads, L = 1 << 16, 4, 31, 2, 128, 150
    mp.spawn(cpu_worker, args=(world, free_port(), str(tmp_path), bits, h, k, n_reads, L), nprocs=world, join=True)
    body = expected(oracle, bits, h, k, world, n_reads, L)
    got = np.concatenate([np.load(tmp_path / ("body%d.npy" % r)) for r in range(world)])
    assert (got == body).all(), "concatenated shard bodies differ from the single-filter body"
    check_queries(oracle, body, bits, h, k, world, n_reads, L, str(tmp_path))


@pytest.mark.parametrize("mode", ["exchange", "gather"])
def test_sharded_uneven_read_counts_gloo_cpu(oracle, tmp_path, mode):
    """ranks that hold different numbers of reads (rank 1 runs out of batches first) must still take part
    in every collective: world 3, the direct position exchange and the gather path"""
    bits, h, k, world, n_reads, L, uneven = 3 << 14, 3, 25, 3, 150, 150, 60
    mp.spawn(cpu_worker, args=(world, free_port(), str(tmp_path), bits, h, k, n_reads, L, mode, uneven), nprocs=world,
             join=True)
    body = np.zeros(bits // 8, np.uint8)
    for rank in range(world):
        for r in oracle.synth_reads(42, rank * n_reads, n_reads - rank * uneven, L).reshape(-1, L):
            oracle.bf_insert_seq(body, bits, h, k, r.tobytes())
    got = np.concatenate([np.load(tmp_path / ("body%d.npy" % r)) for r in range(world)])
    assert (got == body).all()
    for rank in range(world):
        n = n_reads - rank * uneven
        q = np.concatenate([oracle.synth_reads(42, rank * n_reads, n, L), oracle.synth_reads(43, rank * n_reads, n, L)])
        eh = []
        for r in q.reshape(-1, L):
            a, _ = oracle.bf_contains_seq_dense(body, bits, h, k, r.tobytes())
            eh.append(np.concatenate([a, np.zeros(k - 1, np.uint8)]))
        hit = np.unpackbits(np.load(tmp_path / ("hit%d.npy" % rank)).view(np.uint8), bitorder="little")[: q.size]
        assert (hit == np.concatenate(eh)).all(), rank


@pytest.mark.parametrize("world,bits", [(2, 1 << 16), (3, 3 << 14)])
def test_sharded_gather_mode_gloo_cpu(oracle, tmp_path, world, bits):
    """gather mode over gloo with the oracle stand-in: several rounds of the read gather (the last one
    padded), window-partial answers exchanged and ANDed at the reads' owner"""
    h, k, n_reads, L = 4, 31, 100, 150
    mp.spawn(cpu_worker, args=(world, free_port(), str(tmp_path), bits, h, k, n_reads, L, "gather"), nprocs=world,
             join=True)
    body = expected(oracle, bits, h, k, world, n_reads, L)
    got = np.concatenate([np.load(tmp_path / ("body%d.npy" % r)) for r in range(world)])
    assert (got == body).all(), "concatenated shard bodies differ from the single-filter body"
    check_queries(oracle, body, bits, h, k, world, n_reads, L, str(tmp_path))


def test_sharded_world1_matches_plain_filter_cpu(oracle):
    """world_size 1 without a process group: the same code path degenerates to a plain filter"""
    import torch

    from btl_bloomfilter_amd.sharded import ShardedBloomFilter
    from shard_helpers import OracleShardOps

    bits, h, k, L, n = 1 << 14, 3, 25, 100, 40
    ops = OracleShardOps(bits, h, k, 0, 1)
    f = ShardedBloomFilter(bits, h, k, ops=ops, batch_reads=16)
    reads = oracle.synth_reads(7, 0, n, L)
    f.insert_reads(torch.from_numpy(reads), L)
    body = np.zeros(bits // 8, np.uint8)
    for r in reads.reshape(-1, L):
        oracle.bf_insert_seq(body, bits, h, k, r.tobytes())
    assert (ops.local_body() == body).all()


@pytest.mark.gpu
def test_sharded_hip_two_ranks_one_gpu(oracle, tmp_path):
    g = load_golden("digests.json")["bf_small"]  # 20000 reads, 2^24 bits, k=31, h=4 (genuine reference)
    bits, h, k, L, world = g["bits"], g["h"], g["k"], g["read_len"], 2
    n_reads = g["n_reads"] // world
    mp.spawn(gpu_worker, args=(world, free_port(), str(tmp_path), bits, h, k, n_reads, L), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / ("body%d.npy" % r)) for r in range(world)])
    assert sha(got) == g["body_sha256"]
    raw = open(tmp_path / "sharded.bf", "rb").read()
    i = raw.index(b"[HeaderEnd]\n") + 12
    assert raw[:i] == oracle.bf_header(bits, h, k) and sha(np.frombuffer(raw[i:], np.uint8)) == g["body_sha256"]
    # contains(): first 64 reads of every rank checked window by window, totals for the rest
    for rank in range(world):
        cnt = np.load(tmp_path / ("cnt%d.npy" % rank))
        assert cnt[0] == 2 * n_reads * (L - k + 1) and cnt[1] >= n_reads * (L - k + 1)
        hit = np.unpackbits(np.load(tmp_path / ("hit%d.npy" % rank)).view(np.uint8), bitorder="little")
        q = oracle.synth_reads(42, rank * n_reads, 64, L)
        for j, r in enumerate(q.reshape(-1, L)):
            a, _ = oracle.bf_contains_seq_dense(got, bits, h, k, r.tobytes())
            assert (hit[j * L: j * L + len(a)] == a).all()


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [None, True])
def test_sharded_routed_partitioned_two_ranks_one_gpu(tmp_path, pipeline, monkeypatch):
    """routed path: pass A per origin, fixed-size exchange of 4-byte entries, owner splits + LDS apply;
    queries return only failed positions.  Checked against the reference's golden digest and against
    the single-GPU direct kernels (all-hit, few-miss and miss-heavy queries)."""
    g = load_golden("digests.json")["bf_medium"]  # 200000 reads, 2^30 bits, k=31, h=4
    bits, h, k, L, world = g["bits"], g["h"], g["k"], g["read_len"], 2
    n_reads = g["n_reads"] // world
    # pipeline=True: the schedule of the RCCL path (exchange of batch i in flight while batch i+1 is
    # routed, two buffer sets, deferred apply) over the synchronous test exchange, and every block
    # exchanged in slices of 16 MiB
    if pipeline:
        monkeypatch.setenv("BTLBF_TEST_MSG_BYTES", str(16 << 20))
    mp.spawn(gpu_worker_routed, args=(world, free_port(), str(tmp_path), bits, h, k, n_reads, L, 0, pipeline),
             nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / ("body%d.npy" % r)) for r in range(world)])
    assert sha(got) == g["body_sha256"]
    for rank in range(world):
        res = eval(str(np.load(tmp_path / ("res%d.npy" % rank))[0]))
        for name, (same, cnt, exp) in res.items():
            assert same, (rank, name)
            assert cnt == exp, (rank, name, cnt, exp)
        assert res["hits"][1][0] == res["hits"][1][1] == n_reads * (L - k + 1)
        assert res["few_misses"][1][1] < res["few_misses"][1][0]


@pytest.mark.gpu
def test_sharded_routed_four_ranks_one_gpu(tmp_path):
    """four ranks (four origin blocks per owner, 128 level-0 bins per shard) with the double-buffered
    schedule of the RCCL path; the concatenated shard bodies must be the reference's filter"""
    g = load_golden("digests.json")["bf_medium"]
    bits, h, k, L, world = g["bits"], g["h"], g["k"], g["read_len"], 4
    n_reads = g["n_reads"] // world
    mp.spawn(gpu_worker_routed, args=(world, free_port(), str(tmp_path), bits, h, k, n_reads, L, 0, True),
             nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / ("body%d.npy" % r)) for r in range(world)])
    assert sha(got) == g["body_sha256"]
    for rank in range(world):
        res = eval(str(np.load(tmp_path / ("res%d.npy" % rank))[0]))
        for name, (same, cnt, exp) in res.items():
            assert same, (rank, name)
            assert cnt == exp, (rank, name, cnt, exp)


@pytest.mark.gpu
@pytest.mark.parametrize("world,bits", [(2, 1 << 30), (4, 1 << 30), (3, 3 << 28), (2, 5 * (1 << 27))])
def test_sharded_gather_mode_one_gpu(tmp_path, world, bits):
    """gather mode (the default for 2..4 ranks): reads are all-gathered, every shard hashes all of them and
    keeps the probes inside its window (WINDOW pass A + local split/apply); query partials are ANDed at
    the reads' owner.  Several rounds (4 MiB pieces), power-of-two and other geometries; shard bodies and
    all-hit / few-miss / miss-heavy queries against one filter built by the direct kernels, and the
    reference's golden digest where there is one."""
    g = load_golden("digests.json")["bf_medium"]
    h, k, L = g["h"], g["k"], g["read_len"]
    n_reads = g["n_reads"] // world
    mp.spawn(gpu_worker_routed, args=(world, free_port(), str(tmp_path), bits, h, k, n_reads, L, 0, None, "gather"),
             nprocs=world, join=True)
    if bits == g["bits"] and n_reads * world == g["n_reads"]:
        got = np.concatenate([np.load(tmp_path / ("body%d.npy" % r)) for r in range(world)])
        assert sha(got) == g["body_sha256"]
    for rank in range(world):
        res = eval(str(np.load(tmp_path / ("res%d.npy" % rank))[0]))
        for name, (same, cnt, exp) in res.items():
            assert same, (rank, name)
            assert cnt == exp, (rank, name, cnt, exp)
        assert res["hits"][1][0] == res["hits"][1][1] == n_reads * (L - k + 1)
        assert res["few_misses"][1][1] < res["few_misses"][1][0]


@pytest.mark.gpu
def test_sharded_routed_two_split_levels_and_32bit_entries(tmp_path):
    """the owner-side geometry of a 1 TiB filter on 8 GPUs (32-bit entries, two split passes),
    reproduced at 2^36 bits on 2 ranks by using only 16 level-0 bins (BTLBF_ROUTE_BINS)"""
    import torch

    free, _ = torch.cuda.mem_get_info()
    if free < 40 << 30:
        pytest.skip("needs ~30 GiB of HBM")
    bits, h, k, L, world, n_reads = 1 << 36, 4, 31, 150, 2, 60000
    mp.spawn(gpu_worker_routed, args=(world, free_port(), str(tmp_path), bits, h, k, n_reads, L, 16), nprocs=world, join=True)
    for rank in range(world):
        res = eval(str(np.load(tmp_path / ("res%d.npy" % rank))[0]))
        for name, (same, cnt, exp) in res.items():
            assert same, (rank, name)
            assert cnt == exp, (rank, name, cnt, exp)
    # shard bodies against a single filter built by the direct kernel
    import btl_bloomfilter_amd as m

    ref = m.BloomFilter(bits, h, k)
    ref.setInsertMode("direct")
    ref.insertSeqs(m.synth_reads_device(42, 0, world * n_reads, L), read_len=L)
    body = ref.download()
    got = np.concatenate([np.load(tmp_path / ("body%d.npy" % r)) for r in range(world)])
    assert (got == body).all()


@pytest.mark.gpu
def test_sharded_routed_position_windows_c4_geometry(tmp_path):
    """BASELINE config 4 (2^43 bits on 8 GPUs) has TWO position windows of 2^42 bits, each owned by four
    shards, 32-bit entries and two split passes at the owner.  Reproduced on one GPU with 4 ranks: 2^37
    bits, windows of 2^36 (BTLBF_ROUTE_WINDOW_BITS) with 16 level-0 bins each (2^32 positions per bin).
    Shard bodies against a single filter built by the direct kernel; all-hit, few-miss and miss-heavy
    queries against the direct kernel."""
    import torch

    free, _ = torch.cuda.mem_get_info()
    if free < 120 << 30:
        pytest.skip("needs ~100 GiB of HBM (4 shards + 4 whole reference filters of 16 GiB)")
    bits, h, k, L, world, n_reads = 1 << 37, 4, 31, 150, 4, 60000
    mp.spawn(gpu_worker_routed, args=(world, free_port(), str(tmp_path), bits, h, k, n_reads, L, 16, None, "exchange", 36),
             nprocs=world, join=True)
    for rank in range(world):
        res = eval(str(np.load(tmp_path / ("res%d.npy" % rank))[0]))
        for name, (same, cnt, exp) in res.items():
            assert same, (rank, name)
            assert cnt == exp, (rank, name, cnt, exp)
        assert res["hits"][1][0] == res["hits"][1][1] == n_reads * (L - k + 1)


@pytest.mark.gpu
def test_sharded_routed_skewed_reads_overflow_the_spill_lists(tmp_path):
    """20 000 copies of one read per rank: their probes cannot be staged at the origin and come back as
    explicit positions, far more than a spill list of 4096 holds -- the job is routed again into a larger
    list, nothing is lost and nothing raises (bit filter: the body equals the direct kernel's)"""
    bits, h, k, L, world, n_reads = 1 << 30, 4, 31, 150, 2, 20000
    mp.spawn(gpu_worker_routed, args=(world, free_port(), str(tmp_path), bits, h, k, n_reads, L, 0, None, "exchange", 0,
                                      4096, 20000), nprocs=world, join=True)
    for rank in range(world):
        res = eval(str(np.load(tmp_path / ("res%d.npy" % rank))[0]))
        for name, (same, cnt, exp) in res.items():
            assert same, (rank, name)
            assert cnt == exp, (rank, name, cnt, exp)


def test_sharded_routed_mode_is_never_silently_replaced(oracle):
    """mode="routed" with ops (or a geometry) that have no routed path raises instead of falling back to the
    direct position exchange"""
    from shard_helpers import OracleShardOps

    from btl_bloomfilter_amd.sharded import ShardedBloomFilter

    with pytest.raises(ValueError, match="routed"):
        ShardedBloomFilter(1 << 16, 4, 31, ops=OracleShardOps(1 << 16, 4, 31, 0, 1), mode="routed")


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal(tmp_path):
    """bench.py's N > 1 path (launch contract, sharded filter, max-over-ranks timing, the JSON line) with
    two ranks on the one GPU and the gloo backend; RCCL itself needs one GPU per rank"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BTLBF_BENCH_REHEARSAL="1", PYTHONPATH=root)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1",
           "--warmup", "1", "--reads", "200000", "--log2-bits", "30", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["kmers_per_pass"] == 2 * 200000 * 120


@pytest.mark.gpu
def test_bench_four_ranks_routed_windows_rehearsal(tmp_path):
    """bench.py's N = 8 code path -- the routed exchange with position windows, as config 4 takes it -- rehearsed
    with four ranks on the one GPU over gloo: 2^37 bits in windows of 2^36, every k-mer must come back a hit"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BTLBF_BENCH_REHEARSAL="1", PYTHONPATH=root, BTLBF_SHARD_MODE="routed",
               BTLBF_ROUTE_WINDOW_BITS="36")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1",
           "--warmup", "1", "--reads", "300000", "--log2-bits", "35", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["value"] > 0 and d["config"]["collective_world_size"] == 4
    assert d["config"]["kmers_per_pass"] == 4 * 300000 * 120
    assert "partitioned routing" in d["config"]["parallelism"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode,world", [("exchange", 2), ("gather", 2), ("gather", 4)])
def test_sharded_counting_filter_two_ranks_one_gpu(tmp_path, mode, world):
    """the counting filter shards like the bit filter -- incrementAll is shard-local at the
    owners and exact (saturating), contains() = minimum >= threshold; against one unsharded filter.
    Routed path and gather mode (counters inside the shard's window)."""
    counters, h, k, thr, L, n_reads = 1 << 28, 3, 25, 2, 150, 80000 // world
    mp.spawn(gpu_worker_counting, args=(world, free_port(), str(tmp_path), counters, h, k, thr, n_reads, L, mode),
             nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / ("body%d.npy" % r)) for r in range(world)])
    ref = np.load(tmp_path / "ref.npy")
    assert got.shape == ref.shape and (got == ref).all() and got.max() >= 2
    # the file the shards wrote together is the unsharded filter's file, byte for byte
    assert open(tmp_path / "sharded.bf", "rb").read() == open(tmp_path / "ref.bf", "rb").read()
    for rank in range(world):
        same, cnt, exp = eval(str(np.load(tmp_path / ("res%d.npy" % rank))[0]))
        assert same and cnt == exp, (rank, cnt, exp)
        assert (n_reads // 2) * (L - k + 1) <= cnt[1] < cnt[0]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["exchange", "gather"])
def test_sharded_counting_filter_golden_digest(tmp_path, mode):
    """the sharded counting filter against the REFERENCE's own counters: golden fixture cbf_medium_all (2^27
    uint8_t counters, k=25, h=3, 200 000 reads through incrementAll and the first 100 000 a second time) built by
    two shards, routed and gathered; rank 1 has nothing to insert in the second pass and takes part empty-handed"""
    g = load_golden("digests.json")["cbf_medium_all"]
    world = 2
    mp.spawn(gpu_worker_counting, args=(world, free_port(), str(tmp_path), g["bytes"], g["h"], g["k"], g["thr"],
                                        g["n_reads"] // world, g["read_len"], mode, "golden"), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / ("body%d.npy" % r)) for r in range(world)])
    assert got.size == g["bytes"] and sha(got) == g["body_sha256"]
    assert int((got != 0).sum()) == g["popcount"] and int((got >= g["thr"]).sum()) == g["filtered_popcount"]


def _nccl_one_rank_worker(rank, port, outdir, mode, shard_mode="exchange"):
    import btl_bloomfilter_amd as m
    from btl_bloomfilter_amd.sharded import ShardedBloomFilter

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BTLBF_FORCE_EXCHANGE=mode)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    bits, h, k, L, n = 1 << 33, 4, 31, 150, 1_500_000
    f = ShardedBloomFilter(bits, h, k, device=0, batch_bytes_cap=80 << 20, mode=shard_mode)  # three batches, pipelined
    assert f.mode == shard_mode
    f.MSG_BYTES = (64 << 20) if shard_mode == "exchange" else (16 << 20)  # several slices per block
    reads = m.synth_reads_device(42, 0, n, L)
    f.insert_reads(reads, L)
    hit = torch.zeros((reads.numel() + 63) // 64, dtype=torch.int64, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int64)
    q = torch.cat([reads[: 1000 * L], m.synth_reads_device(43, 0, 30, L)])
    f.contains_reads(q, L, hit[: (q.numel() + 63) // 64], cnt)
    ref = m.BloomFilter(bits, h, k)
    ref.insertSeqs(reads, read_len=L)
    eh, _, ec = ref.containsSeqs(q, read_len=L, want_valid=False, want_counts=True)
    torch.cuda.synchronize()
    ok = bool((torch.from_numpy(f.ops.local_body()) == torch.from_numpy(ref.download())).all())
    same = bool((hit[: (q.numel() + 63) // 64] == eh).all().item())
    np.save(os.path.join(outdir, "nccl1.npy"), np.array([ok, same, cnt.tolist() == ec.cpu().tolist()]))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1", "2"])
def test_routed_path_over_rccl_with_one_rank(tmp_path, mode):
    """the production exchange calls (torch.distributed on RCCL: grouped all-to-all of slices, async
    work handles, stream waits, the double-buffered schedule) with a one-rank group: every block
    travels GPU -> RCCL -> same GPU (mode 1), or -- as in production -- the rank's own block is a local
    copy and the collective carries the peers' blocks only, here none (mode 2).  Bodies and query results
    against the plain filter."""
    mp.spawn(_nccl_one_rank_worker, args=(free_port(), str(tmp_path), mode), nprocs=1, join=True)
    assert np.load(tmp_path / "nccl1.npy").all()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1", "2"])
def test_gather_mode_over_rccl_with_one_rank(tmp_path, mode):
    """gather mode's collective calls on RCCL with a one-rank group: the read gather in slices (async work
    handles, two buffers, three rounds) and the exchange of the partial bitmaps; mode 1 sends the rank's
    own piece through RCCL, mode 2 copies it locally as in production"""
    mp.spawn(_nccl_one_rank_worker, args=(free_port(), str(tmp_path), mode, "gather"), nprocs=1, join=True)
    assert np.load(tmp_path / "nccl1.npy").all()


@pytest.mark.gpu
@pytest.mark.parametrize("cap", [1_000_000, 2_110_000, 2_500_000, 3_300_000])
def test_routed_small_batches_with_uneven_tiles_per_workgroup(cap):
    """batches of a few hundred pass-A tiles: 257 tiles over 256 workgroups means two tiles for half of
    them and none for the rest -- region capacities must be planned for the fullest region (the first
    version planned for the mean and overflowed into the spill list).  World 1, routed path, no exchange."""
    import torch

    import btl_bloomfilter_amd as m
    from btl_bloomfilter_amd.sharded import ShardedBloomFilter

    bits, h, k, L, n = 1 << 30, 4, 31, 150, 40000
    f = ShardedBloomFilter(bits, h, k, device=0, batch_bytes_cap=cap, mode="exchange")
    assert f._routed()
    reads = m.synth_reads_device(42, 0, n, L)
    f.insert_reads(reads, L)
    ref = m.BloomFilter(bits, h, k)
    ref.setInsertMode("direct")
    ref.insertSeqs(reads, read_len=L)
    assert (f.ops.local_body() == ref.download()).all()
    q = torch.cat([reads[: 2000 * L], m.synth_reads_device(43, 0, 10, L)])
    hit = torch.zeros((q.numel() + 63) // 64, dtype=torch.int64, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int64)
    f.contains_reads(q, L, hit, cnt)
    eh, _, ec = ref.containsSeqs(q, read_len=L, want_valid=False, want_counts=True)
    assert bool((hit == eh).all().item()) and cnt.tolist() == ec.cpu().tolist()


@pytest.mark.gpu
def test_apply_routed_bins_group_by_group_equals_whole_block():
    """C ABI, owner side: the bins of a routed block applied a group at a time from compact per-group
    buffers (btlbf_route_geometry / btlbf_apply_routed_bins) give the same filter as the whole block
    (btlbf_apply_routed); bin ranges that are not whole groups are refused."""
    import ctypes as C

    import torch

    import btl_bloomfilter_amd as m
    from btl_bloomfilter_amd import _lib
    from btl_bloomfilter_amd.sharded import HipShardOps

    bits, h, k, L, n = 1 << 32, 4, 31, 150, 60000
    reads = m.synth_reads_device(42, 0, n, L)
    plan = reads.numel()
    bodies = []
    for grouped in (False, True):
        ops = HipShardOps(bits, h, k, 0, 1, 0)
        ent_b, cnt_b = ops.route_plan(plan, L)
        bins, regions, cap, gb = ops.route_geometry(plan, L)
        assert ent_b == bins * regions * cap * 128 and cnt_b == bins * regions * 4 and bins % gb == 0 and gb < bins
        send_ent = torch.empty(ent_b, dtype=torch.uint8, device="cuda")
        send_cnt = torch.empty(cnt_b, dtype=torch.uint8, device="cuda")
        spill = torch.empty(1 << 16, dtype=torch.int64, device="cuda")
        spill_count = torch.zeros(1, dtype=torch.int64, device="cuda")
        ops.route(reads, L, plan, 0, send_ent, send_cnt, None, None, None, spill, spill_count)
        assert int(spill_count.item()) == 0
        if not grouped:
            ops.apply_routed(send_ent, send_cnt, 1, plan, L, 0, None, None)
        else:
            ge, gc = ent_b // (bins // gb), cnt_b // (bins // gb)
            for g in reversed(range(bins // gb)):  # any order
                e = send_ent[g * ge:(g + 1) * ge].clone()
                c = send_cnt[g * gc:(g + 1) * gc].clone()
                ops.apply_routed_bins(e, c, 1, g * gb, gb, plan, L, 0, None, None)
            with pytest.raises(_lib.BtlbfError):  # half a group
                ops.apply_routed_bins(send_ent, send_cnt, 1, 0, gb // 2 or gb + 1, plan, L, 0, None, None)
            with pytest.raises(_lib.BtlbfError):  # beyond the shard's bins
                ops.apply_routed_bins(send_ent, send_cnt, 1, bins, gb, plan, L, 0, None, None)
        torch.cuda.synchronize()
        bodies.append(ops.local_body())
        ops.close()
    ref = m.BloomFilter(bits, h, k)
    ref.setInsertMode("direct")
    ref.insertSeqs(reads, read_len=L)
    assert (bodies[0] == ref.download()).all() and (bodies[1] == bodies[0]).all()

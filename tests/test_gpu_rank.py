"""Rank structure over a bit filter (btlbf_rank_*): the layout of sdsl::bit_vector_il<512> and the answers of
sdsl::rank_support_il<1> as the reference's miBF uses them (MIBloomFilter.hpp:44,133,144,527,801-803), checked
against numpy.  sdsl-lite is an un-vendored dependency of the reference, so numpy's cumulative sum over the
filter body is the checker here."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

C5_SEEDS = ["1110111011101110111011101110111", "1101101101101101011011011011011",
            "1111001111001111111001111001111", "1011101011101011101011101011101"]


@pytest.fixture(scope="module")
def bf():
    import torch

    assert torch.cuda.is_available()
    torch.zeros(1, device="cuda")
    import btl_bloomfilter_amd as m

    return m


@pytest.mark.parametrize("bits", [512, 1000, 4096, 1 << 20, (1 << 22) + 64, 3 << 21])
def test_rank_structure_against_numpy(bf, bits):
    rng = np.random.RandomState(bits % 9973)
    body = rng.randint(0, 256, bits // 8).astype(np.uint8)
    body[rng.rand(body.size) < 0.5] = 0  # long runs of empty blocks
    f = bf.BloomFilter(bits, 3, 21)
    f.upload(body)
    rs = bf.RankSupport(f)
    b = np.unpackbits(body, bitorder="little")
    csum = np.concatenate([[0], np.cumsum(b, dtype=np.uint64)])  # csum[p] = set bits before p
    assert rs.ones() == int(b.sum()) == f.getPop()
    il = rs.interleaved()
    n_blocks = (bits + 511) // 512
    assert il.shape == (n_blocks, 9)
    assert (il[:, 0] == csum[np.arange(n_blocks) * 512]).all()
    words = np.zeros(n_blocks * 8, np.uint64)
    words[: bits // 64] = body[: bits // 64 * 8].view(np.uint64)
    if bits % 64:
        tail = np.zeros(8, np.uint8)
        tail[: (bits % 64) // 8] = body[bits // 64 * 8:]
        words[bits // 64] = tail.view(np.uint64)[0]
    assert (il[:, 1:].ravel() == words).all()
    pos = np.concatenate([[0, bits - 1, 511 % bits, 512 % bits], rng.randint(0, bits, 5000)]).astype(np.uint64)
    r, bit = rs.rank(pos)
    assert (r == csum[pos.astype(np.int64)]).all() and (bit == b[pos.astype(np.int64)]).all()
    # getRankPos(hash) = rank(hash % size)  (MIBloomFilter.hpp:527)
    hv = rng.randint(0, 2**63, 3000).astype(np.uint64) * np.uint64(2) + np.uint64(1)
    r, bit = rs.rank(hv, hashes=True)
    p = (hv % np.uint64(bits)).astype(np.int64)
    assert (r == csum[p]).all() and (bit == b[p]).all()


def test_rank_over_the_mibf_stage1_filter(bf):
    """miBF construction order: stage 1 sets the bits of the spaced-seed hashes (MIBFConstructSupport.hpp:75-87),
    stage 2 builds the rank structure, stage 3 addresses the ID array with rank(pos) of every hash
    (MIBFConstructSupport.hpp:109-130).  Every hash of an inserted k-mer finds its bit set and distinct
    positions get distinct ranks 0 .. ones-1."""
    import torch

    bits, k = 1 << 26, 31
    reads = bf.synth_reads_device(42, 0, 3000, 150)
    f = bf.BloomFilter(bits, 4, k)
    f.setSpacedSeeds(C5_SEEDS, 1)
    f.insertSeqs(reads, read_len=150)
    torch.cuda.synchronize()
    rs = bf.RankSupport(f)
    assert rs.ones() == f.getPop()
    hv, valid, _ = bf.sthash_seqs(reads[: 300 * 150], C5_SEEDS, 1, k, read_len=150)
    v = bf.bits_to_bool(valid.cpu().numpy().view(np.uint64), 300 * 150)
    hashes = hv.cpu().numpy().view(np.uint64)[v].ravel()
    r, bit = rs.rank(hashes, hashes=True)
    assert bit.all() and int(r.max()) < rs.ones()
    pos = hashes % np.uint64(bits)
    up, first = np.unique(pos, return_index=True)
    assert len(np.unique(r[first])) == len(up)  # rank is injective on set bits
    order = np.argsort(up)
    assert (np.diff(r[first][order].astype(np.int64)) > 0).all()  # and monotone in the position

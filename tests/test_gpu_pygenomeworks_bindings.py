"""The reference's own binding tests (pygenomeworks/test/test_cudapoa_bindings.py:27-152,
test_cudaaligner_bindings.py:27-108), run against the Cython package `genomeworks` built from pygenomeworks/ --
same calls, same assertions. The reference draws its aligner inputs from its read simulators (out of scope here);
random ACGT strings with substitutions and deletions stand in for them."""
import os
import random
import sys
from difflib import SequenceMatcher

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _package():
    from genomeworks_amd import build
    pkg = build.build_bindings()
    sys.path.insert(0, pkg)
    yield
    sys.path.remove(pkg)


def _batch(*args, **kwargs):
    from genomeworks.cudapoa import CudaPoaBatch
    return CudaPoaBatch(*args, **kwargs)


def _mem():
    import genomeworks.cuda as cuda
    device = cuda.cuda_get_device()
    free, total = cuda.cuda_get_mem_info(device)
    assert 0 < free <= total
    return device, free


def test_cudapoa_simple_batch():
    device, free = _mem()
    batch = _batch(10, 1024, 0.9 * free, deivce_id=device, output_mask='consensus')  # the reference's misspelt kwargs are swallowed
    poa_1 = ["ACTGACTG", "ACTTACTG", "ACGGACTG", "ATCGACTG"]
    poa_2 = ["ACTGAC", "ACTTAC", "ACGGAC", "ATCGAC"]
    batch.add_poa_group(poa_1)
    batch.add_poa_group(poa_2)
    batch.generate_poa()
    consensus, coverage, status = batch.get_consensus()
    assert len(consensus) == 2
    assert batch.total_poas == 2
    assert consensus == ["ACTGACTG", "ACTGAC"] and status == [0, 0]


def test_cudapoa_banded_aligned_batch():
    device, free = _mem()
    batch = _batch(10, 1024, 0.9 * free, deivce_id=device, output_mask='consensus', cuda_banded_alignment=True)
    batch.add_poa_group(["ACTGACTG", "ACTTACTG", "ACGGACTG", "ATCGACTG"])
    batch.add_poa_group(["ACTGAC", "ACTTAC", "ACGGAC", "ATCGAC"])
    batch.generate_poa()
    consensus, coverage, status = batch.get_consensus()
    assert len(consensus) == 2
    assert batch.total_poas == 2


def test_cudapoa_incorrect_and_valid_output_type():
    device, free = _mem()
    with pytest.raises(RuntimeError):
        _batch(10, 1024, 0.9 * free, deivce_id=device, output_type='error_input')
    _batch(10, 1024, 0.9 * free, deivce_id=device, output_type='consensus')
    msa_batch = _batch(10, 1024, 0.5 * free, device_id=device, output_type='msa')
    msa_batch.add_poa_group(["ACTGACTG", "ACTTACTG"])
    msa_batch.generate_poa()
    with pytest.raises(RuntimeError):
        msa_batch.get_consensus()  # output type not requested
    msa, status = msa_batch.get_msa()
    assert status == [0] and [r.replace("-", "") for r in msa[0]] == ["ACTGACTG", "ACTTACTG"]


def test_cudapoa_reset_batch():
    device, free = _mem()
    batch = _batch(10, 1024, 0.9 * free, device_id=device)
    batch.add_poa_group(["ACTGACTG", "ACTTACTG", "ACGGACTG", "ATCGACTG"])
    batch.generate_poa()
    consensus, coverage, status = batch.get_consensus()
    assert batch.total_poas == 1
    batch.reset()
    assert batch.total_poas == 0


def test_cudapoa_graph():
    device, free = _mem()
    batch = _batch(10, 1024, 0.9 * free, device_id=device)
    batch.add_poa_group(["ACTGACTG", "ACTTACTG", "ACTCACTG"])
    batch.generate_poa()
    consensus, coverage, status = batch.get_consensus()
    assert batch.total_poas == 1
    graphs, status = batch.get_graphs()
    assert len(graphs) == 1
    digraph = graphs[0]
    assert digraph.number_of_nodes() == 10
    assert digraph.number_of_edges() == 11
    assert sorted(d["label"] for _, d in digraph.nodes(data=True)) == sorted("ACTGACTG" + "TC")


def test_cudapoa_complex_batch():
    random.seed(2)
    read_len = 500
    ref = ''.join([random.choice(['A', 'C', 'G', 'T']) for _ in range(read_len)])
    num_reads = 100
    mutation_rate = 0.02
    reads = []
    for _ in range(num_reads):
        reads.append(''.join([r if random.random() > mutation_rate else random.choice(['A', 'C', 'G', 'T']) for r in ref]))
    device, free = _mem()
    batch = _batch(1000, 1024, 0.9 * free, device_id=device)
    (add_status, seq_status) = batch.add_poa_group(reads)
    assert add_status == 0 and seq_status == [0] * num_reads
    batch.generate_poa()
    consensus, coverage, status = batch.get_consensus()
    consensus = consensus[0]
    assert len(consensus) == len(ref)
    assert SequenceMatcher(None, ref, consensus).ratio() == 1.0


def test_cudapoa_weights_and_streams():
    import genomeworks.cuda as cuda
    device, free = _mem()
    stream = cuda.CudaStream()
    assert isinstance(stream.stream, int) and stream.stream != 0
    batch = _batch(10, 256, 0.2 * free, device_id=device, stream=stream, band_mode="full_band")
    st, seq = batch.add_poa_group(["ACGTACGT", "ACGAACGT", "ACGAACGT"], weights=[[1] * 8, [9] * 8, None])
    assert st == 0 and seq == [0, 0, 0]
    batch.generate_poa()
    stream.sync()
    consensus, coverage, status = batch.get_consensus()
    assert consensus == ["ACGAACGT"] and status == [0]


@pytest.mark.parametrize("query, target, cigar", [
    ("AAAAAAA", "TTTTTTT", "7M"),
    ("AAATC", "TACGTTTT", "3M1I2M2I"),
    ("TACGTA", "ACATAC", "1D5M1I"),
    ("TGCA", "ATACGCT", "1I1M2I3M"),
    pytest.param("ACGT", "TCGA", "5M", marks=pytest.mark.xfail(strict=True)),
])
def test_cudaaligner_simple_batch(query, target, cigar):
    import genomeworks.cuda as cuda
    from genomeworks.cudaaligner import CudaAlignerBatch
    device = cuda.cuda_get_device()
    stream = cuda.CudaStream()
    batch = CudaAlignerBatch(len(query), len(target), 1, alignment_type="global", stream=stream, device_id=device,
                             max_device_memory_allocator_caching_size=1 << 30)
    batch.add_alignment(query, target)
    batch.align_all()
    alignments = batch.get_alignments()
    assert len(alignments) == 1
    assert alignments[0].cigar == cigar
    assert alignments[0].query == query and alignments[0].target == target and alignments[0].status == 0
    assert alignments[0].alignment_type == "global"
    # the formatted alignment spells both sequences with gaps
    q_line, pairing, t_line = alignments[0].format_alignment
    assert q_line.replace("-", "") == query and t_line.replace("-", "") == target and len(pairing) == len(q_line)


def _noisy_pair(rng, length):
    ref = "".join(rng.choice("ACGT") for _ in range(length))

    def read():
        out = []
        for c in ref:
            u = rng.random()
            if u < 0.01:
                continue                      # deletion
            out.append(rng.choice("ACGT") if u < 0.03 else c)
        return "".join(out)
    return read(), read()


@pytest.mark.parametrize("ref_length, num_alignments", [(5000, 30), (10000, 10), (500, 100)])
def test_cudaaligner_long_alignments(ref_length, num_alignments):
    import genomeworks.cuda as cuda
    from genomeworks.cudaaligner import CudaAlignerBatch
    device = cuda.cuda_get_device()
    rng = random.Random(ref_length)
    batch = CudaAlignerBatch(ref_length, ref_length, num_alignments, device_id=device,
                             max_device_memory_allocator_caching_size=8 << 30)
    pairs = [_noisy_pair(rng, ref_length) for _ in range(num_alignments)]
    for query, target in pairs:
        assert batch.add_alignment(query, target) == 0
    batch.align_all()
    res = batch.get_alignments()
    assert len(res) == num_alignments and all(a.status == 0 for a in res)
    for (query, target), a in zip(pairs[:3], res[:3]):
        assert a.query == query and a.target == target
        assert sum(1 for s in a.alignment if s != "i") == len(query) and sum(1 for s in a.alignment if s != "d") == len(target)
    batch.reset()
    assert len(batch.get_alignments()) == 0


@pytest.mark.parametrize("max_seq_len, max_alignments, seq_len, num_alignments, should_succeed", [
    (1000, 100, 10000, 10, False),
    (1000, 100, 100, 10, True),
    (1000, 100, 1000, 100, True),
    (100, 10, 100, 1000, False),
])
def test_cudaaligner_various_arguments(max_seq_len, max_alignments, seq_len, num_alignments, should_succeed):
    import genomeworks.cuda as cuda
    from genomeworks.cudaaligner import CudaAlignerBatch
    device = cuda.cuda_get_device()
    rng = random.Random(seq_len * 31 + num_alignments)
    batch = CudaAlignerBatch(max_seq_len, max_seq_len, max_alignments, device_id=device,
                             max_device_memory_allocator_caching_size=4 << 30)
    success = True
    for _ in range(num_alignments):
        query, target = _noisy_pair(rng, seq_len)
        status = batch.add_alignment(query, target)
        if status != 0:
            success &= False
    batch.align_all()
    assert success is should_succeed

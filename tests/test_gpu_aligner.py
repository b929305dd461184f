"""GPU parity tests of the banded Myers aligner (through the C-ABI) vs the CPU oracle and the known answers."""
import random

import pytest

import oracle_aligner as A
from test_oracle_aligner import KNOWN, _mutate

pytestmark = pytest.mark.gpu


def run(pairs, max_bandwidth=1024):
    from genomeworks_amd import cudaaligner
    al = cudaaligner.CudaAlignerBatch(max_bandwidth=max_bandwidth, max_device_memory_allocator_caching_size=4 << 30)
    for q, t in pairs:
        assert al.add_alignment(q, t) == 0
    al.align_all()
    return al.get_alignments()


def test_known_cigars_banded():
    # Test_AlignerGlobal.cpp:79-148 (MyersBanded, max_bandwidth 1024) + python binding vectors
    res = run([(q, t) for q, t, _, _ in KNOWN])
    for r, (q, t, cigar, dist) in zip(res, KNOWN):
        assert r.status == 0 and r.is_optimal
        assert r.cigar == cigar
        if dist is not None:
            assert r.edit_distance == dist


def test_known_cigars_default_factory():
    # default create_aligner(max_q, max_t, max_alignments, ...) path: same expected CIGARs (Test_AlignerGlobal.cpp:240-340)
    from genomeworks_amd import cudaaligner
    al = cudaaligner.CudaAlignerBatch(64, 64, 100, max_device_memory_allocator_caching_size=1 << 30)
    for q, t, _, _ in KNOWN:
        assert al.add_alignment(q, t) == 0
    assert al.add_alignment("A" * 65, "A") == cudaaligner.exceeded_max_length
    al.align_all()
    res = al.get_alignments()
    for r, (q, t, cigar, dist) in zip(res, KNOWN):
        assert r.status == 0 and r.cigar == cigar
        assert len(r.alignment) == sum(int(x) for x in __import__("re").findall(r"(\d+)[MID]", cigar))


def test_approximate_banded_exact_cigars():
    # Test_ApproximateBandedMyers.cpp:72-120
    res = run([("AACCGGTTAACCGGTTAACCGGTTTT", "AACCGGTTAAAACCCCGGGGGTTAAACGGTT"),
               ("AACCGGTTAACCGGTTAACCGGTTT", "AACCGGTTAAAACCCCGGGGGTTAACCGGTT")], max_bandwidth=7)
    assert (res[0].cigar, res[0].is_optimal) == ("10M2I2M2I7M3I5M2D", False)
    assert (res[1].cigar, res[1].is_optimal) == ("10M2I2M2I3M2I3M1I6M1D", False)


def test_rejected_pair_stays_uninitialized():
    res = run([("ACGT" * 10, "ACGT" * 30), ("ACGT", "ACGT")], max_bandwidth=8)
    assert res[0].status == 1 and res[0].cigar == ""
    assert res[1].status == 0 and res[1].cigar == "4M"


@pytest.mark.parametrize("max_bw", [1024, 64, 2048, 31])
def test_random_pairs_bit_exact_vs_oracle(max_bw):
    rng = random.Random(1234 + max_bw)
    pairs = []
    for _ in range(300):
        n = rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 100, 150, 400, 1000, 1500])
        q = "".join(rng.choice("ACGT") for _ in range(n))
        t = _mutate(rng, q, rng.choice([0, 1, 2, 5, n // 20 + 1, n // 8 + 1])) or "A"
        pairs.append((q, t))
    res = run(pairs, max_bandwidth=max_bw)
    for (q, t), r in zip(pairs, res):
        ref = A.align(q, t, max_bw)
        assert (r.status == 0) == (ref["status"] == 0)
        if ref["status"] == 0:
            assert r.cigar_extended == ref["cigar_extended"], (q, t)
            assert r.cigar == ref["cigar"]
            assert r.is_optimal == ref["optimal"]
            assert r.edit_distance == ref["edit_distance"]


def test_config2_sample_and_cell_counts():
    # BASELINE config 2 generator: 1000-bp query, target = generate_random_sequence(query, rng, 33, 33, 33), minstd_rand(1)
    from genomeworks_amd import cudaaligner, synthetic
    pairs = synthetic.generate_pairs(1, 256, 1000, 33, 33, 33)
    al = cudaaligner.CudaAlignerBatch(max_bandwidth=1024, max_device_memory_allocator_caching_size=8 << 30)
    for q, t in pairs:
        assert al.add_alignment(q, t) == 0
    al.align_all()
    cells = al.band_cells()
    res = al.get_alignments()
    ref_cells = 0
    for (q, t), r in zip(pairs, res):
        ref = A.align(q, t, 1024)
        ref_cells += ref["cells"]
        assert r.cigar_extended == ref["cigar_extended"] and r.is_optimal
    assert cells == ref_cells


@pytest.mark.parametrize("max_len", [700, 70])
def test_default_aligner_bit_exact_vs_hirschberg_oracle(max_len):
    """create_aligner(max_query, max_target, n): Hirschberg + Myers. Every alignment state equals the oracle's, for
    lengths on both sides of the full-Myers leaf threshold (63) and of the word size; the constructor's
    max_query_length takes part (it bounds the leaf matrix), hence the two values."""
    import random
    from genomeworks_amd import cudaaligner
    rng = random.Random(99)
    pairs = []
    for k in range(160):
        n = rng.choice([1, 2, 3, 17, 31, 32, 33, 61, 62, 63, 64, 65, 96, 127, 128, 129, 300, 690])
        n = min(n, max_len - 2)
        q = "".join(rng.choice("ACGT") for _ in range(n))
        if k % 3 == 0:
            t = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, min(max_len - 2, n + n // 2 + 2))))
        else:
            t = list(q)
            for _ in range(max(1, n // 7)):
                op, p = rng.random(), rng.randrange(max(1, len(t)))
                if op < 0.4 and t:
                    t[p] = rng.choice("ACGT")
                elif op < 0.7 and len(t) < max_len - 2:
                    t.insert(p, rng.choice("ACGT"))
                elif t:
                    del t[p]
            t = "".join(t)
        pairs.append((q, t))
    al = cudaaligner.CudaAlignerBatch(max_len, max_len, len(pairs), max_device_memory_allocator_caching_size=4 << 30)
    for q, t in pairs:
        assert al.add_alignment(q, t) == 0
    al.align_all()
    res = al.get_alignments()
    for r, (q, t) in zip(res, pairs):
        ref = A.hirschberg(q, t, max_len)
        assert r.status == 0
        assert list(r.alignment) == ref["states"], (q, t)
        assert r.cigar == ref["cigar"]

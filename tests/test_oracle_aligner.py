"""Pins the banded-Myers oracle to the reference's known answers and to the reference's own CPU code (oracle/_ref)."""
import random

import pytest

import oracle_aligner as A

# Test_AlignerGlobal.cpp:79-148 (query, target, cigar, edit distance); python bindings test_cudaaligner_bindings.py:27-33
KNOWN = [
    ("AAAA", "TTAT", "4M", 3),
    ("ATAAAAAAAA", "AAAAAAAAA", "1M1D8M", 1),
    ("AAAAAAAAA", "ATAAAAAAAA", "1M1I8M", 1),
    ("ACTGA", "GCTAG", "3M1D1M1I", 3),
    ("ACTG", "ACTG", "4M", 0),
    ("A", "T", "1M", 1),
    ("", "GCTAGGCATCGATCGATCAGCTAGCATCGATCGACTACGACTACGT", "46I", 46),
    ("GCTAGGCATCGATCGATCAGCTAGCATCGATCGACTACGACTACGT", "", "46D", 46),
    ("", "", "", 0),
    ("AAAAAAA", "TTTTTTT", "7M", 7),
    ("AAATC", "TACGTTTT", "3M1I2M2I", None),
    ("TACGTA", "ACATAC", "1D5M1I", None),
    ("TGCA", "ATACGCT", "1I1M2I3M", None),
]


@pytest.mark.parametrize("q,t,cigar,dist", KNOWN)
def test_known_cigars(q, t, cigar, dist):
    r = A.align(q, t, 1024)
    assert r["status"] == 0 and r["optimal"]
    assert r["cigar"] == cigar
    if dist is not None:
        assert r["edit_distance"] == dist


def test_approximate_banded_exact_cigars():
    # Test_ApproximateBandedMyers.cpp:72-120: max_bw = 7, both flagged non-optimal
    r = A.align("AACCGGTTAACCGGTTAACCGGTTTT", "AACCGGTTAAAACCCCGGGGGTTAAACGGTT", 7)
    assert (r["cigar"], r["optimal"]) == ("10M2I2M2I7M3I5M2D", False)
    r = A.align("AACCGGTTAACCGGTTAACCGGTTT", "AACCGGTTAAAACCCCGGGGGTTAACCGGTT", 7)
    assert (r["cigar"], r["optimal"]) == ("10M2I2M2I3M2I3M1I6M1D", False)


def test_rejected_when_band_cannot_span_length_difference():
    # myers_gpu.cu:903-911: max_bw - 1 < |t - q| -> no result
    r = A.align("ACGT" * 10, "ACGT" * 30, 8)
    assert r["status"] == 1 and r["runs"] == []


def _mutate(rng, s, n):
    s = list(s)
    for _ in range(n):
        k = rng.random()
        if k < 0.4 and s:
            s[rng.randrange(len(s))] = rng.choice("ACGT")
        elif k < 0.7:
            s.insert(rng.randrange(len(s) + 1), rng.choice("ACGT"))
        elif s:
            del s[rng.randrange(len(s))]
    return "".join(s)


def _consistent(q, t, runs):
    """The run list must spell a valid global alignment of q and t."""
    i = j = 0
    for o, c in runs:
        for _ in range(c):
            if o in (0, 1):
                assert (q[i] == t[j]) == (o == 0)
                i += 1
                j += 1
            elif o == 2:
                j += 1
            else:
                i += 1
    assert i == len(q) and j == len(t)


@pytest.mark.parametrize("seed", range(12))
def test_optimal_results_match_reference_cpu_edit_distance(seed):
    # Test_MyersAlgorithm.cu:142-177 style: Myers == naive NW; here the oracle vs the reference's own CPU code
    R = A.ref()
    if R is None:
        pytest.skip("oracle/_ref/libref_aligner.so not built (no /root/reference here)")
    rng = random.Random(seed)
    n = rng.choice([1, 31, 32, 33, 64, 65, 200, 700, 1500])
    q = "".join(rng.choice("ACGT") for _ in range(n))
    t = _mutate(rng, q, rng.choice([0, 1, 3, n // 20 + 1, n // 5 + 1]))
    if not t:
        t = "A"
    for mbw in (1024, 2048, 64):
        r = A.align(q, t, mbw)
        if r["status"] != 0:
            continue
        _consistent(q, t, r["runs"])
        # ground truth = the reference's naive NW matrix (its tests' own yardstick, Test_MyersAlgorithm.cu:142-177).
        # (myers_cpu.hpp is dead code in the reference and mis-handles 1-character queries: "TT" vs "T" -> 0.)
        true = R.ref_nw_edit_distance(t.encode(), len(t), q.encode(), len(q))
        if len(q) > 1:
            assert true == R.ref_myers_edit_distance(t.encode(), len(t), q.encode(), len(q))
        if r["optimal"]:
            assert r["edit_distance"] == true
        else:
            assert r["edit_distance"] >= true


def test_monotone_over_bandwidths():
    # Test_ApproximateBandedMyers.cpp:122-170: edit distance is monotone non-increasing in max_bandwidth and
    # becomes optimal once the band is wide enough
    rng = random.Random(99)
    q = "".join(rng.choice("ACGT") for _ in range(600))
    t = _mutate(rng, q, 60)
    last = None
    for mbw in (16, 32, 48, 64, 96, 128, 256, 512, 1024, 2048):
        r = A.align(q, t, mbw)
        if r["status"] != 0:
            continue
        _consistent(q, t, r["runs"])
        if last is not None:
            assert r["edit_distance"] <= last
        last = r["edit_distance"]
    R = A.ref()
    final = A.align(q, t, 2048)
    assert final["optimal"]
    if R is not None:
        assert final["edit_distance"] == R.ref_nw_edit_distance(t.encode(), len(t), q.encode(), len(q))

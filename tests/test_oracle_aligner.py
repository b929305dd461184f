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


# ---- default aligner (Hirschberg + Myers) restatement, oracle/hirschberg_oracle.c ----
HIRSCHBERG_KNOWN = [  # cudaaligner/tests/Test_AlignerGlobal.cpp:79-108,145-146 (HirschbergMyers uses the same table)
    ("AAAA", "TTAT", "4M"), ("ATAAAAAAAA", "AAAAAAAAA", "1M1D8M"), ("AAAAAAAAA", "ATAAAAAAAA", "1M1I8M"),
    ("ACTGA", "GCTAG", "3M1D1M1I"), ("ACTG", "ACTG", "4M"), ("A", "T", "1M"),
    ("", "GACTCTCCCCCTCCCCTTTAAATATATAAAAATGGGGTGTAGCTAG", "46I"), ("GACTCTCCCCCTCCCCTTTAAATATATAAAAATGGGGTGTAGCTAG", "", "46D"),
    ("", "", "")]


@pytest.mark.parametrize("q,t,cigar", HIRSCHBERG_KNOWN)
def test_hirschberg_known_cigars(q, t, cigar):
    assert A.hirschberg(q, t)["cigar"] == cigar


def _mutate(rng, q, n_edits):
    t = list(q)
    for _ in range(n_edits):
        op, p = rng.random(), rng.randrange(max(1, len(t)))
        if op < 0.4 and t:
            t[p] = rng.choice("ACGT")
        elif op < 0.7:
            t.insert(p, rng.choice("ACGT"))
        elif t:
            del t[p]
    return "".join(t)


def test_hirschberg_paths_are_valid_and_optimal():
    import random
    rng = random.Random(11)
    ref = A.ref()
    for k in range(200):
        n = rng.choice([2, 5, 31, 32, 33, 62, 63, 64, 65, 127, 200, 400])
        q = "".join(rng.choice("ACGT") for _ in range(n))
        t = _mutate(rng, q, max(1, n // 8)) if k % 2 else "".join(rng.choice("ACGT") for _ in range(max(0, n + rng.randint(-n // 3, n // 3))))
        r = A.hirschberg(q, t, max(len(q), len(t)) + 1)
        qi = ti = 0
        for s in r["states"]:
            if s in (0, 1):
                assert (q[qi] == t[ti]) == (s == 0)
                qi += 1
                ti += 1
            elif s == 2:
                ti += 1
            else:
                qi += 1
        assert (qi, ti) == (len(q), len(t))
        if ref is not None and q and t:
            assert r["edit_distance"] == ref.ref_nw_edit_distance(q.encode(), len(q), t.encode(), len(t))


def test_hirschberg_range_stack_is_bounded():
    # 64 entries suffice for any realistic input: a 60 kbp pair recurses ~10 levels deep with at most depth + 1 live ranges
    import random
    rng = random.Random(5)
    q = "".join(rng.choice("ACGT") for _ in range(3000))
    r = A.hirschberg(q, _mutate(rng, q, 150), 4096)
    assert r["status"] == 0 and len(r["states"]) >= 3000

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


# ---- AlignerGlobalUkkonen / AlignerGlobalMyers restatements (oracle/global_oracle.c) ----
# Test_AlignerGlobal.cpp:79-108 runs this table for the Ukkonen and the Myers class too (:138-139); the empty-sequence
# cases (:145-153) only for Myers ("Ukkonen cannot handle these cases")
GLOBAL_KNOWN = [
    ("AAAA", "TTAT", "4M", 3),
    ("ATAAAAAAAA", "AAAAAAAAA", "1M1D8M", 1),
    ("AAAAAAAAA", "ATAAAAAAAA", "1M1I8M", 1),
    ("ACTGA", "GCTAG", "3M1D1M1I", 3),
    ("ACTG", "ACTG", "4M", 0),
    ("A", "T", "1M", 1),
]
EMPTY_KNOWN = [
    ("", "GACTCTCCCCCTCCCCTTTAAATATATAAAAATGGGGTGTAGCTAG", "46I", 46),
    ("GACTCTCCCCCTCCCCTTTAAATATATAAAAATGGGGTGTAGCTAG", "", "46D", 46),
    ("", "", "", 0),
]


@pytest.mark.parametrize("q,t,cigar,dist", GLOBAL_KNOWN)
def test_ukkonen_known_cigars(q, t, cigar, dist):
    r = A.ukkonen(q, t, 100)
    assert (r["status"], r["cigar"], r["edit_distance"]) == (0, cigar, dist)


@pytest.mark.parametrize("q,t,cigar,dist", GLOBAL_KNOWN + EMPTY_KNOWN)
def test_myers_full_known_cigars(q, t, cigar, dist):
    r = A.myers_full(q, t)
    assert (r["status"], r["cigar"], r["edit_distance"]) == (0, cigar, dist)


def _pairs(seed, n, lo, hi, div=5):
    rng = random.Random(seed)
    for _ in range(n):
        L = rng.randrange(lo, hi)
        q = "".join(rng.choice("ACGT") for _ in range(L))
        t = _mutate(rng, q, rng.randrange(0, max(2, L // div)))
        if t:
            yield q, t


@pytest.mark.parametrize("p,lo,hi", [(0, 1, 300), (1, 1, 300), (3, 1, 300), (10, 30, 200), (30, 80, 300), (100, 150, 500)])
def test_ukkonen_matches_reference_cpu_ukkonen(p, lo, hi):
    """Against the reference's own ukkonen_cpu() (oracle/_ref): the function Test_NeedlemanWunschImplementation.cpp:286-293
    compares the GPU path with. Small p makes most paths run along the band edges (clipped, non-optimal results).
    ukkonen_cpu wants target >= query; the class swaps the roles (and insertion <-> deletion) otherwise."""
    if A.ref() is None:
        pytest.skip("oracle/_ref not built")
    swap = {0: 0, 1: 1, 2: 3, 3: 2}
    clipped = 0
    for q, t in _pairs(100 + p, 120, lo, hi):
        mine = A.ukkonen(q, t, p)["states"]
        if len(t) >= len(q):
            ref = A.ref_ukkonen_cpu(q, t, p)
        else:
            ref = [swap[x] for x in A.ref_ukkonen_cpu(t, q, p)]
        assert mine == ref
        clipped += sum(1 for s in mine if s != 0) != A.ref().ref_nw_edit_distance(t.encode(), len(t), q.encode(), len(q))
    if p <= 1:
        assert clipped > 10  # the band-clipped regime is really exercised


def test_myers_full_is_optimal_and_equals_wide_banded_myers():
    """The full-matrix class and the banded class share the backtrace rule; with a band that covers the whole matrix
    the banded oracle (pinned separately above) must give the identical path."""
    for q, t in _pairs(77, 150, 1, 400):
        r = A.myers_full(q, t)
        b = A.align(q, t, 1 << 20)
        states = [o for o, c in b["runs"] for _ in range(c)]
        assert r["states"] == states
        if A.ref() is not None:
            assert r["edit_distance"] == A.ref().ref_nw_edit_distance(t.encode(), len(t), q.encode(), len(q))
        assert sum(1 for s in r["states"] if s != 2) == len(q) and sum(1 for s in r["states"] if s != 3) == len(t)


def test_backtrace_step_from_delta_bits_equals_the_three_cell_reads():
    """The kernels' backtrace step reads `left` as the reference does and derives `diag` and `above` from vertical-delta
    bits (gwhip_myers.hip backtrace_banded::fetch3). The oracle evaluates that model next to the reference's three reads in
    every step of every banded backtrace it runs; over narrow and wide bands (one word .. many words, word boundaries inside
    the walk), approximate results, and very unequal lengths the two must never differ."""
    import random
    rng = random.Random(31)
    before = A.delta_identity_mismatches()
    steps = 0
    for k in range(300):
        n = rng.choice([1, 31, 32, 33, 64, 65, 150, 400, 1000, 1500])
        q = "".join(rng.choice("ACGT") for _ in range(n))
        t = list(q)
        for _ in range(rng.choice([0, 1, n // 30 + 1, n // 8 + 1, n // 3 + 1])):
            op, p = rng.random(), rng.randrange(max(1, len(t)))
            if op < 0.4 and t:
                t[p] = rng.choice("ACGT")
            elif op < 0.7:
                t.insert(p, rng.choice("ACGT"))
            elif len(t) > 1:
                del t[p]
        t = "".join(t) or "A"
        for max_bw in (7, 63, 200, 1024, 4096):
            r = A.align(q, t, max_bw)
            steps += len(r.get("cigar_extended", "")) if r.get("status", 1) == 0 else 0
    assert steps > 100000
    assert A.delta_identity_mismatches() == before

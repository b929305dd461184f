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


def test_bases_outside_acgt_and_odd_offsets_equal_the_oracle():
    """The batch goes to the device two bases per byte (include/gwhip.h, gwhip_unpack_bases): a query base keeps what the
    kernels can tell apart ('A', 'C', 'T', 'G' or anything else, myers_gpu.cu:196-208), a target base its pattern index
    (c >> 1) & 3 (myers_gpu.cu:210-241). Lower case, 'N', IUPAC codes and arbitrary bytes in both sequences, odd lengths (so
    queries and targets start on either half of a byte), in a batch large enough for the chunked upload when forced:
    CIGARs, flags and distances equal the oracle, which follows the reference on the unpacked characters."""
    import os
    rng = random.Random(4242)
    alphabet = "ACGTACGTACGTNnacgtRYKMSWBDHV-*xU@~ "
    pairs = []
    for k in range(700):
        n = rng.choice([1, 2, 3, 15, 16, 17, 31, 33, 64, 65, 99, 150, 151, 301, 1000])
        q = "".join(rng.choice(alphabet) for _ in range(n))
        t = "".join(c if rng.random() > 0.06 else rng.choice(alphabet) for c in q)
        if rng.random() < 0.3:
            cut = rng.randrange(len(t))
            t = t[:cut] + t[cut + 1:]
        pairs.append((q, t or "n"))
    ref = [A.align(q, t, 256) for q, t in pairs]
    for chunks in (None, "5"):
        if chunks is None:
            os.environ.pop("GW_ALIGNER_CHUNKS", None)
        else:
            os.environ["GW_ALIGNER_CHUNKS"] = chunks
        try:
            res = run(pairs, max_bandwidth=256)
        finally:
            os.environ.pop("GW_ALIGNER_CHUNKS", None)
        for (q, t), r, e in zip(pairs, res, ref):
            assert (r.status == 0) == (e["status"] == 0), (q, t)
            if e["status"] == 0:
                assert r.cigar_extended == e["cigar_extended"], (q, t)
                assert (r.is_optimal, r.edit_distance) == (e["optimal"], e["edit_distance"])
        # the Alignment objects still hold the characters the caller passed
        assert [(r.query, r.target) for r in res[:50]] == pairs[:50]


def test_chunked_batch_runs_reach_the_host_through_the_mirror_or_the_copy(monkeypatch):
    """A chunked batch's runs are written to a pinned mirror by the chunks' kernels (gwhip_myers_args::results_host);
    sync_alignments() copies for itself when the batch has more runs than the mirror holds. Both ways, and one chunk, give the
    same alignments as the oracle: pairs with a run per base (every other base substituted) next to identical pairs, so that
    the capacity cut falls inside a chunk, and a batch that fits."""
    rng = random.Random(99)
    swap = {"A": "C", "C": "G", "G": "T", "T": "A"}
    pairs = []
    for k in range(640):
        n = rng.choice([40, 97, 150, 151, 300])
        q = "".join(rng.choice("ACGT") for _ in range(n))
        if k % 3 == 0:
            t = "".join(swap[c] if i % 2 else c for i, c in enumerate(q))
        elif k % 3 == 1:
            t = q
        else:
            t = _mutate(rng, q, 3) or "A"
        pairs.append((q, t))
    ref = [A.align(q, t, 512) for q, t in pairs]
    want = [(e["status"], e["cigar_extended"], e["optimal"], e["edit_distance"]) for e in ref]
    assert sum(len(e["runs"]) for e in ref) > 16 * len(pairs)  # more than the default capacity of a small batch
    for chunks, mirror_runs in (("1", None), ("5", None), ("5", "1000"), ("5", "10000000"), ("7", "3")):
        monkeypatch.setenv("GW_ALIGNER_CHUNKS", chunks)
        if mirror_runs is None:
            monkeypatch.delenv("GW_ALIGNER_MIRROR_RUNS", raising=False)
        else:
            monkeypatch.setenv("GW_ALIGNER_MIRROR_RUNS", mirror_runs)
        got = [(r.status, r.cigar_extended, r.is_optimal, r.edit_distance) for r in run(pairs, max_bandwidth=512)]
        assert got == want, (chunks, mirror_runs)


def test_group_kernel_equals_one_lane_kernel(monkeypatch):
    """A/B of the two banded Myers kernels: eight lanes per pair (carry-lookahead over the lanes; picked for small batches
    of long pairs) against one lane per pair (GWHIP_MYERS_GROUP=0), and the group kernel forced on short queries too
    (=1): same CIGARs, optimality flags and edit distances, including bands wider than the group (more than 8 words:
    its one-lane fallback) and pairs the band rejects."""
    rng = random.Random(77)
    pairs = [("A" * 300, "C" * 300), ("ACGT" * 100, "ACGT" * 100)]
    for _ in range(260):
        n = rng.choice([1, 33, 64, 255, 256, 257, 300, 512, 777, 1000, 1024, 1500])
        q = "".join(rng.choice("ACGT") for _ in range(n))
        t = _mutate(rng, q, rng.choice([0, 1, 3, n // 25 + 1, n // 6 + 1, n // 3 + 1])) or "A"
        pairs.append((q, t))
    for max_bw in (2048, 200):
        got = {}
        for mode in ("0", "1", None):
            if mode is None:
                monkeypatch.delenv("GWHIP_MYERS_GROUP", raising=False)
            else:
                monkeypatch.setenv("GWHIP_MYERS_GROUP", mode)
            got[mode] = [(r.status, r.cigar_extended, r.is_optimal, r.edit_distance) for r in run(pairs, max_bandwidth=max_bw)]
        assert got["0"] == got["1"]
        assert got["0"] == got[None]
        # six lanes per pair (ten pairs per wavefront; groups cross the 16-lane rows, four lanes of a wavefront idle)
        monkeypatch.setenv("GWHIP_MYERS_GROUP", "1")
        monkeypatch.setenv("GWHIP_MYERS_GROUP_LANES", "6")
        six = [(r.status, r.cigar_extended, r.is_optimal, r.edit_distance) for r in run(pairs, max_bandwidth=max_bw)]
        monkeypatch.delenv("GWHIP_MYERS_GROUP_LANES", raising=False)
        assert got["0"] == six


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


def test_default_aligner_kernels_agree(monkeypatch):
    """The default aligner's two kernels -- one wavefront per pair (default: the words of a query part across the lanes,
    carry lookahead over the lanes, parts beyond 64 words in chunks) and one lane per pair (GWHIP_HIRSCHBERG_WAVE=0) -- must
    give identical alignments, also for queries of several thousand bases (two and more chunks per part, carries and
    deltas handed from chunk to chunk) and for very unequal lengths; short ones are checked against the oracle as well."""
    import random
    from genomeworks_amd import cudaaligner
    rng = random.Random(4242)
    pairs = []
    for k in range(48):
        n = rng.choice([70, 500, 1000, 2040, 2049, 2500, 4100, 4500, 6000])
        q = "".join(rng.choice("ACGT") for _ in range(n))
        t = list(q)
        for _ in range(max(1, n // rng.choice([8, 15, 40]))):
            op, p = rng.random(), rng.randrange(max(1, len(t)))
            if op < 0.4 and t:
                t[p] = rng.choice("ACGT")
            elif op < 0.7:
                t.insert(p, rng.choice("ACGT"))
            elif t:
                del t[p]
        if k % 7 == 0:
            t = t[: len(t) // 3]                      # target much shorter than the query
        if k % 11 == 0:
            q = q[: max(2, len(q) // 4)]              # query much shorter than the target
        pairs.append((q, "".join(t)))
    max_len = 8192
    out = {}
    for name, flag in (("wave", None), ("lane", "0")):
        if flag is None:
            monkeypatch.delenv("GWHIP_HIRSCHBERG_WAVE", raising=False)
        else:
            monkeypatch.setenv("GWHIP_HIRSCHBERG_WAVE", flag)
        al = cudaaligner.CudaAlignerBatch(max_len, max_len, len(pairs), max_device_memory_allocator_caching_size=8 << 30)
        for q, t in pairs:
            assert al.add_alignment(q, t) == 0
        al.align_all()
        out[name] = [(r.status, list(r.alignment)) for r in al.get_alignments()]
    assert out["wave"] == out["lane"]
    for (st, states), (q, t) in zip(out["wave"], pairs):
        assert st == 0
        if len(q) <= 1000:
            assert states == A.hirschberg(q, t, max_len)["states"]


def test_default_aligner_span_path_equals_the_depth_first_kernel(monkeypatch):
    """Long single pairs (the reference's BM_SingleAlignment shapes, cudaaligner/benchmarks/main.cpp:39-67): the span path
    -- the top of Hirschberg's tree grown level by level across blocks, one wavefront per part below -- against the
    depth-first wavefront kernel alone (GWHIP_HIRSCHBERG_SPAN=0): identical states for pairs of 2.1 k to 60 k bases, balanced
    and very unbalanced, alone and in small batches next to short pairs; the oracle on the ones it finishes quickly."""
    import random
    from genomeworks_amd import cudaaligner
    rng = random.Random(777)

    def pair(n, div, cut_t=None, cut_q=None):
        q = "".join(rng.choice("ACGT") for _ in range(n))
        t = list(q)
        for _ in range(max(1, n // div)):
            op, p = rng.random(), rng.randrange(max(1, len(t)))
            if op < 0.4 and t:
                t[p] = rng.choice("ACGT")
            elif op < 0.7:
                t.insert(p, rng.choice("ACGT"))
            elif t:
                del t[p]
        t = "".join(t)
        if cut_t:
            t = t[: len(t) // cut_t]
        if cut_q:
            q = q[: max(2, len(q) // cut_q)]
        return q, t

    batches = [
        (65536, [pair(60000, 12)]),
        (32768, [pair(20000, 8), pair(2100, 10), pair(700, 9), pair(20000, 30, cut_t=5)]),
        (16384, [pair(9000, 10), pair(16000, 15, cut_q=3), pair(4097, 7), pair(10, 3), pair(12000, 20, cut_t=50)]),
        (6400, [pair(5999, 9), pair(3000, 11), pair(2049, 6)]),
    ]
    for max_len, pairs in batches:
        out = {}
        for name, flag in (("span", None), ("depth_first", "0")):
            if flag is None:
                monkeypatch.delenv("GWHIP_HIRSCHBERG_SPAN", raising=False)
            else:
                monkeypatch.setenv("GWHIP_HIRSCHBERG_SPAN", flag)
            al = cudaaligner.CudaAlignerBatch(max_len, max_len, len(pairs), max_device_memory_allocator_caching_size=16 << 30)
            for q, t in pairs:
                assert al.add_alignment(q, t) == 0
            al.align_all()
            out[name] = [(r.status, list(r.alignment)) for r in al.get_alignments()]
        assert [st for st, _ in out["span"]] == [0] * len(pairs)
        for k, (a, b) in enumerate(zip(out["span"], out["depth_first"])):
            assert a == b, (max_len, k)
        for (st, states), (q, t) in zip(out["span"], pairs):
            assert states == A.hirschberg(q, t, max_len)["states"]


def test_default_aligner_benchmark_shapes_equal_the_golden():
    """Every shape bench.py publishes for the default aligner (the reference's BM_SingleAlignment pairs of 100 .. 100 000
    bases, BM_SingleBatchAlignment 1024 x 2048 bases, cudaaligner/benchmarks/main.cpp:39-143, and 2 000 x 1 kbp) against the
    committed oracle goldens (tests/golden/make_default_aligner_goldens.py: sha256 over every pair's status and state sequence;
    edit-distance sums) -- the 10 kbp and 100 kbp single pairs run through the span path."""
    import golden_io as G
    from genomeworks_amd import cudaaligner
    gold = G.default_aligner_goldens()
    for n, size in G.aligner_gen.SHAPES:
        pairs = G.aligner_gen.shape_pairs(n, size)
        al = cudaaligner.CudaAlignerBatch(size, size, n, max_device_memory_allocator_caching_size=16 << 30)
        for q, t in pairs:
            assert al.add_alignment(q, t) == 0
        al.align_all()
        res = al.get_alignments()
        g = gold["%dx%d" % (n, size)]
        assert len(res) == n and all(r.status == 0 for r in res)
        assert sum(sum(1 for x in r.alignment if x != 0) for r in res) == g["edit_distance_sum"], (n, size)
        assert sum(len(r.alignment) for r in res) == g["states_total"], (n, size)
        assert G.aligner_gen.digest(G.aligner_gen.pair_record(r.status, r.alignment) for r in res) == g["states_sha256"], (n, size)


def test_default_aligner_level_by_level_kernel_agrees(monkeypatch):
    """Aligners for queries of up to 2 048 bases grow Hirschberg's tree a level at a time (all parts of a level side by side
    in the wavefront, default) -- the alignments must equal the depth-first kernel's (GWHIP_HIRSCHBERG_LEVELS=0) and the
    oracle's: lengths around the word and leaf thresholds, single characters, very unequal lengths (targets beyond the
    kernel's LDS rows stay with the depth-first kernel inside the same batch), identical and unrelated sequences."""
    import random
    from genomeworks_amd import cudaaligner
    rng = random.Random(777)
    pairs = []
    for n in [1, 2, 3, 31, 32, 33, 62, 63, 64, 65, 125, 126, 127, 128, 129, 250, 500, 999, 1000, 1023, 1024, 1025, 2000, 2047, 2048]:
        for div in (3, 10, 40):
            q = "".join(rng.choice("ACGT") for _ in range(n))
            t = list(q)
            for _ in range(max(1, n // div)):
                op, p = rng.random(), rng.randrange(max(1, len(t)))
                if op < 0.4 and t:
                    t[p] = rng.choice("ACGT")
                elif op < 0.7 and len(t) < 2048:
                    t.insert(p, rng.choice("ACGT"))
                elif len(t) > 1:
                    del t[p]
            pairs.append((q, "".join(t)[:2048]))
    for n in (5, 40, 300, 1500):
        q = "".join(rng.choice("ACGT") for _ in range(n))
        pairs.append((q, q))                                                      # identical
        pairs.append((q, "".join(rng.choice("ACGT") for _ in range(n))))          # unrelated
        pairs.append((q, "".join(rng.choice("ACGT") for _ in range(2048))))       # target much longer than the query
        pairs.append(("".join(rng.choice("ACGT") for _ in range(2048)), q))       # query much longer than the target
        pairs.append((q, q[: max(1, n // 20)]))
    max_len = 2048
    out = {}
    for name, flag in (("levels", None), ("depth_first", "0")):
        if flag is None:
            monkeypatch.delenv("GWHIP_HIRSCHBERG_LEVELS", raising=False)
        else:
            monkeypatch.setenv("GWHIP_HIRSCHBERG_LEVELS", flag)
        al = cudaaligner.CudaAlignerBatch(max_len, max_len, len(pairs), max_device_memory_allocator_caching_size=8 << 30)
        for q, t in pairs:
            assert al.add_alignment(q, t) == 0
        al.align_all()
        out[name] = [(r.status, list(r.alignment)) for r in al.get_alignments()]
    bad = [i for i, (x, y) in enumerate(zip(out["levels"], out["depth_first"])) if x != y]
    assert not bad, [(i, len(pairs[i][0]), len(pairs[i][1])) for i in bad[:10]]
    for (st, states), (q, t) in zip(out["levels"], pairs):
        assert st == 0
        assert states == A.hirschberg(q, t, max_len)["states"]
    # an aligner whose targets may be much longer than its queries: the kernel's LDS rows are sized from max_query_length, so
    # pairs with longer targets are left to the depth-first kernel launched behind it -- inside the same batch
    mixed = []
    for k in range(40):
        n = rng.choice([20, 100, 300, 500])
        q = "".join(rng.choice("ACGT") for _ in range(n))
        tl = rng.choice([n, n + 30, 600, 700, 1500, 4000])
        t = (q + "".join(rng.choice("ACGT") for _ in range(4000)))[:tl]
        mixed.append((q, t))
    got = {}
    for name, flag in (("levels", None), ("depth_first", "0")):
        if flag is None:
            monkeypatch.delenv("GWHIP_HIRSCHBERG_LEVELS", raising=False)
        else:
            monkeypatch.setenv("GWHIP_HIRSCHBERG_LEVELS", flag)
        al = cudaaligner.CudaAlignerBatch(500, 4000, len(mixed), max_device_memory_allocator_caching_size=8 << 30)
        for q, t in mixed:
            assert al.add_alignment(q, t) == 0
        al.align_all()
        got[name] = [(r.status, list(r.alignment)) for r in al.get_alignments()]
    assert got["levels"] == got["depth_first"]
    for (st, states), (q, t) in zip(got["levels"], mixed):
        assert st == 0
        assert states == A.hirschberg(q, t, 500)["states"]


# ---- the non-default classes: AlignerGlobalUkkonen / AlignerGlobalMyers ----
def _run_algorithm(algorithm, pairs, max_len):
    from genomeworks_amd import cudaaligner
    al = cudaaligner.CudaAlignerBatch(max_len, max_len, len(pairs), algorithm=algorithm,
                                      max_device_memory_allocator_caching_size=6 << 30)
    for q, t in pairs:
        assert al.add_alignment(q, t) == 0, (len(q), len(t))
    al.align_all()
    return al.get_alignments()


@pytest.mark.parametrize("algorithm", ["ukkonen", "myers", "hirschberg_myers"])
def test_algorithm_classes_known_cigars(algorithm):
    # Test_AlignerGlobal.cpp:79-153: one table for every class; the empty-sequence cases not for Ukkonen
    from test_oracle_aligner import GLOBAL_KNOWN, EMPTY_KNOWN
    table = GLOBAL_KNOWN + ([] if algorithm == "ukkonen" else EMPTY_KNOWN)
    res = _run_algorithm(algorithm, [(q, t) for q, t, _, _ in table] * 4, 500)  # :110-124 repeats the table in one batch
    for r, (q, t, cigar, dist) in zip(res, table * 4):
        assert r.status == 0 and r.is_optimal
        assert (r.cigar, r.edit_distance) == (cigar, dist)


def _global_pairs(seed, n, max_len, max_diff):
    rng = random.Random(seed)
    pairs = []
    while len(pairs) < n:
        L = rng.choice([1, 2, 5, 31, 32, 33, 64, 65, 100, 199, 200, 201, 333, 640, max_len - 1])
        L = min(L, max_len - 1)
        q = "".join(rng.choice("ACGT") for _ in range(L))
        t = _mutate(rng, q, rng.randrange(0, max(2, L // 4)))[:max_len]
        if rng.random() < 0.3:
            q, t = t, q  # query longer than target: the Ukkonen class swaps roles and gap states
        if q and t and abs(len(q) - len(t)) <= max_diff:
            pairs.append((q, t))
    return pairs


def test_ukkonen_class_bit_exact_vs_oracle():
    """Every alignment state equals the restatement of ukkonen_gpu.cu (p = 100): in-band optimal paths, both
    orientations, bands wider than one wavefront pass (bw up to 100 + 35 slots per anti-diagonal)."""
    max_len = 700
    pairs = _global_pairs(7, 150, max_len, int(max_len * 0.1))
    res = _run_algorithm("ukkonen", pairs, max_len)
    for r, (q, t) in zip(res, pairs):
        ref = A.ukkonen(q, t, 100)
        assert r.status == 0 and r.is_optimal
        assert list(r.alignment) == ref["states"], (len(q), len(t))
        assert r.cigar == ref["cigar"]


def test_ukkonen_class_band_clipped_paths():
    """Pairs whose optimal path leaves the p = 100 band (a 150-base block moved from the front to the back): the class
    returns the band-restricted path, identical to the oracle's, including steps along the band edges."""
    rng = random.Random(3)
    pairs = []
    for _ in range(12):
        core = "".join(rng.choice("ACGT") for _ in range(rng.randrange(700, 900)))
        block = "".join(rng.choice("ACGT") for _ in range(rng.randrange(130, 180)))
        pairs.append((block + core, core + block))
        pairs.append((core + block, block + core[:len(core) - 20]))
    res = _run_algorithm("ukkonen", pairs, 1100)
    clipped = 0
    for r, (q, t) in zip(res, pairs):
        ref = A.ukkonen(q, t, 100)
        assert list(r.alignment) == ref["states"]
        clipped += ref["edit_distance"] != A.myers_full(q, t)["edit_distance"]
    assert clipped >= len(pairs) // 2


def test_ukkonen_class_rejects_large_length_difference():
    from genomeworks_amd import cudaaligner
    al = cudaaligner.CudaAlignerBatch(100, 100, 4, algorithm="ukkonen", max_device_memory_allocator_caching_size=1 << 30)
    # aligner_global_ukkonen.cpp:53-59: |q - t| > int(max_target_length * 0.1f)
    assert al.add_alignment("A" * 50, "A" * 61) == cudaaligner.exceeded_max_alignment_difference
    assert al.add_alignment("A" * 50, "A" * 60) == 0
    assert al.add_alignment("A" * 101, "A" * 100) == cudaaligner.exceeded_max_length
    al.align_all()
    assert al.get_alignments()[0].cigar == A.ukkonen("A" * 50, "A" * 60)["cigar"]


def test_myers_class_bit_exact_vs_oracle():
    max_len = 700
    pairs = _global_pairs(11, 150, max_len, max_len) + [("", "ACGT"), ("ACGT", ""), ("", "")]
    res = _run_algorithm("myers", pairs, max_len)
    for r, (q, t) in zip(res, pairs):
        ref = A.myers_full(q, t)
        assert r.status == 0 and r.is_optimal
        assert list(r.alignment) == ref["states"], (len(q), len(t))
        assert r.cigar == ref["cigar"]


def test_myers_class_bands_wider_than_a_wavefront_equal_the_oracle(monkeypatch):
    """Band attempts of more than 64 words (round 6: several words per lane of the wavefront that owns the pair,
    multi_attempt in gwhip_myers.hip; before: one lane). Long pairs whose first attempts fail -- unrelated sequences, pairs at
    30 % divergence, different lengths -- so that the doubling reaches bands of 2 .. 8 rounds of 64 words, horizontal stripes
    alone (band = query) and with the sliding stripe; against the full-matrix oracle, and against the one-lane stripes
    (GWHIP_MYERS_SKIP=8) on the same pairs."""
    from genomeworks_amd import cudaaligner
    rng = random.Random(66)
    rand = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    pairs = []
    q = rand(8192)
    pairs.append((q, _mutate(rng, q, 2400)[:8192]))           # 30 % divergence: attempts of 13, 26, 52, 103 words
    pairs.append((rand(9000), rand(12000)))                    # unrelated, |dlen| 3000: the band ends up covering the query
    pairs.append((rand(16000), rand(15000)))                   # unrelated: 500 words, eight rounds
    q = rand(5000)
    pairs.append((q, _mutate(rng, q, 2000)[:6000]))
    pairs.append((rand(4200), rand(4100)))
    max_len = 16384

    def run():
        al = cudaaligner.CudaAlignerBatch(max_len, max_len, len(pairs), algorithm="myers", max_device_memory_allocator_caching_size=64 << 30)
        for a, b in pairs:
            assert al.add_alignment(a, b) == 0
        al.align_all()
        return [(r.status, bool(r.is_optimal), list(r.alignment)) for r in al.get_alignments()]

    got = run()
    for (st, opt, states), (a, b) in zip(got, pairs):
        ref = A.myers_full(a, b)
        assert st == 0 and opt
        assert states == ref["states"], (len(a), len(b))
    monkeypatch.setenv("GWHIP_MYERS_SKIP", "8")
    assert run() == got


def test_aligner_matrix_cells_equal_the_golden():
    """The cells of the reference's aligner benchmark matrix that bench.py publishes (cudaaligner/benchmarks/main.cpp:69-168:
    AlignerGlobalUkkonen, AlignerGlobalMyers, AlignerGlobalMyersBanded, AlignerGlobalHirschbergMyers at 1024 x 2048 bases;
    Ukkonen and Hirschberg at 256 x 8192) against each class's committed oracle golden
    (tests/golden/make_aligner_matrix_goldens.py)."""
    import golden_io as G
    from genomeworks_amd import cudaaligner
    gold = G.aligner_matrix_goldens()
    assert all(G.matrix_gen.cell_key(*c) in gold for c in G.matrix_gen.CELLS + G.matrix_gen.CORNER_CELLS)
    # (round 6: + the corners of the reference's grid -- 1024 x 512, 32 x 32768 and 32 x 65536 for every class)
    for algorithm, n, size in G.matrix_gen.CELLS + G.matrix_gen.CORNER_CELLS:
        pairs = G.aligner_gen.shape_pairs(n, size)
        if algorithm == "myers_banded":
            al = cudaaligner.CudaAlignerBatch(max_bandwidth=G.matrix_gen.BANDED_MAX_BANDWIDTH, max_device_memory_allocator_caching_size=96 << 30)
        else:
            # (the full-matrix Myers class at 32 x 65 536: 2 048 band words x 65 537 columns x 12 B for each of the 64 slots of a
            # workspace region, whatever the number of pairs: 103 GB)
            al = cudaaligner.CudaAlignerBatch(size, size, n, algorithm=algorithm, max_device_memory_allocator_caching_size=(160 if size > 32768 else 96) << 30)
        for q, t in pairs:
            assert al.add_alignment(q, t) == 0
        al.align_all()
        res = al.get_alignments()
        g = gold[G.matrix_gen.cell_key(algorithm, n, size)]
        assert len(res) == n, (algorithm, n, size)
        if size <= 8192:
            assert all(r.status == 0 for r in res), (algorithm, n, size)
        assert sum(sum(1 for x in r.alignment if x != 0) for r in res) == g["edit_distance_sum"], (algorithm, n, size)
        assert G.aligner_gen.digest(G.aligner_gen.pair_record(r.status, r.alignment) for r in res) == g["states_sha256"], (algorithm, n, size)
        del res, al  # (before the next cell's aligner asks for its memory)


def test_hip_aligners_equal_the_reference_itself_on_the_simt_goldens():
    """tests/golden/reference_simt_alignments.json.gz holds what the REFERENCE's own cudaaligner library answered (its CUDA sources
    compiled from /root/reference and run on the CPU by the SIMT emulator of oracle/simt;
    tests/golden/make_reference_simt_alignments.py): the default aligner (Hirschberg + Myers) up to 5 kbp, the banded Myers aligner
    at five band widths with pairs the band rejects or only approximates, the Ukkonen and the full-matrix Myers classes. The HIP
    aligners give the same statuses, optimality flags and alignment states, pair by pair."""
    import gzip
    import importlib.util
    import json
    import os
    from genomeworks_amd import cudaaligner
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_reference_simt_alignments", os.path.join(here, "golden", "make_reference_simt_alignments.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    with gzip.open(os.path.join(here, "golden", "reference_simt_alignments.json.gz"), "rb") as f:
        rows = json.loads(f.read().decode())["batches"]
    assert len(rows) >= 12
    bad = []
    for k, row in enumerate(rows):
        b, ref = row["batch"], row["reference"]
        pairs = [tuple(p) for p in b["pairs"]]
        max_len = max(max(len(q), len(t)) for q, t in pairs)
        if b["kind"] == "banded":
            al = cudaaligner.CudaAlignerBatch(max_bandwidth=b["max_bandwidth"], max_device_memory_allocator_caching_size=2 << 30)
        elif b["kind"] == "default":
            al = cudaaligner.CudaAlignerBatch(max_len, max_len, len(pairs), max_device_memory_allocator_caching_size=2 << 30)
        else:
            al = cudaaligner.CudaAlignerBatch(max_len, max_len, len(pairs), algorithm=b["kind"], max_device_memory_allocator_caching_size=2 << 30)
        added = [al.add_alignment(q, t) for q, t in pairs]
        if added != [r["add_status"] for r in ref]:
            bad.append((k, "add_alignment", added))
            continue
        al.align_all()
        got = al.get_alignments()
        kept = [r for r in ref if r["add_status"] == 0]
        assert len(got) == len(kept)
        for i, (g, r) in enumerate(zip(got, kept)):
            if g.status != r["status"]:
                bad.append((k, i, "status", g.status, r["status"]))
            elif r["status"] == 0 and (list(g.alignment) != gen.unrle(r["alignment"]) or bool(g.is_optimal) != bool(r["optimal"])):
                bad.append((k, i, "alignment"))
    assert not bad, "pairs where a HIP aligner differs from the reference: %s" % bad[:10]

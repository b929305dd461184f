"""The reference's remaining cudaaligner vectors on the GPU (tests/golden/cudaaligner_vectors.json, extracted from its
test sources by tests/golden/make_aligner_vectors.py):
  * Test_HirschbergMyers.cu:94-211 -- exact pattern words of myers_preprocess, and the shifted views of
    get_query_pattern, through the gwhip_myers_test_* hooks (the production device functions);
  * Test_MyersAlgorithm.cu:179-270 -- neighbouring cells of the banded Myers matrices differ by at most 1, for every
    pair of cudaaligner_test_cases.cpp (11 fixed + 10 drawn with minstd_rand(5827349));
  * the same 21 pairs through every aligner class: CIGAR == oracle, edit distance == the reference's CPU NW whenever
    the result is optimal (the yardstick of Test_MyersAlgorithm.cu:179-190 / Test_AlignerGlobal.cpp)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_aligner as A

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "cudaaligner_vectors.json")) as f:
    V = json.load(f)
PAIRS = V["test_pairs"]


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def dev_str(s):
    b = s.encode() if isinstance(s, str) else bytes(s)
    return dev(np.frombuffer(b + b"\0" * 8, np.uint8).copy())


def lib():
    from genomeworks_amd import _native
    L = _native.gwhip()
    vp, i32 = C.c_void_p, C.c_int32
    L.gwhip_myers_test_patterns.argtypes = [vp, i32, vp, vp]
    L.gwhip_myers_test_get_pattern.argtypes = [vp, i32, i32, C.c_char, i32, vp, vp, vp]
    L.gwhip_myers_test_banded_matrices_words.restype = C.c_size_t
    L.gwhip_myers_test_banded_matrices_words.argtypes = [i32, i32, i32]
    L.gwhip_myers_test_banded_matrices.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp]
    return L


def pattern_matrix(query):
    import torch
    n_words = (len(query) + 31) // 32
    out = torch.zeros(max(n_words, 1) * 8, dtype=torch.int32, device="cuda")
    q = dev_str(query)
    assert lib().gwhip_myers_test_patterns(q.data_ptr(), len(query), out.data_ptr(), None) == 0
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint32)[:n_words * 8].reshape(n_words, 8)


def get_pattern(query, idx, x, reverse):
    import torch
    n_words = (len(query) + 31) // 32
    scratch = torch.zeros(max(n_words, 1) * 4, dtype=torch.int32, device="cuda")
    out = torch.zeros(32, dtype=torch.int32, device="cuda")
    q = dev_str(query)
    assert lib().gwhip_myers_test_get_pattern(q.data_ptr(), len(query), idx, x.encode(), 1 if reverse else 0,
                                              scratch.data_ptr(), out.data_ptr(), None) == 0
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint32)


def test_myers_preprocess_pattern_words():
    # Test_HirschbergMyers.cu:94-148
    hp = V["hirschberg_patterns"]
    query = hp["query"]
    p = pattern_matrix(query)
    assert p.shape == (2, 8)
    for key, want in hp["words"].items():
        r, c = (int(x) for x in key.split(","))
        assert int(p[r, c]) == want, key
    rev = pattern_matrix(query[::-1])
    for r in range(2):
        for c in range(4):
            assert rev[r, c] == p[r, c + 4] and rev[r, c + 4] == p[r, c]


@pytest.mark.parametrize("reverse", [False, True])
def test_myers_get_query_pattern_shifts(reverse):
    # Test_HirschbergMyers.cu:150-211: the view shifted by i equals the patterns of the query with i characters dropped
    # from its front (forward) or from its end (reverse)
    query = V["hirschberg_patterns"]["query"]
    p0 = get_pattern(query, 0, "A", reverse)
    p1 = get_pattern(query, 1, "A", reverse)
    for i in range(32):
        shifted = query[:len(query) - i] if reverse else query[i:]
        sp = pattern_matrix(shifted)
        col = 4 if reverse else 0
        assert p0[i] == sp[0, col], i
        assert p1[i] == (sp[1, col] if len(shifted) > 32 else 0), i


def banded_matrices(query, target, band_width, p):
    import torch
    L = lib()
    words = L.gwhip_myers_test_banded_matrices_words(len(query), len(target), band_width)
    ws = torch.zeros(words, dtype=torch.int32, device="cuda")
    diag = torch.zeros(2, dtype=torch.int32, device="cuda")
    q, t = dev_str(query), dev_str(target)
    assert L.gwhip_myers_test_banded_matrices(q.data_ptr(), t.data_ptr(), len(query), len(target), band_width, p,
                                              ws.data_ptr(), diag.data_ptr(), None) == 0
    torch.cuda.synchronize()
    nwb = (band_width + 31) // 32
    me = nwb * (len(target) + 1)
    lane0 = ws[::64].cpu().numpy()  # element k of lane 0 sits at word k * 64
    pv = lane0[:me].view(np.uint32).reshape(len(target) + 1, nwb)   # [column][word]
    mv = lane0[me:2 * me].view(np.uint32).reshape(len(target) + 1, nwb)
    score = lane0[2 * me:3 * me].reshape(len(target) + 1, nwb)
    return pv, mv, score


def myers_scores(pv, mv, score, band_width):
    """get_myers_score (Test_MyersAlgorithm.cu:142-154) for every i in [1, band_width], every column: [column][i-1]."""
    n_cols, nwb = score.shape
    last_mask = ((1 << (band_width % 32)) - 1) if band_width % 32 else 0xffffffff
    out = np.zeros((n_cols, band_width), np.int64)
    bits = np.arange(32, dtype=np.uint64)
    for w in range(nwb):
        rows = min(32, band_width - 32 * w)
        # mask(bit) = (~1 << bit), restricted to the valid bits of the last word
        masks = (np.uint64(0xffffffff) & (~np.uint64(1) << bits)).astype(np.uint64)
        if w == nwb - 1:
            masks &= np.uint64(last_mask)
        p = pv[:, w].astype(np.uint64)[:, None] & masks[None, :rows]
        m = mv[:, w].astype(np.uint64)[:, None] & masks[None, :rows]
        popc = lambda a: np.unpackbits(a.astype(">u8").view(np.uint8).reshape(a.shape + (8,)), axis=-1).sum(-1)
        out[:, 32 * w:32 * w + rows] = score[:, w][:, None] - popc(p) + popc(m)
    return out


@pytest.mark.parametrize("k", range(len(PAIRS)))
def test_banded_matrix_neighbours_differ_by_at_most_one(k):
    # Test_MyersAlgorithm.cu:196-270
    t, q = PAIRS[k]["target"], PAIRS[k]["query"]
    if not t or not q:
        pytest.skip("myers_banded is not defined for empty sequences (the reference skips them too)")
    ts, qs = len(t), len(q)
    estimate = max(ts, qs) // 4
    p = min(ts, qs, int((estimate - abs(ts - qs)) / 2))  # C++ integer division truncates towards zero
    bw = min(1 + 2 * p + abs(ts - qs), qs)
    if bw % 32 == 1 and bw != qs:
        p += 1
        bw = min(1 + 2 * p + abs(ts - qs), qs)
    pv, mv, score = banded_matrices(q, t, bw, p)
    s = myers_scores(pv, mv, score, bw)              # [column j][band row i - 1]
    # along rows: |s(i, j) - s(i, j - 1)| <= 1, the first column continuing from the row above (0 above row 1)
    assert (np.abs(np.diff(s, axis=0)) <= 1).all()
    first_col = np.concatenate([[0], s[0, :]])
    assert (np.abs(np.diff(first_col)) <= 1).all()
    # along columns: |s(i, j) - s(i - 1, j)| <= 1, the first band row continuing from the column to its left (1 before column 0)
    assert (np.abs(np.diff(s, axis=1)) <= 1).all()
    first_row = np.concatenate([[1], s[:, 0]])
    assert (np.abs(np.diff(first_row)) <= 1).all()


def _run(aligner_kwargs):
    from genomeworks_amd import cudaaligner
    al = cudaaligner.CudaAlignerBatch(max_device_memory_allocator_caching_size=16 << 30, **aligner_kwargs)
    status = [al.add_alignment(p["query"], p["target"]) for p in PAIRS]
    al.align_all()
    return status, al.get_alignments()


def test_all_test_pairs_banded_myers():
    status, res = _run(dict(max_bandwidth=8192))
    assert status == [0] * len(PAIRS) and len(res) == len(PAIRS)
    for p, r in zip(PAIRS, res):
        ref = A.align(p["query"], p["target"], 8192)
        # the host clamps max_bandwidth to the query length (aligner_global_myers_banded.cpp:174-178); a pair whose length
        # difference no longer fits that band (query "C" vs a 10-mer) is refused by the kernel and stays uninitialized
        assert r.status == (0 if ref["status"] == 0 else 1), (len(p["query"]), len(p["target"]))
        if ref["status"] != 0:
            assert abs(len(p["query"]) - len(p["target"])) > len(p["query"]) and r.cigar == ""
            continue
        assert r.cigar_extended == ref["cigar_extended"] and r.is_optimal == ref["optimal"]
        assert r.is_optimal and r.edit_distance == p["edit_distance"]


def test_all_test_pairs_default_hirschberg_myers():
    status, res = _run(dict(max_query_length=6000, max_target_length=6000, max_alignments=len(PAIRS)))
    assert status == [0] * len(PAIRS)
    for p, r in zip(PAIRS, res):
        ref = A.hirschberg(p["query"], p["target"], 6000)
        assert list(r.alignment) == ref["states"]
        assert r.edit_distance == p["edit_distance"]  # Hirschberg's divide and conquer is exact


def test_all_test_pairs_full_myers():
    status, res = _run(dict(max_query_length=6000, max_target_length=6000, max_alignments=len(PAIRS), algorithm="myers"))
    assert status == [0] * len(PAIRS)
    for p, r in zip(PAIRS, res):
        assert r.status == 0 and r.edit_distance == p["edit_distance"]
        assert list(r.alignment) == A.myers_full(p["query"], p["target"])["states"]


def test_all_test_pairs_ukkonen():
    status, res = _run(dict(max_query_length=6000, max_target_length=6000, max_alignments=len(PAIRS), algorithm="ukkonen"))
    # pairs whose lengths differ by more than 10 % of max_target_length are refused at add time (aligner_global_ukkonen.cpp:53-54)
    accepted = [p for p, st in zip(PAIRS, status) if st == 0]
    assert len(accepted) == len(res) and len(accepted) >= 19
    for p, r in zip(accepted, res):
        ref = A.ukkonen(p["query"], p["target"], 100)
        assert list(r.alignment) == ref["states"]
        assert r.edit_distance >= p["edit_distance"]
        if abs(len(p["query"]) - len(p["target"])) + 2 * p["edit_distance"] < 100:  # the optimum lies inside the band
            assert r.edit_distance == p["edit_distance"]


def test_full_myers_accepts_targets_twice_as_long_as_the_query():
    """AlignerGlobalMyers admits every pair inside its limits (advisor finding r1: a target >= 2 x query was rejected)."""
    from genomeworks_amd import cudaaligner
    al = cudaaligner.CudaAlignerBatch(64, 64, 4, algorithm="myers", max_device_memory_allocator_caching_size=1 << 30)
    pairs = [("ACGT", "ACGTACGTAC"), ("A", "TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT"), ("ACGTACGTAC", "ACG"), ("GATTACA", "GATTACAGATTACAGATTACAGATT")]
    for q, t in pairs:
        assert al.add_alignment(q, t) == 0
    al.align_all()
    for (q, t), r in zip(pairs, al.get_alignments()):
        ref = A.myers_full(q, t)
        assert r.status == 0 and list(r.alignment) == ref["states"]
        R = A.ref()
        if R is not None:
            assert r.edit_distance == R.ref_nw_edit_distance(t.encode(), len(t), q.encode(), len(q))


def test_aligner_filled_to_its_memory_limit_still_runs():
    """add_alignment's memory estimate covers what align_all allocates (advisor finding r1): fill a small budget until
    exceeded_max_alignments, then align."""
    from genomeworks_amd import cudaaligner, synthetic
    pairs = synthetic.generate_pairs(11, 4000, 150, 5, 5, 5)
    al = cudaaligner.CudaAlignerBatch(max_bandwidth=32, max_device_memory_allocator_caching_size=16 << 20)
    n = 0
    for q, t in pairs:
        st = al.add_alignment(q, t)
        if st == cudaaligner.exceeded_max_alignments:
            break
        assert st == 0
        n += 1
    assert 100 < n < len(pairs)
    al.align_all()
    res = al.get_alignments()
    assert len(res) == n and all(r.status == 0 for r in res)
    assert res[0].cigar_extended == A.align(pairs[0][0], pairs[0][1], 32)["cigar_extended"]

"""Known answers for the seeded sequence generators (include/.../utils/genomeutils.hpp): the ten random pairs of the
reference's aligner test cases (cudaaligner/tests/cudaaligner_test_cases.cpp:25-41, std::minstd_rand(5827349)) as the
reference's own generator printed them (tests/golden/make_aligner_vectors.py compiled that file where it lies). They
pin the number and order of engine values every generator decision consumes -- the contract that makes a seed name the
same synthetic reads here, in the oracle runs behind the goldens, and in the reference."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_test_cases_of_the_reference_are_reproduced():
    from genomeworks_amd import synthetic
    with open(os.path.join(ROOT, "tests", "golden", "cudaaligner_vectors.json")) as f:
        golden = json.load(f)["test_pairs"][11:]
    assert len(golden) == 10
    ours = synthetic.random_length_pairs(5827349, 10, 5000)
    for k, (g, (t, q)) in enumerate(zip(golden, ours)):
        assert t.decode() == g["target"], "target of random pair %d" % k
        assert q.decode() == g["query"], "query of random pair %d" % k


def test_generated_reads_only_differ_by_the_requested_edit_budget():
    from genomeworks_amd import synthetic
    w = synthetic.generate_window(42, 300, 6, 5, 3, 2)
    assert len(w) == 6 and len(w[0]) == 300 and set(w[0]) <= set(b"ACGT")
    for r in w[1:]:
        assert 300 - 2 <= len(r) <= 300 + 3
    assert synthetic.generate_window(42, 300, 6, 5, 3, 2) == w          # deterministic
    assert synthetic.generate_window(43, 300, 6, 5, 3, 2) != w

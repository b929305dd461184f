"""A randomised sweep of the HIP path against the oracle (tools/fuzz_gpu_vs_oracle.py) with a fixed seed: band mode x band width x
output type x scores x read counts / lengths / divergence drawn at random, and the aligner classes over random batches. Round 6 ran
three seeds x (400 + 100) cases without a difference (profiles/r06_gpu_fuzz_vs_oracle.jsonl); the suite keeps a small one."""
import importlib.util
import os
import random

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_gpu_vs_oracle", os.path.join(ROOT, "tools", "fuzz_gpu_vs_oracle.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_random_poa_configurations_equal_the_oracle():
    F = _fuzz()
    rng = random.Random(7)
    seen, bad = set(), []
    for k in range(120):
        c = F.poa_case(rng, k)
        seen.add((c["mode"], c["msa"]))
        if not F.run_poa(c):
            bad.append((k, c["mode"], c["band"], c["max_seq"], c["msa"], len(c["reads"])))
    assert len(seen) == 10  # every band mode, consensus and MSA
    assert not bad, bad


def test_random_aligner_batches_equal_the_oracles():
    F = _fuzz()
    rng = random.Random(8)
    kinds, bad = set(), []
    for k in range(40):
        c = F.aligner_case(rng)
        kinds.add(c["kind"])
        if not F.run_aligner(c):
            bad.append((k, c["kind"], len(c["pairs"]), c["max_bandwidth"]))
    assert kinds == {"banded", "default", "ukkonen", "myers"}
    assert not bad, bad

"""The oracle against THE REFERENCE ITSELF: the reference's cudapoa library (its CUDA sources compiled by g++ from where they lie,
kernels run on the CPU by the SIMT emulator oracle/simt) answered the windows of tests/golden/reference_simt_windows.json.gz
(tests/golden/make_reference_simt_goldens.py). The oracle must give the same statuses, consensus, coverage and MSA rows for every
window; where the reference library is present (oracle/_ref/libref_cudapoa_simt.so: this container, and the GPU box when the
prebuilt file travelled) a sample of the file is regenerated and windows that are not in the file are compared as well (a fixed seed; GW_SIMT_SEED picks others)."""
import gzip
import json
import os
import random

import pytest

import oracle_poa as O
import ref_cudapoa as R

HERE = os.path.dirname(os.path.abspath(__file__))


def golden():
    with gzip.open(os.path.join(HERE, "golden", "reference_simt_windows.json.gz"), "rb") as f:
        return json.loads(f.read().decode())["windows"]


def accepted_reads(c, ref):
    """add_poa_group refuses single reads (too long, too many: its per-read statuses say which); the window is what it accepted"""
    keep = [i for i, st in enumerate(ref["read_status"]) if st == 0]
    return dict(c, reads=[c["reads"][i] for i in keep], weights=None if c["weights"] is None else [c["weights"][i] for i in keep])


def oracle_answer(c):
    cfg = O.make_cfg(c["max_seq"], c["max_seqs"], c["band_width"], c["band_mode"], gap=c["gap"], mismatch=c["mismatch"], match=c["match"],
                     output_mask=c["output_mask"], max_pred=c.get("max_pred", 0))
    with O.Workspace(cfg) as ws:
        return ws.process(c["reads"], c["weights"])


def same(c, ref, mine):
    if ref["status"] != mine["status"]:
        return False
    if mine["status"] != 0:
        return True
    if c["output_mask"] & 1:
        return ref["consensus"] == mine["consensus"] and list(ref["coverage"]) == [int(x) for x in mine["coverage"]]
    return ref["msa"] == mine["msa"]


def test_oracle_equals_the_reference_on_every_golden_window():
    rows = golden()
    assert len(rows) >= 87 and sum(1 for r in rows if r["case"].get("max_pred") == 8) >= 7 and {r["case"]["band_mode"] for r in rows} == {0, 1, 2, 3, 4}
    assert any(r["case"]["weights"] for r in rows) and {r["case"]["output_mask"] for r in rows} == {1, 2}
    assert sum(r["reference"]["status"] == 4 for r in rows) >= 3          # node_count_exceeded_maximum_graph_size
    assert any(2 in r["reference"]["read_status"] for r in rows)          # exceeded_maximum_sequence_size
    assert any(3 in r["reference"]["read_status"] for r in rows)          # exceeded_maximum_sequences_per_poa
    bad = [i for i, r in enumerate(rows)
           if r["reference"]["add_status"] == 0 and not same(r["case"], r["reference"], oracle_answer(accepted_reads(r["case"], r["reference"])))]
    assert not bad, "windows where the oracle differs from the reference: %s" % bad


@pytest.mark.skipif(not R.available(), reason="the reference library (oracle/_ref/libref_cudapoa_simt.so) is not built here")
def test_reference_library_reproduces_a_sample_of_the_golden_file():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_reference_simt_goldens", os.path.join(HERE, "golden", "make_reference_simt_goldens.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    rows, cases = golden(), gen.cases()
    assert [r["case"] for r in rows] == cases  # the generator still describes the file
    for i in (0, 9, 17, 30, 44, 58, 70, 72, 75, 76, 78, 85):
        assert gen.run_reference(cases[i]) == rows[i]["reference"], i


@pytest.mark.skipif(not R.available(), reason="the reference library (oracle/_ref/libref_cudapoa_simt.so) is not built here")
def test_oracle_equals_the_reference_on_fresh_random_windows():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_reference_simt_goldens", os.path.join(HERE, "golden", "make_reference_simt_goldens.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    seed = int(os.environ.get("GW_SIMT_SEED", "424242"))  # (GW_SIMT_SEED=<n>: other windows; tools/explore_reference_simt.py sweeps seeds)
    rng = random.Random(seed)
    for k in range(12):
        L = rng.choice([20, 70, 130, 220])
        base = "".join(rng.choice("ACGT") for _ in range(L))
        div = rng.choice([0, 15, 8, 5])
        c = dict(max_seq=256, max_seqs=12, band_width=rng.choice([128, 256]), band_mode=rng.randint(0, 4), output_mask=rng.choice([1, 2]),
                 gap=-8, mismatch=-6, match=8, reads=[gen.mutate(rng, base, L // div if div else 0) for _ in range(rng.randint(1, 10))], weights=None)
        ref = gen.run_reference(c)
        assert ref["add_status"] == 0
        assert same(c, ref, oracle_answer(c)), "seed %d window %d: %s" % (seed, k, c)


@pytest.mark.skipif(not R.available(), reason="the reference library (oracle/_ref/libref_cudapoa_simt.so) is not built here")
def test_batch_config_constructor_equals_the_reference():
    """BatchConfig(max_seq_sz, max_seq_per_poa, band_width, banding, adaptive_storage_factor, graph_length_factor, max_pred_dist):
    the host library's constructor (gw_poa_batch_config_default, no device call) against the reference's own, field by field,
    over band modes, widths that are and are not multiples of 128, the factors and predecessor distances."""
    import ctypes as C
    from genomeworks_amd import _native, cudapoa
    L = cudapoa._bind(_native.host())
    keys = ("max_sequence_size", "max_consensus_size", "max_nodes_per_graph", "matrix_sequence_dimension", "alignment_band_width",
            "max_sequences_per_poa", "band_mode", "max_banded_pred_distance")
    checked = 0
    for mode in range(5):
        for max_seq in (32, 100, 1024, 5000, 40000):
            for band in (1, 128, 200, 256, 1000):
                for sf, gf, pred in ((2.0, 3.0, 0), (1.5, 4.0, 0), (4.0, 3.0, 100), (2.0, 2.5, 1000)):
                    ref = R.config(max_seq, 31, band, mode, sf, gf, pred)
                    cfg = _native.PoaBatchConfig()
                    assert L.gw_poa_batch_config_default(C.byref(cfg), max_seq, 31, band, mode, sf, gf, pred) == 0
                    assert [getattr(cfg, k) for k in keys] == [ref[k] for k in keys], (mode, max_seq, band, sf, gf, pred)
                    checked += 1
    assert checked == 500


# ---- cudaaligner: the reference's own aligners (tests/golden/make_reference_simt_alignments.py) ------------------------------
import oracle_aligner as A
import ref_cudaaligner as RA


def alignment_golden():
    with gzip.open(os.path.join(HERE, "golden", "reference_simt_alignments.json.gz"), "rb") as f:
        return json.loads(f.read().decode())["batches"]


def _aligner_generator():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_reference_simt_alignments", os.path.join(HERE, "golden", "make_reference_simt_alignments.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    return gen


def oracle_alignment(kind, q, t, max_bandwidth=None, max_len=None):
    """-> (status == success, optimal, [AlignmentState...]) of the oracle that restates that aligner"""
    if kind == "banded":
        r = A.align(q, t, max_bandwidth)
        states = [o for o, k in r["runs"] for _ in range(k)] if r["status"] == 0 else None
        return r["status"] == 0, r["optimal"] if r["status"] == 0 else None, states
    if kind in ("default", "hirschberg_myers"):
        r = A.hirschberg(q, t, max_len)
    elif kind == "ukkonen":
        r = A.ukkonen(q, t, 100)
    else:
        r = A.myers_full(q, t)
    return True, True, list(r["states"])


def differences(gen, b, ref):
    max_len = max(max(len(q), len(t)) for q, t in b["pairs"])
    bad = []
    for i, ((q, t), r) in enumerate(zip(b["pairs"], ref)):
        ok, optimal, states = oracle_alignment(b["kind"], q, t, b.get("max_bandwidth"), max_len)
        if r["add_status"] != 0:
            continue  # refused by add_alignment: the host classes are compared on the GPU
        ref_ok = r["status"] == 0 and r["alignment"] is not None and (r["alignment"] != "" or (q == "" and t == ""))
        if ok != ref_ok or (ok and (states != gen.unrle(r["alignment"]) or bool(optimal) != bool(r["optimal"]))):
            bad.append(i)
    return bad


def test_aligner_oracles_equal_the_reference_on_every_golden_batch():
    gen, rows = _aligner_generator(), alignment_golden()
    assert {r["batch"]["kind"] for r in rows} == {"default", "banded", "ukkonen", "myers"}
    assert sum(len(r["batch"]["pairs"]) for r in rows) >= 220
    assert any(max(len(q) for q, _ in r["batch"]["pairs"]) >= 5000 for r in rows)           # the Hirschberg recursion, several levels
    assert any(x["status"] != 0 for r in rows for x in r["reference"])                      # pairs a band rejected
    assert any(x["optimal"] is False for r in rows for x in r["reference"] if x["status"] == 0)  # approximate results of a narrow band
    for k, r in enumerate(rows):
        bad = differences(gen, r["batch"], r["reference"])
        assert not bad, "batch %d (%s, max_bandwidth %s): pairs where the oracle differs from the reference: %s" % (
            k, r["batch"]["kind"], r["batch"].get("max_bandwidth"), bad)


@pytest.mark.skipif(not RA.available(), reason="the reference library (oracle/_ref/libref_cudaaligner_simt.so) is not built here")
def test_reference_aligner_library_reproduces_the_golden_file_and_fresh_pairs():
    gen, rows = _aligner_generator(), alignment_golden()
    assert [r["batch"] for r in rows] == gen.batches()
    for k in (1, 2, 6, 10, 11, 12, 13, 15):
        assert gen.run_reference(rows[k]["batch"]) == rows[k]["reference"], k
    seed = int(os.environ.get("GW_SIMT_SEED", "434343"))
    rng = random.Random(seed)
    for kind, bw in (("default", None), ("banded", 31), ("banded", 256), ("ukkonen", None), ("myers", None)):
        b = dict(kind=kind, pairs=gen.random_pairs(rng, 16, [3, 30, 64, 129, 260], 300))
        if bw:
            b["max_bandwidth"] = bw
        assert not differences(gen, b, gen.run_reference(b)), "seed %d %s %s" % (seed, kind, bw)


def test_config_goldens_were_checked_against_the_reference():
    """tests/golden/reference_simt_config_check.json is the record of tests/golden/check_goldens_against_reference.py: which units
    of the committed BASELINE-config goldens the reference itself (on the SIMT emulator) was asked for, and that none differed.
    Where the reference library is present, a few pairs of configs[1] are asked again."""
    with open(os.path.join(HERE, "golden", "reference_simt_config_check.json")) as f:
        rec = json.load(f)
    assert rec["config3"]["windows_checked"] == list(range(1024)) and rec["config3"]["windows_differing"] == []   # every window of the metric
    assert len(rec["config4"]["windows_checked"]) >= 424 and rec["config4"]["windows_differing"] == []            # long reads: the smallest
    assert sum(hi - lo for lo, hi in rec["config2"]["pair_ranges_checked"]) == 10000 and rec["config2"]["ranges_differing"] == []
    assert sum(hi - lo for lo, hi in rec["config5"]["pair_ranges_checked"]) == 1000000 and rec["config5"]["ranges_differing"] == []
    if RA.available():
        import importlib.util
        spec = importlib.util.spec_from_file_location("check_goldens_against_reference", os.path.join(HERE, "golden", "check_goldens_against_reference.py"))
        chk = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(chk)
        assert chk.check_pairs(("config2", 5000, 5012)) == ("config2", [5000, 5012], [])


@pytest.mark.skipif(not R.available(), reason="the reference library (oracle/_ref/libref_cudapoa_simt.so) is not built here")
def test_multi_batch_binning_equals_the_reference():
    """get_multi_batch_sizes (cudapoa/src/utils.cu:30-135): the host library's binning rule (gw_poa_bin_groups, no device call) fed
    with the per-group capacities the reference itself estimates (BatchBlock::estimate_max_poas over the stub's 8 GB of free
    memory) gives the reference's batches: the same BatchConfig per batch and the same groups in the same order. (The capacities
    themselves differ by design -- our device layout has other byte counts per window)"""
    import ctypes as C
    from genomeworks_amd import cudapoa
    L = R.lib()
    L.ref_poa_estimate_max_poas.restype = C.c_longlong
    L.ref_poa_estimate_max_poas.argtypes = [C.c_int] * 4 + [C.c_float, C.c_float, C.c_int, C.c_int, C.c_float] + [C.c_int] * 3
    pi = C.POINTER(C.c_int)
    L.ref_poa_multi_batch_sizes.argtypes = [C.c_int, pi, pi] + [C.c_int] * 3 + [C.c_float, C.c_float, C.c_int, C.c_float] + [C.c_int] * 3 + [pi, pi, pi]
    rng = random.Random(2718)
    keys = ("max_sequence_size", "max_consensus_size", "max_nodes_per_graph", "matrix_sequence_dimension", "alignment_band_width",
            "max_sequences_per_poa", "band_mode", "max_banded_pred_distance")
    modes = {0: "full_band", 1: "static_band", 2: "adaptive_band", 3: "static_band_traceback", 4: "adaptive_band_traceback"}
    for trial in range(40):
        n = rng.randint(1, 60)
        longest = [rng.choice([100, 300, 1000, 3000, 9000, 20000, 30000]) + rng.randint(0, 99) for _ in range(n)]
        reads = [rng.randint(1, 60) for _ in range(n)]
        mode, msa = rng.randint(0, 4), rng.randint(0, 1)
        band = rng.choice([128, 256, 512])
        cfg8, per, ids = (C.c_int * (8 * n))(), (C.c_int * n)(), (C.c_int * n)()
        nb = L.ref_poa_multi_batch_sizes(n, (C.c_int * n)(*longest), (C.c_int * n)(*reads), msa, band, mode, 2.0, 3.0, 0, 0.9, -6, -8, 8, cfg8, per, ids)
        ref_cfgs = [dict(zip(keys, list(cfg8[8 * b:8 * b + 8]))) for b in range(nb)]
        ref_groups, at = [], 0
        for b in range(nb):
            ref_groups.append(list(ids[at:at + per[b]]))
            at += per[b]
        capacity = [int(min(L.ref_poa_estimate_max_poas(longest[g], reads[g], band, mode, 2.0, 3.0, 0, msa, 0.9, -6, -8, 8), 2 ** 31 - 1)) for g in range(n)]
        cfgs, groups = cudapoa.bin_poa_groups(capacity, longest, reads, band, modes[mode])
        assert groups == ref_groups, (trial, capacity)
        assert [{k: c[k] for k in keys} for c in cfgs] == ref_cfgs, trial

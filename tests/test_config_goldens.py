"""CPU checks of the committed config goldens: the files are what config_goldens.json says they are, and the oracle
still reproduces a sample of every file (a drifting oracle or generator fails here, without a GPU)."""
import hashlib

import numpy as np

import golden_io as G
import oracle_aligner as A
import oracle_poa as O


def test_fingerprint_function_is_order_and_content_sensitive():
    fp = G.run_fingerprints([0, 2, 4, 4, 5], np.array([0, 1, 1, 0, 2], np.int8), np.array([5, 1, 5, 1, 7], np.int32))
    assert len(fp) == 4 and fp[2] == 0 and len({int(x) for x in fp}) == 4
    swapped = G.run_fingerprints([0, 2], np.array([1, 0], np.int8), np.array([1, 5], np.int32))
    assert swapped[0] != fp[0]


def test_config3_file_matches_summary_and_oracle_sample():
    rows, sha = G.config3_windows()
    s = G.summary()["config3"]
    assert sha == s["sha256"] and len(rows) == s["windows"] == 1024
    assert sum(r["cells"] for r in rows) == s["cells"]
    assert all(r["status"] == 0 and len(r["consensus"]) == len(r["coverage"]) for r in rows)
    from genomeworks_amd import synthetic
    cfg = O.make_cfg(s["max_seq"], s["max_seqs"], s["band"], s["band_mode"])
    with O.Workspace(cfg) as ws:
        for w in (0, 1, 77, 512, 1023):
            ref = ws.process([r.decode() for r in synthetic.generate_window(s["first_seed"] + w)])
            assert (ref["status"], ref["cells"], ref["consensus"], list(ref["coverage"])) == \
                   (rows[w]["status"], rows[w]["cells"], rows[w]["consensus"], rows[w]["coverage"])


def _check_pairs(name, cfg, golden, sample):
    from genomeworks_amd import synthetic
    pairs = synthetic.generate_pairs(cfg["seed"], max(sample) + 1, cfg["length"], cfg["mut"], cfg["ins"], cfg["dele"])
    for i in sample:
        r = A.align(pairs[i][0], pairs[i][1], cfg["max_bandwidth"])
        ops = np.array([o for o, _ in r["runs"]], np.int8)
        cnt = np.array([c for _, c in r["runs"]], np.int32)
        assert r["edit_distance"] == int(golden["edit_distance"][i]), (name, i)
        if "fingerprint" in golden:
            assert int(G.run_fingerprints([0, len(ops)], ops, cnt)[0]) == int(golden["fingerprint"][i]), (name, i)


def test_config2_file_matches_summary_and_oracle_sample():
    g, s = G.config2_pairs(), G.summary()["config2"]
    assert hashlib.sha256(g["fingerprint"].tobytes()).hexdigest() == s["fingerprint_sha256"]
    assert len(g["fingerprint"]) == s["pairs"] == 10000 and int(g["cells"].sum()) == s["band_cells"]
    assert int((g["status"] == 0).sum()) == 10000
    _check_pairs("config2", s, g, [0, 1, 499, 1500])


def test_config5_file_matches_summary_and_oracle_sample():
    g, s = G.config5_pairs(), G.summary()["config5"]
    assert len(g["edit_distance"]) == s["pairs"] == 1000000
    assert len(g["block_sha"]) == (s["pairs"] + s["block"] - 1) // s["block"]
    assert int(g["edit_distance"].astype(np.int64).sum()) == s["edit_distance_sum"]
    _check_pairs("config5", s, g, [0, 3, 1023, 1024, 4999])
    # the first block's digest from the oracle
    from genomeworks_amd import synthetic
    pairs = synthetic.generate_pairs(s["seed"], s["block"], s["length"], s["mut"], s["ins"], s["dele"])
    offs, ops, cnt = [0], [], []
    for q, t in pairs:
        for o, c in A.align(q, t, s["max_bandwidth"])["runs"]:
            ops.append(o)
            cnt.append(c)
        offs.append(len(ops))
    fp = G.run_fingerprints(offs, np.array(ops, np.int8), np.array(cnt, np.int32))
    assert G.block_digests(fp, s["block"])[0] == str(g["block_sha"][0])
    # the scalar model of the kernels' backtrace step (delta bits instead of three cell reads) ran next to the reference's reads
    assert A.delta_identity_mismatches() == 0


def test_config4_plan_is_reproducible_and_oracle_sample_matches():
    import importlib.util
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_long_read_goldens", os.path.join(here, "make_long_read_goldens.py"))
    lr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lr)
    with open(os.path.join(here, "config4_long_reads.json")) as f:
        golden = json.load(f)
    s = G.summary()["config4"]
    det = golden["windows_detail"]
    assert len(det) == 598 and sum(d["cells"] for d in det) == s["cells"]
    assert hashlib.sha256("".join(d["msa_sha"] for d in det).encode()).hexdigest() == s["digest"]
    assert sum(1 for d in det if d["status"] != 0) * 50 < len(det)  # < 2 % of the windows end in an error status
    # the plan only needs the lengths: the first 40 windows are enough to check the generator, the oracle takes the smallest
    from genomeworks_amd import synthetic
    small = sorted(range(598), key=lambda w: det[w]["cells"])[:2]
    for w in small:
        reads = synthetic.long_read_window(w, golden["max_len"])
        c = golden["batch_configs"][det[w]["cfg"]]
        with O.Workspace(lr.oracle_cfg(c)) as ws:
            ref = ws.process(reads[:c["max_sequences_per_poa"]])
        assert ref["status"] == det[w]["status"] and ref["cells"] == det[w]["cells"]
        assert lr.msa_digest(ref["msa"]) == det[w]["msa_sha"]
    # the scalar model of the kernel's MSA rows (scatter over the nodes) ran next to the reference's walk on these windows too
    assert O.Workspace.msa_scatter_mismatches() == 0


def test_long_read_prefix_fixture_of_the_reference_and_oracle_sample():
    """tests/golden/reference_simt_long_prefixes.json (the reference's own answers for cut-down windows of the two largest
    size classes of configs[3]): the file covers both classes, every row was equal to the oracle when it was written, the oracle
    still reproduces the cheapest rows of every set, and reference_simt_config_check.json lists these windows beside the 424 whole
    windows of all four size classes that the reference has answered."""
    import importlib.util
    import json
    import os
    from genomeworks_amd import synthetic
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_long_read_goldens", os.path.join(here, "make_long_read_goldens.py"))
    lr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lr)
    with open(os.path.join(here, "reference_simt_long_prefixes.json")) as f:
        fixture = json.load(f)
    with open(os.path.join(here, "config4_long_reads.json")) as f:
        golden = json.load(f)
    with open(os.path.join(here, "reference_simt_config_check.json")) as f:
        check = json.load(f)
    rows = fixture["windows"]
    assert fixture["batch_configs"] == golden["batch_configs"][:2]
    assert len(rows) >= 60 and all(r["oracle_equal"] and r["add_status"] == 0 for r in rows)
    # the record of what the reference has answered lists them (beside the whole windows: 245 when these were cut, 424 since)
    assert sorted(r["w"] for r in rows) == check["config4_prefixes"]["windows_checked"] and not check["config4_prefixes"]["windows_differing"]
    assert len(check["config4"]["windows_checked"]) >= 424 and not check["config4"]["windows_differing"]
    by_class = {}
    for w in check["config4"]["windows_checked"]:
        by_class[golden["windows_detail"][w]["cfg"]] = by_class.get(golden["windows_detail"][w]["cfg"], 0) + 1
    assert by_class[3] == 130 and by_class[2] == 160 and by_class[1] >= 93 and by_class[0] >= 41  # every size class
    assert all(golden["windows_detail"][r["w"]]["cfg"] == r["cfg"] for r in rows)
    for n_reads in sorted({r["reads"] for r in rows}):
        cheapest = min((r for r in rows if r["reads"] == n_reads), key=lambda r: r["cells"])
        if cheapest["cells"] > 150e6:
            continue
        with O.Workspace(lr.oracle_cfg(fixture["batch_configs"][cheapest["cfg"]])) as ws:
            ref = ws.process(synthetic.long_read_window(cheapest["w"], golden["max_len"])[:n_reads])
        assert ref["status"] == cheapest["status"] and ref["cells"] == cheapest["cells"]
        assert lr.msa_digest(ref["msa"]) == cheapest["msa_sha"] and len(ref["msa"]) == cheapest["msa_rows"]


def test_band_mode_table_file_matches_summary_and_oracle_sample():
    """tests/golden/band_mode_goldens.{npz,json}: 4 banded modes x 4 band widths x 1024 windows. The file is what its
    summary says, the static-band / 256 cell is the config-3 golden, and the oracle still reproduces two windows of
    every cell."""
    from genomeworks_amd import synthetic
    s, g = G.band_mode_summary(), G.band_mode_goldens()
    assert s["windows"] == 1024 and g["fingerprint"].shape == (len(s["modes"]), len(s["widths"]), 1024)
    for mi, mode in enumerate(s["modes"]):
        for wi, width in enumerate(s["widths"]):
            cell = s["cells"]["%s/%d" % (mode, width)]
            assert int(g["cells"][mi, wi].sum()) == cell["cells"]
            assert G.band_gen.cell_digest(g["fingerprint"][mi, wi]) == cell["fingerprint_sha256"]
            assert cell["oracle_int16_overflow_events"] == 0
            with O.Workspace(G.band_gen.cell_cfg(mode, width)) as ws:
                for w in (3, 1000):
                    ref = ws.process([r.decode() for r in synthetic.generate_window(s["first_seed"] + w)])
                    assert ref["status"] == int(g["status"][mi, wi, w]) and ref["cells"] == int(g["cells"][mi, wi, w])
                    assert G.band_gen.fingerprint(ref["status"], ref.get("consensus", ""), ref.get("coverage", [])) == int(g["fingerprint"][mi, wi, w])
    # the metric cell equals the full-text config-3 golden
    rows, _ = G.config3_windows()
    mi, wi = s["modes"].index("static_band"), s["widths"].index(256)
    assert s["cells"]["static_band/256"]["cells"] == G.summary()["config3"]["cells"]
    for w in (0, 511, 1023):
        assert G.band_gen.fingerprint(rows[w]["status"], rows[w]["consensus"], rows[w]["coverage"]) == int(g["fingerprint"][mi, wi, w])


def test_full_band_file_matches_summary_and_oracle_sample():
    """tests/golden/full_band_goldens.{npz,json}: BatchConfig(1024, 200) = full band on the 1024 metric windows."""
    from genomeworks_amd import synthetic
    s, g = G.full_band_summary(), G.full_band_goldens()
    assert s["windows"] == 1024 and g["fingerprint"].shape == (1024,)
    assert int(g["cells"].sum()) == s["cells"] and G.band_gen.cell_digest(g["fingerprint"]) == s["fingerprint_sha256"]
    assert s["oracle_int16_overflow_events"] == 0 and int((g["status"] == 0).sum()) == 1024
    with O.Workspace(G.full_gen.full_band_cfg()) as ws:
        for w in (5, 700):
            ref = ws.process([r.decode() for r in synthetic.generate_window(s["first_seed"] + w)])
            assert ref["status"] == int(g["status"][w]) and ref["cells"] == int(g["cells"][w])
            assert G.band_gen.fingerprint(ref["status"], ref.get("consensus", ""), ref.get("coverage", [])) == int(g["fingerprint"][w])


def test_default_aligner_golden_file_matches_oracle_sample():
    """tests/golden/default_aligner_goldens.json: the oracle still reproduces the short single-pair shapes and the first pairs
    of the batch shapes (the 100 kbp pair takes the oracle 30 s and is left to the generator)."""
    gold = G.default_aligner_goldens()
    assert sorted(gold) == sorted("%dx%d" % s for s in G.aligner_gen.SHAPES)
    for n, size in [(1, 100), (1, 1000), (1, 10000)]:
        (q, t), = G.aligner_gen.shape_pairs(n, size)
        ref = A.hirschberg(q, t, size)
        assert G.aligner_gen.digest([G.aligner_gen.pair_record(ref["status"], ref["states"])]) == gold["%dx%d" % (n, size)]["states_sha256"]
        assert ref["edit_distance"] == gold["%dx%d" % (n, size)]["edit_distance_sum"]


def test_aligner_matrix_golden_file_matches_oracle_sample():
    """tests/golden/aligner_matrix_goldens.json: the cells bench.py publishes; the Hirschberg cell at 1024 x 2048 is the
    default aligner's golden of the same pairs, and each class's oracle still reproduces the first pairs of a cell."""
    gold, dgold = G.aligner_matrix_goldens(), G.default_aligner_goldens()
    assert sorted(gold) == sorted(G.matrix_gen.cell_key(*c) for c in G.matrix_gen.CELLS + G.matrix_gen.CORNER_CELLS)
    # the long cells: each class's oracle on the first pair of the 32 x 32768 corner (the full-matrix Myers oracle takes 6 s there)
    long_pair = G.aligner_gen.shape_pairs(32, 32768)[0]
    for algorithm in ("ukkonen", "myers_banded", "hirschberg_myers"):
        rec = G.matrix_gen._one((algorithm, long_pair[0], long_pair[1], 32768))
        assert rec[1] >= 0 and rec[2] >= 32768
    assert gold["hirschberg_myers/1024x2048"]["states_sha256"] == dgold["1024x2048"]["states_sha256"]
    pairs = G.aligner_gen.shape_pairs(1024, 2048)[:3]
    for algorithm in ("ukkonen", "myers", "myers_banded", "hirschberg_myers"):
        recs = [G.matrix_gen._one((algorithm, q, t, 2048)) for q, t in pairs]
        assert all(r[1] >= 0 and r[2] >= 2048 for r in recs)
    # all four classes are optimal on these pairs: equal edit-distance sums
    assert len({gold[G.matrix_gen.cell_key(a, 1024, 2048)]["edit_distance_sum"] for a in ("ukkonen", "myers", "myers_banded", "hirschberg_myers")}) == 1


def _oracle_runs(name, lo, hi):
    """What CudaAlignerBatch.get_runs() returns for the pairs [lo, hi) of a config, from the oracle."""
    import oracle_aligner as A
    pairs = G.gen.pairs_of(name)[lo:hi]
    c = G.gen.CONFIG2 if name == "config2" else G.gen.CONFIG5
    offs, ops, cnts, status, opt = [0], [], [], [], []
    for q, t in pairs:
        r = A.align(q, t, c["max_bandwidth"])
        status.append(r["status"])
        opt.append(1 if r["optimal"] else 0)
        for o, k in r["runs"]:
            ops.append(o)
            cnts.append(k)
        offs.append(len(ops))
    return {"offsets": np.array(offs, np.int64), "ops": np.array(ops, np.int8), "counts": np.array(cnts, np.int32),
            "status": np.array(status, np.int8), "optimal": np.array(opt, np.uint8)}


def test_bench_aligner_golden_verdict_accepts_the_oracle_and_rejects_a_changed_run():
    """bench.py's checker of the configs[1] / configs[4] records (aligner_golden_verdict): a rank's range of pairs -- block
    aligned or not, the start, the middle or the end of the config -- passes with the oracle's runs and fails when one run
    length, one operation, one flag or one status differs."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for key, lo, hi in (("config2", 0, 40), ("config2", 9990, 10000), ("config5", 0, 2048), ("config5", 1000, 3100),
                        ("config5", 1000000 - 1700, 1000000)):
        runs = _oracle_runs(key, lo, hi)
        ok, what = bench.aligner_golden_verdict(key, runs, lo, hi)
        assert ok, (key, lo, hi, what)
        for field, at in (("counts", len(runs["counts"]) // 2), ("ops", len(runs["ops"]) // 3), ("optimal", 5), ("status", 7)):
            bad = {k: v.copy() for k, v in runs.items()}
            bad[field][at] = bad[field][at] + 1 if field == "counts" else (bad[field][at] + 1) % 3 if field == "ops" else bad[field][at] ^ 1
            assert not bench.aligner_golden_verdict(key, bad, lo, hi)[0], (key, lo, hi, field)
    # a block that is wrong in the middle of an unaligned range
    runs = _oracle_runs("config5", 1000, 3100)
    i = int(runs["offsets"][1500])
    runs["ops"][i] = (runs["ops"][i] + 1) % 3
    assert not bench.aligner_golden_verdict("config5", runs, 1000, 3100)[0]

"""cudapoa/utils.hpp row of SURVEY 8(f): get_multi_batch_sizes binning, window-file / FASTA readers, the CLI.

The binning rule is checked against a line-by-line Python restatement of the reference's rule
(cudapoa/src/utils.cu:62-143) on random capacities, plus the worked example of its source comment."""
import os
import random
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genomeworks_amd import cudapoa  # noqa: E402


def reference_binning(capacity, longest, reads, bins=None):
    """Restatement of utils.cu:62-143: returns [(max_len, max_reads, [group ids])]."""
    if bins is None:
        bins = [1 << j for j in range(20)]
    nb = len(bins)
    freq, blen, bnum, glist = [0] * nb, [0] * nb, [0] * nb, [[] for _ in range(nb)]
    for i, cap in enumerate(capacity):
        for j in range(nb):
            if cap <= bins[j] or j == nb - 1:
                freq[j] += 1
                glist[j].append(i)
                blen[j] = max(blen[j], longest[i])
                bnum[j] = max(bnum[j], reads[i])
                break
    out = []
    for j in range(nb):
        if freq[j] > 0:
            merged = list(glist[j])
            for k in range(j + 1, nb):
                if freq[k] > 0:
                    if bins[j] >= freq[k]:
                        merged += glist[k]
                        freq[k] = 0
                    else:
                        break
            out.append((blen[j], bnum[j], merged))
    return out


def check(capacity, longest, reads, bins=None):
    cfgs, groups = cudapoa.bin_poa_groups(capacity, longest, reads, band_width=256, band_mode="static_band", bins_capacity=bins)
    ref = reference_binning(capacity, longest, reads, bins)
    assert len(cfgs) == len(ref)
    for cfg, ids, (mlen, mreads, rids) in zip(cfgs, groups, ref):
        assert ids == rids
        assert cfg["max_sequence_size"] == mlen and cfg["max_sequences_per_poa"] == mreads
        assert cfg["alignment_band_width"] == 256 and cfg["band_mode"] == 1
    assert sorted(i for g in groups for i in g) == list(range(len(capacity)))  # every group in exactly one batch


def test_binning_worked_example():
    # utils.cu:100-110: 10 groups of capacity 64 (len 5120), 51 of capacity 128 (len 3604) -> one batch takes both bins
    capacity = [64] * 10 + [128] * 51
    longest = [5120] * 10 + [3604] * 51
    reads = [8] * 61
    cfgs, groups = cudapoa.bin_poa_groups(capacity, longest, reads, band_mode="static_band")
    assert len(cfgs) == 1 and len(groups[0]) == 61 and cfgs[0]["max_sequence_size"] == 5120
    check(capacity, longest, reads)


def test_binning_random_vs_reference_rule():
    rng = random.Random(7)
    for _ in range(50):
        n = rng.randint(1, 60)
        capacity = [rng.choice([1, 2, 3, 5, 17, 64, 100, 300, 1000, 5000, 10 ** 6]) for _ in range(n)]
        longest = [rng.randint(10, 5000) for _ in range(n)]
        reads = [rng.randint(1, 50) for _ in range(n)]
        check(capacity, longest, reads)
        check(capacity, longest, reads, bins=[2, 10, 50, 400])


def test_window_file_reader_and_resize(tmp_path):
    p = tmp_path / "w.txt"
    p.write_text("2\nACGT\nACGA\n3\nTTTT\nTTTA\nTTAA\n")
    w = cudapoa.parse_cudapoa_file(str(p))
    assert w == [["ACGT", "ACGA"], ["TTTT", "TTTA", "TTAA"]]
    assert cudapoa.parse_cudapoa_file(str(p), 1) == [["ACGT", "ACGA"]]
    # fewer windows than requested: the windows read are repeated in order
    assert cudapoa.parse_cudapoa_file(str(p), 5) == [w[0], w[1], w[0], w[1], w[0]]
    with pytest.raises(RuntimeError):
        cudapoa.parse_cudapoa_file(str(tmp_path / "missing.txt"))


def test_fasta_reader(tmp_path):
    a = tmp_path / "a.fa"
    b = tmp_path / "b.fa"
    a.write_text(">r0 first\nACGT\nAC\n>r1\nGGGG\n\n")
    b.write_text(">x\r\nTTTT\r\n")
    assert cudapoa.parse_fasta_files([str(a), str(b)]) == [["ACGTAC", "GGGG"], ["TTTT"]]
    assert cudapoa.parse_fasta_files([str(a), str(b)], 3) == [["ACGTAC", "GGGG"], ["TTTT"], ["ACGTAC", "GGGG"]]
    with pytest.raises(RuntimeError):
        cudapoa.parse_fasta_files([str(tmp_path / "nope.fa")])


def _cli():
    return os.path.join(ROOT, "genomeworks_amd", "bin", "cudapoa")


def test_cli_argument_validation(tmp_path):
    exe = _cli()
    assert os.path.exists(exe)
    r = subprocess.run([exe, "-h"], capture_output=True, text=True)
    assert r.returncode == 0 and "Usage: cudapoa" in r.stderr and "--band-mode" in r.stderr
    p = tmp_path / "w.txt"
    p.write_text("1\nACGT\n")
    for args, msg in ((["-i", str(p), "-b", "7"], "band-mode must be"), (["-i", str(p), "-M", "0"], "max-groups cannot be 0"),
                      (["-i", str(p), "-n", "3"], "mismatch score must be non-positive"),
                      (["-i", str(tmp_path / "none")], "Invalid input file")):
        r = subprocess.run([exe] + args, capture_output=True, text=True)
        assert r.returncode != 0 and msg in r.stderr


@pytest.mark.gpu
def test_get_multi_batch_sizes_on_device():
    groups = [["A" * 1000] * 8] * 5 + [["A" * 300] * 4] * 40
    cfgs, ids = cudapoa.get_multi_batch_sizes(groups, band_mode="static_band")
    assert sorted(i for g in ids for i in g) == list(range(45))
    assert cfgs[0]["max_sequence_size"] == 1000 and cfgs[0]["max_sequences_per_poa"] == 8


@pytest.mark.gpu
def test_cli_matches_batch_api(tmp_path):
    from genomeworks_amd import synthetic
    windows = [[r.decode() for r in synthetic.generate_window(4000 + w, 200, 8, 8, 4, 4)] for w in range(6)]
    p = tmp_path / "windows.txt"
    p.write_text("".join("%d\n%s\n" % (len(w), "\n".join(w)) for w in windows))
    b = cudapoa.CudaPoaBatch(8, 256, 1 << 30, band_mode="static_band", alignment_band_width=128)
    for w in windows:
        assert b.add_poa_group(w)[0] == 0
    b.generate_poa()
    cons, _, status = b.get_consensus()
    assert all(s == 0 for s in status)
    dot = tmp_path / "g.dot"
    r = subprocess.run([_cli(), "-i", str(p), "-b", "1", "-w", "128", "-d", str(dot)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == cons
    assert "Processed groups 0 - 5 (batch 0)" in r.stderr
    assert dot.read_text().count("digraph") == 6
    # MSA mode prints one row per read
    r = subprocess.run([_cli(), "-i", str(p), "-b", "1", "-w", "128", "-a", "-M", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert len(r.stdout.split()) == 16


def test_size_class_plan_is_a_partition_with_geometric_shapes():
    """cudapoa::plan_size_classes (host-only): windows binned by their longest read into classes (L / 2^(k+1), L / 2^k],
    one BatchConfig per class sized by the class's own largest member."""
    from genomeworks_amd import cudapoa
    lengths = [30000, 29000, 15001, 15000, 14000, 7600, 7000, 3000, 2999, 400, 90]
    reads = [8, 32, 10, 12, 9, 31, 8, 20, 8, 8, 3]
    windows = [["A" * n] + ["C" * max(n - 7, 1)] * (r - 1) for n, r in zip(lengths, reads)]
    plan = cudapoa.SizeClassPlan(windows, msa_flag=True, adaptive_storage_factor=4.0)
    assert sorted(w for g in plan.groups for w in g) == list(range(len(windows)))
    # L = 30000: (15000, 30000], (7500, 15000], (3750, 7500], (1875, 3750], (937.5, 1875] (empty here), everything below
    assert plan.groups == [[0, 1, 2], [3, 4, 5], [6], [7, 8], [9, 10]]
    for cfg, g, b in zip(plan.configs, plan.groups, plan.bytes_per_window):
        assert cfg["max_sequence_size"] == max(max(lengths[w] for w in g), 256)
        assert cfg["max_sequences_per_poa"] == max(reads[w] for w in g)
        assert cfg["max_nodes_per_graph"] == (3 * cfg["max_sequence_size"] + 3) // 4 * 4 and cfg["matrix_sequence_dimension"] == 4 * 264
        assert b > 0
    sizes = [b for b in plan.bytes_per_window]
    assert sizes == sorted(sizes, reverse=True)
    assert plan.total_bytes == sum(len(g) * b for g, b in zip(plan.groups, plan.bytes_per_window))
    plan.keep([1, 6, 10])
    assert plan.groups == [[1], [], [6], [], [10]] and plan.total_bytes == sizes[0] + sizes[2] + sizes[4]
    assert cudapoa.SizeClassPlan([], msa_flag=False).configs == []


def test_size_class_admission_gates():
    """cudapoa::size_class_admission_gates (host-only): classes are admitted in plan order while their windows fit 1.25 x the
    compute units; the classes of the next group wait for the last class of the group before. The long-read set of
    BASELINE configs[3] has 157 / 151 / 160 / 130 windows per class: on 256 units the two heavy classes start at once and
    the two light ones wait for the second."""
    from genomeworks_amd import cudapoa
    def plan_of(counts):
        lengths = []
        for k, c in enumerate(counts):
            lengths += [30000 >> k] * c
        return cudapoa.SizeClassPlan([["A" * n, "C" * (n - 1)] for n in lengths], msa_flag=True, adaptive_storage_factor=4.0)
    plan = plan_of([157, 151, 160, 130])
    assert [len(g) for g in plan.groups] == [157, 151, 160, 130]
    assert plan.admission_gates(256) == [-1, -1, 1, 1]
    assert plan.admission_gates(1024) == [-1, -1, -1, -1]
    assert plan.admission_gates(64) == [-1, 0, 1, 2]      # nothing fits beside anything: a chain
    plan.keep(list(range(157)) + list(range(157 + 151, 157 + 151 + 160)))
    assert [len(g) for g in plan.groups] == [157, 0, 160, 0]
    assert plan.admission_gates(256) == [-1, -1, -1, -1]  # 317 windows fit 320; empty classes gate nothing
    assert plan.admission_gates(200) == [-1, -1, 0, -1]


"""The alignment stage of cudamapper as a consumer of cudaaligner: `align_overlaps` tool,
PAF in -> PAF with cg:Z: CIGARs out (reference: cudamapper/src/main.cu:54-187, utils.cpp:41-124)."""
import os
import random
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "genomeworks_amd", "bin", "align_overlaps")
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def mutate(rng, s, rate):
    out = []
    for c in s:
        r = rng.random()
        if r < rate / 3:
            continue                                   # deletion
        if r < 2 * rate / 3:
            out.append(rng.choice("ACGT"))             # substitution
            continue
        out.append(c)
        if r < rate:
            out.append(rng.choice("ACGT"))             # insertion
    return "".join(out)


def make_case(tmp_path, n_reads=14, seed=11):
    """Reads sampled from a random genome (some stored reverse-complemented) and every pairwise overlap of their
    genome intervals as PAF records with approximate read coordinates."""
    rng = random.Random(seed)
    genome = "".join(rng.choice("ACGT") for _ in range(9000))
    reads = []
    for i in range(n_reads):
        a = rng.randrange(0, 7000)
        b = a + rng.randrange(900, 2000)
        seq = mutate(rng, genome[a:b], 0.06)
        rev = i % 3 == 2
        reads.append(dict(name="read_%d desc" % i, a=a, b=b, seq=revcomp(seq) if rev else seq, rev=rev))
    fasta = tmp_path / "reads.fasta"
    with open(fasta, "w") as f:
        for r in reads:
            f.write(">%s\n" % r["name"])
            for k in range(0, len(r["seq"]), 70):
                f.write(r["seq"][k:k + 70] + "\n")
    lines, expect = [], []
    for i, q in enumerate(reads):
        for j, t in enumerate(reads):
            if i >= j:
                continue
            lo, hi = max(q["a"], t["a"]), min(q["b"], t["b"])
            if hi - lo < 300:
                continue

            def span(r):  # read coordinates of genome interval [lo, hi), scaled, on the stored strand
                n, g = len(r["seq"]), r["b"] - r["a"]
                s, e = (lo - r["a"]) * n // g, (hi - r["a"]) * n // g
                return (n - e, n - s) if r["rev"] else (s, e)
            qs, qe = span(q)
            ts, te = span(t)
            strand = "-" if q["rev"] != t["rev"] else "+"
            lines.append("\t".join(map(str, ["read_%d" % i, len(q["seq"]), qs, qe, strand, "read_%d" % j, len(t["seq"]), ts, te,
                                             7, max(qe - qs, te - ts), 255])))
            tsub = t["seq"][ts:te]
            expect.append((q["seq"][qs:qe], revcomp(tsub) if strand == "-" else tsub))
    paf = tmp_path / "overlaps.paf"
    paf.write_text("\n".join(lines) + "\n")
    return str(fasta), str(paf), lines, expect


def test_tool_is_built_and_rejects_bad_input(tmp_path):
    assert os.access(TOOL, os.X_OK), "build it with __graft_entry__.build()"
    r = subprocess.run([TOOL], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage: align_overlaps" in r.stderr
    fasta, paf, lines, _ = make_case(tmp_path, n_reads=4)
    bad = tmp_path / "bad.paf"
    bad.write_text(lines[0].replace("read_0", "nobody", 1) + "\n")
    r = subprocess.run([TOOL, fasta, fasta, str(bad)], capture_output=True, text=True)  # fails before any device call
    assert r.returncode == 1 and "unknown read name" in r.stderr
    bad.write_text("read_0\t10\tx\n")
    r = subprocess.run([TOOL, fasta, fasta, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "malformed PAF line 1" in r.stderr
    r = subprocess.run([TOOL, str(tmp_path / "missing.fasta"), fasta, paf], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open FASTA file" in r.stderr


def cigar_lengths(cigar):
    q = t = 0
    for n, op in re.findall(r"(\d+)([MIDX=])", cigar):
        n = int(n)
        if op in "M=X":
            q += n
            t += n
        elif op == "I":  # cudaaligner's convention (cudaaligner.hpp:47-53): insertion = absent in query, present in target
            t += n
        else:
            q += n
    return q, t


@pytest.mark.gpu
def test_paf_cigars_equal_direct_aligner_calls_for_any_engine_count(tmp_path):
    from genomeworks_amd import cudaaligner
    fasta, paf, lines, expect = make_case(tmp_path)
    assert len(lines) > 20 and any("\t-\t" in l for l in lines)
    outs = []
    for engines, batch in ((1, 0), (3, 4), (2, 1000)):
        r = subprocess.run([TOOL, "-a", str(engines)] + (["-b", str(batch)] if batch else []) + [fasta, fasta, paf],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "Aligning %d overlaps" % len(lines) in r.stderr
        outs.append(r.stdout)
    assert outs[0] == outs[1] == outs[2]          # record order and CIGARs do not depend on engines / batch size
    got = outs[0].strip().split("\n")
    assert len(got) == len(lines)
    # the same pairs straight through the Python mirror of the Aligner interface (default create_aligner)
    mq, mt = max(len(q) for q, _ in expect), max(len(t) for _, t in expect)
    batch = cudaaligner.CudaAlignerBatch(mq, mt, len(expect))
    for q, t in expect:
        assert batch.add_alignment(q, t) == 0
    batch.align_all()
    ref = batch.get_alignments()
    for line, src, (q, t), a in zip(got, lines, expect, ref):
        cols = line.split("\t")
        assert cols[:9] == src.split("\t")[:9] and cols[11] == "255"
        assert cols[12].startswith("cg:Z:")
        cigar = cols[12][5:]
        assert cigar == a.cigar
        assert cigar_lengths(cigar) == (len(q), len(t))


@pytest.mark.gpu
def test_sam_records_carry_the_paf_cigars(tmp_path):
    """`align_overlaps -S` (cudamapper's -S / print_sam, cudamapper/src/utils.cpp:190-318, without htslib): one @SQ line per
    distinct target read, the @PG line, and per overlap a record whose CIGAR is the `cg:Z:` tag of the PAF output, whose flag
    says the strand, whose RNAME / POS are the target read and the 1-based target start, with the whole query sequence."""
    fasta, paf, lines, _expect = make_case(tmp_path)
    p = subprocess.run([TOOL, "-a", "2", fasta, fasta, paf], capture_output=True, text=True)
    s = subprocess.run([TOOL, "-a", "2", "-S", fasta, fasta, paf], capture_output=True, text=True)
    assert p.returncode == 0 and s.returncode == 0, (p.stderr, s.stderr)
    paf_rows = [l.split("\t") for l in p.stdout.strip().split("\n")]
    sam = s.stdout.strip().split("\n")
    header = [l for l in sam if l.startswith("@")]
    records = [l.split("\t") for l in sam if not l.startswith("@")]
    targets_in_order = []
    for row in paf_rows:
        if row[5] not in targets_in_order:
            targets_in_order.append(row[5])
    assert [h.split("\t")[1][3:] for h in header if h.startswith("@SQ")] == targets_in_order
    assert sum(1 for h in header if h.startswith("@PG\tID:cudamapper")) == 1 and header[-1].startswith("@PG")
    assert len(records) == len(paf_rows)
    seqs = {}
    name = None
    for line in open(fasta):
        if line.startswith(">"):
            name = line[1:].split()[0]
            seqs[name] = ""
        else:
            seqs[name] += line.strip()
    for rec, row in zip(records, paf_rows):
        assert len(rec) == 11
        assert rec[0] == row[0] and rec[2] == row[5]
        assert rec[1] == ("16" if row[4] == "-" else "0")
        assert int(rec[3]) == int(row[7]) + 1 and rec[4] == "255"
        # the PAF CIGAR between soft clips for the unaligned ends of the read, with I / D in SAM's sense (cudaaligner names them
        # from the other sequence's side); on the reverse strand SEQ is the reverse complement and the clips swap sides
        qlen, qs, qe = int(row[1]), int(row[2]), int(row[3])
        head, tail = (qs, qlen - qe) if row[4] == "+" else (qlen - qe, qs)
        swapped = row[12][5:].translate(str.maketrans("ID", "DI"))
        assert rec[5] == ("%dS" % head if head else "") + swapped + ("%dS" % tail if tail else "")
        seq = seqs[row[0]] if row[4] == "+" else seqs[row[0]][::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))
        assert rec[6:9] == ["*", "0", "0"] and rec[9] == seq and rec[10] == "*"
        # SAM validity: the operators that consume the query (M I S = X) add up to len(SEQ), those that consume the reference
        # (M D N = X) to the aligned target span
        ops = re.findall(r"(\d+)([MIDNSHP=X])", rec[5])
        assert "".join(n + o for n, o in ops) == rec[5]
        assert sum(int(n) for n, o in ops if o in "MIS=X") == len(rec[9])
        assert sum(int(n) for n, o in ops if o in "MDN=X") == int(row[8]) - int(row[7])

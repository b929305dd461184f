"""Writes tests/golden/reference_simt_alignments.json.gz: pairs of sequences and what the REFERENCE ITSELF answers for them -- its
cudaaligner library (aligner*.cpp, myers_gpu.cu, hirschberg_myers_gpu.cu, ukkonen_gpu.cu) compiled from /root/reference where it
lies and run on the CPU by the SIMT emulator of oracle/simt (`make -C oracle -f Makefile.ref ref_cudaaligner_simt`,
tests/ref_cudaaligner.py). Batches: the default aligner (create_aligner(max_query, max_target, n): Hirschberg + Myers) on short
pairs, on pairs of 1.2, 3 and 5 kbp (the Hirschberg recursion above the Myers threshold) and on extreme shapes; the banded Myers
aligner (create_aligner(global_alignment, max_bandwidth, ...)) at five band widths, with pairs the band rejects or only
approximates; the Ukkonen and the full-matrix Myers classes. The aligner oracles (tests/test_reference_simt.py, CPU) and the HIP
aligners (tests/test_gpu_aligner.py, GPU) are compared with this file.
usage: python tests/golden/make_reference_simt_alignments.py   (about one minute)"""
import gzip
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
OUT = os.path.join(HERE, "reference_simt_alignments.json.gz")


def mutate(rng, s, k):
    s = list(s)
    for _ in range(k):
        i = rng.randrange(len(s))
        op = rng.random()
        if op < 0.4:
            s[i] = rng.choice("ACGT")
        elif op < 0.7:
            s.insert(i, rng.choice("ACGT"))
        elif len(s) > 1:
            del s[i]
    return "".join(s)


def random_pairs(rng, n, lengths, max_len=None):
    pairs = []
    while len(pairs) < n:
        L = rng.choice(lengths)
        q = "".join(rng.choice("ACGT") for _ in range(L))
        t = mutate(rng, q, rng.choice([0, 1, 3, L // 20 + 1, L // 6 + 1]))
        if rng.random() < 0.3:
            q, t = t, q
        if max_len is None or (len(q) <= max_len and len(t) <= max_len):
            pairs.append([q, t])
    return pairs


def batches():
    rng = random.Random(20260928)
    out = []
    out.append(dict(kind="default", pairs=random_pairs(rng, 40, [1, 2, 5, 31, 32, 33, 64, 65, 100, 199, 333, 400])))
    out.append(dict(kind="default", pairs=[["A", "T"], ["ACGT" * 30, "TGCA" * 30], ["A" * 200, "A" * 200], ["A" * 150, "A" * 20], ["ACGTT", "ACGTTACGTTACGTTAACCGGTTACGT" * 6],
                                           ["GATTACA" * 20, "GATTACA" * 19 + "GATTTACA"]]))
    for L in (1200, 3000, 5000):
        q = "".join(rng.choice("ACGT") for _ in range(L))
        out.append(dict(kind="default", pairs=[[q, mutate(rng, q, L // 12)]]))
    for bw in (7, 31, 64, 256, 1024):
        pairs = random_pairs(rng, 24, [1, 5, 33, 64, 100, 150, 300, 700])
        pairs += [["ACGT" * 10, "ACGT" * 30], ["A" * 90, "C" * 90]]  # a length difference the narrow bands reject; nothing in common
        out.append(dict(kind="banded", max_bandwidth=bw, pairs=pairs))
    out.append(dict(kind="ukkonen", pairs=random_pairs(rng, 24, [2, 5, 31, 33, 64, 100, 199, 250], 300)))
    out.append(dict(kind="myers", pairs=random_pairs(rng, 24, [1, 2, 5, 31, 33, 64, 100, 199, 250], 300)))
    # characters outside ACGT ('N', lower case, IUPAC codes): the bit-vector kernels see the query through four patterns and the target
    # through its pattern index (c >> 1) & 3; the Hirschberg kernel's single-character leaf and the Ukkonen kernel compare the
    # characters themselves. (The reference's debug build asserts ACGT-only targets, myers_gpu.cu:214; its release build does this.)
    # The HIP default and full-Myers aligners have not met such input on a GPU yet: those batches are marked "gpu": False.
    rng2 = random.Random(77)

    def odd_pairs(n):
        alphabet, pairs = "ACGTNacgtRY", []
        for _ in range(n):
            L = rng2.choice([1, 2, 7, 33, 64, 130, 260])
            q = [rng2.choice(alphabet) for _ in range(L)]
            t = list(q)
            for _ in range(rng2.choice([0, 1, 3, L // 10 + 1])):
                i, op = rng2.randrange(len(t)), rng2.random()
                if op < 0.4:
                    t[i] = rng2.choice(alphabet)
                elif op < 0.7:
                    t.insert(i, rng2.choice(alphabet))
                elif len(t) > 1:
                    del t[i]
            pairs.append(["".join(q), "".join(t)])
        return pairs
    out.append(dict(kind="banded", max_bandwidth=256, pairs=odd_pairs(20)))
    out.append(dict(kind="default", pairs=odd_pairs(30), gpu=False))
    out.append(dict(kind="myers", pairs=odd_pairs(20), gpu=False))
    out.append(dict(kind="ukkonen", pairs=[p for p in odd_pairs(40) if abs(len(p[0]) - len(p[1])) <= 2][:16], gpu=False))
    return out


def rle(states):
    """'12=1X3I2D' with = match, X mismatch, I insertion, D deletion (AlignmentState 0..3)"""
    out, i = [], 0
    while i < len(states):
        j = i
        while j < len(states) and states[j] == states[i]:
            j += 1
        out.append("%d%s" % (j - i, "=XID"[states[i]]))
        i = j
    return "".join(out)


def unrle(text):
    import re
    out = []
    for n, c in re.findall(r"(\d+)([=XID])", text):
        out += ["=XID".index(c)] * int(n)
    return out


def run_reference(b):
    import ref_cudaaligner as R
    res = R.align([tuple(p) for p in b["pairs"]], b["kind"], max_bandwidth=b.get("max_bandwidth"))
    return [dict(add_status=r["add_status"], status=r["status"], optimal=r["optimal"], alignment=None if r["states"] is None else rle(r["states"])) for r in res]


def main():
    rows = []
    for i, b in enumerate(batches()):
        rows.append(dict(batch=b, reference=run_reference(b)))
        print(i, b["kind"], b.get("max_bandwidth"), len(b["pairs"]), "pairs", flush=True)
    with gzip.GzipFile(OUT, "wb", mtime=0, compresslevel=9) as f:
        f.write(json.dumps(dict(generator="tests/golden/make_reference_simt_alignments.py", source="the reference's cudaaligner library on oracle/simt",
                                batches=rows), separators=(",", ":")).encode())
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()

"""Writes tests/golden/reference_simt_windows.json.gz: POA windows and what the REFERENCE ITSELF answers for them -- its cudapoa
library (batch.cu, cudapoa_kernels.cuh ...) compiled from /root/reference where it lies and run on the CPU by the SIMT emulator
of oracle/simt (`make -C oracle -f Makefile.ref ref_cudapoa_simt`, tests/ref_cudapoa.py). Every band mode, consensus and MSA
output, match / mismatch / gap scores other than the defaults, per-base weights, reads that overflow the graph or the band.
The oracle (tests/test_reference_simt.py, CPU) and the HIP path (tests/test_gpu_poa.py, GPU) are compared with this file; on
a machine that has the reference, test_reference_simt.py also regenerates a sample and runs fresh random windows.
usage: python tests/golden/make_reference_simt_goldens.py   (about two minutes)"""
import gzip
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
OUT = os.path.join(HERE, "reference_simt_windows.json.gz")
BAND_MODES = {0: "full_band", 1: "static_band", 2: "adaptive_band", 3: "static_band_traceback", 4: "adaptive_band_traceback"}


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


def cases():
    """Deterministic list of dict(config..., reads, weights)."""
    rng = random.Random(20260927)
    out = []
    for mode in range(5):
        for k in range(14):
            L = rng.choice([12, 40, 90, 150, 260, 400])
            n = rng.randint(1, 12)
            div = rng.choice([0, 25, 12, 7, 4])
            base = "".join(rng.choice("ACGT") for _ in range(L))
            reads = [mutate(rng, base, L // div if div else 0) for _ in range(n)]
            if k % 5 == 4:  # reads of very different lengths: band shifts, adaptive reruns, short prefixes
                reads = [r[:max(4, int(len(r) * rng.uniform(0.3, 1.0)))] for r in reads]
            c = dict(max_seq=512, max_seqs=16, band_width=rng.choice([128, 256]), band_mode=mode, output_mask=1 if k % 2 == 0 else 2,
                     gap=-8, mismatch=-6, match=8, reads=reads, weights=None)
            if k % 7 == 3:
                c.update(gap=-4, mismatch=-3, match=5)
            if k % 6 == 5:
                c["weights"] = [[rng.randint(1, 40) for _ in r] for r in reads]
            out.append(c)
    # limits: a graph that outgrows max_nodes_per_graph (3 x max_seq), a window of the maximum number of reads
    rng2 = random.Random(7)
    far = ["".join(rng2.choice("ACGT") for _ in range(60)) for _ in range(8)]
    out.append(dict(max_seq=128, max_seqs=8, band_width=128, band_mode=1, output_mask=1, gap=-8, mismatch=-6, match=8, reads=far, weights=None))
    base = "".join(rng2.choice("ACGT") for _ in range(120))
    out.append(dict(max_seq=256, max_seqs=16, band_width=128, band_mode=2, output_mask=2, gap=-8, mismatch=-6, match=8,
                    reads=[mutate(rng2, base, 9) for _ in range(16)], weights=None))
    # graphs that outgrow max_nodes_per_graph (unrelated reads, max_seq 128 -> 384 nodes), in banded modes and in the full-band mode
    for mode in (1, 0, 4):
        unrelated = ["".join(rng2.choice("ACGT") for _ in range(120)) for _ in range(16)]
        out.append(dict(max_seq=128, max_seqs=16, band_width=128, band_mode=mode, output_mask=1, gap=-8, mismatch=-6, match=8, reads=unrelated, weights=None))
    # a read longer than max_sequence_size is refused by add_poa_group (its status says so), the rest of the window is processed
    base = "".join(rng2.choice("ACGT") for _ in range(100))
    reads = [mutate(rng2, base, 6) for _ in range(5)]
    reads.insert(2, base + base)
    out.append(dict(max_seq=128, max_seqs=8, band_width=128, band_mode=1, output_mask=1, gap=-8, mismatch=-6, match=8, reads=reads, weights=None))
    # more reads than max_sequences_per_poa
    out.append(dict(max_seq=128, max_seqs=4, band_width=128, band_mode=2, output_mask=2, gap=-8, mismatch=-6, match=8,
                    reads=[mutate(rng2, base, 5) for _ in range(7)], weights=None))
    # traceback-buffer modes with a predecessor window of 8 and 16 rows and indel-heavy reads: first predecessors at a multiple of
    # the window alias the row's own ring slot (cudapoa_nw_tb_banded.cuh:456; oracle/poa_nw_tb.inc on the pass order). The HIP
    # path has not met these windows on a GPU yet: the GPU test of this file leaves out the cases marked "gpu": False.
    rng3 = random.Random(88)

    def indels(s, k):
        s = list(s)
        for _ in range(k):
            i = rng3.randrange(len(s))
            op = rng3.random()
            if op < 0.3:
                s[i] = rng3.choice("ACGT")
            elif op < 0.65:
                s[i:i] = [rng3.choice("ACGT") for _ in range(rng3.choice([1, 1, 2, 8, 30]))]
            else:
                del s[i:i + rng3.choice([1, 1, 2, 8, 30])]
        return ("".join(s) or "A")[:500]
    for k in range(10):
        L = rng3.choice([100, 200, 350])
        base = "".join(rng3.choice("ACGT") for _ in range(L))
        out.append(dict(max_seq=512, max_seqs=16, band_width=128, band_mode=3 + k % 2, output_mask=1 + k % 2, gap=-8, mismatch=-6, match=8,
                        reads=[indels(base, L // 8) for _ in range(rng3.randint(4, 10))], weights=None, max_pred=8 if k < 7 else 16, gpu=False))
    return out


def run_reference(c):
    import ref_cudapoa as R
    with R.RefBatch(c["max_seq"], c["max_seqs"], c["band_width"], c["band_mode"], gap=c["gap"], mismatch=c["mismatch"], match=c["match"],
                    output_mask=c["output_mask"], max_pred=c.get("max_pred", 0)) as b:
        add_status, read_status = b.add_poa_group(c["reads"], c["weights"])
        # (the eight fields the reference's BatchConfig constructor derived: what a caller of the explicit constructor passes)
        res = dict(add_status=add_status, read_status=read_status, batch_config=R.config(c["max_seq"], c["max_seqs"], c["band_width"], c["band_mode"], max_pred=c.get("max_pred", 0)))
        if add_status == 0:
            b.generate_poa()
            if c["output_mask"] & 1:
                res.update(b.get_consensus()[0])
            else:
                res.update(b.get_msa()[0])
        return res


def main():
    rows = []
    for i, c in enumerate(cases()):
        rows.append(dict(case=c, reference=run_reference(c)))
        print(i, BAND_MODES[c["band_mode"]], "reads", len(c["reads"]), "status", rows[-1]["reference"].get("status"), flush=True)
    with gzip.GzipFile(OUT, "wb", mtime=0, compresslevel=9) as f:
        f.write(json.dumps(dict(generator="tests/golden/make_reference_simt_goldens.py", source="the reference's cudapoa library on oracle/simt",
                                windows=rows), separators=(",", ":")).encode())
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()

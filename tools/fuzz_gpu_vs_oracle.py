#!/usr/bin/env python3
"""Randomised parity sweep on a GPU box: the HIP path against the CPU oracle over configurations the fixed tests do not enumerate --
every band mode x band width, consensus and MSA, scores inside the reference's limits, 2-32 reads of 40-1000 bases at 0-30 %
divergence, graphs that outgrow max_nodes_per_graph, reads the batch refuses -- and the aligners over random pairs (banded Myers at
several max_bandwidths, the default aligner, Ukkonen, full Myers). TEST INFRASTRUCTURE (the oracle is the checker); prints one JSON
line and exits non-zero on the first differences.

  python tools/fuzz_gpu_vs_oracle.py [seed=1] [poa_cases=300] [aligner_cases=120]
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_aligner as A  # noqa: E402
import oracle_poa as O  # noqa: E402
from genomeworks_amd import cudaaligner, cudapoa, synthetic  # noqa: E402

BAND = {"full_band": 0, "static_band": 1, "adaptive_band": 2, "static_band_traceback": 3, "adaptive_band_traceback": 4}


def poa_case(rng, k):
    mode = rng.choice(list(BAND))
    max_seq = rng.choice([256, 512, 1024])
    band = rng.choice([b for b in (128, 256, 384, 512) if b <= max_seq]) if mode != "full_band" else 256
    backbone = rng.randint(40, max_seq - 24 - 8)
    n_reads = rng.randint(2, 32)
    div = rng.choice([0.0, 0.02, 0.1, 0.2, 0.3])
    mut, ins, dele = (int(backbone * div * f) for f in rng.choice([(0.4, 0.3, 0.3), (0.1, 0.45, 0.45), (1.0, 0.0, 0.0)]))
    ins = min(ins, 24)
    reads = [r.decode() for r in synthetic.generate_window(rng.randint(1, 1 << 30), backbone, n_reads, mut, ins, dele)]
    reads = [r for r in reads if 0 < len(r) <= max_seq] or ["ACGT"]
    msa = rng.random() < 0.35
    match, mismatch, gap = rng.choice([(8, -6, -8), (8, -6, -8), (5, -4, -8), (2, -3, -2), (10, -1, -12)])
    nodes = rng.choice([3 * max_seq, 3 * max_seq, max_seq])  # sometimes a graph that may outgrow its limit (the smallest the batch accepts)
    return dict(mode=mode, band=band, max_seq=max_seq, reads=reads, msa=msa, match=match, mismatch=mismatch, gap=gap, nodes=nodes)


def run_poa(c):
    b = cudapoa.CudaPoaBatch(32, c["max_seq"], 2 << 30, output_type="msa" if c["msa"] else "consensus", band_mode=c["mode"],
                             alignment_band_width=c["band"], max_nodes_per_graph=c["nodes"], gap_score=c["gap"], mismatch_score=c["mismatch"],
                             match_score=c["match"])
    st, _ = b.add_poa_group(c["reads"])
    cfg = O.make_cfg(c["max_seq"], 32, c["band"], BAND[c["mode"]], gap=c["gap"], mismatch=c["mismatch"], match=c["match"],
                     output_mask=2 if c["msa"] else 1)
    cfg.max_nodes_per_graph = c["nodes"]
    cfg.matrix_sequence_dimension = b.batch_size.matrix_sequence_dimension
    cfg.max_banded_pred_distance = b.batch_size.max_banded_pred_distance
    O.lib().poa_cfg_select_types(cfg)
    with O.Workspace(cfg) as ws:
        ref = ws.process(c["reads"])
    if st != 0:
        return False  # (every read fits the batch: nothing to refuse)
    b.generate_poa()
    if c["msa"]:
        rows, status = b.get_msa()
        return status[0] == ref["status"] and (status[0] != 0 or rows[0] == list(ref["msa"]))
    cons, cov, status = b.get_consensus()
    return status[0] == ref["status"] and (status[0] != 0 or (cons[0] == ref["consensus"] and list(cov[0]) == list(ref["coverage"])))


def aligner_case(rng):
    kind = rng.choice(["banded", "banded", "default", "ukkonen", "myers"])
    n = rng.randint(1, 40)
    L = rng.choice([20, 150, 400, 1000, 2500])
    div = rng.choice([0.0, 0.03, 0.1, 0.3])
    pairs = [(q.decode(), t.decode()) for q, t in
             synthetic.generate_pairs(rng.randint(1, 1 << 30), n, L, int(L * div * 0.4), int(L * div * 0.3), int(L * div * 0.3))]
    return dict(kind=kind, pairs=pairs, max_bandwidth=rng.choice([32, 150, 512, 1024]))


def run_aligner(c):
    pairs, kind = c["pairs"], c["kind"]
    mx = max(max(len(q), len(t)) for q, t in pairs) + 1
    if kind == "banded":
        al = cudaaligner.CudaAlignerBatch(max_bandwidth=c["max_bandwidth"], max_device_memory_allocator_caching_size=4 << 30)
    elif kind == "default":
        al = cudaaligner.CudaAlignerBatch(mx, mx, len(pairs), max_device_memory_allocator_caching_size=4 << 30)
    else:
        al = cudaaligner.CudaAlignerBatch(mx, mx, len(pairs), algorithm=kind, max_device_memory_allocator_caching_size=8 << 30)
    for q, t in pairs:
        if al.add_alignment(q, t) != 0:
            return False
    al.align_all()
    for (q, t), r in zip(pairs, al.get_alignments()):
        if kind == "banded":
            ref = A.align(q, t, c["max_bandwidth"])
            if r.status != ref["status"] or (r.status == 0 and r.cigar_extended != ref["cigar_extended"]):
                return False
        else:
            ref = A.hirschberg(q, t, mx) if kind == "default" else (A.ukkonen(q, t, 100) if kind == "ukkonen" else A.myers_full(q, t))
            if r.status != ref["status"] or (r.status == 0 and list(r.alignment) != ref["states"]):
                return False
    return True


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n_poa = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    n_al = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    rng = random.Random(seed)
    bad_poa, bad_al, by_mode = [], [], {}
    for k in range(n_poa):
        c = poa_case(rng, k)
        by_mode[c["mode"]] = by_mode.get(c["mode"], 0) + 1
        try:
            ok = run_poa(c)
        except Exception as e:  # a refused configuration is an answer too, a crash is not
            ok = False
            c["error"] = repr(e)[:200]
        if not ok:
            bad_poa.append({k2: v for k2, v in c.items() if k2 != "reads"} | {"case": k, "n_reads": len(c["reads"]), "first_len": len(c["reads"][0])})
    for k in range(n_al):
        c = aligner_case(rng)
        try:
            ok = run_aligner(c)
        except Exception as e:
            ok = False
            c["error"] = repr(e)[:200]
        if not ok:
            bad_al.append({"case": k, "kind": c["kind"], "pairs": len(c["pairs"]), "max_bandwidth": c["max_bandwidth"], "error": c.get("error")})
    print(json.dumps({"seed": seed, "poa_cases": n_poa, "poa_by_mode": by_mode, "poa_differing": bad_poa[:10], "poa_differing_count": len(bad_poa),
                      "aligner_cases": n_al, "aligner_differing": bad_al[:10], "aligner_differing_count": len(bad_al)}))
    return 1 if (bad_poa or bad_al) else 0


if __name__ == "__main__":
    sys.exit(main())

"""Dry run of the two GPU tests that compare the HIP path with the reference fixtures: the device classes are replaced by doubles
that answer from the oracles, so the tests' own code (keys, indexing, comparisons) is exercised on the CPU."""
import sys, ctypes as C
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_poa as O, oracle_aligner as A
from genomeworks_amd import cudapoa, cudaaligner, _native

BAND = {"full_band": 0, "static_band": 1, "adaptive_band": 2, "static_band_traceback": 3, "adaptive_band_traceback": 4}

class FakePoa:
    @classmethod
    def from_batch_config(cls, max_seq, max_seqs, band, mode, mem, output_type="consensus", gap_score=-8, mismatch_score=-6, match_score=8, max_banded_pred_distance=0, **kw):
        self = cls()
        L = cudapoa._bind(_native.host())
        cfg = _native.PoaBatchConfig()
        assert L.gw_poa_batch_config_default(C.byref(cfg), max_seq, max_seqs, band, BAND[mode], 2.0, 3.0, max_banded_pred_distance) == 0
        self.batch_size = cfg
        self.o = O.make_cfg(max_seq, max_seqs, band, BAND[mode], gap=gap_score, mismatch=mismatch_score, match=match_score, output_mask=1 if output_type == "consensus" else 2, max_pred=max_banded_pred_distance)
        self.max_seq, self.max_seqs = max_seq, max_seqs
        return self
    def add_poa_group(self, reads, weights=None):
        st, kept = [], []
        for i, r in enumerate(reads):
            if len(r) > self.max_seq: st.append(2)
            elif len(kept) >= self.max_seqs: st.append(3)
            else: st.append(0); kept.append(i)
        self.reads = [reads[i] for i in kept]; self.weights = None if weights is None else [weights[i] for i in kept]
        return 0, st
    def generate_poa(self):
        with O.Workspace(self.o) as ws: self.res = ws.process(self.reads, self.weights)
    def get_consensus(self):
        r = self.res
        return [r.get("consensus", "")], [[int(x) for x in r.get("coverage", [])]], [r["status"]]
    def get_msa(self):
        return [self.res.get("msa", [])], [self.res["status"]]

class FakeAl:
    def __init__(self, *a, max_bandwidth=None, algorithm=None, **kw):
        self.kind = "banded" if max_bandwidth is not None else (algorithm or "default"); self.bw = max_bandwidth; self.pairs = []
        self.max_len = a[0] if a else None
    def add_alignment(self, q, t): self.pairs.append((q, t)); return 0
    def align_all(self): pass
    def get_alignments(self):
        out = []
        for q, t in self.pairs:
            if self.kind == "banded":
                r = A.align(q, t, self.bw); ok = r["status"] == 0
                states = [o for o, k in r["runs"] for _ in range(k)] if ok else []
                out.append(cudaaligner.CudaAlignment(q, t, "", "", 0 if ok else 1, bool(r["optimal"]) if ok else False, 0, states))
            else:
                r = A.hirschberg(q, t, self.max_len) if self.kind == "default" else (A.ukkonen(q, t, 100) if self.kind == "ukkonen" else A.myers_full(q, t))
                out.append(cudaaligner.CudaAlignment(q, t, "", "", 0, True, 0, list(r["states"])))
        return out

cudapoa.CudaPoaBatch = FakePoa
cudaaligner.CudaAlignerBatch = FakeAl
import test_gpu_poa, test_gpu_aligner
test_gpu_poa.test_hip_path_equals_the_reference_itself_on_the_simt_goldens()
print("POA fixture test: ok")
test_gpu_aligner.test_hip_aligners_equal_the_reference_itself_on_the_simt_goldens()
print("aligner fixture test: ok")

"""TEST INFRASTRUCTURE: ctypes binding of oracle/_ref/libref_cudaaligner_simt.so -- the REFERENCE's own cudaaligner library (its
CUDA sources compiled by g++ where they lie, kernels run on the CPU by oracle/simt/simt.hpp; `make -C oracle -f Makefile.ref
ref_cudaaligner_simt`). Only what checks the aligner oracles and writes tests/golden/reference_simt_alignments.json.gz uses it."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "_ref", "libref_cudaaligner_simt.so")
KINDS = {"default": 0, "ukkonen": 1, "myers": 2, "hirschberg_myers": 3}
_lib = None


def available():
    return os.path.exists(PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(PATH)
        L.ref_aligner_create.restype = C.c_void_p
        L.ref_aligner_create.argtypes = [C.c_int] * 4
        L.ref_aligner_create_banded.restype = C.c_void_p
        L.ref_aligner_create_banded.argtypes = [C.c_int, C.c_longlong]
        L.ref_aligner_destroy.argtypes = [C.c_void_p]
        L.ref_aligner_add.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.ref_aligner_run.argtypes = [C.c_void_p]
        for f in ("ref_aligner_states_length", "ref_aligner_is_optimal", "ref_aligner_alignment_status"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
        L.ref_aligner_states.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
        L.ref_aligner_states.restype = None
        _lib = L
    return _lib


def align(pairs, kind="default", max_query=None, max_target=None, max_bandwidth=None):
    """The reference's aligner on `pairs` [(query, target)]: kind "default" (create_aligner: Hirschberg + Myers), "ukkonen",
    "myers", "hirschberg_myers", or "banded" (create_aligner(global_alignment, max_bandwidth, ...): banded Myers).
    -> list of dict(add_status, status, optimal, states) in input order; pairs that add_alignment refused have states None."""
    L = lib()
    raw = [(q.encode() if isinstance(q, str) else bytes(q), t.encode() if isinstance(t, str) else bytes(t)) for q, t in pairs]
    if kind == "banded":
        h = L.ref_aligner_create_banded(max_bandwidth, 1 << 28)
    else:
        h = L.ref_aligner_create(KINDS[kind], max_query or max(len(q) for q, _ in raw), max_target or max(len(t) for _, t in raw), len(raw))
    if not h:
        raise RuntimeError("the reference's constructor threw")
    try:
        out, kept = [], []
        for q, t in raw:
            st = L.ref_aligner_add(h, q, len(q), t, len(t))
            out.append(dict(add_status=st, status=None, optimal=None, states=None))
            if st == 0:
                kept.append(len(out) - 1)
        n = L.ref_aligner_run(h)
        if n != len(kept):
            raise RuntimeError("align_all / sync_alignments of the reference: %d" % n)
        for k, i in enumerate(kept):
            ln = L.ref_aligner_states_length(h, k)
            buf = C.create_string_buffer(max(ln, 1))
            L.ref_aligner_states(h, k, buf)
            out[i].update(status=L.ref_aligner_alignment_status(h, k), optimal=bool(L.ref_aligner_is_optimal(h, k)), states=list(buf.raw[:ln]))
        return out
    finally:
        L.ref_aligner_destroy(h)

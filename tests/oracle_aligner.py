"""ctypes view of the aligner oracle (oracle/build/libaligner_oracle.so) and of the reference's own CPU aligner code
built into oracle/_ref/libref_aligner.so -- TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_L = None
_R = None

OPS = "MXID"  # match, mismatch, insertion, deletion (AlignmentState order)


def lib():
    global _L
    if _L is None:
        path = os.path.join(ROOT, "oracle", "build", "libaligner_oracle.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "build/libaligner_oracle.so"], check=True)
        _L = C.CDLL(path)
        _L.aligner_oracle_myers_banded.restype = C.c_int32
        _L.aligner_oracle_myers_banded.argtypes = [C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p,
                                                   C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                                   C.POINTER(C.c_int64)]
        _L.aligner_oracle_host_max_bandwidth.restype = C.c_int32
        _L.aligner_oracle_delta_identity_mismatches.restype = C.c_int64
        _L.aligner_oracle_delta_identity_mismatches.argtypes = []
    return _L


def ref():
    """The reference's own CPU code (None when oracle/_ref was not built: no /root/reference and no prebuilt .so)."""
    global _R
    if _R is None:
        path = os.path.join(ROOT, "oracle", "_ref", "libref_aligner.so")
        if not os.path.exists(path):
            return None
        _R = C.CDLL(path)
        for f in (_R.ref_myers_edit_distance, _R.ref_nw_edit_distance):
            f.restype = C.c_int32
            f.argtypes = [C.c_char_p, C.c_int32, C.c_char_p, C.c_int32]
        _R.ref_needleman_wunsch_cpu.restype = C.c_int32
        _R.ref_needleman_wunsch_cpu.argtypes = [C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.c_void_p, C.c_int32]
        _R.ref_ukkonen_cpu.restype = C.c_int32
        _R.ref_ukkonen_cpu.argtypes = [C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
    return _R


def align(query, target, max_bandwidth):
    """Returns dict(status, optimal, runs=[(op, count)...] in forward order, states, cigar, edit_distance, cells)."""
    q = query.encode() if isinstance(query, str) else bytes(query)
    t = target.encode() if isinstance(target, str) else bytes(target)
    mbw = lib().aligner_oracle_host_max_bandwidth(max_bandwidth, len(q))
    cap = len(q) + len(t) + 4
    ops = np.zeros(cap, np.int8)
    cnt = np.zeros(cap, np.int32)
    n, opt, cells = C.c_int32(0), C.c_int32(0), C.c_int64(0)
    rc = lib().aligner_oracle_myers_banded(q, len(q), t, len(t), mbw, ops.ctypes.data, cnt.ctypes.data, C.byref(n),
                                           C.byref(opt), C.byref(cells))
    runs = [(int(ops[i]), int(cnt[i])) for i in range(n.value)][::-1]  # host reverses (aligner_global_myers_banded.cpp:423)
    out = dict(status=rc, optimal=bool(opt.value), runs=runs, cells=cells.value)
    out["cigar"] = cigar(runs, False)
    out["cigar_extended"] = cigar(runs, True)
    out["edit_distance"] = sum(c for o, c in runs if o != 0)
    return out


def cigar(runs, extended):
    """AlignmentImpl::convert_to_cigar (alignment_impl.cpp:70-127): basic M/I/D merges match+mismatch; extended =/X/I/D."""
    sym = "=XID" if extended else "MMID"
    out, last, acc = [], None, 0
    for o, c in runs:
        s = sym[o]
        if s == last:
            acc += c
        else:
            if last is not None:
                out.append("%d%s" % (acc, last))
            last, acc = s, c
    if last is not None:
        out.append("%d%s" % (acc, last))
    return "".join(out)


def pattern_view(query, target):
    """What the Myers bit-vector kernels see of a pair (myers_gpu.cu:196-241, hirschberg_myers_gpu.cu: the same tables): the query
    through four patterns -- 'A', 'C', 'T', 'G'; any other query character matches nothing -- and a target character through its
    pattern index (c >> 1) & 3, i.e. as one of those four letters. The value-level oracles below compare characters, so they are
    given this view of the pair; for sequences over ACGT it is the identity."""
    q = query.encode() if isinstance(query, str) else bytes(query)
    t = target.encode() if isinstance(target, str) else bytes(target)
    return bytes(c if c in b"ACGT" else 0 for c in q), bytes(b"ACTG"[(c >> 1) & 3] for c in t)


def hirschberg(query, target, max_query_length=None):
    """The default aligner (Hirschberg + Myers restatement, oracle/hirschberg_oracle.c):
    dict(status, states (forward order), cigar, cigar_extended, edit_distance)."""
    L = lib()
    L.hirschberg_oracle_align_raw.restype = C.c_int32
    L.hirschberg_oracle_align_raw.argtypes = [C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.c_char_p, C.c_char_p, C.c_int32, C.c_void_p,
                                              C.POINTER(C.c_int32)]
    raw_q = query.encode() if isinstance(query, str) else bytes(query)
    raw_t = target.encode() if isinstance(target, str) else bytes(target)
    q, t = pattern_view(query, target)
    if max_query_length is None:
        max_query_length = max(len(q), len(t)) + 1
    path = np.zeros(len(q) + len(t) + 8, np.int8)
    n = C.c_int32(0)
    rc = L.hirschberg_oracle_align_raw(q, len(q), t, len(t), raw_q, raw_t, max_query_length, path.ctypes.data, C.byref(n))
    states = [int(x) for x in path[:n.value]][::-1]  # the host reverses (aligner_global.cpp:180)
    runs = []
    for s in states:
        if runs and runs[-1][0] == s:
            runs[-1] = (s, runs[-1][1] + 1)
        else:
            runs.append((s, 1))
    return dict(status=rc, states=states, cigar=cigar(runs, False), cigar_extended=cigar(runs, True),
                edit_distance=sum(1 for s in states if s != 0))


def _states_result(states, rc=0):
    runs = []
    for s in states:
        if runs and runs[-1][0] == s:
            runs[-1] = (s, runs[-1][1] + 1)
        else:
            runs.append((s, 1))
    return dict(status=rc, states=states, cigar=cigar(runs, False), cigar_extended=cigar(runs, True),
                edit_distance=sum(1 for s in states if s != 0))


def _global(fn_name, query, target, *extra):
    L = lib()
    fn = getattr(L, fn_name)
    fn.restype = C.c_int32
    q = query.encode() if isinstance(query, str) else bytes(query)
    t = target.encode() if isinstance(target, str) else bytes(target)
    path = np.zeros(len(q) + len(t) + 8, np.int8)
    n = C.c_int32(0)
    rc = fn(C.c_char_p(q), C.c_int32(len(q)), C.c_char_p(t), C.c_int32(len(t)), *[C.c_int32(x) for x in extra],
            C.c_void_p(path.ctypes.data), C.byref(n))
    return _states_result([int(x) for x in path[:n.value]][::-1], rc)  # the host reverses (aligner_global.cpp:180)


def ukkonen(query, target, p=100):
    """AlignerGlobalUkkonen restatement (oracle/global_oracle.c), band parameter p (the class fixes p = 100)."""
    return _global("ukkonen_oracle_align", query, target, p)


def myers_full(query, target):
    """AlignerGlobalMyers restatement (oracle/global_oracle.c)."""
    return _global("myers_full_oracle_align", *pattern_view(query, target))


def ref_ukkonen_cpu(query, target, p):
    """The reference's own ukkonen_cpu() (target, query, p) -> states in forward order, or None without oracle/_ref."""
    R = ref()
    if R is None:
        return None
    q = query.encode() if isinstance(query, str) else bytes(query)
    t = target.encode() if isinstance(target, str) else bytes(target)
    out = np.zeros(len(q) + len(t) + 8, np.int8)
    n = R.ref_ukkonen_cpu(t, len(t), q, len(q), p, out.ctypes.data, len(out))
    return [int(x) for x in out[:n]]


def delta_identity_mismatches():
    """Backtrace steps (process-wide) in which the kernels' way of getting the neighbour scores -- `left` read, `diag` and
    `above` from vertical-delta bits -- differed from the reference's three cell reads (oracle/aligner_oracle.c model_step)."""
    return lib().aligner_oracle_delta_identity_mismatches()

"""Seeded synthetic inputs with the reference generator's semantics (genomeutils.hpp:32-127), via the host library."""
import ctypes as C

import numpy as np

from . import _native


def generate_window(seed, backbone_len=960, n_reads=32, max_mut=48, max_ins=24, max_del=24):
    """One POA window: backbone + (n_reads-1) mutated copies, std::minstd_rand(seed). Returns list[bytes]."""
    cap = (backbone_len + max_ins + 8) * n_reads
    buf = np.zeros(cap, np.uint8)
    lens = np.zeros(n_reads, np.int32)
    n = _native.host().gw_generate_window(seed, backbone_len, n_reads, max_mut, max_ins, max_del,
                                          buf.ctypes.data, cap, lens.ctypes.data)
    if n < 0:
        raise RuntimeError("gw_generate_window failed: %d" % n)
    out, off = [], 0
    for l in lens:
        out.append(bytes(buf[off:off + l]))
        off += int(l)
    return out


def config3_windows(n_windows=1024, first_seed=1000):
    """window w uses seed 1000+w, backbone 960, 32 reads, <=48 subs / 24 ins / 24 dels."""
    return [generate_window(first_seed + w) for w in range(n_windows)]


def generate_pairs(seed, n_pairs, length, max_mut, max_ins, max_del):
    cap = n_pairs * (2 * length + max_ins + 8)
    buf = np.zeros(cap, np.uint8)
    ql = np.zeros(n_pairs, np.int32)
    tl = np.zeros(n_pairs, np.int32)
    n = _native.host().gw_generate_pairs(seed, n_pairs, length, max_mut, max_ins, max_del, buf.ctypes.data, cap,
                                         ql.ctypes.data, tl.ctypes.data)
    if n < 0:
        raise RuntimeError("gw_generate_pairs failed: %d" % n)
    out, off = [], 0
    for a, b in zip(ql, tl):
        q = bytes(buf[off:off + a]); off += int(a)
        t = bytes(buf[off:off + b]); off += int(b)
        out.append((q, t))
    return out


def long_read_window(w, max_len=32768, min_len=2000):
    """Window `w` of the long-read MSA set (BASELINE configs[3], SURVEY.md 8(d) "Config 4"): 8..32 reads, backbone
    length log-uniform in [min_len, 0.93 max_len], 8-12 % divergence split 1 : 2 : 2 over substitutions :
    insertions : deletions, seed 2000 + w. Reads the batch API would reject (>= max_len) are left out.
    Returns list[str]."""
    import math
    import random
    rng = random.Random(2000 + w)
    n_reads = rng.randint(8, 32)
    backbone = int(round(math.exp(rng.uniform(math.log(min_len), math.log(max_len * 0.93)))))
    div = rng.uniform(0.08, 0.12)
    # the generator fires each of its max_* trials with p = 0.5 (genomeutils.hpp:47-127): 2 x for the expected count
    mut, ins, dele = (int(2 * backbone * div * f) for f in (0.2, 0.4, 0.4))
    reads = [r.decode() for r in generate_window(2000 + w, backbone, n_reads, mut, ins, dele)]
    return [s for s in reads if len(s) < max_len]


def random_length_pairs(seed, n_pairs, max_len):
    """The random pairs of the reference's aligner test cases (cudaaligner_test_cases.cpp:29-41): list of
    (target, query) bytes."""
    cap = n_pairs * (2 * max_len + 8) * 2
    buf = np.zeros(cap, np.uint8)
    tl = np.zeros(n_pairs, np.int32)
    ql = np.zeros(n_pairs, np.int32)
    L = _native.host()
    L.gw_generate_random_length_pairs.restype = C.c_int64
    n = L.gw_generate_random_length_pairs(C.c_uint32(seed), C.c_int32(n_pairs), C.c_int32(max_len), C.c_void_p(buf.ctypes.data),
                                          C.c_int64(cap), C.c_void_p(tl.ctypes.data), C.c_void_p(ql.ctypes.data))
    if n < 0:
        raise RuntimeError("gw_generate_random_length_pairs failed: %d" % n)
    out, off = [], 0
    for a, b in zip(tl, ql):
        t = bytes(buf[off:off + a]); off += int(a)
        q = bytes(buf[off:off + b]); off += int(b)
        out.append((t, q))
    return out

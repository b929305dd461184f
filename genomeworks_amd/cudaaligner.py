"""Python mirror of pygenomeworks' genomeworks.cudaaligner (cudaaligner.pyx) over the object-level C API.

CudaAlignerBatch keeps the reference's constructor (the default, fixed-stride factory); `max_bandwidth=` selects the
banded Myers aligner (create_aligner(global_alignment, max_bandwidth, ...)), which is what the benchmarks use."""
import ctypes as C

import numpy as np

from . import _native
from .cuda import CudaStream

# cudaaligner::StatusType (cudaaligner.hpp:34-42)
success = 0
uninitialized = 1
exceeded_max_alignments = 2
exceeded_max_length = 3
exceeded_max_alignment_difference = 4
generic_error = 5

_STATUS = {0: "success", 1: "uninitialized", 2: "exceeded_max_alignments", 3: "exceeded_max_length",
           4: "exceeded_max_alignment_difference", 5: "generic_error"}


def status_to_str(status):
    if status not in _STATUS:
        raise RuntimeError("Unknown error status : " + str(status))
    return _STATUS[status]


def _bind(L):
    if getattr(L, "_gw_aln_bound", False):
        return L
    vp, i32 = C.c_void_p, C.c_int32
    L.gw_aligner_create_banded.restype = vp
    L.gw_aligner_create_banded.argtypes = [i32, vp, i32, C.c_int64]
    L.gw_aligner_create.restype = vp
    L.gw_aligner_create.argtypes = [i32, i32, i32, vp, i32, C.c_int64]
    L.gw_aligner_create_algorithm.restype = vp
    L.gw_aligner_create_algorithm.argtypes = [C.c_char_p, i32, i32, i32, vp, i32, C.c_int64]
    L.gw_aligner_destroy.argtypes = [vp]
    L.gw_aligner_add_alignment.argtypes = [vp, C.c_char_p, i32, C.c_char_p, i32, C.c_int, C.c_int]
    for n in ("gw_aligner_align_all", "gw_aligner_sync_alignments", "gw_aligner_num_alignments", "gw_aligner_reset",
              "gw_aligner_relaunch"):
        getattr(L, n).argtypes = [vp]
    for n in ("gw_alignment_status", "gw_alignment_is_optimal", "gw_alignment_edit_distance"):
        getattr(L, n).argtypes = [vp, i32]
    L.gw_alignment_cigar.restype = C.POINTER(C.c_char)
    L.gw_alignment_cigar.argtypes = [vp, i32, i32, C.POINTER(i32)]
    L.gw_alignment_states.argtypes = [vp, i32, vp, i32]
    L.gw_aligner_band_cells.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.gw_aligner_relaunch_timed.argtypes = [vp, C.POINTER(C.c_float)]
    L.gw_aligner_get_runs.restype = C.c_int64
    L.gw_aligner_get_runs.argtypes = [vp, vp, vp, vp, C.c_int64, vp, vp]
    L.gw_aligner_device_alignments.argtypes = [vp, C.POINTER(i32), C.POINTER(C.c_int64)]
    L.gw_aligner_copy_device_alignments.argtypes = [vp, vp, vp, vp, vp]
    L._gw_aln_bound = True
    return L


class CudaAlignment:
    """One alignment result (pygenomeworks CudaAlignment): query, target, cigar, status, alignment states."""

    def __init__(self, query, target, cigar, cigar_extended, status, is_optimal, edit_distance, states):
        self.query = query
        self.target = target
        self.cigar = cigar
        self.cigar_extended = cigar_extended
        self.status = status
        self.is_optimal = is_optimal
        self.edit_distance = edit_distance
        self.alignment = states

    def format_alignment(self):
        """(query line, pairing line, target line) like Alignment::format_alignment."""
        q, p, t, qi, ti = [], [], [], 0, 0
        for s in self.alignment:
            if s in (0, 1):
                q.append(self.query[qi]); t.append(self.target[ti]); p.append("|" if s == 0 else "x"); qi += 1; ti += 1
            elif s == 3:
                q.append(self.query[qi]); t.append("-"); p.append(" "); qi += 1
            else:
                q.append("-"); t.append(self.target[ti]); p.append(" "); ti += 1
        return "".join(q), "".join(p), "".join(t)


class CudaAlignerBatch:
    """Python API for GPU-accelerated global pairwise alignment (pygenomeworks CudaAlignerBatch)."""

    def __init__(self, max_query_length=None, max_target_length=None, max_alignments=None, alignment_type="global",
                 stream=None, device_id=0, max_device_memory_allocator_caching_size=-1, max_bandwidth=None,
                 algorithm=None):
        """algorithm (not in pygenomeworks): one of the reference's non-public aligner classes that its C++ tests and
        benchmarks construct directly -- "hirschberg_myers" (the default), "ukkonen", "myers"."""
        self._L = _bind(_native.host())
        if alignment_type != "global":
            raise RuntimeError("Unknown alignment_type provided. Must be global.")
        if stream is not None and not isinstance(stream, CudaStream):
            raise RuntimeError("Type for stream option must be CudaStream")
        self.stream = stream
        st = stream.stream if stream is not None else None
        if max_bandwidth is not None:
            self._h = self._L.gw_aligner_create_banded(int(max_bandwidth), st, device_id,
                                                       int(max_device_memory_allocator_caching_size))
        elif algorithm is not None:
            self._h = self._L.gw_aligner_create_algorithm(algorithm.encode(), int(max_query_length), int(max_target_length),
                                                          int(max_alignments), st, device_id,
                                                          int(max_device_memory_allocator_caching_size))
        else:
            self._h = self._L.gw_aligner_create(int(max_query_length), int(max_target_length), int(max_alignments), st,
                                                device_id, int(max_device_memory_allocator_caching_size))
        if not self._h:
            raise RuntimeError(self._L.gw_last_error().decode())
        self._pairs = []

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.gw_aligner_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def add_alignment(self, query, target, reverse_complement_query=False, reverse_complement_target=False):
        """Queue one pair. Returns the StatusType (exceeded_max_alignments: run the batch, reset, retry)."""
        q = query.encode("utf-8") if isinstance(query, str) else bytes(query)
        t = target.encode("utf-8") if isinstance(target, str) else bytes(target)
        st = self._L.gw_aligner_add_alignment(self._h, q, len(q), t, len(t), int(reverse_complement_query),
                                              int(reverse_complement_target))
        if st < 0:
            raise RuntimeError(self._L.gw_last_error().decode())
        if st == success:
            self._pairs.append((q.decode(), t.decode()))
        return st

    def align_all(self):
        st = self._L.gw_aligner_align_all(self._h)
        if st != 0:
            raise RuntimeError("align_all failed: %s" % (self._L.gw_last_error().decode() if st < 0 else status_to_str(st)))

    def relaunch(self):
        """Benchmark helper: run the kernels again on the inputs resident in HBM (before get_alignments)."""
        if self._L.gw_aligner_relaunch(self._h) != 0:
            raise RuntimeError(self._L.gw_last_error().decode())

    def relaunch_timed(self):
        """Benchmark helper: relaunch and return the kernels' time in ms (HIP events on the aligner's stream)."""
        ms = C.c_float(0)
        if self._L.gw_aligner_relaunch_timed(self._h, C.byref(ms)) != 0:
            raise RuntimeError(self._L.gw_last_error().decode())
        return ms.value

    def device_sync(self):
        """Wait for align_all() and report what get_alignments_device() holds: (n_alignments, total runs)."""
        n, total = C.c_int32(0), C.c_int64(0)
        if self._L.gw_aligner_device_alignments(self._h, C.byref(n), C.byref(total)) < 0:
            raise RuntimeError(self._L.gw_last_error().decode())
        return n.value, total.value

    def band_cells(self):
        v = C.c_uint64(0)
        if self._L.gw_aligner_band_cells(self._h, C.byref(v)) != 0:
            raise RuntimeError(self._L.gw_last_error().decode())
        return v.value

    def sync(self):
        """sync_alignments() only (host materialisation in the library, no Python marshalling). Returns the count."""
        st = self._L.gw_aligner_sync_alignments(self._h)
        if st != 0:
            raise RuntimeError("sync_alignments failed")
        return self._L.gw_aligner_num_alignments(self._h)

    def get_runs(self):
        """sync_alignments() + every alignment as a run-length CIGAR in forward order, without per-alignment Python
        objects: dict(offsets[n + 1], ops, counts, status[n], optimal[n]) of numpy arrays."""
        n = self.sync()
        total = self._L.gw_aligner_get_runs(self._h, None, None, None, 0, None, None)
        if total < 0:
            raise RuntimeError(self._L.gw_last_error().decode())
        offsets = np.zeros(n + 1, np.int64)
        ops = np.zeros(max(total, 1), np.int8)
        counts = np.zeros(max(total, 1), np.int32)
        status = np.zeros(max(n, 1), np.int32)
        optimal = np.zeros(max(n, 1), np.int32)
        self._L.gw_aligner_get_runs(self._h, offsets.ctypes.data, ops.ctypes.data, counts.ctypes.data, total,
                                    status.ctypes.data, optimal.ctypes.data)
        return dict(offsets=offsets, ops=ops[:total], counts=counts[:total], status=status[:n], optimal=optimal[:n])

    def get_alignments_device(self):
        """Aligner::get_alignments_device() read back for inspection (stream synchronised first): dict of numpy arrays
        cigar_operations, cigar_runlengths (each alignment back to front), cigar_offsets[n + 1], metadata[n] (bit 31
        is_optimal, bits 26-0 index as added), or None for aligners without a device-resident form."""
        n, total = C.c_int32(0), C.c_int64(0)
        rc = self._L.gw_aligner_device_alignments(self._h, C.byref(n), C.byref(total))
        if rc < 0:
            raise RuntimeError(self._L.gw_last_error().decode())
        if rc != 0:
            return None
        ops = np.zeros(max(total.value, 1), np.int8)
        runs = np.zeros(max(total.value, 1), np.int32)
        offs = np.zeros(n.value + 1, np.int32)
        meta = np.zeros(max(n.value, 1), np.uint32)
        if self._L.gw_aligner_copy_device_alignments(self._h, ops.ctypes.data, runs.ctypes.data, offs.ctypes.data, meta.ctypes.data) != 0:
            raise RuntimeError(self._L.gw_last_error().decode())
        return dict(cigar_operations=ops[:total.value], cigar_runlengths=runs[:total.value], cigar_offsets=offs,
                    metadata=meta[:n.value], total_length=total.value, n_alignments=n.value)

    def get_alignments(self):
        """sync_alignments() + list of CudaAlignment in the order the pairs were added."""
        n = self.sync()
        out = []
        for i in range(n):
            ln = C.c_int32(0)
            p = self._L.gw_alignment_cigar(self._h, i, 0, C.byref(ln))
            cigar = C.string_at(p, ln.value).decode()
            p = self._L.gw_alignment_cigar(self._h, i, 1, C.byref(ln))
            cigar_x = C.string_at(p, ln.value).decode()
            ns = self._L.gw_alignment_states(self._h, i, None, 0)
            states = np.zeros(max(ns, 1), np.int8)
            self._L.gw_alignment_states(self._h, i, states.ctypes.data, ns)
            q, t = self._pairs[i] if i < len(self._pairs) else ("", "")
            st, opt, ed = (self._L.gw_alignment_status(self._h, i), self._L.gw_alignment_is_optimal(self._h, i),
                           self._L.gw_alignment_edit_distance(self._h, i))
            if ns < 0 or st < 0 or opt < 0 or ed < 0:  # the C ABI reports a bad index / a thrown accessor as -1 + error string
                raise RuntimeError("alignment %d: %s" % (i, self._L.gw_last_error().decode()))
            out.append(CudaAlignment(q, t, cigar, cigar_x, st, bool(opt), ed, [int(x) for x in states[:ns]]))
        return out

    def reset(self):
        self._L.gw_aligner_reset(self._h)
        self._pairs = []

"""TEST INFRASTRUCTURE: ctypes binding of oracle/_ref/libref_cudapoa_simt.so -- the REFERENCE's own cudapoa library (its CUDA
sources compiled by g++ where they lie, kernels run on the CPU by oracle/simt/simt.hpp; `make -C oracle -f Makefile.ref
ref_cudapoa_simt`). Only what checks the oracle and writes tests/golden/reference_simt_*.json uses it; /root/reference does not
exist on the GPU box, where the prebuilt library travels with the snapshot (or the tests that need it skip)."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "_ref", "libref_cudapoa_simt.so")
_lib = None


def available():
    return os.path.exists(PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(PATH)
        L.ref_poa_create.restype = C.c_void_p
        L.ref_poa_create.argtypes = [C.c_int] * 4 + [C.c_float, C.c_float] + [C.c_int] * 5 + [C.c_longlong]
        L.ref_poa_config.argtypes = [C.c_int] * 4 + [C.c_float, C.c_float, C.c_int, C.POINTER(C.c_int)]
        L.ref_poa_destroy.argtypes = [C.c_void_p]
        L.ref_poa_add_group.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        for f in ("ref_poa_total_poas", "ref_poa_fetch_consensus", "ref_poa_fetch_msa"):
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("ref_poa_generate", "ref_poa_reset"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = None
        for f in ("ref_poa_consensus_length", "ref_poa_window_status", "ref_poa_msa_status", "ref_poa_msa_rows"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
        L.ref_poa_consensus.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_ushort)]
        L.ref_poa_consensus.restype = None
        L.ref_poa_msa_row_length.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ref_poa_msa_row.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
        L.ref_poa_msa_row.restype = None
        _lib = L
    return _lib


def config(max_seq=1024, max_seqs=100, band_width=256, band_mode=0, storage_factor=2.0, graph_factor=3.0, max_pred=0):
    """The reference's BatchConfig(max_seq_sz, max_seq_per_poa, band_width, banding, ...) -> its eight fields."""
    out = (C.c_int * 8)()
    lib().ref_poa_config(max_seq, max_seqs, band_width, band_mode, storage_factor, graph_factor, max_pred, out)
    keys = ("max_sequence_size", "max_consensus_size", "max_nodes_per_graph", "matrix_sequence_dimension", "alignment_band_width",
            "max_sequences_per_poa", "band_mode", "max_banded_pred_distance")
    return dict(zip(keys, list(out)))


class RefBatch:
    """create_batch(...) of the reference with BatchConfig(max_seq, max_seqs, band_width, band_mode, ...)."""

    def __init__(self, max_seq=1024, max_seqs=100, band_width=256, band_mode=0, storage_factor=2.0, graph_factor=3.0, max_pred=0,
                 gap=-8, mismatch=-6, match=8, output_mask=1, max_mem=1 << 28):
        self.h = lib().ref_poa_create(max_seq, max_seqs, band_width, band_mode, storage_factor, graph_factor, max_pred, output_mask, gap, mismatch,
                                      match, max_mem)
        if not self.h:
            raise RuntimeError("the reference's create_batch threw")
        self.output_mask = output_mask

    def close(self):
        if self.h:
            lib().ref_poa_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def add_poa_group(self, reads, weights=None):
        """-> (StatusType, [per-read StatusType])"""
        n = len(reads)
        raw = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
        arr = (C.c_char_p * n)(*raw)
        lens = (C.c_int * n)(*[len(r) for r in raw])
        wkeep, wptr = [], None
        if weights is not None:
            wptr = (C.c_void_p * n)()
            for i, w in enumerate(weights):
                if w is None:
                    wptr[i] = None
                else:
                    buf = (C.c_byte * len(w))(*[int(x) for x in w])
                    wkeep.append(buf)
                    wptr[i] = C.cast(buf, C.c_void_p)
        st = (C.c_int * n)()
        rc = lib().ref_poa_add_group(self.h, n, arr, lens, wptr, st)
        return rc, list(st)

    def generate_poa(self):
        lib().ref_poa_generate(self.h)

    def reset(self):
        lib().ref_poa_reset(self.h)

    def get_consensus(self):
        """-> list of dict(status, consensus, coverage) per window"""
        L = lib()
        L.ref_poa_fetch_consensus(self.h)
        out = []
        for w in range(L.ref_poa_total_poas(self.h)):
            n = L.ref_poa_consensus_length(self.h, w)
            bases = C.create_string_buffer(max(n, 1))
            cov = (C.c_ushort * max(n, 1))()
            L.ref_poa_consensus(self.h, w, bases, cov)
            out.append(dict(status=L.ref_poa_window_status(self.h, w), consensus=bases.raw[:n].decode("latin1"), coverage=list(cov)[:n]))
        return out

    def get_msa(self):
        """-> list of dict(status, msa rows) per window"""
        L = lib()
        L.ref_poa_fetch_msa(self.h)
        out = []
        for w in range(L.ref_poa_total_poas(self.h)):
            rows = []
            for r in range(L.ref_poa_msa_rows(self.h, w)):
                n = L.ref_poa_msa_row_length(self.h, w, r)
                buf = C.create_string_buffer(max(n, 1))
                L.ref_poa_msa_row(self.h, w, r, buf)
                rows.append(buf.raw[:n].decode("latin1"))
            out.append(dict(status=L.ref_poa_msa_status(self.h, w), msa=rows))
        return out

"""CPU-side checks: the C-ABI libraries load and export every symbol the headers declare (no compute calls)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    with open(os.path.join(ROOT, "include", header)) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def libs():
    from genomeworks_amd import build as gb
    gb.build_all()
    from genomeworks_amd import _native
    return _native.gwhip(), _native.host()


def test_gwhip_exports_every_declared_symbol(libs):
    gwhip, _ = libs
    names = declared("gwhip.h", "gwhip_")
    assert len(names) >= 7
    missing = [n for n in names if not hasattr(gwhip, n)]
    assert not missing, missing


def test_host_exports_every_declared_symbol(libs):
    _, host = libs
    names = declared("gw_capi.h", "gw_")
    missing = [n for n in names if not hasattr(host, n)]
    assert not missing, missing


def test_build_arch_is_gfx950(libs):
    gwhip, _ = libs
    assert gwhip.gwhip_build_arch() == b"gfx950"
    assert gwhip.gwhip_abi_version() >= 1


def test_workspace_sizing_matches_survey_scale(libs):
    # config 3: <int16,int16,int16>, 3072 nodes, msd 264; host-only sizing function
    from genomeworks_amd import _native
    gwhip, _ = libs
    c = _native.PoaConfig(1024, 2048, 3072, 264, 256, 32, 1, 512, -8, -6, 8, 1, 0, 0, 1, 0)
    per_poa, per_matrix = C.c_int64(0), C.c_int64(0)
    gwhip.gwhip_poa_bytes_per_window(C.byref(c), C.byref(per_poa), C.byref(per_matrix))
    assert per_matrix.value == 3072 * 264 * 2
    one = gwhip.gwhip_poa_workspace_bytes(C.byref(c), 1, 0)
    assert gwhip.gwhip_poa_workspace_bytes(C.byref(c), 1024, 0) == 1024 * one
    assert 2.0e6 < one < 4.5e6  # ~3 MB per window, as the reference's own carving


def test_batch_config_math_matches_reference(libs):
    # batch.cu:34-70: BatchConfig(1024, 32, 256, static_band) -> 3072 nodes, msd 264, consensus 2048, pred 512
    from genomeworks_amd import _native, cudapoa
    _, host = libs
    cudapoa._bind(host)
    cfg = _native.PoaBatchConfig()
    assert host.gw_poa_batch_config_default(C.byref(cfg), 1024, 32, 256, 1, 2.0, 3.0, 0) == 0
    assert (cfg.max_nodes_per_graph, cfg.matrix_sequence_dimension, cfg.max_consensus_size,
            cfg.max_banded_pred_distance, cfg.alignment_band_width) == (3072, 264, 2048, 512, 256)
    assert host.gw_poa_batch_config_default(C.byref(cfg), 1024, 100, 200, 0, 2.0, 3.0, 0) == 0  # BM_SingleBatchTest
    assert (cfg.alignment_band_width, cfg.matrix_sequence_dimension) == (256, 1024)
    assert host.gw_poa_batch_config_default(C.byref(cfg), 32768, 32, 256, 2, 2.0, 3.0, 0) == 0  # config 4
    assert (cfg.max_nodes_per_graph, cfg.matrix_sequence_dimension) == (98304, 528)
    # explicit ctor validation (batch.cu:88-98)
    assert host.gw_poa_batch_config_full(C.byref(cfg), 1024, 512, 3072, 256, 10, 264, 1, 512) == -1
    assert b"max_consensus_size" in host.gw_last_error()


def test_synthetic_generator_is_seed_stable(libs):
    from genomeworks_amd import synthetic
    a = synthetic.generate_window(1000)
    b = synthetic.generate_window(1000)
    c = synthetic.generate_window(1001)
    assert a == b and a != c
    assert len(a) == 32 and len(a[0]) == 960 and all(900 <= len(r) <= 984 for r in a)
    assert set(b"".join(a)) <= set(b"ACGT")


def test_banded_myers_workspace_sized_in_pieces_equals_the_whole(libs):
    """gwhip_myers_banded_workspace_words over pieces of whole waves (64 slots) + ..._bytes_of_words is what
    gwhip_myers_banded_workspace_bytes_ordered returns for the batch: align_all() sizes a chunk's workspace piece by piece on
    host threads (host functions, no device call)."""
    import random
    import numpy as np
    gwhip, _ = libs
    gwhip.gwhip_myers_banded_workspace_bytes_ordered.restype = C.c_size_t
    gwhip.gwhip_myers_banded_workspace_bytes_of_words.restype = C.c_size_t
    gwhip.gwhip_myers_banded_workspace_words.restype = C.c_int64
    p64, p32 = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    gwhip.gwhip_myers_banded_workspace_bytes_ordered.argtypes = [C.c_int32, p64, p32, p32]
    gwhip.gwhip_myers_banded_workspace_words.argtypes = [C.c_int32, C.c_int32, p64, p32, p32]
    gwhip.gwhip_myers_banded_workspace_bytes_of_words.argtypes = [C.c_int32, C.c_int64, C.c_int64]
    rng = random.Random(5)
    for n in (1, 63, 64, 65, 1000, 4099):
        starts = [0]
        for _ in range(n):
            q = rng.choice([1, 31, 32, 150, 151, 1000])
            starts.append(starts[-1] + q)
            starts.append(starts[-1] + max(1, q + rng.randint(-3, 3)))
        starts = np.array(starts, np.int64)
        bws = np.array([rng.choice([7, 150, 512, 1024]) for _ in range(n)], np.int32)
        order = np.array(sorted(range(n), key=lambda i: -(starts[2 * i + 2] - starts[2 * i])), np.int32)
        for sched in (order, None):
            so = sched.ctypes.data_as(p32) if sched is not None else None
            whole = gwhip.gwhip_myers_banded_workspace_bytes_ordered(n, starts.ctypes.data_as(p64), bws.ctypes.data_as(p32), so)
            for share in (64, 192, 4096):
                words = 0
                for lo in range(0, n, share):
                    words += gwhip.gwhip_myers_banded_workspace_words(lo, min(n, lo + share) - lo, starts.ctypes.data_as(p64), bws.ctypes.data_as(p32), so)
                assert gwhip.gwhip_myers_banded_workspace_bytes_of_words(n, int(starts[-1] - starts[0]), words) == whole


def test_python_mirror_of_gwhip_myers_args_has_the_layout_of_the_header(tmp_path):
    """_native.MyersArgs (ctypes) against include/gwhip.h compiled by gcc: size and the offset of every field."""
    import subprocess
    from genomeworks_amd import _native
    names = [f[0] for f in _native.MyersArgs._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gwhip.h"\nint main(void) {\n  printf("%zu", sizeof(gwhip_myers_args));\n'
                   + "".join('  printf(" %%zu", offsetof(gwhip_myers_args, %s));\n' % n for n in names) + '  return 0;\n}\n')
    exe = str(tmp_path / "layout")
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", exe, str(src)], check=True)
    got = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    assert got[0] == C.sizeof(_native.MyersArgs)
    assert got[1:] == [getattr(_native.MyersArgs, n).offset for n in names]

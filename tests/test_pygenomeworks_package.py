"""The Cython package `genomeworks` (pygenomeworks/): builds in-tree against include/ and imports with the reference's
module / class / method names (pygenomeworks/genomeworks/{cuda,cudapoa,cudaaligner}). No device calls here."""
import inspect
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gw():
    from genomeworks_amd import build
    pkg = build.build_bindings()
    sys.path.insert(0, pkg)
    import genomeworks.cuda
    import genomeworks.cudaaligner
    import genomeworks.cudapoa
    yield genomeworks
    sys.path.remove(pkg)


def test_modules_are_compiled_extensions_linked_with_the_host_library(gw):
    for mod in (gw.cuda.cuda, gw.cudapoa.cudapoa, gw.cudaaligner.cudaaligner):
        assert mod.__file__.endswith(".so") and os.path.dirname(mod.__file__).startswith(os.path.join(ROOT, "pygenomeworks"))
    maps = open("/proc/self/maps").read()
    assert "libgenomeworks_amd.so" in maps and "libamdhip64" in maps


def test_api_surface_matches_pygenomeworks(gw):
    # pygenomeworks/genomeworks/cuda/cuda.pyx
    for name in ("CudaStream", "CudaRuntimeError", "cuda_get_device_count", "cuda_set_device", "cuda_get_device", "cuda_get_mem_info"):
        assert hasattr(gw.cuda, name), name
    assert issubclass(gw.cuda.CudaRuntimeError, Exception)
    # cudapoa.pyx:69-334
    poa = gw.cudapoa.CudaPoaBatch
    for name in ("add_poa_group", "generate_poa", "get_consensus", "get_msa", "get_graphs", "reset", "total_poas", "batch_id"):
        assert hasattr(poa, name), name
    doc = poa.__init__.__doc__ or poa.__doc__ or ""
    assert gw.cudapoa.status_to_str(0) == "success" and gw.cudapoa.status_to_str(9) == "output_type_unavailable"
    with pytest.raises(RuntimeError):
        gw.cudapoa.status_to_str(99)
    # cudaaligner.pyx
    aln = gw.cudaaligner.CudaAlignerBatch
    for name in ("add_alignment", "align_all", "get_alignments", "reset"):
        assert hasattr(aln, name), name
    assert gw.cudaaligner.status_to_str(2) == "exceeded_max_alignments"
    a = gw.cudaaligner.CudaAlignment("ACGT", "ACGA", "4M", 0, 0, [0, 0, 0, 1], ["ACGT", "|||x", "ACGA"])
    assert a.alignment == ["m", "m", "m", "mm"] and a.alignment_type == "global" and str(a) == "ACGT\n|||x\nACGA\n"
    assert inspect.isclass(gw.cudaaligner.CudaAlignment)


def test_argument_errors_are_raised_before_any_device_work(gw):
    # these checks come first in the constructors (cudapoa.pyx:118-135, cudaaligner.pyx:196-205)
    with pytest.raises(RuntimeError, match="output_type"):
        gw.cudapoa.CudaPoaBatch(10, 1024, 1 << 30, output_type="error_input")
    with pytest.raises(RuntimeError, match="band_mode"):
        gw.cudapoa.CudaPoaBatch(10, 1024, 1 << 30, band_mode="diagonal")
    with pytest.raises(RuntimeError, match="stream"):
        gw.cudapoa.CudaPoaBatch(10, 1024, 1 << 30, stream=object())
    with pytest.raises(RuntimeError, match="alignment_type"):
        gw.cudaaligner.CudaAlignerBatch(10, 10, 1, alignment_type="local")
    with pytest.raises(RuntimeError, match="stream"):
        gw.cudaaligner.CudaAlignerBatch(10, 10, 1, stream=object())


def test_runtime_errors_carry_name_and_text(gw):
    import torch
    if torch.cuda.is_available():
        assert gw.cuda.cuda_get_device_count() >= 1
    else:
        with pytest.raises(gw.cuda.CudaRuntimeError, match="hipErrorNoDevice"):
            gw.cuda.cuda_get_device_count()

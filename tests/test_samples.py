"""samples/: the C++ samples compile against include/ exactly as they would against the reference headers
(source compatibility of the Batch / Aligner interfaces), and all four samples run on the GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "genomeworks_amd", "lib")


def build_sample(name, outdir):
    exe = os.path.join(str(outdir), name)
    cmd = ["g++", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "samples", name + ".cpp"), "-L", LIB, "-lgenomeworks_amd", "-lgwhip", "-L", "/opt/rocm/lib",
           "-lamdhip64", "-Wl,-rpath," + LIB, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("name", ["sample_cudapoa", "sample_cudaaligner"])
def test_cpp_samples_compile_against_public_headers(name, tmp_path):
    assert os.path.exists(build_sample(name, tmp_path))


@pytest.mark.gpu
def test_cpp_samples_run(tmp_path):
    r = subprocess.run([build_sample("sample_cudapoa", tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["ACGTTGCAACGTACGTTAGC", "TTGACCATTG"]
    r = subprocess.run([build_sample("sample_cudaaligner", tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "cigar" in r.stdout and r.stdout.count("cigar") == 3


@pytest.mark.gpu
def test_python_samples_run():
    for script, args in (("sample_cudapoa.py", ["-p"]), ("sample_cudapoa.py", ["-m", "-p"]), ("sample_cudaaligner.py", ["-n", "20", "-p"])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "samples", script)] + args, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert len(r.stdout) > 100

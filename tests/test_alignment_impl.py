"""The host Alignment classes against the reference's own vectors (cudaaligner/tests/Test_AlignmentImpl.cpp:36-204,
extracted into tests/golden/cudaaligner_vectors.json): CIGAR (basic / extended), format_alignment, getters -- for
AlignmentImpl and for the PackedAlignment views that sync_alignments() of the banded aligner hands out."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    from genomeworks_amd import build
    build.build_host()
    exe = str(tmp_path_factory.mktemp("aln") / "alignment_impl_driver")
    lib = os.path.join(ROOT, "genomeworks_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "genomeworks_amd", "host"), "-I", os.path.join(ROCM, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "alignment_impl_driver.cpp"), "-L", lib, "-lgenomeworks_amd", "-lgwhip",
                    "-L", os.path.join(ROCM, "lib"), "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath," + os.path.join(ROCM, "lib"),
                    "-pthread"], check=True)
    return exe


def run(driver, cases):
    text = "".join("%s %s %s %d\n" % (c["query"] or "-", c["target"] or "-", "".join(str(s) for s in c["alignment"]) or "-",
                                        1 if c["is_optimal"] else 0) for c in cases)
    out = subprocess.run([driver], input=text, capture_output=True, text=True, check=True).stdout
    return [line.split("\t") for line in out.split("\n") if line]


def test_reference_vectors(driver):
    with open(os.path.join(ROOT, "tests", "golden", "cudaaligner_vectors.json")) as f:
        cases = json.load(f)["alignment_impl"]
    rows = run(driver, cases)
    assert rows[0] == ["initial", "1", "1"]      # StatusType::uninitialized, AlignmentType::unset
    assert rows[1] == ["after_set", "0", "0"]    # success, global_alignment
    rows = rows[2:]
    per_case = 4  # AlignmentImpl, PackedAlignment(runs), runs, PackedAlignment(states)
    assert len(rows) == per_case * len(cases)
    for k, c in enumerate(cases):
        impl, packed_runs, runs, packed_states = rows[per_case * k:per_case * k + per_case]
        states = "".join(str(s) for s in c["alignment"])
        ed = sum(1 for s in c["alignment"] if s != 0)
        for got in (impl, packed_states):
            assert got[1:5] == [c["query"], c["target"], states, "1" if c["is_optimal"] else "0"], got[0]
            assert got[7] == c["cigar_basic"] and got[8] == c["cigar_extended"] and int(got[9]) == ed, got[0]
            assert got[10:13] == c["formatted"], got[0]
        # run-length form (what the banded aligner produces): same CIGARs and distance; per-position states are not held
        assert packed_runs[1:3] == [c["query"], c["target"]] and packed_runs[3] == ""
        assert packed_runs[7] == c["cigar_basic"] and packed_runs[8] == c["cigar_extended"] and int(packed_runs[9]) == ed
        # its runs, forward order, re-expand to the states
        expanded = "".join(op * int(n) for n, op in (r.split("x") for r in runs[1].split(",") if r))
        assert expanded == states


def test_line_wrapped_output_and_empty_alignment(driver):
    rows = run(driver, [dict(query="ACGTA", target="ACTA", alignment=[0, 0, 3, 0, 0], is_optimal=True),
                        dict(query="", target="", alignment=[], is_optimal=True)])[2:]
    impl = rows[0]
    # operator<< with linebreak_after = 3: query / pairing / target in blocks of three columns
    assert impl[13] == "ACG/|| /AC-/TA/||/TA//"
    empty = rows[4]
    assert empty[0] == "AlignmentImpl" and empty[7] == "" and empty[8] == "" and empty[9] == "0"


def test_worker_pool_runs_every_task_once(tmp_path):
    """gwhost::parallel_tasks (the pool behind the un-reversal of get_consensus()): tests/cpp/parallel_tasks_driver.cpp."""
    from genomeworks_amd import build
    build.build_host()
    exe = str(tmp_path / "parallel_tasks_driver")
    lib = os.path.join(ROOT, "genomeworks_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "genomeworks_amd", "host"), "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "parallel_tasks_driver.cpp"), "-L", lib, "-lgenomeworks_amd", "-lgwhip",
                    "-L", os.path.join(ROCM, "lib"), "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath," + os.path.join(ROCM, "lib"),
                    "-pthread"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


def test_two_bases_per_byte_round_trip(tmp_path):
    """The banded aligner's upload format (genomeworks_amd/host/base_packing.hpp) against a restatement of the device-side
    expansion: tests/cpp/base_packing_driver.cpp."""
    exe = str(tmp_path / "base_packing_driver")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "genomeworks_amd", "host"), "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "base_packing_driver.cpp")], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


def test_chunk_order_on_host_threads_equals_a_plain_stable_sort(tmp_path):
    """The banded aligner's processing order and workspace sizing in pieces on pool threads (genomeworks_amd/host/chunk_order.hpp)
    against one stable sort and one sizing call per chunk: tests/cpp/chunk_order_driver.cpp."""
    from genomeworks_amd import build
    build.build_all()
    exe = str(tmp_path / "chunk_order_driver")
    lib = os.path.join(ROOT, "genomeworks_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "genomeworks_amd", "host"), "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "chunk_order_driver.cpp"), "-L", lib, "-lgenomeworks_amd", "-lgwhip",
                    "-L", os.path.join(ROCM, "lib"), "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath," + os.path.join(ROCM, "lib"),
                    "-pthread"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr

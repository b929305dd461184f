#!/usr/bin/env python3
"""Regenerate tests/golden/cudaaligner_vectors.json from the reference's own cudaaligner test sources.

Runs only in the build container (needs /root/reference and g++). TEST INFRASTRUCTURE.
  * test pairs: the reference's cudaaligner/tests/cudaaligner_test_cases.cpp (11 fixed pairs + 10 pairs drawn with
    std::minstd_rand(5827349) through its genomeutils generators) is compiled WHERE IT LIES together with a small
    printer (written to a temp dir), so the pairs -- including the random ones -- are the reference's own output.
    The edit distance of each pair comes from the reference's CPU code (needleman_wunsch_build_score_matrix_naive,
    the yardstick of Test_MyersAlgorithm.cu:179-190) in the same program.
  * AlignmentImpl cases (Test_AlignmentImpl.cpp:66-142) and the pattern words of Test_HirschbergMyers.cu:107-123:
    transcribed / extracted by regex; every expected literal is checked to occur verbatim in the cited source.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

REF = os.environ.get("GW_REFERENCE", "/root/reference")
T = os.path.join(REF, "cudaaligner", "tests")
HERE = os.path.dirname(os.path.abspath(__file__))

PRINTER = r'''
#include <cstdio>
#include <string>
#include <vector>
#include "cudaaligner_test_cases.hpp"
#include "needleman_wunsch_cpu.hpp"
using namespace claraparabricks::genomeworks;
int main()
{
    for (const TestCaseData& t : create_cudaaligner_test_cases())
    {
        auto m = cudaaligner::needleman_wunsch_build_score_matrix_naive(t.target, t.query);
        std::printf("%s\t%s\t%d\n", t.target.c_str(), t.query.c_str(), (int)m(m.num_rows() - 1, m.num_cols() - 1));
    }
    return 0;
}
'''


def must_contain(text, literal, where):
    if literal not in text:
        sys.exit("literal %r not found in %s" % (literal, where))


def test_pairs():
    with tempfile.TemporaryDirectory() as tmp:
        stub = os.path.join(tmp, "stub")
        os.makedirs(stub)
        with open(os.path.join(stub, "cuda_runtime_api.h"), "w") as f:
            f.write("#pragma once\n#define __host__\n#define __device__\n#define __forceinline__ inline\ntypedef void* cudaStream_t;\n")
        src = os.path.join(tmp, "printer.cpp")
        with open(src, "w") as f:
            f.write(PRINTER)
        exe = os.path.join(tmp, "printer")
        subprocess.run(["g++", "-O2", "-std=c++17", "-I", stub, "-I", os.path.join(REF, "common", "base", "include"),
                        "-I", os.path.join(REF, "cudaaligner", "include"), "-I", os.path.join(REF, "cudaaligner", "src"), "-I", T,
                        "-o", exe, src, os.path.join(T, "cudaaligner_test_cases.cpp"),
                        os.path.join(REF, "cudaaligner", "src", "needleman_wunsch_cpu.cpp")], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    pairs = []
    for line in out.split("\n"):
        if not line:
            continue
        target, query, dist = line.split("\t")
        pairs.append(dict(target=target, query=query, edit_distance=int(dist)))
    assert len(pairs) == 21, len(pairs)
    return pairs


def alignment_impl_cases():
    src = open(os.path.join(T, "Test_AlignmentImpl.cpp")).read()
    M, X, I, D = 0, 1, 2, 3  # AlignmentState
    cases = [
        dict(query="AAAA", target="TTATG", alignment=[X, X, M, X, I], is_optimal=True,
             formatted=["AAAA-", "xx|x ", "TTATG"], cigar_basic="4M1I", cigar_extended="2X1=1X1I"),
        dict(query="CGATAATG", target="CATAA", alignment=[D, X, M, M, M, M, D, D], is_optimal=True,
             formatted=["CGATAATG", " x||||  ", "-CATAA--"], cigar_basic="1D5M2D", cigar_extended="1D1X4=2D"),
        dict(query="GTTAG", target="AAGTCTAGAA", alignment=[I, I, M, M, I, M, M, M, I, I], is_optimal=True,
             formatted=["--GT-TAG--", "  || |||  ", "AAGTCTAGAA"], cigar_basic="2I2M1I3M2I", cigar_extended="2I2=1I3=2I"),
        dict(query="GTTACA", target="GATTCA", alignment=[M, I, M, M, D, M, M], is_optimal=False,
             formatted=["G-TTACA", "| || ||", "GATT-CA"], cigar_basic="1M1I2M1D2M", cigar_extended="1=1I2=1D2="),
    ]
    names = {M: "match", X: "mismatch", I: "insertion", D: "deletion"}
    for c in cases:
        must_contain(src, 'data.query     = "%s";' % c["query"], "Test_AlignmentImpl.cpp")
        must_contain(src, 'data.target    = "%s";' % c["target"], "Test_AlignmentImpl.cpp")
        must_contain(src, 'FormattedAlignment{"%s", "%s", "%s"}' % tuple(c["formatted"]), "Test_AlignmentImpl.cpp")
        must_contain(src, '"%s"' % c["cigar_basic"], "Test_AlignmentImpl.cpp")
        must_contain(src, '"%s"' % c["cigar_extended"], "Test_AlignmentImpl.cpp")
        # the state list, in order, as it is written in the source
        block = src[src.index('data.query     = "%s";' % c["query"]):]
        block = block[:block.index("data.is_optimal")]
        states = re.findall(r"AlignmentState::(\w+)", block)
        assert states == [names[s] for s in c["alignment"]], (c["query"], states)
    return cases


def pattern_words():
    src = open(os.path.join(T, "Test_HirschbergMyers.cu")).read()
    m = re.search(r'std::string query =\s*"([ACGT]+)"\s*"([ACGT]+)"\s*"([ACGT]+)";', src)
    query = "".join(m.groups())
    words = {}
    for r, c, bits in re.findall(r"EXPECT_EQ\(patterns\((\d), (\d)\), 0b([01]+)u\);", src):
        words["%s,%s" % (r, c)] = int(bits, 2)
    assert len(words) == 16, len(words)
    return dict(query=query, words=words, source="Test_HirschbergMyers.cu:100-123 (A=0, C=1, T=2, G=3; +4 reversed)")


def main():
    out = dict(test_pairs=test_pairs(), alignment_impl=alignment_impl_cases(), hirschberg_patterns=pattern_words(),
               source="cudaaligner/tests/{cudaaligner_test_cases.cpp:25-99, Test_AlignmentImpl.cpp:66-142, Test_HirschbergMyers.cu:100-123}")
    with open(os.path.join(HERE, "cudaaligner_vectors.json"), "w") as f:
        json.dump(out, f, indent=0)
        f.write("\n")
    print("pairs", len(out["test_pairs"]), [(len(p["query"]), len(p["target"]), p["edit_distance"]) for p in out["test_pairs"]])


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Regenerate tests/golden/cudapoa_vectors.json from the reference's own test sources.

Runs only in the build container (needs /root/reference). The small hand graphs are transcribed from the
cited initializer lists; every expected-answer literal is checked to occur verbatim in the cited source file
so a transcription slip fails loudly. The two long NWbandedTest strings are extracted by regex.
"""
import json, os, re, sys

REF = os.environ.get("GW_REFERENCE", "/root/reference")
T = os.path.join(REF, "cudapoa", "tests")


def src(name):
    with open(os.path.join(T, name)) as f:
        return f.read()


def must_contain(text, literal, where):
    if literal not in text:
        sys.exit(f"literal {literal!r} not found in {where}")


nw_src = src("Test_CudapoaNW.cu")
nw_cases = [  # Test_CudapoaNW.cu:100-187
    dict(name="NW1", nodes="AAAA", sorted=[0, 1, 2, 3], outgoing=[[1], [2], [3], []], read="AATA",
         graph_ans="3,2,1,0", read_ans="3,2,1,0"),
    dict(name="NW2", nodes="ATCG", sorted=[0, 1, 2, 3], outgoing=[[1], [2], [3], []], read="ATCGA",
         graph_ans="-1,3,2,1,0", read_ans="4,3,2,1,0"),
    dict(name="NW3", nodes="AACGC", sorted=[0, 4, 1, 2, 3], outgoing=[[1, 4], [2], [3], [], [2]], read="ATCG",
         graph_ans="3,2,1,0", read_ans="3,2,1,0"),
    dict(name="NW4", nodes="ATTGA", sorted=[0, 1, 2, 3, 4], outgoing=[[1], [2], [3], [4], []], read="AA",
         graph_ans="4,3,2,1,0", read_ans="1,-1,-1,-1,0"),
    dict(name="NW5", nodes="ATGTACA", sorted=[0, 5, 1, 6, 2, 3, 4], outgoing=[[1, 5], [2], [3], [4], [], [6], [3]],
         read="ACTTA", graph_ans="4,3,6,5,0", read_ans="4,3,2,1,0"),
]
for c in nw_cases:
    must_contain(nw_src, f'("{c["graph_ans"]}", "{c["read_ans"]}")', "Test_CudapoaNW.cu")

m_nodes = re.search(r'std::string nodes_str\s*=\s*"([ACGT]+)"', nw_src)
m_read = re.search(r'std::string read_str\s*=\s*"([ACGT]+)"', nw_src)
nwb = dict(nodes=m_nodes.group(1), read=m_read.group(1),  # Test_CudapoaNW.cu:453-454
           config=dict(max_sequence_size=1024, max_sequences_per_poa=2, band_width=128))  # :326-327

ts_src = src("Test_CudapoaTopSort.cu")
topsort_cases = [  # Test_CudapoaTopSort.cu:48-58
    dict(outgoing=[[], [], [3], [1], [0, 1], [0, 2]], answer="4-5-0-2-3-1"),
    dict(outgoing=[[1, 3], [2, 3], [3, 4, 5], [4, 5], [5], []], answer="0-1-2-3-4-5"),
    dict(outgoing=[[], [], [3], [1], [0, 1, 7], [0, 2], [4], [5]], answer="6-4-7-5-0-2-3-1"),
]
for c in topsort_cases:
    must_contain(ts_src, f'"{c["answer"]}"', "Test_CudapoaTopSort.cu")

aa_src = src("Test_CudapoaAddAlignment.cu")
add_cases = [  # Test_CudapoaAddAlignment.cu:127-229 (harness :233-340: default BatchConfig, s=0, non-MSA)
    dict(nodes="AAAA", outgoing=[[], [0], [1], [2]], coverage=[1, 1, 1, 1], read="AATA", weights=[0, 0, 1, 2],
         alignment_graph=[0, 1, 2, 3], alignment_read=[0, 1, 2, 3], answer=[[], [0], [1], [2, 4], [1]]),
    dict(nodes="ATCG", outgoing=[[], [0], [1], [2]], coverage=[1, 1, 1, 1], read="ATCGA", weights=[0, 1, 2, 3, 4],
         alignment_graph=[0, 1, 2, 3, -1], alignment_read=[0, 1, 2, 3, 4], answer=[[], [0], [1], [2], [3]]),
    dict(nodes="AACGC", outgoing=[[], [0], [1, 4], [2], [0]], coverage=[2, 1, 2, 2, 1], read="ATCG", weights=[0, 1, 1, 5],
         alignment_graph=[0, 4, 2, 3], alignment_read=[0, 1, 2, 3], answer=[[], [0], [1, 4, 5], [2], [0], [0]]),
    dict(nodes="ATTGA", outgoing=[[], [0], [1], [2], [3]], coverage=[1, 1, 1, 1, 1], read="AA", weights=[5, 1],
         alignment_graph=[0, 1, 2, 3, 4], alignment_read=[0, -1, -1, -1, 1], answer=[[], [0], [1], [2], [3, 0]]),
    dict(nodes="ATGTACA", outgoing=[[], [0], [1], [2, 6], [3], [0], [5]], coverage=[2, 1, 1, 2, 2, 1, 1], read="ACTTA",
         weights=[10, 9, 8, 7, 6], alignment_graph=[0, 5, 6, 3, 4], alignment_read=[0, 1, 2, 3, 4],
         answer=[[], [0], [1], [2, 6, 7], [3], [0], [5], [5]]),
]
for c in add_cases:
    lit = "Int16Vec2D({" + ", ".join("{" + ", ".join(map(str, r)) + "}" for r in c["answer"]) + "})"
    must_contain(aa_src, lit, "Test_CudapoaAddAlignment.cu")

gc_src = src("Test_CudapoaGenerateConsensus.cu")
consensus_cases = [  # Test_CudapoaGenerateConsensus.cu:95-160; weight placement quirk :62-73
    dict(nodes="AAAAT", sorted=[0, 1, 2, 4, 3], node_alignments=[[], [], [4], [], [2]],
         outgoing=[[1], [2, 4], [3], [], [3]], coverage=[2, 2, 1, 2, 1], outgoing_w=[[5], [4, 3], [2], [], [1]], answer="ATAA"),
    dict(nodes="ATCGA", sorted=[0, 1, 2, 3, 4], node_alignments=[[], [], [], [], []],
         outgoing=[[1], [2], [3], [4], []], coverage=[1, 1, 1, 1, 1], outgoing_w=[[4], [3], [2], [1], []], answer="AGCTA"),
    dict(nodes="AACGCT", sorted=[0, 1, 4, 5, 2, 3], node_alignments=[[], [4, 5], [], [], [1, 5], [1, 4]],
         outgoing=[[1, 4, 5], [2], [3], [], [2], [2]], coverage=[3, 1, 3, 3, 1, 1],
         outgoing_w=[[7, 6, 5], [4], [3], [], [2], [1]], answer="GCCA"),
    dict(nodes="ATTGA", sorted=[0, 1, 2, 3, 4], node_alignments=[[], [], [], [], []],
         outgoing=[[1, 4], [2], [3], [4], []], coverage=[2, 1, 1, 1, 2], outgoing_w=[[5, 4], [3], [2], [1], []], answer="AGTTA"),
    dict(nodes="ATGTACAT", sorted=[0, 1, 5, 2, 6, 7, 3, 4], node_alignments=[[], [5], [6, 7], [], [], [1], [2, 7], [2, 6]],
         outgoing=[[1, 5], [2], [3], [4], [], [6, 7], [3], [3]], coverage=[3, 1, 1, 3, 3, 2, 1, 1],
         outgoing_w=[[9, 8], [7], [6], [5], [], [4, 3], [2], [1]], answer="ATTCA"),
]
for c in consensus_cases:
    must_contain(gc_src, f'= "{c["answer"]}";', "Test_CudapoaGenerateConsensus.cu")

out = dict(
    _source="generated by tests/golden/make_reference_vectors.py from /root/reference/cudapoa/tests (v0.6.0)",
    scores=dict(gap=-8, mismatch=-6, match=8),
    nw=nw_cases, nw_banded=nwb, topsort=topsort_cases, add_alignment=add_cases, consensus=consensus_cases,
)
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cudapoa_vectors.json")
with open(dst, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", dst, "nwb lens", len(nwb["nodes"]), len(nwb["read"]))

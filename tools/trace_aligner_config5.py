#!/usr/bin/env python3
"""Host-side timeline of configs[4] (1 000 000 pairs x 150 bp): GW_ALIGNER_TRACE=1 python tools/trace_aligner_config5.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomeworks_amd import cudaaligner, synthetic

CONFIG5 = dict(seed=3, pairs=int(os.environ.get("PAIRS", 1000000)), length=150, mut=2, ins=1, dele=1, max_bandwidth=150)
pairs = synthetic.generate_pairs(CONFIG5["seed"], CONFIG5["pairs"], CONFIG5["length"], CONFIG5["mut"], CONFIG5["ins"], CONFIG5["dele"])
al = cudaaligner.CudaAlignerBatch(max_bandwidth=CONFIG5["max_bandwidth"], max_device_memory_allocator_caching_size=32 << 30)
add = al._L.gw_aligner_add_alignment
for rep in range(4):
    al.reset()
    for q, t in pairs:
        assert add(al._h, q, len(q), t, len(t), 0, 0) == 0
    sys.stderr.write("---- rep %d\n" % rep)
    t0 = time.perf_counter()
    al.align_all()
    t1 = time.perf_counter()
    n = al.sync()
    t2 = time.perf_counter()
    sys.stderr.write("align_all %.3f ms, sync %.3f ms, total %.3f ms, %d alignments\n" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, n))

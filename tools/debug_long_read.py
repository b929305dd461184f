#!/usr/bin/env python3
"""Debug aid: synthetic long-read windows (tools/bench_long_read_msa.py make_window) through ONE GPU batch and the CPU
oracle under an explicit BatchConfig; prints status / node count of both per window.
  python tools/debug_long_read.py <first> <count> <max_sequence_size> <max_sequences_per_poa> [gpu_mem_gib]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_long_read_msa as B  # noqa: E402
import oracle_poa as O  # noqa: E402
from genomeworks_amd import cudapoa  # noqa: E402

first, count, max_seq, max_seqs = (int(x) for x in sys.argv[1:5])
mem = int(sys.argv[5]) if len(sys.argv) > 5 else 8
ocfg = O.make_cfg(max_seq, max_seqs, 256, 2, output_mask=2)
batch = cudapoa.CudaPoaBatch(max_seqs, max_seq, mem << 30, output_type="msa", band_mode="adaptive_band",
                             max_consensus_size=ocfg.max_consensus_size, max_nodes_per_graph=ocfg.max_nodes_per_graph,
                             matrix_sequence_dimension=ocfg.matrix_sequence_dimension,
                             max_banded_pred_distance=ocfg.max_banded_pred_distance)
if os.environ.get("PRELAUNCH"):  # a first launch with other windows, then reset: exposes state carried over by a reused batch
    for w in range(first + count, first + 2 * count):
        reads = [s for s in B.make_window(w, 32768) if len(s) < max_seq][:max_seqs]
        if reads:
            batch.add_poa_group(reads)
    batch.generate_poa()
    batch.get_msa()
    batch.reset()
wins, ids = [], []
order = range(first, first + count)
if os.environ.get("ORDER"):
    order = [int(x) for x in os.environ["ORDER"].split(",")]
for w in order:
    reads = [s for s in B.make_window(w, 32768) if len(s) < 32768]
    if not os.environ.get("RAW"):
        reads = [s for s in reads if len(s) < max_seq][:max_seqs]
    if not reads:
        continue
    st, seq_st = batch.add_poa_group(reads)
    if st != 0:
        print("window %d not added: %d" % (w, st))
        continue
    wins.append([s for s, ss in zip(reads, seq_st) if ss == 0])
    ids.append(w)
batch.generate_poa()
msa, status = batch.get_msa()
bad = 0
with O.Workspace(ocfg) as ws:
    for k, (w, reads) in enumerate(zip(ids, wins)):
        if status[k] == 0 and os.environ.get("ONLY_FAILED"):
            continue
        ref = ws.process(reads)
        ok = status[k] == ref["status"] and (status[k] != 0 or msa[k] == ref["msa"])
        bad += not ok
        print("window %4d slot %3d reads %2d longest %5d: gpu %d cpu %d nodes %d %s" % (
            w, k, len(reads), max(map(len, reads)), status[k], ref["status"], ref["node_count"], "" if ok else "  <-- MISMATCH"))
print("mismatches:", bad, "of", len(ids))

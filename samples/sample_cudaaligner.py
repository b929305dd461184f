#!/usr/bin/env python3
"""Global alignment of random query/target pairs with the Python API (counterpart of
pygenomeworks/samples/sample_cudaaligner): pairs are queued until the batch reports exceeded_max_alignments,
aligned, read back and the batch reset.  usage: sample_cudaaligner.py [-n pairs] [-l length] [-p]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomeworks_amd import cudaaligner, synthetic  # noqa: E402


def run(pairs, max_bandwidth, print_output):
    batch = cudaaligner.CudaAlignerBatch(max_bandwidth=max_bandwidth, max_device_memory_allocator_caching_size=2 << 30)
    aligned = 0

    def drain():
        nonlocal aligned
        batch.align_all()
        for a in batch.get_alignments():
            aligned += 1
            if print_output:
                q, m, t = a.format_alignment()
                print("%s\n%s\n%s\ncigar %s  edit distance %d\n" % (q, m, t, a.cigar, a.edit_distance))
        batch.reset()

    for q, t in pairs:
        st = batch.add_alignment(q.decode(), t.decode())
        if st == cudaaligner.exceeded_max_alignments:
            drain()
            st = batch.add_alignment(q.decode(), t.decode())
        if st != cudaaligner.success:
            print("pair skipped: %s" % cudaaligner.status_to_str(st), file=sys.stderr)
    drain()
    print("aligned %d pairs" % aligned, file=sys.stderr)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", "--pairs", type=int, default=100)
    ap.add_argument("-l", "--length", type=int, default=200)
    ap.add_argument("-w", "--max-bandwidth", type=int, default=256)
    ap.add_argument("-p", "--print", dest="print_output", action="store_true")
    a = ap.parse_args()
    run(synthetic.generate_pairs(7, a.pairs, a.length, a.length // 20, a.length // 40, a.length // 40), a.max_bandwidth, a.print_output)

#!/usr/bin/env python3
"""Consensus / MSA for the POA groups of a window file (or of a few synthetic windows) with the Python API.

Counterpart of pygenomeworks/samples/sample_cudapoa: the batch is sized from the groups, filled until it reports
exceeded_maximum_poas, run, drained and reset.  usage: sample_cudapoa.py [-i windows.txt] [-m] [-p]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomeworks_amd import cuda, cudapoa, synthetic  # noqa: E402


def run_cudapoa(groups, msa, print_output):
    free, _total = cuda.cuda_get_mem_info(cuda.cuda_get_device())
    # 90 % of the free memory as in the reference sample, capped: a few windows do not need 250 GB
    batch = cudapoa.CudaPoaBatch(max(len(g) for g in groups), max(len(s) for g in groups for s in g) + 1,
                                 min(0.9 * free, 4 << 30),
                                 output_type="msa" if msa else "consensus", band_mode="static_band")
    done, first = 0, 0

    def drain(upto):
        batch.generate_poa()
        if msa:
            results, status = batch.get_msa()
        else:
            results, _coverage, status = batch.get_consensus()
        for g, (r, st) in enumerate(zip(results, status)):
            if st != cudapoa.success:
                print("group %d: %s" % (first + g, cudapoa.status_to_str(st)), file=sys.stderr)
            elif print_output:
                print("\n".join(r) if msa else r)
        batch.reset()
        print("Processed groups %d - %d" % (first, upto), file=sys.stderr)

    i = 0
    while i < len(groups):
        status, _seq_status = batch.add_poa_group(groups[i])
        if status == cudapoa.exceeded_maximum_poas:
            drain(i - 1)
            first = i
            continue
        if status != cudapoa.success:
            print("group %d skipped: %s" % (i, cudapoa.status_to_str(status)), file=sys.stderr)
        i += 1
        done += 1
    if batch.total_poas > 0:
        drain(len(groups) - 1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-i", "--input", help="cudapoa window file (default: 8 synthetic windows)")
    ap.add_argument("-m", "--msa", action="store_true")
    ap.add_argument("-p", "--print", dest="print_output", action="store_true")
    args = ap.parse_args()
    if args.input:
        groups = cudapoa.parse_cudapoa_file(args.input)
    else:
        groups = [[r.decode() for r in synthetic.generate_window(100 + w, 400, 12, 20, 8, 8)] for w in range(8)]
    run_cudapoa(groups, args.msa, args.print_output)

#!/usr/bin/env python3
"""BASELINE configs[3] on MI355X: cudapoa long-read MSA, adaptive band, through get_multi_batch_sizes and the
multi-batch loop (the flow of cudapoa/src/main.cpp:197-326), on synthetic windows (SURVEY.md 8(d) "Config 4"):

  window w (seed 2000 + w): 8..32 reads, backbone length log-uniform in [2 k, 32 k], 8-12 % divergence split
  1 : 2 : 2 over substitutions : insertions : deletions; band 256, adaptive_band, MSA output
  => BatchConfig types <int32, int32, int16> (32-bit scores and node ids), HBM row tables, multi-pass bands.

Timed region per batch: generate_poa() + get_msa() with the groups already added (H2D included in generate_poa, as
the reference benchmark times it). Cells are the kernels' own count (every band pass of every read).

  python tools/bench_long_read_msa.py [--windows 48] [--max-len 32768] [--check 2] [--json out.json]
  multi-GPU: launch under torchrun; rank r takes windows r, r + world, ... (index split, no collective)
"""
import argparse
import json
import math
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genomeworks_amd import cuda, cudapoa, synthetic  # noqa: E402


def make_window(w, max_len):
    rng = random.Random(2000 + w)
    n_reads = rng.randint(8, 32)
    backbone = int(round(math.exp(rng.uniform(math.log(2000), math.log(max_len * 0.93)))))
    div = rng.uniform(0.08, 0.12)
    # the generator fires each of its max_* trials with p = 0.5 (genomeutils.hpp:47-127): 2 x for the expected count
    mut, ins, dele = (int(2 * backbone * div * f) for f in (0.2, 0.4, 0.4))
    return [r.decode() for r in synthetic.generate_window(2000 + w, backbone, n_reads, mut, ins, dele)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=48)
    ap.add_argument("--max-len", type=int, default=32768)
    ap.add_argument("--check", type=int, default=0, help="verify this many (shortest) windows against the CPU oracle")
    ap.add_argument("--json")
    ap.add_argument("--kernel-time", action="store_true", help="one extra launch per batch, timed with HIP events")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cuda.cuda_set_device(local_rank)

    ids = list(range(rank, args.windows, world))
    windows = [make_window(w, args.max_len) for w in ids]
    windows = [[s for s in g if len(s) < args.max_len] for g in windows]
    cfgs, groups = cudapoa.get_multi_batch_sizes(windows, msa_flag=True, band_width=256, band_mode="adaptive_band")
    free, _total = cuda.cuda_get_mem_info(local_rank)

    msa_of, cfg_of, accepted, slot_of = {}, {}, {}, {}
    placed = sorted(g for members in groups for g in members)
    assert placed == sorted(set(placed)), "get_multi_batch_sizes placed a window twice"
    rows, cells_total, t_total, n_done = [], 0, 0.0, 0
    for cfg, members in zip(cfgs, groups):
        batch = cudapoa.CudaPoaBatch(cfg["max_sequences_per_poa"], cfg["max_sequence_size"], int(0.9 * free),
                                     output_type="msa", band_mode="adaptive_band", device_id=local_rank,
                                     alignment_band_width=cfg["alignment_band_width"],
                                     max_consensus_size=cfg["max_consensus_size"],
                                     max_nodes_per_graph=cfg["max_nodes_per_graph"],
                                     matrix_sequence_dimension=cfg["matrix_sequence_dimension"],
                                     max_banded_pred_distance=cfg["max_banded_pred_distance"])
        pending = list(members)
        if os.environ.get("GW_DEBUG_ORDER"):
            print("order", cfg, [ids[g] for g in members][:int(os.environ["GW_DEBUG_ORDER"])], flush=True)
        bin_cells, bin_t, bin_n, launches, kernel_ms = 0, 0.0, 0, 0, 0.0
        while pending:
            taken = []
            while pending:
                st, seq_st = batch.add_poa_group(windows[pending[0]])
                if st == cudapoa.exceeded_maximum_poas:
                    break
                g = pending.pop(0)
                if st == cudapoa.success:
                    taken.append(g)
                    # a bin's BatchConfig may hold fewer reads than its deepest window: the batch keeps the reads it
                    # accepted (the others report exceeded_maximum_sequences_per_poa), and so does the oracle check
                    accepted[g] = [s for s, ss in zip(windows[g], seq_st) if ss == cudapoa.success]
                else:
                    print("window %d skipped: %s" % (ids[g], cudapoa.status_to_str(st)), file=sys.stderr)
                    if st == cudapoa.empty_poa_group and windows[g]:
                        # every read was rejected AFTER the batch opened a POA for the group: as in the reference
                        # (cudapoa_batch.cuh:122-150) that empty POA stays in the batch and owns an output slot
                        taken.append(None)
            if not any(g is not None for g in taken):
                raise RuntimeError("a batch of this bin cannot hold a single window")
            t0 = time.perf_counter()
            batch.generate_poa()
            n_out = batch.get_msa_native()  # D2H + row unpack in the library = Batch::get_msa
            dt = time.perf_counter() - t0
            msa, status = batch.collect_msa(n_out)  # Python marshalling, not part of the reference's timed region
            lens = [max(len(r) for r in windows[g]) for g in taken if g is not None]
            print("[rank %d]   launch %d: %d windows, longest read %d..%d, %.1f ms" % (rank, launches, len(lens), min(lens), max(lens), dt * 1e3),
                  file=sys.stderr, flush=True)
            if args.kernel_time:
                k_ms, o_ms = batch.relaunch_timed()
                kernel_ms += k_ms + o_ms
            assert len(msa) == len(status) == len(taken)
            for g, m, st in zip(taken, msa, status):
                if g is None:
                    continue
                if st != cudapoa.success:
                    print("window %d: %s" % (ids[g], cudapoa.status_to_str(st)), file=sys.stderr)
                msa_of[g] = (m, st)
                cfg_of[g] = cfg
                slot_of[g] = (launches, taken.index(g))
            bin_cells += batch.total_cells()
            bin_t += dt
            bin_n += sum(1 for g in taken if g is not None)
            launches += 1
            batch.reset()
        rows.append({"max_sequence_size": cfg["max_sequence_size"], "max_nodes_per_graph": cfg["max_nodes_per_graph"],
                     "windows": bin_n, "launches": launches, "cells": bin_cells, "ms": round(bin_t * 1e3, 2),
                     "gcups": round(bin_cells / bin_t / 1e9, 3) if bin_t else None,
                     "kernels_ms": round(kernel_ms, 2) if args.kernel_time else None})
        print("[rank %d] bin max_seq %6d: %3d windows in %d launch(es), %.3e cells, %9.1f ms, %7.2f GCUPS"
              % (rank, cfg["max_sequence_size"], bin_n, launches, bin_cells, bin_t * 1e3, bin_cells / bin_t / 1e9), flush=True)
        cells_total += bin_cells
        t_total += bin_t
        n_done += bin_n
        del batch

    checked = 0
    if args.check:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_poa as O
        order = sorted(msa_of, key=lambda g: sum(len(s) for s in windows[g]))[:args.check]
        order += [g for g in sorted(msa_of) if msa_of[g][1] != cudapoa.success and g not in order][:2]  # error statuses too
        for g in order:
            c = cfg_of[g]  # the oracle runs the window under the BatchConfig of the bin it was placed in
            ocfg = O.make_cfg(c["max_sequence_size"], c["max_sequences_per_poa"], 256, 2, output_mask=2)  # 2 = adaptive_band
            assert (ocfg.max_nodes_per_graph, ocfg.matrix_sequence_dimension) == (c["max_nodes_per_graph"], c["matrix_sequence_dimension"])
            with O.Workspace(ocfg) as ws:
                ref = ws.process(accepted[g])
            got, st = msa_of[g]
            assert st == ref["status"], (ids[g], st, ref["status"], slot_of[g], len(accepted[g]), len(windows[g]))
            if st == cudapoa.success:
                assert got == ref["msa"], "window %d: MSA differs from the oracle" % ids[g]
            checked += 1
        print("[rank %d] %d window(s) bit-exact vs the CPU oracle" % (rank, checked), flush=True)

    out = {"config": "BASELINE configs[3]: long-read MSA, adaptive band 256, synthetic windows (seeds 2000+w)",
           "rank": rank, "world": world, "windows": n_done, "cells": cells_total, "ms": round(t_total * 1e3, 2),
           "gcups": round(cells_total / t_total / 1e9, 3), "windows_per_s": round(n_done / t_total, 3),
           "checked_vs_oracle": checked, "bins": rows,
           "reads_per_window": [len(g) for g in windows], "longest_read": [max(len(s) for s in g) for g in windows]}
    print(json.dumps({k: v for k, v in out.items() if k not in ("reads_per_window", "longest_read")}))
    if args.json:
        with open(args.json if world == 1 else "%s.rank%d" % (args.json, rank), "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

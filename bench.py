#!/usr/bin/env python3
"""bench.py -- the north-star metric on MI355X: GCUPS (+ POA windows/s) of the cudapoa consensus hot path on the
1024-window short-read batch (BASELINE.json configs[2]: 1024 windows x 32 reads <= 1024 bp, static band 256).

A "step" = one pass of the hot path over one 1024-window batch with the inputs already resident in HBM:
graph-build kernel (NW + merge + topsort per read) + consensus kernel + D2H of the consensus/coverage and host
un-reversal (get_consensus). Batch filling and the H2D upload are outside the timed region (the PCIe-inclusive
rate is printed as `pcie_inclusive_*`, never as `value`).

Multi-GPU (torchrun, one rank per GPU): windows are independent, so each rank runs its own 1024-window batch
(seeds rank*1024 + w): weak scaling, no data-path collective; RCCL is used only for the barrier and the max
over ranks of the elapsed time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_CELL = 4             # 2 x sizeof(int16) per DP cell (one write, one predecessor read)
WINDOWS = 1024


_CPU_SHARED = {}


def _cpu_worker(k):
    """One host core: its own oracle workspace over its share of the windows, again and again for `seconds`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_poa as O
    windows, cores, seconds = _CPU_SHARED["windows"], _CPU_SHARED["cores"], _CPU_SHARED["seconds"]
    share = windows[k::cores] or windows[:1]
    cells = n = 0
    with O.Workspace(O.make_cfg(1024, 32, 256, 1)) as ws:
        t0 = time.perf_counter()
        while True:
            for w in share:
                cells += ws.process(w)["cells"]
                n += 1
                if time.perf_counter() - t0 > seconds:
                    return cells, n, time.perf_counter() - t0


def cpu_baseline(windows, seconds=6.0):
    """CPU oracle (port of the reference semantics, scalar C) on the GPU box's host cores: one process per core, each
    with its own workspace over a share of the same windows for `seconds` (`value` = sum of the per-core rates,
    `cores`), next to the single-core rate (`single_core`). spoa, the CPU path the reference names, is not vendored
    in its checkout. Runs before the first device call of the process (the workers are forked)."""
    import multiprocessing as mp
    cores = max(1, os.cpu_count() or 1)
    _CPU_SHARED.update(windows=windows, cores=1, seconds=seconds)
    c1, n1, dt1 = _cpu_worker(0)
    _CPU_SHARED.update(cores=cores)
    try:
        with mp.get_context("fork").Pool(cores) as pool:
            res = pool.map(_cpu_worker, range(cores), chunksize=1)
    except Exception as e:  # no room for the workers (process limits): the single-core rate is the baseline then
        print("bench.py: all-core CPU baseline unavailable (%s), reporting one core" % e, file=sys.stderr)
        cores, res = 1, [(c1, n1, dt1)]
    rate = sum(c / dt for c, _, dt in res)
    return {"value": round(rate / 1e9, 4), "unit": "GCUPS", "cores": cores, "kind": "port",
            "windows_per_s": round(sum(n / dt for _, n, dt in res), 3),
            "sample": "%d window passes over %d processes, %.1f s each, drawn from the %d config-3 windows (gcc -O2 scalar oracle)"
                      % (sum(n for _, n, _ in res), cores, seconds, len(windows)),
            "single_core": {"value": round(c1 / dt1 / 1e9, 4), "windows_per_s": round(n1 / dt1, 3),
                            "sample": "%d windows, %.1f s" % (n1, dt1)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--windows", type=int, default=WINDOWS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    from genomeworks_amd import synthetic
    first_seed = 1000 + rank * args.windows
    windows = [[r.decode() for r in synthetic.generate_window(first_seed + w)] for w in range(args.windows)]
    # the CPU baseline forks one worker per core: before this process makes its first device call
    cpu = cpu_baseline(windows) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from genomeworks_amd import cudapoa
    from genomeworks_amd.cuda import cuda_set_device
    cuda_set_device(local_rank)

    batch = cudapoa.CudaPoaBatch(32, 1024, 8 << 30, output_type="consensus", band_mode="static_band",
                                 alignment_band_width=256, max_nodes_per_graph=3072, device_id=local_rank)
    for w in windows:
        st, _ = batch.add_poa_group(w)
        assert st == 0, st

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # first pass includes the H2D upload: PCIe-inclusive rate, reported separately
    sync()
    t0 = time.perf_counter()
    batch.generate_poa()
    n_ok = batch.get_consensus_native()
    t_pcie = time.perf_counter() - t0
    cells = batch.total_cells()
    assert n_ok == args.windows

    for _ in range(args.warmup):
        batch.relaunch()
        batch.get_consensus_native()

    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.relaunch()
        batch.get_consensus_native()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([float(cells)], device="cuda", dtype=torch.float64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_cells = float(c.item())
    else:
        total_cells = float(cells)

    # dominant kernel, timed live with HIP events on the batch's own stream (outside the timed region above)
    kms, oms = [], []
    for _ in range(max(3, args.steps)):
        a, b = batch.relaunch_timed()
        kms.append(a)
        oms.append(b)
    k_ms = sum(kms) / len(kms)
    o_ms = sum(oms) / len(oms)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        gcups = total_cells * args.steps / elapsed / 1e9
        achieved = cells * BYTES_PER_CELL / (k_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC pass of this same workload (rocprofv3 cannot run inside the
        # timed region; tools/pmc_passes.sh collects FETCH_SIZE / WRITE_SIZE in their own runs)
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if pmc.get("windows") == args.windows:
                traffic = pmc["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            traffic = None
        out = {
            "metric": "cudapoa consensus GCUPS, 1024-window short-read batch (static band 256, 32 reads <= 1024 bp)",
            "value": round(gcups, 3), "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16", "data": "synthetic",
            "windows_per_s": round(world * args.windows * args.steps / elapsed, 1),
            "config": {"workload": "BASELINE configs[2]: cudapoa single-batch consensus, %d windows x 32 reads, "
                                   "backbone 960 bp, <=48 sub/24 ins/24 del, BatchConfig(1024,32,256,static_band), "
                                   "scores 8/-6/-8" % args.windows,
                       "windows_per_gpu": args.windows, "cells_per_gpu": cells, "parallelism": "index-split x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "poa_window_kernel<int16,int16,static_band>",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": cells * BYTES_PER_CELL,
                         "kernel_ms": round(k_ms, 3), "output_kernel_ms": round(o_ms, 3),
                         "algorithmic_bytes_per_cell": BYTES_PER_CELL},
            "pcie_inclusive_ms": round(t_pcie * 1e3, 3),
            "pcie_inclusive_gcups": round(cells / t_pcie / 1e9, 3),
        }
        if cpu is not None:  # timed on rank 0 of the 1-GPU run only
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

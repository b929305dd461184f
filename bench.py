#!/usr/bin/env python3
"""bench.py -- the north-star metric on MI355X: GCUPS (+ POA windows/s) of the cudapoa consensus hot path on the
1024-window short-read batch (BASELINE.json configs[2]: 1024 windows x 32 reads <= 1024 bp, static band 256).

A "step" is the timed region of the reference's own benchmark (cudapoa/benchmarks/single_batch.hpp:86-93):
generate_poa() -- H2D of the batch from pinned host memory, graph-build kernel (NW + merge + topsort per read),
consensus kernel -- followed by get_consensus() (D2H + host un-reversal), in a steady-state loop on one Batch object.
Batch filling (add_poa_group) is outside, as in the reference. `value` is that loop; the same kernels on inputs already
resident in HBM are reported next to it as `kernel_only` (no H2D), and the dominant kernel's own duration (HIP events
on the batch's stream) feeds `roofline`.

The same JSON line carries sub-records for the other BASELINE configs, each with its own roofline and CPU baseline:
  configs[1]  cudaaligner banded Myers, 10 000 pairs x 1 kbp (align_all + sync_alignments; kernels only; device resident)
  configs[4]  cudaaligner 1 000 000 pairs x 150 bp, index-split over the ranks
  configs[3]  cudapoa long-read MSA, adaptive band, 598 windows through the multi-batch loop, batch-sharded over the ranks
and, with more than one GPU, a strong-scaling record of the metric config (the same 1024 windows split over the ranks
by estimated cost, consensus gathered by global index and hashed against the committed oracle golden).

Multi-GPU (torchrun, one rank per GPU): windows and pairs are independent, so there is no data-path collective. The
headline is weak scaling -- each rank runs the SAME 1024 windows (seeds 1000 + w: the batch the oracle golden covers, so
every rank's output is checked inside the run) -- because one
window is one chain of 31 dependent alignments: 1024 windows already leave a single MI355X at one wavefront per SIMD,
and splitting them further only idles SIMDs (the strong-scaling record shows exactly that). RCCL carries the barrier
and the max / sum reductions of the timing scalars only.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_CELL = 4             # 2 x sizeof(int16) per DP cell (one write, one predecessor read)
BYTES_PER_CELL_LONG = 8        # long reads: 2 x sizeof(int32)
BYTES_PER_MYERS_CELL = 0.375   # 12 B (pv, mv, score) per 32-cell band word and column
WINDOWS = 1024
CONFIG2 = dict(seed=1, pairs=10000, length=1000, mut=33, ins=33, dele=33, max_bandwidth=1024)
CONFIG5 = dict(seed=3, pairs=1000000, length=150, mut=2, ins=1, dele=1, max_bandwidth=150)


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


def cpu_baseline_pairs(pairs, max_bandwidth, budget_s):
    """Aligner CPU baseline on one host core, bounded sample of the same pairs: the reference's own CPU aligner
    (needleman_wunsch_cpu, compiled in place into oracle/_ref: kind "reference") when that library travelled to the
    box, else the C port of the banded Myers kernel (kind "port")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import numpy as np
    import oracle_aligner as A
    R = A.ref()
    n = cells = 0
    out = np.zeros(2 * max(len(q) + len(t) for q, t in pairs[:64]) + 64, np.int8)
    t0 = time.perf_counter()
    for q, t in pairs:
        if R is not None:
            R.ref_needleman_wunsch_cpu(t, len(t), q, len(q), out.ctypes.data, len(out))
            cells += len(q) * len(t)
        else:
            cells += A.align(q, t, max_bandwidth)["cells"]
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 1), "unit": "pairs/s", "cores": 1, "kind": "reference" if R is not None else "port",
            "gcups": round(cells / dt / 1e9, 4),
            "sample": "first %d pairs of the config, %.1f s, %s" % (
                n, dt, "needleman_wunsch_cpu of the reference (full |q| x |t| matrix + backtrace)" if R is not None
                else "C port of the banded Myers kernel (band cells)")}


def config3_golden_digest():
    """sha256 of the oracle's consensus strings of the 1024 config-3 windows, joined by newlines (tests/golden)."""
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "config_goldens.json")))["config3"]["consensus_sha256"]
    except (OSError, ValueError, KeyError):
        return None


def consensus_digest(consensus_strings):
    return hashlib.sha256("\n".join(consensus_strings).encode()).hexdigest()


_PMC_PROFILE = None


def pmc_profile():
    """The committed PMC session of this round (tools/gpu_session.sh steps pmc + issue + traffic -> tools/pmc_profile.py): HBM
    traffic, issued instructions, wave cycles, wait share and LDS bank conflicts per launch of every record's kernels. rocprofv3
    cannot run inside the timed region, so these come from their own passes -- and they only describe THESE kernels if the device
    sources have not changed since: the session stamps `kernel_source_sha256` (genomeworks_amd.build.kernel_source_digest()),
    which is recomputed here. A profile without the stamp or with another one is STALE: nothing of it is attached to a line
    (`"traffic": null, "stale": true`)."""
    global _PMC_PROFILE
    if _PMC_PROFILE is None:
        _PMC_PROFILE = {}
        try:
            from genomeworks_amd.build import kernel_source_digest
            now = kernel_source_digest()
            for name in ("r06_pmc_profile.json", "r05_pmc_profile.json", "r04_pmc_profile.json"):
                path = os.path.join(ROOT, "profiles", name)
                if os.path.exists(path):
                    prof = json.load(open(path))
                    if prof.get("kernel_source_sha256") == now:
                        _PMC_PROFILE = prof
                        _PMC_PROFILE["file"] = "profiles/" + name
                    else:
                        _PMC_PROFILE = {"stale": True, "file": "profiles/" + name, "stale_commit": prof.get("commit")}
                    break
        except (OSError, ValueError):
            _PMC_PROFILE = {}
    return _PMC_PROFILE


def pmc_stale():
    return bool(pmc_profile().get("stale"))


def sub_traffic(key):
    """HBM bytes of a record's kernels from the committed PMC session when it describes these kernels (pmc_profile()); else None."""
    e = pmc_profile().get(key)
    if isinstance(e, dict) and "hbm_bytes" in e:
        return e["hbm_bytes"]
    return None


def roofline_issue(key):
    """The roofline that explains these kernels: a window / pair is a chain of dependent steps on ONE wavefront, and a lone
    wavefront issues one instruction per ~4.1 cycles (profiles/r03_microbench_instruction_size.json). issued instructions per
    wavefront x 4.1 cycles is the floor of its run time; `frac` = that floor / the cycles a wavefront was resident, the rest is
    exposed latency (`wait_share` = SQ_WAIT_ANY / SQ_WAVE_CYCLES). From the committed PMC session (pmc_profile())."""
    p = pmc_profile()
    e = p.get(key)
    if not isinstance(e, dict) or "issue" not in e:
        return None
    out = {"bound": "instruction issue of a lone wavefront", "kernel": e["kernel"], "per": e.get("per"),
           "instructions": e["instructions"], "waves": e.get("waves"),
           "instructions_per_wave": e["issue"]["instructions_per_wave"], "cycles_per_wave": e["issue"]["cycles_per_wave"],
           "cycles_per_instruction": e["issue"]["cycles_per_instruction"],
           "lone_wave_cycles_per_instruction": e["issue"]["lone_wave_cycles_per_instruction"],
           "frac": e["issue"]["frac_of_lone_wave_issue_bound"], "wait_share": e.get("wait_share"),
           "lone_wave_cycles_per_instruction_source": e["issue"].get("lone_wave_cycles_per_instruction_source"),
           "source": "%s (commit %s, tag %s)" % (p.get("file"), p.get("commit"), p.get("tag"))}
    if "valu_busy" in e:
        out["valu_busy"] = e["valu_busy"].get("by_instruction_count")
        out["closest_limit"] = ("none saturated: a dependency chain on one wavefront per SIMD (VALU busy %.2f, HBM see roofline.frac, "
                                "issue port %.2f of its lone-wave bound)" % (out["valu_busy"], out["frac"]))
    if "lds" in e:
        out["lds_bank_conflict_share_of_lds_active"] = e["lds"]["bank_conflict_share_of_lds_active"]
        out["lds_bank_conflict_share_of_wave_cycles"] = e["lds"].get("bank_conflict_share_of_wave_cycles")
    return out


def reduce_scalars(dist, torch, values, op):
    if dist is None:
        return list(values)
    # RCCL reduces device tensors; the gloo group of the single-device test mode takes host tensors
    t = torch.tensor(list(values), device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
    dist.all_reduce(t, op=getattr(dist.ReduceOp, op))
    return [float(x) for x in t.tolist()]


def aligner_golden_verdict(key, runs, lo, hi):
    """The checker of a banded-aligner record: `runs` (CudaAlignerBatch.get_runs() of the pairs [lo, hi) of the config) against
    the committed oracle golden of BASELINE configs[1] (`key` "config2": a CIGAR fingerprint, the optimality flag and the edit
    distance of every pair) or configs[4] ("config5": flags and edit distances of every pair, digests of the fingerprints in
    blocks of 1024 pairs -- the blocks that lie inside [lo, hi) -- and one digest over all pairs when the range is the whole
    config). tests/golden/make_config_goldens.py wrote them; tests/test_gpu_config_goldens.py makes the same comparison.
    -> (ok, what was compared)"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_io as G  # the checker
    s = G.summary()[key]
    g = G.config2_pairs() if key == "config2" else G.config5_pairs()
    n = hi - lo
    ok = len(runs["status"]) == n and bool((np.asarray(runs["status"]) == 0).all())
    ok = ok and bool((np.asarray(runs["optimal"]) == g["optimal"][lo:hi]).all())
    ok = ok and bool((G.edit_distances(runs["offsets"], runs["ops"], runs["counts"]) == g["edit_distance"][lo:hi]).all())
    fp = G.run_fingerprints(runs["offsets"], runs["ops"], runs["counts"])
    if key == "config2":
        ok = ok and bool((fp == g["fingerprint"][lo:hi]).all())
        return ok, "CIGAR fingerprint, optimality flag and edit distance of each of the %d pairs" % n
    block = int(s["block"])
    first, last = (lo + block - 1) // block, hi // block  # whole blocks inside the range
    if hi == int(s["pairs"]) and hi % block:
        last += 1  # the config's short last block
    for b in range(first, last):
        part = fp[b * block - lo:min((b + 1) * block, hi) - lo]
        ok = ok and hashlib.sha256(np.ascontiguousarray(part).tobytes()).hexdigest()[:32] == str(g["block_sha"][b])
    if lo == 0 and hi == int(s["pairs"]):
        ok = ok and hashlib.sha256(fp.tobytes()).hexdigest() == s["fingerprint_sha256"]
    return ok, ("optimality flag and edit distance of each of the %d pairs, CIGAR fingerprints in %d blocks of %d pairs"
                % (n, max(0, last - first), block))


def bench_aligner(name, cfg, rank, world, local_rank, sync, dist, torch, reps, cpu_budget_s, cpu_all_cores=None):
    """One aligner config, index-split over the ranks. Timed regions: align_all() + sync_alignments() (the reference
    benchmark's, cudaaligner/benchmarks/main.cpp:96-143), align_all() + stream sync with the results left on the
    device (get_alignments_device), and the kernels alone (HIP events, inputs resident)."""
    from genomeworks_amd import cudaaligner, multi_gpu, synthetic
    pairs = synthetic.generate_pairs(cfg["seed"], cfg["pairs"], cfg["length"], cfg["mut"], cfg["ins"], cfg["dele"])
    # the all-core baseline was taken before the process's first device call (its workers are forked); else one core, in process
    cpu = cpu_all_cores if cpu_all_cores is not None else (
        cpu_baseline_pairs(pairs, cfg["max_bandwidth"], cpu_budget_s) if (rank == 0 and cpu_budget_s > 0) else None)
    lo, hi = multi_gpu.shard_range(len(pairs), rank, world)
    if cfg is CONFIG5 and world > 1:
        # the ranks' ranges on the grid of the golden's fingerprint blocks (1024 pairs; under 1 % of a rank's pairs at N = 8):
        # every pair's CIGAR then lies in a block that aligner_golden_verdict() compares
        block = 1024
        lo = 0 if rank == 0 else min(len(pairs), (lo + block - 1) // block * block)
        hi = len(pairs) if rank == world - 1 else min(len(pairs), (hi + block - 1) // block * block)
    mine = pairs[lo:hi]
    al = cudaaligner.CudaAlignerBatch(max_bandwidth=cfg["max_bandwidth"], max_device_memory_allocator_caching_size=32 << 30,
                                      device_id=local_rank)
    add = al._L.gw_aligner_add_alignment

    def fill():
        for q, t in mine:
            st = add(al._h, q, len(q), t, len(t), 0, 0)
            assert st == 0, st

    fill()
    al.align_all()
    cells = al.band_cells()
    k_ms = sum(al.relaunch_timed() for _ in range(reps)) / reps
    # device resident: align_all (H2D + kernels), results stay on the device
    t_dev = []
    for _ in range(reps):
        al.reset()
        fill()
        sync()
        t0 = time.perf_counter()
        al.align_all()
        n_dev, _total = al.device_sync()
        sync()
        t_dev.append(time.perf_counter() - t0)
        assert n_dev == len(mine)
    # host materialised: align_all + sync_alignments
    t_full = []
    for _ in range(reps):
        al.reset()
        fill()
        sync()
        t0 = time.perf_counter()
        al.align_all()
        n_host = al.sync()
        sync()
        t_full.append(time.perf_counter() - t0)
        assert n_host == len(mine)
    # outside the clock: this rank's pairs once more, every alignment against the committed oracle golden
    golden_flag, golden_what = -1.0, None
    try:
        al.reset()
        fill()
        al.align_all()
        ok, golden_what = aligner_golden_verdict("config2" if cfg is CONFIG2 else "config5", al.get_runs(), lo, hi)
        golden_flag = 1.0 if ok else 0.0
    except Exception as e:  # the record then says that the check did not run, and why
        golden_what = "golden check failed to run: %r" % (e,)
    al.reset()
    full, dev = min(t_full), min(t_dev)
    (golden_all,) = reduce_scalars(dist, torch, [golden_flag], "MIN")
    full, dev, k_max = reduce_scalars(dist, torch, [full, dev, k_ms], "MAX")
    (cells_all,) = reduce_scalars(dist, torch, [float(cells)], "SUM")
    if rank != 0:
        return None
    achieved = cells * BYTES_PER_MYERS_CELL / (k_ms * 1e-3) / 1e9
    out = {"workload": name, "pairs": len(pairs), "pairs_per_gpu": len(mine), "max_bandwidth": cfg["max_bandwidth"],
           "metric": "pairs/s, align_all() + sync_alignments()", "value": round(len(pairs) / full, 1), "unit": "pairs/s",
           "ms": round(full * 1e3, 3), "band_cells": int(cells_all), "band_gcups": round(cells_all / full / 1e9, 2),
           "device_resident": {"pairs_per_s": round(len(pairs) / dev, 1), "ms": round(dev * 1e3, 3),
                               "what": "align_all() + stream sync; results read through get_alignments_device()"},
           "kernel_only": {"pairs_per_s": round(len(pairs) / (k_max * 1e-3), 1), "ms": round(k_max, 3),
                           "band_gcups": round(cells_all / (k_max * 1e-3) / 1e9, 2)},
           "sync_over_kernel": round((full * 1e3 - k_max) / k_max, 2),
           "roofline": {"bound": "hbm", "kernel": "myers_banded_group_kernel<6, 4> (six lanes per pair, four wavefronts per block; + scan, compaction)" if cfg is CONFIG2 else "myers_banded_kernel (+ scan, compaction)",
                        "achieved": round(achieved, 2),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                        "traffic": sub_traffic("configs[1]" if cfg is CONFIG2 else "configs[4]"),
                        "algorithmic_bytes_per_cell": BYTES_PER_MYERS_CELL, "kernel_ms": round(k_ms, 3)},
           "roofline_issue": roofline_issue("configs[1]" if cfg is CONFIG2 else "configs[4]")}
    out["equals_oracle_golden"] = None if golden_all < 0 else bool(golden_all > 0.5)
    out["golden_compared"] = golden_what  # (rank 0's words; every rank compared its own pairs)
    if cpu is not None:
        out["cpu_baseline"] = cpu
    return out


def _load_long_read_plan():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_long_read_goldens", os.path.join(ROOT, "tests", "golden", "make_long_read_goldens.py"))
    lr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lr)
    return lr


def _cpu_long_worker(k):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_poa as O
    lr, windows, cfg_of, sample, cores, seconds = (_CPU_SHARED[x] for x in ("lr", "lr_windows", "lr_cfg_of", "lr_sample", "cores", "seconds"))
    cells = n = 0
    t0 = time.perf_counter()
    for w in sample[k::cores]:
        c = cfg_of[w]
        with O.Workspace(lr.oracle_cfg(c)) as ws:
            cells += ws.process(windows[w][:c["max_sequences_per_poa"]])["cells"]
        n += 1
        if time.perf_counter() - t0 > seconds:
            break
    return cells, n, time.perf_counter() - t0


def cpu_baseline_long_reads(n_windows, seconds=25.0):
    """CPU oracle on ALL host cores over a cost-stratified sample of the long-read set: the windows sorted by estimated
    cells, every m-th taken (so cheap, middling and heavy windows appear in the set's own proportions), dealt round-robin
    to one forked process per core; a process stops after the window during which its `seconds` ran out. `value` = sum of
    the per-process rates. Runs before the process's first device call."""
    import multiprocessing as mp
    from genomeworks_amd import multi_gpu
    lr = _load_long_read_plan()
    windows, cfgs, groups = lr.plan(max(n_windows, lr.CONFIG4["windows"]))
    cfg_of = {w: c for c, members in zip(cfgs, groups) for w in members}
    cost = [multi_gpu.poa_window_cost(w, 256) for w in windows[:n_windows]]
    order = sorted(range(n_windows), key=lambda w: cost[w])
    cores = max(1, os.cpu_count() or 1)
    # one window per process, every m-th of the cost-sorted set; a window that one core would need more than ~`seconds` for
    # (about 0.25 GCUPS per core; the cost estimate is ~6 x too low for wide adaptive bands) is left out so that the default
    # bench run stays within minutes -- the per-cell rate does not depend on the window's size
    limit = seconds * 0.25e9 / 6.0
    light = [w for w in order if cost[w] <= limit]
    step = max(1, -(-len(light) // cores))
    sample = light[step // 2::step] or order[:1]
    skipped = n_windows - len(light)
    _CPU_SHARED.update(lr=lr, lr_windows=windows, lr_cfg_of=cfg_of, lr_sample=sample, cores=min(cores, len(sample)), seconds=seconds)
    used = min(cores, len(sample))
    try:
        with mp.get_context("fork").Pool(used) as pool:
            res = pool.map(_cpu_long_worker, range(used), chunksize=1)
    except Exception as e:
        print("bench.py: long-read CPU baseline unavailable (%s)" % e, file=sys.stderr)
        return None
    res = [r for r in res if r[1] > 0]
    return {"value": round(sum(c / dt for c, _, dt in res) / 1e9, 4), "unit": "GCUPS", "cores": used, "kind": "port",
            "windows_per_s": round(sum(n / dt for _, n, dt in res), 3),
            "sample": "%d windows (every %d-th by estimated cost of the %d that one core finishes in about %.0f s; the %d heaviest left "
                      "out) over %d processes, %.0f-%.0f s each (gcc -O2 scalar oracle, the window's own BatchConfig)"
                      % (sum(n for _, n, _ in res), step, len(light), seconds, skipped, used, min(dt for _, _, dt in res), max(dt for _, _, dt in res))}


def bench_long_reads(n_windows, rank, world, local_rank, sync, dist, torch, cpu, ranks_per_device=1):
    """BASELINE configs[3]: the long-read MSA set, windows dealt to the ranks by estimated cost (no collective). The set is
    planned into size classes (cudapoa::plan_size_classes: geometric in the longest read, one BatchConfig per class) and
    all classes run at once, one host thread + stream + Batch each (process_windows_size_classes) -- the multi-batch
    pattern of the reference (cudapoa/benchmarks/multi_batch.hpp) with its size binning (cudapoa/src/utils.cu:66-146).
    Timed region: from the moment every class has filled its batch (add_poa_group excluded, as in the reference
    benchmarks) to the last class's end of generate_poa() + get_msa()."""
    from genomeworks_amd import cudapoa, multi_gpu
    lr = _load_long_read_plan()
    # the plan (and with it every window's BatchConfig) is that of the whole 598-window set, also when only the first
    # n_windows are run or when the windows are dealt to several ranks: the goldens are keyed to it
    windows, cfgs, groups = lr.plan(max(n_windows, lr.CONFIG4["windows"]))
    plan = lr.size_plan(windows)
    cost = [multi_gpu.poa_window_cost(w, 256) if k < n_windows else 0 for k, w in enumerate(windows)]
    mine = [w for w in multi_gpu.balanced_partition(cost, world)[rank] if w < n_windows]
    if world > 1 or n_windows < len(windows):
        plan.keep(mine)
    golden = {}
    try:
        with open(os.path.join(ROOT, "tests", "golden", "config4_long_reads.json")) as f:
            golden = {d["w"]: d for d in json.load(f)["windows_detail"]}
    except (OSError, ValueError):
        pass
    sync()
    out = cudapoa.process_windows_size_classes(windows, plan, device=local_rank,
                                               memory_budget=lr.CONFIG4["memory_budget_bytes"] // ranks_per_device,
                                               output_type="msa", digest=lr.msa_digest)
    sync()
    my_cells = sum(golden[w]["cells"] for w in mine if w in golden)  # the kernels' counters equal the oracle's (asserted by the GPU tests)
    failed_cells = sum(golden[w]["cells"] for w in mine if w in golden and out["status"][w] != 0)
    n_ok = sum(1 for w in mine if out["status"][w] == 0)
    checked = sum(1 for w in mine if w in golden and out["status"][w] == golden[w]["status"] and
                  (out["status"][w] != 0 or out["msa"][w] == golden[w]["msa_sha"]))
    mismatched = sum(1 for w in mine if w in golden) - checked
    seconds, total_s, fill_s = reduce_scalars(dist, torch, [out["compute_seconds"], out["seconds"], out["seconds_after_creation"]], "MAX")
    cells, n_done, n_ok, checked, mismatched, failed_cells = reduce_scalars(
        dist, torch, [float(my_cells), float(len(mine)), float(n_ok), float(checked), float(mismatched), float(failed_cells)], "SUM")
    if rank != 0:
        return None
    achieved = my_cells * BYTES_PER_CELL_LONG / out["compute_seconds"] / 1e9
    rec = {"workload": "BASELINE configs[3]: cudapoa long-read MSA, %d windows (8-32 reads, 2-30 kbp, 8-12 %% indel-heavy "
                       "divergence, seeds 2000+w), adaptive band 256, adaptive_storage_factor 4, %d size classes resident and "
                       "running at once (%.0f GB of slabs on rank 0)" % (n_windows, len(cfgs), plan.total_bytes / 1e9),
           "metric": "GCUPS, generate_poa() + get_msa() of all size classes (concurrent batches)", "value": round(cells / seconds / 1e9, 3),
           "unit": "GCUPS", "windows": int(n_done), "windows_ok": int(n_ok), "windows_per_s": round(n_done / seconds, 2),
           "ms": round(seconds * 1e3, 1),
           # the reference's multi-batch benchmark (cudapoa/benchmarks/multi_batch.hpp:72-177) times filling + generate_poa() +
           # get_msa() of batches that exist already: that clock, next to the kernels-and-results clock of `value`
           "fill_inclusive": {"ms": round(fill_s * 1e3, 1), "gcups": round(cells / max(fill_s, 1e-9) / 1e9, 3),
                              "what": "add_poa_group() of every window + generate_poa() + get_msa(), batches created before the clock starts"},
           "ms_with_batch_creation_and_filling": round(total_s * 1e3, 1), "cells": int(cells),
           # windows that end with an error status (exceeded_adaptive_banded_matrix_size, in the oracle too) stop at the read that
           # failed: the cells they computed up to there are in `cells`; this is how many
           "cells_of_windows_with_error_status": int(failed_cells),
           "value_without_those_cells": round((cells - failed_cells) / seconds / 1e9, 3),
           "launches_rank0": out["launches"], "dtype": "int32",
           "windows_equal_to_oracle_golden": int(checked), "windows_differing_from_golden": int(mismatched),
           "equals_oracle_golden": bool(int(mismatched) == 0 and int(checked) == n_windows),
           "size_classes": [{"max_sequence_size": c["max_sequence_size"], "windows": len(g)} for c, g in zip(cfgs, plan.groups)],
           "roofline": {"bound": "hbm", "kernel": "poa_window_kernel<int32,int32,adaptive_band,HBM tables, 8 waves per window> (4 launches, admitted by residency)",
                        "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": sub_traffic("configs[3]"),
                        "algorithmic_bytes_per_cell": BYTES_PER_CELL_LONG,
                        "kernel_ms": round(out["compute_seconds"] * 1e3, 1),
                        "note": "the launches of the classes overlap: the duration is the concurrent region's (host clock), "
                                "of which the kernels are all but the D2H and unpacking of the MSA rows"},
           "roofline_issue": roofline_issue("configs[3]")}
    if cpu is not None:
        rec["cpu_baseline"] = cpu
    return rec


def cpu_baseline_pairs_all_cores(pairs, max_bandwidth, seconds, max_cores=None):
    """The aligner CPU baseline on ALL host cores: one forked process per core, each running the reference's own
    needleman_wunsch_cpu (oracle/_ref, kind "reference"; the C port of the banded kernel if that library is absent) over
    its share of the pairs for `seconds`. Runs before the process's first device call."""
    import multiprocessing as mp
    cores = max(1, os.cpu_count() or 1)
    if max_cores is not None:  # (the reference's CPU aligner keeps a full int matrix per pair: 268 MB at 8 192 bases)
        cores = max(1, min(cores, max_cores))
    _CPU_SHARED.update(pairs=pairs, cores=cores, seconds=seconds, max_bandwidth=max_bandwidth)
    try:
        with mp.get_context("fork").Pool(cores) as pool:
            res = pool.map(_cpu_pairs_worker, range(cores), chunksize=1)
    except Exception as e:
        print("bench.py: all-core aligner CPU baseline unavailable (%s)" % e, file=sys.stderr)
        return None
    kind = res[0][3]
    return {"value": round(sum(n / dt for n, _, dt, _ in res), 1), "unit": "pairs/s", "cores": cores, "kind": kind,
            "gcups": round(sum(c / dt for _, c, dt, _ in res) / 1e9, 3),
            "sample": "%d pairs over %d processes, %.1f s each, %s" % (
                sum(n for n, _, _, _ in res), cores, seconds,
                "needleman_wunsch_cpu of the reference (full |q| x |t| matrix + backtrace)" if kind == "reference"
                else "C port of the banded Myers kernel (band cells)")}


def _cpu_pairs_worker(k):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_aligner as A
    pairs, cores, seconds, max_bandwidth = (_CPU_SHARED[x] for x in ("pairs", "cores", "seconds", "max_bandwidth"))
    share = pairs[k::cores] or pairs[:1]
    R = A.ref()
    out = np.zeros(2 * max(len(q) + len(t) for q, t in share[:64]) + 4096, np.int8)
    n = cells = 0
    t0 = time.perf_counter()
    while True:
        for q, t in share:
            if R is not None and 2 * (len(q) + len(t)) + 64 <= len(out):
                R.ref_needleman_wunsch_cpu(t, len(t), q, len(q), out.ctypes.data, len(out))
                cells += len(q) * len(t)
            else:
                cells += A.align(q, t, max_bandwidth)["cells"]
            n += 1
            if time.perf_counter() - t0 > seconds:
                return n, cells, time.perf_counter() - t0, "reference" if R is not None else "port"


def bench_default_aligner(local_rank, sync, cpu_all_cores=None):
    """The default aligner -- create_aligner(max_query, max_target, n): Hirschberg + Myers, the one the Python bindings reach
    -- on the reference's benchmark shapes (cudaaligner/benchmarks/main.cpp:39-67 BM_SingleAlignment: one pair of 100 ..
    100 000 bases; :69-143 BM_SingleBatchAlignment: 1024 pairs x 2048 bases) and on 2 000 pairs x 1 kbp. Timed region as
    there: align_all() + sync_alignments() with the pairs queued. Rank 0 only."""
    from genomeworks_amd import cudaaligner, synthetic
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_io as G  # the checker: committed oracle goldens of every shape (tests/golden/make_default_aligner_goldens.py)
    gold = G.default_aligner_goldens()
    shapes = G.aligner_gen.SHAPES
    rows = []
    cpu = cpu_all_cores
    for n, size in shapes:
        pairs = G.aligner_gen.shape_pairs(n, size)
        al = cudaaligner.CudaAlignerBatch(size, size, n, max_device_memory_allocator_caching_size=32 << 30, device_id=local_rank)
        best = None
        for _ in range(3):
            for q, t in pairs:
                assert al.add_alignment(q, t) == 0
            sync()
            t0 = time.perf_counter()
            al.align_all()
            assert al.sync() == n
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            # every pair's state sequence of the LAST timed run against the oracle's (outside the clock)
            res = al.get_alignments()
            k_ms = min(al.relaunch_timed() for _ in range(3))   # the kernels alone, HIP events on the aligner's stream
            al.reset()
        del al
        g = gold["%dx%d" % (n, size)]
        sha = G.aligner_gen.digest(G.aligner_gen.pair_record(r.status, r.alignment) for r in res)
        cells = sum(len(q) * len(t) for q, t in pairs)
        rows.append({"pairs": n, "length": size, "ms": round(best * 1e3, 3), "pairs_per_s": round(n / best, 1),
                     "full_matrix_gcups": round(cells / best / 1e9, 2), "kernel_ms": round(k_ms, 3),
                     "states_sha256": sha, "equals_oracle_golden": bool(sha == g["states_sha256"])})
        del res
    head = rows[-1]
    achieved = 2000 * 1000 * 1000 * BYTES_PER_MYERS_CELL / (head["kernel_ms"] * 1e-3) / 1e9
    out = {"workload": "default aligner (Hirschberg + Myers bit vectors, one wavefront per pair, the tree grown level by level for queries of up to 2 048 bases): reference benchmark shapes",
           "metric": "pairs/s, align_all() + sync_alignments(), 2 000 pairs x 1 kbp (about 10 % divergence)",
           "value": head["pairs_per_s"], "unit": "pairs/s", "ms": head["ms"], "shapes": rows,
           "all_equal_oracle_golden": all(r["equals_oracle_golden"] for r in rows),
           "kernel_only": {"pairs_per_s": round(2000 / (head["kernel_ms"] * 1e-3), 1), "ms": head["kernel_ms"]},
           "roofline": {"bound": "hbm", "kernel": "hirschberg_levels_kernel (+ hirschberg_wave_kernel for what it leaves; the 10 kbp / 100 kbp single pairs: hb_span_rows_mw / split / parts / join kernels)", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": sub_traffic("default_aligner"), "algorithmic_bytes_per_cell": BYTES_PER_MYERS_CELL,
                        "note": "|q| x |t| cells of the full matrix at the bit-vector cost of 12 B per 32-cell word column; the divide "
                                "and conquer computes every cell about twice and keeps its state in registers and LDS, so HBM "
                                "carries little: the kernel is bound by the dependent column steps of a wavefront",
                        "kernel_ms": head["kernel_ms"]},
           "roofline_issue": roofline_issue("default_aligner")}
    if cpu is not None:
        out["cpu_baseline"] = cpu
    return out


def bench_aligner_matrix(local_rank, sync, cpu_by_size=None, corners=False):
    """Six cells of the reference's aligner benchmark matrix (cudaaligner/benchmarks/main.cpp:69-143, registered :150-168:
    AlignerGlobalUkkonen / AlignerGlobalMyers / AlignerGlobalMyersBanded / AlignerGlobalHirschbergMyers x alignments per batch
    x genome size): all four classes at 1024 pairs x 2048 bases, Ukkonen and Hirschberg + Myers at 256 x 8192. Timed region as
    there: align_all() + sync_alignments() with the pairs queued; the kernels alone by HIP events. Every cell's state
    sequences are compared with the committed oracle golden of that class (tests/golden/make_aligner_matrix_goldens.py)."""
    import ctypes as C
    import numpy as np
    from genomeworks_amd import _native, cudaaligner
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_io as G  # the checker
    gold = G.aligner_matrix_goldens()
    rows = []
    # corners=True (--sub-configs aligner_grid): the corners of the reference's grid instead -- 1024 x 512, 32 x 32768 and 32 x 65536 for
    # every class (cudaaligner/benchmarks/main.cpp:150-168); cells without a committed golden are skipped
    for algorithm, n, size in (G.matrix_gen.CORNER_CELLS if corners else G.matrix_gen.CELLS):
        if G.matrix_gen.cell_key(algorithm, n, size) not in gold:
            continue
        pairs = G.aligner_gen.shape_pairs(n, size)
        if algorithm == "myers_banded":
            al = cudaaligner.CudaAlignerBatch(max_bandwidth=G.matrix_gen.BANDED_MAX_BANDWIDTH, max_device_memory_allocator_caching_size=96 << 30,
                                              device_id=local_rank)
        else:
            # (full-matrix Myers at 32 x 65 536: 103 GB of workspace, a region is 64 slots wide whatever the number of pairs)
            al = cudaaligner.CudaAlignerBatch(size, size, n, algorithm=algorithm, max_device_memory_allocator_caching_size=(160 if size > 32768 else 96) << 30,
                                              device_id=local_rank)
        best = None
        for _ in range(3):
            for q, t in pairs:
                assert al.add_alignment(q, t) == 0
            sync()
            t0 = time.perf_counter()
            al.align_all()
            assert al.sync() == n
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            al.reset()
        # outside the clock: the kernels alone (HIP events on the aligner's stream, inputs resident), the band cells, and every
        # pair's state sequence for the golden verdict
        for q, t in pairs:
            assert al.add_alignment(q, t) == 0
        al.align_all()
        k_ms = min(al.relaunch_timed() for _ in range(3))
        band_cells = al.band_cells() if algorithm == "myers_banded" else None
        res = al.get_alignments()
        al.reset()
        del al
        g = gold[G.matrix_gen.cell_key(algorithm, n, size)]
        sha = G.aligner_gen.digest(G.aligner_gen.pair_record(r.status, r.alignment) for r in res)
        del res
        full_cells = sum(len(q) * len(t) for q, t in pairs)
        if algorithm == "ukkonen":
            # the band Ukkonen's class stores (int16 per slot, every slot written once): what gwhip_ukkonen_workspace_bytes sizes
            starts = np.cumsum([0] + [x for q, t in pairs for x in (len(q), len(t))]).astype(np.int64)
            Gw = _native.gwhip()
            Gw.gwhip_ukkonen_workspace_bytes.restype = C.c_size_t
            Gw.gwhip_ukkonen_workspace_bytes.argtypes = [C.c_int32, C.c_void_p, C.c_int32]
            alg_bytes, what = float(Gw.gwhip_ukkonen_workspace_bytes(n, starts.ctypes.data, 100)), "band slots x 2 B (each written once)"
        elif algorithm == "myers_banded":
            alg_bytes, what = band_cells * BYTES_PER_MYERS_CELL, "band cells x 0.375 B (pv, mv, score per 32-cell word column)"
        else:
            alg_bytes, what = full_cells * BYTES_PER_MYERS_CELL, "|q| x |t| cells x 0.375 B (pv, mv, score per 32-cell word column)"
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        row = {"algorithm": algorithm, "pairs": n, "length": size, "ms": round(best * 1e3, 3), "pairs_per_s": round(n / best, 1),
               "full_matrix_gcups": round(full_cells / best / 1e9, 2), "kernel_ms": round(k_ms, 3),
               "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(achieved / HBM_PEAK_GBS, 5), "algorithmic_bytes": int(alg_bytes), "algorithmic_bytes_are": what,
                            "kernel_ms": round(k_ms, 3), "traffic": sub_traffic("aligner_matrix/" + G.matrix_gen.cell_key(algorithm, n, size))},
               "roofline_issue": roofline_issue("aligner_matrix/" + G.matrix_gen.cell_key(algorithm, n, size)),
               "states_sha256": sha, "equals_oracle_golden": bool(sha == g["states_sha256"])}
        if cpu_by_size and size in cpu_by_size and cpu_by_size[size] is not None:
            row["cpu_baseline"] = cpu_by_size[size]
        rows.append(row)
    return {"workload": "cells of the reference's aligner benchmark matrix (BM_SingleBatchAlignment: every aligner class x alignments per "
                        "batch x genome size; genome pairs at about 10 % divergence)",
            "metric": "pairs/s, align_all() + sync_alignments()", "all_equal_oracle_golden": all(r["equals_oracle_golden"] for r in rows),
            "rows": rows}


def bench_band_modes(windows, local_rank, sync):
    """Every banded mode x band width of the reference's parameter space (multiples of 128, cudapoa/src/batch.cu:41; the
    traceback-buffer modes of cudapoa_nw_tb_banded.cuh:264-643) on the 1024 config-3 windows: generate_poa() +
    get_consensus(), steady state, plus the graph-build kernel's own duration. Every cell runs a packed 16-bit forward pass
    and the sheared-tile walk (round 4: poa_forward_moves.h for bands 128 / 256, poa_forward_moves_wide.h for 384 / 512,
    poa_forward_moves_tb.h for the traceback-buffer modes) and is compared with its committed oracle golden."""
    from genomeworks_amd import cudapoa
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
    import golden_io as G  # the checker: committed oracle goldens of every cell (tests/golden/make_band_mode_goldens.py)
    gsum, gold = G.band_mode_summary(), G.band_mode_goldens()
    assert gsum["windows"] == len(windows) == 1024 and gsum["first_seed"] == 1000
    rows = []
    for mode in ("static_band", "adaptive_band", "static_band_traceback", "adaptive_band_traceback"):
        for bw in (128, 256, 384, 512):
            b = cudapoa.CudaPoaBatch(32, 1024, 24 << 30, output_type="consensus", band_mode=mode, alignment_band_width=bw,
                                     max_nodes_per_graph=3072, device_id=local_rank)
            for w in windows:
                st, _ = b.add_poa_group(w)
                assert st == 0, st
            b.generate_poa()
            n_ok = b.get_consensus_native()
            cells = b.total_cells()
            sync()
            t0 = time.perf_counter()
            for _ in range(2):
                b.generate_poa()
                b.get_consensus_native()
            sync()
            dt = (time.perf_counter() - t0) / 2
            k_ms, _o = b.relaunch_timed()
            cons, cov, status = b.get_consensus()  # of the last (timed) launch
            del b
            cell = gsum["cells"]["%s/%d" % (mode, bw)]
            fp = G.band_mode_fingerprints(cons, cov, status)
            mi, wi = gsum["modes"].index(mode), gsum["widths"].index(bw)
            rows.append({"band_mode": mode, "band_width": bw, "ms": round(dt * 1e3, 2), "kernel_ms": round(k_ms, 2),
                         "gcups": round(cells / dt / 1e9, 1), "cells": cells, "windows_ok": sum(1 for x in status if int(x) == 0),
                         "windows_equal_oracle_golden": int((fp == gold["fingerprint"][mi, wi]).sum()),
                         "equals_oracle_golden": bool(G.band_gen.cell_digest(fp) == cell["fingerprint_sha256"] and cells == cell["cells"])})
    ref = next(r for r in rows if r["band_mode"] == "static_band" and r["band_width"] == 256)
    for r in rows:
        r["gcups_vs_static_256"] = round(r["gcups"] / ref["gcups"], 3)
    return {"workload": "the 1024 config-3 windows through every banded mode and band width; every row's 1024 consensus / "
                        "coverage / status triples and its cell count are compared with the committed oracle golden of that cell",
            "all_equal_oracle_golden": all(r["equals_oracle_golden"] for r in rows), "rows": rows}


def bench_reference_shapes(windows, local_rank, sync, steps):
    """The reference's own cudapoa benchmark shapes on the config-3 inputs (no published numbers exist for them):
    BM_SingleBatchTest -- one batch, BatchConfig(1024, 200) = full band, generate_poa() + get_consensus()
    (cudapoa/benchmarks/single_batch.hpp:52-54,86-93); BM_MultiBatchTest -- 1, 2, 4, 8 concurrent batches on host
    threads sharing one device (multi_batch.hpp:41-61,72-177), here through process_windows_multi_device."""
    from genomeworks_amd import cudapoa
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_io as G  # the checker: the full-band golden (tests/golden/make_full_band_goldens.py) and the config-3 golden
    fgold, fsum = G.full_band_goldens(), G.full_band_summary()
    b = cudapoa.CudaPoaBatch(200, 1024, 16 << 30, output_type="consensus", band_mode="full_band", device_id=local_rank,
                             max_nodes_per_graph=3072, matrix_sequence_dimension=1024)
    for w in windows:
        st, _ = b.add_poa_group(w)
        assert st == 0, st
    b.generate_poa()
    b.get_consensus_native()
    cells = b.total_cells()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.generate_poa()
        b.get_consensus_native()
    sync()
    dt = (time.perf_counter() - t0) / steps
    k_ms, _o = b.relaunch_timed()
    cons, cov, status = b.get_consensus()  # of the last launch
    del b
    fp = G.band_mode_fingerprints(cons, cov, status)
    single = {"shape": "BM_SingleBatchTest: %d windows, BatchConfig(1024, 200) full band, consensus" % len(windows),
              "ms": round(dt * 1e3, 2), "gcups": round(cells / dt / 1e9, 2), "windows_per_s": round(len(windows) / dt, 1),
              "cells": cells, "kernel_ms": round(k_ms, 2),
              "roofline": {"bound": "hbm", "kernel": "poa_window_kernel<int16,int16,full_band>", "achieved": round(cells * BYTES_PER_CELL / (k_ms * 1e-3) / 1e9, 2),
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(cells * BYTES_PER_CELL / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                           "algorithmic_bytes_per_cell": BYTES_PER_CELL, "kernel_ms": round(k_ms, 3), "traffic": sub_traffic("full_band")},
              "roofline_issue": roofline_issue("full_band"),
              "windows_equal_oracle_golden": int((fp == fgold["fingerprint"]).sum()) if len(windows) == fsum["windows"] else None,
              "equals_oracle_golden": bool(len(windows) == fsum["windows"] and G.band_gen.cell_digest(fp) == fsum["fingerprint_sha256"]
                                           and cells == fsum["cells"])}
    # BM_SingleBatchTest's own sweep (cudapoa/benchmarks/main.cpp:68-71: RangeMultiplier(4), Range(1, 1024)): the first n windows
    # as one batch, in the benchmark's full band and in the metric configuration (static band 256). One window is one chain of
    # 31 dependent alignments on ONE wavefront, so a batch that leaves SIMDs empty takes as long as a full one: the sweep is the
    # measured form of "a 1024-window batch does not strong-scale" (DESIGN.md 5).
    sweep = []
    for n in (1, 4, 16, 64, 256, 1024):
        if n > len(windows):
            break
        row = {"windows": n}
        for label, mk in (("full_band", lambda: cudapoa.CudaPoaBatch(200, 1024, 16 << 30, output_type="consensus", band_mode="full_band",
                                                                     device_id=local_rank, max_nodes_per_graph=3072, matrix_sequence_dimension=1024)),
                          ("static_band_256", lambda: cudapoa.CudaPoaBatch(32, 1024, 8 << 30, output_type="consensus", band_mode="static_band",
                                                                           alignment_band_width=256, device_id=local_rank, max_nodes_per_graph=3072))):
            sb = mk()
            for w in windows[:n]:
                st, _ = sb.add_poa_group(w)
                assert st == 0, st
            sb.generate_poa()
            sb.get_consensus_native()
            c_n = sb.total_cells()
            sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                sb.generate_poa()
                sb.get_consensus_native()
            sync()
            dt_n = (time.perf_counter() - t0) / steps
            cons, cov, status = sb.get_consensus()
            del sb
            if label == "full_band":
                ok = bool(len(windows) == fsum["windows"] and (G.band_mode_fingerprints(cons, cov, status) == fgold["fingerprint"][:n]).all())
            else:
                rows3, _ = G.config3_windows()
                ok = all(c == rows3[i]["consensus"] and list(v) == list(rows3[i]["coverage"]) and int(stt) == rows3[i]["status"]
                         for i, (c, v, stt) in enumerate(zip(cons, cov, status)))
            row[label] = {"ms": round(dt_n * 1e3, 3), "gcups": round(c_n / dt_n / 1e9, 2), "windows_per_s": round(n / dt_n, 1),
                          "equals_oracle_golden": ok}
        sweep.append(row)
    multi = []
    twice = windows + windows
    gold_fp = fgold["fingerprint"]
    # the reference's own size: 5 500 windows over 1 .. 16 batches (cudapoa/benchmarks/main.cpp:48-66), here the 1024 config-3
    # windows cyclically; every block of results is compared with the same golden fingerprints
    big = [windows[i % len(windows)] for i in range(5500)]
    multi_5500 = []
    for nb in (1, 2, 4, 8, 16):
        sync()
        out = cudapoa.process_windows_multi_device(big, 200, 1024, devices=(local_rank,), batches_per_device=nb,
                                                   memory_per_device=int(96e9), band_mode="full_band",
                                                   max_nodes_per_graph=3072, matrix_sequence_dimension=1024)
        fp5 = G.band_mode_fingerprints(out["consensus"], out["coverage"], out["status"])
        ok = bool(len(windows) == fsum["windows"] and all(s == 0 for s in out["status"])
                  and all((fp5[k:k + len(windows)] == gold_fp[:len(fp5[k:k + len(windows)])]).all() for k in range(0, len(big), len(windows))))
        multi_5500.append({"batches": nb, "ms": round(out["seconds_after_creation"] * 1e3, 1),
                           "windows_per_s": round(len(big) / out["seconds_after_creation"], 1), "launches": out["launches"],
                           "gcups": round(cells * len(big) / len(windows) / out["seconds_after_creation"] / 1e9, 1),
                           "ms_with_batch_creation": round(out["seconds"] * 1e3, 1), "equals_oracle_golden": ok})
    for nb in (1, 2, 4, 8):
        sync()
        # BatchConfig(1024, 200) = full band and one share of the device's memory split evenly over the batches, as the
        # reference's multi-batch benchmark does (multi_batch.hpp:49-57: 0.9 x free / batches; here 32 GB in all, 11 MB per
        # window: every batch holds its share of the 2048 windows in one fill)
        out = cudapoa.process_windows_multi_device(twice, 200, 1024, devices=(local_rank,), batches_per_device=nb,
                                                   memory_per_device=int(32e9), band_mode="full_band",
                                                   max_nodes_per_graph=3072, matrix_sequence_dimension=1024)
        dt, dt_all = out["seconds_after_creation"], out["seconds"]  # the reference times process_batches(): the batches exist
        assert all(s == 0 for s in out["status"])
        fp2 = G.band_mode_fingerprints(out["consensus"], out["coverage"], out["status"])
        ok = bool(len(windows) == fsum["windows"] and (fp2[:len(windows)] == gold_fp).all() and (fp2[len(windows):] == gold_fp).all())
        multi.append({"batches": nb, "ms": round(dt * 1e3, 1), "windows_per_s": round(len(twice) / dt, 1), "launches": out["launches"],
                      "gcups": round(2 * cells / dt / 1e9, 1), "ms_with_batch_creation": round(dt_all * 1e3, 1), "equals_oracle_golden": ok})
    return {"single_batch_full_band": single,
            "single_batch_sweep": {"shape": "BM_SingleBatchTest sweep: the first n of the 1024 config-3 windows as one batch, generate_poa() + "
                                            "get_consensus(); full band = BatchConfig(1024, 200) as in the reference, static_band_256 = the metric "
                                            "configuration (cudapoa/benchmarks/main.cpp:68-71)", "rows": sweep},
            "multi_batch_5500": {"shape": "BM_MultiBatchTest: 5500 windows (the 1024 config-3 windows cyclically), BatchConfig(1024, 200) full band, "
                                          "1 .. 16 batches on host threads sharing the device and 96 GB of it (cudapoa/benchmarks/main.cpp:48-66, "
                                          "multi_batch.hpp:41-61,165-176); timed like process_batches()", "runs": multi_5500},
            "multi_batch": {"shape": "BM_MultiBatchTest pattern: %d windows (the 1024 config-3 windows twice), BatchConfig(1024, 200) full "
                                     "band as in the reference, N batches on host threads sharing the device and 32 GB of it split evenly "
                                     "(multi_batch.hpp:49-57); timed like process_batches() there: filling (under the window mutex), kernels, "
                                     "result unpacking of batches that exist (`ms_with_batch_creation` includes their construction)" % len(twice),
                            "runs": multi}}


FINAL_LINE_LIMIT = 4096   # bytes: the driver keeps a bounded tail of stdout and parses the LAST line


def emit(headline, strong, strong8, sub, record_dir):
    """Output protocol. Every sub-record goes out FIRST, one `{"sub_record": name, ...}` JSON line each, and the whole run
    (headline + sub-records) is also written to <record_dir>/bench_full_record.json. The LAST line of stdout is the headline
    alone -- the reference benchmark prints one number per benchmark (cudapoa/benchmarks/single_batch.hpp:86-93) -- and is
    kept under FINAL_LINE_LIMIT bytes: metric, value, unit, n_gpus, steps, warmup, ms_per_step, dtype, config, roofline,
    cpu_baseline, the golden verdict and a pointer to the file. Returns the final line."""
    extras = {}
    if strong is not None:
        extras["strong_scaling"] = strong
    if strong8 is not None:
        extras["strong_scaling_8x"] = strong8
    extras.update(sub or {})
    for name, rec in extras.items():
        print(json.dumps({"sub_record": name, "record": rec}), flush=True)
    full = dict(headline)
    if extras:
        full["sub_records"] = extras
    pointer = None
    try:
        os.makedirs(record_dir, exist_ok=True)
        pointer = os.path.join(record_dir, "bench_full_record.json")
        with open(pointer, "w") as f:
            json.dump(full, f, indent=1)
        pointer = os.path.relpath(pointer, ROOT)
    except OSError as e:
        print("bench.py: could not write the full record (%s)" % e, file=sys.stderr)
        pointer = None
    final = dict(headline)
    if isinstance(final.get("roofline_issue"), dict):  # (the whole object is in the file; the line keeps the ratios)
        keep = ("bound", "instructions_per_wave", "cycles_per_wave", "cycles_per_instruction", "lone_wave_cycles_per_instruction", "frac",
                "wait_share", "valu_busy", "closest_limit", "source")
        final["roofline_issue"] = {k: final["roofline_issue"][k] for k in keep if k in final["roofline_issue"]}
    # what the sub-records said, in one small object: name -> golden verdict (every row of the record ANDed) + its value
    final["sub_records"] = {"file": pointer, "lines": "one {\"sub_record\": ...} stdout line each, before this line",
                            "summary": {k: sub_summary(v) for k, v in extras.items()}}
    line = json.dumps(final, separators=(",", ":"))
    # shed the optional members (never metric / roofline / cpu_baseline) until the line fits
    for victim in ("roofline_issue", "extension_get_consensus_in_place", "kernel_only", "timed_region", "scaling_note",
                   "consensus_sha256", "cold_first_pass_ms"):
        if len(line.encode()) < FINAL_LINE_LIMIT:
            break
        final.pop(victim, None)
        line = json.dumps(final, separators=(",", ":"))
    if len(line.encode()) >= FINAL_LINE_LIMIT:
        final["sub_records"] = {"file": pointer}
        line = json.dumps(final, separators=(",", ":"))
    assert len(line.encode()) < FINAL_LINE_LIMIT, len(line)
    sys.stdout.flush()
    print(line, flush=True)
    return line


def golden_verdicts(rec):
    """Every `equals_oracle_golden` found anywhere inside a record."""
    found = []
    if isinstance(rec, dict):
        for k, v in rec.items():
            if k == "equals_oracle_golden":
                found.append(v)
            else:
                found.extend(golden_verdicts(v))
    elif isinstance(rec, (list, tuple)):
        for v in rec:
            found.extend(golden_verdicts(v))
    return found


def sub_summary(rec):
    v = golden_verdicts(rec)
    out = {"golden": [sum(1 for x in v if x is True), len(v)]}   # [rows equal to the golden, rows checked]
    if isinstance(rec, dict):
        for k in ("value", "gcups", "ms", "ms_per_step"):
            if isinstance(rec.get(k), (int, float)):
                out[k] = rec[k]
        if isinstance(rec.get("unit"), str):
            out["unit"] = rec["unit"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--windows", type=int, default=WINDOWS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sub-configs", default="aligner,default_aligner,aligner_matrix,long_reads,reference_shapes,band_modes",
                    help="comma list of the sub-records to measure next to the metric config: aligner, default_aligner, "
                         "aligner_matrix, long_reads, reference_shapes, band_modes, none")
    ap.add_argument("--long-read-windows", type=int, default=598)
    ap.add_argument("--record-dir", default=os.path.join(ROOT, "gpurun_out"),
                    help="where bench_full_record.json (headline + every sub-record) is written; stdout's last line is the headline")
    args = ap.parse_args()
    subs = set(x for x in args.sub_configs.split(",") if x and x != "none")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # test aid (tools/r02_gpu_run_multirank.sh): several ranks on ONE device over gloo, to exercise the multi-rank code
    # paths on a single-GPU box; never set by the driver
    ranks_per_device = max(1, int(os.environ.get("GW_BENCH_RANKS_PER_DEVICE", "1")))
    if ranks_per_device > 1:
        local_rank = local_rank // ranks_per_device

    from genomeworks_amd import synthetic
    first_seed = 1000 # every rank the same batch: the one tests/golden/config3_windows.txt.gz holds the oracle's output for
    windows = [[r.decode() for r in synthetic.generate_window(first_seed + w)] for w in range(args.windows)]
    # the CPU baseline forks one worker per core: before this process makes its first device call
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    cpu = cpu_baseline(windows) if want_cpu else None
    cpu_pairs = {}
    cpu_long = None
    if want_cpu:  # the aligner baselines on all host cores too (reference needleman_wunsch_cpu from oracle/_ref when it travelled)
        if "aligner" in subs:
            cpu_pairs["configs[1]"] = cpu_baseline_pairs_all_cores(
                synthetic.generate_pairs(CONFIG2["seed"], CONFIG2["pairs"], CONFIG2["length"], CONFIG2["mut"], CONFIG2["ins"], CONFIG2["dele"]),
                CONFIG2["max_bandwidth"], 3.0)
            cpu_pairs["configs[4]"] = cpu_baseline_pairs_all_cores(
                synthetic.generate_pairs(CONFIG5["seed"], 65536, CONFIG5["length"], CONFIG5["mut"], CONFIG5["ins"], CONFIG5["dele"]),
                CONFIG5["max_bandwidth"], 3.0)
        if "long_reads" in subs:
            cpu_long = cpu_baseline_long_reads(args.long_read_windows, 25.0)
        if "default_aligner" in subs:
            p1k = synthetic.generate_pairs(1, 2000, 1000, 33, 33, 33)
            cpu_pairs["default_aligner"] = cpu_baseline_pairs_all_cores([(q, t[:1000]) for q, t in p1k], 1024, 3.0)
        if "aligner_matrix" in subs:  # the reference's own needleman_wunsch_cpu on the matrix's two genome sizes
            for size, count in ((2048, 1024), (8192, 256)):
                pm = synthetic.generate_pairs(1, count, size, size // 30, size // 30, size // 30)
                cpu_pairs["matrix_%d" % size] = cpu_baseline_pairs_all_cores([(q, t[:size]) for q, t in pm], 1024, 3.0,
                                                                             max_cores=None if size <= 2048 else 32)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if ranks_per_device > 1 else "nccl", rank=rank, world_size=world)

    from genomeworks_amd import cudapoa, multi_gpu
    from genomeworks_amd.cuda import cuda_set_device
    cuda_set_device(local_rank)

    def new_batch(max_mem=8 << 30):
        return cudapoa.CudaPoaBatch(32, 1024, max_mem, output_type="consensus", band_mode="static_band",
                                    alignment_band_width=256, max_nodes_per_graph=3072, device_id=local_rank)

    batch = new_batch()
    for w in windows:
        st, _ = batch.add_poa_group(w)
        assert st == 0, st

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # cold pass (first launch of the process: code object load, first touch of the slabs)
    sync()
    t0 = time.perf_counter()
    batch.generate_poa()
    n_ok = batch.get_consensus_native()
    t_cold = time.perf_counter() - t0
    cells = batch.total_cells()
    assert n_ok == args.windows

    # ---- the metric: steady-state generate_poa() + get_consensus() ----
    for _ in range(args.warmup):
        batch.generate_poa()
        batch.get_consensus_native()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.generate_poa()
        batch.get_consensus_native()
    sync()
    elapsed = time.perf_counter() - t0
    (elapsed,) = reduce_scalars(dist, torch, [elapsed], "MAX")
    (total_cells,) = reduce_scalars(dist, torch, [float(cells)], "SUM")
    # the line certifies itself: the consensus of the LAST timed step of every rank against the oracle golden (outside the clock)
    cons_last, _cov_last, status_last = batch.get_consensus()
    my_digest = consensus_digest(cons_last)
    golden = config3_golden_digest()
    checkable = golden is not None and args.windows == WINDOWS
    ok_here = 1.0 if (checkable and my_digest == golden and all(int(x) == 0 for x in status_last)) else 0.0
    (ok_all,) = reduce_scalars(dist, torch, [ok_here], "MIN")
    equals_golden = (ok_all == 1.0) if checkable else None

    # ---- extension, reported next to the metric and never as the metric: the same step with the library's
    # get_consensus_in_place (results into the storage of the previous call; cudapoa::Batch has no such entry point) ----
    batch.generate_poa()
    batch.get_consensus_native(in_place=True)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.generate_poa()
        batch.get_consensus_native(in_place=True)
    sync()
    in_place = time.perf_counter() - t0
    (in_place,) = reduce_scalars(dist, torch, [in_place], "MAX")

    # ---- the same kernels on inputs resident in HBM (no H2D) ----
    batch.relaunch()
    batch.get_consensus_native()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.relaunch()
        batch.get_consensus_native()
    sync()
    resident = time.perf_counter() - t0
    (resident,) = reduce_scalars(dist, torch, [resident], "MAX")

    # dominant kernel, timed live with HIP events on the batch's own stream (outside the timed regions above)
    kms, oms = [], []
    for _ in range(max(3, args.steps)):
        a, b = batch.relaunch_timed()
        kms.append(a)
        oms.append(b)
    k_ms = sum(kms) / len(kms)
    o_ms = sum(oms) / len(oms)

    # ---- strong scaling of the metric config: the SAME windows over the ranks, at 1024 windows (one wavefront per SIMD of ONE
    # device already: splitting them further idles SIMDs) and at 8 x 1024 windows (the size at which an index split can scale:
    # the 1024 golden windows eight times over, so every block of 1024 results is checked against the same golden) ----
    strong = strong8 = None
    if world > 1:
        base_windows = [[r.decode() for r in synthetic.generate_window(1000 + w)] for w in range(args.windows)]

        def strong_record(copies):
            all_windows = base_windows * copies
            cost = [multi_gpu.poa_window_cost(w, 256) for w in all_windows]
            timing = {}

            def process(units, idx):
                sb = new_batch(max(8 << 30, len(units) * (5 << 20)))
                for w in units:
                    st, _ = sb.add_poa_group(w)
                    assert st == 0, st
                sb.generate_poa()
                sb.get_consensus_native()
                sync()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    sb.generate_poa()
                    sb.get_consensus_native()
                sync()
                timing["s"] = time.perf_counter() - t1
                cons, _cov, status = sb.get_consensus()
                timing["cells"] = sb.total_cells()
                return [(c, st) for c, st in zip(cons, status)]

            gathered = multi_gpu.run_sharded(all_windows, process, gather=True, costs=cost)
            s_elapsed, = reduce_scalars(dist, torch, [timing["s"]], "MAX")
            s_cells, = reduce_scalars(dist, torch, [float(timing["cells"])], "SUM")
            if rank != 0:
                return None
            n = args.windows
            digests = [consensus_digest([c for c, _ in gathered[k * n:(k + 1) * n]]) for k in range(copies)]
            return {"scaling": "strong", "windows": len(all_windows), "ms_per_step": round(s_elapsed / args.steps * 1e3, 3),
                    "gcups": round(s_cells * args.steps / s_elapsed / 1e9, 3),
                    "windows_per_s": round(len(all_windows) * args.steps / s_elapsed, 1),
                    "consensus_sha256": digests[0],
                    "equals_oracle_golden": all(d == golden for d in digests) if checkable else None,
                    "split": "balanced_partition by estimated cells, results gathered by global window index"}

        strong = strong_record(1)
        strong8 = strong_record(8)

    # ---- sub-records: the other BASELINE configs ----
    sub = {}
    cpu_s = 0 if args.no_cpu_baseline else 1
    if "aligner" in subs:
        sub["configs[1]"] = bench_aligner("BASELINE configs[1]: cudaaligner banded Myers, 10 000 pairs x 1 kbp, <=33 sub/ins/del, "
                                          "max_bandwidth 1024", CONFIG2, rank, world, local_rank, sync, dist, torch, 3, 3.0 * cpu_s,
                                          cpu_pairs.get("configs[1]"))
        sub["configs[4]"] = bench_aligner("BASELINE configs[4]: cudaaligner 1 000 000 pairs x 150 bp, <=2 sub, <=1 ins, <=1 del, "
                                          "max_bandwidth 150, index split over the ranks", CONFIG5, rank, world, local_rank, sync, dist, torch, 2,
                                          3.0 * cpu_s, cpu_pairs.get("configs[4]"))
    if "default_aligner" in subs and rank == 0:
        sub["default_aligner"] = bench_default_aligner(local_rank, sync if world == 1 else (lambda: torch.cuda.synchronize()),
                                                       cpu_pairs.get("default_aligner"))
    if "aligner_matrix" in subs and rank == 0:
        sub["aligner_matrix"] = bench_aligner_matrix(local_rank, sync if world == 1 else (lambda: torch.cuda.synchronize()),
                                                     {2048: cpu_pairs.get("matrix_2048"), 8192: cpu_pairs.get("matrix_8192")})
    if "aligner_grid" in subs and rank == 0:  # (not in the default line: the 65 kbp cells take seconds each)
        sub["aligner_grid"] = bench_aligner_matrix(local_rank, sync if world == 1 else (lambda: torch.cuda.synchronize()), None, corners=True)
    if "band_modes" in subs and world == 1:
        sub["band_modes"] = bench_band_modes(windows, local_rank, sync)
    if "reference_shapes" in subs and world == 1:
        sub["reference_benchmark_shapes"] = bench_reference_shapes(windows, local_rank, sync, 2)
    if "long_reads" in subs:
        sub["configs[3]"] = bench_long_reads(args.long_read_windows, rank, world, local_rank, sync, dist, torch, cpu_long,
                                             ranks_per_device)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        gcups = total_cells * args.steps / elapsed / 1e9
        achieved = cells * BYTES_PER_CELL / (k_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC pass of this same workload (rocprofv3 cannot run inside the
        # timed region; tools/pmc_passes.sh collects FETCH_SIZE / WRITE_SIZE in their own runs)
        traffic = sub_traffic("headline") if args.windows == WINDOWS else None
        out = {
            "metric": "cudapoa consensus GCUPS, 1024-window short-read batch (static band 256, 32 reads <= 1024 bp)",
            "value": round(gcups, 3), "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16", "data": "synthetic",
            "equals_oracle_golden": equals_golden, "consensus_sha256": my_digest,
            "windows_per_s": round(world * args.windows * args.steps / elapsed, 1),
            "timed_region": "generate_poa() [H2D from pinned host + graph-build kernel + consensus kernel] + the PUBLIC "
                            "Batch::get_consensus() into three fresh vectors per step [D2H + host un-reversal; the previous step's "
                            "results destroyed], steady state on one Batch: the body of SingleBatch::process_consensus(), "
                            "cudapoa/benchmarks/single_batch.hpp:86-93",
            "config": {"workload": "BASELINE configs[2]: cudapoa single-batch consensus, %d windows x 32 reads, "
                                   "backbone 960 bp, <=48 sub/24 ins/24 del, BatchConfig(1024,32,256,static_band), "
                                   "scores 8/-6/-8" % args.windows,
                       "windows_per_gpu": args.windows, "cells_per_gpu": cells, "parallelism": "index-split x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "poa_window_kernel<int16,int16,static_band>",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "stale": pmc_stale(),
                         "traffic_source": pmc_profile().get("file"),
                         "algorithmic_bytes_per_launch": cells * BYTES_PER_CELL,
                         "kernel_ms": round(k_ms, 3), "output_kernel_ms": round(o_ms, 3),
                         "algorithmic_bytes_per_cell": BYTES_PER_CELL},
            "roofline_issue": roofline_issue("headline"),
            "kernel_only": {"what": "relaunch on inputs resident in HBM + get_consensus() (no H2D)",
                            "ms_per_step": round(resident / args.steps * 1e3, 3),
                            "gcups": round(total_cells * args.steps / resident / 1e9, 3),
                            "windows_per_s": round(world * args.windows * args.steps / resident, 1)},
            "extension_get_consensus_in_place": {
                "what": "the same step with gw_poa_get_consensus_in_place (result storage of the previous call reused; not part of "
                        "cudapoa::Batch, not the metric)",
                "ms_per_step": round(in_place / args.steps * 1e3, 3),
                "gcups": round(total_cells * args.steps / in_place / 1e9, 3)},
            "cold_first_pass_ms": round(t_cold * 1e3, 3),
        }
        if world > 1:
            out["scaling_note"] = ("weak scaling by construction: every rank runs the same 1024-window batch; the comparable "
                                   "strong-scaling records are strong_scaling (1024 windows) and strong_scaling_8x (8 x 1024 windows "
                                   "index-split over the ranks), see the sub-record lines / file")
        if cpu is not None:  # timed on rank 0 of the 1-GPU run only
            out["cpu_baseline"] = cpu
        emit(out, strong, strong8, sub, args.record_dir)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

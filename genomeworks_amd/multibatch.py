"""The multi-batch loop of the reference's cudapoa tool (cudapoa/src/main.cpp:197-326) for a planned set of windows:
one Batch per BatchConfig of the plan, filled until exceeded_maximum_poas, generate_poa + get_msa / get_consensus,
reset, next fill. Used by bench.py (BASELINE configs[3]), tools/bench_long_read_msa.py and the GPU parity tests."""
import time

from . import cudapoa


def run_plan(windows, cfgs, groups, max_gpu_mem, output_type="msa", band_mode="adaptive_band", device_id=0,
             collect=True, kernel_time=False, on_launch=None, digest=None, longest_first=True):
    """Run every window of the plan. Returns dict(results={w: (rows_or_consensus, status)}, cells, seconds (the
    reference benchmark's timed region: generate_poa + get_msa, H2D included), kernel_ms (graph-build kernels of one
    extra resident launch per fill, HIP events; None unless kernel_time), launches, accepted={w: reads the batch took}).
    `digest(rows)` replaces the rows of a window by a digest as soon as they are fetched (long MSAs are large).
    longest_first: a launch lasts as long as its heaviest window (every window is one chain of dependent alignments),
    so the windows of a bin are dealt to the fills heaviest first -- the later fills then end early instead of each
    waiting for one long window (the reference bins windows by size for the same reason, cudapoa/src/utils.cu:66-146).
    Results are keyed by window index, so the order of the fills is not observable."""
    results, accepted = {}, {}
    cells = 0
    seconds, kernel_ms, launches = 0.0, 0.0, 0
    for cfg, members in zip(cfgs, groups):
        batch = cudapoa.CudaPoaBatch(cfg["max_sequences_per_poa"], cfg["max_sequence_size"], int(max_gpu_mem),
                                     output_type=output_type, band_mode=band_mode, device_id=device_id,
                                     alignment_band_width=cfg["alignment_band_width"],
                                     max_consensus_size=cfg["max_consensus_size"],
                                     max_nodes_per_graph=cfg["max_nodes_per_graph"],
                                     matrix_sequence_dimension=cfg["matrix_sequence_dimension"],
                                     max_banded_pred_distance=cfg["max_banded_pred_distance"])
        pending = list(members)
        if longest_first:
            pending.sort(key=lambda g: (-sum(len(r) for r in windows[g]) * max((len(r) for r in windows[g]), default=0), g))
        while pending:
            taken = []
            batch_full = False
            while pending:
                st, seq_st = batch.add_poa_group(windows[pending[0]])
                if st == cudapoa.exceeded_maximum_poas:
                    batch_full = True
                    break
                g = pending.pop(0)
                if st == cudapoa.success:
                    taken.append(g)
                    # a bin's BatchConfig may hold fewer reads than its deepest window: the batch keeps the reads it
                    # accepted (the others report exceeded_maximum_sequences_per_poa)
                    accepted[g] = [s for s, ss in zip(windows[g], seq_st) if ss == cudapoa.success]
                elif st == cudapoa.empty_poa_group and windows[g]:
                    # every read was rejected AFTER the batch opened a POA for the group: as in the reference
                    # (cudapoa_batch.cuh:122-150) that empty POA stays in the batch and owns an output slot
                    taken.append(None)
                    results[g] = (None, st)
                else:
                    # rejected by the batch (e.g. its longest read exceeds the bin's BatchConfig): recorded with its
                    # status like the C++ drivers do (out.status[w]), never dropped
                    results[g] = (None, st)
            if not taken:
                if batch_full:  # the batch was full before it held one window
                    raise RuntimeError("a batch of this plan cannot hold a single window")
                continue  # every window of this fill was rejected: nothing to launch
            t0 = time.perf_counter()
            batch.generate_poa()
            n_out = batch.get_msa_native() if output_type == "msa" else batch.get_consensus_native()
            dt = time.perf_counter() - t0
            seconds += dt
            cells += batch.total_cells()
            assert n_out == len(taken)
            if collect:
                if output_type == "msa":
                    for slot, g in enumerate(taken):
                        if g is None:
                            continue
                        rows, st = batch.collect_msa_one(slot)
                        results[g] = (digest(rows) if (digest and st == cudapoa.success) else rows, st)
                else:
                    cons, _cov, status = batch.get_consensus()
                    for g, c, st in zip(taken, cons, status):
                        if g is not None:
                            results[g] = (c, st)
            if kernel_time:
                k_ms, _o_ms = batch.relaunch_timed()
                kernel_ms += k_ms
            if on_launch:
                on_launch(launches, [g for g in taken if g is not None], dt)
            launches += 1
            batch.reset()
        del batch
    return dict(results=results, cells=cells, seconds=seconds, kernel_ms=kernel_ms if kernel_time else None,
                launches=launches, accepted=accepted)

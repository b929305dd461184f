#!/usr/bin/env python3
"""One JSON per measurement session from the summary of the PMC passes (tools/pmc_passes.sh -> tools/pmc_summary.py) over the
headline kernel and the sub-records' kernels: what bench.py needs next to its live timings and cannot collect inside the timed
region -- HBM traffic per launch (FETCH_SIZE x 2, WRITE_SIZE x 0.97: the calibration of profiles/r03_pmc_traffic.json), issued
instructions, wave cycles, wait share and LDS bank-conflict share per launch.

  python tools/pmc_profile.py <pmc_summary.csv> <tag> <out_dir> [<commit>]   ->  <out_dir>/pmc_profile.json
"""
import csv
import json
import os
import sys

rows = {}
for r in csv.DictReader(open(sys.argv[1])):
    rows.setdefault(r["kernel"], {})[r["counter"]] = (float(r["mean_per_dispatch"]), int(r["dispatches"]))
# optional fifth argument: the summary of the passes over the full-band kernel alone (gpu_session.sh step pmcfull); its kernel
# names get a prefix so that they cannot mix with the main session's
if len(sys.argv) > 5 and os.path.exists(sys.argv[5]):
    for r in csv.DictReader(open(sys.argv[5])):
        rows.setdefault("full_band_session::" + r["kernel"], {})[r["counter"]] = (float(r["mean_per_dispatch"]), int(r["dispatches"]))
tag = sys.argv[2] if len(sys.argv) > 2 else "session"
out_dir = sys.argv[3] if len(sys.argv) > 3 else "."
commit = sys.argv[4] if len(sys.argv) > 4 else os.environ.get("GW_COMMIT", "unknown")

# a CONSTANT, not a counter: the issue rate of a lone wavefront measured once by a microbenchmark in the production launch geometry
# (profiles/r03_microbench_instruction_size.json: 4.1-4.2 cycles per 4-byte instruction); every record says so
LONE_WAVE_CYCLES_PER_INST = 4.1
LONE_WAVE_SOURCE = "constant from the round-3 microbenchmark profiles/r03_microbench_instruction_size.json, not measured in this session"


def entry(match, what="mean", note=None, launches_per_call=1):
    ks = sorted(k for k in rows if match(k))
    if not ks:
        return None

    def tot(counter):
        v, seen = 0.0, False
        for k in ks:
            if counter in rows[k]:
                m, n = rows[k][counter]
                # (summaries written before round 6 counted a counter once per pass that collected it: the launches of the set are
                # those of SQ_WAVES, which one pass collects)
                n_set = rows[k]["SQ_WAVES"][1] if "SQ_WAVES" in rows[k] else n
                v += m * (min(n, n_set) if what == "sum" else launches_per_call)
                seen = True
        return v if seen else None
    e = {"kernel": " + ".join(k.replace("void ", "").replace("full_band_session::", "") for k in ks),
         "per": "sum over the launches of the set" if what == "sum" else ("launch (mean)" if launches_per_call == 1 else "call (%d launches)" % launches_per_call)}
    f, w = tot("FETCH_SIZE"), tot("WRITE_SIZE")
    if f is not None and w is not None:
        e["read_bytes"], e["write_bytes"] = int(f * 1024 * 2), int(w * 1024 * 0.97)
        e["hbm_bytes"] = e["read_bytes"] + e["write_bytes"]
    insts = [tot(c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM")]
    if all(v is not None for v in insts):
        e["instructions"] = {"valu": int(insts[0]), "salu": int(insts[1]), "lds": int(insts[2]), "vmem_rd": int(insts[3]),
                             "vmem_wr": int(insts[4]), "smem": int(insts[5]), "total": int(sum(insts))}
        br = tot("SQ_INSTS_BRANCH")
        if br is not None:
            e["instructions"]["branch_of_salu"] = int(br)
    waves, wc, wa = tot("SQ_WAVES"), tot("SQ_WAVE_CYCLES"), tot("SQ_WAIT_ANY")
    if waves:
        e["waves"] = int(waves)
    if wc:
        e["wave_cycles"] = int(wc * 4)  # the counter ticks once per 4 cycles of a resident wave
        if wa is not None:
            e["wait_share"] = round(wa / wc, 4)
        if waves and "instructions" in e:
            per_wave = e["instructions"]["total"] / waves
            e["issue"] = {"instructions_per_wave": round(per_wave, 1), "cycles_per_wave": round(wc * 4 / waves, 1),
                          "cycles_per_instruction": round(wc * 4 / waves / per_wave, 2),
                          "lone_wave_cycles_per_instruction": LONE_WAVE_CYCLES_PER_INST,
                          "lone_wave_cycles_per_instruction_source": LONE_WAVE_SOURCE,
                          "frac_of_lone_wave_issue_bound": round(per_wave * LONE_WAVE_CYCLES_PER_INST / (wc * 4 / waves), 4)}
    # the compute-side fraction: how busy the vector ALUs are. By instruction count (every wave64 VALU instruction
    # holds its SIMD's issue port for 4 cycles; wave cycles = cycles a wavefront was resident, and these kernels keep one wavefront
    # per SIMD) and, when the `active` pass ran, by the SQ's own counter of cycles with a VALU instruction in flight
    if wc and "instructions" in e:
        e["valu_busy"] = {"by_instruction_count": round(e["instructions"]["valu"] * 4.0 / (wc * 4), 4),
                          "what": "VALU instructions x 4 cycles / resident wave cycles (one wavefront per SIMD in these kernels)"}
        av, awc = tot("SQ_ACTIVE_INST_VALU"), tot("SQ_WAVE_CYCLES")
        if av is not None and awc:
            e["valu_busy"]["sq_active_inst_valu_over_wave_cycles"] = round(av / awc, 4)
    bc, ia = tot("SQ_LDS_BANK_CONFLICT"), tot("SQ_LDS_IDX_ACTIVE")
    if bc is not None and ia:
        e["lds"] = {"bank_conflict_cycles": int(bc), "idx_active_cycles": int(ia), "bank_conflict_share_of_lds_active": round(bc / ia, 4)}
        if wc:
            e["lds"]["bank_conflict_share_of_wave_cycles"] = round(bc / (wc * 4), 5)
    if note:
        e["note"] = note
    return e


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomeworks_amd.build import kernel_source_digest  # noqa: E402

# optional: the lone-wave issue constant measured in THIS session (tools/microbench_issue.hip, gpu_session.sh step issue)
issue_json = os.path.join(out_dir, "microbench_issue.json")
if os.path.exists(issue_json):
    try:
        m = json.load(open(issue_json))
        LONE_WAVE_SOURCE = "measured in this session by tools/microbench_issue.hip (%s)" % m.get("what", "independent 4-byte VALU instructions, one wavefront per SIMD")
        LONE_WAVE_CYCLES_PER_INST = float(m["lone_wave_cycles_per_instruction"])
    except (OSError, ValueError, KeyError):
        pass

out = {"tag": tag, "commit": commit, "kernel_source_sha256": kernel_source_digest(),
       "source": "rocprofv3 --kernel-trace --pmc, one counter group per pass (tools/pmc_passes.sh: insts, waits, lds, fetch, write) over "
                 "`bench.py --steps 1 --warmup 0 --no-cpu-baseline --sub-configs aligner,default_aligner,long_reads`; FETCH_SIZE x 2 and "
                 "WRITE_SIZE x 0.97 as calibrated in profiles/r03_pmc_traffic.json on microkernels of known size"}
for key, e in (("headline", entry(lambda k: "poa_window_kernel<short, short, signed char, 1, false, tru" in k and not k.startswith("full_band_session::"))),
               ("full_band", entry(lambda k: k.startswith("full_band_session::") and "poa_window_kernel<short, short, signed char, 0, false, tru" in k, "mean",
                                   "the 1024 metric windows under BatchConfig(1024, 200), launches of the full batch only")),
               ("configs[1]", entry(lambda k: "myers_banded_group_kernel" in k)),
               # (round 5: align_all() cuts the million pairs into six chunks, one launch each: per call = 6 x the mean launch)
               ("configs[4]", entry(lambda k: "myers_banded_kernel<true>" in k, "mean",
                                    "six chunk launches per align_all(): the mean launch x 6", launches_per_call=6)),
               ("default_aligner", entry(lambda k: "hirschberg_levels_kernel" in k or "hirschberg_wave_kernel" in k or "hb_span_" in k, "sum",
                                         "all Hirschberg kernels of the sub-record's shapes (1 .. 2000 pairs; the hb_span_* kernels are the long "
                                         "single pairs' span path), summed over their launches")),
               ("configs[3]", entry(lambda k: "poa_window_kernel<" in k and "2, true, false" in k, "sum",
                                    "sum over the launches of the set's size classes"))):
    if e:
        out[key] = e
os.makedirs(out_dir, exist_ok=True)
with open(os.path.join(out_dir, "pmc_profile.json"), "w") as f:
    json.dump(out, f, indent=1)
    f.write("\n")
print(json.dumps({k: {kk: v[kk] for kk in ("hbm_bytes", "wait_share") if kk in v} | ({"issue": v["issue"]} if "issue" in v else {})
                  for k, v in out.items() if isinstance(v, dict)}, indent=1))

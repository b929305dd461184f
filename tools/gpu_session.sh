#!/bin/bash
# One parameterised GPU-box session (replaces the per-call tools/r0N_gpu_run_*.sh scripts of rounds 2-3, which are in the
# git history). Run from the repo root through gpurun:
#
#   gpurun --timeout 1500 -- 'STEPS="tests bench stats" bash tools/gpu_session.sh r04a'
#
# STEPS (any subset, run in this order):
#   tests        pytest -m gpu (PYTEST_ARGS = files, PYTEST_K = a -k expression narrow it)
#   phases       per-phase cycle breakdown of the metric kernel (tools/profile_phases.py 1024)
#   bench        the driver's bench line (BENCH_ARGS, default none = the full line with all sub-records)
#   stats        rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline $STATS_ARGS` -> kernel_stats.csv
#   pmc          tools/pmc_passes.sh with PASSES (default "insts waits lds fetch write active") and SUBS (default none)
#   pmcfull      the same passes over the full-band kernel alone (tools/profile_phases.py 1024 full_band) -> pmc_summary_full_band.csv
#   issue        measure the lone-wavefront issue rate (tools/microbench_fetch.hip) -> microbench_issue.json, which the traffic step
#                writes into pmc_profile.json instead of the round-3 constant
#   traffic      derive pmc_profile.json (HBM traffic, issue and LDS counters per record) from the pmc step's summary; pass
#                GW_COMMIT=$(git rev-parse --short HEAD) from the submitting side (the box has no .git)
#   extra        run $EXTRA_CMD (one-off measurements)
# Everything lands in gpurun_out/<tag>/; copy what should be judged into profiles/.
set -u
TAG=${1:-session}
STEPS=${STEPS:-tests bench}
REPO=$(pwd)
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { case " $STEPS " in *" $1 "*) return 0;; *) return 1;; esac; }

if has tests; then
    # PYTEST_K: a -k expression (may contain spaces)
    ( timeout ${TESTS_TIMEOUT:-1500} python -m pytest ${PYTEST_ARGS:-tests} -m gpu -q ${PYTEST_K:+-k "$PYTEST_K"} 2>&1 | tail -${PYTEST_TAIL:-25} ) > $OUT/pytest.log
    tail -${PYTEST_TAIL:-25} $OUT/pytest.log
fi
if has phases; then
    timeout 300 python tools/profile_phases.py 1024 2>/dev/null | tail -1 > $OUT/phase_breakdown.json
    cut -c1-900 $OUT/phase_breakdown.json; echo
fi
if has bench; then
    ( timeout ${BENCH_TIMEOUT:-1500} python bench.py --record-dir $OUT ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err ); echo "bench rc=$?"; tail -3 $OUT/bench.err
    python - $OUT/bench.json <<'PY'
import json, sys
import os
lines = [x for x in open(sys.argv[1]) if x.startswith('{')]
if not lines:
    sys.exit("no bench line")
d = json.loads(lines[-1])
print("final line bytes:", len(lines[-1].encode()))
full = os.path.join(os.path.dirname(sys.argv[1]), "bench_full_record.json")
if os.path.exists(full):
    d = json.load(open(full))
print("headline", d['value'], d['unit'], "step ms", d['ms_per_step'], "kernel ms", d['roofline']['kernel_ms'], "frac", d['roofline']['frac'],
      "golden", d.get('equals_oracle_golden'))
s = d.get('sub_records', {})
for k in ('configs[1]', 'configs[4]', 'default_aligner', 'configs[3]'):
    if k in s:
        print(k, s[k].get('value'), s[k].get('ms'), s[k].get('kernel_only'), s[k]['roofline'].get('frac'))
if 'band_modes' in s:
    for r in s['band_modes']['rows']:
        print("band", r['band_mode'], r['band_width'], "kernel ms", r['kernel_ms'], "gcups", r['gcups'], "golden", r.get('equals_oracle_golden'))
PY
fi
if has stats; then
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/stats -- python $REPO/bench.py --no-cpu-baseline ${STATS_ARGS:-} > $REPO/$OUT/bench_under_rocprof.json 2> $REPO/$OUT/stats.log)
    DB=$(find $OUT/stats -name "*.db" | head -1)
    [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" > $OUT/kernel_stats.csv && head -14 $OUT/kernel_stats.csv | cut -c1-160
    rm -rf $OUT/stats
fi
if has pmc; then
    SUBS=${SUBS:-none} PASSES="${PASSES:-insts waits lds fetch write active}" bash tools/pmc_passes.sh $OUT/pmc > $OUT/pmc.log 2>&1
    python tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.csv 2>/dev/null
    rm -rf $OUT/pmc
    grep -c . $OUT/pmc_summary.csv
    grep -E "LDS|WAIT_ANY|WAVE_CYCLES|FETCH|WRITE_SIZE" $OUT/pmc_summary.csv | cut -c1-170 | head -60
fi
if has pmcfull; then
    # the full-band kernel of the reference benchmarks' BatchConfig(1024, 200) on the 1024 metric windows, alone
    PMC_CMD="python $REPO/tools/profile_phases.py 1024 full_band" PASSES="${PASSES:-insts waits lds fetch write}" bash tools/pmc_passes.sh $OUT/pmcfull > $OUT/pmcfull.log 2>&1
    python tools/pmc_summary.py $OUT/pmcfull > $OUT/pmc_summary_full_band.csv 2>/dev/null
    rm -rf $OUT/pmcfull
    grep -E "WAVE_CYCLES|FETCH|WRITE_SIZE|WAIT_ANY" $OUT/pmc_summary_full_band.csv | cut -c1-170 | head -12
fi
if has issue; then
    # the lone-wavefront issue rate that roofline_issue prices instructions with, measured in THIS session
    mkdir -p tools/bin
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/microbench_fetch.hip -o tools/bin/microbench_fetch 2> $OUT/issue_build.log \
        && timeout 300 tools/bin/microbench_fetch > $OUT/microbench_instruction_size.json
    python - $OUT <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/microbench_instruction_size.json"))["cycles_per_instruction"]
lone = [r for r in d if r["waves_per_simd"] == 1]
pick = next(r for r in lone if r["chain"].startswith("v_add_u32 e32, 4 independent"))
json.dump({"lone_wave_cycles_per_instruction": pick["cycles"], "what": pick["chain"] + ", one wavefront per SIMD",
           "all_lone_wave_rows": lone}, open(sys.argv[1] + "/microbench_issue.json", "w"), indent=1)
print("lone-wave cycles per instruction:", pick["cycles"])
PY
fi
if has traffic; then
    python tools/pmc_profile.py $OUT/pmc_summary.csv "$TAG" $OUT "${GW_COMMIT:-unknown}" $OUT/pmc_summary_full_band.csv
fi
if has extra; then
    bash -c "${EXTRA_CMD:-true}" > $OUT/extra.log 2>&1; echo "extra rc=$?"; tail -${EXTRA_TAIL:-30} $OUT/extra.log
fi
du -sh $OUT

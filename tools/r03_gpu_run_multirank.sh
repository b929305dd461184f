#!/bin/bash
# Exercises bench.py's multi-rank paths (index split of the aligner and long-read sub-records, both strong-scaling records with
# the gather by global index and the golden checks, the per-rank golden digests of the weak line) on a single-GPU box: 2 ranks
# on device 0. The numbers mean nothing (two ranks share one GPU); the JSON line must be complete and every check must pass.
set -u
TAG=${1:-r03mr}
mkdir -p gpurun_out/${TAG}
export TMPDIR=/tmp GW_BENCH_RANKS_PER_DEVICE=2
( timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --long-read-windows 150 > gpurun_out/${TAG}/bench2.json 2> gpurun_out/${TAG}/bench2.err ) ; echo "rc=$?" >> gpurun_out/${TAG}/bench2.err
tail -c 1500 gpurun_out/${TAG}/bench2.err
python - gpurun_out/${TAG}/bench2.json <<'PY'
import json,sys
d=json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][0])
print({k:d[k] for k in ('value','n_gpus','scaling','equals_oracle_golden') if k in d})
print('strong', {k:(v.get('equals_oracle_golden'), v.get('ms') or v.get('ms_per_step')) for k,v in d.items() if k.startswith('strong')})
print('subs', list(d['sub_records'].keys()))
for k,v in d['sub_records'].items():
    if isinstance(v,dict): print(k, v.get('value'), v.get('windows_equal_to_oracle_golden'))
PY

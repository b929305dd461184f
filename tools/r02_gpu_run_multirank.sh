#!/bin/bash
# Exercises bench.py's multi-rank paths (index split of the aligner and long-read sub-records, strong-scaling record with the
# gather by global index and the golden check) on a single-GPU box: 2 ranks on device 0 over gloo. The numbers mean nothing
# (two ranks share one GPU); the JSON line must be complete and the golden checks must pass.
set -u
TAG=${1:-r02mr}
mkdir -p gpurun_out/${TAG}
export TMPDIR=/tmp GW_BENCH_RANKS_PER_DEVICE=2
( timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --long-read-windows 150 > gpurun_out/${TAG}/bench2.json 2> gpurun_out/${TAG}/bench2.err ) ; echo "rc=$?" >> gpurun_out/${TAG}/bench2.err
tail -c 3000 gpurun_out/${TAG}/bench2.err

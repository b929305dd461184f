#!/bin/bash
# Same-box A/B of two builds of the native libraries for an arbitrary command: genomeworks_amd/lib (candidate) against
# genomeworks_amd/lib_old (baseline), alternating. usage (GPU box, repo root): AB_CMD='python tools/...' bash tools/ab_cmd.sh [rounds]
ROUNDS=${1:-2}
for i in $(seq $ROUNDS); do
    echo "candidate: $(bash -c "$AB_CMD" 2>/dev/null | tail -1 | cut -c1-${AB_CUT:-300})"
    mv genomeworks_amd/lib genomeworks_amd/lib_new && mv genomeworks_amd/lib_old genomeworks_amd/lib
    echo "baseline:  $(bash -c "$AB_CMD" 2>/dev/null | tail -1 | cut -c1-${AB_CUT:-300})"
    mv genomeworks_amd/lib genomeworks_amd/lib_old && mv genomeworks_amd/lib_new genomeworks_amd/lib
done

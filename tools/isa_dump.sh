#!/bin/bash
# Dump the gfx950 ISA of one poa_window_kernel instantiation (default: the config-3 kernel) to $OUT (default /tmp/isa/k.s).
# usage: tools/isa_dump.sh ["int16_t,int16_t,int8_t,1,false,true,1"]
set -e
INST=${1:-"int16_t,int16_t,int8_t,1,false,true,1,0"}
OUT=${OUT:-/tmp/isa/k.s}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$(dirname "$OUT")"
TMP=$(mktemp --suffix=.hip)
cat > "$TMP" <<EOS
#define GWHIP_DEVICE_ONLY
#include "$ROOT/genomeworks_amd/csrc/gwhip_poa.hip"
template __global__ void gwhip::poa_window_kernel<$INST>(gwhip::KernelArgs);
EOS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off \
    -fhip-fp32-correctly-rounded-divide-sqrt -I "$ROOT/include" --cuda-device-only -S "$TMP" -o "$OUT"
rm -f "$TMP"
grep -E "NumVgprs|NumSgprs|ScratchSize|Occupancy|sgpr_spill|vgpr_spill" "$OUT" | head -8
wc -l "$OUT"

#!/bin/bash
# The tile kernel's sphere filter on bf16 MFMAs (3-way split operands) against binary32 MFMAs, and S / W choices of the bf16 build.
# Variants (scripts/build_variant.sh with ALL=1): f32 = -DRT_TILE_BF16=0 -DRT_TUNING, bf_s4 = -DRT_TUNING, bf_s8 = -DRT_TILE_S=8 -DRT_TUNING
run() { # name, RT_TILE_W, synth spec...
  local name=$1 w=$2; shift 2
  RT_TILE_W=$w RAYTRACER_AMD_LIB=$PWD/raytracer_amd/csrc/build/variants/$name/libraytracer_amd.so timeout -k 10 300 python3 scripts/run_configs.py --skip-samples "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l)
    print('$name W=$w', j['config'], j['size'], j['kernel'], round(j['frame_ms'],1), 'ms', round(j['Mrays_per_s']), 'Mrays/s', j['rays'])" || exit 1
}
for v in f32 bf_s4 bf_s8; do
  for w in 1 2 4; do
    run $v $w --synth 512 1920 1080 4 || exit 1
  done
  for w in 2 4 8; do
    run $v $w --synth 4096 480 136 4 || exit 1
  done
done

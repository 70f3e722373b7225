#!/bin/bash
# waves per workgroup of the tile kernel (RT_TILE_W) on N = 4096 (68-row slice), 1024 and 512
# The RT_* environment knobs exist only in -DRT_TUNING builds:  scripts/build_variant.sh tuning -DRT_TUNING, then
# RAYTRACER_AMD_LIB=$PWD/raytracer_amd/csrc/build/variants/tuning/libraytracer_amd.so <this script> (the shipped library ignores them).
for spec in "4096 68" "1024 270" "512 1080"; do
  set -- $spec
  for w in 1 2 4 8; do
    RT_TILE_W=$w timeout -k 10 300 python3 -u bench.py --no-configs --no-cpu-baseline --steps 1 --spheres $1 --height $2 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=$1 W=$w', round(j['ms_per_step'],1), round(j['value'],1), j['roofline'].get('kernel'))" || exit 1
  done
done

#!/bin/bash
# where the tile kernel starts to beat the per-lane scan of the wave kernel: small synthetic worlds with RT_WIDE_MIN = 16 and 1000
# The RT_* environment knobs exist only in -DRT_TUNING builds:  scripts/build_variant.sh tuning -DRT_TUNING, then
# RAYTRACER_AMD_LIB=$PWD/raytracer_amd/csrc/build/variants/tuning/libraytracer_amd.so <this script> (the shipped library ignores them).
for n in 16 24 32 48 64; do
  for m in 16 1000; do
    RT_WIDE_MIN=$m timeout -k 10 200 python3 -u bench.py --no-configs --no-cpu-baseline --steps 2 --spheres $n --height 540 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=$n wide_min=$m', round(j['ms_per_step'],1), round(j['value'],1), j['roofline'].get('kernel'))" || exit 1
  done
done

#!/bin/bash
# C3 / C4 frame times of a variant library at several waves-per-SIMD settings of the wave kernel:  scripts/ab_occ.sh VARIANT OCC...
# The RT_* environment knobs exist only in -DRT_TUNING builds:  scripts/build_variant.sh tuning -DRT_TUNING, then
# RAYTRACER_AMD_LIB=$PWD/raytracer_amd/csrc/build/variants/tuning/libraytracer_amd.so <this script> (the shipped library ignores them).
name=$1; shift
lib=""; [ "$name" != product ] && lib=raytracer_amd/csrc/build/variants/$name/libraytracer_amd.so
for occ in "$@"; do
  RT_WAVEPATH_OCC=$occ RAYTRACER_AMD_LIB=$lib timeout -k 10 120 python3 scripts/run_configs.py 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l)
    if j['mode']=='path': print('$name occ=$occ', j['config'], round(j['frame_ms'],2), j['rays'])" || exit 1
done

#!/bin/bash
# C3 / C4 frame times of one variant library at several RT_WAVEPATH_OCC values (a tuning aid of -DRT_EXP_OCC builds):  scripts/ab_occ_env.sh VARIANT OCC...
# The RT_* environment knobs exist only in -DRT_TUNING builds:  scripts/build_variant.sh tuning -DRT_TUNING, then
# RAYTRACER_AMD_LIB=$PWD/raytracer_amd/csrc/build/variants/tuning/libraytracer_amd.so <this script> (the shipped library ignores them).
v=$1; shift
for occ in "$@"; do
  RT_WAVEPATH_OCC=$occ RAYTRACER_AMD_LIB=$PWD/raytracer_amd/csrc/build/variants/$v/libraytracer_amd.so timeout -k 10 120 python3 scripts/run_configs.py 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l)
    if j['mode']=='path': print('occ $occ', j['config'], round(j['frame_ms'],2), j['frame_sha256'])"
done

#!/bin/bash
# Fresh pixel samples a wave may take per iteration (RenderParams::fresh_max).  Needs the tuning build (ALL=1 scripts/build_variant.sh tuning -DRT_TUNING).
lib=$PWD/raytracer_amd/csrc/build/variants/tuning/libraytracer_amd.so
for f in 64 32 16 8 4 2 1; do
  RT_FRESH_MAX=$f RAYTRACER_AMD_LIB=$lib timeout -k 10 120 python3 scripts/run_configs.py 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l)
    if j['mode']=='path': print('fresh_max=$f', j['config'], round(j['frame_ms'],2), 'ms', j['rays'], j['frame_sha256'])" || exit 1
done
for f in 64 16 4 2 1; do
  RT_FRESH_MAX=$f RAYTRACER_AMD_LIB=$lib timeout -k 10 300 python3 scripts/run_configs.py --skip-samples --synth 512 1920 1080 4 --synth 4096 480 136 4 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l)
    print('fresh_max=$f', j['config'], j['size'], round(j['frame_ms'],1), 'ms', round(j['Mrays_per_s']), 'Mrays/s', j['rays'], j['frame_sha256'])" || exit 1
done

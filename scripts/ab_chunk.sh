#!/bin/bash
# Sample reservation size (RenderParams::sample_chunk): RT_CHUNK_DIV = reservations per wave and launch for small worlds,
# RT_CHUNK_LARGE = samples per reservation for large worlds.  Needs the tuning build (scripts/build_variant.sh tuning -DRT_TUNING).
lib=$PWD/raytracer_amd/csrc/build/variants/tuning/libraytracer_amd.so
for div in 1 8 16 32 64 128; do
  RT_CHUNK_DIV=$div RAYTRACER_AMD_LIB=$lib timeout -k 10 120 python3 scripts/run_configs.py 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l)
    if j['mode']=='path': print('div=$div', j['config'], round(j['frame_ms'],2), 'ms', j['rays'], j['frame_sha256'])" || exit 1
done
for large in 128 16 4 1; do
  RT_CHUNK_LARGE=$large RAYTRACER_AMD_LIB=$lib timeout -k 10 300 python3 scripts/run_configs.py --skip-samples --synth 512 1920 1080 4 --synth 4096 480 136 4 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l)
    print('large=$large', j['config'], j['size'], round(j['frame_ms'],1), 'ms', round(j['Mrays_per_s']), 'Mrays/s', j['rays'], j['frame_sha256'])" || exit 1
done

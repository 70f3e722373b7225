#!/bin/bash
# What one more launch costs: the same N=4096 480x272 AA=4 frame as 1, 2, 4, 8 row-cyclic slices, with the large-world sample
# reservation at 1 (what the idle lanes ask for: the product) and at 128 (rounds 1-2).  Needs the tuning build
# (scripts/build_variant.sh tuning -DRT_TUNING).
lib=$PWD/raytracer_amd/csrc/build/variants/tuning/libraytracer_amd.so
for large in 1 128; do
  for n in 1 2 4 8; do
    RT_CHUNK_LARGE=$large RAYTRACER_AMD_LIB=$lib timeout -k 10 200 python3 -u scripts/full_frame.py --width 480 --height 272 --slices $n 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
print('reservation $large: slices', j['slices'], 'kernel_s', round(j['kernel_s_one_gpu'],3), 'Mrays/s', round(j['Mrays_per_s']), 'rays', j['rays'])" || exit 1
  done
done

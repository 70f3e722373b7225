#!/bin/bash
# N=512 frame and the N=4096 slice for the product library and each named variant:  scripts/ab_variants.sh VARIANT...
for name in product "$@"; do
  lib=""; [ "$name" != product ] && lib=raytracer_amd/csrc/build/variants/$name/libraytracer_amd.so
  for args in "--steps 2" "--steps 1 --spheres 4096 --height 68"; do
    RAYTRACER_AMD_LIB=$lib timeout -k 10 300 python3 -u bench.py --no-configs --no-cpu-baseline $args 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', j['config']['objects'], round(j['ms_per_step'],1), round(j['value'],1), j['roofline'].get('kernel'))" || exit 1
  done
done

#!/bin/bash
# A/B of a variant library against the product one on the two wide-kernel workloads:  scripts/ab_wide.sh VARIANT
v=raytracer_amd/csrc/build/variants/$1/libraytracer_amd.so
for lib in "" "$v"; do
  for args in "--steps 2" "--steps 1 --spheres 4096 --height 68"; do
    RAYTRACER_AMD_LIB=$lib timeout -k 10 300 python3 bench.py --no-configs --no-cpu-baseline $args 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-product}'.split('/')[-2] if '$lib' else 'product', j['config']['objects'], round(j['ms_per_step'],1), round(j['value'],1), j['config']['rays_per_frame'])"
  done
done

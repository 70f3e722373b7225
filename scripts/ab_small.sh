#!/bin/bash
# A/B of a variant library against the product one on C3 / C4:  scripts/ab_small.sh VARIANT
v=raytracer_amd/csrc/build/variants/$1/libraytracer_amd.so
for lib in "" "$v"; do
  RAYTRACER_AMD_LIB=$lib timeout -k 10 120 python3 scripts/run_configs.py 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l)
    if j['mode']=='path': print('${lib:-product}'.split('/')[-2] if '$lib' else 'product', j['config'], round(j['frame_ms'],2), j['rays'])"
done

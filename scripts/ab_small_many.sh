#!/bin/bash
# C3 / C4 frame times of several variant libraries against the product one:  scripts/ab_small_many.sh VARIANT...
for name in product "$@"; do
  lib=""; [ "$name" != product ] && lib=raytracer_amd/csrc/build/variants/$name/libraytracer_amd.so
  RAYTRACER_AMD_LIB=$lib timeout -k 10 120 python3 scripts/run_configs.py 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l)
    if j['mode']=='path': print('$name', j['config'], round(j['frame_ms'],2), j['rays'], j.get('frame_sha256'))" || exit 1
done

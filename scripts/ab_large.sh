#!/bin/bash
# Large-world frame times (N=512 1080p AA=4; N=4096 480x270 AA=4 = 1/16 of the pixels of the 1080p frame) of variant libraries against the product one:
#   scripts/ab_large.sh VARIANT...
for name in product "$@"; do
  lib=""; [ "$name" != product ] && lib=$PWD/raytracer_amd/csrc/build/variants/$name/libraytracer_amd.so
  RAYTRACER_AMD_LIB=$lib timeout -k 10 300 python3 scripts/run_configs.py --skip-samples --synth 512 1920 1080 4 --synth 4096 480 270 4 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l)
    print('$name', j['config'], j['size'], round(j['frame_ms'],1), 'ms', round(j['Mrays_per_s']), 'Mrays/s', j['rays'], j.get('frame_sha256'))" || exit 1
done

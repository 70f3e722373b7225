#!/bin/bash
# tuning sweep of the tile kernel: S tiles per wave x W waves (x 4 or 5 waves per SIMD) on N=512 and the N=4096 slice
run() { # lib S W args...
  lib=$1; s=$2; w=$3; shift 3
  RAYTRACER_AMD_LIB=$lib RT_TILE_S=$s RT_TILE_W=$w timeout -k 10 300 python3 -u bench.py --no-configs --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:+occ5 }S=$s W=$w', j['config']['objects'], round(j['ms_per_step'],1), round(j['value'],1), j['roofline'].get('kernel'))" || exit 1
}
for sw in "8 1" "8 2" "16 1" "16 2" "4 2" "4 1"; do run "" $sw --steps 1 --spheres 4096 --height 68; done
for sw in "8 1" "8 2" "4 1"; do run "" $sw --steps 2 --spheres 1024 --height 270; done
for sw in "8 1" "4 1"; do run "" $sw --steps 2 --spheres 128 --height 540; done

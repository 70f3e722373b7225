#!/bin/bash
# Builds an instrumented / re-parameterised copy of the library next to the product one:
#   scripts/build_variant.sh NAME -DRT_BVH_STATS ...   ->  raytracer_amd/csrc/build/variants/NAME/libraytracer_amd.so
# Use it with RAYTRACER_AMD_LIB=<that path> (raytracer_amd/lib.py).  Only rt_api.hip and rt_scene.hip are recompiled (dispatch and
# scene tables), unless ALL=1 is set (then the wide-kernel translation units too, e.g. for -DRT_PROFILE_PHASES or -DRT_TILE_BF16=0).
set -e
cd "$(dirname "$0")/../raytracer_amd/csrc"
name=$1; shift
make -s ../libraytracer_amd.so
out=build/variants/$name
mkdir -p "$out"
for f in rt_api rt_scene; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -fvisibility=hidden "$@" -c $f.hip -o "$out/$f.o" &
done
wait
wide="build/rt_wide_iso.o build/rt_wide_gen.o build/rt_wide_tile.o"
if [ -n "$ALL" ]; then
  for f in rt_wide_iso rt_wide_gen rt_wide_tile; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -fvisibility=hidden "$@" -c $f.hip -o "$out/$f.o" &
  done
  wait
  wide="$out/rt_wide_iso.o $out/rt_wide_gen.o $out/rt_wide_tile.o"
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$out/rt_api.o" "$out/rt_scene.o" build/rt_multi.o build/rt_tonemap_api.o $wide -o "$out/libraytracer_amd.so"
echo "$PWD/$out/libraytracer_amd.so"

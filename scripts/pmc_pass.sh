#!/bin/bash
# One rocprofv3 counter pass of C3 / C4 (scripts/run_configs.py) for one library variant, printed per ray:
#   scripts/pmc_pass.sh TAG VARIANT "COUNTER COUNTER ..."      (VARIANT "product" = the in-tree library; at most 8 SQ counters)
tag=$1; name=$2; counters=$3
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/$tag
lib=""; [ "$name" != product ] && lib=$PWD/raytracer_amd/csrc/build/variants/$name/libraytracer_amd.so
export RAYTRACER_AMD_LIB=$lib
timeout -s KILL 120 rocprofv3 --pmc $counters --output-format csv -d gpurun_out/$tag/$name -- python3 scripts/run_configs.py > gpurun_out/$tag/$name.log 2>&1 || { echo "$name: rocprofv3 failed"; tail -n 5 gpurun_out/$tag/$name.log; exit 1; }
python3 scripts/pmc_summary.py gpurun_out/$tag/$name.json gpurun_out/$tag/$name > /dev/null
python3 - "$name" gpurun_out/$tag/$name.json gpurun_out/$tag/$name.log <<'PY'
import json, sys
name, js, log = sys.argv[1:4]
rays = {}
for l in open(log):
    if l.startswith('{'):
        j = json.loads(l)
        if j['mode'] == 'path':
            rays[j['config'][:2]] = (j['rays'], j['frame_ms'])
for k, e in json.load(open(js)).items():
    if 'k_render_path' not in k:
        continue
    cfg = 'C4' if 'true, true>' in k and 'wave' in k else 'C3'
    n = 4 * rays.get(cfg, (1, 0))[0]  # warm-up + 3 timed launches
    print(f"{name} {cfg} {rays.get(cfg,(0,0))[1]:.2f} ms per ray: " + "  ".join(f"{c.replace('SQ_','')}={v/n:.3f}" for c, v in sorted(e['counters'].items())))
PY

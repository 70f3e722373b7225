#!/bin/bash
# One SQ counter pass (8 counters) of C3 / C4 per library variant:  scripts/pmc_variants.sh TAG VARIANT...   ("product" = the in-tree library)
# Prints per kernel: VALU / SALU / LDS wave-instructions per ray, waiting share, LDS conflict ratio.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/$tag
for name in "$@"; do
  lib=""; [ "$name" != product ] && lib=$PWD/raytracer_amd/csrc/build/variants/$name/libraytracer_amd.so
  export RAYTRACER_AMD_LIB=$lib
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU \
     --output-format csv -d gpurun_out/$tag/$name -- python3 scripts/run_configs.py > gpurun_out/$tag/$name.log 2>&1 || { echo "$name: rocprofv3 failed"; tail -n 5 gpurun_out/$tag/$name.log; exit 1; }
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
res = json.load(open(js))
for k, e in res.items():
    if 'k_render_path' not in k:
        continue
    c = e['counters']
    cfg = 'C4' if 'true, true>' in k and 'wave' in k else 'C3'
    n = 4 * rays.get(cfg, (1, 0))[0]  # warm-up + 3 timed launches
    print(f"{name:14s} {cfg} {rays.get(cfg,(0,0))[1]:7.2f} ms  VALU/ray {c['SQ_INSTS_VALU']/n:6.2f}  SALU/ray {c['SQ_INSTS_SALU']/n:5.2f}  LDS/ray {c['SQ_INSTS_LDS']/n:5.3f}  "
          f"wait {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.3f}  conflict {c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1):.3f}  LDScyc/ray {c['SQ_LDS_IDX_ACTIVE']/n:.2f} vgpr {e['vgprs']} scratch {e['scratch_bytes']}")
PY
done

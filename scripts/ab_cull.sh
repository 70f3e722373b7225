#!/bin/bash
# The cluster-culled tile kernel against the same kernel scanning all tiles: same frames (sha), times.  Sizes: "N W H AA" ...
set -e
for cfg in "${@:-4096 480 136 4}"; do
  for flag in "" "--scan-all"; do
    timeout -k 10 300 python3 scripts/run_configs.py --skip-samples $flag --synth $cfg 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l)
    print(j['config'], j['size'], j['kernel'], round(j['frame_ms'],1), 'ms', round(j['Mrays_per_s']), 'Mrays/s', j['rays'], j['frame_sha256'])"
  done
done

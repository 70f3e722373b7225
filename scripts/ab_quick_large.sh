#!/bin/bash
# Quick check of the large-world kernels of the product library: N=512 whole frame, N=4096 480x136 AA=4 (1/32 of the 1080p frame's pixels)
timeout -k 10 300 python3 scripts/run_configs.py --skip-samples --synth 512 1920 1080 4 --synth 4096 480 136 4 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l)
    print(j['config'], j['size'], j['kernel'], round(j['frame_ms'],1), 'ms', round(j['Mrays_per_s']), 'Mrays/s', j['rays'], j['frame_sha256'])"

#!/bin/bash
# the MFMA filter (default) against the VALU filter (--valu-only = RT_FLAG_VALU_ONLY) on the two wide-kernel workloads, then the tests that look at them
for v in "" --valu-only; do
  for args in "--steps 2" "--steps 1 --spheres 4096 --height 68"; do
    timeout -k 10 300 python3 -u bench.py --no-configs --no-cpu-baseline $v $args 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', j['config']['objects'], round(j['ms_per_step'],1), round(j['value'],1), j['roofline']['frac'], j['roofline'].get('kernel'))" || exit 1
  done
done


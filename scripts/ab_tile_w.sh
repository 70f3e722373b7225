for w in 1 2; do
RT_TILE_W=$w timeout -k 10 300 python3 -u bench.py --no-configs --no-cpu-baseline --steps 2 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('W=$w', round(j['ms_per_step'],1), round(j['value'],1), j['roofline'].get('kernel'))"
done

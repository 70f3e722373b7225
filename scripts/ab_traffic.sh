set -e
for args in "--steps 2" "--steps 1 --spheres 4096 --height 68"; do
  timeout -k 10 300 python3 -u bench.py --no-configs --no-cpu-baseline $args 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['config']['objects'], round(j['ms_per_step'],1), round(j['value'],1), j['roofline']['frac'])"
done
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for pass in FETCH_SIZE WRITE_SIZE; do
timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/tcc_$pass -- python3 bench.py --steps 1 --warmup 0 --no-configs --no-cpu-baseline > gpurun_out/tcc_$pass.log 2>&1
done
python3 scripts/pmc_summary.py gpurun_out/tcc.json gpurun_out/tcc_FETCH_SIZE gpurun_out/tcc_WRITE_SIZE | tail -30

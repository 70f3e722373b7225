#!/bin/bash
# Rehearsal of bench.py's N>1 path on ONE GPU: 1, 2 and 3 ranks sharing device 0 (gloo, host-staged gather).
# bench.py --gpus N starts its own ranks (no launcher around it).  The frame checksum must be the same for every rank count.
set -e
args="--spheres 128 --width 640 --height 360 --aa 2 --steps 1 --warmup 1 --no-cpu-baseline --no-configs --backend gloo"
for n in 1 2 3; do
  timeout -k 10 180 python3 bench.py --gpus $n $args > gpurun_out/rehearse_$n.json
done
python3 - <<'PY'
import json
rows = [json.loads(open(f"gpurun_out/rehearse_{n}.json").read().strip().splitlines()[-1]) for n in (1, 2, 3)]
for r in rows:
    print(r["n_gpus"], r["frame_sha256"], r["config"]["rays_per_frame"], round(r["ms_per_step"], 1))
assert len({r["frame_sha256"] for r in rows}) == 1 and len({r["config"]["rays_per_frame"] for r in rows}) == 1
print("OK: identical frame and ray count for 1, 2, 3 ranks")
PY

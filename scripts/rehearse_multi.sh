#!/bin/bash
# Rehearsal of bench.py's N>1 path on ONE GPU: 1, 2 and 3 ranks sharing device 0 (gloo, host-staged gather).
# bench.py --gpus N starts its own ranks (no launcher around it).  A frame of 6 shares is walked through completely by every
# rank count (warm-up step + timed steps = 6 / N steps); the assembled frame must be the same file for every rank count.
set -e
mkdir -p gpurun_out
args="--spheres 128 --width 640 --height 360 --aa 2 --shares 6 --warmup 1 --no-cpu-baseline --no-configs --backend gloo"
for n in 1 2 3; do
  timeout -k 10 180 python3 bench.py --gpus $n --steps $((6 / n - 1)) $args --save-pfm gpurun_out/rehearse_$n.pfm > gpurun_out/rehearse_$n.json
done
python3 - <<'PY'
import hashlib, json
rows = [json.loads(open(f"gpurun_out/rehearse_{n}.json").read().strip().splitlines()[-1]) for n in (1, 2, 3)]
frames = [hashlib.sha256(open(f"gpurun_out/rehearse_{n}.pfm", "rb").read()).hexdigest() for n in (1, 2, 3)]
for r, f in zip(rows, frames):
    print(r["n_gpus"], f[:16], r["share0_sha256"][:16], round(r["ms_per_step"], 1), r["kernel_ms_per_step"], r["gather_ms_per_step"])
assert len(set(frames)) == 1 and len({r["share0_sha256"] for r in rows}) == 1
print("OK: identical frame for 1, 2, 3 ranks")
PY

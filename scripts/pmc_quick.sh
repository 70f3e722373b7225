#!/bin/bash
# One SQ counter pass over a bench.py configuration, summarised per kernel.   scripts/pmc_quick.sh TAG <bench.py arguments>
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES \
  --output-format csv -d gpurun_out/pmcq_${tag} -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs "$@" > gpurun_out/pmcq_${tag}.log 2>&1
python3 scripts/pmc_summary.py gpurun_out/pmcq_${tag}.json gpurun_out/pmcq_${tag} > /dev/null
python3 - <<PY
import json
j = json.load(open("gpurun_out/pmcq_${tag}.json"))
line = [l for l in open("gpurun_out/pmcq_${tag}.log") if l.startswith("{")][-1]
b = json.loads(line)
rays = b["roofline"]["rays_per_launch"]
for k, e in j.items():
    if "resolve" in k: continue
    c = e["counters"]; n = 2  # stats pass + 1 timed step
    print(k[:70], "vgprs", e["vgprs"], "scratch", e["scratch_bytes"], "lds", e["lds_bytes"])
    for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        print(f"  {name}: {c[name] / n / rays:.2f} wave-instr per ray = {c[name] / n / rays * 64:.0f} per 64 rays")
    print(f"  wait share {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.2f}; kernel ms {b['roofline']['kernel_ms']:.2f}")
PY

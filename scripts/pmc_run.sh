#!/bin/bash
# Counter passes for one bench configuration (each pass its own rocprofv3 run, counters only).
#   scripts/pmc_run.sh TAG -- <bench.py arguments>
set -e
tag=$1; shift; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS" \
            "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 400 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 bench.py "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "pass $i done"
done
python3 scripts/pmc_summary.py gpurun_out/pmc_${tag}.json gpurun_out/pmc_${tag}_1 gpurun_out/pmc_${tag}_2 gpurun_out/pmc_${tag}_3 gpurun_out/pmc_${tag}_4 > /dev/null
cat gpurun_out/pmc_${tag}.json

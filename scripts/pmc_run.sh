#!/bin/bash
# Counter passes for one bench configuration (each pass its own rocprofv3 run, counters only): SQ x8 | GRBM + SQ x7 | FETCH_SIZE |
# WRITE_SIZE | VALU instruction classes | activity, lane use, branches, LDS atomics | matrix-pipe busy cycles and MFMA counts.
#   scripts/pmc_run.sh TAG -- <bench.py arguments>
set -e
tag=$1; shift; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS" \
            "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" \
            "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA" \
            "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_CYCLES"; do
  i=$((i+1))
  timeout -s KILL 400 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 bench.py "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "pass $i done"
done
python3 scripts/pmc_summary.py gpurun_out/pmc_${tag}.json gpurun_out/pmc_${tag}_1 gpurun_out/pmc_${tag}_2 gpurun_out/pmc_${tag}_3 gpurun_out/pmc_${tag}_4 gpurun_out/pmc_${tag}_5 gpurun_out/pmc_${tag}_6 gpurun_out/pmc_${tag}_7 > /dev/null
cat gpurun_out/pmc_${tag}.json

#!/bin/bash
# SQ / TCP counter passes over the inference step (B=64, 1024^2, bf16): usage scripts/dbg/pmc_r04.sh TAG -> gpurun_out/r04_pmc_sq_TAG.txt
export TMPDIR=/tmp
TAG=${1:-v1}
INF="--steps 1 --warmup 1 --no_cpu_baseline --no_layer_events --no_other_dtype --no_train_leg --no_fp32_leg --no_pipelined_leg"
i=0
for set in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1)); rm -rf gpurun_out/pmc4_$i
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc4_$i -- python3 bench.py $INF > gpurun_out/pmc4_$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmc4_$i.log; }
done
python scripts/pmc_summary.py gpurun_out/pmc4_1 gpurun_out/pmc4_2 gpurun_out/pmc4_3 > gpurun_out/r04_pmc_sq_$TAG.txt 2>&1
rm -rf gpurun_out/pmc4_1 gpurun_out/pmc4_2 gpurun_out/pmc4_3
head -40 gpurun_out/r04_pmc_sq_$TAG.txt

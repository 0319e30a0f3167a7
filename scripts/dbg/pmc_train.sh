#!/bin/bash
# SQ counters of the training step's kernels (one pass per counter set): scripts/dbg/pmc_train.sh [size]
export TMPDIR=/tmp
S=${1:-1024}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU"; do
  i=$((i+1)); rm -rf gpurun_out/pmct_$i
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmct_$i -- python3 bench.py --mode train --train_size $S --steps 1 --warmup 1 --no_cpu_baseline > gpurun_out/pmct_$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmct_$i.log; }
done
python scripts/pmc_summary.py gpurun_out/pmct_1 gpurun_out/pmct_2 gpurun_out/pmct_3 > gpurun_out/pmc_train.txt 2>&1
rm -rf gpurun_out/pmct_1 gpurun_out/pmct_2 gpurun_out/pmct_3
grep -n "wgrad_bf16_kernel<3, 1" -A26 gpurun_out/pmc_train.txt | head -30

#!/bin/bash
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT" \
           "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_$i
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$i -- python3 bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_layer_events > gpurun_out/pmc_$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmc_$i.log; }
done
python scripts/pmc_summary.py gpurun_out/pmc_1 gpurun_out/pmc_2 gpurun_out/pmc_3 gpurun_out/pmc_4 > gpurun_out/r02_pmc_sq.txt 2>&1
head -60 gpurun_out/r02_pmc_sq.txt

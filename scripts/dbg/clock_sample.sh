#!/bin/bash
# Samples rocm-smi (shader clock, package power) while bench.py times one 16-bit storage type: usage scripts/dbg/clock_sample.sh bf16|fp16 OUT
# (the question: same kernels, same instruction counts -- why is the half-precision path ~4 % slower?  DESIGN.md section 2)
DT=$1
OUT=$2
python bench.py --dtype $DT --steps 300 --warmup 20 --no_cpu_baseline --no_other_dtype --no_train_leg --no_fp32_leg --no_pipelined_leg > $OUT.bench 2>&1 &
PID=$!
sleep 6   # import + warm-up
: > $OUT.smi
while kill -0 $PID 2>/dev/null; do
    /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" >> $OUT.smi
    sleep 0.2
done
wait $PID
grep -c sclk $OUT.smi

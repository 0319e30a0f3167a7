#!/bin/bash
# Regenerates the judged profile artefacts on the GPU box: usage scripts/profile_round.sh TAG   (writes gpurun_out/<files>_TAG;
# copy them into profiles/ afterwards).  AY_GIT_HEAD=<short sha> is recorded in the traffic json files next to the kernel-source hash.
set -e
export TMPDIR=/tmp
TAG=$1
R=${ROUND:-r03}
INF="--no_cpu_baseline --no_other_dtype --no_train_leg --no_fp32_leg --no_pipelined_leg --dtype ${AY_PROFILE_DTYPE:-bf16}"
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write
# ---- inference (the timed dtype of bench.py's default line: bf16; AY_PROFILE_DTYPE=fp16 profiles the other one)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --steps 5 --warmup 2 $INF > gpurun_out/bench_kt_$TAG.log 2>&1
cp gpurun_out/prof_kt/*/*_kernel_stats.csv gpurun_out/${R}_bench_b64_kernel_stats_$TAG.csv
python scripts/layer_times.py gpurun_out/prof_kt > gpurun_out/${R}_bench_b64_layer_times_$TAG.txt
echo "inference kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 1 --warmup 1 $INF > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 1 --warmup 1 $INF > /dev/null 2>&1
python scripts/hbm_traffic.py gpurun_out/prof_fetch gpurun_out/prof_write --json gpurun_out/traffic_$TAG.json > gpurun_out/${R}_hbm_traffic_$TAG.txt
tail -3 gpurun_out/${R}_hbm_traffic_$TAG.txt
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write
# ---- the fp32 parity path (B=8)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_other_dtype --no_train_leg --no_pipelined_leg > gpurun_out/bench_kt_fp32_$TAG.log 2>&1
cp gpurun_out/prof_kt/*/*_kernel_stats.csv gpurun_out/${R}_bench_fp32leg_kernel_stats_$TAG.csv
rm -rf gpurun_out/prof_kt
echo "fp32 leg kernel trace done"
# ---- training (configs[2]: B=32, 1024^2)
TRN="--mode train --train_size 1024 --no_cpu_baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py $TRN --steps 5 --warmup 2 > gpurun_out/bench_kt_train_$TAG.log 2>&1
cp gpurun_out/prof_kt/*/*_kernel_stats.csv gpurun_out/${R}_train_b32_s1024_kernel_stats_$TAG.csv
rm -rf gpurun_out/prof_kt
echo "training kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py $TRN --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 bench.py $TRN --steps 1 --warmup 1 > /dev/null 2>&1
python scripts/hbm_traffic_train.py gpurun_out/prof_fetch gpurun_out/prof_write --json gpurun_out/traffic_train_$TAG.json > gpurun_out/${R}_hbm_traffic_train_$TAG.txt
tail -3 gpurun_out/${R}_hbm_traffic_train_$TAG.txt
rm -rf gpurun_out/prof_fetch gpurun_out/prof_write

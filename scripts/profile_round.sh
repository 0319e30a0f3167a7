#!/bin/bash
# Regenerates the judged profile artefacts on the GPU box: usage scripts/profile_round.sh TAG   (writes gpurun_out/<files>_TAG;
# copy them into profiles/ afterwards).  AY_GIT_HEAD=<short sha> is recorded in traffic.json next to the kernel-source hash.
set -e
export TMPDIR=/tmp
TAG=$1
R=${ROUND:-r03}
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_other_dtype --no_train_leg --no_fp32_leg > gpurun_out/bench_kt_$TAG.log 2>&1
cp gpurun_out/prof_kt/*/*_kernel_stats.csv gpurun_out/${R}_bench_b64_kernel_stats_$TAG.csv
python scripts/layer_times.py gpurun_out/prof_kt > gpurun_out/${R}_bench_b64_layer_times_$TAG.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_other_dtype --no_train_leg --no_fp32_leg > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_other_dtype --no_train_leg --no_fp32_leg > /dev/null 2>&1
python scripts/hbm_traffic.py gpurun_out/prof_fetch gpurun_out/prof_write --json gpurun_out/traffic_$TAG.json > gpurun_out/${R}_hbm_traffic_$TAG.txt
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write
tail -4 gpurun_out/${R}_hbm_traffic_$TAG.txt

#!/bin/bash
# phase clock of the ring kernels (instrumented build ab/lib_clock.so): AY_DBG=8 normal, 12 = without output stores
L=amyloid_yolo_paper_amd/libamyloid_yolo_hip.so
mkdir -p ab; cp $L ab/lib_keep.so; cp ab/lib_clock.so $L
for d in 8 12; do
  AY_DBG=$d timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no_cpu_baseline 2> gpurun_out/phase_$d.log > /dev/null
  echo "== AY_DBG=$d"; grep "ay phase" gpurun_out/phase_$d.log | sort | uniq -c | sort -k2 | awk '{c[$0]++} END{for(k in c) print k}' | sort -k3 | head -60
done
cp ab/lib_keep.so $L

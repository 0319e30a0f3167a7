#!/bin/bash
# same-box A/B of library builds on the training step: scripts/dbg/ab_train.sh SIZE name1 name2 ...  (ab/lib_<name>.so)
L=amyloid_yolo_paper_amd/libamyloid_yolo_hip.so
S=$1; shift
cp $L ab/lib_keep.so
for rep in 1 2; do
  for v in "$@"; do
    cp ab/lib_$v.so $L
    timeout -k 10 300 python bench.py --mode train --train_size $S --steps 6 --warmup 2 --no_cpu_baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('$v', d['ms_per_step'], 'ms/step', d['value'], 'imgs/s  wgrad', r.get('avg_launch_ms'), 'ms', r.get('frac'), ' conv', r.get('conv_family',{}).get('frac'))"
  done
done
cp ab/lib_keep.so $L

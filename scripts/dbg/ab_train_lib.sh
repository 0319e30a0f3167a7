#!/bin/bash
# same-box A/B of library builds on the training step (B=32, 1024^2): scripts/dbg/ab_train_lib.sh nameA nameB ...  (ab/lib_<name>.so)
export TMPDIR=/tmp
L=amyloid_yolo_paper_amd/libamyloid_yolo_hip.so
cp $L ab/lib_keep.so
for rep in 1 2 3; do
  for v in "$@"; do
    cp ab/lib_$v.so $L
    r=$(timeout -k 10 300 python bench.py --mode train --train_size 1024 --steps 10 --warmup 2 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))")
    echo "$v -> $r"
  done
done
cp ab/lib_keep.so $L

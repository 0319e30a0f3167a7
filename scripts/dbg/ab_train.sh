#!/bin/bash
L=amyloid_yolo_paper_amd/libamyloid_yolo_hip.so
mkdir -p ab; cp $L ab/lib_keep.so
for rep in 1 2; do for v in "$@"; do cp ab/lib_$v.so $L
 r=$(timeout -k 10 300 python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
 echo "$v -> $r"; done; done
cp ab/lib_keep.so $L

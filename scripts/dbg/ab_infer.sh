#!/bin/bash
# same-box A/B of library builds on the inference step only: scripts/dbg/ab_infer.sh nameA nameB ...   (ab/lib_<name>.so, 3 rounds, interleaved)
export TMPDIR=/tmp
L=amyloid_yolo_paper_amd/libamyloid_yolo_hip.so
cp $L ab/lib_keep.so
for rep in 1 2 3; do
  for v in "$@"; do
    cp ab/lib_$v.so $L
    r=$(timeout -k 10 300 python bench.py --steps ${AB_STEPS:-20} --warmup 5 --no_cpu_baseline --no_other_dtype --no_train_leg --no_fp32_leg --no_pipelined_leg ${AB_FLAGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))")
    echo "$v -> $r"
  done
done
cp ab/lib_keep.so $L

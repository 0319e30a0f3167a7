#!/bin/bash
# same-box comparison of two source trees (each with its own built library): scripts/dbg/ab_trees.sh DIR_A DIR_B  -- the default bench
# line's inference and training legs, 3 rounds, interleaved
export TMPDIR=/tmp
for rep in 1 2 3; do
  for d in "$@"; do
    r=$(cd $d && timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_other_dtype --no_fp32_leg --no_pipelined_leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('infer', d['ms_per_step'], d['value'], d['roofline']['frac'], '| train', d['train']['ms_per_step'], d['train']['value'])")
    echo "$d -> $r"
  done
done

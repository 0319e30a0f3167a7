#!/bin/bash
# same-box A/B of bench.py flag sets on the inference step: scripts/dbg/ab_flags.sh "" "--serial_nms" ...   (3 rounds, interleaved)
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in "$@"; do
    r=$(timeout -k 10 300 python bench.py --steps ${AB_STEPS:-20} --warmup 5 --no_cpu_baseline --no_other_dtype --no_train_leg --no_fp32_leg --no_pipelined_leg $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))")
    echo "[$v] -> $r"
  done
done

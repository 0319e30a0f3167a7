#!/bin/bash
# same-box A/B of two environments on the inference step: scripts/dbg/ab_env.sh "VAR=0" "VAR=1"   (3 rounds, interleaved)
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in "$@"; do
    r=$(env $v timeout -k 10 300 python bench.py --steps ${AB_STEPS:-20} --warmup 5 --no_cpu_baseline --no_other_dtype --no_train_leg --no_fp32_leg --no_pipelined_leg ${AB_FLAGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))")
    echo "$v -> $r"
  done
done

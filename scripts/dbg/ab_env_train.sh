#!/bin/bash
# same-box A/B of two environments on the training step (B=32, 1024^2): scripts/dbg/ab_env_train.sh "VAR=0" "VAR=1"   (3 rounds, interleaved)
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in "$@"; do
    r=$(env $v timeout -k 10 300 python bench.py --mode train --train_size 1024 --steps 10 --warmup 2 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))")
    echo "$v -> $r"
  done
done

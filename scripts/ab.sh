#!/bin/bash
# usage: scripts/ab.sh "ENV=.." "ENV2=.." ...   runs bench twice per setting, prints ms/step
for rep in 1 2; do
  for setting in "$@"; do
    v=$(env $setting timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline $AB_EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
    echo "$setting $AB_EXTRA -> $v"
  done
done

#!/bin/bash
# same-box A/B of builds of the library: scripts/dbg/ab_lib.sh old new ...  (ab/lib_<name>.so; bench x2 each, then a kernel trace each)
set -e
export TMPDIR=/tmp
L=amyloid_yolo_paper_amd/libamyloid_yolo_hip.so
mkdir -p ab; cp $L ab/lib_keep.so
for rep in 1 2; do
  for v in "$@"; do
    cp ab/lib_$v.so $L
    r=$(timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
    echo "$v -> $r"
  done
done
for v in "$@"; do
  cp ab/lib_$v.so $L
  rm -rf gpurun_out/prof_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$v -- python3 bench.py --steps 5 --warmup 2 --no_cpu_baseline > /dev/null 2>&1
  echo "== $v"; python scripts/kernel_stats.py gpurun_out/prof_$v 7 12
done
cp ab/lib_keep.so $L

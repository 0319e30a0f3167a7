#!/bin/bash
# usage: ab_conv.sh lib1 lib2 ... : times a few convolution shapes with each build of the library (ab/lib_<name>.so)
L=amyloid_yolo_paper_amd/libamyloid_yolo_hip.so
mkdir -p ab; cp $L ab/lib_keep.so
for v in "$@"; do cp ab/lib_$v.so $L; echo "== $v"
  for shape in "256 512 3 1 64 64" "128 256 3 1 128 64" "512 1024 3 1 32 64" "256 512 3 1 64 64 res"; do timeout -k 10 100 python scripts/dbg/time_conv.py $shape; done
done
cp ab/lib_keep.so $L

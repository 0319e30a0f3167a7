#!/bin/bash
# usage: scripts/dbg/build_variant.sh NAME file.hip "extra flags"  -> ab/lib_NAME.so (the current objects with file.hip rebuilt)
set -e
cd "$(dirname "$0")/../.."
NAME=$1; SRC=$2; EXTRA=$3
C=amyloid_yolo_paper_amd/csrc
mkdir -p ab /tmp/abobj_$NAME
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-function $EXTRA -c $C/$SRC -o /tmp/abobj_$NAME/${SRC%.hip}.o
OBJS=$(ls $C/*.o | grep -v "/${SRC%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib_$NAME.so $OBJS /tmp/abobj_$NAME/${SRC%.hip}.o
echo ab/lib_$NAME.so

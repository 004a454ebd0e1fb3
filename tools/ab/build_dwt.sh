#!/bin/bash
# usage: bash tools/ab/build_dwt.sh NAME [extra hipcc flags...]  -> tools/ab/lib_NAME.so = the in-tree objects with dwtile.hip rebuilt under the flags
set -e
R=/root/repo; C=$R/mobilenet-yolo-pytorch_amd/csrc; N=$1; shift
mkdir -p /tmp/hz
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c $C/dwtile.hip -o /tmp/hz/dwtile_$N.o 2>&1 | grep -E "error" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/lib_$N.so /tmp/hz/dwtile_$N.o $(ls $C/_obj/*.o | grep -v dwtile.o)

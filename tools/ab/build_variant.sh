#!/bin/bash
# usage: bash tools/ab/build_variant.sh NAME [extra hipcc flags...]  -> tools/ab/lib_NAME.so = the in-tree objects with pwgemm.hip rebuilt under the flags
set -e
R=/root/repo; C=$R/mobilenet-yolo-pytorch_amd/csrc; N=$1; shift
mkdir -p /tmp/hz
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c $C/pwgemm.hip -o /tmp/hz/pwgemm_$N.o 2>&1 | grep -E "error" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/lib_$N.so /tmp/hz/pwgemm_$N.o $(ls $C/_obj/*.o | grep -v pwgemm.o)

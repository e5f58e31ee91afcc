#!/bin/bash
# usage: tools/build_variant.sh NAME [extra hipcc flags...]   -> primus-fhe_amd/variants/libpfhe_hip_NAME.so
# Tuning / ablation builds of the same library; selected at run time with PFHE_LIB_PATH.
set -e
cd "$(dirname "$0")/../primus-fhe_amd"
name=$1; shift
mkdir -p variants build_$name
for f in csrc/*.hip csrc/*.cpp; do
  o=build_$name/$(basename $f).o
  x=""; [[ $f == *.cpp ]] && x="-x hip"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed "$@" $x -c $f -o $o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libpfhe_hip_$name.so build_$name/*.o
rm -rf build_$name
echo built variants/libpfhe_hip_$name.so

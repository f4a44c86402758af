#!/bin/bash
# A/B build of focf_step.hip alone: scratch/build_step_variant.sh <name> <extra flags...> -> scratch/lib/libfairrec_hip_<name>.so
# (the other objects are the product build's: run `make -C recbole-fairrec_amd/csrc` first)
set -e
name=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
src=$root/recbole-fairrec_amd/csrc
mkdir -p $root/scratch/lib $src/build_$name
/opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$root/include -I$src -Wall -Wno-unused-function -fvisibility=hidden \
  -D__HIP_PLATFORM_AMD__ -mllvm -amdgpu-kernarg-preload-count=8 -c $src/focf_step.hip -o $src/build_$name/focf_step.o
objs=$(ls $src/build/*.o | grep -v focf_step.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/scratch/lib/libfairrec_hip_$name.so $objs $src/build_$name/focf_step.o
echo built $root/scratch/lib/libfairrec_hip_$name.so

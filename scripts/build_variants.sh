#!/bin/bash
# Kernel experiments: builds libdebwt_hip variants into build/variants/ (git-ignored, travels with gpurun).
#   scripts/build_variants.sh NAME "-DFLAG ..." [git-rev for the kernel sources]
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; rev=$3
tmp=$(mktemp -d)
cp debwt_amd/csrc/*.hip debwt_amd/csrc/*.h debwt_amd/csrc/*.cpp "$tmp"/
if [ -n "$rev" ]; then
  for f in radix_sort.hip radix_sort.h stage_kernels.h debwt_hip.hip common.h special_kernels.h verify_kernels.h; do
    git show "$rev:debwt_amd/csrc/$f" > "$tmp/$f"
  done
  mkdir -p "$tmp/inc"; git show "$rev:include/debwt_hip.h" > "$tmp/inc/debwt_hip.h"; cp include/debwt_synth.h "$tmp/inc/"
  inc="$tmp/inc"
else
  inc="$PWD/include"
fi
sed -i "s#\"../../include/#\"$inc/#" "$tmp"/*.hip "$tmp"/*.cpp
F="-O3 -std=c++17 -fPIC -Wno-unused-function --offload-arch=gfx950 $flags"
mkdir -p build/variants
( cd "$tmp"
  /opt/rocm/bin/hipcc $F -c radix_sort.hip -o radix_sort.o &
  /opt/rocm/bin/hipcc $F -c debwt_hip.hip -o debwt_hip.o &
  /opt/rocm/bin/hipcc $F -c multi_host.cpp -o multi_host.o &
  /opt/rocm/bin/hipcc $F -x c++ -c special_host.cpp -o special_host.o &
  /opt/rocm/bin/hipcc $F -x c++ -c fasta_host.cpp -o fasta_host.o &
  /opt/rocm/bin/hipcc $F -x c++ -c gz_parallel.cpp -o gz_parallel.o &
  /opt/rocm/bin/hipcc $F -x c++ -c synth_host.cpp -o synth_host.o & wait )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libdebwt_$name.so "$tmp"/radix_sort.o "$tmp"/debwt_hip.o \
  "$tmp"/special_host.o "$tmp"/fasta_host.o "$tmp"/gz_parallel.o "$tmp"/synth_host.o "$tmp"/multi_host.o -lz -lpthread
rm -rf "$tmp"
echo built build/variants/libdebwt_$name.so

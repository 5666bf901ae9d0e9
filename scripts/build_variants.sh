#!/bin/bash
# Kernel experiments: builds libdebwt_hip variants into build/variants/ (git-ignored, travels with gpurun).
#   scripts/build_variants.sh NAME "-DFLAG ..." [git-rev for radix_sort.hip]
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; rev=$3
tmp=$(mktemp -d)
cp debwt_amd/csrc/*.hip debwt_amd/csrc/*.h debwt_amd/csrc/*.cpp "$tmp"/
mkdir -p "$tmp/../../include_tmp"
if [ -n "$rev" ]; then git show "$rev:debwt_amd/csrc/radix_sort.hip" > "$tmp/radix_sort.hip"; git show "$rev:debwt_amd/csrc/radix_sort.h" > "$tmp/radix_sort.h"; fi
sed -i "s#\"../../include/debwt_hip.h\"#\"$PWD/include/debwt_hip.h\"#" "$tmp/debwt_hip.hip"
F="-O3 -std=c++17 -fPIC -Wno-unused-function --offload-arch=gfx950 $flags"
( cd "$tmp" && /opt/rocm/bin/hipcc $F -c radix_sort.hip -o radix_sort.o & 
  cd "$tmp" && /opt/rocm/bin/hipcc $F -c debwt_hip.hip -o debwt_hip.o &
  cd "$tmp" && /opt/rocm/bin/hipcc $F -x c++ -c special_host.cpp -o special_host.o &
  cd "$tmp" && /opt/rocm/bin/hipcc $F -x c++ -c fasta_host.cpp -o fasta_host.o & wait )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libdebwt_$name.so "$tmp"/radix_sort.o "$tmp"/debwt_hip.o "$tmp"/special_host.o "$tmp"/fasta_host.o -lz -lpthread
rm -rf "$tmp"
echo built build/variants/libdebwt_$name.so

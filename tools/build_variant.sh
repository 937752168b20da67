#!/bin/bash
# Builds an A/B variant of the product library that differs only in the flags given to wna16_gemm.hip (the int4 GEMM/GEMV kernels):
#   tools/build_variant.sh <name> [-DFLAG=..]...   ->  vllm_rs_amd/libvra_<name>.so   (other objects are reused from csrc/build)
set -e
cd "$(dirname "$0")/../vllm_rs_amd/csrc"
name=$1; shift
B=build_$name
mkdir -p $B
for o in build/*.o; do b=$(basename $o); case $b in wna16_gemm.o|*-hip-amdgcn-*|*-host-x86_64-*) ;; *) ln -sf ../$o $B/$b ;; esac; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -ffp-contract=off "$@" -save-temps=obj -c wna16_gemm.hip -o $B/wna16_gemm.o
[ -n "$NOLINT" ] || python3 ../../tools/check_mfma_overlap.py $B/wna16_gemm-hip-amdgcn-amd-amdhsa-gfx950.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvra_$name.so $(ls $B/*.o | grep -v -e -hip-amdgcn- -e -host-x86_64-) -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built libvra_$name.so

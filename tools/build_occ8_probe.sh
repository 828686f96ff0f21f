#!/bin/bash
# Experiment: rebuilds libanx with amdgpu_waves_per_eu(8,8) forced on every k_filter_score instance
# (the configuration of commit f28d0ff whose GPU suite "did not finish") into build/libanx_occ8.so.  Product sources untouched.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/analiticcl_amd $T/include $R/build
cp -r $R/analiticcl_amd/csrc $T/analiticcl_amd/csrc
cp $R/include/anx.h $T/include/
sed -i 's/__global__ __launch_bounds__(256) void k_filter_score(/__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_filter_score(/' $T/analiticcl_amd/csrc/kernels_score.hpp
make -C $T/analiticcl_amd/csrc -s OUT=$R/build/libanx_occ8.so
if [ "$1" = "asm" ]; then make -C $T/analiticcl_amd/csrc -s asm && cp $T/analiticcl_amd/csrc/engine.s $R/build/engine_occ8.s; fi
rm -rf $T
ls -la $R/build/libanx_occ8.so

#!/bin/bash
# Experiment helper: builds libanx from the working tree with extra compiler flags into build/libanx_<name>.so: build_flags.sh <name> "<flags>"
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/analiticcl_amd $T/include $R/build
cp -r $R/analiticcl_amd/csrc $T/analiticcl_amd/csrc
rm -rf $T/analiticcl_amd/csrc/obj
cp $R/include/anx.h $T/include/
# only the kernels differ between such builds: reuse the host objects of the tree when they are current
make -C $T/analiticcl_amd/csrc -s -j${J:-4} OUT=$R/build/libanx_$1.so CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wextra -ffp-contract=off -pthread $2"
rm -rf $T
ls -la $R/build/libanx_$1.so

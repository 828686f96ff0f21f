#!/bin/bash
# Experiment helper: builds libanx from the sources of a git revision into build/libanx_<name>.so (A/B against the working tree).
# usage: build_rev.sh <name> <revision> [extra CXXFLAGS]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $R/build
git -C $R archive $2 analiticcl_amd/csrc include | tar -x -C $T
make -C $T/analiticcl_amd/csrc -s -j8 OUT=$R/build/libanx_$1.so CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wextra -ffp-contract=off -pthread $3"
rm -rf $T
ls -la $R/build/libanx_$1.so

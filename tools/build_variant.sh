#!/bin/bash
# Experiment helper: builds libanx from a temporary copy of csrc with a sed expression applied to one file, into build/libanx_<name>.so.
# usage: build_variant.sh <name> <file in csrc> <sed expression>.  Product sources untouched.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/analiticcl_amd $T/include $R/build
cp -r $R/analiticcl_amd/csrc $T/analiticcl_amd/csrc
rm -rf $T/analiticcl_amd/csrc/obj
cp $R/include/anx.h $T/include/
sed -i "$3" $T/analiticcl_amd/csrc/$2
make -C $T/analiticcl_amd/csrc -s -j8 OUT=$R/build/libanx_$1.so
rm -rf $T
ls -la $R/build/libanx_$1.so

#!/bin/bash
# builds the host-only benchmark of the confusable rescoring (g++, the host sources + the device stubs of the sanitizer harness)
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${1:-/tmp/conf_bench}
S=$R/analiticcl_amd/csrc
g++ -std=c++17 -O3 ${CONF_BENCH_FLAGS:-} -pthread -I $R/include -o $OUT $S/host_model.cpp $S/capi.cpp $S/search.cpp $S/confusables.cpp $S/contextrules.cpp \
    $S/index_cache.cpp $R/tests/host_sanitize/stub_engine.cpp $R/tools/conf_bench/main.cpp && echo built $OUT

#!/bin/bash
# ThreadSanitizer over the multi-threaded staged expansion (host C++ only; no GPU, no Python).
set -e
cd "$(dirname "$0")/.."
H=tetrex_amd/csrc/host
mkdir -p /tmp/tetrex_tsan
g++ -O1 -g -std=c++20 -fsanitize=thread -fno-omit-frame-pointer -pthread -o /tmp/tetrex_tsan/tsan_staged \
    tests/native/tsan_staged.cpp $H/encoder.cpp $H/regex_front.cpp $H/kgraph.cpp $H/compiler.cpp $H/staged.cpp
TSAN_OPTIONS=halt_on_error=1 /tmp/tetrex_tsan/tsan_staged

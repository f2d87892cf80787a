#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (C++ host front-end and the oracle); GPU sanitizers
# are not available on this pool.  Builds instrumented copies under /tmp and runs the CPU tests on them.
set -e
cd "$(dirname "$0")/.."
H=tetrex_amd/csrc/host
mkdir -p /tmp/tetrex_asan
g++ -O1 -g -std=c++20 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -shared -pthread \
    -o /tmp/tetrex_asan/libtetrex_host.so $H/encoder.cpp $H/regex_front.cpp $H/kgraph.cpp $H/compiler.cpp $H/staged.cpp $H/index_file.cpp $H/matcher.cpp $H/host_capi.cpp
make -C oracle liboracle_asan.so > /dev/null
# python does not link libstdc++, so it must be preloaded next to libasan for __cxa_throw interception
env LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" ASAN_OPTIONS=detect_leaks=0 \
    TETREX_HOST_LIB=/tmp/tetrex_asan/libtetrex_host.so TETREX_ORACLE_LIB=$PWD/oracle/liboracle_asan.so \
    python -m pytest tests/test_index_file.py tests/test_oracle_fixture.py tests/test_host_frontend.py tests/test_host_staged.py \
    tests/test_host_gaps.py tests/test_host_dense.py tests/test_host_shards.py tests/test_host_matcher.py tests/test_fuzz_parity.py -q -p no:cacheprovider -p no:faulthandler
# the native driver under ASan with leak detection on (LSan cannot be used under Python)
g++ -O1 -g -std=c++20 -fsanitize=address,undefined -fno-omit-frame-pointer -pthread -o /tmp/tetrex_asan/asan_staged \
    tests/native/tsan_staged.cpp $H/encoder.cpp $H/regex_front.cpp $H/kgraph.cpp $H/compiler.cpp $H/staged.cpp
/tmp/tetrex_asan/asan_staged

#!/bin/bash
# Timing experiment (WRONG masks): how much of the dense / sparse kernels' time is hashing?  Builds libtxq.so with a
# three-multiply stand-in for hash_row (TXQ_EXPERIMENTS + TXQ_CHEAP_HASH) in the box's scratch copy and times the 1000-motif
# k = 4 batch and the k = 6 batch with both libraries.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3b; mkdir -p $O
run() {
  REPS=8 timeout -k 10 120 python3 tools/e2e_profile.py 2>/dev/null | grep "^rep" | awk '{print $3}' | sort -n | head -4 | tr '\n' ' '; echo
  K6_NO_CHECK=1 timeout -k 10 200 python3 tools/k6_profile.py 2>/dev/null | grep -o '"seconds": [0-9.]*'
}
echo "product hash"; run
rm -f tetrex_amd/csrc/*.o
make -j16 EXPERIMENTS=1 EXPERIMENT_FLAGS=-DTXQ_CHEAP_HASH tetrex_amd/libtxq.so > $O/cheap_build.log 2>&1 || { tail $O/cheap_build.log; exit 1; }
echo "cheap hash (wrong masks)"; run

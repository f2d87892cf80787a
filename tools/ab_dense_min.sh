#!/bin/bash
# With the batches bound by the host's expansion: does it pay to hand shorter state lists to the device as blocks?
# TETREX_DENSE_MIN (smallest list that becomes a block, default 32) x TETREX_DENSE_SPARSE_BELOW (largest shape enumerated again, 16).
cd "$GRAFT_REPO_ROOT" || exit 1
for kn in "TETREX_DENSE_MIN=32 TETREX_DENSE_SPARSE_BELOW=16" "TETREX_DENSE_MIN=16 TETREX_DENSE_SPARSE_BELOW=8" "TETREX_DENSE_MIN=8 TETREX_DENSE_SPARSE_BELOW=4" "TETREX_DENSE_MIN=4 TETREX_DENSE_SPARSE_BELOW=2" "TETREX_DENSE_MIN=64 TETREX_DENSE_SPARSE_BELOW=32"; do
  echo "$kn"
  env $kn REPS=10 timeout -k 10 120 python3 tools/e2e_profile.py 2>/dev/null | grep "^rep" | sort -n -k3 | head -3 | cut -c1-220
  env $kn REPS=5 timeout -k 10 120 python3 tools/e2e_profile.py 10000 2>/dev/null | grep "^rep" | sort -n -k3 | head -2 | cut -c1-220
done

#!/bin/bash
# A/B of the host-side thresholds of the dense DP steps on the end-to-end batch (run on the GPU box).
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "128 24" "32 16" "48 16" "24 12" "32 24" "64 32"; do
  set -- $cfg
  echo "min_states=$1 sparse_below=$2"
  TXQ_TRACE=1 TETREX_DENSE_MIN=$1 TETREX_DENSE_SPARSE_BELOW=$2 timeout -k 10 120 python tools/e2e_profile.py 2>&1 | grep -E "^rep|session: 1000" | tail -2
done

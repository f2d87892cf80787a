#!/bin/bash
# A/B of the dense kernel's launch geometry on the end-to-end batch (run on the GPU box).
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "4 2" "4 1" "2 2" "1 2" "8 1" "8 2"; do
  set -- $cfg
  echo "slices=$1 tile_rounds=$2"
  TXQ_DENSE_SLICES=$1 TXQ_DENSE_TILE_ROUNDS=$2 timeout -k 10 120 python tools/e2e_profile.py 2>&1 | grep "^rep" | tail -2
done

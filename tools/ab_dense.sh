#!/bin/bash
# A/B of the dense kernel's launch geometry on the end-to-end batch (run on the GPU box).
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "2 2 3" "2 2 6" "2 1 6" "4 2 6" "1 2 6" "2 2 2" "2 4 3" "2 4 6"; do
  set -- $cfg
  echo "slices=$1 tile_rounds=$2 unroll=$3"
  TXQ_DENSE_SLICES=$1 TXQ_DENSE_TILE_ROUNDS=$2 TXQ_DENSE_UNROLL=$3 timeout -k 10 120 python tools/e2e_profile.py 2>&1 | grep "^rep" | tail -2 | cut -c1-20,150-230
done

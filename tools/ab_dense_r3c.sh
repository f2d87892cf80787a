#!/bin/bash
# Tile shapes of dense_kernel once more, on the final configuration (thresholds 8 / 4, table of all k-mers' masks): the
# 1000-motif batch is one wave of 65 levels now.  Best five of ten 1000-motif runs, best two of four 10000-motif runs, ms.
cd "$GRAFT_REPO_ROOT" || exit 1
for kn in "TXQ_DENSE_TILE_ROUNDS=2 TXQ_DENSE_SLICES=2" "TXQ_DENSE_TILE_ROUNDS=4 TXQ_DENSE_SLICES=2" "TXQ_DENSE_TILE_ROUNDS=8 TXQ_DENSE_SLICES=2" "TXQ_DENSE_TILE_ROUNDS=16 TXQ_DENSE_SLICES=2" "TXQ_DENSE_TILE_ROUNDS=32 TXQ_DENSE_SLICES=2" "TXQ_DENSE_TILE_ROUNDS=4 TXQ_DENSE_SLICES=1" "TXQ_DENSE_TILE_ROUNDS=8 TXQ_DENSE_SLICES=1" "TXQ_DENSE_TILE_ROUNDS=8 TXQ_DENSE_SLICES=2 TXQ_DENSE_UNROLL=5" "TXQ_DENSE_TILE_ROUNDS=8 TXQ_DENSE_SLICES=2 TXQ_KMER_TABLE_MB=0"; do
  echo "$kn"
  env $kn REPS=10 timeout -k 10 120 python3 tools/e2e_profile.py 2>/dev/null | grep "^rep" | awk '{print $3}' | sort -n | head -5 | tr '\n' ' '
  env $kn REPS=4 timeout -k 10 120 python3 tools/e2e_profile.py 10000 2>/dev/null | grep "^rep" | awk '{print $3}' | sort -n | head -2 | tr '\n' ' '
  echo
done

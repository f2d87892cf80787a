#!/bin/bash
# A/B of the dense step's tile shape on the 1000-motif end-to-end batch (tools/e2e_profile.py): slices x tile rounds x unroll.
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo "== $*"; env "$@" REPS=7 timeout -k 10 120 python tools/e2e_profile.py 1000 2>&1 | grep "^rep" | tail -4 | cut -c1-16; }
run A=default
for sl in 1 2 4; do for tr in 1 2 4; do for ua in 2 3 6; do run TXQ_DENSE_SLICES=$sl TXQ_DENSE_TILE_ROUNDS=$tr TXQ_DENSE_UNROLL=$ua; done; done; done
run TETREX_WAVE_OPS=98304
run TETREX_WAVE_OPS=131072
run TETREX_WAVE_OPS=262144
run TETREX_THREADS=12
run TETREX_THREADS=14

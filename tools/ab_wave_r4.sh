#!/bin/bash
# Round 4 (the host's expansion is twice as fast with fused residue classes): A/B of the first wave's size (TETREX_WAVE_OPS) and
# of how the later waves grow (TETREX_WAVE_GROWTH, percent of the ops emitted before) on the 1000- and 10000-motif batches:
# five best of ten / three best of six runs, ms.
cd "$GRAFT_REPO_ROOT" || exit 1
for kn in "TETREX_WAVE_OPS=40000 TETREX_WAVE_GROWTH=100" "TETREX_WAVE_OPS=40000 TETREX_WAVE_GROWTH=50" "TETREX_WAVE_OPS=40000 TETREX_WAVE_GROWTH=25" "TETREX_WAVE_OPS=40000 TETREX_WAVE_GROWTH=0" "TETREX_WAVE_OPS=80000 TETREX_WAVE_GROWTH=50" "TETREX_WAVE_OPS=80000 TETREX_WAVE_GROWTH=0" "TETREX_WAVE_OPS=60000 TETREX_WAVE_GROWTH=33" "TETREX_WAVE_OPS=120000 TETREX_WAVE_GROWTH=0" "TETREX_WAVE_OPS=20000 TETREX_WAVE_GROWTH=100"; do
  echo "$kn"
  env $kn REPS=10 timeout -k 10 120 python3 tools/e2e_profile.py 2>/dev/null | grep "^rep" | awk '{print $3}' | sort -n | head -5 | tr '\n' ' '
  env $kn REPS=6 timeout -k 10 120 python3 tools/e2e_profile.py 10000 2>/dev/null | grep "^rep" | awk '{print $3}' | sort -n | head -3 | tr '\n' ' '
  echo
done

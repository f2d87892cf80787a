#!/bin/bash
# A/B of the first wave's size (TETREX_WAVE_OPS) and of how the later waves grow (TETREX_WAVE_GROWTH: a wave is at least this
# share, in percent, of the ops emitted before it) on the 1000- and 10000-motif batches: five best of ten / three best of five runs, ms.
cd "$GRAFT_REPO_ROOT" || exit 1
for kn in "TETREX_WAVE_OPS=98304 TETREX_WAVE_GROWTH=50" "TETREX_WAVE_OPS=40000 TETREX_WAVE_GROWTH=50" "TETREX_WAVE_OPS=40000 TETREX_WAVE_GROWTH=100" "TETREX_WAVE_OPS=28000 TETREX_WAVE_GROWTH=100" "TETREX_WAVE_OPS=40000 TETREX_WAVE_GROWTH=200" "TETREX_WAVE_OPS=28000 TETREX_WAVE_GROWTH=200" "TETREX_WAVE_OPS=20000 TETREX_WAVE_GROWTH=200"; do
  echo "$kn"
  env $kn REPS=10 timeout -k 10 120 python3 tools/e2e_profile.py 2>/dev/null | grep "^rep" | awk '{print $3}' | sort -n | head -5 | tr '\n' ' '
  env $kn REPS=5 timeout -k 10 120 python3 tools/e2e_profile.py 10000 2>/dev/null | grep "^rep" | awk '{print $3}' | sort -n | head -3 | tr '\n' ' '
  echo
done

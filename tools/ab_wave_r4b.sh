#!/bin/bash
# second pass of tools/ab_wave_r4.sh around its best settings
cd "$GRAFT_REPO_ROOT" || exit 1
for kn in "TETREX_WAVE_OPS=60000 TETREX_WAVE_GROWTH=33" "TETREX_WAVE_OPS=60000 TETREX_WAVE_GROWTH=0" "TETREX_WAVE_OPS=60000 TETREX_WAVE_GROWTH=15" "TETREX_WAVE_OPS=60000 TETREX_WAVE_GROWTH=50" "TETREX_WAVE_OPS=70000 TETREX_WAVE_GROWTH=20" "TETREX_WAVE_OPS=50000 TETREX_WAVE_GROWTH=25" "TETREX_WAVE_OPS=80000 TETREX_WAVE_GROWTH=20" "TETREX_WAVE_OPS=64000 TETREX_WAVE_GROWTH=10" "TETREX_WAVE_OPS=40000 TETREX_WAVE_GROWTH=100"; do
  echo "$kn"
  env $kn REPS=10 timeout -k 10 120 python3 tools/e2e_profile.py 2>/dev/null | grep "^rep" | awk '{print $3}' | sort -n | head -5 | tr '\n' ' '
  env $kn REPS=6 timeout -k 10 120 python3 tools/e2e_profile.py 10000 2>/dev/null | grep "^rep" | awk '{print $3}' | sort -n | head -3 | tr '\n' ' '
  echo
done

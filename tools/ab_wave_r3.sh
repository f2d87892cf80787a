#!/bin/bash
# A/B of the first wave's size (TETREX_WAVE_OPS) on the 1000- and 10000-motif end-to-end batches (best runs, ms).  Round 3, after
# the dense thresholds went to 8 / 4: the 1000-motif batch is 78 k ops in all, so the 96 k of the first A/B
# (profiles/r3_wave_size_ab.txt, first part) had become a single wave.
cd "$GRAFT_REPO_ROOT" || exit 1
for w in 98304 49152 32768 24576 16384 12288 8192; do
  echo "TETREX_WAVE_OPS=$w"
  TETREX_WAVE_OPS=$w REPS=10 timeout -k 10 120 python3 tools/e2e_profile.py 2>/dev/null | grep "^rep" | awk '{print $3}' | sort -n | head -5 | tr '\n' ' '
  TETREX_WAVE_OPS=$w REPS=5 timeout -k 10 120 python3 tools/e2e_profile.py 10000 2>/dev/null | grep "^rep" | awk '{print $3}' | sort -n | head -3 | tr '\n' ' '
  echo
done

#!/bin/bash
# A/B of the first wave's size (TETREX_WAVE_OPS) on the 1000-motif end-to-end batch, after staging buffers stopped draining the
# device when they grow (a stage of the next wave now really runs beside the previous one).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3b; mkdir -p $O
for w in 196608 131072 98304 65536 49152 32768 16384; do
  echo "TETREX_WAVE_OPS=$w"
  TETREX_WAVE_OPS=$w REPS=10 timeout -k 10 120 python3 tools/e2e_profile.py 2>/dev/null | grep "^rep" | awk '{print $3}' | sort -n | head -5 | tr '\n' ' '
  echo
done
echo "10000 motifs"
for w in 196608 65536 32768; do
  echo "TETREX_WAVE_OPS=$w"
  TETREX_WAVE_OPS=$w REPS=5 timeout -k 10 120 python3 tools/e2e_profile.py 10000 2>/dev/null | grep "^rep" | awk '{print $3}' | sort -n | head -3 | tr '\n' ' '
  echo
done

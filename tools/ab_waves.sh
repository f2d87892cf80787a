#!/bin/bash
# A/B of the wave size of the staged expansion (TETREX_WAVE_OPS; 0 = every query begins in the first stage) on the
# bench's 1000-motif batch and on 10 000 motifs.  Run on the GPU box: tools/ab_waves.sh > gpurun_out/ab_waves.txt
for n in 1000 10000; do
  for w in 0 65536 98304 196608 393216; do
    echo "== motifs $n TETREX_WAVE_OPS=$w"
    TETREX_WAVE_OPS=$w timeout -k 10 120 python tools/e2e_profile.py $n 2>&1 | grep "^rep" | cut -c1-90 || exit 1
  done
done

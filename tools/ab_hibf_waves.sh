#!/bin/bash
# A/B on the GPU box: how many waves hibf_fused_kernel is launched with in all (TXQ_HIBF_WAVES; 0 = the default of 64 per CU),
# on the HIBF legs of bench.py alone.  Results: gpurun_out/r4b/hibf_waves_<n>.json (hibf_irregular.probe_user_order / probe_layout_order).
mkdir -p gpurun_out/r4b
for w in 0 14336 16128 17920 21504 28672; do
  TXQ_HIBF_WAVES=$w timeout -k 10 200 python bench.py --no-queries --no-k6 --no-hbm-leg --no-dna-batch --no-verification --no-big-batch --no-cpu --steps 5 --warmup 2 > gpurun_out/r4b/hibf_waves_$w.json 2> gpurun_out/r4b/hibf_waves_$w.err || exit 1
done

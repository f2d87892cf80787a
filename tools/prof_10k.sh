#!/bin/bash
# Device timeline of the 10000-motif batch with and without the table of all k-mers' masks.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3c; mkdir -p $O
for mb in 0 512; do
  TXQ_KMER_TABLE_MB=$mb REPS=3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_10k_$mb -o e2e -- python3 tools/e2e_profile.py 10000 > $O/e2e_10k_$mb.txt 2>&1
  python tools/trace_timeline.py $O/prof_10k_$mb/e2e_kernel_trace.csv > $O/e2e_10k_timeline_$mb.txt 2>&1
  rm -f $O/prof_10k_$mb/*.db $O/prof_10k_$mb/e2e_kernel_trace.csv
  grep "^rep" $O/e2e_10k_$mb.txt | tail -1 | cut -c1-300
  head -8 $O/e2e_10k_timeline_$mb.txt
done

#!/bin/bash
# The table of all k-mers' masks (TXQ_KMER_TABLE_MB) against row gathers, and predecessors per trip (TXQ_DENSE_UNROLL), on the
# 1000-motif k = 4 batch (best five of ten runs, ms) and on the regular-tree batch of bench.py's hibf_batch leg.
cd "$GRAFT_REPO_ROOT" || exit 1
for kn in "TXQ_KMER_TABLE_MB=0 TXQ_DENSE_UNROLL=2" "TXQ_KMER_TABLE_MB=0 TXQ_DENSE_UNROLL=3" "TXQ_KMER_TABLE_MB=0 TXQ_DENSE_UNROLL=5" "TXQ_KMER_TABLE_MB=512 TXQ_DENSE_UNROLL=2" "TXQ_KMER_TABLE_MB=512 TXQ_DENSE_UNROLL=3" "TXQ_KMER_TABLE_MB=512 TXQ_DENSE_UNROLL=5"; do
  echo "$kn"
  env $kn REPS=10 timeout -k 10 120 python3 tools/e2e_profile.py 2>/dev/null | grep "^rep" | awk '{print $3}' | sort -n | head -5 | tr '\n' ' '; echo
done

#!/bin/bash
# The best tile shapes of ab_dense_r3.sh on the bench's end-to-end legs (flat 1000 motifs, HIBF batch) and on 10 000 motifs.
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "2 2 3" "1 1 2" "2 4 3" "1 1 6" "1 2 2" "2 4 2"; do
  set -- $cfg
  echo "== slices=$1 tile_rounds=$2 unroll=$3"
  export TXQ_DENSE_SLICES=$1 TXQ_DENSE_TILE_ROUNDS=$2 TXQ_DENSE_UNROLL=$3
  timeout -k 10 200 python bench.py --steps 3 --no-hbm-leg --no-k6 --no-verification --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); e=d['end_to_end']; print('flat 1k %.2f ms  hibf 1k %.2f ms' % (e['batch']['seconds']*1e3, e['hibf_batch']['seconds']*1e3))"
  REPS=5 timeout -k 10 200 python tools/e2e_profile.py 10000 2>&1 | grep "^rep" | tail -3 | cut -c1-16 | tr '\n' ' '; echo
done

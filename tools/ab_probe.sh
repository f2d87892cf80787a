set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "2:" "1:" "4:" "8:" "2:nt" "4:nt"; do
  u=${cfg%%:*}; nt=${cfg##*:}
  for rows in 0 62500000; do
    if [ "$rows" = "0" ]; then extra=""; else extra="--rows $rows --kmer-bits 40 --per-bin 20000"; fi
    if [ -n "$nt" ]; then export TXQ_PROBE_NT=1; else unset TXQ_PROBE_NT; fi
    TXQ_PROBE_UNROLL=$u timeout -k 10 200 python bench.py --no-cpu --no-queries --steps 30 $extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('unroll=$u nt=$nt rows=$rows', round(d['value']/1e9,3), 'Gprobes/s frac', round(d['roofline']['frac'],3))"
  done
done

#!/bin/bash
# The thresholds of tools/ab_dense_min.sh on the other workloads: bench legs (k = 4 batch, HIBF batch, k = 6 batch), the CLI on
# Swissprot-shaped bins at k = 6, config 5 (Murphy, 65 536-bin tree), a ragged 65 536-bin tree.
cd "$GRAFT_REPO_ROOT" || exit 1
for kn in "TETREX_DENSE_MIN=32 TETREX_DENSE_SPARSE_BELOW=16" "TETREX_DENSE_MIN=8 TETREX_DENSE_SPARSE_BELOW=4" "TETREX_DENSE_MIN=16 TETREX_DENSE_SPARSE_BELOW=8"; do
  echo "== $kn"
  env $kn timeout -k 10 400 python3 bench.py --no-cpu --no-verification 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['end_to_end']
for k in ('batch','batch_10x','hibf_batch','k6_batch'):
    print(k, round(e[k]['seconds']*1e3,2),'ms', e[k].get('ops'), e[k].get('dense_ops'))
"
  env $kn timeout -k 10 300 python3 tests/perf_cli_swissprot_shape.py /tmp/sp 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cli k6 batch', d['motif_file_batch'])"
  env $kn timeout -k 10 300 python3 tests/perf_config5_queries.py 2>/dev/null | cut -c1-700
  env $kn timeout -k 10 300 python3 tests/perf_hibf_ragged.py 1048576 256 2>/dev/null | cut -c1-500
done

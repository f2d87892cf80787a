#!/bin/bash
# Timing experiments on the pushed steps of the k = 6 batch (sparse_units_kernel): what the batch costs with every destination
# atomic issued twice (1), every bitmap atomic twice (2), the row gathers of a second k-mer beside each unit's (4) — the masks
# stay the same (OR is idempotent, the extra rows are looked at but never change a product); needs the experiments library
# (build/exp/libtxq.so: txq_exec.hip compiled with -DTXQ_EXPERIMENTS), which this script puts in place of the product library
# IN THE GPU BOX'S SCRATCH COPY.  Usage on the GPU box: tools/ab_sparse_steps.sh
cd "$GRAFT_REPO_ROOT" || exit 1
cp build/exp/libtxq.so tetrex_amd/libtxq.so || exit 1
run() { echo "== $*"; for i in 1 2; do env K6_NO_CHECK=1 "$@" timeout -k 10 200 python tools/k6_profile.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f ms (execute_us %d)' % (d['seconds']*1e3, d['execute_us']))"; done; }
run TXQ_STEP_EXPERIMENT=0
run TXQ_STEP_EXPERIMENT=1


run TXQ_STEP_EXPERIMENT=4
run TXQ_STEP_EXPERIMENT=5

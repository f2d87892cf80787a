#!/bin/bash
# NOTE: the TXQ_HIBF_STORE=16/32/48 lines are timing experiments that compute wrong masks: they need a library built with `make clean && make EXPERIMENTS=1`.
# A/B of the HIBF descent kernels on S-HIBF-65536 (run on the GPU box): child-stationary vs k-mer-stationary, and its knobs.
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo "== $*"; env "$@" timeout -k 10 200 python tests/perf_hibf.py 1048576 300 1 65536 256 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e k-mers/s  %.2f ms  %.0f GB/s of rows' % (d['kmers_per_s'], d['seconds_per_batch']*1e3, d['mask_zero_fill_GBps']))"; }
run TXQ_HIBF_STATIONARY=0
for u in 1 2 3 4; do run TXQ_HIBF_UNROLL=$u; done
run TXQ_HIBF_LANE_HASH=1 TXQ_HIBF_UNROLL=1
run TXQ_HIBF_LANE_HASH=1 TXQ_HIBF_UNROLL=2
run TXQ_HIBF_TILE=512
run TXQ_HIBF_TILE=1024
run TXQ_HIBF_TILE=4096
run PERF_HIBF_NO_CHECK=1 TXQ_HIBF_STORE=16
run PERF_HIBF_NO_CHECK=1 TXQ_HIBF_STORE=32
run PERF_HIBF_NO_CHECK=1 TXQ_HIBF_STORE=48

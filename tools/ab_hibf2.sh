#!/bin/bash
# NOTE: the TXQ_HIBF_STORE=16/32/48 lines are timing experiments that compute wrong masks: they need a library built with `make clean && make EXPERIMENTS=1`.
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo "== $*"; env PERF_HIBF_NO_CHECK=1 "$@" timeout -k 10 200 python tests/perf_hibf.py 1048576 300 1 65536 256 2>&1 | tail -1 | sed 's/.*seconds_per_batch": \([0-9.e-]*\).*/\1 s/'; }
run TXQ_HIBF_STORE=0
run TXQ_HIBF_STORE=16
run TXQ_HIBF_STORE=32
run TXQ_HIBF_STORE=48
run TXQ_HIBF_STORE=16 TXQ_HIBF_UNROLL=1
run TXQ_HIBF_STORE=32 TXQ_HIBF_UNROLL=1
run TXQ_HIBF_STORE=32 TXQ_HIBF_UNROLL=4

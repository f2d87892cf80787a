#!/bin/bash
# Timing experiments on hibf_layout_level_kernel (general HIBFs in layout order; tests/perf_hibf_ragged.py, 2^20 k-mers, 65 536 user
# bins, tmax 256): the stage without row gathers (16), without stores (32), without gate loads (64: a level passes on a fixed bit of
# the k-mer instead) — WRONG rows: needs the experiments library (build/exp/libtxq.so), which this script puts in place of the
# product library IN THE GPU BOX'S SCRATCH COPY.  Then per-launch durations of the product kernel under rocprofv3.
cd "$GRAFT_REPO_ROOT" || exit 1
cp tetrex_amd/libtxq.so /tmp/libtxq_product.so
cp build/exp/libtxq.so tetrex_amd/libtxq.so || exit 1
run() { echo "== $*"; env PERF_HIBF_NO_CHECK=1 PERF_HIBF_PROBE_ONLY=1 "$@" timeout -k 10 200 python tests/perf_hibf_ragged.py 1048576 256 65536 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('layout-order stage %.2f ms' % (d['probe_layout_order']['seconds']*1e3))"; }
run TXQ_HIBF_STORE=0
run TXQ_HIBF_STORE=16
run TXQ_HIBF_STORE=32
run TXQ_HIBF_STORE=64
run TXQ_HIBF_STORE=80
run TXQ_HIBF_STORE=112
cp /tmp/libtxq_product.so tetrex_amd/libtxq.so
cd /tmp && export TMPDIR=/tmp && PERF_HIBF_NO_CHECK=1 PERF_HIBF_PROBE_ONLY=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ragged -o ragged -- python3 $GRAFT_REPO_ROOT/tests/perf_hibf_ragged.py 1048576 256 65536 > /dev/null 2>&1
python3 - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('/tmp/ragged/ragged_kernel_trace.csv')) if 'hibf_layout_level' in r['Kernel_Name']]
for r in rows[-6:]:
    print('level launch: grid %s wg %s  %.1f us' % (r.get('Grid_Size_X'), r.get('Workgroup_Size_X'), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY

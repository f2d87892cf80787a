#!/bin/bash
# Round-3 evidence run (on the GPU box): the default bench line, rocprofv3 kernel summaries of its legs, the k = 6 batch
# (tracked blocks) with its PMC traffic, the command line on Swissprot-shaped data, general HIBFs in layout order, the
# end-to-end batch at 1 k and 10 k motifs.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3_final
mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_probe -o probe -- python3 $GRAFT_REPO_ROOT/bench.py --no-queries --no-hibf --no-cpu > $O/bench_probe_legs.json 2> /dev/null)
(cd /tmp && K6_NO_CHECK=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k6 -o k6 -- python3 $GRAFT_REPO_ROOT/tools/k6_profile.py > $O/k6_under_rocprof.json 2> /dev/null)
python tools/trace_timeline.py $O/prof_k6/k6_kernel_trace.csv > $O/k6_timeline.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_e2e -o e2e -- python3 tools/e2e_profile.py > $O/e2e_1k.txt 2>&1
python tools/trace_timeline.py $O/prof_e2e/e2e_kernel_trace.csv > $O/e2e_timeline.txt 2>&1
rm -f $O/*/*_kernel_trace.csv $O/*/*.db
timeout -k 10 500 python tools/pmc_sparse.py $O/pmc_sparse_kernel.json > /dev/null 2> $O/pmc_sparse.err
TXQ_TRACE=1 timeout -k 10 300 python tools/e2e_profile.py 10000 > $O/e2e_10k.txt 2>&1
timeout -k 10 400 python tests/perf_cli_swissprot_shape.py /tmp/sp > $O/cli_swissprot_shape.json 2> /dev/null
timeout -k 10 300 python tests/perf_hibf_ragged.py 1048576 256 > $O/hibf_ragged_tmax256.json 2> /dev/null
timeout -k 10 300 python tests/perf_hibf_ragged.py 1048576 64 > $O/hibf_ragged_tmax64.json 2> /dev/null
timeout -k 10 300 python tests/perf_config5_queries.py > $O/config5.json 2> /dev/null
timeout -k 10 100 python tools/single_query_latency.py > $O/single_query_latency.txt 2> /dev/null
tail -c 500 $O/bench_default.json; echo; cat $O/k6_timeline.txt | head -12; cat $O/e2e_timeline.txt | head -8; grep "^rep" $O/e2e_1k.txt | tail -2; grep -E "^rep" $O/e2e_10k.txt | tail -2 | cut -c1-300; cut -c1-400 $O/config5.json; cat $O/single_query_latency.txt; python -c "
import json; d=json.load(open('$O/pmc_sparse_kernel.json')); print(json.dumps(d.get('roofline')))"
(cd /tmp && PERF_HIBF_NO_CHECK=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ragged -o ragged -- python3 $GRAFT_REPO_ROOT/tests/perf_hibf_ragged.py 1048576 256 > $O/hibf_ragged_under_rocprof.json 2> /dev/null)
rm -f $O/prof_ragged/*_kernel_trace.csv $O/prof_ragged/*.db
head -12 $O/prof_ragged/ragged_kernel_stats.csv | cut -c1-160

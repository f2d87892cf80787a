#!/bin/bash
# Round-4 evidence run (on the GPU box): the default bench line; rocprofv3 kernel summaries of the ISOLATED main leg (probe kernel on
# S-IBF-1024 only), of the k = 6 batch (sparse_units_kernel) with its timeline, of the general-HIBF probes (layout order by one wave
# per k-mer) and query batch; the L2-side counters of dense_kernel on the 1000-motif batch (what replaces the HBM roofline for a
# kernel whose rows come out of the caches); the command line's cold mask stage.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_final
mkdir -p $O
timeout -k 10 700 python bench.py > $O/bench_default.json 2> $O/bench_default.err
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_main -o main -- python3 $GRAFT_REPO_ROOT/bench.py --no-queries --no-hibf --no-cpu --no-hbm-leg > $O/bench_main_leg.json 2> /dev/null)
(cd /tmp && K6_NO_CHECK=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k6 -o k6 -- python3 $GRAFT_REPO_ROOT/tools/k6_profile.py > $O/k6_under_rocprof.json 2> /dev/null)
python tools/trace_timeline.py $O/prof_k6/k6_kernel_trace.csv > $O/k6_timeline.txt 2>&1
(cd /tmp && PERF_HIBF_NO_CHECK=1 PERF_HIBF_PROBE_ONLY=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ragged -o ragged -- python3 $GRAFT_REPO_ROOT/tests/perf_hibf_ragged.py 1048576 256 65536 > $O/hibf_ragged_probes_under_rocprof.json 2> /dev/null)
(cd /tmp && PERF_HIBF_QUERIES_ONLY=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_raggedq -o q -- python3 $GRAFT_REPO_ROOT/tests/perf_hibf_ragged.py 1048576 256 65536 > $O/hibf_ragged_queries_under_rocprof.json 2> /dev/null)
python tools/trace_timeline.py $O/prof_raggedq/q_kernel_trace.csv > $O/hibf_ragged_queries_timeline.txt 2>&1
rm -f $O/*/*_kernel_trace.csv $O/*/*.db
# dense_kernel (k = 4 batch, table of all k-mers' masks): L2 hits / misses and requests (one --pmc pass; program directly after --)
(cd /tmp && REPS=4 timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_dense_l2 -o l2 -- python3 $GRAFT_REPO_ROOT/tools/e2e_profile.py > $O/e2e_under_pmc.txt 2> /dev/null)
python tools/sum_pmc.py $O/pmc_dense_l2 $O/pmc_dense_l2.json > /dev/null
rm -rf $O/pmc_dense_l2
TETREX_TRACE=1 timeout -k 10 300 python tools/verify_leg.py 2 > $O/verify_leg.txt 2>&1
tail -c 400 $O/bench_default.json; echo; head -8 $O/k6_timeline.txt; head -8 $O/hibf_ragged_queries_timeline.txt; cat $O/pmc_dense_l2.json | head -30; tail -3 $O/verify_leg.txt | cut -c1-300

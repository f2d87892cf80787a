#!/bin/bash
# Round-2 evidence run (on the GPU box): the default bench line, rocprofv3 kernel summaries of the bench legs, the
# end-to-end batch at 1 k and 10 k motifs, the configs[2] HIBF batch under rocprof, the matcher's throughput.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_final
mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_probe -o probe -- python3 bench.py --no-queries --no-hibf --no-cpu > $O/bench_probe_legs.json 2> /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_all -o all -- python3 bench.py --no-cpu > $O/bench_all_legs_under_rocprof.json 2> /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python3 tests/perf_config3_queries.py > $O/config3.json 2> /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_e2e -o e2e -- python3 tools/e2e_profile.py > $O/e2e_1k.txt 2>&1
python tools/trace_timeline.py $O/prof_e2e/e2e_kernel_trace.csv > $O/e2e_timeline.txt
rm -f $O/*/*_kernel_trace.csv $O/*/*.db
TXQ_TRACE=1 timeout -k 10 300 python tools/e2e_profile.py 10000 > $O/e2e_10k.txt 2>&1
timeout -k 10 300 python tests/perf_config5_queries.py > $O/config5.json 2> /dev/null
timeout -k 10 100 python tools/single_query_latency.py > $O/single_query_latency.txt 2> /dev/null
g++ -O2 -std=c++20 -o /tmp/mf tests/native/matcher_fuzz.cpp tetrex_amd/csrc/host/matcher.cpp && /tmp/mf speed 400 > $O/matcher_speed.txt
timeout -k 10 200 python tests/perf_hibf.py 1048576 300 1 65536 256 > $O/hibf_65536.json 2>/dev/null
timeout -k 10 200 python tests/perf_hibf.py 1048576 300 8 65536 256 > $O/hibf_65536_shard0of8.json 2>/dev/null
tail -c 600 $O/bench_default.json; echo; cat $O/e2e_timeline.txt; grep "^rep" $O/e2e_1k.txt | tail -2; grep -E "^rep|session: 10000" $O/e2e_10k.txt | tail -3 | cut -c1-400; cat $O/matcher_speed.txt; head -8 $O/prof_c3/c3_kernel_stats.csv | cut -c1-200; cat $O/hibf_65536_shard0of8.json | cut -c1-300; cut -c1-700 $O/config5.json; cat $O/single_query_latency.txt

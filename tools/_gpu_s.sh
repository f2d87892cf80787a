cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/split
timeout -k 10 600 python -m pytest tests/test_gpu_layout_order.py -x -q 2>&1 | tail -3
timeout -k 10 300 python tools/repro_fuzz52.py 2>&1 | grep -c MISMATCH
FUZZ_ONLY=hibf-layout-900,hibf-irregular-300 timeout -k 10 400 python tools/gpu_regex_fuzz.py 600 52 2>&1 | tail -1
PERF_HIBF_NO_CHECK=0 timeout -k 10 400 python tests/perf_hibf_ragged.py 1048576 256 65536 > gpurun_out/split/ragged.json 2> gpurun_out/split/ragged.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/split/ragged.json').read().strip().splitlines()[-1])
for k in ('probe_user_order','probe_layout_order','queries_layout_order','queries_user_order','query_masks_identical','queries_layout_order_sub_tree_shards_on_this_gpu'):
    v=d.get(k); print(k, {x:(v[x] if not isinstance(v[x],dict) else v[x].get('seconds')) for x in v if x in ('seconds','kmers_per_s','queries_per_s','2','8')} if isinstance(v,dict) else v)
PY

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuse
BENCH_CLI_ROCPROF_HIP=1 BENCH_CLI_ROCPROF=$GRAFT_REPO_ROOT/gpurun_out/fuse/cold2 python tools/verify_leg.py 1 2>&1 | tail -2
ls -la gpurun_out/fuse/cold2
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/fuse/cold2/*hip_api_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
k = list(csv.DictReader(open(glob.glob('gpurun_out/fuse/cold2/*kernel_trace.csv')[0])))
ks = min(int(r['Start_Timestamp']) for r in k); ke = max(int(r['End_Timestamp']) for r in k)
print('kernels from', 0, 'to', (ke-ks)/1e6, 'ms')
big = [r for r in rows if int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 300000]
for r in big:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    print('%-40s start %+9.2f ms  dur %8.2f ms' % (r['Function'], (s-ks)/1e6, (e-s)/1e6))
PY

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/split
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/split/pytest_all.txt 2>&1; echo "rc=$?" >> gpurun_out/split/pytest_all.txt; tail -4 gpurun_out/split/pytest_all.txt
for s in 52 55 56; do timeout -k 10 330 python tools/gpu_regex_fuzz.py 600 $s > gpurun_out/split/fuzz_$s.txt 2>&1; tail -1 gpurun_out/split/fuzz_$s.txt; grep -c "compared with the oracle" gpurun_out/split/fuzz_$s.txt; done

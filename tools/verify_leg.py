"""bench.py's end_to_end.with_verification leg alone, a few times (mask / verify seconds of the CLI with -t 1 and -t 16)."""
import os, sys, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bench
class A: no_cpu = True
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    w = bench.verified_end_to_end(A)
    print({t: {k: w[t][k] for k in ("seconds", "mask_seconds", "verify_seconds")} for t in ("threads_1", "threads_16")}, flush=True)

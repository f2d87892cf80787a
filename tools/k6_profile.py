"""The k = 6 end-to-end leg of bench.py alone (end_to_end.k6_batch), for profilers:
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT -- python3 $REPO/tools/k6_profile.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from tetrex_amd import capi
import bench
capi.init(0)
class A: pass
print(json.dumps(bench.k6_end_to_end(capi, torch, A, check=not os.environ.get("K6_NO_CHECK"))))

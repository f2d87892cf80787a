#!/bin/bash
# How the staged expansion scales over host threads on this box (no GPU involved): tests/native/expand_bench.cpp on 10 000 motifs.
cd "$GRAFT_REPO_ROOT" || exit 1
python3 -c "
import sys; sys.path.insert(0,'tests')
from motifs import random_prosite_motifs
open('/tmp/m10k.txt','w').write('\n'.join(random_prosite_motifs(10000, 6))+'\n')"
g++ -O2 -march=native -std=c++20 -pthread -o /tmp/expand_bench tests/native/expand_bench.cpp tetrex_amd/csrc/host/{encoder,regex_front,kgraph,compiler,staged}.cpp || exit 1
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for t in 1 4 8 16 32; do TETREX_DENSE_EVIDENCE=dense EB_DENSE=1 /tmp/expand_bench /tmp/m10k.txt $t | tail -1; done
TETREX_TRACE=1 TETREX_DENSE_EVIDENCE=dense EB_DENSE=1 /tmp/expand_bench /tmp/m10k.txt 16 2>&1 | tail -60 | grep -v "frontier\|execute\|prune"

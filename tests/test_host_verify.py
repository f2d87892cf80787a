"""host/verify.cpp verify_batch — a `-f` batch verified bin-major, the motif's required run of residue classes searched over the
whole bin before any automaton runs (host/fasta.cpp records_with_run, AVX2) — against the motif-by-motif verify_bins (the
reference's order: include/query.h:329-346 over :126-138), row for row, through the native harness tests/native/verify_bench.cpp
(no GPU: candidate masks are random)."""
import os
import re
import subprocess

from conftest import ROOT


def test_bin_major_verification_writes_the_rows_of_the_motif_by_motif_one(tmp_path):
    exe = str(tmp_path / "verify_bench")
    host = os.path.join(ROOT, "tetrex_amd", "csrc", "host")
    subprocess.run(["g++", "-O2", "-std=c++20", "-fopenmp", "-o", exe, os.path.join(ROOT, "tests", "native", "verify_bench.cpp")] +
                   [os.path.join(host, f) for f in ("verify.cpp", "fasta.cpp", "matcher.cpp", "regex_front.cpp", "encoder.cpp")] + ["-lz"],
                   check=True, timeout=600)
    env = dict(os.environ, VERIFY_BENCH_BINS="128", VERIFY_BENCH_PER_BIN="30000")
    for threads in ("1", "3"):
        r = subprocess.run([exe, threads, "120"], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        m = re.search(r"compared with verify_bins: (\d+) of (\d+) motifs differ, (\d+) have rows", r.stdout)
        assert m, r.stdout[-2000:]
        assert int(m.group(1)) == 0 and int(m.group(2)) == 120 and int(m.group(3)) >= 10, r.stdout[-500:]

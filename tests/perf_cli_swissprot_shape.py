#!/usr/bin/env python3
"""Secondary measurement: the `tetrex` command line on Swissprot-SHAPED synthetic data — the scenario
of the reference's README (README.md:84-109: Swissprot split into 1024 bins, HIBF, k = 6, motif
LMA(E|Q)GLYN, "Query Time" 0.007 s incl. verification of the hit bins).  Writes 1024 FASTA files of
random protein sequences (uniform residues, ~200 k residues per bin, one planted LMAEGLYN / LMAQGLYN
occurrence in three bins), then times `tetrex index`, one `tetrex query -v`, and a 200-motif `-f` batch.
Prints one JSON line.  Usage: perf_cli_swissprot_shape.py [workdir] [bins] [residues_per_bin]"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
TETREX = os.path.join(ROOT, "bin", "tetrex")
AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)


def main():
    work = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="tetrex_sp_")
    bins = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    per_bin = int(sys.argv[3]) if len(sys.argv) > 3 else 200000
    os.makedirs(work, exist_ok=True)
    rng = np.random.default_rng(11)
    planted = {17: b"LMAEGLYN", 400: b"LMAQGLYN", 901 % bins: b"LMAEGLYN"}
    files = []
    t0 = time.perf_counter()
    for b in range(bins):
        seq = AA[rng.integers(0, 20, size=per_bin)].copy()
        if b in planted:
            seq[1000:1000 + len(planted[b])] = np.frombuffer(planted[b], dtype=np.uint8)
        path = os.path.join(work, "bin%04d.fa" % b)
        with open(path, "wb") as f:
            for i, start in enumerate(range(0, per_bin, 360)):  # ~Swissprot's mean protein length
                f.write(b">sp|%04d_%d\n" % (b, i))
                f.write(seq[start:start + 360].tobytes())
                f.write(b"\n")
        files.append(path)
    t_gen = time.perf_counter() - t0

    def run(*args, **kw):
        t = time.perf_counter()
        r = subprocess.run([TETREX, *args], capture_output=True, text=True, cwd=work, **kw)
        return r, time.perf_counter() - t

    r, t_index = run("index", "-k", "6", "sp", *files)
    assert r.returncode == 0 and "DONE" in r.stderr, r.stderr[-2000:]
    r, t_query = run("query", "-v", "-S", "sp.ibf", "LMA(E|Q)GLYN")
    assert r.returncode == 0, r.stderr[-2000:]
    hits = sorted({line.split("\t")[0] for line in r.stdout.splitlines() if line})
    qt = re.search(r"Query Time: ([0-9.eE+-]+)", r.stderr)
    nb = re.search(r"Narrowed Search to (\d+) possible bins", r.stderr)
    stats = [json.loads(ln) for ln in r.stderr.splitlines() if ln.startswith("{")]
    from motifs import random_prosite_motifs
    motifs = random_prosite_motifs(200, 3, wildcard=0.05, ranges=0.02, min_len=8, max_len=14)
    with open(os.path.join(work, "motifs.tsv"), "w") as f:
        for i, m in enumerate(motifs):
            f.write("M%03d\t%s\n" % (i, m))
    r, t_batch = run("query", "-S", "-f", "sp.ibf", "motifs.tsv")
    assert r.returncode == 0, r.stderr[-2000:]
    if os.environ.get("PERF_STDERR"):  # the batch run's stderr (TETREX_TRACE / TXQ_TRACE output) for whoever profiles it
        with open(os.environ["PERF_STDERR"], "w") as f:
            f.write(r.stderr)
    bstats = [json.loads(ln) for ln in r.stderr.splitlines() if ln.startswith("{")]
    print(json.dumps({
        "scenario": "Swissprot-shaped synthetic data through the tetrex CLI (HIBF, k=6, h=3, fpr 0.05)",
        "bins": bins, "residues_per_bin": per_bin, "generate_s": round(t_gen, 2),
        "index_wall_s": round(t_index, 2), "index_bytes": os.path.getsize(os.path.join(work, "sp.ibf")),
        "single_query": {"motif": "LMA(E|Q)GLYN", "process_wall_s": round(t_query, 3), "reported_query_time_s": float(qt.group(1)) if qt else None,
                         "candidate_bins": int(nb.group(1)) if nb else None, "files_with_matches": len(hits), "mask_stage": stats[0] if stats else None},
        "motif_file_batch": {"motifs": len(motifs), "process_wall_s": round(t_batch, 3), "mask_stage": bstats[0] if bstats else None},
    }))


if __name__ == "__main__":
    main()

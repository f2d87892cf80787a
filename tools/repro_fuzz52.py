"""Reproduction of a gpu_regex_fuzz.py mismatch: one regex of seed 52 on the hibf-layout-900 tree in its batch of 7, under knobs."""
import os, sys, warnings, subprocess
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
warnings.filterwarnings("ignore")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import oracle as O
    from tetrex_amd import capi
    from helpers import layout_hibf
    capi.init(0)
    ox, descs, _ = layout_hibf(O, 5, user_bins=900, tmax=32, n_values=30)
    batch = eval(os.environ["REPRO_BATCH"])
    ix = capi.Index.upload_hibf(900, descs)
    for rep in range(2):
        got, status, stats = ix.query_masks(batch, False, 4)
        for q, g, s_ in zip(batch, got, status):
            try:
                w = ox.expected_mask(q)[0]
            except Exception:
                continue
            if s_ != 0 or not np.array_equal(g, w):
                d = np.unpackbits((g ^ w).view(np.uint8), bitorder="little")
                print("  MISMATCH rep %d %r: %d bins differ (got %d bits, want %d), first %s; %s" % (rep, q, int(d.sum()), int(np.unpackbits(g.view(np.uint8)).sum()),
                      int(np.unpackbits(w.view(np.uint8)).sum()), np.nonzero(d)[0][:6].tolist(), {k: stats[k] for k in ("stages", "ops", "dense_ops", "tracked_queries")}))
    print("  done", flush=True)
    sys.exit(0)
import hypothesis
from test_fuzz_parity import regex_strategy, AA
sys.argv = [sys.argv[0]]
def draw(strategy, count, seed):
    out = []
    @hypothesis.seed(seed)
    @hypothesis.settings(max_examples=count, database=None, deadline=None, suppress_health_check=list(hypothesis.HealthCheck), phases=[hypothesis.Phase.generate])
    @hypothesis.given(strategy)
    def collect(rx):
        out.append(rx)
    collect()
    return list(dict.fromkeys(out))
qs = draw(regex_strategy(AA, max_leaves=6), 600, 52)
i = qs.index("((...){2,3})+")
batch = qs[i - i % 7: i - i % 7 + 7]
print("batch:", batch)
for label, env in [("defaults", {}), ("alone", {"ALONE": "1"}), ("TETREX_DENSE=0", {"TETREX_DENSE": "0"}), ("user order", {"TXQ_HIBF_LAYOUT_ORDER": "0"}),
                   ("untracked", {"TETREX_DENSE_TRACKED": "-1"}), ("tracked", {"TETREX_DENSE_TRACKED": "1"}), ("table", {"TXQ_KMER_TABLE_MIN": "1"}),
                   ("one stream", {"TXQ_ONE_STREAM": "1"}), ("no fused rows", {"TXQ_HIBF_LAYOUT_FUSED": "0"}), ("threads 1", {"TETREX_THREADS": "1"}),
                   ("one wave", {"TETREX_WAVE_OPS": "0"}), ("final via device", {"TXQ_FINAL_PINNED": "0"})]:
    e = dict(os.environ, **env)
    e["REPRO_BATCH"] = repr(["((...){2,3})+"] if "ALONE" in env else batch)
    print(label, flush=True)
    subprocess.run([sys.executable, __file__, "child"], env=e)

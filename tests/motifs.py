"""Query corpora for the parity tests."""
import numpy as np

AA = "ACDEFGHIKLMNPQRSTVWY"

# hand-picked: the survey's goldens plus operator coverage (k >= 3 where * / + are used)
PEPTIDE_QUERIES = [
    "LMA(E|Q)GLYN", "LMAEGLYN", "AC+G", "AB?C", "A{2,4}C", "(AB)*CDE", "AC{0,1}GH", "A.CD", "[^P]ACD", "C.{0,3}DE",
    "LM(A|C|D)E(F|G)HIK", "L[MA]E[GLY]NK", "(LM|AE)(GL|YN)K", "L(MA)+EG", "L(MA)*EG", "LMA?EGL", "L.{2}EGLY",
    "^MAEG$", ".*LMAE.+", "[^ACD]LMA[^E]G", "LMA{3}E", "L(M|A){2}EG", "W.{2}[LIVM]D[VFY]", "K[RK]{2,3}DE", "LMAE",
    "LMA", "LM", "C.{2,4}C.{3}[LIVMFYWC]", "[ST].[RK]", "N[^P][ST][^P]", "R.{2}[ST]", "[RK]{2}.[ST]",
]

DNA_QUERIES = [
    "A(C+|G+)T", "AC+G", "ACGT", "A(C|G)T", "AC?GT", "ACG{2}T", "(AC)+GT", "(AC)*GT", "A.T", "[AC]G[GT]A", "AC{1,3}G",
    "TTGACA.{3}TATAAT"[:14], "GATTACA", "A[^C]GT", "ACGTACGTAC",
]


def random_prosite_motifs(n, seed, wildcard=0.1, classes=0.3, ranges=0.05, min_len=6, max_len=14):
    """PROSITE-style motifs in POSIX form (SURVEY.md §8d 'Motif batches')."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        parts = []
        for _ in range(int(rng.integers(min_len, max_len + 1))):
            r = rng.random()
            if r < wildcard:
                parts.append(".")
            elif r < wildcard + classes:
                m = int(rng.integers(2, 6))
                parts.append("[" + "".join(rng.choice(list(AA), size=m, replace=False)) + "]")
            elif r < wildcard + classes + ranges:
                lo = int(rng.integers(0, 3))
                hi = lo + int(rng.integers(1, 3))
                parts.append(".{%d,%d}" % (lo, min(hi, 4)))
            else:
                parts.append(str(rng.choice(list(AA))))
        # keep both ends informative so trimming does not eat the motif
        parts[0] = str(rng.choice(list(AA)))
        parts[-1] = str(rng.choice(list(AA)))
        out.append("".join(parts))
    return out

"""Shared builders for the parity tests: the same synthetic index in the oracle and in HBM."""
import numpy as np

MERGED = 0xFFFFFFFFFFFFFFFF


def splitmix64(seed, n):
    """Vectorised splitmix64 stream (the PRNG SURVEY.md §8d prescribes for synthetic inputs)."""
    with np.errstate(over="ignore"):
        x = (np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def random_words(bins, bin_size, density, seed):
    """Row-major [bin_size][ceil(bins/64)] matrix with Bernoulli(density) bits; bits >= bins are 0."""
    rng = np.random.default_rng(seed)
    W = (bins + 63) // 64
    bits = rng.random((bin_size, W * 64)) < density
    bits[:, bins:] = False
    packed = np.packbits(bits.reshape(bin_size, W, 64), axis=-1, bitorder="little")
    return packed.view("<u8").reshape(bin_size * W).copy()


def oracle_ibf_from_words(O, bins, bin_size, h, words, dna=False, k=4, reduction=0):
    ix = O.Index.ibf(bins, bin_size, h, dna=dna, k=k, reduction=reduction)
    ix.set_words(words)
    return ix


def random_hibf(O, seed, user_bins=300, tmax=128, h=2, n_values=40, value_bits=20, levels=3):
    """A random HIBF: the same tree in the oracle and as upload descriptors.

    Layout rule (own, simple): user bins are dealt over `tmax`-wide IBFs; some technical bins of
    an inner IBF are merged bins pointing at a child; some user bins are split over 2-3
    consecutive technical bins.  Values of a user bin are inserted in its leaf technical bins
    and in every merged bin on the path to the root.
    """
    rng = np.random.default_rng(seed)
    values = [rng.integers(0, 1 << value_bits, size=n_values, dtype=np.uint64) for _ in range(user_bins)]
    ibfs = []  # dicts: bins, bin_size, hash_funs, tb (list of (kind, payload)), content per tb

    def build(ub_list, level):
        my = len(ibfs)
        ibfs.append(None)
        tbs = []  # (user_bin or MERGED, child id, values array)
        if level + 1 < levels and len(ub_list) > 4:
            # split the list: ~half directly here, the rest into 1-3 merged children
            rng.shuffle(ub_list)
            cut = max(1, len(ub_list) // 2)
            direct, rest = ub_list[:cut], ub_list[cut:]
            n_child = int(rng.integers(1, 4))
            parts = [list(p) for p in np.array_split(np.array(rest, dtype=np.int64), n_child) if len(p)]
        else:
            direct, parts = ub_list, []
        entries = []
        for ub in direct:
            split = int(rng.integers(1, 4)) if rng.random() < 0.3 else 1
            chunks = np.array_split(values[ub], split)
            entries.append([(ub, 0, c) for c in chunks])
        for part in parts:
            child = build([int(x) for x in part], level + 1)
            allv = np.concatenate([values[u] for u in part])
            entries.append([(MERGED, child, allv)])
        order = rng.permutation(len(entries))
        for i in order:
            tbs.extend(entries[i])
        bins = len(tbs)
        n_max = max(len(t[2]) for t in tbs) if tbs else 1
        bin_size = max(8, int(np.ceil(-max(n_max, 1) * np.log(0.05) / np.log(2) ** 2)))
        ibfs[my] = dict(bins=bins, bin_size=bin_size, hash_funs=h, tbs=tbs)
        return my

    build(list(range(user_bins)), 0)
    ox = O.Index.hibf(user_bins, dna=False, k=4)
    descs = []
    for f in ibfs:
        nxt = np.array([t[1] if t[0] == MERGED else 0 for t in f["tbs"]], dtype=np.uint64)
        tbu = np.array([t[0] for t in f["tbs"]], dtype=np.uint64)
        i = ox.add_ibf(f["bins"], f["bin_size"], f["hash_funs"], nxt, tbu)
        for tb, t in enumerate(f["tbs"]):
            if len(t[2]):
                ox.hibf_emplace(i, t[2], tb)
        descs.append(dict(bins=f["bins"], bin_size=f["bin_size"], hash_funs=f["hash_funs"],
                          words=None, next_ibf_id=nxt, tb_to_user=tbu))
    for i, d in enumerate(descs):
        d["words"] = ox.hibf_words(i)
    return ox, descs, values


def layout_hibf(O, seed, user_bins, tmax=64, h=2, n_values=20, value_bits=20, direct=6, fpr=0.05, k=4, plant=None):
    """An HIBF shaped like the layouts seqan::hibf computes (reference include/index_hibf.h:114-129 hands the layout to it):
    every IBF has at most `tmax` technical bins; user bins are taken in a SHUFFLED order (the layout sorts them by size, so
    the user bins of a leaf are no run of ids); an IBF that cannot hold its user bins directly keeps a few of them as
    technical bins of its own (`direct`, every third one split over two or three bins) and merges the rest into evenly
    sized children — as many levels as that takes.  Returns (oracle index, upload descriptors, values per user bin)."""
    rng = np.random.default_rng(seed)
    values = [rng.integers(0, 1 << value_bits, size=n_values, dtype=np.uint64) for _ in range(user_bins)]
    if plant is not None:
        plant(values)  # (a split bin's parts are runs of its value list: the first and the last value lie in different parts)
    ibfs = []

    def build(ubs):
        my = len(ibfs)
        ibfs.append(None)
        entries = []  # per technical bin: (user bin or MERGED, child, values)
        def direct_bins(lst):
            for j, ub in enumerate(lst):
                parts = 1 if j % 3 else int(rng.integers(2, 4))
                for c in np.array_split(values[ub], parts):
                    entries.append((ub, 0, c))
        if len(ubs) <= tmax:  # a leaf: every user bin a technical bin, some split while there is room (the layout fills its IBFs)
            room = tmax - len(ubs)
            for j, ub in enumerate(ubs):
                parts = 1
                if j % 3 == 0 and room >= 2:
                    parts = int(rng.integers(2, 4))
                    room -= parts - 1
                for c in np.array_split(values[ub], parts):
                    entries.append((ub, 0, c))
        else:
            mine, rest = ubs[:direct], ubs[direct:]
            direct_bins(mine)
            room = tmax - len(entries)
            n_parts = min(room, max(2, int(np.ceil(len(rest) / (tmax * 0.75)))))
            for part in np.array_split(np.array(rest, dtype=np.int64), n_parts):
                if len(part) == 0:
                    continue
                part = [int(x) for x in part]
                if len(part) == 1:
                    entries.append((part[0], 0, values[part[0]]))
                else:
                    child = build(part)
                    entries.append((MERGED, child, np.concatenate([values[u] for u in part])))
        order = rng.permutation(len(entries))
        tbs = [entries[i] for i in order]
        n_max = max(len(t[2]) for t in tbs)
        bin_size = max(8, int(np.ceil(-max(n_max, 1) * np.log(fpr) / np.log(2) ** 2)))
        ibfs[my] = dict(bins=len(tbs), bin_size=bin_size, hash_funs=h, tbs=tbs)
        return my

    build([int(x) for x in rng.permutation(user_bins)])
    ox = O.Index.hibf(user_bins, dna=False, k=k)
    descs = []
    for f in ibfs:
        nxt = np.array([t[1] if t[0] == MERGED else 0 for t in f["tbs"]], dtype=np.uint64)
        tbu = np.array([t[0] for t in f["tbs"]], dtype=np.uint64)
        i = ox.add_ibf(f["bins"], f["bin_size"], f["hash_funs"], nxt, tbu)
        for tb, t in enumerate(f["tbs"]):
            if len(t[2]):
                ox.hibf_emplace(i, t[2], tb)
        descs.append(dict(bins=f["bins"], bin_size=f["bin_size"], hash_funs=f["hash_funs"], words=None, next_ibf_id=nxt, tb_to_user=tbu))
    for i, d in enumerate(descs):
        d["words"] = ox.hibf_words(i)
    return ox, descs, values


NO_KMER = 0xFFFFFFFF


def make_blob(kmers, programs):
    """Serialise a txq_program.h blob.  programs: list of (n_slots, [(kmer, dst, a, b), ...])."""
    import struct
    kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
    n_ops = sum(len(p[1]) for p in programs)
    hdr = 48
    k_off = hdr
    p_off = k_off + kmers.size * 8
    o_off = p_off + len(programs) * 16
    out = bytearray()
    out += struct.pack("<6I3Q", 0x50515854, 1, len(programs), kmers.size, n_ops, 0, k_off, p_off, o_off)
    out += kmers.tobytes()
    first = 0
    for n_slots, ops in programs:
        out += struct.pack("<4I", first, len(ops), n_slots, 0)
        first += len(ops)
    for _, ops in programs:
        for (k, d, a, b) in ops:
            out += struct.pack("<4I", k, d, a, b)
    return bytes(out)


def eval_program(n_slots, ops, M, ones):
    """numpy evaluation of one mask-DAG program; M: (n_kmers, W) masks, ones: (W,) ONES slot."""
    W = ones.size
    S = np.zeros((n_slots, W), dtype=np.uint64)
    S[1] = ones
    for (k, d, a, b) in ops:
        x = S[a].copy()
        if k != NO_KMER:
            x &= M[k]
        S[d] = x | S[b]
    return S[2]


def ones_mask(user_bins, word0=0, words=None):
    W = (user_bins + 63) // 64
    m = np.full(W, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    if user_bins % 64:
        m[-1] = np.uint64((1 << (user_bins % 64)) - 1)
    if words is None:
        return m
    return m[word0:word0 + words]


class SessionSimulator:
    """CPU stand-in for a txq session (test double): keeps every program's slot masks, runs a
    stage's ops with numpy over oracle-probed masks and answers the alive questions.  Understands
    version-4 blobs (dense DP steps, include/txq_program.h): a program's dense blocks are arrays of their own, laid out
    inside their geometry (untracked blocks: the whole alphabet at every position; tracked blocks: what their ZERO says)."""

    DENSE_OP = 0xFFFFFFFE
    DENSE_BIT = 0x40000000
    BLOCK_SHIFT = 22
    INDEX_MASK = 0x3FFFFF

    def __init__(self, oracle_index, n_programs, dgram_index=None):
        from tetrex_amd import host
        self.host = host
        self.ox = oracle_index
        self.dg = dgram_index  # oracle flat IBF over d-gram codes (the session's auxiliary index)
        self.W = oracle_index.words_per_mask
        self.ones = ones_mask(oracle_index.bins)
        self.slots = [dict() for _ in range(n_programs)]
        # per program: block id -> dict(arr [cap, W], valid [cap] (which entries hold defined values: a shaped DENSE_ZERO of an
        # untracked block leaves the rest undefined, and nothing may touch them), geom = per position the sorted codes)
        self.blocks = [dict() for _ in range(n_programs)]
        self.stages = 0
        self.dense_steps = 0
        self.dense_kinds = [0, 0, 0, 0]  # ZERO, STEP, REDUCE, FILL ops seen
        self.tracked = [False] * n_programs  # TXQ_PROGRAM_TRACKED_BIT: the program's blocks carry live lists
        self.tracked_ops = 0
        self.block_entries = []  # capacities of the tracked blocks created

    def _entry(self, p, s):
        blk = self.blocks[p][(s & ~self.DENSE_BIT) >> self.BLOCK_SHIFT]
        i = s & self.INDEX_MASK
        assert i < blk["arr"].shape[0], "dense slot beyond the capacity of its block"
        return blk, i

    def _get(self, p, s):
        if s & self.DENSE_BIT:
            blk, i = self._entry(p, s)
            assert blk["valid"][i], "ordinary op reads a dense slot outside the zeroed shape of its block"
            return blk["arr"][i]
        if s == 0:
            return np.zeros(self.W, dtype=np.uint64)
        if s == 1:
            return self.ones
        return self.slots[p].get(s, np.zeros(self.W, dtype=np.uint64) if s == 2 else None)

    def _set(self, p, s, v):
        if s & self.DENSE_BIT:
            blk, i = self._entry(p, s)
            assert blk["valid"][i], "ordinary op writes a dense slot outside the zeroed shape of its block"
            blk["arr"][i] = v
        else:
            self.slots[p][s] = v

    @staticmethod
    def _codes(mask):
        return [c for c in range(32) if (mask >> c) & 1]

    def _canonical(self, fwd, k):
        rc = np.zeros_like(fwd)
        f = fwd.copy()
        for _ in range(k):
            rc = (rc << np.uint64(2)) | ((f & np.uint64(3)) ^ np.uint64(2))
            f >>= np.uint64(2)
        return np.minimum(fwd, rc)

    @staticmethod
    def _ranks(geom):
        """per position: code -> rank within the geometry's set (-1: not in it)"""
        out = []
        for cs in geom:
            r = np.full(32, -1, dtype=np.int64)
            r[np.array(cs, dtype=np.int64)] = np.arange(len(cs))
            out.append(r)
        return out

    def _indices(self, blk, code_lists):
        """entries of the product code_lists[0] x .. x code_lists[pos-1] in the block, and their packed k-mer bits"""
        rk = self._ranks(blk["geom"])
        idx = np.zeros(1, dtype=np.int64)
        for j, cs in enumerate(code_lists):
            c = np.array(cs, dtype=np.int64)
            assert (rk[j][c] >= 0).all(), "code outside the geometry of the block"
            idx = (idx[:, None] * len(blk["geom"][j]) + rk[j][c][None, :]).reshape(-1)
        return idx

    def _dense_op(self, p, par, row):
        kind, dst, src, r_mask = (int(x) for x in row[:4])
        k, bits, A = par["k"], par["bits"], par["alphabet"]
        pos = k - 1
        N = A ** pos
        B = self.blocks[p]
        self.dense_kinds[kind] += 1
        # every dense op of a tracked program says so, and only those (include/txq_program.h TXQ_DENSE_TRACKED)
        assert int(row[15]) & 1 == (1 if self.tracked[p] else 0), "dense op and program disagree about tracking"
        assert not (int(row[15]) & 2) or (kind == 1 and self.tracked[p]), "TXQ_DENSE_NOPROBE on something that is no tracked STEP"
        assert int(row[15]) < 4
        self.tracked_ops += int(row[15]) & 1
        noprobe = bool(int(row[15]) & 2)
        self.noprobe_steps = getattr(self, "noprobe_steps", 0) + int(noprobe)
        shape = [self._codes(int(row[4 + j])) for j in range(pos)]
        assert all(c < A for cs in shape for c in cs)

        def block_of(s):
            assert (s & self.DENSE_BIT) and (s & self.INDEX_MASK) == 0, "dense op on something that is not a block"
            return B[(s & ~self.DENSE_BIT) >> self.BLOCK_SHIFT]

        if kind == 0:  # ZERO
            b = (dst & ~self.DENSE_BIT) >> self.BLOCK_SHIFT
            assert (dst & self.DENSE_BIT) and (dst & self.INDEX_MASK) == 0
            if self.tracked[p]:  # (re)creates the block inside the geometry its shape gives; src = capacity
                entries = int(np.prod([len(cs) for cs in shape]))
                assert all(shape) and entries <= src <= (1 << self.BLOCK_SHIFT), "tracked ZERO: capacity below its geometry"
                assert b not in B or B[b]["arr"].shape[0] == src, "a block id changed its capacity"
                B[b] = dict(arr=np.zeros((src, self.W), dtype=np.uint64), valid=np.zeros(src, dtype=bool), geom=shape)
                B[b]["valid"][:entries] = True
                self.block_entries.append(src)
                return
            blk = B[b]
            if not r_mask:
                blk["arr"][:] = 0
                blk["valid"][:] = True
                return
            idx = self._indices(blk, shape)
            blk["arr"][:] = np.uint64(0xFFFFFFFFFFFFFFFF)  # nobody may touch the rest: undefined (and poisoned, should a check miss it)
            blk["arr"][idx] = 0
            blk["valid"][:] = False
            blk["valid"][idx] = True
            return
        if kind == 3:  # FILL: every entry inside the shape |= the ordinary slot src
            blk = block_of(dst)
            assert not (src & self.DENSE_BIT)
            idx = self._indices(blk, shape)
            v = self._get(p, src)
            assert v is not None, "DENSE_FILL spreads a slot that was never written"
            assert blk["valid"][idx].all(), "DENSE_FILL writes outside the zeroed shape of its block"
            blk["arr"][idx] |= v
            return
        sblk = block_of(src)
        if self.tracked[p]:
            shape = sblk["geom"]  # the work follows the live list: whatever the block holds, inside its geometry
        if kind == 2:  # REDUCE: slot dst |= OR of the entries inside the shape
            idx = self._indices(sblk, shape)
            assert sblk["valid"][idx].all(), "DENSE_REDUCE reads outside the zeroed shape of its block"
            acc = np.bitwise_or.reduce(sblk["arr"][idx], axis=0) if idx.size else np.zeros(self.W, dtype=np.uint64)
            cur = self._get(p, dst)
            assert cur is not None
            self._set(p, dst, cur | acc)
            return
        assert kind == 1
        self.dense_steps += 1
        dblk = block_of(dst)
        assert dblk is not sblk
        # (x1 .. x_{k-2}) over shape[1:]: codes per position as flat arrays
        mids = [np.zeros(1, dtype=np.int64)] * 0
        grid = np.zeros((1, 0), dtype=np.int64)
        for cs in shape[1:]:
            c = np.array(cs, dtype=np.int64)
            grid = np.concatenate([np.repeat(grid, c.size, axis=0), np.tile(c, grid.shape[0])[:, None]], axis=1)
        if grid.shape[0] == 0 or not shape[0]:
            return
        midv = np.zeros(grid.shape[0], dtype=np.uint64)
        for j in range(grid.shape[1]):
            midv = (midv << np.uint64(bits)) | grid[:, j].astype(np.uint64)
        srk, drk = self._ranks(sblk["geom"]), self._ranks(dblk["geom"])
        sn = [len(cs) for cs in sblk["geom"]]
        dn = [len(cs) for cs in dblk["geom"]]
        smid = np.zeros(grid.shape[0], dtype=np.int64)   # rank number of (x1 .. x_{k-2}) at positions 1 .. pos-1 of src
        dmid = np.zeros(grid.shape[0], dtype=np.int64)   # ... at positions 0 .. pos-2 of dst
        for j in range(grid.shape[1]):
            assert (srk[j + 1][grid[:, j]] >= 0).all()
            smid = smid * sn[j + 1] + srk[j + 1][grid[:, j]]
            dr = drk[j][grid[:, j]]
            assert (dr >= 0).all(), "DENSE_STEP leaves the geometry of its destination block"
            dmid = dmid * dn[j] + dr
        s_stride0 = int(np.prod(sn[1:])) if pos > 1 else 1
        for r in self._codes(r_mask):
            assert r < A and drk[pos - 1][r] >= 0
            acc = np.zeros((grid.shape[0], self.W), dtype=np.uint64)
            for ai in shape[0]:
                fwd = (np.uint64(ai) << np.uint64(bits * (k - 1))) | (midv << np.uint64(bits)) | np.uint64(r)
                val = self._canonical(fwd, k) if par["canonical"] else fwd
                si = int(srk[0][ai]) * s_stride0 + smid
                assert sblk["valid"][si].all(), "DENSE_STEP reads outside the zeroed shape of its source block"
                acc |= sblk["arr"][si] if noprobe else sblk["arr"][si] & self.ox.probe(val)
            di = dmid * dn[pos - 1] + int(drk[pos - 1][r])
            assert dblk["valid"][di].all(), "DENSE_STEP accumulates outside the zeroed shape of its destination block"
            dblk["arr"][di] |= acc

    def _check_level_races(self, blob, progs, dense):
        """The device runs the ops of one dependency level concurrently (txq_program.h, version 2/3 rules): within a level
        no op may read what another op writes, plain writes are unique, only accumulations (dst |= src, and DENSE_REDUCE)
        may share a destination — with dense ops counted by their whole blocks (a step reads all of src, reads and writes
        all of dst).  This simulator executes sequentially, so it checks the rule instead of depending on it."""
        levels = self.host.blob_levels(blob)
        if levels is None:
            return
        BIT, DOP = self.DENSE_BIT, self.DENSE_OP
        N = 1 << self.BLOCK_SHIFT  # a block's slot numbers
        for p, ((n_slots, ops), ends) in enumerate(zip(progs, levels)):
            begin = 0
            for end in ends:
                writes, accs, reads = {}, set(), {}   # single slots
                wr_ranges, rd_ranges = [], []         # (lo, hi, op) over dense slot ids
                for i in range(begin, end):
                    k, d, a, b = (int(x) for x in ops[i])
                    if k == DOP:
                        kind, dst, src = (int(x) for x in dense[1][d][:3])
                        if kind == 0:
                            wr_ranges.append((dst, dst + N, i))
                        elif kind == 3:
                            wr_ranges.append((dst, dst + N, i))
                            reads.setdefault(src, set()).add(i)
                        elif kind == 1:
                            wr_ranges.append((dst, dst + N, i))
                            rd_ranges.append((src, src + N, i))
                        else:
                            accs.add(dst)
                            rd_ranges.append((src, src + N, i))
                    elif k == NO_KMER and (d == a or d == b):
                        accs.add(d)
                        reads.setdefault(b if d == a else a, set()).add(i)
                    else:
                        assert d not in writes, "two plain writes to one slot in a level"
                        writes[d] = i
                        reads.setdefault(a, set()).add(i)
                        reads.setdefault(b, set()).add(i)
                for d, i in writes.items():
                    assert d not in accs, "slot written and accumulated in one level"
                    assert reads.get(d, set()) <= {i}, "slot written and read by different ops of one level"
                for d in accs:
                    assert not reads.get(d), "accumulated slot read in the same level"
                inside = lambda s, r: (s & BIT) and r[0] <= s < r[1]
                for r in wr_ranges:  # a block a dense op writes: nobody else touches it in this level
                    for s_ in list(writes) + list(accs) + list(reads):
                        assert not inside(s_, r), "ordinary op touches a block that a dense op of the same level writes"
                    for q in wr_ranges + rd_ranges:
                        assert q[2] == r[2] or q[1] <= r[0] or r[1] <= q[0], "two dense ops on one block in a level"
                for r in rd_ranges:  # a block a dense op reads: nobody writes into it in this level
                    for s_ in list(writes) + list(accs):
                        assert not inside(s_, r), "ordinary op writes into a block that a dense op of the same level reads"
                begin = end

    def stage(self, blob, qp, qs):
        kmers, progs = self.host.parse_blob(blob)
        assert len(progs) == len(self.slots)
        dense = self.host.blob_dense(blob)
        self._check_level_races(blob, progs, dense)
        if dense is not None:
            A, pos = dense[0]["alphabet"], dense[0]["k"] - 1
            for p, want in enumerate(dense[2]):
                if not self.blocks[p]:
                    self.tracked[p] = bool(want & 0x80000000)
                assert (not (want & 0x7FFFFFFF)) or self.tracked[p] == bool(want & 0x80000000), "a program changed its tracking"
                want &= 0x7FFFFFFF
                assert want <= 256
                if not self.tracked[p]:  # untracked blocks exist from the moment the program counts them: A^(k-1) entries, full geometry
                    for b in range(want):
                        if b not in self.blocks[p]:
                            self.blocks[p][b] = dict(arr=np.zeros((A ** pos, self.W), dtype=np.uint64), valid=np.zeros(A ** pos, dtype=bool),
                                                     geom=[list(range(A))] * pos)
        n_aux = self.host.blob_aux_kmers(blob)
        n_main = kmers.size - n_aux
        M = self.ox.probe(kmers[:n_main]) if n_main else np.zeros((0, self.W), dtype=np.uint64)
        if n_aux:
            assert self.dg is not None, "blob has d-gram k-mers but no auxiliary index is attached"
            M = np.concatenate([M, self.dg.probe(kmers[n_main:])])
        for p, (n_slots, ops) in enumerate(progs):
            for k, d, a, b in ops:
                if k == self.DENSE_OP:
                    assert dense is not None
                    self._dense_op(p, dense[0], dense[1][int(d)])
                    continue
                x = self._get(p, int(a))
                assert x is not None, "slot read before written"
                x = x.copy()
                if k != NO_KMER:
                    x &= M[k]
                y = self._get(p, int(b))
                assert y is not None
                assert (int(d) & self.DENSE_BIT) or (int(d) < n_slots and int(d) >= 2)
                self._set(p, int(d), x | y)
        self.stages += 1
        out = []
        for p, s in zip(qp, qs):
            v = self._get(p, s)
            assert v is not None
            out.append(int(np.unpackbits(v.view(np.uint8)).sum()).bit_length())  # 0, or 1 + floor(log2(bits set)) like the device
        return out

    def result(self, p):
        return self._get(p, 2)


def regular_hibf(O, user_bins, children, per_bin, value_fn, h=2, fpr=0.05, dna=False, k=4, reduction=0, mixed=False):
    """A regular two-level HIBF (root: `children` merged technical bins; child c: one technical bin
    per user bin of its contiguous range), the layout `tetrex index` writes.  value_fn(user_bin) ->
    uint64 array of that bin's values.  mixed: the children differ in their hash counts (h, h + 1) and row counts,
    and the root has h + 1 hash functions.  Returns (oracle index, upload descriptors)."""
    per_child = -(-user_bins // children)
    ranges = [(c * per_child, min(user_bins, (c + 1) * per_child)) for c in range(children) if c * per_child < user_bins]
    values = [value_fn(b) for b in range(user_bins)]
    ox = O.Index.hibf(user_bins, dna=dna, k=k, reduction=reduction)
    root_vals = [np.concatenate([values[b] for b in range(lo, hi)]) for lo, hi in ranges]
    m_root = max(1, O.compute_bitcount(max(len(v) for v in root_vals), fpr))
    nxt = np.arange(1, len(ranges) + 1, dtype=np.uint64)
    tbu = np.full(len(ranges), MERGED, dtype=np.uint64)
    h_root = h + 1 if mixed else h
    descs = [dict(bins=len(ranges), bin_size=m_root, hash_funs=h_root, next_ibf_id=nxt, tb_to_user=tbu, words=None)]
    ox.add_ibf(len(ranges), m_root, h_root, nxt, tbu)
    for tb, v in enumerate(root_vals):
        ox.hibf_emplace(0, v, tb)
    for c, (lo, hi) in enumerate(ranges):
        m = max(1, O.compute_bitcount(max(len(values[b]) for b in range(lo, hi)), fpr))
        hc = h
        if mixed:
            m += 7 * (c % 3)
            hc = h + c % 2
        nx = np.zeros(hi - lo, dtype=np.uint64)
        tb = np.arange(lo, hi, dtype=np.uint64)
        i = ox.add_ibf(hi - lo, m, hc, nx, tb)
        for t, b in enumerate(range(lo, hi)):
            ox.hibf_emplace(i, values[b], t)
        descs.append(dict(bins=hi - lo, bin_size=m, hash_funs=hc, next_ibf_id=nx, tb_to_user=tb, words=None))
    for i, d in enumerate(descs):
        d["words"] = ox.hibf_words(i)
    return ox, descs, values

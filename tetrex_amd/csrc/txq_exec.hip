// Mask-DAG executor for gfx950: the bit algebra of OTFCollector::collect()
// (reference include/otf_collector.h:341-393) for a whole batch of queries.
//
// A *session* holds the slot masks of a batch of programs in HBM while the host streams the
// expanded frontier to the device in stages (BASELINE north star: "the NFA k-mer-path frontier
// ... is expanded on the host and streamed to the device in batches"):
//   per stage  (1) probe every DISTINCT k-mer of the stage once (flat IBF: txq_probe.hip;
//                  HIBF: txq_hibf.hip)  ->  M[n_kmers][W].  The reference's kmer_cache_, batch-wide.
//              (2) the new ops run level by level (the host orders them into dependency levels,
//                  txq_program.h v2): small programs get one workgroup each (__syncthreads between
//                  levels), big ones are cut into units and every level is one launch over the
//                  whole GPU.  Every lane owns fixed mask-word columns: bins are independent in
//                  every operation of the collector, so an op is W independent 64-bit lanes.
//              (3) optional feedback: "is slot s of program p all zero?" (path_.none(),
//                  include/otf_collector.h:383) for the host to prune dead frontier states before
//                  it expands them further — answered as 0 or 1 + floor(log2(bits set)): how full the
//                  surviving masks are tells the host whether dense blocks pay on this index.
// Most queries finish in one stage; txq_run_programs is exactly that case.
// Format of a stage's op list: include/txq_program.h.
//
// Stages are not waited for one by one: a session owns two staging sets (blob, tables, unit and tile-group lists,
// the probe's output) and uploads on its own stream, so stage n+1 is submitted while the kernels of stage n run —
// and runs BESIDE them, on the session's second stream, when it continues nothing they work on (the next wave of
// queries of a batch).  Regions given back by finished programs are reused two stages later.
//
// Dense DP steps (blob version 3): where a query's state set saturates, the host keeps it as a block of
// A^(k-1) slots in the program's DENSE REGION and sends one op per residue set instead of one per state and
// residue.  dense_kernel runs such a step as the fused probe-AND-OR of the collector: for every destination
// suffix it hashes the k-mers of all predecessor states, gathers their h IBF rows, ANDs them with the
// predecessor's mask and ORs the result into the destination — the per-k-mer masks M[k] never exist in HBM
// (algorithmic bytes per state visit: h*W*8 of rows + W*8 of source mask, the latter L2-resident).
// Programs with dense ops always run level by level: a dense op is cut into tiles, one workgroup each (written on
// the device from one group per op, make_tiles_kernel), and the level's ordinary ops ride in the same launch.
// Where the rows come from is a policy of the kernel: a flat IBF (FlatRows), a regular two-level HIBF (TreeRows,
// TreeRowsByLane) or the interleaved children of a small uniform one (InterleavedRows).
#include "txq_internal.hpp"
#include <algorithm>
#include <thread>
#include <atomic>
#include "../../include/txq_program.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace txq {

// Slot words are shared between the waves of a workgroup across levels: every access is a relaxed
// workgroup-scope atomic, so that the barrier between two levels orders them whatever the cache
// policy of plain loads would be, and concurrent accumulations into one slot are exact.
__device__ __forceinline__ uint64_t slot_load(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void slot_store(uint64_t* p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// A dense block in HBM: [cap][W] mask words, then its live list (include/txq_program.h, tracked programs): a 64-byte
// header — number of listed entries, capacity, the block's geometry —, a bitmap of cap bits ("entry is listed"), the list.
struct BlockMeta { uint32_t* count; uint32_t* geom; uint64_t* bitmap; uint32_t* list; };
static constexpr uint32_t kBlockHeaderWords = 8;
__host__ __device__ __forceinline__ size_t block_meta_words(uint32_t cap) { return kBlockHeaderWords + ((size_t)cap + 63) / 64 + ((size_t)cap + 1) / 2; }
__device__ __forceinline__ BlockMeta block_meta(uint64_t* block, uint32_t cap, uint32_t W) {
    uint64_t* m = block + (size_t)cap * W;
    uint32_t* h = reinterpret_cast<uint32_t*>(m);
    return BlockMeta{h, h + 2, m + kBlockHeaderWords, reinterpret_cast<uint32_t*>(m + kBlockHeaderWords + ((size_t)cap + 63) / 64)};
}
// entry idx has received a bit: true for the one caller that makes it a listed entry
__device__ __forceinline__ bool mark_live(const BlockMeta& m, uint32_t idx) {
    const uint64_t bit = 1ULL << (idx & 63u);
    return !(__hip_atomic_fetch_or(m.bitmap + (idx >> 6), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit);
}
__device__ __forceinline__ void append_live(const BlockMeta& m, uint32_t idx) {
    m.list[__hip_atomic_fetch_add(m.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = idx;
}

// Where a program's slots live: ordinary slots in its slot region S; slots with TXQ_DENSE_SLOT_BIT (BIT | block << 22 |
// index) in its dense blocks, found through its row of the stage's block table: bt[0] = flags (bit 0: tracked),
// bt[1 + 2 b] = block b, bt[2 + 2 b] = its capacity in entries.
struct DenseRow {
    uint64_t* const* bt;
    __device__ __forceinline__ bool tracked() const { return (reinterpret_cast<uintptr_t>(bt[0]) & 1u) != 0; }
};
__device__ __forceinline__ uint64_t* slot_ptr(uint64_t* S, const DenseRow& D, uint32_t s, uint32_t W) {
    if (!(s & TXQ_DENSE_SLOT_BIT)) return S + (size_t)s * W;
    const uint32_t b = (s & ~TXQ_DENSE_SLOT_BIT) >> TXQ_DENSE_BLOCK_SHIFT;
    return D.bt[1 + 2 * b] + (size_t)(s & TXQ_DENSE_INDEX_MASK) * W;
}
// an ordinary op has ORed `x` into word w of dense slot s of a tracked program: the entry joins its block's live list
__device__ __forceinline__ void note_dense_write(const DenseRow& D, uint32_t s, uint32_t W, uint64_t x) {
    if (!x || !(s & TXQ_DENSE_SLOT_BIT) || !D.tracked()) return;
    const uint32_t b = (s & ~TXQ_DENSE_SLOT_BIT) >> TXQ_DENSE_BLOCK_SHIFT, idx = s & TXQ_DENSE_INDEX_MASK;
    const BlockMeta m = block_meta(D.bt[1 + 2 * b], (uint32_t)reinterpret_cast<uintptr_t>(D.bt[2 + 2 * b]), W);
    if (mark_live(m, idx)) append_live(m, idx);
}

template <int G>
__device__ __forceinline__ void run_op(const txq_op o, uint64_t* S, const DenseRow& D, const uint64_t* __restrict__ M, uint32_t W, uint32_t sub,
                                        bool concurrent) {
    const bool accumulate = o.kmer == TXQ_NO_KMER && (o.dst == o.a || o.dst == o.b);
    uint64_t* pd = slot_ptr(S, D, o.dst, W);
    if (concurrent && accumulate) {  // slot[dst] |= slot[src]; other ops of the level may hit dst too
        const uint64_t* ps = slot_ptr(S, D, o.dst == o.a ? o.b : o.a, W);
        for (uint32_t w = sub; w < W; w += G) {
            const uint64_t x = slot_load(ps + w);
            if (x) __hip_atomic_fetch_or(pd + w, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            note_dense_write(D, o.dst, W, x);
        }
        return;
    }
    const uint64_t* pa = slot_ptr(S, D, o.a, W);
    const uint64_t* pb = slot_ptr(S, D, o.b, W);
    for (uint32_t w = sub; w < W; w += G) {
        uint64_t x = slot_load(pa + w);
        if (o.kmer != TXQ_NO_KMER) x &= M[(size_t)o.kmer * W + w];
        x |= slot_load(pb + w);
        slot_store(pd + w, x);
        note_dense_write(D, o.dst, W, x);
    }
}

// One workgroup per program.  G lanes per op (pow2 >= W, at most the workgroup); lane `sub` owns words sub, sub+G, ...
// A level's ops are dealt round-robin to the workgroup's lane groups; __syncthreads() separates
// levels.  Programs without a level table run in op order on lane group 0.
template <int G>
__global__ __launch_bounds__(1024) void exec_kernel(const DevProgram* __restrict__ progs, const txq_op* __restrict__ ops,
                                                    const uint32_t* __restrict__ levels, uint64_t* const* __restrict__ slot_base,
                                                    uint32_t n_programs, const uint64_t* __restrict__ M, uint32_t W) {
    const uint32_t sub = threadIdx.x % G;
    const uint32_t group = threadIdx.x / G, n_groups = blockDim.x / G;
    for (uint32_t p = blockIdx.x; p < n_programs; p += gridDim.x) {
        const DevProgram pr = progs[p];
        if (pr.n_ops == 0) continue;
        uint64_t* S = slot_base[p];  // [slots][W]
        const DenseRow D{reinterpret_cast<uint64_t* const*>(slot_base[n_programs + p])};
        const txq_op* op = ops + pr.first_op;
        if (pr.n_levels == 0) {
            if (group == 0)
                for (uint32_t i = 0; i < pr.n_ops; ++i) run_op<G>(op[i], S, D, M, W, sub, false);
            __syncthreads();
            continue;
        }
        const uint32_t* lv = levels + pr.first_level;
        uint32_t begin = 0;
        for (uint32_t l = 0; l < pr.n_levels; ++l) {
            const uint32_t end = lv[l];
            for (uint32_t i = begin + group; i < end; i += n_groups) run_op<G>(op[i], S, D, M, W, sub, true);
            __syncthreads();
            begin = end;
        }
    }
}

// Big programs: one launch per dependency level, the level's ops of ALL big programs cut into
// units of <= unit_ops(W) ops; one workgroup per unit, G lanes per op (up to the whole workgroup for wide masks).  The kernel boundary is the
// barrier between levels, so slot words are plain loads/stores; concurrent accumulations use
// agent-scope atomics (units of one program may run on different XCDs).
struct ExecUnit { uint32_t program, begin, end; };
static constexpr uint32_t kUnitWords = 2048;  // mask words one unit moves per operand: 128 ops of a 1024-bin index, 2 ops at 65536 bins
static inline uint32_t unit_ops(uint32_t W) { return W >= kUnitWords ? 1u : kUnitWords / W; }

// g_log2: log2 of the lanes per op (a power of two <= 256)
__device__ __forceinline__ void run_unit(const ExecUnit u, const txq_op* __restrict__ ops, uint64_t* const* __restrict__ slot_base,
                                         uint32_t n_programs, const uint64_t* __restrict__ M, uint32_t W, uint32_t g_log2) {
    const uint32_t G = 1u << g_log2;
    const uint32_t sub = threadIdx.x & (G - 1), group = threadIdx.x >> g_log2, n_groups = blockDim.x >> g_log2;
    uint64_t* S = slot_base[u.program];
    const DenseRow D{reinterpret_cast<uint64_t* const*>(slot_base[n_programs + u.program])};
    for (uint32_t i = u.begin + group; i < u.end; i += n_groups) {
        const txq_op o = ops[i];
        uint64_t* pd = slot_ptr(S, D, o.dst, W);
        if (o.kmer == TXQ_NO_KMER && (o.dst == o.a || o.dst == o.b)) {
            const uint64_t* ps = slot_ptr(S, D, o.dst == o.a ? o.b : o.a, W);
            for (uint32_t w = sub; w < W; w += G) {
                const uint64_t x = ps[w];
                if (x) __hip_atomic_fetch_or(pd + w, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                note_dense_write(D, o.dst, W, x);
            }
        } else {
            const uint64_t* pa = slot_ptr(S, D, o.a, W);
            const uint64_t* pb = slot_ptr(S, D, o.b, W);
            for (uint32_t w = sub; w < W; w += G) {
                uint64_t x = pa[w];
                if (o.kmer != TXQ_NO_KMER) x &= M[(size_t)o.kmer * W + w];
                x |= pb[w];
                pd[w] = x;
                note_dense_write(D, o.dst, W, x);
            }
        }
    }
}

__global__ __launch_bounds__(256) void exec_units_kernel(const ExecUnit* __restrict__ units, const txq_op* __restrict__ ops,
                                                         uint64_t* const* __restrict__ slot_base, uint32_t n_programs,
                                                         const uint64_t* __restrict__ M, uint32_t W, uint32_t g_log2) {
    run_unit(units[blockIdx.x], ops, slot_base, n_programs, M, W, g_log2);
}

// The units of a level that also has dense tiles ride in the dense launch (its first `n_units` workgroups): the ordinary
// and the dense ops of one level are independent, and a level costs one kernel boundary instead of two.
struct LevelUnits { const ExecUnit* units; const txq_op* ops; const uint64_t* M; uint32_t n_units, g_log2; };

// ---- dense DP steps ---------------------------------------------------------------------------
// One workgroup per tile: `count` work entries of one dense op, starting at `first`.
//   ZERO    entries = slots of the block
//   REDUCE  entries = suffixes inside shape[0] x .. x shape[k-2]
//   STEP    entries = destination suffixes (x1 .. x_{k-2}, r) inside shape[1] x .. x shape[k-2] x R; each is
//           handled by G lanes (a lane owns 16 bytes of the mask: WIDE, or one word) which loop over the
//           predecessors a in shape[0], two at a time: 2 * (H row gathers + 1 source mask) loads in flight
struct DenseTile { uint32_t program, op, first, count; };
// The host does not spell the tiles of a stage out (the bench batch: 210 000 of them for 6 800 dense ops): it sends one
// group per dense op — its tiles are [first_tile, first_tile + ceil(entries / per_tile)) of the stage's tile array, the
// groups of one level back to back — and make_tiles_kernel writes them (one workgroup per group).
struct TileGroup { uint32_t program, op, entries, per_tile; uint64_t first_tile; };
static constexpr uint32_t kRootWordsLds = 4096;  // 32 KB of root verdicts per workgroup (TreeRowsByLane)
struct DenseParams { uint32_t k, bits, A, canonical, pos; uint32_t pow_a[TXQ_DENSE_MAX_POSITIONS + 1]; uint32_t nt; };  // nt: A/B bits (TXQ_DENSE_NT): 1 destination stores, 2 destination loads
// Where the blocks (and slots) of a stage's dense op live, resolved by the host side when it plans the stage:
// dst = the block written (ZERO, STEP, FILL) or the slot accumulated into (REDUCE); src = the block read (STEP, REDUCE) or
// the slot spread (FILL).  A tile reads this next to the op itself: no pointer chase through the program's tables.
struct DenseOpPtr { uint64_t* dst; const uint64_t* src; uint32_t dst_cap, src_cap; };  // (capacities of the blocks: where their live lists sit)

__global__ __launch_bounds__(256) void make_tiles_kernel(const TileGroup* __restrict__ groups, DenseTile* __restrict__ tiles) {
    const TileGroup g = groups[blockIdx.x];
    const uint32_t n = (g.entries + g.per_tile - 1) / g.per_tile;
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
        const uint32_t first = j * g.per_tile, left = g.entries - first;
        tiles[g.first_tile + j] = DenseTile{g.program, g.op, first, left < g.per_tile ? left : g.per_tile};
    }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool WIDE> struct Lane;
template <> struct Lane<true> {
    using T = u32x4;
    static constexpr uint32_t kWords = 2;
    static __device__ __forceinline__ T load(const uint64_t* p) { return *reinterpret_cast<const T*>(p); }
    static __device__ __forceinline__ void store(uint64_t* p, T v) { *reinterpret_cast<T*>(p) = v; }
    static __device__ __forceinline__ void store_nt(uint64_t* p, T v) { __builtin_nontemporal_store(v, reinterpret_cast<T*>(p)); }
    static __device__ __forceinline__ T load_nt(const uint64_t* p) { return __builtin_nontemporal_load(reinterpret_cast<const T*>(p)); }
    static __device__ __forceinline__ T zero() { return T{0u, 0u, 0u, 0u}; }
    static __device__ __forceinline__ bool any(T v) { return (v.x | v.y | v.z | v.w) != 0u; }
    static __device__ __forceinline__ bool test(T v, uint32_t bit) { return (((bit & 64u) ? ((bit & 32u) ? v.w : v.z) : ((bit & 32u) ? v.y : v.x)) >> (bit & 31u)) & 1u; }
    static __device__ __forceinline__ T with_bit(T v, uint32_t bit) {
        const uint32_t m = 1u << (bit & 31u);
        if (bit & 64u) { if (bit & 32u) v.w |= m; else v.z |= m; } else { if (bit & 32u) v.y |= m; else v.x |= m; }
        return v;
    }
    static __device__ __forceinline__ T keep(T v, bool word0, bool word1) {  // zero the words that are not kept
        const uint32_t m0 = word0 ? ~0u : 0u, m1 = word1 ? ~0u : 0u;
        return T{v.x & m0, v.y & m0, v.z & m1, v.w & m1};
    }
    static __device__ __forceinline__ T shfl_xor(T v, uint32_t o) {
        return T{(uint32_t)__shfl_xor((int)v.x, (int)o), (uint32_t)__shfl_xor((int)v.y, (int)o), (uint32_t)__shfl_xor((int)v.z, (int)o),
                 (uint32_t)__shfl_xor((int)v.w, (int)o)};
    }
};
template <> struct Lane<false> {
    using T = uint64_t;
    static constexpr uint32_t kWords = 1;
    static __device__ __forceinline__ T load(const uint64_t* p) { return *p; }
    static __device__ __forceinline__ void store(uint64_t* p, T v) { *p = v; }
    static __device__ __forceinline__ void store_nt(uint64_t* p, T v) { __builtin_nontemporal_store(v, p); }
    static __device__ __forceinline__ T load_nt(const uint64_t* p) { return __builtin_nontemporal_load(p); }
    static __device__ __forceinline__ T zero() { return 0; }
    static __device__ __forceinline__ bool any(T v) { return v != 0; }
    static __device__ __forceinline__ bool test(T v, uint32_t bit) { return (v >> (bit & 63u)) & 1ULL; }
    static __device__ __forceinline__ T with_bit(T v, uint32_t bit) { return v | (1ULL << (bit & 63u)); }
    static __device__ __forceinline__ T keep(T v, bool word0, bool) { return word0 ? v : 0; }
    static __device__ __forceinline__ T shfl_xor(T v, uint32_t o) {
        return ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), (int)o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, (int)o);
    }
};

__device__ __forceinline__ uint64_t canonical_dna(uint64_t fwd, uint32_t k) {
    uint64_t rc = 0, f = fwd;
    for (uint32_t i = 0; i < k; ++i) { rc = (rc << 2) | ((f & 3u) ^ 2u); f >>= 2; }
    return fwd <= rc ? fwd : rc;
}

// Where a step's rows M[k-mer] come from.  A lane owns chunk c (16 bytes: WIDE, or one word) of the mask; per
// predecessor it issues its loads (`issue`: the predecessor's slot chunk and the rows) and combines them later
// (`combine`: slot & rows), so that several predecessors' loads are in flight at once.
//
// Flat IBF: (src & row_0 & .. & row_{H-1}) — the fused probe-AND of SURVEY.md §7 step 6.
template <int H, bool WIDE>
struct FlatRows {
    using L = Lane<WIDE>;
    using T = typename L::T;
    struct Loads { T x[H + 1]; };
    static constexpr bool kRootByLane = false;
    static constexpr bool kPullsSplit = false;
    static constexpr int kPushUnroll = 3;
    IbfDev f;
    uint32_t c;
    __device__ __forceinline__ void prepare(uint32_t chunk) { c = chunk; }
    // SRC = false: the rows only (the caller holds the predecessor's mask: pushed steps, sparse_kernel)
    template <bool SRC = true>
    __device__ __forceinline__ void issue(const uint64_t* src_slot, uint64_t value, Loads& l) const {
        if constexpr (SRC) l.x[H] = L::load(src_slot + (size_t)c * L::kWords);
        else l.x[H] = ~L::zero();
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const uint64_t row = hash_row(value, kSeeds[i], f.hash_shift, f.bin_size);
            l.x[i] = L::load(f.words + row * f.stride + (size_t)c * L::kWords);
        }
    }
    template <bool SRC = true>
    __device__ __forceinline__ void issue_late(Loads&) const {}
    __device__ __forceinline__ T combine(const Loads& l) const {
        T y = l.x[H];
#pragma unroll
        for (int h = 0; h < H; ++h) y &= l.x[h];
        return y;
    }
};

// The same index through its table of all k-mers' masks (Index::kmer_table, ensure_kmer_table below): M[k-mer] is ONE row, read
// at the k-mer's packed value — what the reference's kmer_cache_ (include/otf_collector.h:247-262) is per query, here complete
// and resident: a step moves two rows per predecessor instead of 1 + hash_funs, out of a table that is a fraction of the matrix
// (every 4-mer of 1024 protein bins: 20 MB touched, against a 160 MB matrix).  H is not used (instantiated with 1).
template <int H, bool WIDE>
struct TableRows {
    using L = Lane<WIDE>;
    using T = typename L::T;
    struct Loads { T x[2]; };
    static constexpr bool kRootByLane = false;
    static constexpr bool kPullsSplit = false;
    static constexpr int kPushUnroll = 3;
    const uint64_t* table;
    uint32_t stride;  // words per row (the session's mask width)
    uint32_t c;
    __device__ __forceinline__ void prepare(uint32_t chunk) { c = chunk; }
    template <bool SRC = true>
    __device__ __forceinline__ void issue(const uint64_t* src_slot, uint64_t value, Loads& l) const {
        if constexpr (SRC) l.x[1] = L::load(src_slot + (size_t)c * L::kWords);
        else l.x[1] = ~L::zero();
        l.x[0] = L::load(table + value * stride + (size_t)c * L::kWords);
    }
    template <bool SRC = true>
    __device__ __forceinline__ void issue_late(Loads&) const {}
    __device__ __forceinline__ T combine(const Loads& l) const { return l.x[0] & l.x[1]; }
};

// Regular two-level HIBF (txq_internal.hpp ChildRec; membership_for(·, 1) of reference include/index_hibf.h:132-147 on
// that shape): the lane's mask words are technical bins of ONE child, so M[k-mer] there = the child's rows ANDed,
// if the k-mer is in the child's merged bin of the root (the root's rows ANDed, one bit of one word).  Two rounds of
// loads: the root words of all predecessors in flight (`issue`; the lanes of a suffix read the same words), then — only
// in the lanes whose child the root lets through — the child's rows and the predecessor's slot chunk (`issue_late`).
// With 64-bin children a full-width row would be one cache line PER CHILD and hash function; the root keeps that to
// the few children a k-mer can be in.  H = the most hash functions of any IBF of the tree; an IBF with fewer skips the others.
template <int H, bool WIDE>
struct TreeRows {
    using L = Lane<WIDE>;
    using T = typename L::T;
    struct Loads { T x[H + 1]; uint64_t r[H]; uint64_t value; const uint64_t* src; bool hit; };
    static constexpr bool kRootByLane = false;
    static constexpr bool kPullsSplit = false;
    static constexpr int kPushUnroll = 3;
    HibfNode root;
    const ChildRec* children;
    uint32_t wpr_log2;
    // the lane's child
    const uint64_t* cw;
    uint32_t c, c_rows, c_shift, c_hf, col, r_word, r_bit;
    __device__ __forceinline__ void prepare(uint32_t chunk) {
        c = chunk;
        const uint32_t word = chunk * L::kWords;
        const ChildRec rec = children[word >> wpr_log2];
        cw = (const uint64_t*)rec.words;
        c_rows = rec.bin_size;
        c_shift = rec.packed & 0xFFu;
        c_hf = (rec.packed >> 8) & 0xFu;
        col = word & ((1u << wpr_log2) - 1u);
        r_word = (rec.packed >> 12) >> 6;
        r_bit = (rec.packed >> 12) & 63u;
    }
    template <bool SRC = true>
    __device__ __forceinline__ void issue(const uint64_t* src_slot, uint64_t value, Loads& l) const {
        l.value = value;
        l.src = src_slot;
        const uint64_t* rw = (const uint64_t*)root.words;
        const uint32_t r_hf = root.hash_funs(), r_stride = root.stride(), r_shift = root.hash_shift();
#pragma unroll
        for (int i = 0; i < H; ++i)
            l.r[i] = (uint32_t)i < r_hf ? rw[hash_row_seeded(value * kSeeds[i], r_shift, root.bin_size) * r_stride + r_word] : ~0ULL;
    }
    template <bool SRC = true>
    __device__ __forceinline__ void issue_late(Loads& l) const {
        uint64_t rb = l.r[0];
#pragma unroll
        for (int h = 1; h < H; ++h) rb &= l.r[h];
        l.hit = (rb >> r_bit) & 1ULL;
        if (l.hit) {
            if constexpr (SRC) l.x[H] = L::load(l.src + (size_t)c * L::kWords);
            else l.x[H] = ~L::zero();
#pragma unroll
            for (int i = 0; i < H; ++i) {
                if ((uint32_t)i < c_hf) l.x[i] = L::load(cw + ((hash_row_seeded(l.value * kSeeds[i], c_shift, c_rows) << wpr_log2) + col));
                else l.x[i] = ~L::zero();
            }
        }
    }
    __device__ __forceinline__ T combine(const Loads& l) const {
        if (!l.hit) return L::zero();
        T y = l.x[H];
#pragma unroll
        for (int h = 0; h < H; ++h) y &= l.x[h];
        return y;
    }
    // (TreeRowsByLane) one word of the root's verdict on a k-mer: bit b = the k-mer may be in the child behind merged bin 64 * word + b
    __device__ __forceinline__ uint64_t root_word(uint64_t value, uint32_t word) const {
        const uint64_t* rw = (const uint64_t*)root.words;
        const uint32_t r_hf = root.hash_funs(), r_stride = root.stride(), r_shift = root.hash_shift();
        uint64_t x = ~0ULL;
#pragma unroll
        for (int i = 0; i < H; ++i)
            if ((uint32_t)i < r_hf) x &= rw[hash_row_seeded(value * kSeeds[i], r_shift, root.bin_size) * r_stride + word];
        return x;
    }
    __device__ __forceinline__ void issue_child(const uint64_t* src_slot, uint64_t value, Loads& l) const {
        l.x[H] = L::load(src_slot + (size_t)c * L::kWords);
#pragma unroll
        for (int i = 0; i < H; ++i) {
            if ((uint32_t)i < c_hf) l.x[i] = L::load(cw + ((hash_row_seeded(value * kSeeds[i], c_shift, c_rows) << wpr_log2) + col));
            else l.x[i] = ~L::zero();
        }
    }
    __device__ __forceinline__ T combine_child(const Loads& l) const {
        T y = l.x[H];
#pragma unroll
        for (int h = 0; h < H; ++h) y &= l.x[h];
        return y;
    }
};

// A small regular tree whose children are uniform has them interleaved (Index::interleaved): one row segment per hash
// function holds the lane's words of ALL its children — the load pattern of a flat IBF — and the root's word (at most 64
// merged bins) clears the words of the children the k-mer cannot be in.  One round of loads.
template <int H, bool WIDE>
struct InterleavedRows {
    using L = Lane<WIDE>;
    using T = typename L::T;
    struct Loads { T x[H + 1]; uint64_t r[H]; };
    static constexpr bool kRootByLane = false;
    static constexpr bool kPullsSplit = false;
    static constexpr int kPushUnroll = 3;
    IbfDev f;  // the interleaved children
    HibfNode root;
    const ChildRec* children;
    uint32_t wpr_log2;
    uint32_t c, bit0, bit1;  // root bins of the children behind the lane's first / second word
    __device__ __forceinline__ void prepare(uint32_t chunk) {
        c = chunk;
        const uint32_t word = chunk * L::kWords;
        bit0 = children[word >> wpr_log2].packed >> 12;
        bit1 = WIDE ? children[(word + 1) >> wpr_log2].packed >> 12 : bit0;
    }
    template <bool SRC = true>
    __device__ __forceinline__ void issue(const uint64_t* src_slot, uint64_t value, Loads& l) const {
        if constexpr (SRC) l.x[H] = L::load(src_slot + (size_t)c * L::kWords);
        else l.x[H] = ~L::zero();
        const uint64_t* rw = (const uint64_t*)root.words;
        const uint32_t r_hf = root.hash_funs(), r_stride = root.stride(), r_shift = root.hash_shift();
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const uint64_t sv = value * kSeeds[i];
            if ((uint32_t)i < f.hash_funs) l.x[i] = L::load(f.words + hash_row_seeded(sv, f.hash_shift, f.bin_size) * f.stride + (size_t)c * L::kWords);
            else l.x[i] = ~L::zero();
            l.r[i] = (uint32_t)i < r_hf ? rw[hash_row_seeded(sv, r_shift, root.bin_size) * r_stride] : ~0ULL;
        }
    }
    template <bool SRC = true>
    __device__ __forceinline__ void issue_late(Loads&) const {}
    __device__ __forceinline__ T combine(const Loads& l) const {
        T y = l.x[H];
        uint64_t rb = l.r[0];
#pragma unroll
        for (int h = 0; h < H; ++h) { y &= l.x[h]; rb &= l.r[h]; }
        return L::keep(y, (rb >> bit0) & 1ULL, (rb >> bit1) & 1ULL);
    }
};

// The same tree with the root's verdicts shared: the lanes of a suffix first share out the (predecessor, root word) pairs —
// every root row is fetched once per suffix, all predecessors in one round, instead of once per lane chunk — and leave
// the ANDed words in LDS; each lane then visits only the predecessors that the root lets into ITS child.  What the
// 65536-bin tree needs: its mask is 512 lane chunks wide, its root row 4 words.  (dense_kernel, kRootByLane)
template <int H, bool WIDE>
struct TreeRowsByLane : TreeRows<H, WIDE> {
    static constexpr bool kRootByLane = true;
    static constexpr bool kPullsSplit = false;
};

// A general HIBF in layout order (txq_internal.hpp VChunk): the lane's 16 bytes are two row words of ONE IBF of the tree, and
// M[k-mer] there = that IBF's rows ANDed, if the k-mer gets that far — if, in every ancestor from the root down, the rows
// ANDed have the bit of the merged bin that leads towards it (membership_for(·, 1), reference include/index_hibf.h:132-147,
// restated per technical bin).  Two rounds of loads like TreeRows: the gate words of all ancestors (8 bytes each; the lanes
// of a sub-tree read the same ones), then — where every gate is open — the predecessor's chunk and the IBF's own rows.
template <int H, bool WIDE>
struct PathRows {
    using L = Lane<WIDE>;
    using T = typename L::T;
    struct Loads { T x[H + 1]; uint64_t g[kMaxVDepth][H]; uint64_t sv[H]; const uint64_t* src; bool hit; };
    static constexpr bool kRootByLane = false;
    static constexpr bool kPullsSplit = true;  // split user bins: pull_split below
    static constexpr int kPushUnroll = 1;  // (a lane keeps its ancestors' gates in registers: one residue's loads at a time)
    const VChunk* chunks;
    const VPath* paths;
    const VSplitRange* split_range;  // per chunk: its representatives of split user bins (null: the tree has none)
    const VSplit* splits;
    uint32_t sp_first, sp_count, sp_bit0, sp_stride;
    const uint64_t* sp_side;
    T sp_reps;  // the chunk's representatives
    // the lane's chunk: its IBF and the ancestors' gates
    const uint64_t* cw;
    uint32_t c, c_rows, c_packed, col, depth;
    const uint64_t* aw[kMaxVDepth];
    uint32_t a_rows[kMaxVDepth], a_packed[kMaxVDepth], a_word[kMaxVDepth], a_bit[kMaxVDepth];
    __device__ __forceinline__ void prepare(uint32_t chunk) {
        c = chunk;
        sp_first = sp_count = 0;
        const VChunk rec = chunks[chunk];
        if (split_range && ((rec.packed >> 30) & 1u)) {
            const VSplitRange r = split_range[chunk];
            sp_first = r.first;
            sp_count = r.count;
            sp_bit0 = r.bit0;
            sp_stride = r.side_stride;
            sp_side = (const uint64_t*)r.side;
            if constexpr (WIDE) sp_reps = T{r.reps[0], r.reps[1], r.reps[2], r.reps[3]};
            else sp_reps = (uint64_t)r.reps[0] | ((uint64_t)r.reps[1] << 32);
        }
        cw = (const uint64_t*)rec.words;
        c_rows = rec.bin_size;
        c_packed = rec.packed;
        col = rec.col;
        const VPath* p = paths + rec.ibf;
        depth = p->depth;
#pragma unroll
        for (uint32_t a = 0; a < kMaxVDepth; ++a) {
            aw[a] = (const uint64_t*)p->anc[a].words;
            a_rows[a] = p->anc[a].bin_size;
            a_packed[a] = p->anc[a].packed;
            a_word[a] = p->anc[a].word;
            a_bit[a] = p->anc[a].bit;
        }
    }
    template <bool SRC = true>
    __device__ __forceinline__ void issue(const uint64_t* src_slot, uint64_t value, Loads& l) const {
        l.src = src_slot;
#pragma unroll
        for (int i = 0; i < H; ++i) l.sv[i] = value * kSeeds[i];
#pragma unroll
        for (uint32_t a = 0; a < kMaxVDepth; ++a)
#pragma unroll
            for (int i = 0; i < H; ++i) {
                l.g[a][i] = ~0ULL;
                if (a < depth && (uint32_t)i < ((a_packed[a] >> 26) & 7u))
                    l.g[a][i] = aw[a][hash_row_seeded32(l.sv[i], (a_packed[a] >> 20) & 63u, a_rows[a]) * (a_packed[a] & 0xFFFFFu) + a_word[a]];
            }
    }
    template <bool SRC = true>
    __device__ __forceinline__ void issue_late(Loads& l) const {
        bool open = true;
#pragma unroll
        for (uint32_t a = 0; a < kMaxVDepth; ++a) {
            uint64_t gw = l.g[a][0];
#pragma unroll
            for (int i = 1; i < H; ++i) gw &= l.g[a][i];
            if (a < depth) open = open && ((gw >> a_bit[a]) & 1ULL);
        }
        const uint32_t stride = c_packed & 0xFFFFFu, shift = (c_packed >> 20) & 63u, hf = (c_packed >> 26) & 7u;
        open = open && hf != 0;  // (the row's padding word belongs to no IBF)
        l.hit = open;
        if (!open) return;
        if constexpr (SRC) l.x[H] = L::load(l.src + (size_t)c * L::kWords);
        else l.x[H] = ~L::zero();
        const bool single = (c_packed >> 29) & 1u;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            l.x[i] = ~L::zero();
            if ((uint32_t)i >= hf) continue;
            const uint64_t* p = cw + (size_t)hash_row_seeded32(l.sv[i], shift, c_rows) * stride + col;
            if constexpr (WIDE) {
                if (single) { const uint64_t w = *p; l.x[i] = T{(uint32_t)w, (uint32_t)(w >> 32), 0u, 0u}; }
                else l.x[i] = L::load(p);
            } else l.x[i] = L::load(p);
        }
    }
    __device__ __forceinline__ T combine(const Loads& l) const {
        if (!l.hit) return L::zero();
        T y = l.x[H];
#pragma unroll
        for (int h = 0; h < H; ++h) y &= l.x[h];
        return y;
    }
    // Split user bins (txq_internal.hpp VSplit).  y = mask & M[k-mer] of this chunk as the IBF's rows give it, `mask` in the
    // session's form — a split bin is its representative's bit.  A representative that is set in `mask` also stays when ANOTHER
    // part of its bin holds the k-mer: the k-mer's h rows of the IBF's side matrix, ANDed, are those parts' hits — one word per
    // row for the whole chunk, asked only when a representative is open; a hit names its representative through the chunk's
    // entries (the other parts' own bits are zero in `mask`, so they vanish from y by themselves).
    __device__ __forceinline__ T pull_split(const Loads& l, T mask, T y) const {
        if (!sp_count || !l.hit) return y;
        const T open = mask & sp_reps & ~y;  // representatives that are asked for and that their own part does not answer
        if (!L::any(open)) return y;
        const uint32_t shift = (c_packed >> 20) & 63u, hf = (c_packed >> 26) & 7u;
        const bool two = sp_bit0 + sp_count > 64u;  // (more parts than the first word has room for: they go on in the next)
        uint64_t lo = ~0ULL, hi = two ? ~0ULL : 0ULL;
#pragma unroll
        for (int i = 0; i < H; ++i)
            if ((uint32_t)i < hf) {
                const uint64_t* p = sp_side + (size_t)hash_row_seeded32(l.sv[i], shift, c_rows) * sp_stride;
                lo &= p[0];
                if (two) hi &= p[1];
            }
        const uint32_t in_lo = two ? 64u - sp_bit0 : sp_count;
        lo = (lo >> sp_bit0) & (in_lo >= 64u ? ~0ULL : ((1ULL << in_lo) - 1ULL));
        if (two) hi &= (1ULL << (sp_count - in_lo)) - 1ULL;  // (fewer than 64 there: a chunk has at most 127 parts)
        for (; lo; lo &= lo - 1) {
            const uint32_t rep = splits[sp_first + (uint32_t)__builtin_ctzll(lo)].rep_bit;
            if (L::test(open, rep)) y = L::with_bit(y, rep);
        }
        for (; hi; hi &= hi - 1) {
            const uint32_t rep = splits[sp_first + in_lo + (uint32_t)__builtin_ctzll(hi)].rep_bit;
            if (L::test(open, rep)) y = L::with_bit(y, rep);
        }
        return y;
    }
};

template <int H, bool WIDE, int UA, class ROWS>
__global__ __launch_bounds__(256) void dense_kernel(ROWS rows, const DenseTile* __restrict__ tiles, const txq_dense_op* __restrict__ dops,
                                                    const DenseOpPtr* __restrict__ optr, uint64_t* const* __restrict__ slot_base, uint32_t n_programs,
                                                    uint32_t W, uint32_t G, uint32_t SL, DenseParams P, LevelUnits U) {
    using L = Lane<WIDE>;
    using T = typename L::T;
    __shared__ uint8_t codes[TXQ_DENSE_MAX_POSITIONS + 1][32];  // [j < pos]: codes of shape[j]; [pos]: codes of r_mask
    __shared__ uint32_t cnt[TXQ_DENSE_MAX_POSITIONS + 1];
    __shared__ uint64_t root_words[ROWS::kRootByLane ? kRootWordsLds : 1];  // [suffix of the pass][predecessor < 32][root word]
    if (blockIdx.x < U.n_units) {  // the whole workgroup: no barrier has been reached
        run_unit(U.units[blockIdx.x], U.ops, slot_base, n_programs, U.M, W, U.g_log2);
        return;
    }
    const DenseTile t = tiles[blockIdx.x - U.n_units];
    const txq_dense_op d = dops[t.op];
    const DenseOpPtr q = optr[t.op];
    if (d.kind == TXQ_DENSE_ZERO && !d.r_mask) {  // the whole block: entries = consecutive slots
        uint64_t* blk = q.dst;
        const size_t end = ((size_t)t.first + t.count) * W;
        for (size_t i = (size_t)t.first * W + threadIdx.x; i < end; i += blockDim.x) blk[i] = 0;
        return;
    }
    if (threadIdx.x <= P.pos) {
        const uint32_t j = threadIdx.x;
        const uint32_t mask = j < P.pos ? dops[t.op].shape[j] : d.r_mask;  // (read where it lies: indexing the copy would put it in scratch)
        uint32_t n = 0;
        for (uint32_t c = 0; c < 32; ++c)
            if ((mask >> c) & 1u) codes[j][n++] = (uint8_t)c;
        cnt[j] = n;
    }
    __syncthreads();
    const uint32_t end = t.first + t.count;
    if (d.kind == TXQ_DENSE_ZERO || d.kind == TXQ_DENSE_FILL) {  // the entries inside the shape: := 0, or |= the slot that is spread
        uint64_t* blk = q.dst;
        const bool fill = d.kind == TXQ_DENSE_FILL;
        uint32_t wl = 1;
        while (wl < W && wl < blockDim.x) wl <<= 1;
        const uint32_t sub = threadIdx.x % wl, grp = threadIdx.x / wl, groups = blockDim.x / wl;
        for (uint32_t e = t.first + grp; e < end; e += groups) {
            uint32_t r = e, idx = 0;
            for (uint32_t j = P.pos; j-- > 0;) {
                idx += (uint32_t)codes[j][r % cnt[j]] * P.pow_a[P.pos - 1 - j];
                r /= cnt[j];
            }
            if (fill) for (uint32_t w = sub; w < W; w += wl) blk[(size_t)idx * W + w] |= q.src[w];
            else for (uint32_t w = sub; w < W; w += wl) blk[(size_t)idx * W + w] = 0;
        }
        return;
    }
    const uint64_t* src = q.src;
    if (d.kind == TXQ_DENSE_REDUCE) {
        uint64_t* dst = q.dst;
        uint32_t wl = 1;
        while (wl < W && wl < blockDim.x) wl <<= 1;
        const uint32_t sub = threadIdx.x % wl, grp = threadIdx.x / wl, groups = blockDim.x / wl;
        for (uint32_t w = sub; w < W; w += wl) {
            uint64_t acc = 0;
            for (uint32_t e = t.first + grp; e < end; e += groups) {
                uint32_t r = e, idx = 0;
                for (uint32_t j = P.pos; j-- > 0;) {
                    idx += (uint32_t)codes[j][r % cnt[j]] * P.pow_a[P.pos - 1 - j];
                    r /= cnt[j];
                }
                acc |= src[(size_t)idx * W + w];
            }
            if (acc) __hip_atomic_fetch_or(dst + w, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    // STEP.  A destination suffix gets SL lane groups of G lanes: group `slice` takes every SL-th predecessor (three
    // at a time: 3 * (H row gathers + 1 source mask) loads in flight per lane), the slices are ORed with xor-shuffles.
    // One suffix per lane-group set and pass; a tile is short so that a level with few programs is not one long chain.
    uint64_t* dstb = q.dst;
    const uint32_t lanes = G * SL;  // per suffix, a power of two <= 64
    const uint32_t sub = threadIdx.x % G, slice = (threadIdx.x / G) % SL, grp = threadIdx.x / lanes, groups = blockDim.x / lanes;
    const uint32_t chunks = (W + L::kWords - 1) / L::kWords;
    const uint32_t n_r = cnt[P.pos], n_a = cnt[0];
    const uint32_t a_stride = P.pow_a[P.pos - 1];
    const uint32_t a_shift = P.bits * P.pos;
    const uint32_t rounds = (t.count + groups - 1) / groups;  // every lane takes part in the shuffles of every round
    for (uint32_t it = 0; it < rounds; ++it) {
        const uint32_t e = t.first + it * groups + grp;
        const bool live = e < end;
        uint32_t q = live ? e / n_r : 0;
        const uint32_t r = codes[P.pos][live ? e % n_r : 0];
        uint32_t mid = 0;
        uint64_t mid_val = 0;
        for (uint32_t j = P.pos; j-- > 1;) {
            const uint32_t c = codes[j][q % cnt[j]];
            q /= cnt[j];
            mid += c * P.pow_a[P.pos - 1 - j];
            mid_val |= (uint64_t)c << (P.bits * (P.pos - 1 - j));
        }
        const uint64_t low = (mid_val << P.bits) | r;  // the k-mer without its oldest residue
        uint64_t* dst = dstb + ((size_t)mid * P.A + r) * W;
        const uint64_t* srcm = src + (size_t)mid * W;
        if constexpr (ROWS::kRootByLane) {
            // Round 1: the root's verdict on every predecessor's k-mer, once per suffix — the lanes of the suffix share out the
            // (predecessor, root word) pairs and leave the ANDed words in LDS.  Round 2: per mask chunk, only the
            // predecessors the root lets into the lane's child are visited.
            const uint32_t ls = threadIdx.x % lanes, rs = rows.root.stride();
            uint64_t* mine_root = root_words + (size_t)grp * 32u * rs;
            __syncthreads();  // the previous pass has read its words
            if (live)
                for (uint32_t idx = ls; idx < n_a * rs; idx += lanes) {
                    const uint32_t j = idx / rs, w = idx - j * rs;
                    uint64_t v = ((uint64_t)codes[0][j] << a_shift) | low;
                    if (P.canonical) v = canonical_dna(v, P.k);
                    mine_root[idx] = rows.root_word(v, w);
                }
            __syncthreads();
            uint32_t my_share = 0;  // my slice's share of the predecessors: slice, slice + SL, ...
            for (uint32_t j = slice; j < n_a; j += SL) my_share |= 1u << j;
            for (uint32_t c0 = 0; c0 < chunks; c0 += G) {
                const uint32_t c = c0 + sub;
                const bool mine = live && c < chunks;
                T acc = L::zero();
                T old = L::zero();  // (travels with the first trip of gathers, see below)
                uint64_t* const p = dst + (size_t)c * L::kWords;
                if (mine && slice == 0) old = (P.nt & 2u) ? L::load_nt(p) : L::load(p);
                if (mine) {
                    rows.prepare(c);
                    uint32_t todo = 0;  // bit j: the root lets predecessor j of this suffix into my child
                    for (uint32_t j = 0; j < n_a; ++j) todo |= (uint32_t)((mine_root[j * rs + rows.r_word] >> rows.r_bit) & 1ULL) << j;
                    todo &= my_share;
                    while (todo) {
                        typename ROWS::Loads x[UA];
                        bool have[UA];
#pragma unroll
                        for (int u = 0; u < UA; ++u) {
                            have[u] = todo != 0;
                            if (have[u]) {
                                const uint32_t j = (uint32_t)__builtin_ctz(todo);
                                todo &= todo - 1;
                                const uint32_t a = codes[0][j];
                                uint64_t v = ((uint64_t)a << a_shift) | low;
                                if (P.canonical) v = canonical_dna(v, P.k);
                                rows.issue_child(srcm + (size_t)a * a_stride * W, v, x[u]);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < UA; ++u)
                            if (have[u]) acc |= rows.combine_child(x[u]);
                    }
                }
                for (uint32_t o = G; o < lanes; o <<= 1) acc |= L::shfl_xor(acc, o);
                if (mine && slice == 0 && L::any(acc)) {
                    if (P.nt & 1u) L::store_nt(p, old | acc); else L::store(p, old | acc);
                }
            }
        } else
        for (uint32_t c0 = 0; c0 < chunks; c0 += G) {
            const uint32_t c = c0 + sub;
            const bool mine = live && c < chunks;
            T acc = L::zero();
            // what the destination holds (other steps of earlier levels OR into the same entries) travels with the first trip
            // of gathers instead of being one more dependent trip behind the last
            T old = L::zero();
            uint64_t* const p = dst + (size_t)c * L::kWords;
            if (mine && slice == 0) old = (P.nt & 2u) ? L::load_nt(p) : L::load(p);
            if (mine) {
                rows.prepare(c);
                // UA predecessors per trip: UA * (H + 1) loads in flight per lane; the last trip is a partial one (its missing
                // predecessors are skipped lane group by lane group), not a tail of single predecessors — n_a = 20 residues
                // over two slices are 10 / UA trips, rounded up
                for (uint32_t i = slice; i < n_a; i += UA * SL) {
                    typename ROWS::Loads x[UA];
                    bool have[UA];
#pragma unroll
                    for (int u = 0; u < UA; ++u) {
                        have[u] = i + u * SL < n_a;
                        if (have[u]) {
                            const uint32_t a = codes[0][i + u * SL];
                            uint64_t v = ((uint64_t)a << a_shift) | low;
                            if (P.canonical) v = canonical_dna(v, P.k);
                            rows.issue(srcm + (size_t)a * a_stride * W, v, x[u]);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UA; ++u)
                        if (have[u]) rows.issue_late(x[u]);
#pragma unroll
                    for (int u = 0; u < UA; ++u)
                        if (have[u]) {
                            T y = rows.combine(x[u]);
                            if constexpr (ROWS::kPullsSplit) y = rows.pull_split(x[u], x[u].x[H], y);
                            acc |= y;
                        }
                }
            }
            for (uint32_t o = G; o < lanes; o <<= 1) acc |= L::shfl_xor(acc, o);
            if (mine && slice == 0 && L::any(acc)) {
                if (P.nt & 1u) L::store_nt(p, old | acc); else L::store(p, old | acc);
            }
        }
    }
}

__device__ __forceinline__ void dense_codes(const txq_dense_op* __restrict__ d, uint32_t pos, uint8_t (*codes)[32], uint32_t* cnt);

// ---- tracked (sparse) blocks: dense ops whose work follows the live list --------------------------
// The dense ops of tracked programs (include/txq_program.h) are not cut into tiles by the host — how many entries a
// block's list holds is known on the device only.  Per level: sparse_plan_kernel reads the counts of the level's
// groups (one group per op), turns them into chunks of kSparseChunk entries and leaves counts and the running chunk
// total in the stage's tables; sparse_kernel's workgroups share the chunks out evenly.  A ZERO's count is reset by
// the plan kernel (the chunks work from the snapshot): nothing else of the level touches that block.
struct SparseGroup { uint32_t op; uint32_t fixed; };  // fixed != kNotFixed: the host knows the entries (FILL: its shape)
static constexpr uint32_t kNotFixed = 0xFFFFFFFFu;
static constexpr uint32_t kSparseChunk = 64;
static constexpr uint32_t kUnitChunk = 256;      // most entries per chunk of sparse_units_kernel: one decoding thread each
static constexpr uint32_t kUnitFresh = 2048;     // fresh destination entries it collects per chunk (more: appended one by one)
// sparse_units_kernel cuts a group's list into chunks of about 512 UNITS (entry x residue), not of a fixed number of entries: a
// chunk is the same work whether the step rolls one residue in or twenty (a 64-entry chunk of a wildcard step is forty times
// the work of a literal's; with 256 entries the heaviest chunks were a level's critical path), and a step with few residues
// amortises a chunk's fixed trips (list, bitmap words, the append) over more entries
__host__ __device__ __forceinline__ uint32_t unit_chunk_entries(uint32_t n_r, uint32_t target) {  // target: units per chunk (TXQ_SPARSE_UNITS, 512)
    uint32_t c = n_r ? target / n_r : kUnitChunk;
    if (c > kUnitChunk) c = kUnitChunk;
    if (c < 16u) c = 16u;
    return c & ~7u;
}
static constexpr uint32_t kUnitStepWords = 32;  // masks up to this wide (2048 bins) step by units (sparse_kernel, narrow masks)
static constexpr uint32_t kMaxSparseGroups = 1024;  // per launch (the chunk totals sit in LDS)

__global__ __launch_bounds__(1024) void sparse_plan_kernel(const SparseGroup* __restrict__ groups, uint32_t n_groups, const txq_dense_op* __restrict__ dops,
                                                           const DenseOpPtr* __restrict__ optr, uint32_t W, uint32_t pos, uint32_t chunk, uint32_t* __restrict__ counts,
                                                           uint32_t* __restrict__ prefix) {
    __shared__ uint32_t scan[1024];
    const uint32_t t = threadIdx.x;
    uint32_t chunks = 0;
    if (t < n_groups) {
        const SparseGroup g = groups[t];
        uint32_t n = g.fixed;
        if (n == kNotFixed) {
            const txq_dense_op d = dops[g.op];
            const DenseOpPtr q = optr[g.op];
            if (d.kind == TXQ_DENSE_ZERO) {  // (re)creates the block: what it listed is cleared by this level's chunks, its geometry is the op's shape from now on
                uint32_t* h = reinterpret_cast<uint32_t*>(q.dst + (size_t)q.dst_cap * W);
                n = h[0];
                h[0] = 0;
                h[1] = q.dst_cap;
                for (uint32_t j = 0; j < TXQ_DENSE_MAX_POSITIONS; ++j) h[2 + j] = j < pos ? d.shape[j] : 0u;
            } else
                n = *reinterpret_cast<const uint32_t*>(q.src + (size_t)q.src_cap * W);
        }
        counts[t] = n;
        // (bit 31: by units — sparse_units_kernel — the low bits are the units per chunk)
        const uint32_t per = chunk >> 31 ? unit_chunk_entries((uint32_t)__popc(dops[g.op].r_mask), chunk & 0x7FFFFFFFu) : chunk;
        chunks = (n + per - 1) / per;
    }
    scan[t] = chunks;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        const uint32_t v = t >= o ? scan[t - o] : 0;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    if (t < n_groups) prefix[t + 1] = scan[t];
    if (t == 0) prefix[0] = 0;
}

template <bool WIDE>
__device__ __forceinline__ void atomic_or_chunk(uint64_t* p, typename Lane<WIDE>::T v) {
    if constexpr (WIDE) {
        const uint64_t lo = ((uint64_t)v.y << 32) | v.x, hi = ((uint64_t)v.w << 32) | v.z;
        if (lo) __hip_atomic_fetch_or(p, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (hi) __hip_atomic_fetch_or(p + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        if (v) __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Entries that became live in this wave join the list with ONE atomic on the block's count: `n_new` per lane (most
// lanes: 0), inclusive wave scan, the last lane reserves the range.  Returns the lane's first position.
__device__ __forceinline__ uint32_t reserve_live(const BlockMeta& m, uint32_t n_new) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t incl = n_new;
    for (uint32_t o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, (int)o);
        if (lane >= o) incl += v;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, 63);
    uint32_t base = 0;
    if (lane == 63 && total) base = __hip_atomic_fetch_add(m.count, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    base = (uint32_t)__shfl((int)base, 63);
    return base + incl - n_new;
}

// A block's geometry as the chunks use it: per position the codes in rank order (entry -> suffix) and the rank of every
// code (suffix -> entry), and the mixed-radix strides.
struct GeomTables {
    uint8_t code[TXQ_DENSE_MAX_POSITIONS][32];
    uint8_t rank[TXQ_DENSE_MAX_POSITIONS][32];  // 0xFF: the code is not in the set
    uint32_t cnt[TXQ_DENSE_MAX_POSITIONS], stride[TXQ_DENSE_MAX_POSITIONS];
};
__device__ __forceinline__ void load_geometry(GeomTables& g, const uint32_t* __restrict__ geom, uint32_t pos) {
    if (threadIdx.x < pos) {
        const uint32_t j = threadIdx.x, mask = geom[j];
        uint32_t n = 0;
        for (uint32_t c = 0; c < 32; ++c) {
            const bool in = (mask >> c) & 1u;
            g.rank[j][c] = in ? (uint8_t)n : (uint8_t)0xFF;
            if (in) g.code[j][n++] = (uint8_t)c;
        }
        g.cnt[j] = n;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t st = 1;
        for (uint32_t j = pos; j-- > 0;) { g.stride[j] = st; st *= g.cnt[j]; }
    }
    __syncthreads();
}

// WITH_STEP = false: the launch holds no STEP group (on flat indexes and tables of k-mer masks a level's STEP groups get a
// launch of their own) — the STEP code, its tables and its registers are compiled out (ROWS is not used then).
template <int H, bool WIDE, class ROWS, bool WITH_STEP = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void sparse_kernel(ROWS rows, const SparseGroup* __restrict__ groups, uint32_t n_groups, const uint32_t* __restrict__ counts,
                                                     const uint32_t* __restrict__ prefix, const txq_dense_op* __restrict__ dops,
                                                     const DenseOpPtr* __restrict__ optr, uint64_t* const* __restrict__ slot_base, uint32_t n_programs,
                                                     uint32_t W, uint32_t G, DenseParams P, LevelUnits U, uint32_t chunk) {
    using L = Lane<WIDE>;
    using T = typename L::T;
    constexpr int UA = ROWS::kPushUnroll;  // residues in flight per lane: UA * H row gathers
    __shared__ uint32_t pre[kMaxSparseGroups + 1];
    __shared__ uint8_t codes[TXQ_DENSE_MAX_POSITIONS + 1][32];  // of the op's shape (FILL) and, [pos], of its r_mask (STEP)
    __shared__ uint32_t cnt[TXQ_DENSE_MAX_POSITIONS + 1];
    __shared__ GeomTables sg, dg;  // geometry of the block read / written
    // destination entries a chunk's pushes have made live: collected here and appended to dst's list with ONE atomic on the
    // block's count per chunk (one per wave and round, thousands on one address per launch, was what the big steps waited for)
    __shared__ uint32_t fresh_list[WITH_STEP ? kSparseChunk * 32 : 1];
    __shared__ uint32_t fresh_n, fresh_at;
    if (blockIdx.x < U.n_units) {  // the level's ordinary ops ride along (the whole workgroup: no barrier has been reached)
        run_unit(U.units[blockIdx.x], U.ops, slot_base, n_programs, U.M, W, U.g_log2);
        return;
    }
    for (uint32_t i = threadIdx.x; i <= n_groups; i += blockDim.x) pre[i] = prefix[i];
    __syncthreads();
    const uint32_t total = pre[n_groups];
    const uint32_t j = blockIdx.x - U.n_units, J = gridDim.x - U.n_units;
    const uint32_t lo = (uint32_t)((uint64_t)total * j / J), hi = (uint32_t)((uint64_t)total * (j + 1) / J);
    if (lo >= hi) return;
    uint32_t g = 0;  // the group of chunk lo: the last one that starts at or before it
    for (uint32_t b = n_groups; b - g > 1;) {
        const uint32_t m = (g + b) / 2;
        if (pre[m] <= lo) g = m; else b = m;
    }
    const uint32_t chunks_w = (W + L::kWords - 1) / L::kWords;
    uint32_t loaded = 0xFFFFFFFFu;
    for (uint32_t t = lo; t < hi; ++t) {
        while (pre[g + 1] <= t) ++g;  // (groups without a chunk)
        const uint32_t first = (t - pre[g]) * chunk;  // (chunk: entries per chunk, <= kSparseChunk — the plan kernel cut the lists by it)
        const uint32_t n = counts[g];
        const uint32_t end = first + chunk < n ? first + chunk : n;
        const SparseGroup sgr = groups[g];
        const txq_dense_op d = dops[sgr.op];
        const DenseOpPtr q = optr[sgr.op];
        if (loaded != g && (d.kind == TXQ_DENSE_STEP || d.kind == TXQ_DENSE_FILL)) {
            __syncthreads();  // the previous chunk has read its tables
            dense_codes(dops + sgr.op, P.pos, codes, cnt);
            load_geometry(dg, block_meta(q.dst, q.dst_cap, W).geom, P.pos);
            if constexpr (WITH_STEP)
                if (d.kind == TXQ_DENSE_STEP) load_geometry(sg, block_meta(const_cast<uint64_t*>(q.src), q.src_cap, W).geom, P.pos);
            loaded = g;
        }
        if (d.kind == TXQ_DENSE_ZERO) {  // the listed entries := 0, their bits in the bitmap cleared (the plan kernel has reset the count)
            const BlockMeta m = block_meta(q.dst, q.dst_cap, W);
            uint32_t wl = 1;
            while (wl < W && wl < blockDim.x) wl <<= 1;
            const uint32_t sub = threadIdx.x % wl, grp = threadIdx.x / wl, ngrp = blockDim.x / wl;
            for (uint32_t e = first + grp; e < end; e += ngrp) {
                const uint32_t idx = m.list[e];
                for (uint32_t w = sub; w < W; w += wl) q.dst[(size_t)idx * W + w] = 0;
                // every bit of the bitmap is a listed entry and the ZERO clears all of them: the whole word goes (no atomic needed —
                // nothing else of the level touches this block, and the other entries of the word store the same zero)
                if (sub == 0) m.bitmap[idx >> 6] = 0;
            }
            continue;
        }
        if (d.kind == TXQ_DENSE_REDUCE) {  // slot dst |= OR of the listed entries
            const BlockMeta m = block_meta(const_cast<uint64_t*>(q.src), q.src_cap, W);
            uint32_t wl = 1;
            while (wl < W && wl < blockDim.x) wl <<= 1;
            const uint32_t sub = threadIdx.x % wl, grp = threadIdx.x / wl, ngrp = blockDim.x / wl;
            for (uint32_t w = sub; w < W; w += wl) {
                uint64_t acc = 0;
                for (uint32_t e = first + grp; e < end; e += ngrp) acc |= q.src[(size_t)m.list[e] * W + w];
                if (acc) __hip_atomic_fetch_or(q.dst + w, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            continue;
        }
        if (d.kind == TXQ_DENSE_FILL) {  // entries first .. end of the shape |= the slot; those that were empty join the list
            const BlockMeta m = block_meta(q.dst, q.dst_cap, W);
            uint32_t wl = 1;
            while (wl < W && wl < 64u) wl <<= 1;  // the lanes of an entry stay within a wave (reserve_live)
            const uint32_t sub = threadIdx.x % wl, grp = threadIdx.x / wl, ngrp = blockDim.x / wl;
            const uint32_t rounds = (end - first + ngrp - 1) / ngrp;
            for (uint32_t it = 0; it < rounds; ++it) {
                const uint32_t e = first + it * ngrp + grp;
                bool live = e < end;
                uint32_t r = live ? e : 0, idx = 0;
                for (uint32_t jj = P.pos; jj-- > 0;) {  // entry number e of the shape -> its codes -> their ranks in the block's geometry
                    const uint32_t rk = dg.rank[jj][codes[jj][r % cnt[jj]]];
                    live = live && rk != 0xFFu;
                    idx += rk * dg.stride[jj];
                    r /= cnt[jj];
                }
                live = live && idx < q.dst_cap;
                uint64_t any = 0;
                if (live)
                    for (uint32_t w = sub; w < W; w += wl) {
                        const uint64_t v = q.src[w];
                        if (v) q.dst[(size_t)idx * W + w] |= v;
                        any |= v;
                    }
                for (uint32_t o = 1; o < wl; o <<= 1) any |= Lane<false>::shfl_xor(any, o);
                const uint32_t fresh = live && sub == 0 && any && mark_live(m, idx) ? 1u : 0u;
                const uint32_t at = reserve_live(m, fresh);
                if (fresh) m.list[at] = idx;
            }
            continue;
        }
        if constexpr (!WITH_STEP) continue;  // (no such group in this launch)
        else {
        // STEP, pushed: every listed entry (a, x1 .. x_{k-2}) of src is rolled forward by the residues of r_mask —
        // dst[(x1 .. x_{k-2}, r)] |= src[entry] & M[k-mer(entry, r)] — G lanes per entry, UA residues in flight.  A product
        // that is empty is not written (the collector's path_.none() pruning, include/otf_collector.h:383); a destination entry
        // that receives its first bit joins dst's list.  Entries are numbered inside the blocks' geometries (sg, dg).
        const BlockMeta sm = block_meta(const_cast<uint64_t*>(q.src), q.src_cap, W), dm = block_meta(q.dst, q.dst_cap, W);
        const uint32_t sub = threadIdx.x % G, grp = threadIdx.x / G, ngrp = blockDim.x / G;
        const uint32_t n_r = cnt[P.pos];
        const bool noprobe = (d.reserved & TXQ_DENSE_NOPROBE) != 0;  // states that are still filling their first k-mer: the mask moves on as it is
        const uint32_t rounds = (end - first + ngrp - 1) / ngrp;
        __syncthreads();  // (the previous chunk has copied its fresh entries out)
        if (threadIdx.x == 0) fresh_n = 0;
        __syncthreads();
        for (uint32_t it = 0; it < rounds; ++it) {
            const uint32_t e = first + it * ngrp + grp;
            bool live = e < end;
            const uint32_t idx = live ? sm.list[e] : 0;
            uint64_t high = 0;   // the k-mer without the residue rolled in
            uint32_t dst0 = 0;   // the destination entry without that residue's rank
            for (uint32_t jj = P.pos, rest = idx; jj-- > 0;) {
                const uint32_t c = sg.code[jj][rest % sg.cnt[jj]];
                rest /= sg.cnt[jj];
                high |= (uint64_t)c << (P.bits * (P.pos - 1 - jj));
                if (jj > 0) {
                    const uint32_t rk = dg.rank[jj - 1][c];
                    live = live && rk != 0xFFu;
                    dst0 += rk * dg.stride[jj - 1];
                }
            }
            high <<= P.bits;
            live = live && idx < q.src_cap;
            uint32_t hit = 0;  // bit i: residue codes[pos][i] left a bit in this lane's chunk
            // A lane owns one chunk of every pass of G chunks over the mask.  Wide masks are sparse (a layout-order row of the
            // 65 536-bin trees: 20 passes, a handful of chunks with bits): first the lane finds WHICH of its chunks hold a bit —
            // four passes' loads in flight at a time, nothing else done for them —, then it works on those alone (the chunk comes
            // out of the cache again; the row source is only prepared — PathRows: the chunk's record and its ancestors' — for a
            // chunk that needs it).  Lanes whose bits sit in different passes work side by side: the wave takes as many turns as
            // its busiest lane has chunks, not one per pass.
            for (uint32_t pb = 0; pb < chunks_w; pb += G * 32u) {
                uint32_t nz = 0;
                for (uint32_t p0 = 0; p0 < 32u && pb + p0 * G < chunks_w; p0 += 4u) {
                    T s4[4];
#pragma unroll
                    for (uint32_t jj = 0; jj < 4u; ++jj) {
                        const uint32_t c = pb + (p0 + jj) * G + sub;
                        s4[jj] = live && c < chunks_w ? L::load(q.src + (size_t)idx * W + (size_t)c * L::kWords) : L::zero();
                    }
#pragma unroll
                    for (uint32_t jj = 0; jj < 4u; ++jj) nz |= (L::any(s4[jj]) ? 1u : 0u) << (p0 + jj);
                }
                for (; nz; nz &= nz - 1) {
                const uint32_t c = pb + (uint32_t)__builtin_ctz(nz) * G + sub;
                rows.prepare(c);
                const T sv = L::load(q.src + (size_t)idx * W + (size_t)c * L::kWords);
                uint32_t i = 0;
                if (noprobe) {
                    for (; i < n_r; ++i) {
                        const uint32_t rk = dg.rank[P.pos - 1][codes[P.pos][i]];
                        if (rk == 0xFFu) continue;
                        atomic_or_chunk<WIDE>(q.dst + (size_t)(dst0 + rk) * W + (size_t)c * L::kWords, sv);
                        hit |= 1u << i;
                    }
                    continue;
                }
                for (; i + UA <= n_r; i += UA) {
                    typename ROWS::Loads x[UA];
#pragma unroll
                    for (int u = 0; u < UA; ++u) {
                        uint64_t v = high | codes[P.pos][i + u];
                        if (P.canonical) v = canonical_dna(v, P.k);
                        rows.template issue<false>(nullptr, v, x[u]);
                    }
#pragma unroll
                    for (int u = 0; u < UA; ++u) rows.template issue_late<false>(x[u]);
#pragma unroll
                    for (int u = 0; u < UA; ++u) {
                        T y = sv & rows.combine(x[u]);
                        if constexpr (ROWS::kPullsSplit) y = rows.pull_split(x[u], sv, y);
                        const uint32_t rk = dg.rank[P.pos - 1][codes[P.pos][i + u]];
                        if (L::any(y) && rk != 0xFFu) {
                            atomic_or_chunk<WIDE>(q.dst + (size_t)(dst0 + rk) * W + (size_t)c * L::kWords, y);
                            hit |= 1u << (i + u);
                        }
                    }
                }
                for (; i < n_r; ++i) {
                    typename ROWS::Loads x0;
                    uint64_t v = high | codes[P.pos][i];
                    if (P.canonical) v = canonical_dna(v, P.k);
                    rows.template issue<false>(nullptr, v, x0);
                    rows.template issue_late<false>(x0);
                    T y = sv & rows.combine(x0);
                    if constexpr (ROWS::kPullsSplit) y = rows.pull_split(x0, sv, y);
                    const uint32_t rk = dg.rank[P.pos - 1][codes[P.pos][i]];
                    if (L::any(y) && rk != 0xFFu) {
                        atomic_or_chunk<WIDE>(q.dst + (size_t)(dst0 + rk) * W + (size_t)c * L::kWords, y);
                        hit |= 1u << i;
                    }
                }
                }
            }
            for (uint32_t o = 1; o < G; o <<= 1) hit |= (uint32_t)__shfl_xor((int)hit, (int)o);
            if (sub == 0 && hit) {
                // the entry's first lane notes the destinations that were empty until now.  An entry's residues land in consecutive
                // destination entries (the rolled-in residue is the last digit): the bitmap word of dst0 or the next one — TWO
                // returning atomics side by side, not one per residue in sequence (twenty round trips for a wildcard step)
                const uint32_t w0 = dst0 >> 6;
                unsigned long long b0 = 0, b1 = 0;
                for (uint32_t h = hit; h; h &= h - 1) {
                    const uint32_t entry = dst0 + dg.rank[P.pos - 1][codes[P.pos][__builtin_ctz(h)]];
                    if ((entry >> 6) == w0) b0 |= 1ULL << (entry & 63u);
                    else b1 |= 1ULL << (entry & 63u);
                }
                unsigned long long* const bm = reinterpret_cast<unsigned long long*>(dm.bitmap) + w0;
                const unsigned long long was0 = b0 ? __hip_atomic_fetch_or(bm, b0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ULL;
                const unsigned long long was1 = b1 ? __hip_atomic_fetch_or(bm + 1, b1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ULL;
                for (unsigned long long f = b0 & ~was0; f; f &= f - 1) fresh_list[atomicAdd(&fresh_n, 1u)] = w0 * 64u + (uint32_t)__builtin_ctzll(f);
                for (unsigned long long f = b1 & ~was1; f; f &= f - 1) fresh_list[atomicAdd(&fresh_n, 1u)] = (w0 + 1u) * 64u + (uint32_t)__builtin_ctzll(f);
            }
        }
        __syncthreads();
        const uint32_t n_fresh = fresh_n;  // at most 64 entries x 32 residues
        if (n_fresh) {
            if (threadIdx.x == 0) fresh_at = __hip_atomic_fetch_add(dm.count, n_fresh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < n_fresh; i += blockDim.x) dm.list[fresh_at + i] = fresh_list[i];
        }
        }
    }
}

struct StepParams { uint32_t k, bits, pos, canonical; uint32_t units; uint32_t experiment; };  // units: per chunk (unit_chunk_entries);  // experiment: TXQ_EXPERIMENTS builds only (timing experiments: wrong masks)

// ---- pushed steps on narrow masks: by units -------------------------------------------------------------------------
// The STEP groups of a level when a mask is a cache line or two (W <= kUnitStepWords: up to 2048 bins per shard), where every
// lane group touches the same lines whichever of its lanes hold bits.  sparse_kernel above walks a chunk of 64 entries in
// rounds of one entry per lane group — list -> mask chunk -> rows -> listing, twice per chunk, with one k-mer in flight per
// group for a literal residue.  Here the chunk's entries are decoded by one thread each, and the lane groups then share out
// the chunk's UNITS — (entry, residue) pairs — UA at a time: a unit's loads are the entry's mask chunk and the h rows of its
// k-mer, all in ONE trip, and UA units are in flight per group whatever the step's number of residues.  Which destination
// entries received bits is collected in LDS as bits of dst's bitmap words and asked of the bitmap with one returning atomic per
// word at the end of the chunk.  A chunk is: list (coalesced) -> its units' trips -> bitmap words -> one append of the fresh entries.
template <int H, bool WIDE, int UA, class ROWS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void sparse_units_kernel(ROWS rows, const SparseGroup* __restrict__ groups, uint32_t n_groups, const uint32_t* __restrict__ counts,
                                                           const uint32_t* __restrict__ prefix, const txq_dense_op* __restrict__ dops,
                                                           const DenseOpPtr* __restrict__ optr, uint64_t* const* __restrict__ slot_base, uint32_t n_programs,
                                                           uint32_t W, uint32_t G, StepParams P, LevelUnits U, unsigned long long* __restrict__ ctr) {
    using L = Lane<WIDE>;
    using T = typename L::T;
    __shared__ uint32_t pre[kMaxSparseGroups + 1];
    __shared__ uint8_t rcode[32], rrank[32];  // the step's residues: code, and rank in the last position of dst's geometry (0xFF: not in it)
    __shared__ GeomTables sg, dg;
    __shared__ uint32_t fresh_list[kUnitFresh];
    __shared__ uint32_t fresh_n, fresh_at;
    __shared__ uint32_t e_idx[kUnitChunk], e_dst0[kUnitChunk];  // the chunk's entries, decoded
    __shared__ uint64_t e_high[kUnitChunk];
    // the destination entries the chunk's units left bits in, as bits of dst's bitmap words: an entry's residues land in
    // consecutive destination entries (the rolled-in residue is the last digit), i.e. in the bitmap word of dst0 or the next one
    __shared__ unsigned long long live_bits[kUnitChunk][2];
    if (blockIdx.x < U.n_units) {  // the level's ordinary ops ride along (the whole workgroup: no barrier has been reached)
        run_unit(U.units[blockIdx.x], U.ops, slot_base, n_programs, U.M, W, U.g_log2);
        return;
    }
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i <= n_groups; i += blockDim.x) pre[i] = prefix[i];
    __syncthreads();
    const uint32_t total = pre[n_groups];
    const uint32_t j = blockIdx.x - U.n_units, J = gridDim.x - U.n_units;
    const uint32_t lo = (uint32_t)((uint64_t)total * j / J), hi = (uint32_t)((uint64_t)total * (j + 1) / J);
    if (lo >= hi) return;
    uint32_t g = 0;  // the group of chunk lo: the last one that starts at or before it
    for (uint32_t b = n_groups; b - g > 1;) {
        const uint32_t m = (g + b) / 2;
        if (pre[m] <= lo) g = m; else b = m;
    }
    const uint32_t chunks_w = (W + L::kWords - 1) / L::kWords;
    const uint32_t sub = tid % G, grp = tid / G, ngrp = blockDim.x / G;
    const bool lane_on = sub < chunks_w;
    if (lane_on) rows.prepare(sub);
    const unsigned long long group_lanes = (G >= 64u ? ~0ULL : ((1ULL << G) - 1ULL)) << ((tid & 63u) & ~(G - 1u));
    uint32_t loaded = 0xFFFFFFFFu, n_r = 0;
    bool noprobe = false;
    DenseOpPtr q{};
    BlockMeta dm{};
    const uint32_t* src_list = nullptr;
    unsigned long long c_entries = 0, c_units = 0;  // (TXQ_TRACE; thread 0)
    uint32_t c_products = 0, c_hits = 0;            // this lane's non-empty products; this group's units that left a bit
    for (uint32_t t = lo; t < hi; ++t) {
        while (pre[g + 1] <= t) ++g;  // (groups without a chunk)
        __syncthreads();  // the previous chunk has copied its fresh entries out and is done with the tables
        if (tid == 0) fresh_n = 0;
        if (loaded != g) {
            const uint32_t op = groups[g].op;
            q = optr[op];
            dm = block_meta(q.dst, q.dst_cap, W);
            const BlockMeta sm = block_meta(const_cast<uint64_t*>(q.src), q.src_cap, W);
            src_list = sm.list;
            noprobe = (dops[op].reserved & TXQ_DENSE_NOPROBE) != 0;  // states that are still filling their first k-mer: the mask moves on as it is
            load_geometry(dg, dm.geom, P.pos);
            load_geometry(sg, sm.geom, P.pos);
            const uint32_t r_mask = dops[op].r_mask;
            if (tid == 0) {
                uint32_t m = 0;
                for (uint32_t c = 0; c < 32; ++c)
                    if ((r_mask >> c) & 1u) { rcode[m] = (uint8_t)c; rrank[m] = dg.rank[P.pos - 1][c]; ++m; }
            }
            n_r = (uint32_t)__builtin_popcount(r_mask);
            loaded = g;
        }
        const uint32_t per = unit_chunk_entries(n_r, P.units), first = (t - pre[g]) * per, n = counts[g];
        const uint32_t end = first + per < n ? first + per : n;
        const uint32_t n_e = end > first ? end - first : 0u;
        if (tid < n_e) {  // decode: one thread per entry
            const uint32_t idx = src_list[first + tid];
            bool live = idx < q.src_cap;
            uint64_t high = 0;   // the k-mer without the residue rolled in
            uint32_t dst0 = 0;   // the destination entry without that residue's rank
            for (uint32_t jj = P.pos, rest = idx; jj-- > 0;) {
                const uint32_t cn = sg.cnt[jj];
                const uint32_t c = sg.code[jj][rest % cn];
                rest /= cn;
                high |= (uint64_t)c << (P.bits * (P.pos - 1 - jj));
                if (jj > 0) {
                    const uint32_t rk = dg.rank[jj - 1][c];
                    live = live && rk != 0xFFu;
                    dst0 += rk * dg.stride[jj - 1];
                }
            }
            e_idx[tid] = live ? idx : 0xFFFFFFFFu;
            e_high[tid] = high << P.bits;
            e_dst0[tid] = dst0;
            live_bits[tid][0] = live_bits[tid][1] = 0ULL;
        }
        __syncthreads();
        const uint32_t units = n_e * n_r;
        if (tid == 0) { c_entries += n_e; c_units += units; }
        for (uint32_t u0 = grp * UA; u0 < units; u0 += ngrp * UA) {
            typename ROWS::Loads x[UA];
#ifdef TXQ_EXPERIMENTS
            typename ROWS::Loads x2[UA];
#endif
            uint32_t dent[UA], del[UA];  // the unit's destination entry (or none) and its entry of the chunk
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const uint32_t unit = u0 + u;
                dent[u] = 0xFFFFFFFFu;
                if (unit < units) {
                    const uint32_t el = unit / n_r, ri = unit - el * n_r;
                    const uint32_t idx = e_idx[el], rk = rrank[ri];
                    del[u] = el;
                    if (idx != 0xFFFFFFFFu && rk != 0xFFu) {
                        dent[u] = e_dst0[el] + rk;
                        if (lane_on) {
                            const uint64_t* src_slot = q.src + (size_t)idx * W;
                            if (noprobe) x[u].x[0] = L::load(src_slot + (size_t)sub * L::kWords);
                            else {
                                uint64_t v = e_high[el] | rcode[ri];
                                if (P.canonical) v = canonical_dna(v, P.k);
                                rows.template issue<true>(src_slot, v, x[u]);
#ifdef TXQ_EXPERIMENTS
                                if (P.experiment & 4u) rows.template issue<false>(src_slot, v ^ 0x2A5u, x2[u]);  // the row gathers of another k-mer beside them
#endif
                            }
                        }
                    }
                }
            }
            if (!noprobe) {
#pragma unroll
                for (int u = 0; u < UA; ++u)
                    if (dent[u] != 0xFFFFFFFFu && lane_on) rows.template issue_late<true>(x[u]);
            }
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                bool nz = false;
                if (dent[u] != 0xFFFFFFFFu && lane_on) {
                    const T y = noprobe ? x[u].x[0] : rows.combine(x[u]);
                    nz = L::any(y);
#ifdef TXQ_EXPERIMENTS
                    // (timing experiments that keep the masks: every destination atomic twice — OR is idempotent —; the doubled row
                    // gathers looked at, so that they are not optimised away: a product is dropped for a value no row ever has)
                    if ((P.experiment & 4u) && !noprobe) { const T z = rows.combine(x2[u]); if (!L::any(z ^ ~L::zero())) nz = false; }  // (a row of all ones: no such row)
                    if ((P.experiment & 1u) && nz) atomic_or_chunk<WIDE>(q.dst + (size_t)dent[u] * W + (size_t)sub * L::kWords, y);
#endif
                    if (nz) atomic_or_chunk<WIDE>(q.dst + (size_t)dent[u] * W + (size_t)sub * L::kWords, y);
                    c_products += nz;
                }
                const bool group_hit = (__ballot(nz) & group_lanes) != 0ULL;
                if (group_hit && sub == 0) {  // a product of this unit is not empty: its destination entry is (now) live — noted in LDS
                    ++c_hits;
                    atomicOr(&live_bits[del[u]][(dent[u] >> 6) - (e_dst0[del[u]] >> 6)], 1ULL << (dent[u] & 63u));
                }
            }
        }
        __syncthreads();
        // ONE returning atomic per bitmap word the chunk touched (at most two per entry, whatever the number of residues — the
        // per-residue atomics were a third of this kernel's time): the bits that were clear until now are the fresh entries
        for (uint32_t p = tid; p < 2u * n_e; p += blockDim.x) {
            const uint32_t el = p >> 1, w = p & 1u;
            const unsigned long long bits = live_bits[el][w];
            if (bits) {
                const uint32_t word = (e_dst0[el] >> 6) + w;
                const unsigned long long was = __hip_atomic_fetch_or(reinterpret_cast<unsigned long long*>(dm.bitmap) + word, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (unsigned long long f = bits & ~was; f; f &= f - 1) {
                    const uint32_t entry = word * 64u + (uint32_t)__builtin_ctzll(f), at = atomicAdd(&fresh_n, 1u);
                    if (at < kUnitFresh) fresh_list[at] = entry;
                    else append_live(dm, entry);  // (more fresh entries than the list holds: one by one)
                }
            }
        }
        __syncthreads();
        const uint32_t n_fresh = fresh_n < kUnitFresh ? fresh_n : kUnitFresh;
        if (n_fresh) {
            if (tid == 0) fresh_at = __hip_atomic_fetch_add(dm.count, n_fresh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            for (uint32_t i = tid; i < n_fresh; i += blockDim.x) dm.list[fresh_at + i] = fresh_list[i];
        }
    }
    if (ctr) {  // what the pushed steps amount to (TXQ_TRACE; DESIGN.md section 3: the algorithmic bytes of this kernel)
        if (tid == 0) { atomicAdd(ctr + 0, c_entries); atomicAdd(ctr + 1, c_units); }
        if (c_products) atomicAdd(ctr + 2, (unsigned long long)c_products);
        if (c_hits) atomicAdd(ctr + 3, (unsigned long long)c_hits);
    }
}

// ---- dense steps on an HIBF --------------------------------------------------------------------
// M[k-mer] of an HIBF is a tree descent (txq_hibf.hip), not h row gathers, so a step cannot be fused.  It runs as
// three launches over a chunk of step tiles: the predecessor k-mers of every destination suffix are written out
// (pair p of a tile = suffix first + p / n_a, predecessor p % n_a), descended in one hibf_probe batch, and the
// combine kernel ANDs each mask with its predecessor's slot and ORs the result into the destination suffix.
// pair_base[tile] = index of the tile's first pair in the chunk's k-mer / mask arrays.
// (the op is read where it lies: a by-value copy indexed by threadIdx.x would live in scratch memory)
__device__ __forceinline__ void dense_codes(const txq_dense_op* __restrict__ d, uint32_t pos, uint8_t (*codes)[32], uint32_t* cnt) {
    if (threadIdx.x <= pos) {
        const uint32_t j = threadIdx.x;
        const uint32_t mask = j < pos ? d->shape[j] : d->r_mask;
        uint32_t n = 0;
        for (uint32_t c = 0; c < 32; ++c)
            if ((mask >> c) & 1u) codes[j][n++] = (uint8_t)c;
        cnt[j] = n;
    }
    __syncthreads();
}
// destination suffix number e of a step -> index of (x1 .. x_{k-2}) in A^(k-2), the rolled-in code r, and the k-mer without its oldest residue
__device__ __forceinline__ void dense_entry(const DenseParams& P, const uint8_t (*codes)[32], const uint32_t* cnt, uint32_t e, uint32_t* mid,
                                            uint32_t* r, uint64_t* low) {
    const uint32_t n_r = cnt[P.pos];
    uint32_t q = e / n_r;
    *r = codes[P.pos][e % n_r];
    uint32_t m = 0;
    uint64_t mv = 0;
    for (uint32_t j = P.pos; j-- > 1;) {
        const uint32_t c = codes[j][q % cnt[j]];
        q /= cnt[j];
        m += c * P.pow_a[P.pos - 1 - j];
        mv |= (uint64_t)c << (P.bits * (P.pos - 1 - j));
    }
    *mid = m;
    *low = (mv << P.bits) | *r;
}

__global__ __launch_bounds__(256) void dense_hibf_kmers_kernel(const DenseTile* __restrict__ tiles, const uint32_t* __restrict__ pair_base,
                                                               const txq_dense_op* __restrict__ dops, DenseParams P, uint64_t* __restrict__ kmers) {
    __shared__ uint8_t codes[TXQ_DENSE_MAX_POSITIONS + 1][32];
    __shared__ uint32_t cnt[TXQ_DENSE_MAX_POSITIONS + 1];
    const DenseTile t = tiles[blockIdx.x];
    dense_codes(dops + t.op, P.pos, codes, cnt);
    const uint32_t n_a = cnt[0];
    uint64_t* out = kmers + pair_base[blockIdx.x];
    for (uint32_t p = threadIdx.x; p < t.count * n_a; p += blockDim.x) {
        uint32_t mid, r;
        uint64_t low;
        dense_entry(P, codes, cnt, t.first + p / n_a, &mid, &r, &low);
        uint64_t v = ((uint64_t)codes[0][p % n_a] << (P.bits * P.pos)) | low;
        if (P.canonical) v = canonical_dna(v, P.k);
        out[p] = v;
    }
}

__global__ __launch_bounds__(256) void dense_hibf_combine_kernel(const DenseTile* __restrict__ tiles, const uint32_t* __restrict__ pair_base,
                                                                 const txq_dense_op* __restrict__ dops, const DenseOpPtr* __restrict__ optr,
                                                                 uint32_t W, DenseParams P, const uint64_t* __restrict__ masks) {
    __shared__ uint8_t codes[TXQ_DENSE_MAX_POSITIONS + 1][32];
    __shared__ uint32_t cnt[TXQ_DENSE_MAX_POSITIONS + 1];
    const DenseTile t = tiles[blockIdx.x];
    dense_codes(dops + t.op, P.pos, codes, cnt);
    const uint64_t* src = optr[t.op].src;
    uint64_t* dstb = optr[t.op].dst;
    const uint64_t* M = masks + (size_t)pair_base[blockIdx.x] * W;
    uint32_t wl = 1;  // lanes per suffix: one word each
    while (wl < W && wl < blockDim.x) wl <<= 1;
    const uint32_t sub = threadIdx.x % wl, grp = threadIdx.x / wl, groups = blockDim.x / wl;
    const uint32_t n_a = cnt[0], a_stride = P.pow_a[P.pos - 1];
    for (uint32_t e = grp; e < t.count; e += groups) {
        uint32_t mid, r;
        uint64_t low;
        dense_entry(P, codes, cnt, t.first + e, &mid, &r, &low);
        uint64_t* dst = dstb + ((size_t)mid * P.A + r) * W;
        for (uint32_t w = sub; w < W; w += wl) {
            uint64_t acc = 0;
            for (uint32_t i = 0; i < n_a; ++i)
                acc |= src[((size_t)codes[0][i] * a_stride + mid) * W + w] & M[((size_t)e * n_a + i) * W + w];
            if (acc) dst[w] |= acc;
        }
    }
}

// grown slot regions keep their contents: all moves of a stage in ONE launch (a hipMemcpyAsync per program cost 20 ms
// of host time when a thousand programs' dense regions doubled in the same stage); blockIdx.y cuts a move into 16 slices
struct RegionMove { uint64_t* dst; const uint64_t* src; size_t words; };
__global__ __launch_bounds__(256) void move_regions_kernel(const RegionMove* __restrict__ moves) {
    const RegionMove m = moves[blockIdx.x];
    const size_t per = (m.words + gridDim.y - 1) / gridDim.y;
    const size_t lo = per * blockIdx.y, hi = lo + per < m.words ? lo + per : m.words;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) m.dst[i] = m.src[i];
}

// blocks a tracked program takes over start all zero (masks, list, bitmap): all of a stage's in ONE launch (a memset per block
// was 550 calls of the runtime per 200-motif batch at k = 6); blockIdx.y cuts a block into 16 slices
__global__ __launch_bounds__(256) void clear_blocks_kernel(const RegionMove* __restrict__ jobs) {
    const RegionMove m = jobs[blockIdx.x];
    const size_t per = (m.words + gridDim.y - 1) / gridDim.y;
    const size_t lo = per * blockIdx.y, hi = lo + per < m.words ? lo + per : m.words;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) m.dst[i] = 0;
}

// constants of programs that just received their first slot region
__global__ __launch_bounds__(256) void init_slots_kernel(uint64_t* const* __restrict__ slot_base, const uint32_t* __restrict__ which,
                                                         uint32_t n, uint32_t W, uint64_t user_bins, uint64_t word0, const uint64_t* __restrict__ ones_words) {
    const size_t total = (size_t)n * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t p = which[i / W], w = (uint32_t)(i % W);
        uint64_t* S = slot_base[p];
        // ONES = hit_vector(bin_count, true): the bits of this shard's word that are real bins
        const uint64_t first_bin = (word0 + w) * 64;
        uint64_t ones = 0;
        if (first_bin < user_bins) ones = (user_bins - first_bin >= 64) ? ~0ULL : ((1ULL << (user_bins - first_bin)) - 1ULL);
        if (ones_words) ones = ones_words[w];  // a layout-order session: the technical bins that are user bins
        S[(size_t)TXQ_SLOT_ZERO * W + w] = 0;
        S[(size_t)TXQ_SLOT_ONES * W + w] = ones;
        S[(size_t)TXQ_SLOT_RESULT * W + w] = 0;
    }
}

// alive[i] = 0 when slot q_slot[i] of program q_prog[i] has no bit set, else 1 + floor(log2(bits set)); one wave per query
__global__ __launch_bounds__(256) void slot_alive_kernel(uint64_t* const* __restrict__ slot_base, const uint32_t* __restrict__ q_prog,
                                                         const uint32_t* __restrict__ q_slot, uint32_t n, uint32_t W,
                                                         uint8_t* __restrict__ alive) {
    const uint32_t lane = threadIdx.x & 63;
    const size_t n_waves = (size_t)gridDim.x * (blockDim.x >> 6);
    for (size_t i = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < n; i += n_waves) {
        const uint64_t* s = slot_base[q_prog[i]] + (size_t)q_slot[i] * W;
        uint32_t bits = 0;
        for (uint32_t w = lane; w < W; w += 64) bits += (uint32_t)__popcll(s[w]);
        for (uint32_t o = 32; o; o >>= 1) bits += (uint32_t)__shfl_xor((int)bits, (int)o);
        if (lane == 0) alive[i] = bits ? (uint8_t)(32 - __clz((int)bits)) : 0;
    }
}

__global__ __launch_bounds__(256) void gather_result_kernel(uint64_t* const* __restrict__ slot_base, uint32_t n_programs, uint32_t W,
                                                            uint64_t* __restrict__ final_masks) {
    const size_t total = (size_t)n_programs * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / W, w = i % W;
        const uint64_t* region = slot_base[p];
        final_masks[i] = region ? region[(size_t)TXQ_SLOT_RESULT * W + w] : 0;
    }
}

#define TXQ_HIP(call)                                        \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return fail_hip(e_, #call);    \
    } while (0)

// Host-side validation: nothing malformed may reach the GPU (a stray slot or k-mer index would
// be an out-of-bounds access there).  Accepts version 1 (op order), 2 (levels) and 3 (levels + dense
// ops) blobs and normalises the program table.
struct BlobView {
    uint32_t n_kmers = 0, n_ops = 0, n_levels = 0, n_dense = 0;
    uint64_t kmers_offset = 0, ops_offset = 0, levels_offset = 0, n_aux_kmers = 0, dense_offset = 0;
    DenseParams dense{};
    uint32_t block_slots = 0;  // A^(k-1) when the blob is version 3
    std::vector<DevProgram> programs;
    std::vector<uint32_t> n_slots, n_blocks;  // per program: ordinary slots; dense blocks (ids 0 .. n-1)
    std::vector<uint8_t> has_dense;  // the program has dense ops in this stage
    std::vector<uint8_t> tracked;    // TXQ_PROGRAM_TRACKED_BIT
};

static int validate_blob(const unsigned char* blob, size_t bytes, size_t n_programs, const Knobs& kn, BlobView* out) {
    if (bytes < sizeof(txq_blob_header)) return fail(TXQ_ERR_PROGRAM, "blob shorter than its header");
    if ((uintptr_t)blob % 8) return fail(TXQ_ERR_PROGRAM, "blob must be 8-byte aligned");
    const txq_blob_header* h1 = (const txq_blob_header*)blob;
    if (h1->magic != TXQ_PROGRAM_MAGIC) return fail(TXQ_ERR_PROGRAM, "bad blob magic");
    const bool v3 = h1->version == TXQ_PROGRAM_VERSION_DENSE;  // (levels + dense ops)
    const bool v2 = v3 || h1->version == TXQ_PROGRAM_VERSION_LEVELS;
    if (!v2 && h1->version != TXQ_PROGRAM_VERSION) return fail(TXQ_ERR_PROGRAM, "unsupported blob version %u", h1->version);
    if (bytes < (v3 ? sizeof(txq_blob_header_v3) : v2 ? sizeof(txq_blob_header_v2) : sizeof(txq_blob_header)))
        return fail(TXQ_ERR_PROGRAM, "blob shorter than its header");
    const txq_blob_header_v2* h2 = (const txq_blob_header_v2*)blob;
    const txq_blob_header_v3* h3 = (const txq_blob_header_v3*)blob;
    BlobView v;
    uint64_t programs_offset;
    if (v2) {
        v.n_kmers = h2->n_kmers; v.n_ops = h2->n_ops; v.n_levels = h2->n_levels;
        v.kmers_offset = h2->kmers_offset; v.ops_offset = h2->ops_offset; v.levels_offset = h2->levels_offset;
        v.n_aux_kmers = h2->n_aux_kmers;
        if (v.n_aux_kmers > v.n_kmers) return fail(TXQ_ERR_PROGRAM, "more auxiliary k-mers than k-mers");
        programs_offset = h2->programs_offset;
        if (h2->n_programs != n_programs) return fail(TXQ_ERR_PROGRAM, "blob holds %u programs, caller says %zu", h2->n_programs, n_programs);
    } else {
        v.n_kmers = h1->n_kmers; v.n_ops = h1->n_ops;
        v.kmers_offset = h1->kmers_offset; v.ops_offset = h1->ops_offset;
        programs_offset = h1->programs_offset;
        if (h1->n_programs != n_programs) return fail(TXQ_ERR_PROGRAM, "blob holds %u programs, caller says %zu", h1->n_programs, n_programs);
    }
    auto in_range = [&](uint64_t off, uint64_t count, uint64_t elem) {
        return off % 4 == 0 && off <= bytes && count <= (bytes - off) / elem;
    };
    if (v.kmers_offset % 8 || !in_range(v.kmers_offset, v.n_kmers, 8) || !in_range(v.ops_offset, v.n_ops, sizeof(txq_op)) ||
        !in_range(programs_offset, n_programs, v2 ? sizeof(txq_program_v2) : sizeof(txq_program)) ||
        (v2 && !in_range(v.levels_offset, v.n_levels, 4)))
        return fail(TXQ_ERR_PROGRAM, "blob table outside the blob");
    if (v3) {
        v.n_dense = h3->n_dense;
        v.dense_offset = h3->dense_offset;
        if (v.dense_offset % 8 || !in_range(v.dense_offset, v.n_dense, sizeof(txq_dense_op))) return fail(TXQ_ERR_PROGRAM, "dense table outside the blob");
        DenseParams& P = v.dense;
        P.k = h3->k; P.bits = h3->bits; P.A = h3->alphabet; P.canonical = h3->canonical ? 1u : 0u;
        if (P.k < 2 || P.k - 1 > TXQ_DENSE_MAX_POSITIONS || P.bits < 1 || P.bits > 8 || (uint64_t)P.bits * P.k > 64 || P.A < 1 || P.A > 32 ||
            P.A > (1u << P.bits) || (P.canonical && P.bits != 2))
            return fail(TXQ_ERR_PROGRAM, "dense parameters out of range (k %u, %u bits, alphabet %u)", P.k, P.bits, P.A);
        P.pos = P.k - 1;
        P.nt = (uint32_t)kn.dense_nt;
        uint64_t n = 1;
        P.pow_a[0] = 1;
        for (uint32_t j = 1; j <= P.pos; ++j) {
            n *= P.A;
            if (n > (1u << 22)) return fail(TXQ_ERR_PROGRAM, "dense block of %u^%u slots is too large", P.A, P.pos);
            P.pow_a[j] = (uint32_t)n;
        }
        v.block_slots = (uint32_t)n;
    }
    v.programs.resize(n_programs);
    v.n_slots.resize(n_programs);
    v.n_blocks.assign(n_programs, 0);
    v.has_dense.assign(n_programs, 0);
    v.tracked.assign(n_programs, 0);
    const txq_op* ops = (const txq_op*)(blob + v.ops_offset);
    const txq_dense_op* dops = v3 ? (const txq_dense_op*)(blob + v.dense_offset) : nullptr;
    const uint32_t* levels = v2 ? (const uint32_t*)(blob + v.levels_offset) : nullptr;
    for (uint32_t p = 0; p < n_programs; ++p) {
        DevProgram d{};
        if (v2) {
            const txq_program_v2& s = ((const txq_program_v2*)(blob + programs_offset))[p];
            d = DevProgram{s.first_op, s.n_ops, s.first_level, s.n_levels};
            v.n_slots[p] = s.n_slots;
            if (v3) {
                v.tracked[p] = (s.reserved & TXQ_PROGRAM_TRACKED_BIT) != 0;
                v.n_blocks[p] = s.reserved & ~TXQ_PROGRAM_TRACKED_BIT;
                if (v.n_blocks[p] > TXQ_DENSE_MAX_BLOCKS) return fail(TXQ_ERR_PROGRAM, "program %u: more than %u dense blocks", p, TXQ_DENSE_MAX_BLOCKS);
            }
        } else {
            const txq_program& s = ((const txq_program*)(blob + programs_offset))[p];
            d = DevProgram{s.first_op, s.n_ops, 0, 0};
            v.n_slots[p] = s.n_slots;
        }
        const uint32_t n_slots = v.n_slots[p];
        if (n_slots < TXQ_SLOT_FIRST_FREE || n_slots >= TXQ_DENSE_SLOT_BIT) return fail(TXQ_ERR_PROGRAM, "program %u: n_slots out of range", p);
        if (d.first_op > v.n_ops || d.n_ops > v.n_ops - d.first_op) return fail(TXQ_ERR_PROGRAM, "program %u: ops out of range", p);
        if (d.n_levels) {
            if (d.first_level > v.n_levels || d.n_levels > v.n_levels - d.first_level) return fail(TXQ_ERR_PROGRAM, "program %u: levels out of range", p);
            uint32_t prev = 0;
            for (uint32_t l = 0; l < d.n_levels; ++l) {
                const uint32_t e = levels[d.first_level + l];
                if (e < prev || e > d.n_ops) return fail(TXQ_ERR_PROGRAM, "program %u: level table not ascending", p);
                prev = e;
            }
            if (prev != d.n_ops) return fail(TXQ_ERR_PROGRAM, "program %u: levels do not cover the ops", p);
        }
        v.programs[p] = d;
    }
    // every op of every program: operands inside the program's slot regions, k-mer inside the table, dense ops on
    // whole blocks.  Large stages (hundreds of MB of ops) are checked by several threads, each taking whole programs.
    struct Bad { uint32_t program = 0xFFFFFFFFu, op = 0; int kind = 0; };
    auto check_program = [&](uint32_t p, Bad& bad) {
        const DevProgram& d = v.programs[p];
        const uint32_t n_slots = v.n_slots[p], n_blocks = v.n_blocks[p];
        const bool tracked = v.tracked[p] != 0;
        const txq_op* o = ops + d.first_op;
        // a dense slot: an existing block id; its index inside A^(k-1) for untracked blocks (a tracked block's capacity is only
        // known to the session: plan_units checks those)
        auto slot_ok = [&](uint32_t s) {
            if (s & 0x80000000u) return false;
            if (!(s & TXQ_DENSE_SLOT_BIT)) return s < n_slots;
            return ((s & ~TXQ_DENSE_SLOT_BIT) >> TXQ_DENSE_BLOCK_SHIFT) < n_blocks && (tracked || (s & TXQ_DENSE_INDEX_MASK) < v.block_slots);
        };
        auto block_ok = [&](uint32_t s) {
            return !(s & 0x80000000u) && (s & TXQ_DENSE_SLOT_BIT) && (s & TXQ_DENSE_INDEX_MASK) == 0 && ((s & ~TXQ_DENSE_SLOT_BIT) >> TXQ_DENSE_BLOCK_SHIFT) < n_blocks;
        };
        for (uint32_t i = 0; i < d.n_ops; ++i) {
            int kind = 0;
            if (o[i].kmer == TXQ_DENSE_OP) {
                if (!v3 || o[i].dst >= v.n_dense || d.n_levels == 0) kind = 4;
                else {
                    const txq_dense_op& x = dops[o[i].dst];
                    const uint32_t code_mask = v.dense.A >= 32 ? 0xFFFFFFFFu : ((1u << v.dense.A) - 1u);
                    bool ok = x.kind <= TXQ_DENSE_FILL;
                    if (ok) ok = ((x.reserved & TXQ_DENSE_TRACKED) != 0) == (v.tracked[p] != 0) && (x.reserved & ~(TXQ_DENSE_TRACKED | TXQ_DENSE_NOPROBE)) == 0;
                    if (ok && (x.reserved & TXQ_DENSE_NOPROBE)) ok = x.kind == TXQ_DENSE_STEP && v.tracked[p] != 0;  // (only the pushed steps of tracked programs)
                    if (ok && x.kind != TXQ_DENSE_REDUCE) ok = block_ok(x.dst);
                    if (ok && (x.kind == TXQ_DENSE_STEP || x.kind == TXQ_DENSE_REDUCE)) ok = block_ok(x.src);
                    if (ok && x.kind == TXQ_DENSE_FILL) ok = !(x.src & TXQ_DENSE_SLOT_BIT) && x.src < n_slots;
                    if (ok && (x.kind != TXQ_DENSE_ZERO || x.r_mask))
                        for (uint32_t j = 0; ok && j < v.dense.pos; ++j) ok = (x.shape[j] & ~code_mask) == 0;
                    if (ok && x.kind == TXQ_DENSE_STEP) ok = x.src != x.dst && (x.r_mask & ~code_mask) == 0;
                    if (ok && x.kind == TXQ_DENSE_REDUCE) ok = slot_ok(x.dst) && x.dst != TXQ_SLOT_ZERO && x.dst != TXQ_SLOT_ONES && !(tracked && (x.dst & TXQ_DENSE_SLOT_BIT));
                    if (ok && x.kind == TXQ_DENSE_ZERO && tracked) {  // (re)creates the block: geometry in shape[], capacity in src
                        uint64_t entries = 1;
                        for (uint32_t j = 0; j < v.dense.pos; ++j) entries *= (uint64_t)__builtin_popcount(x.shape[j]);
                        ok = entries >= 1 && entries <= x.src && x.src <= (1u << TXQ_DENSE_BLOCK_SHIFT);
                    }
                    if (!ok) kind = 4;
                    v.has_dense[p] = 1;
                }
            } else if (!slot_ok(o[i].dst) || !slot_ok(o[i].a) || !slot_ok(o[i].b)) kind = 1;
            else if (o[i].dst == TXQ_SLOT_ZERO || o[i].dst == TXQ_SLOT_ONES) kind = 2;
            else if (o[i].kmer != TXQ_NO_KMER && o[i].kmer >= v.n_kmers) kind = 3;
            else if ((o[i].dst | o[i].a | o[i].b) & TXQ_DENSE_SLOT_BIT) {  // an ordinary op on block entries: the program runs level by level, like one with dense ops
                if (d.n_levels == 0) kind = 4;
                v.has_dense[p] = 1;
            }
            if (kind) { if (p < bad.program) bad = Bad{p, i, kind}; return; }
        }
    };
    Bad bad;
    unsigned n_threads = v.n_ops >= (1u << 20) ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    if (n_threads <= 1) {
        for (uint32_t p = 0; p < n_programs && bad.program == 0xFFFFFFFFu; ++p) check_program(p, bad);
    } else {
        std::vector<Bad> found(n_threads);
        std::atomic<uint32_t> next{0};
        std::vector<std::thread> workers;
        for (unsigned t = 0; t < n_threads; ++t)
            workers.emplace_back([&, t]() {
                for (uint32_t p; (p = next.fetch_add(1)) < n_programs;) check_program(p, found[t]);
            });
        for (auto& w : workers) w.join();
        for (const Bad& b : found) if (b.program < bad.program) bad = b;
    }
    if (bad.program != 0xFFFFFFFFu) {
        static const char* const what[] = {"", "slot out of range", "writes a constant slot", "k-mer index out of range", "malformed dense op"};
        return fail(TXQ_ERR_PROGRAM, "program %u op %u: %s", bad.program, bad.op, what[bad.kind]);
    }
    *out = std::move(v);
    return TXQ_OK;
}

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

Session::~Session() {
    if (kn.trace)
        fprintf(stderr, "[txq] session: %zu programs, %zu stages, %.1f MB uploaded, %.1f MB of slots; validate %.3f s, upload %.3f s, device+sync %.3f s; "
                        "(regions %.3f, plan %.3f, wait for the staging set %.3f, buffers %.3f) "
                        "%zu levels, %zu unit launches (%zu units), %zu dense launches (%zu tiles), step rows: %s\n",
                n_programs, n_stages, bytes_uploaded / 1e6, arena_words * 8 / 1e6, t_validate, t_upload, t_device, t_grow, t_plan, t_wait, t_alloc, n_levels, n_unit_launches, n_units,
                n_dense_launches, n_dense_tiles, row_source);
    if (kn.trace && n_step_pairs)
        fprintf(stderr, "[txq]   dense work: %llu predecessor visits for %llu destination suffixes, %llu slots zeroed, %llu entries reduced; mask %u words\n",
                (unsigned long long)n_step_pairs, (unsigned long long)n_step_suffixes, (unsigned long long)n_zero_slots, (unsigned long long)n_reduce_entries, W);
    if (kn.trace && n_beside) fprintf(stderr, "[txq]   %zu stage(s) ran beside the previous one (second stream)\n", n_beside);
    if (kn.trace && n_blocks_made + n_block_memsets + n_blocks_relisted)
        fprintf(stderr, "[txq]   dense blocks: %zu made (%.1f MB in all; %zu chunks of block memory, %.2f ms in hipMalloc), %zu cleared for tracked programs, %zu taken over as a tracked program left them; %zu sparse launches (%zu groups)\n",
                n_blocks_made, block_bytes_made / 1e6, block_chunks.size(), block_alloc_seconds * 1e3, n_block_memsets, n_blocks_relisted, n_sparse_launches, n_sparse_groups);
    if (aux) --aux->open_sessions;
    if (ix) --ix->open_sessions;
    for (Index::StagingSet& t : set)  // nothing of the session may still be running when its buffers change hands
        if (t.pending) { (void)hipEventSynchronize(t.done); t.pending = false; }
    if (d_step_ctr) {
        unsigned long long c[4] = {0, 0, 0, 0};
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(c, d_step_ctr, sizeof c, hipMemcpyDeviceToHost);
        (void)hipFree(d_step_ctr);
        if (c[0] && ix) {
            // algorithmic bytes of the pushed steps: per live entry its list index and its mask, per item and
            // residue the rows' 16-byte (8-byte) pieces, per non-empty product a 16-byte read-modify-write of the destination
            const uint64_t piece = W % 2 == 0 ? 16 : 8;
            const uint64_t rows_per_unit = ix->kmer_table ? 1 : (ix->is_hibf ? ix->tree_hash_max : ix->ibf[0].hash_funs);
            const double bytes = (double)c[0] * (4 + 8.0 * W) + (double)c[1] * rows_per_unit * 8.0 * W + (double)c[2] * 2 * piece + (double)c[3] * 8;
            fprintf(stderr, "[txq]   sparse steps: %llu live entries, %llu units (entry x residue), %llu non-empty %llu-byte products, %llu units that left a bit; "
                            "algorithmic bytes %.0f (per entry its list index and mask, per unit %llu row segment(s) of %u B, per product a read-modify-write, "
                            "per unit with a bit one bitmap word)\n", c[0], c[1], c[2], (unsigned long long)piece, c[3], bytes, (unsigned long long)rows_per_unit, W * 8);
        }
    }
    const double t_retire = now_s();
    for (void* p : retired) (void)hipFree(p);
    struct Lap {  // TXQ_TRACE: what releasing the session costs (the members' own destructors come after this)
        bool on; double t0; size_t retired;
        ~Lap() { if (on) fprintf(stderr, "[txq]   release: %zu outgrown staging buffers freed, then %.2f ms for buffers and blocks going back to the index\n", retired, (now_s() - t0) * 1e3); }
    } lap{kn.trace, now_s(), retired.size()};
    if (kn.trace && !retired.empty()) fprintf(stderr, "[txq]   release: hipFree of outgrown staging buffers %.2f ms\n", (lap.t0 - t_retire) * 1e3);
    if (owns_cache && ix) {  // hand the buffers back for the next session (the chunks up to a total of kArenaKeepBytes)
        Index::SessionCache& c = ix->session_cache;
        size_t kept = 0;
        for (const Index::ArenaChunk& k : chunks) {
            if (kept + k.cap * 8 <= Index::kArenaKeepBytes) { c.chunks.push_back(k); kept += k.cap * 8; }
            else (void)hipFree(k.p);
        }
        // the blocks: every one this session holds goes into the index's pool (a tracked program's are all zero outside their
        // lists), unless a stage failed (their state is unknown) or slots and blocks together outgrow what an index keeps
        size_t block_bytes = 0;
        for (const Index::ArenaChunk& k : block_chunks) block_bytes += k.cap * 8;
        if (!failed && kept + block_bytes <= Index::kArenaKeepBytes) {
            c.block_chunks.swap(block_chunks);
            c.block_cur = bcur;
            c.block_used = bused;
            c.blocks_W = W;
            c.blocks.swap(pool);
            for (size_t p = 0; p < blocks.size(); ++p)
                for (const DenseBlock& b : blocks[p])
                    if (b.p) c.blocks.put(DenseBlock{b.p, b.cap, (uint8_t)(tracked[p] ? kListed : kGarbage)});
            c.blocks.absorb(free_blocks);
            for (const std::vector<DenseBlock>& v : given_back)
                for (const DenseBlock& b : v) c.blocks.put(b);
        } else
            for (const Index::ArenaChunk& k : block_chunks) (void)hipFree(k.p);
        c.set[0] = set[0];
        c.set[1] = set[1];
        c.upload = upload;
        c.side = side;
        c.in_use = false;
        return;
    }
    for (const Index::ArenaChunk& k : chunks) (void)hipFree(k.p);
    for (const Index::ArenaChunk& k : block_chunks) (void)hipFree(k.p);
    for (Index::StagingSet& t : set) {
        if (t.done) (void)hipEventDestroy(t.done);
        for (void* p : {(void*)t.d_blob, (void*)t.d_aux, (void*)t.d_masks}) if (p) (void)hipFree(p);
    }
    if (upload) (void)hipStreamDestroy(upload);
    if (side) (void)hipStreamDestroy(side);
}


// bump allocation of `words` 64-bit words of slot storage (an even number wherever W is even: 16-byte lanes)
static int arena_alloc(Session& s, size_t words, uint64_t** out) {
    while (s.cur < s.chunks.size() && s.chunk_used + words > s.chunks[s.cur].cap) { ++s.cur; s.chunk_used = 0; }  // adopted chunks
    if (s.cur >= s.chunks.size()) {
        // 8 MiB first, then as much again as the session already holds (at least 64 MiB): few, large chunks
        size_t cap = s.chunks.empty() ? (size_t)1 << 20 : std::max((size_t)8 << 20, s.arena_words);
        if (words > cap) cap = words;
        uint64_t* c = nullptr;
        TXQ_HIP(hipMalloc((void**)&c, cap * 8));
        s.chunks.push_back(Index::ArenaChunk{c, cap});
        s.arena_words += cap;
        s.cur = s.chunks.size() - 1;
        s.chunk_used = 0;
    }
    *out = s.chunks[s.cur].p + s.chunk_used;
    s.chunk_used += words;
    return TXQ_OK;
}

int session_begin(Index& ix, size_t n_programs, Session** out) {
    if (n_programs >> 30) return fail(TXQ_ERR_ARG, "too many programs");
    Session* s = new (std::nothrow) Session();
    if (!s) return fail(TXQ_ERR_NOMEM, "out of host memory");
    s->ix = &ix;
    s->kn = knobs();
    ++ix.open_sessions;
    s->n_programs = n_programs;
    s->vspace = ix.layout_order(s->kn);  // a general HIBF: the session's masks are rows in layout order (txq_internal.hpp VChunk)
    s->W = s->vspace ? ix.v_words : (uint32_t)ix.shard_words;
    s->base.assign(2 * n_programs, nullptr);
    s->cap.assign(n_programs, 0);
    s->blocks.assign(n_programs, {});
    s->tracked.assign(n_programs, 0);
    Index::SessionCache& c = ix.session_cache;
    if (!c.in_use) {  // adopt the previous session's buffers
        c.in_use = true;
        s->owns_cache = true;
        s->chunks.swap(c.chunks);
        for (const Index::ArenaChunk& k : s->chunks) s->arena_words += k.cap;
        s->block_chunks.swap(c.block_chunks);
        for (const Index::ArenaChunk& k : s->block_chunks) s->block_arena_words += k.cap;
        if (c.blocks_W == s->W) {  // the pooled blocks fit this session's masks: take them over, go on allocating behind them
            s->pool.swap(c.blocks);
            s->bcur = c.block_cur;
            s->bused = c.block_used;
        }  // (else: another mask width — the chunks are reused from their beginning)
        s->set[0] = c.set[0];
        s->set[1] = c.set[1];
        s->upload = c.upload;
        s->side = c.side;
        c = Index::SessionCache{};
        c.in_use = true;
    }
    s->last_stage.assign(n_programs, 0);
    if (s->kn.trace && hipMalloc((void**)&s->d_step_ctr, 4 * sizeof(unsigned long long)) == hipSuccess)
        (void)hipMemset(s->d_step_ctr, 0, 4 * sizeof(unsigned long long));
    else { (void)hipGetLastError(); s->d_step_ctr = nullptr; }
    const double t_streams = now_s();
    for (hipStream_t* st : {&s->upload, &s->side})
        if (!*st && !(*st = take_spare_stream(ix.device))) {
            hipError_t e = hipStreamCreateWithFlags(st, hipStreamNonBlocking);
            if (e != hipSuccess) { delete s; return fail_hip(e, "hipStreamCreate(session)"); }
        }
    const double t_events = now_s();
    for (Index::StagingSet& t : s->set)
        if (!t.done) {
            hipError_t e = hipEventCreateWithFlags(&t.done, hipEventDisableTiming);
            if (e != hipSuccess) { delete s; return fail_hip(e, "hipEventCreate(session)"); }
        }
    if (s->kn.trace) fprintf(stderr, "[txq] session begin: streams %.3f ms, events %.3f ms\n", (t_events - t_streams) * 1e3, (now_s() - t_events) * 1e3);
    *out = s;
    return TXQ_OK;
}

// (re)size the programs' slot regions to what the stage needs; a grown region keeps its contents
// (host side only: the caller uploads `moves` and the base table with the stage and launches move_regions_kernel).
// Dense blocks: an untracked program gets the blocks it counts (A^(k-1) entries each); a tracked program gets a block
// when a ZERO of this stage creates it, with the capacity the op names (a block id keeps its capacity).  Blocks come
// from those that finished programs gave back, or from the arena; `to_clear` = blocks that go to a tracked program and
// must be all zero first (the caller memsets them on the stage's stream).
static size_t block_alloc_words(uint32_t cap, uint32_t W) { return ((size_t)cap * W + block_meta_words(cap) + 1) & ~(size_t)1; }

// the blocks' own arena (kept with the index between sessions together with the pool of blocks inside it)
static int block_arena_alloc(Session& s, size_t words, uint64_t** out) {
    while (s.bcur < s.block_chunks.size() && s.bused + words > s.block_chunks[s.bcur].cap) { ++s.bcur; s.bused = 0; }
    if (s.bcur >= s.block_chunks.size()) {
        size_t cap = std::max((size_t)8 << 20, s.block_arena_words);  // 64 MiB first, then as much again as there is
        if (words > cap) cap = words;
        uint64_t* c = nullptr;
        const double t0 = now_s();
        TXQ_HIP(hipMalloc((void**)&c, cap * 8));
        s.block_alloc_seconds += now_s() - t0;
        s.block_chunks.push_back(Index::ArenaChunk{c, cap});
        s.block_arena_words += cap;
        s.bcur = s.block_chunks.size() - 1;
        s.bused = 0;
    }
    *out = s.block_chunks[s.bcur].p + s.bused;
    s.bused += words;
    return TXQ_OK;
}

static int take_block(Session& s, uint32_t cap, Session::DenseBlock* out) {
    for (auto* from : {&s.free_blocks, &s.pool})  // given back in this session; left by earlier sessions on this index
        if (from->take(cap, out)) return TXQ_OK;
    Session::DenseBlock b{nullptr, cap, Session::kGarbage};
    if (int rc = block_arena_alloc(s, block_alloc_words(cap, s.W), &b.p)) return rc;
    ++s.n_blocks_made;
    s.block_bytes_made += block_alloc_words(cap, s.W) * 8;
    *out = b;
    return TXQ_OK;
}

static int grow_slot_regions(Session& s, const BlobView& bv, const unsigned char* blob, std::vector<uint32_t>* fresh, std::vector<RegionMove>* moves_out,
                             std::vector<std::pair<uint64_t*, size_t>>* to_clear) {
    std::vector<RegionMove>& moves = *moves_out;
    if (bv.block_slots) {
        if (!s.block_slots) s.block_slots = bv.block_slots;
        else if (s.block_slots != bv.block_slots)
            return fail(TXQ_ERR_PROGRAM, "the block size changed within a session (%u -> %u slots)", s.block_slots, bv.block_slots);
    }
    // Blocks given back two stages ago serve other programs now: whatever used them has finished (a stage waits for the
    // stage before the previous one, whose staging set it takes over), so a recycled block ties its new owner to nobody.
    for (const Session::DenseBlock& b : s.given_back[1]) s.free_blocks.put(b);
    s.given_back[1].swap(s.given_back[0]);
    s.given_back[0].clear();
    // a program that reports no dense blocks any more is finished with them
    if (bv.block_slots)
        for (size_t p = 0; p < s.n_programs; ++p)
            if (bv.n_blocks[p] == 0 && !s.blocks[p].empty()) {
                for (const Session::DenseBlock& b : s.blocks[p])
                    if (b.p) {
                        s.given_back[0].push_back(Session::DenseBlock{b.p, b.cap, (uint8_t)(s.tracked[p] ? Session::kListed : Session::kGarbage)});
                        --s.n_blocks_live;
                    }
                s.blocks[p].clear();
                s.base[s.n_programs + p] = nullptr;
            }
    const txq_op* ops = (const txq_op*)(blob + bv.ops_offset);
    const txq_dense_op* dops = bv.n_dense ? (const txq_dense_op*)(blob + bv.dense_offset) : nullptr;
    for (size_t p = 0; p < s.n_programs; ++p) {
        // a program gets its region with its first ops (a query of a later wave would otherwise get eight slots now and
        // outgrow them — a move, tied to this stage's init kernel — the moment it begins)
        const uint32_t need = s.cap[p] || bv.programs[p].n_ops ? bv.n_slots[p] : 0;
        if (need > s.cap[p]) {
            uint32_t cap = s.cap[p] ? s.cap[p] * 2 : 8;
            if (cap < need) cap = need;
            uint64_t* region = nullptr;
            if (int rc = arena_alloc(s, (size_t)cap * s.W, &region)) return rc;
            if (s.cap[p]) moves.push_back(RegionMove{region, s.base[p], (size_t)s.cap[p] * s.W});
            else fresh->push_back((uint32_t)p);
            s.base[p] = region;
            s.cap[p] = cap;
        }
        const size_t bneed = bv.block_slots ? bv.n_blocks[p] : 0;
        if (!bneed) continue;
        if (s.blocks[p].empty()) s.tracked[p] = bv.tracked[p];
        else if (s.tracked[p] != bv.tracked[p]) return fail(TXQ_ERR_PROGRAM, "program %zu: tracked and untracked blocks in one program", p);
        if (s.blocks[p].size() < bneed) s.blocks[p].resize(bneed, Session::DenseBlock{nullptr, 0, Session::kGarbage});
        if (!s.tracked[p]) {
            for (Session::DenseBlock& b : s.blocks[p])
                if (!b.p) {
                    if (int rc = take_block(s, bv.block_slots, &b)) return rc;
                    ++s.n_blocks_live;
                }
            continue;
        }
        if (!bv.has_dense[p]) continue;
        const DevProgram& d = bv.programs[p];
        for (uint32_t i = 0; i < d.n_ops; ++i) {
            const txq_op& o = ops[d.first_op + i];
            if (o.kmer != TXQ_DENSE_OP || dops[o.dst].kind != TXQ_DENSE_ZERO) continue;
            const txq_dense_op& z = dops[o.dst];
            Session::DenseBlock& b = s.blocks[p][(z.dst & ~TXQ_DENSE_SLOT_BIT) >> TXQ_DENSE_BLOCK_SHIFT];
            if (b.p) {
                if (b.cap != z.src) return fail(TXQ_ERR_PROGRAM, "program %zu: a tracked block changed its capacity (%u -> %u entries)", p, b.cap, z.src);
                continue;
            }
            if (int rc = take_block(s, z.src, &b)) return rc;
            ++s.n_blocks_live;
            // a block a tracked program left behind is all zero outside its list, and the ZERO that creates the block here
            // clears what is listed (sparse_plan_kernel resets the count, the chunks clear entries and bitmap bits): as it is
            if (b.state == Session::kListed) ++s.n_blocks_relisted;
            else to_clear->emplace_back(b.p, block_alloc_words(b.cap, s.W) * 8);
            b.state = Session::kGarbage;  // (what it is while its program runs; tracked[p] decides what it is given back as)
        }
    }
    return TXQ_OK;
}

// Big level-scheduled programs, and every program with dense ops, leave the one-workgroup-per-program kernel:
// their ops are cut into units per dependency level (units of level l, all programs, are contiguous in `units`),
// their dense ops into tiles, and every level becomes one launch of each kind over the whole GPU.
// Returns the number of programs left to exec_kernel.
struct LevelPlan {
    size_t units = 0, tiles = 0, hsteps = 0, sparse = 0, sparse_chunks = 0;
    // split_steps (flat indexes, tables of k-mer masks): the level's sparse groups are ordered [others | STEPs]; the first
    // sparse_misc go to the sparse_kernel without step code, the STEPs to the one with it (sparse_chunks counts the others' chunks then)
    size_t sparse_misc = 0, step_chunks = 0;
};
// hibf: STEP tiles go to their own list (`hsteps`, with the number of predecessors per suffix in `hstep_na`): on an
// HIBF a step is three launches (dense_hibf_*), not a tile of dense_kernel.
// The dense ops of tracked programs become sparse groups (one per op; sparse_kernel), and every dense op's blocks are
// resolved to pointers here (`optr`, indexed like the stage's dense table).
static size_t plan_units(const Session& s, BlobView& bv, const unsigned char* blob, uint32_t W, uint32_t G_dense, bool hibf, bool split_steps, std::vector<ExecUnit>* units,
                         std::vector<TileGroup>* groups, size_t* n_tiles, uint64_t (*work)[4], std::vector<DenseTile>* hsteps, std::vector<uint32_t>* hstep_na,
                         std::vector<SparseGroup>* sparse, std::vector<DenseOpPtr>* optr, std::vector<LevelPlan>* plan) {
    const uint32_t per_unit = unit_ops(W);
    const uint32_t* levels_host = bv.n_levels ? (const uint32_t*)(blob + bv.levels_offset) : nullptr;
    const txq_op* ops = (const txq_op*)(blob + bv.ops_offset);
    const txq_dense_op* dops = bv.n_dense ? (const txq_dense_op*)(blob + bv.dense_offset) : nullptr;
    optr->assign(bv.n_dense, DenseOpPtr{nullptr, nullptr, 0, 0});
    std::vector<std::vector<ExecUnit>> per_level;
    std::vector<std::vector<TileGroup>> groups_level;
    std::vector<std::vector<DenseTile>> hsteps_level;
    std::vector<std::vector<SparseGroup>> sparse_level, step_level;  // (step_level: the STEP groups when split_steps)
    std::vector<size_t> sparse_chunks, step_chunks;
    // entries per tile: every lane-group set of the workgroup gets two destination suffixes of a step (TXQ_DENSE_TILE_ROUNDS)
    const uint32_t step_tile = (uint32_t)s.kn.dense_tile_rounds * (256 / (G_dense ? G_dense : 1));
    size_t n_small = 0;
    int bad_program = -1;
    for (size_t p = 0; p < bv.programs.size(); ++p) {
        DevProgram& d = bv.programs[p];
        const bool dense = bv.has_dense[p] != 0;
        // small = less work than a unit launch is worth: 2048 ops of a 1024-bin index, 32 ops at 65536 bins
        if (!dense && (d.n_levels == 0 || (uint64_t)d.n_ops * W < 2048u * 16u)) { n_small += d.n_ops != 0; continue; }
        if (per_level.size() < d.n_levels) {
            per_level.resize(d.n_levels); groups_level.resize(d.n_levels); hsteps_level.resize(d.n_levels);
            sparse_level.resize(d.n_levels); sparse_chunks.resize(d.n_levels, 0);
            step_level.resize(d.n_levels); step_chunks.resize(d.n_levels, 0);
        }
        // (validate_blob has checked that block operands name existing block ids; grow_slot_regions has given the program its
        // blocks — a tracked block exists once a ZERO has created it: an op on one that was never created is refused here)
        auto block_of = [&](uint32_t slot) -> const Session::DenseBlock& {
            const Session::DenseBlock& b = s.blocks[p][(slot & ~TXQ_DENSE_SLOT_BIT) >> TXQ_DENSE_BLOCK_SHIFT];
            if (!b.p) bad_program = (int)p;
            return b;
        };
        auto slot_of = [&](uint32_t slot) -> uint64_t* {
            if (!(slot & TXQ_DENSE_SLOT_BIT)) return s.base[p] + (size_t)slot * W;
            const Session::DenseBlock& b = block_of(slot);
            if ((slot & TXQ_DENSE_INDEX_MASK) >= b.cap) bad_program = (int)p;
            return b.p + (size_t)(slot & TXQ_DENSE_INDEX_MASK) * W;
        };
        const bool check_slots = dense && bv.tracked[p];  // ordinary ops on dense slots of tracked blocks: inside the block's capacity?
        uint32_t begin = 0;
        for (uint32_t l = 0; l < d.n_levels; ++l) {
            const uint32_t end = levels_host[d.first_level + l];
            auto cut = [&](uint32_t from, uint32_t to) {  // a run of ordinary ops -> units
                for (uint32_t at = from; at < to; at += per_unit)
                    per_level[l].push_back(ExecUnit{(uint32_t)p, d.first_op + at, d.first_op + (to - at < per_unit ? to : at + per_unit)});
            };
            if (!dense) cut(begin, end);
            else {
                uint32_t run = begin;
                for (uint32_t i = begin; i < end; ++i) {
                    const txq_op& o = ops[d.first_op + i];
                    if (o.kmer != TXQ_DENSE_OP) {
                        if (check_slots)
                            for (uint32_t operand : {o.dst, o.a, o.b})
                                if (operand & TXQ_DENSE_SLOT_BIT) (void)slot_of(operand);
                        continue;
                    }
                    cut(run, i);
                    run = i + 1;
                    const txq_dense_op& x = dops[o.dst];
                    DenseOpPtr& q = (*optr)[o.dst];
                    if (x.kind == TXQ_DENSE_REDUCE) q.dst = slot_of(x.dst);
                    else { const Session::DenseBlock& b = block_of(x.dst); q.dst = b.p; q.dst_cap = b.cap; }
                    if (x.kind == TXQ_DENSE_STEP || x.kind == TXQ_DENSE_REDUCE) { const Session::DenseBlock& b = block_of(x.src); q.src = b.p; q.src_cap = b.cap; }
                    else if (x.kind == TXQ_DENSE_FILL) q.src = slot_of(x.src);
                    uint64_t shape_entries = 1;
                    for (uint32_t j = 0; j < bv.dense.pos; ++j) shape_entries *= (uint64_t)__builtin_popcount(x.shape[j]);
                    if (x.reserved & TXQ_DENSE_TRACKED) {  // work follows the block's live list (FILL: its shape)
                        const bool fixed = x.kind == TXQ_DENSE_FILL;
                        if (fixed && !shape_entries) continue;
                        const bool to_steps = split_steps && x.kind == TXQ_DENSE_STEP;
                        (to_steps ? step_level : sparse_level)[l].push_back(SparseGroup{o.dst, fixed ? (uint32_t)shape_entries : kNotFixed});
                        // most chunks this group can turn out to have: a list never outgrows its block
                        const uint64_t most = fixed ? shape_entries : x.kind == TXQ_DENSE_ZERO ? q.dst_cap : q.src_cap;
                        (to_steps ? step_chunks : sparse_chunks)[l] += (size_t)((most + kSparseChunk - 1) / kSparseChunk);
                        continue;
                    }
                    uint64_t entries = 1, per_tile = step_tile;
                    if (x.kind == TXQ_DENSE_ZERO || x.kind == TXQ_DENSE_FILL) {
                        per_tile = std::max<uint64_t>(1, 8192 / W);
                        entries = x.kind == TXQ_DENSE_ZERO && !x.r_mask ? bv.block_slots : shape_entries;
                    } else {
                        for (uint32_t j = x.kind == TXQ_DENSE_STEP ? 1 : 0; j < bv.dense.pos; ++j) entries *= (uint64_t)__builtin_popcount(x.shape[j]);
                        if (x.kind == TXQ_DENSE_STEP) entries *= (uint64_t)__builtin_popcount(x.r_mask) * (__builtin_popcount(x.shape[0]) ? 1 : 0);
                        else per_tile = 1024;
                    }
                    if (x.kind == TXQ_DENSE_STEP) { (*work)[0] += entries * (uint64_t)__builtin_popcount(x.shape[0]); (*work)[1] += entries; }
                    else if (x.kind == TXQ_DENSE_REDUCE) (*work)[3] += entries;
                    else (*work)[2] += entries;
                    const bool hstep = hibf && x.kind == TXQ_DENSE_STEP;
                    if (hstep) per_tile = 256;  // 256 suffixes x up to 32 predecessors: at most 8192 k-mers per tile
                    if (!hstep) {
                        if (entries) groups_level[l].push_back(TileGroup{(uint32_t)p, o.dst, (uint32_t)entries, (uint32_t)per_tile, 0});
                        continue;
                    }
                    for (uint64_t at = 0; at < entries; at += per_tile)
                        hsteps_level[l].push_back(DenseTile{(uint32_t)p, o.dst, (uint32_t)at, (uint32_t)std::min<uint64_t>(per_tile, entries - at)});
                }
                cut(run, end);
            }
            begin = end;
        }
        d.n_ops = 0;  // the per-program kernel skips it
    }
    if (bad_program >= 0) {
        (void)fail(TXQ_ERR_PROGRAM, "program %d: an op on a dense block that no ZERO has created, or beyond its capacity", bad_program);
        return (size_t)-1;
    }
    plan->resize(per_level.size());
    for (size_t l = 0; l < per_level.size(); ++l) {
        (*plan)[l].units = per_level[l].size();
        size_t level_tiles = 0;
        for (TileGroup& g : groups_level[l]) {
            g.first_tile = *n_tiles + level_tiles;
            level_tiles += (g.entries + g.per_tile - 1) / g.per_tile;
        }
        (*plan)[l].tiles = level_tiles;
        *n_tiles += level_tiles;
        groups->insert(groups->end(), groups_level[l].begin(), groups_level[l].end());
        (*plan)[l].hsteps = hsteps_level[l].size();
        (*plan)[l].sparse = sparse_level[l].size() + step_level[l].size();
        (*plan)[l].sparse_misc = sparse_level[l].size();
        (*plan)[l].sparse_chunks = sparse_chunks[l];
        (*plan)[l].step_chunks = step_chunks[l];
        sparse->insert(sparse->end(), sparse_level[l].begin(), sparse_level[l].end());
        sparse->insert(sparse->end(), step_level[l].begin(), step_level[l].end());
        units->insert(units->end(), per_level[l].begin(), per_level[l].end());
        for (const DenseTile& t : hsteps_level[l]) {
            hsteps->push_back(t);
            hstep_na->push_back((uint32_t)__builtin_popcount(dops[t.op].shape[0]));
        }
    }
    return n_small;
}

// rows_of(H) makes the kernel's row source for H hash functions (FlatRows / TreeRows)
template <bool WIDE, template <int, bool> class ROWS, class MAKE>
static hipError_t launch_dense(int ua, uint32_t hash_funs, MAKE rows_of, const DenseTile* tiles, size_t n_tiles, const txq_dense_op* dops, const DenseOpPtr* optr,
                               uint64_t* const* base, uint32_t n_programs, uint32_t W, uint32_t G, uint32_t SL, const DenseParams& P, const LevelUnits& U, hipStream_t st) {
    const size_t grid = n_tiles + U.n_units;
    // ua: predecessors in flight per lane (TXQ_DENSE_UNROLL: A/B knob)
#define TXQ_DENSE(H) \
    do { \
        ROWS<H, WIDE> rows{}; \
        rows_of(rows); \
        if (ua >= 5) dense_kernel<H, WIDE, 5, ROWS<H, WIDE>><<<(unsigned)grid, 256, 0, st>>>(rows, tiles, dops, optr, base, n_programs, W, G, SL, P, U); \
        else if (ua <= 2) dense_kernel<H, WIDE, 2, ROWS<H, WIDE>><<<(unsigned)grid, 256, 0, st>>>(rows, tiles, dops, optr, base, n_programs, W, G, SL, P, U); \
        else dense_kernel<H, WIDE, 3, ROWS<H, WIDE>><<<(unsigned)grid, 256, 0, st>>>(rows, tiles, dops, optr, base, n_programs, W, G, SL, P, U); \
    } while (0)
    switch (hash_funs) {
        case 1: TXQ_DENSE(1); break;
        case 2: TXQ_DENSE(2); break;
        case 3: TXQ_DENSE(3); break;
        case 4: TXQ_DENSE(4); break;
        case 5: TXQ_DENSE(5); break;
        default: return hipErrorInvalidValue;
    }
#undef TXQ_DENSE
    return hipGetLastError();
}

// the level's sparse groups [groups, groups + n_groups) (n_groups <= kMaxSparseGroups): `grid` workgroups share their chunks out
template <bool WIDE, template <int, bool> class ROWS, class MAKE>
static hipError_t launch_sparse(uint32_t hash_funs, MAKE rows_of, const SparseGroup* groups, uint32_t n_groups, const uint32_t* counts, const uint32_t* prefix,
                                size_t grid, const txq_dense_op* dops, const DenseOpPtr* optr, uint64_t* const* base, uint32_t n_programs, uint32_t W, uint32_t G,
                                const DenseParams& P, const LevelUnits& U, uint32_t chunk, hipStream_t st) {
#define TXQ_SPARSE(H) \
    do { \
        ROWS<H, WIDE> rows{}; \
        rows_of(rows); \
        sparse_kernel<H, WIDE, ROWS<H, WIDE>><<<(unsigned)grid, 256, 0, st>>>(rows, groups, n_groups, counts, prefix, dops, optr, base, n_programs, W, G, P, U, chunk); \
    } while (0)
    switch (hash_funs) {
        case 1: TXQ_SPARSE(1); break;
        case 2: TXQ_SPARSE(2); break;
        case 3: TXQ_SPARSE(3); break;
        case 4: TXQ_SPARSE(4); break;
        case 5: TXQ_SPARSE(5); break;
        default: return hipErrorInvalidValue;
    }
#undef TXQ_SPARSE
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void iota_kernel(uint64_t* __restrict__ v, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) v[i] = i;
}

// the STEP groups of a level on narrow masks (sparse_units_kernel)
template <bool WIDE, template <int, bool> class ROWS, class MAKE>
static hipError_t launch_sparse_units(uint32_t hash_funs, MAKE rows_of, const SparseGroup* groups, uint32_t n_groups, const uint32_t* counts, const uint32_t* prefix,
                                      size_t grid, const txq_dense_op* dops, const DenseOpPtr* optr, uint64_t* const* base, uint32_t n_programs, uint32_t W, uint32_t G,
                                      const StepParams& P, const LevelUnits& U, unsigned long long* ctr, int ua, hipStream_t st) {
    // ua: units in flight per lane group (TXQ_SPARSE_UNROLL: A/B knob; 3 fits the 128 VGPRs of four waves per SIMD up to h = 3)
#define TXQ_UNITS(H) \
    do { \
        ROWS<H, WIDE> rows{}; \
        rows_of(rows); \
        if (ua <= 2 || H >= 4) sparse_units_kernel<H, WIDE, 2, ROWS<H, WIDE>><<<(unsigned)grid, 256, 0, st>>>(rows, groups, n_groups, counts, prefix, dops, optr, base, n_programs, W, G, P, U, ctr); \
        else sparse_units_kernel<H, WIDE, 3, ROWS<H, WIDE>><<<(unsigned)grid, 256, 0, st>>>(rows, groups, n_groups, counts, prefix, dops, optr, base, n_programs, W, G, P, U, ctr); \
    } while (0)
    switch (hash_funs) {
        case 1: TXQ_UNITS(1); break;
        case 2: TXQ_UNITS(2); break;
        case 3: TXQ_UNITS(3); break;
        case 4: TXQ_UNITS(4); break;
        case 5: TXQ_UNITS(5); break;
        default: return hipErrorInvalidValue;
    }
#undef TXQ_UNITS
    return hipGetLastError();
}

// The table of all k-mers' masks of an index (Index::kmer_table), built once per index when the first stage with dense steps
// arrives: bulk_contains of every packed value below 2^(bits * k) — values with a residue code outside the alphabet included,
// nobody reads their rows — with the probe kernel itself; for an HIBF the descent's user-bin masks, so that dense steps on ANY
// tree whose table fits read one row per k-mer like a flat index's.  Returns whether the steps may read it.  (Synchronous: a
// later stage on the other stream reads the table, too.)
static bool ensure_kmer_table(Index& ix, const Knobs& kn, const DenseParams& P, uint32_t W, hipStream_t st) {
    const uint32_t vb = P.bits * P.k;
    if (kn.kmer_table_mb <= 0 || vb == 0 || vb > 24 || W != ix.shard_words) return false;
    std::lock_guard<std::mutex> lock(ix.table_mutex);  // two sessions of one index may arrive here at once: one builds, the other finds it
    if (ix.kmer_table) return ix.kmer_table_bits == vb;  // (one encoder per index; a session with another k gathers rows)
    const uint64_t n = 1ULL << vb, bytes = n * (uint64_t)W * 8;
    if (ix.kmer_table_refused || bytes > ((uint64_t)kn.kmer_table_mb << 20)) return false;
    const double t0 = now_s();
    uint64_t *table = nullptr, *values = nullptr;
    if (hipMalloc((void**)&table, bytes) != hipSuccess || hipMalloc((void**)&values, n * 8) != hipSuccess) {
        (void)hipGetLastError();
        if (table) (void)hipFree(table);
        ix.kmer_table_refused = true;
        return false;
    }
    iota_kernel<<<(unsigned)std::min<uint64_t>((n + 255) / 256, 4096), 256, 0, st>>>(values, n);
    hipError_t e = hipSuccess;
    if (ix.is_hibf) {  // membership_for(., 1) of every value, in user-bin order: any tree, whatever its shape
        // the descent keeps its frontiers in the INDEX's scratch: a stage of this or another session that is still descending
        // (hibf_probe on another stream) must have finished before this descent overwrites them — once per index
        e = hipDeviceSynchronize();
        if (e == hipSuccess && hibf_probe(ix, kn, values, n, table, nullptr, st) != TXQ_OK) e = hipErrorUnknown;
    } else e = launch_probe(ix.ibf[0], values, n, table, nullptr, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(values);
    if (e != hipSuccess) {
        (void)hipFree(table);
        ix.kmer_table_refused = true;
        return false;
    }
    ix.kmer_table = table;
    ix.kmer_table_bits = vb;
    if (kn.trace) fprintf(stderr, "[txq] table of all %llu k-mer masks (%u bits per k-mer): %.1f MB, built in %.2f ms\n", (unsigned long long)n, vb, bytes / 1e6, (now_s() - t0) * 1e3);
    return true;
}

int session_stage(Session& s, const void* blob_v, size_t bytes, const uint32_t* q_prog, const uint32_t* q_slot, size_t n_q,
                  uint8_t* alive, hipStream_t caller_stream) {
    Index& ix = *s.ix;
    const unsigned char* blob = (const unsigned char*)blob_v;
    BlobView bv;
    double t0 = now_s();
    if (int rc = validate_blob(blob, bytes, s.n_programs, s.kn, &bv)) return rc;
    s.t_validate += now_s() - t0;
    t0 = now_s();
    ++s.n_stages;
    s.bytes_uploaded += bytes;
    const BlobView* h = &bv;
    bool any_dense = false;
    for (uint8_t d : bv.has_dense) any_dense |= d != 0;
    // Dense steps through the index's table of all k-mers' masks, where it fits (TXQ_DENSE_TREE set: the tree paths are asked for).
    // A session on a general HIBF that would keep its masks in layout order goes back to user-bin order for it — in its first
    // stage, before anything has been laid out.
    // A session of a few queries does not build it (0.4-1 ms: more than a single query's steps cost); it uses one that is there.
    bool table = false;
    if (any_dense && (!ix.is_hibf || s.kn.dense_tree < 0) && (ix.kmer_table || (long long)s.n_programs >= s.kn.kmer_table_min)) {
        if (!s.vspace) table = ensure_kmer_table(ix, s.kn, bv.dense, s.W, s.upload);
        else if (s.n_stages == 1 && !s.aux && ensure_kmer_table(ix, s.kn, bv.dense, (uint32_t)ix.shard_words, s.upload)) {
            s.vspace = false;
            s.W = (uint32_t)ix.shard_words;
            table = true;
        }
    }
    const uint32_t W = s.W;
    for (size_t i = 0; i < n_q; ++i) {
        if (q_prog[i] >= s.n_programs) return fail(TXQ_ERR_ARG, "feedback query %zu: program out of range", i);
        if (q_slot[i] >= bv.n_slots[q_prog[i]]) return fail(TXQ_ERR_ARG, "feedback query %zu: slot out of range", i);
    }
    if (W == 0 || s.n_programs == 0) {
        for (size_t i = 0; i < n_q; ++i) alive[i] = 0;
        return TXQ_OK;
    }
    if (any_dense && !ix.is_hibf && (ix.ibf[0].bin_size >> 32))
        return fail(TXQ_ERR_PROGRAM, "dense ops need fewer than 2^32 rows");
    // Does this stage continue anything the previous stage — possibly still running — works on?  Programs with ops in both,
    // feedback questions, grown regions (moves), an HIBF that is descended or a d-gram index (scratch of the index) tie it
    // to the previous stage's stream; a stage of other programs only (the next wave of queries) runs beside it.
    bool continues = n_q != 0 || (ix.is_hibf && !ix.probes_interleaved(s.kn) && !s.vspace) || s.aux != nullptr;
    for (size_t p = 0; p < s.n_programs; ++p)
        if (bv.programs[p].n_ops) {
            continues = continues || s.last_stage[p] + 1 == s.n_stages;
            s.last_stage[p] = (uint32_t)s.n_stages;
        }
    // dense steps on a regular two-level HIBF run fused, too (TreeRows; TXQ_DENSE_TREE=0 sends them through the generic HIBF path)
    const bool tree = !table && index_fuses_tree_steps(ix, s.kn);
    bool any_tracked = false;
    for (size_t p = 0; p < s.n_programs; ++p) any_tracked |= bv.tracked[p] != 0 && bv.has_dense[p] != 0;
    const bool vspace = s.vspace;
    if (any_tracked && ix.is_hibf && !tree && !vspace && !table)
        return fail(TXQ_ERR_PROGRAM, "tracked blocks need an index whose dense steps run fused (txq_index_supports_dense() == 2)");
    std::vector<uint32_t> fresh;  // programs that got their first region: ZERO/ONES/RESULT need initialising
    std::vector<RegionMove> moves;
    std::vector<std::pair<uint64_t*, size_t>> to_clear;
    const double t_begin = t0;
    double t_mark[6] = {0, 0, 0, 0, 0, 0};  // (TXQ_TRACE_STAGES) where the host's time of this stage goes
    if (int rc = grow_slot_regions(s, bv, blob, &fresh, &moves, &to_clear)) return rc;
    s.t_grow += now_s() - t0;
    t_mark[0] = now_s();
    for (size_t i = 0; i < n_q; ++i)
        if (!s.base[q_prog[i]]) return fail(TXQ_ERR_ARG, "feedback query %zu: program %u has not run an op yet", i, q_prog[i]);

    // dense steps: 16-byte lanes where masks and rows allow it, G lanes per destination suffix
    if (any_dense) s.row_source = tree ? "regular tree, fused" : vspace ? "general tree in layout order, fused" : ix.is_hibf ? "HIBF descent" : "flat IBF, fused";
    if (any_dense && tree && ix.interleaved.words && ix.interleaved.shard_words == W && ix.root_node.bins <= 64 && s.kn.dense_tree < 0)
        s.row_source = "regular tree, interleaved children, fused";
    const int tree_knob = s.kn.dense_tree;  // 0: generic HIBF steps, 1: TreeRows, 2: TreeRowsByLane where it applies; -1 (default): best fit
    const bool interleaved = tree && ix.interleaved.words && ix.interleaved.shard_words == W && ix.root_node.bins <= 64 && tree_knob < 0;
    if (table) s.row_source = ix.is_hibf ? "HIBF through its table of all k-mers' masks" : "flat IBF through its table of all k-mers' masks";
    const bool wide = W % 2 == 0 && (vspace ? ix.v_chunk_words == 2 : table ? true : (interleaved ? ix.interleaved.stride % 2 == 0 : tree ? ix.child_row_words >= 2 : !ix.is_hibf && ix.ibf[0].stride % 2 == 0));
    uint32_t g_dense = 1;
    while (g_dense < 64 && g_dense < (wide ? W / 2 : W)) g_dense <<= 1;
    // ... and two such lane groups share the predecessors of one suffix (TXQ_DENSE_SLICES: A/B knob; on the bench batch
    // 1 / 2 / 4 / 8 slices took 34 / 30 / 34 / 44+ ms end to end: more slices shorten a tile's load chain but multiply the tiles)
    const uint32_t want_slices = (uint32_t)s.kn.dense_slices;
    uint32_t sl_dense = 1;
    while (sl_dense * 2 <= want_slices && g_dense * sl_dense * 2 <= 64) sl_dense <<= 1;
    std::vector<ExecUnit> units;
    std::vector<TileGroup> tile_groups;
    size_t n_tiles = 0;
    uint64_t work[4] = {0, 0, 0, 0};
    std::vector<DenseTile> hsteps;
    std::vector<uint32_t> hstep_na;
    std::vector<LevelPlan> plan;
    std::vector<SparseGroup> sparse_groups;
    std::vector<DenseOpPtr> optr;
    double t1 = now_s();
    // a level's STEP groups get a launch of their own (chunked by work, below); the ZERO / REDUCE / FILL groups run in the
    // sparse_kernel variant without step code (66 VGPRs)
    const bool split_steps = true;
    // ... and where a mask is a cache line or two, that launch shares the steps out by units (sparse_units_kernel; TXQ_SPARSE_STEPS=0:
    // sparse_kernel's rounds of entries, A/B and tests)
    const bool by_units = s.kn.sparse_steps && W <= kUnitStepWords && !vspace && !tree && (table || !ix.is_hibf);
    const size_t n_small = plan_units(s, bv, blob, W, g_dense * sl_dense, ix.is_hibf && !tree && !vspace && !table, split_steps, &units, &tile_groups, &n_tiles, &work, &hsteps, &hstep_na,
                                      &sparse_groups, &optr, &plan);
    if (n_small == (size_t)-1) return TXQ_ERR_PROGRAM;
    size_t n_sparse_launches = 0;
    for (const LevelPlan& lp : plan)
        n_sparse_launches += (lp.sparse_misc + kMaxSparseGroups - 1) / kMaxSparseGroups + (lp.sparse - lp.sparse_misc + kMaxSparseGroups - 1) / kMaxSparseGroups;
    // the stage's block table: per program with blocks a row [flags | block 0 | its capacity | block 1 | ..] (DenseRow)
    std::vector<uint64_t*> block_table;
    std::vector<size_t> row_of(s.n_programs, 0);
    for (size_t p = 0; p < s.n_programs; ++p)
        if (!s.blocks[p].empty()) {
            row_of[p] = block_table.size();
            block_table.push_back(reinterpret_cast<uint64_t*>((uintptr_t)(s.tracked[p] ? 1 : 0)));
            for (const Session::DenseBlock& b : s.blocks[p]) {
                block_table.push_back(b.p);
                block_table.push_back(reinterpret_cast<uint64_t*>((uintptr_t)b.cap));
            }
        }
    s.n_step_pairs += work[0]; s.n_step_suffixes += work[1]; s.n_zero_slots += work[2]; s.n_reduce_entries += work[3];
    // HIBF steps run in chunks of tiles whose masks fit the scratch (2 GiB): chunk c = tiles [chunk_first[c], chunk_first[c+1]),
    // never across a level; pair_base[tile] = first pair of the tile within its chunk
    std::vector<uint32_t> pair_base(hsteps.size(), 0);
    std::vector<size_t> chunk_first;
    std::vector<uint32_t> chunk_pairs;
    size_t most_pairs = 0;
    {
        const uint64_t budget = std::max<uint64_t>(((uint64_t)2 << 30) / ((uint64_t)W * 8), 8192);
        size_t at = 0;
        for (const LevelPlan& lp : plan) {
            uint64_t pairs = 0;
            for (size_t i = 0; i < lp.hsteps; ++i, ++at) {
                const uint64_t mine = (uint64_t)hsteps[at].count * hstep_na[at];
                if (i == 0 || pairs + mine > budget) {
                    if (!chunk_first.empty() && !chunk_pairs.empty()) most_pairs = std::max<size_t>(most_pairs, chunk_pairs.back());
                    chunk_first.push_back(at);
                    chunk_pairs.push_back(0);
                    pairs = 0;
                }
                pair_base[at] = (uint32_t)pairs;
                pairs += mine;
                chunk_pairs.back() = (uint32_t)pairs;
            }
        }
        for (uint32_t c : chunk_pairs) most_pairs = std::max<size_t>(most_pairs, c);
        chunk_first.push_back(hsteps.size());
    }

    // Staging set of this stage (the other one may still be read by the previous stage's kernels; this one was last used
    // two stages ago): blob, and aux = program table | fresh-program list | feedback queries | alive bytes | units | tiles |
    // HIBF steps | their pair bases | region moves | region bases.  Everything is copied on the upload stream and the host
    // waits for THOSE copies only (pageable sources, some of them locals) — not for the kernels of the previous stage.
    s.t_plan += now_s() - t1;
    t1 = now_s();
    t_mark[1] = t1;
    Index::StagingSet& S = s.set[(s.n_stages - 1) & 1];
    if (S.pending) {
        TXQ_HIP(hipEventSynchronize(S.done));
        S.pending = false;
    }
    s.t_wait += now_s() - t1;
    t1 = now_s();
    t_mark[2] = t1;
    size_t aux_bytes = 0;
    auto place = [&](size_t bytes) { const size_t at = aux_bytes; aux_bytes = (aux_bytes + bytes + 15) & ~(size_t)15; return at; };
    const size_t prog_bytes = s.n_programs * sizeof(DevProgram);
    const size_t at_progs = place(prog_bytes), at_fresh = place(fresh.size() * 4), at_qp = place(n_q * 4), at_qs = place(n_q * 4), at_alive = place(n_q);
    const size_t at_units = place(units.size() * sizeof(ExecUnit)), at_tiles = place(n_tiles * sizeof(DenseTile)), at_groups = place(tile_groups.size() * sizeof(TileGroup));
    const size_t at_hsteps = place(hsteps.size() * sizeof(DenseTile)), at_pair_base = place(hsteps.size() * 4);
    const size_t at_moves = place(moves.size() * sizeof(RegionMove)), at_base = place(2 * s.n_programs * sizeof(uint64_t*));
    const size_t at_optr = place(optr.size() * sizeof(DenseOpPtr)), at_bt = place(block_table.size() * sizeof(uint64_t*));
    const size_t at_sgroups = place(sparse_groups.size() * sizeof(SparseGroup)), at_scounts = place(sparse_groups.size() * 4);
    const size_t at_sprefix = place((sparse_groups.size() + n_sparse_launches) * 4);
    std::vector<RegionMove> clears;
    for (const auto& b : to_clear) clears.push_back(RegionMove{b.first, nullptr, b.second / 8});
    const size_t at_clears = place(clears.size() * sizeof(RegionMove));
    // a small stage (a single query: a few hundred bytes of blob, a dozen small tables) travels as ONE copy: the blob
    // rides behind the tables in `aux`
    const bool packed = aux_bytes + bytes <= ((size_t)256 << 10);
    const size_t at_blob = packed ? place(bytes) : 0;
    // The set's buffers are idle (S.done was waited for), but hipFree drains the WHOLE device — the previous stage, which this
    // stage may want to run beside: a buffer that has to grow (how the queries fall into waves depends on the host's timing, so
    // stage sizes differ from batch to batch) is replaced generously and the old one freed with the session.
    auto ensure_idle = [&](void** p, size_t* cap, size_t need) -> int {
        if (*p && *cap >= need) return TXQ_OK;
        if (*p) s.retired.push_back(*p);
        *p = nullptr; *cap = 0;
        const size_t want = std::max(need + need / 2, (size_t)1 << 20);
        hipError_t e = hipMalloc(p, want);
        if (e != hipSuccess) return fail_hip(e, "hipMalloc(staging set)");
        *cap = want;
        return TXQ_OK;
    };
    if (!packed)
        if (int rc = ensure_idle((void**)&S.d_blob, &S.cap_blob, (bytes + 7) & ~(size_t)7)) return rc;
    if (int rc = ensure_idle((void**)&S.d_aux, &S.cap_aux, aux_bytes + 16)) return rc;
    const unsigned char* dblob = packed ? S.d_aux + at_blob : S.d_blob;
    for (size_t p = 0; p < s.n_programs; ++p)  // (the aux buffer has its final address now)
        s.base[s.n_programs + p] = s.blocks[p].empty() ? nullptr : reinterpret_cast<uint64_t*>(reinterpret_cast<uint64_t**>(S.d_aux + at_bt) + row_of[p]);
    const size_t nk = h->n_kmers;
    // scratch of the INDEX that kernels in flight may still use: replacing it drains the device (ensure), so replace it generously
    auto ensure_scratch = [&](uint64_t** p, size_t* cap, size_t need) -> int {
        if (*p && *cap >= need) return TXQ_OK;
        return ensure((void**)p, cap, std::max(need + need / 2, (size_t)64 << 20));
    };
    if (int rc = ensure_idle((void**)&S.d_masks, &S.cap_masks, std::max((nk ? nk : 1) * (size_t)W * 8, (size_t)16 << 20))) return rc;
    uint64_t* const d_masks = S.d_masks;
    s.t_alloc += now_s() - t1;
    t_mark[3] = now_s();
    hipStream_t up = s.upload;
    if (packed) {
        s.host_aux.resize(aux_bytes);
        unsigned char* hb = s.host_aux.data();
        auto put = [&](size_t at, const void* src, size_t n) { if (n) std::memcpy(hb + at, src, n); };
        put(at_progs, bv.programs.data(), prog_bytes);
        put(at_fresh, fresh.data(), fresh.size() * 4);
        put(at_qp, q_prog, n_q * 4);
        put(at_qs, q_slot, n_q * 4);
        put(at_units, units.data(), units.size() * sizeof(ExecUnit));
        put(at_groups, tile_groups.data(), tile_groups.size() * sizeof(TileGroup));
        put(at_hsteps, hsteps.data(), hsteps.size() * sizeof(DenseTile));
        put(at_pair_base, pair_base.data(), hsteps.size() * 4);
        put(at_moves, moves.data(), moves.size() * sizeof(RegionMove));
        put(at_base, s.base.data(), 2 * s.n_programs * sizeof(uint64_t*));
        put(at_optr, optr.data(), optr.size() * sizeof(DenseOpPtr));
        put(at_bt, block_table.data(), block_table.size() * sizeof(uint64_t*));
        put(at_sgroups, sparse_groups.data(), sparse_groups.size() * sizeof(SparseGroup));
        put(at_clears, clears.data(), clears.size() * sizeof(RegionMove));
        put(at_blob, blob, bytes);
        TXQ_HIP(hipMemcpyAsync(S.d_aux, hb, aux_bytes, hipMemcpyHostToDevice, up));
    } else {
        auto send = [&](size_t at, const void* src, size_t n) -> hipError_t {
            return n ? hipMemcpyAsync(S.d_aux + at, src, n, hipMemcpyHostToDevice, up) : hipSuccess;
        };
        TXQ_HIP(hipMemcpyAsync(S.d_blob, blob, bytes, hipMemcpyHostToDevice, up));
        TXQ_HIP(send(at_progs, bv.programs.data(), prog_bytes));
        TXQ_HIP(send(at_fresh, fresh.data(), fresh.size() * 4));
        TXQ_HIP(send(at_qp, q_prog, n_q * 4));
        TXQ_HIP(send(at_qs, q_slot, n_q * 4));
        TXQ_HIP(send(at_units, units.data(), units.size() * sizeof(ExecUnit)));
        TXQ_HIP(send(at_groups, tile_groups.data(), tile_groups.size() * sizeof(TileGroup)));
        TXQ_HIP(send(at_hsteps, hsteps.data(), hsteps.size() * sizeof(DenseTile)));
        TXQ_HIP(send(at_pair_base, pair_base.data(), hsteps.size() * 4));
        TXQ_HIP(send(at_moves, moves.data(), moves.size() * sizeof(RegionMove)));
        TXQ_HIP(send(at_base, s.base.data(), 2 * s.n_programs * sizeof(uint64_t*)));
        TXQ_HIP(send(at_optr, optr.data(), optr.size() * sizeof(DenseOpPtr)));
        TXQ_HIP(send(at_bt, block_table.data(), block_table.size() * sizeof(uint64_t*)));
        TXQ_HIP(send(at_sgroups, sparse_groups.data(), sparse_groups.size() * sizeof(SparseGroup)));
        TXQ_HIP(send(at_clears, clears.data(), clears.size() * sizeof(RegionMove)));
    }
    DevProgram* d_progs = (DevProgram*)(S.d_aux + at_progs);
    uint32_t* d_fresh = (uint32_t*)(S.d_aux + at_fresh);
    uint32_t* d_qp = (uint32_t*)(S.d_aux + at_qp);
    uint32_t* d_qs = (uint32_t*)(S.d_aux + at_qs);
    uint8_t* d_alive = S.d_aux + at_alive;
    ExecUnit* d_units = (ExecUnit*)(S.d_aux + at_units);
    DenseTile* d_tiles = (DenseTile*)(S.d_aux + at_tiles);
    DenseTile* d_hsteps = (DenseTile*)(S.d_aux + at_hsteps);
    uint32_t* d_pair_base = (uint32_t*)(S.d_aux + at_pair_base);
    const DenseOpPtr* d_optr = (const DenseOpPtr*)(S.d_aux + at_optr);
    const SparseGroup* d_sgroups = (const SparseGroup*)(S.d_aux + at_sgroups);
    uint32_t* d_scounts = (uint32_t*)(S.d_aux + at_scounts);
    uint32_t* d_sprefix = (uint32_t*)(S.d_aux + at_sprefix);
    s.d_base = (uint64_t**)(S.d_aux + at_base);
    if (!hsteps.empty()) {
        if (int rc = ensure_scratch(&ix.scratch_dense_kmers, &ix.cap_dense_kmers, most_pairs * 8)) return rc;
        if (int rc = ensure_scratch(&ix.scratch_dense_masks, &ix.cap_dense_masks, most_pairs * (size_t)W * 8)) return rc;
    }
    t_mark[4] = now_s();
    TXQ_HIP(hipStreamSynchronize(up));
    s.t_upload += now_s() - t0;
    t0 = now_s();
    t_mark[5] = t0;

    const bool one_stream = s.kn.one_stream;  // A/B knob
    const Index::StagingSet& prev = s.set[s.n_stages & 1];
    const bool beside = !continues && moves.empty() && hsteps.empty() && prev.pending && s.n_stages > 1 && !one_stream;
    const int which = beside ? 1 - s.stream_of_last : s.stream_of_last;
    s.stream_of_last = which;
    if (beside) ++s.n_beside;
    if (s.kn.trace_stages)
        fprintf(stderr, "[txq] stage %zu: continues %d (questions %zu), moves %zu, previous pending %d -> stream %d\n", s.n_stages, (int)continues, n_q, moves.size(),
                (int)prev.pending, which);
    hipStream_t st = which ? s.side : caller_stream;
    if (!tile_groups.empty()) {
        make_tiles_kernel<<<(unsigned)tile_groups.size(), 256, 0, st>>>((const TileGroup*)(S.d_aux + at_groups), d_tiles);
        TXQ_HIP(hipGetLastError());
    }
    if (!clears.empty()) {  // blocks a tracked program takes over: all zero, list empty
        clear_blocks_kernel<<<dim3((unsigned)clears.size(), 16), 256, 0, st>>>((const RegionMove*)(S.d_aux + at_clears));
        TXQ_HIP(hipGetLastError());
        s.n_block_memsets += clears.size();
    }
    if (!moves.empty()) {  // after everything earlier stages launched on the regions, before anything of this stage
        move_regions_kernel<<<dim3((unsigned)moves.size(), 16), 256, 0, st>>>((const RegionMove*)(S.d_aux + at_moves));
        TXQ_HIP(hipGetLastError());
    }
    if (!fresh.empty()) {
        size_t blocks = (fresh.size() * W + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        init_slots_kernel<<<(unsigned)blocks, 256, 0, st>>>(s.d_base, d_fresh, (uint32_t)fresh.size(), W, ix.user_bins, ix.shard_word0, s.vspace ? ix.d_vleaf : nullptr);
    }
    const uint64_t* d_kmers = (const uint64_t*)(dblob + h->kmers_offset);
    const size_t n_aux = (size_t)h->n_aux_kmers, n_main = nk - n_aux;
    if (n_aux && !s.aux) return fail(TXQ_ERR_STATE, "the blob has auxiliary (d-gram) k-mers but the session has no auxiliary index");
    if (n_main) {
        if (vspace) {
            if (int rc = hibf_probe_layout_order(ix, d_kmers, n_main, d_masks, st)) return rc;
        } else if (ix.is_hibf) {
            if (int rc = hibf_probe(ix, s.kn, d_kmers, n_main, d_masks, nullptr, st)) return rc;
        } else {
            hipError_t e = launch_probe(ix.ibf[0], d_kmers, n_main, d_masks, nullptr, st);
            if (e != hipSuccess) return fail_hip(e, "probe kernel launch");
        }
    }
    if (n_aux) {  // d-grams: same bins, same column shard, their own flat IBF
        hipError_t e = launch_probe(s.aux->ibf[0], d_kmers + n_main, n_aux, d_masks + n_main * (size_t)W, nullptr, st);
        if (e != hipSuccess) return fail_hip(e, "d-gram probe kernel launch");
    }
    if (h->n_ops) {
        // lanes per op: the mask width rounded up to a power of two, at most the whole workgroup
        // (a 65536-bin mask is 1024 words: one word per thread of exec_kernel, four per thread of a unit)
        int g = 1;
        while (g < 1024 && (uint32_t)g < W) g <<= 1;
        const int g_units = g < 256 ? g : 256;
        uint32_t g_units_log2 = 0;
        while ((1 << g_units_log2) < g_units) ++g_units_log2;
        const bool fuse_units = s.kn.fuse_units;  // A/B knob
        const txq_op* d_ops = (const txq_op*)(dblob + h->ops_offset);
        const uint32_t* d_levels = h->n_levels ? (const uint32_t*)(dblob + h->levels_offset) : nullptr;
        const txq_dense_op* d_dops = h->n_dense ? (const txq_dense_op*)(dblob + h->dense_offset) : nullptr;
        const uint32_t np = (uint32_t)s.n_programs;
        if (n_small) {
            size_t blocks = s.n_programs < 4096 ? s.n_programs : 4096;
#define TXQ_EXEC(G) exec_kernel<G><<<(unsigned)blocks, 1024, 0, st>>>(d_progs, d_ops, d_levels, s.d_base, np, d_masks, W)
            switch (g) {
                case 1: TXQ_EXEC(1); break;
                case 2: TXQ_EXEC(2); break;
                case 4: TXQ_EXEC(4); break;
                case 8: TXQ_EXEC(8); break;
                case 16: TXQ_EXEC(16); break;
                case 32: TXQ_EXEC(32); break;
                case 64: TXQ_EXEC(64); break;
                case 128: TXQ_EXEC(128); break;
                case 256: TXQ_EXEC(256); break;
                case 512: TXQ_EXEC(512); break;
                default: TXQ_EXEC(1024); break;
            }
#undef TXQ_EXEC
        }
        size_t first = 0, first_tile = 0, first_hstep = 0, chunk = 0, first_sparse = 0, sparse_launch = 0;
        uint32_t wpr_log2 = 0;
        while (tree && (1u << wpr_log2) < ix.child_row_words) ++wpr_log2;
        for (size_t l = 0; l < plan.size(); ++l) {
            const size_t cnt = plan[l].units;
            ++s.n_levels;
            // a flat index runs the level's units inside its dense launch (below), or its sparse launch when it has no tiles;
            // otherwise they are a launch of their own
            const bool ride = fuse_units && cnt && plan[l].tiles && (!ix.is_hibf || tree || vspace || table);
            const bool ride_sparse = fuse_units && cnt && !ride && plan[l].sparse && (!ix.is_hibf || tree || vspace || table);
            if (cnt && !ride && !ride_sparse) {
                ++s.n_unit_launches;
                exec_units_kernel<<<(unsigned)cnt, 256, 0, st>>>(d_units + first, d_ops, s.d_base, np, d_masks, W, g_units_log2);
            }
            s.n_units += cnt;
            // HIBF: the level's steps, chunk by chunk: k-mers of all (suffix, predecessor) pairs -> tree descent -> combine
            for (const size_t level_end = first_hstep + plan[l].hsteps; first_hstep < level_end; ++chunk) {
                const size_t c0 = chunk_first[chunk], c1 = chunk_first[chunk + 1];
                const uint32_t pairs = chunk_pairs[chunk];
                if (pairs) {
                    dense_hibf_kmers_kernel<<<(unsigned)(c1 - c0), 256, 0, st>>>(d_hsteps + c0, d_pair_base + c0, d_dops, bv.dense, ix.scratch_dense_kmers);
                    TXQ_HIP(hipGetLastError());
                    if (int rc = hibf_probe(ix, s.kn, ix.scratch_dense_kmers, pairs, ix.scratch_dense_masks, nullptr, st)) return rc;
                    dense_hibf_combine_kernel<<<(unsigned)(c1 - c0), 256, 0, st>>>(d_hsteps + c0, d_pair_base + c0, d_dops, d_optr, W, bv.dense, ix.scratch_dense_masks);
                    TXQ_HIP(hipGetLastError());
                }
                first_hstep = c1;
                s.n_dense_tiles += c1 - c0;
            }
            if (plan[l].tiles) {  // ordinary and dense ops of one level are independent of each other: no order implied
                const LevelUnits lu{d_units + first, d_ops, d_masks, ride ? (uint32_t)cnt : 0u, g_units_log2};
                hipError_t e;
                if (vspace) {  // (rows are whole 16-byte chunks: WIDE; two predecessors in flight — a lane keeps its ancestors' gates in registers)
                    auto rows_path = [&](auto& r) { r.chunks = ix.d_vchunks; r.paths = ix.d_vpaths; r.split_range = ix.d_vsplit_range; r.splits = ix.d_vsplits; };
                    e = wide ? launch_dense<true, PathRows>(2, ix.tree_hash_max, rows_path, d_tiles + first_tile, plan[l].tiles, d_dops, d_optr, s.d_base, np, W, g_dense, sl_dense, bv.dense, lu, st)
                             : launch_dense<false, PathRows>(2, ix.tree_hash_max, rows_path, d_tiles + first_tile, plan[l].tiles, d_dops, d_optr, s.d_base, np, W, g_dense, sl_dense, bv.dense, lu, st);
                } else if (tree) {
                    auto rows_of = [&](auto& r) { r.root = ix.root_node; r.children = (const ChildRec*)ix.d_children; r.wpr_log2 = wpr_log2; };
                    // root of <= 64 merged bins and the suffix's lanes cover the mask: root words by lane (TXQ_DENSE_TREE=1: the general variant)
                    const bool by_lane = (256u / (g_dense * sl_dense)) * 32u * ix.root_node.stride() <= kRootWordsLds && tree_knob != 1;
                    if (interleaved) {
                        auto rows_il = [&](auto& r) { r.f = ix.interleaved; r.root = ix.root_node; r.children = (const ChildRec*)ix.d_children; r.wpr_log2 = wpr_log2; };
                        e = wide ? launch_dense<true, InterleavedRows>(s.kn.dense_unroll, ix.tree_hash_max, rows_il, d_tiles + first_tile, plan[l].tiles, d_dops, d_optr, s.d_base, np, W, g_dense, sl_dense, bv.dense, lu, st)
                                 : launch_dense<false, InterleavedRows>(s.kn.dense_unroll, ix.tree_hash_max, rows_il, d_tiles + first_tile, plan[l].tiles, d_dops, d_optr, s.d_base, np, W, g_dense, sl_dense, bv.dense, lu, st);
                    } else if (by_lane)
                        e = wide ? launch_dense<true, TreeRowsByLane>(s.kn.dense_unroll, ix.tree_hash_max, rows_of, d_tiles + first_tile, plan[l].tiles, d_dops, d_optr, s.d_base, np, W, g_dense, sl_dense, bv.dense, lu, st)
                                 : launch_dense<false, TreeRowsByLane>(s.kn.dense_unroll, ix.tree_hash_max, rows_of, d_tiles + first_tile, plan[l].tiles, d_dops, d_optr, s.d_base, np, W, g_dense, sl_dense, bv.dense, lu, st);
                    else
                        e = wide ? launch_dense<true, TreeRows>(s.kn.dense_unroll, ix.tree_hash_max, rows_of, d_tiles + first_tile, plan[l].tiles, d_dops, d_optr, s.d_base, np, W, g_dense, sl_dense, bv.dense, lu, st)
                                 : launch_dense<false, TreeRows>(s.kn.dense_unroll, ix.tree_hash_max, rows_of, d_tiles + first_tile, plan[l].tiles, d_dops, d_optr, s.d_base, np, W, g_dense, sl_dense, bv.dense, lu, st);
                } else if (table) {  // a flat index whose masks of all k-mers are tabulated: one row per k-mer
                    auto rows_tab = [&](auto& r) { r.table = ix.kmer_table; r.stride = W; };
                    e = wide ? launch_dense<true, TableRows>(s.kn.dense_unroll, 1, rows_tab, d_tiles + first_tile, plan[l].tiles, d_dops, d_optr, s.d_base, np, W, g_dense, sl_dense, bv.dense, lu, st)
                             : launch_dense<false, TableRows>(s.kn.dense_unroll, 1, rows_tab, d_tiles + first_tile, plan[l].tiles, d_dops, d_optr, s.d_base, np, W, g_dense, sl_dense, bv.dense, lu, st);
                } else {  // (an irregular HIBF only has ZERO / REDUCE tiles here: its steps are `hsteps`)
                    auto rows_of = [&](auto& r) { r.f = ix.ibf[0]; };
                    e = wide ? launch_dense<true, FlatRows>(s.kn.dense_unroll, ix.ibf[0].hash_funs, rows_of, d_tiles + first_tile, plan[l].tiles, d_dops, d_optr, s.d_base, np, W, g_dense, sl_dense, bv.dense, lu, st)
                             : launch_dense<false, FlatRows>(s.kn.dense_unroll, ix.ibf[0].hash_funs, rows_of, d_tiles + first_tile, plan[l].tiles, d_dops, d_optr, s.d_base, np, W, g_dense, sl_dense, bv.dense, lu, st);
                }
                if (e != hipSuccess) return fail_hip(e, "dense kernel launch");
                first_tile += plan[l].tiles;
                s.n_dense_tiles += plan[l].tiles;
                ++s.n_dense_launches;
            }
            // the level's sparse groups (dense ops of tracked programs): plan (counts -> chunks), then the chunks.  split_steps: the
            // groups are ordered [others | STEPs] and the STEPs go to the compacted step kernel (two ranges, each in segments of
            // at most kMaxSparseGroups groups); otherwise sparse_kernel takes all of them
            const size_t n_misc = split_steps ? plan[l].sparse_misc : plan[l].sparse;
            bool rode = false;  // the level's ordinary ops ride in its first sparse launch
            for (int range = 0; range < 2; ++range) {
                const size_t r_lo = range == 0 ? 0 : n_misc, r_hi = range == 0 ? n_misc : plan[l].sparse;
                const bool steps = range == 1;
                size_t range_chunks = steps ? plan[l].step_chunks : plan[l].sparse_chunks;
                if (steps && by_units) range_chunks = range_chunks * (kSparseChunk / 16);  // (counted in chunks of kSparseChunk entries; a chunk by units holds 16 at least)
                // The STEPs of sparse_kernel (trees; masks wider than kUnitStepWords): a chunk is ONE round of the workgroup's lane groups
                // on wide masks — the live lists of a layout-order session are short (thousands of entries of 19.8 KB), and a
                // chunk of 64 of them was sixteen rounds of dozens of dependent trips in one workgroup while most of the device
                // idled (level 2 of the 200-motif batch: 2900 entries, 6.1 ms) — and kSparseChunk entries where a round holds that many
                const uint32_t step_chunk = steps && !by_units && W > kUnitStepWords ? std::max<uint32_t>(1u, std::min<uint32_t>(kSparseChunk, 256u / g_dense)) : kSparseChunk;
                if (steps && !by_units) range_chunks = range_chunks * (kSparseChunk / step_chunk);
                for (size_t off = r_lo; off < r_hi; off += kMaxSparseGroups, ++sparse_launch) {
                    const uint32_t ng = (uint32_t)std::min<size_t>(kMaxSparseGroups, r_hi - off);
                    const SparseGroup* gr = d_sgroups + first_sparse + off;
                    uint32_t* counts = d_scounts + first_sparse + off;
                    uint32_t* prefix = d_sprefix + first_sparse + off + sparse_launch;
                    sparse_plan_kernel<<<1, 1024, 0, st>>>(gr, ng, d_dops, d_optr, W, bv.dense.pos, steps && by_units ? (0x80000000u | (uint32_t)s.kn.sparse_units) : step_chunk, counts, prefix);
                    TXQ_HIP(hipGetLastError());
                    const LevelUnits lu{d_units + first, d_ops, d_masks, ride_sparse && !rode ? (uint32_t)cnt : 0u, g_units_log2};
                    rode = true;
                    // as many workgroups as the chunks could be at most, within what the device holds at a time
                    const size_t grid = lu.n_units + std::max<size_t>(1, std::min<size_t>(range_chunks, 2048));
                    hipError_t e;
                    if (steps && by_units) {
                        StepParams sp{bv.dense.k, bv.dense.bits, bv.dense.pos, bv.dense.canonical, (uint32_t)s.kn.sparse_units, 0u};
#ifdef TXQ_EXPERIMENTS
                        if (const char* ex = std::getenv("TXQ_STEP_EXPERIMENT")) sp.experiment = (uint32_t)std::atoi(ex);  // 1: destination atomics twice, 2: bitmap atomics twice, 4: row gathers twice
#endif
                        if (table) {
                            auto rows_tab = [&](auto& r) { r.table = ix.kmer_table; r.stride = W; };
                            e = wide ? launch_sparse_units<true, TableRows>(1, rows_tab, gr, ng, counts, prefix, grid, d_dops, d_optr, s.d_base, np, W, g_dense, sp, lu, s.d_step_ctr, s.kn.sparse_unroll, st)
                                     : launch_sparse_units<false, TableRows>(1, rows_tab, gr, ng, counts, prefix, grid, d_dops, d_optr, s.d_base, np, W, g_dense, sp, lu, s.d_step_ctr, s.kn.sparse_unroll, st);
                        } else {
                            auto rows_of = [&](auto& r) { r.f = ix.ibf[0]; };
                            e = wide ? launch_sparse_units<true, FlatRows>(ix.ibf[0].hash_funs, rows_of, gr, ng, counts, prefix, grid, d_dops, d_optr, s.d_base, np, W, g_dense, sp, lu, s.d_step_ctr, s.kn.sparse_unroll, st)
                                     : launch_sparse_units<false, FlatRows>(ix.ibf[0].hash_funs, rows_of, gr, ng, counts, prefix, grid, d_dops, d_optr, s.d_base, np, W, g_dense, sp, lu, s.d_step_ctr, s.kn.sparse_unroll, st);
                        }
                    } else if (split_steps && !steps) {  // ZERO / REDUCE / FILL only: the variant without the step code (its row source is not used)
                        FlatRows<1, true> none{};
                        if (wide) sparse_kernel<1, true, FlatRows<1, true>, false><<<(unsigned)grid, 256, 0, st>>>(none, gr, ng, counts, prefix, d_dops, d_optr, s.d_base, np, W, g_dense, bv.dense, lu, kSparseChunk);
                        else { FlatRows<1, false> none1{}; sparse_kernel<1, false, FlatRows<1, false>, false><<<(unsigned)grid, 256, 0, st>>>(none1, gr, ng, counts, prefix, d_dops, d_optr, s.d_base, np, W, g_dense, bv.dense, lu, kSparseChunk); }
                        e = hipGetLastError();
                    } else if (vspace) {
                        auto rows_path = [&](auto& r) { r.chunks = ix.d_vchunks; r.paths = ix.d_vpaths; r.split_range = ix.d_vsplit_range; r.splits = ix.d_vsplits; };
                        e = wide ? launch_sparse<true, PathRows>(ix.tree_hash_max, rows_path, gr, ng, counts, prefix, grid, d_dops, d_optr, s.d_base, np, W, g_dense, bv.dense, lu, step_chunk, st)
                                 : launch_sparse<false, PathRows>(ix.tree_hash_max, rows_path, gr, ng, counts, prefix, grid, d_dops, d_optr, s.d_base, np, W, g_dense, bv.dense, lu, step_chunk, st);
                    } else if (interleaved) {
                        auto rows_il = [&](auto& r) { r.f = ix.interleaved; r.root = ix.root_node; r.children = (const ChildRec*)ix.d_children; r.wpr_log2 = wpr_log2; };
                        e = wide ? launch_sparse<true, InterleavedRows>(ix.tree_hash_max, rows_il, gr, ng, counts, prefix, grid, d_dops, d_optr, s.d_base, np, W, g_dense, bv.dense, lu, step_chunk, st)
                                 : launch_sparse<false, InterleavedRows>(ix.tree_hash_max, rows_il, gr, ng, counts, prefix, grid, d_dops, d_optr, s.d_base, np, W, g_dense, bv.dense, lu, step_chunk, st);
                    } else if (tree) {
                        auto rows_of = [&](auto& r) { r.root = ix.root_node; r.children = (const ChildRec*)ix.d_children; r.wpr_log2 = wpr_log2; };
                        e = wide ? launch_sparse<true, TreeRows>(ix.tree_hash_max, rows_of, gr, ng, counts, prefix, grid, d_dops, d_optr, s.d_base, np, W, g_dense, bv.dense, lu, step_chunk, st)
                                 : launch_sparse<false, TreeRows>(ix.tree_hash_max, rows_of, gr, ng, counts, prefix, grid, d_dops, d_optr, s.d_base, np, W, g_dense, bv.dense, lu, step_chunk, st);
                    } else if (table) {
                        auto rows_tab = [&](auto& r) { r.table = ix.kmer_table; r.stride = W; };
                        e = wide ? launch_sparse<true, TableRows>(1, rows_tab, gr, ng, counts, prefix, grid, d_dops, d_optr, s.d_base, np, W, g_dense, bv.dense, lu, step_chunk, st)
                                 : launch_sparse<false, TableRows>(1, rows_tab, gr, ng, counts, prefix, grid, d_dops, d_optr, s.d_base, np, W, g_dense, bv.dense, lu, step_chunk, st);
                    } else {
                        auto rows_of = [&](auto& r) { r.f = ix.ibf[0]; };
                        e = wide ? launch_sparse<true, FlatRows>(ix.ibf[0].hash_funs, rows_of, gr, ng, counts, prefix, grid, d_dops, d_optr, s.d_base, np, W, g_dense, bv.dense, lu, step_chunk, st)
                                 : launch_sparse<false, FlatRows>(ix.ibf[0].hash_funs, rows_of, gr, ng, counts, prefix, grid, d_dops, d_optr, s.d_base, np, W, g_dense, bv.dense, lu, step_chunk, st);
                    }
                    if (e != hipSuccess) return fail_hip(e, "sparse kernel launch");
                    if (s.kn.trace_sync && s.kn.trace_stages) {  // (profiling aid: TXQ_TRACE_SYNC + TXQ_TRACE_STAGES) what each sparse launch amounted to
                        const double t_l = now_s();
                        (void)hipStreamSynchronize(st);
                        const double dt = now_s() - t_l;
                        std::vector<uint32_t> cnts(ng);
                        (void)hipMemcpy(cnts.data(), counts, (size_t)ng * 4, hipMemcpyDeviceToHost);
                        const txq_dense_op* hd = (const txq_dense_op*)(blob + bv.dense_offset);
                        uint64_t by_kind[5] = {0, 0, 0, 0, 0};  // entries: ZERO, STEP, REDUCE, FILL, STEP without probe
                        uint64_t visits = 0;
                        for (uint32_t gi = 0; gi < ng; ++gi) {
                            const txq_dense_op& x = hd[sparse_groups[first_sparse + off + gi].op];
                            const bool np_ = x.kind == TXQ_DENSE_STEP && (x.reserved & TXQ_DENSE_NOPROBE);
                            by_kind[np_ ? 4 : x.kind] += cnts[gi];
                            if (x.kind == TXQ_DENSE_STEP) visits += (uint64_t)cnts[gi] * (uint64_t)__builtin_popcount(x.r_mask);
                        }
                        fprintf(stderr, "[txq]   sparse launch%s: level %zu, %u groups, %.1f us; entries: zero %llu, step %llu (+ %llu without probe; %llu visits), reduce %llu, fill %llu\n",
                                steps ? " (steps)" : "", l, ng, dt * 1e6, (unsigned long long)by_kind[0], (unsigned long long)by_kind[1], (unsigned long long)by_kind[4],
                                (unsigned long long)visits, (unsigned long long)by_kind[2], (unsigned long long)by_kind[3]);
                    }
                    ++s.n_sparse_launches;
                    s.n_sparse_groups += ng;
                }
            }
            first_sparse += plan[l].sparse;
            first += cnt;
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "exec kernel launch");
    if (n_q) {
        size_t blocks = (n_q + 3) / 4;
        if (blocks > 2048) blocks = 2048;
        slot_alive_kernel<<<(unsigned)blocks, 256, 0, st>>>(s.d_base, d_qp, d_qs, (uint32_t)n_q, W, d_alive);
        TXQ_HIP(hipGetLastError());
        TXQ_HIP(hipMemcpyAsync(alive, d_alive, n_q, hipMemcpyDeviceToHost, st));
        TXQ_HIP(hipStreamSynchronize(st));
    }
    TXQ_HIP(hipEventRecord(S.done, st));
    S.pending = true;
    if (s.kn.trace_sync) (void)hipStreamSynchronize(st);  // charges the device time to the stage that caused it
    s.t_device += now_s() - t0;
    if (s.kn.trace_stages)
        fprintf(stderr, "[txq] stage %zu: regions %.0f us, plan %.0f us, staging set %.0f us, buffers %.0f us, copies issued %.0f us, copies done %.0f us, launches %.0f us\n", s.n_stages,
                (t_mark[0] - t_begin) * 1e6, (t_mark[1] - t_mark[0]) * 1e6, (t_mark[2] - t_mark[1]) * 1e6, (t_mark[3] - t_mark[2]) * 1e6, (t_mark[4] - t_mark[3]) * 1e6,
                (t_mark[5] - t_mark[4]) * 1e6, (now_s() - t_mark[5]) * 1e6);
    return TXQ_OK;
}

int session_finish(Session& s, uint64_t* d_final, hipStream_t st) {
    const uint32_t W = s.W;
    if (W == 0 || s.n_programs == 0) return TXQ_OK;
    if (s.n_stages == 0) return fail(TXQ_ERR_STATE, "the session never ran a stage");
    // (a program that never had an op has no region: its RESULT is the initial one, no bin)
    for (Index::StagingSet& t : s.set)  // stages may have run on two streams: all of them before the results are gathered
        if (t.pending) {
            hipError_t e = hipEventSynchronize(t.done);
            if (e != hipSuccess) return fail_hip(e, "waiting for the last stages");
            t.pending = false;
        }
    size_t blocks = (s.n_programs * W + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    uint64_t* gathered = d_final;
    if (s.vspace) {  // RESULT slots are rows in layout order: gathered into scratch, then converted to user-bin order
        if (int rc = ensure((void**)&s.ix->scratch_slots, &s.ix->cap_slots, s.n_programs * (size_t)W * 8)) return rc;
        gathered = s.ix->scratch_slots;
    }
    gather_result_kernel<<<(unsigned)blocks, 256, 0, st>>>(s.d_base, (uint32_t)s.n_programs, W, gathered);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "gather kernel launch");
    if (s.vspace) return hibf_layout_to_user(*s.ix, gathered, s.n_programs, d_final, st);
    return TXQ_OK;
}

void preload_exec_kernels() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&gather_result_kernel));
    (void)hipGetLastError();
}

int run_programs(Index& ix, const void* blob, size_t bytes, size_t n_programs, uint64_t* d_final, hipStream_t st) {
    Session* s = nullptr;
    if (int rc = session_begin(ix, n_programs, &s)) return rc;
    int rc = session_stage(*s, blob, bytes, nullptr, nullptr, 0, nullptr, st);
    if (rc == TXQ_OK) rc = session_finish(*s, d_final, st);
    // the slot arena must outlive the kernels that read it
    if (hipStreamSynchronize(st) != hipSuccess && rc == TXQ_OK) rc = fail(TXQ_ERR_HIP, "stream synchronize failed");
    delete s;
    return rc;
}

}  // namespace txq

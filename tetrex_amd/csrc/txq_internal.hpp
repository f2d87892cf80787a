// Internal (non-ABI) declarations shared by the translation units of libtxq.so.
#pragma once
#include "txq_kernels.hpp"
#include "../../include/txq.h"
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace txq {

// Every environment variable libtxq.so reads (all of them A/B and test switches between code paths that give the
// SAME results; listed in include/txq.h).  They are parsed in ONE place (txq_api.hip read_knobs) at the library's entry
// points — txq_init, txq_index_upload, txq_session_begin and the probe calls — and everything below works from
// that snapshot: nothing in the library reads the environment while it runs a stage.
struct Knobs {
    bool trace = false, trace_stages = false, trace_sync = false;  // TXQ_TRACE, TXQ_TRACE_STAGES, TXQ_TRACE_SYNC
    // executor (txq_exec.hip)
    int dense_tree = -1;        // TXQ_DENSE_TREE: 0 generic HIBF steps, 1 TreeRows, 2 TreeRowsByLane where it applies; -1: best fit
    int dense_unroll = 3;       // TXQ_DENSE_UNROLL: predecessors in flight per lane (2, 3, 5)
    int dense_slices = 2;       // TXQ_DENSE_SLICES: lane groups sharing the predecessors of one suffix
    int dense_tile_rounds = 4;  // TXQ_DENSE_TILE_ROUNDS: destination suffixes per lane-group set and tile (profiles/r3_dense_tile_shapes_ab.txt)
    int dense_nt = 0;           // TXQ_DENSE_NT: bit 0 non-temporal stores, bit 1 non-temporal loads of a dense step's destination entries
    bool fuse_units = true;     // TXQ_FUSE_UNITS=0: a level's ordinary ops get a launch of their own
    bool one_stream = false;    // TXQ_ONE_STREAM: independent stages do not run beside each other
    int sparse_units = 512;     // TXQ_SPARSE_UNITS: (entry x residue) units per chunk of sparse_units_kernel, 64 .. 1536
    int sparse_unroll = 3;      // TXQ_SPARSE_UNROLL: units in flight per lane group in sparse_units_kernel (2 or 3)
    bool sparse_steps = true;   // TXQ_SPARSE_STEPS=0: pushed steps on narrow masks run in sparse_kernel (rounds of one entry per lane group), not by units (sparse_units_kernel)
    long long kmer_table_mb = 512;  // TXQ_KMER_TABLE_MB: most an index's table of ALL k-mers' masks may take (0: dense steps always gather rows)
    long long kmer_table_min = 16;  // TXQ_KMER_TABLE_MIN: the session of fewest programs that builds the table (a single query does not pay for it; once built it is used)
    // HIBF (txq_hibf.hip)
    bool hibf_interleave = true;        // TXQ_HIBF_INTERLEAVE=0: no interleaved copy of uniform children (at upload)
    bool hibf_interleave_probe = true;  // TXQ_HIBF_INTERLEAVE_PROBE=0: plain probes descend the tree
    bool hibf_levels = false;           // TXQ_HIBF_LEVELS=1: level-synchronous descent
    bool hibf_stationary = true;        // TXQ_HIBF_STATIONARY=0: no child-stationary descent
    bool hibf_small = true;             // TXQ_HIBF_SMALL=0: no lane-per-k-mer kernel for small trees
    bool hibf_lane_hash = false;        // TXQ_HIBF_LANE_HASH: per-lane hashing on a uniform tree
    bool final_pinned = true;           // TXQ_FINAL_PINNED=0: a session's final masks through a device buffer and hipMemcpy
    bool hibf_layout_fused = true;      // TXQ_HIBF_LAYOUT_FUSED=0: the layout-order rows of plain k-mers level by level (hibf_layout_level_kernel), not one wave per k-mer
    bool hibf_layout_order = true;      // TXQ_HIBF_LAYOUT_ORDER=0: sessions on general trees work in user-bin order (descent kernels)
    int hibf_steps_per_group = 0, hibf_tile = 0, hibf_unroll = 1, hibf_store = 0;  // TXQ_HIBF_STEPS_PER_GROUP / _TILE / _UNROLL / _STORE_KIND (store instruction: 0-3)
    long long hibf_waves = 0;           // TXQ_HIBF_WAVES
    bool hibf_layout_direct = true;     // TXQ_HIBF_LAYOUT_DIRECT=0: hibf_fused_kernel<G, LAYOUT> keeps the row in LDS instead of writing it directly
    long long hibf_stack_lds = 128;     // TXQ_HIBF_STACK_LDS: entries of hibf_fused_kernel's IBF stack kept in LDS (the rest: in the k-mer's output row)
    // probe (txq_probe.hip)
    int probe_blocks_per_cu = 256, probe_unroll = 2;  // TXQ_PROBE_BLOCKS_PER_CU, TXQ_PROBE_UNROLL
    bool probe_nt = false;                            // TXQ_PROBE_NT
};
Knobs knobs();      // a copy of the snapshot taken at the last entry point (published under a lock: entry points run on several threads)
void read_knobs();  // take it (txq_api.hip)

// Everything about one IBF of an HIBF tree in one 32-byte record (txq_hibf.hip: nodes[e] = the child behind merged
// technical bin e; the dense steps on a regular tree take the root as a kernel argument).
struct HibfNode {  // 32 bytes = two 16-byte loads per lane
    uint64_t words;       // device pointer to the IBF's rows
    uint32_t bin_size;    // rows (< 2^32: the fused kernel is not used for larger IBFs)
    uint32_t packed;      // stride (bits 0-19) | hash_shift (20-25) | hash_funs (26-28) | has merged bins (29)
    uint32_t off;         // first entry of the IBF's technical bins in the flattened maps
    uint32_t moff;        // first word of the IBF in `merged` / `descend`
    uint32_t ident_word;  // see IbfDev::ident_word
    uint32_t bins;        // technical bins
    __host__ __device__ uint32_t stride() const { return packed & 0xFFFFFu; }
    __host__ __device__ uint32_t hash_shift() const { return (packed >> 20) & 63u; }
    __host__ __device__ uint32_t hash_funs() const { return (packed >> 26) & 7u; }
    __host__ __device__ bool has_merged() const { return (packed >> 29) & 1u; }
    __host__ __device__ uint32_t words_per_row() const { return (bins + 63u) >> 6; }
};
static_assert(sizeof(HibfNode) == 32, "two 16-byte pieces per node");

// Regular two-level trees: one record per child in mask-column order (child-stationary descent in txq_hibf.hip, dense
// steps on the tree in txq_exec.hip).
struct ChildRec {   // 16 bytes, one per child in mask-column order
    uint64_t words;     // device pointer to the child's rows (stride = row words, a power of two >= 2)
    uint32_t bin_size;  // rows
    uint32_t packed;    // hash_shift (bits 0-7) | hash_funs (8-11) | root technical bin (12-31)
};
static_assert(sizeof(ChildRec) == 16, "one 16-byte load per lane");

// General HIBFs in LAYOUT ORDER (sessions on trees that are not regular: three and more levels, user bins next to merged
// bins, split bins, user bins in any order — what seqan::hibf's layout produces, reference include/index_hibf.h:114-129).
// A session on such a tree does not work on masks in user-bin order but on rows in the order of the tree's own
// technical bins: the row of every IBF, one after the other (levels ascending, every IBF padded to an even number of
// words), W_v words in all.  In that order every IBF owns an aligned segment of the row, so a k-mer's mask is written
// segment by segment with coalesced stores and no atomics (child-stationary, level by level), and a dense step gathers a
// lane's 16 bytes from ONE IBF.  Every operation of the collector is bin-wise, so the order of the bins does not matter
// until the final masks are handed out: those are converted to user-bin order (split bins ORed) once per query.
// Merged bins keep their bits in the rows (the next level reads them as its gates); they never reach a result because
// the ONES slot of a layout-order session only has the bits of technical bins that ARE user bins.
struct VChunk {          // one chunk of the layout-order row: two row words (16 bytes) of one IBF, or one (Index::v_chunk_words)
    uint64_t words;      // the IBF's rows
    uint32_t bin_size;   // rows (< 2^32)
    uint32_t packed;     // stride (bits 0-19) | hash_shift (20-25) | hash_funs (26-28) | single-word rows (29) | holds representatives of split user bins (30)
    uint32_t col;        // word column of the chunk within the IBF's row
    uint32_t gate_word;  // layout-order word that holds the parent's merged bin leading here (kNoGate: the root)
    uint32_t gate_bit;
    uint32_t ibf;        // IBF id (its VPath)
};
static_assert(sizeof(VChunk) == 32, "two 16-byte loads per lane");
static constexpr uint32_t kNoGate = 0xFFFFFFFFu;
static constexpr uint32_t kMaxVDepth = 3;  // ancestors a fused dense step follows (trees of up to 4 levels)
struct VPath {           // the ancestors of an IBF, root first: whose merged bin (row word, bit) leads towards it
    uint32_t depth, pad;
    struct { uint64_t words; uint32_t bin_size, packed, word, bit; } anc[kMaxVDepth];
};
// Split user bins in layout order.  A user bin that the layout spreads over several technical bins of one IBF holds a k-mer when
// ANY of its parts does, and masks are combined per USER bin (reference include/index_hibf.h:132-147 ORs the parts before the
// collector ANDs anything) — so in a layout-order row a split bin is ONE bit, its first part's (the representative), which
// stands for the OR of the parts; the other parts' bits are always zero.  Rows of plain k-mers are put into that form as they
// are written (hibf_fused_kernel<G, LAYOUT>) or right after (unify_split_rows_kernel behind the level kernels).  A fused step
// (PathRows) works on one 16-byte chunk of an IBF's row and does not see the other parts: for them every IBF with split bins has
// a SIDE matrix — the columns of its non-representative parts once more, packed so that the parts belonging to one chunk's
// representatives are consecutive bits of one 64-bit word per row.  ANDing the k-mer's h side rows gives those parts' hits
// exactly (they are the IBF's own columns), and a hit sets its representative: VSplit entry e of the chunk (sorted by
// representative) is side bit `bit0 + e`.
struct VSplit { uint32_t part_word; uint16_t rep_bit, part_bit; };  // word column and bit of the part in the IBF's row; its representative's bit in the chunk
struct VSplitRange {
    uint32_t first, count;   // the chunk's entries in Index::d_vsplits (fewer than the chunk has bits)
    uint32_t reps[4];        // the chunk's bits that are representatives (bit b of the 128: reps[b >> 5] >> (b & 31))
    uint64_t side;           // device pointer: row 0 of the chunk's (first) word in the IBF's side matrix — entry e is bit bit0 + e from there
    uint32_t side_stride;    // words per side row
    uint32_t bit0;           // the chunk's first bit in that word
};
struct VLevel { uint32_t first_chunk, n_chunks; std::vector<uint32_t> group_first; };  // groups: chunk ranges whose IBFs share an L2's worth of rows

// One HIBF work item: k-mer `kmer` (index into the batch) must be looked up in IBF `ibf`.
struct WorkItem { uint32_t kmer; uint32_t ibf; };

struct Index {
    int device = 0;
    bool is_hibf = false;
    uint64_t user_bins = 0, mask_words = 0, shard_word0 = 0, shard_words = 0, device_bytes = 0;
    std::vector<IbfDev> ibf;  // host copies of the device descriptors ([0] = flat IBF / HIBF root)

    // HIBF tree in HBM
    IbfDev* d_ibf = nullptr;         // [n_ibf]
    uint64_t* d_next = nullptr;      // flattened next_ibf_id
    uint64_t* d_tb_user = nullptr;   // flattened tb_to_user_bin (TXQ_MERGED_BIN for merged)
    uint64_t* d_map_off = nullptr;   // [n_ibf] offset of IBF i's maps in the flattened arrays
    uint64_t* d_merged = nullptr;    // merged-bin bitmask words of every IBF, flattened
    void* d_nodes = nullptr;         // HibfNode[total technical bins + 1] for the fused descent (txq_hibf.hip)
    uint64_t* d_descend = nullptr;   // same layout: merged bins worth descending into for this shard
    uint64_t* d_merged_off = nullptr;
    // Regular two-level trees (root of merged bins over leaf IBFs that each map an aligned run of user bins, all of
    // one row width): the child-stationary descent of txq_hibf.hip (rows of >= 2 words) and the fused dense steps of
    // txq_exec.hip.  d_children = ChildRec[n_children] in mask-column
    // order for THIS shard's columns; empty when the tree does not have that shape.
    void* d_children = nullptr;
    uint32_t n_children = 0;         // children whose columns this shard owns
    uint32_t child_row_words = 0;    // mask words per child (a power of two)
    // Small regular trees with uniform children (root of <= 64 merged bins, mask of <= 32 words): the children's matrices
    // once more, row r of all children side by side ([rows][stride] like a flat IBF over the children's hash parameters),
    // so that a dense step gathers one row segment per hash function instead of one cache line per child.
    IbfDev interleaved{};
    // plain k-mer probes go to the interleaved children too (TXQ_HIBF_INTERLEAVE_PROBE=0: the tree descent kernels; A/B and tests)
    bool probes_interleaved(const Knobs& kn) const {
        return is_hibf && interleaved.words && interleaved.stride >= 2 && !(interleaved.stride & 1) && interleaved.shard_words == shard_words &&
               root_node.bins <= 64 && kn.hibf_interleave_probe;
    }
    HibfNode root_node{};            // host copy of the root's record
    uint32_t tree_hash_max = 0;      // most hash functions of any IBF of the regular tree
    bool children_uniform = false;   // same rows / hash shift / hash count in every child: scalar hashing
    uint64_t children_bytes = 0;     // their matrices
    uint32_t* scratch_crows = nullptr; size_t cap_crows = 0;  // uniform children: per k-mer its row indexes in a child
    uint64_t* scratch_cm = nullptr; size_t cap_cm = 0;  // root pass output: per k-mer the root row (which children to visit)
    // layout order (see VChunk): built at upload for trees that are not regular, one shard, fewer than 2^32 rows per IBF
    VChunk* d_vchunks = nullptr;
    VPath* d_vpaths = nullptr;
    uint64_t* d_vleaf = nullptr;     // [v_words] bits of technical bins that are user bins (the ONES of a layout-order session)
    uint32_t* d_vuser = nullptr;     // [v_words * 64] user bin of a layout-order bit (kNoGate: none)
    uint32_t* d_vgroups = nullptr;   // per level its groups' first chunks, concatenated (+ end)
    uint64_t* d_vnonrep = nullptr;   // [v_words] bits of the parts of split user bins that are not their representative (null: no split bins)
    uint32_t* d_vrep = nullptr;      // [v_words * 64] for such a bit: the representative's bit position in the row
    VSplitRange* d_vsplit_range = nullptr;  // [n_vchunks]
    VSplit* d_vsplits = nullptr;
    uint64_t* d_vside = nullptr;     // the side matrices of all IBFs with split bins
    void* d_vnodes = nullptr;        // HibfNode records (as d_nodes) whose ident_word is the IBF's first word in the layout-order row
    uint32_t v_inner_words = 0;  // of a row: the words of IBFs with merged bins (what the next level reads as gates)
    uint32_t v_words = 0, n_vchunks = 0, v_depth = 0, v_chunk_words = 2;  // (chunks of 16 bytes, or of 8 for trees of narrow IBFs)
    std::vector<VLevel> vlevels;
    bool layout_order(const Knobs& kn) const;  // sessions on this index work in layout order
    uint32_t depth = 1;              // levels of the tree
    uint64_t hibf_total_tbs = 0;     // technical bins over all IBFs of the tree
    uint64_t max_level_width = 1;    // max number of IBFs on one level (bounds the frontier)
    uint32_t max_stride = 1;         // widest row over all IBFs (words)

    // grow-only scratch (owned by the index; one host thread at a time)
    uint64_t* scratch_kmers = nullptr; size_t cap_kmers = 0;
    uint64_t* scratch_masks = nullptr; size_t cap_masks = 0;
    WorkItem* frontier[2] = {nullptr, nullptr}; size_t cap_frontier[2] = {0, 0};
    uint32_t* d_counts = nullptr; size_t cap_counts = 0;
    unsigned char* scratch_blob = nullptr; size_t cap_blob = 0;
    uint64_t* scratch_slots = nullptr; size_t cap_slots = 0;
    uint64_t* scratch_final = nullptr; size_t cap_final = 0;
    uint64_t* host_final = nullptr; size_t cap_host_final = 0;  // pinned: a session's final masks are gathered straight into host memory
    uint64_t* scratch_dense_kmers = nullptr; size_t cap_dense_kmers = 0;  // dense steps on an HIBF: the pairs' k-mers ...
    uint64_t* scratch_dense_masks = nullptr; size_t cap_dense_masks = 0;  // ... and their descended masks
    // A flat index's masks of ALL k-mers, M[v] at kmer_table + v * shard_words for every packed value v < 2^(bits * k) (txq_exec.hip
    // ensure_kmer_table): where that fits TXQ_KMER_TABLE_MB, a dense step reads ONE row per k-mer instead of gathering hash_funs.
    uint64_t* kmer_table = nullptr; uint32_t kmer_table_bits = 0;  // bits = bits per residue * k of the table that is built
    bool kmer_table_refused = false;                               // (the allocation failed once: not tried again)
    std::mutex table_mutex;                                        // building / dropping the table (sessions of one index may run on several threads)

    // Device buffers of the last session, kept for the next one: a single query must not pay
    // hipMalloc/hipFree (they cost more than its kernels).  One session at a time may hold them.
    struct ArenaChunk { uint64_t* p; size_t cap; };  // cap in 64-bit words
    // What one stage of a session uploads: the blob, and `aux` = program table, lists, units, tiles, region moves and the
    // table of region bases.  A session alternates between two sets, so stage n+1 is uploaded (on its own stream) while
    // the kernels of stage n still read theirs; `done` is recorded behind a stage's last kernel.
    struct StagingSet {
        unsigned char* d_blob = nullptr; size_t cap_blob = 0;
        unsigned char* d_aux = nullptr; size_t cap_aux = 0;
        uint64_t* d_masks = nullptr; size_t cap_masks = 0;  // M[k-mer] of the stage's k-mer table (the probe's output)
        hipEvent_t done = nullptr;
        bool pending = false;
    };
    // A dense block as it is handed on: [cap][W] mask words, then its live list.  state: kGarbage (fresh memory, or left by an
    // untracked program), kListed (left by a tracked program: all zero except the entries in its list, which is intact — the
    // ZERO that re-creates it for a tracked program clears exactly those, so such a block needs no clearing at all).
    struct DenseBlock { uint64_t* p; uint32_t cap; uint8_t state; };
    // blocks by capacity: a handful of capacities (powers of two for tracked blocks, A^(k-1) for untracked ones) with thousands
    // of blocks each — a vector per capacity (a multimap's node per block made releasing a 10 000-query session 2.5 ms)
    struct BlockBins {
        std::vector<std::pair<uint32_t, std::vector<DenseBlock>>> bins;
        std::vector<DenseBlock>& of(uint32_t cap) {
            for (auto& b : bins) if (b.first == cap) return b.second;
            bins.emplace_back(cap, std::vector<DenseBlock>());
            return bins.back().second;
        }
        void put(const DenseBlock& b) { of(b.cap).push_back(b); }
        bool take(uint32_t cap, DenseBlock* out) {
            for (auto& b : bins)
                if (b.first == cap) {
                    if (b.second.empty()) return false;
                    *out = b.second.back();
                    b.second.pop_back();
                    return true;
                }
            return false;
        }
        void absorb(BlockBins& other) {  // everything of `other` moves in
            for (auto& b : other.bins) {
                std::vector<DenseBlock>& mine = of(b.first);
                if (mine.empty()) mine.swap(b.second);
                else { mine.insert(mine.end(), b.second.begin(), b.second.end()); b.second.clear(); }
            }
        }
        void swap(BlockBins& o) { bins.swap(o.bins); }
        size_t size() const { size_t n = 0; for (const auto& b : bins) n += b.second.size(); return n; }
        void clear() { bins.clear(); }
    };
    struct SessionCache {
        std::vector<ArenaChunk> chunks;  // slot-arena chunks, at most kArenaKeepBytes in all
        // dense blocks live in chunks of their own, and ALL blocks of a session go back into a pool by capacity when it ends:
        // the next batch on this index takes its blocks from there — no allocation, and for tracked programs no clearing
        // (5.3 GB of memset per 200-motif batch at k = 6 before)
        std::vector<ArenaChunk> block_chunks;
        size_t block_cur = 0, block_used = 0;
        BlockBins blocks;
        uint32_t blocks_W = 0;  // the mask width the pooled blocks were laid out for
        StagingSet set[2];
        hipStream_t upload = nullptr, side = nullptr;
        bool in_use = false;
    } session_cache;
    static constexpr size_t kArenaKeepBytes = (size_t)64 << 30;
    uint64_t user_tag = 0;  // txq_index_set_tag
    int open_sessions = 0;  // txq_index_free refuses while a session still points at this index
    bool join_or = false;   // a sub-tree shard of a general HIBF (txq_index_upload_subtrees): full-width masks, ORed with the other shards'
    int shard_rank = 0, n_shards = 1;

    // txq_probe (host buffers): two streams with their device and pinned bounce buffers
    struct HostPipe {
        hipStream_t stream[2] = {nullptr, nullptr};
        hipEvent_t done[2] = {nullptr, nullptr};
        uint64_t* d_kmers[2] = {nullptr, nullptr}; size_t cap_kmers[2] = {0, 0};
        uint64_t* d_masks[2] = {nullptr, nullptr}; size_t cap_masks[2] = {0, 0};
        uint64_t* bounce[2] = {nullptr, nullptr}; size_t cap_bounce[2] = {0, 0};
    } host_pipe;

    void release();
};

// Does a session on this index run dense steps fused on the tree (txq_exec.hip TreeRows / InterleavedRows)?  A regular
// two-level HIBF whose children tile this shard's mask columns.  TXQ_DENSE_TREE=0 (A/B and tests) sends steps through
// the generic HIBF path instead.
inline bool Index::layout_order(const Knobs& kn) const {
    return is_hibf && d_vchunks && v_words && kn.hibf_layout_order && shard_words == mask_words && shard_word0 == 0;
}
inline bool index_fuses_tree_steps(const Index& ix, const Knobs& kn) {
    return ix.is_hibf && ix.d_children && ix.n_children && kn.dense_tree != 0 && ix.tree_hash_max >= 1 && ix.tree_hash_max <= 5 &&
           (uint64_t)ix.n_children * ix.child_row_words == ix.shard_words;
}

// Normalised program descriptor the executor kernel reads (both blob versions map onto it).
struct DevProgram { uint32_t first_op, n_ops, first_level, n_levels; };

// A batch of programs whose slot masks persist in HBM across stages (txq_exec.hip).
struct Session {
    Index* ix = nullptr;
    Index* aux = nullptr;  // optional d-gram index (flat IBF, same bins and shard as ix)
    Knobs kn;              // the environment switches as they were when the session began
    size_t n_programs = 0;
    uint32_t W = 0;        // words of a slot mask: the shard's mask words, or (vspace) the words of a layout-order row
    bool failed = false;   // a stage failed after it may have launched kernels: the device was drained, further stages are refused
    bool vspace = false;   // the index is a general HIBF: masks are rows in layout order, final masks are converted (Index::layout_order)
    std::vector<Index::ArenaChunk> chunks;  // arena chunks; bump allocation in chunks[cur]
    size_t cur = 0, chunk_used = 0, arena_words = 0;
    std::vector<uint64_t*> base;    // [2 * n_programs]: per program its slot region [cap][W], then (device address of) its row of the stage's block table
    std::vector<uint32_t> cap;      // per program: slots allocated
    // Dense blocks (include/txq_program.h, version 3): block b of program p is blocks[p][b], an allocation of its own —
    // [N][W] mask words, then the block's live list (tracked programs): count | bitmap of N bits | list of N entries.
    // A program that needs more blocks just gets more (nothing ever moves); kernels find a block through the stage's
    // block table (ordinary ops on dense slots) or through the per-op pointers the host side resolves (DenseOpPtr).
    // What a block holds when it is handed on: kGarbage (fresh arena memory, or left by an untracked program),
    // kListed (left by a tracked program: all zero except the entries in its list, which is intact).
    using DenseBlock = Index::DenseBlock;  // cap: entries ([cap][W] mask words, then the live list)
    enum : uint8_t { kGarbage = 0, kListed = 1 };
    std::vector<Index::ArenaChunk> block_chunks;  // the blocks' own arena (bump allocation in block_chunks[bcur]); kept with the index
    size_t bcur = 0, bused = 0, block_arena_words = 0;
    Index::BlockBins pool;     // blocks earlier sessions on this index left behind, by capacity
    std::vector<std::vector<DenseBlock>> blocks;  // per program, by block id (p == nullptr: a tracked block no ZERO has created yet)
    std::vector<uint8_t> tracked;                 // per program: TXQ_PROGRAM_TRACKED_BIT (fixed with its first block)
    Index::BlockBins free_blocks;  // blocks of finished programs by capacity, reusable ...
    // ... two stages after they were given back: the stage before the current one may still be running, on another stream
    std::vector<DenseBlock> given_back[2];
    uint32_t block_slots = 0;       // N = A^(k-1) of this session's blobs (0: no dense blob seen yet): the capacity of untracked blocks
    size_t block_bytes_made = 0;
    double block_alloc_seconds = 0;  // spent in hipMalloc for block chunks (TXQ_TRACE)
    size_t n_blocks_live = 0, n_blocks_made = 0, n_block_memsets = 0, n_blocks_relisted = 0, n_sparse_launches = 0, n_sparse_groups = 0;
    std::vector<uint32_t> last_stage;  // per program: the last stage (1-based) that had ops for it
    hipStream_t side = nullptr;        // a stage that continues nothing of the stage in flight runs beside it, on the other stream
    int stream_of_last = 0;            // 0: the caller's stream, 1: `side`
    uint64_t** d_base = nullptr;  // device copy of `base` as of the last stage (lives in that stage's staging set)
    bool owns_cache = false;      // buffers came from / go back to ix->session_cache
    std::vector<void*> retired;   // staging buffers that were outgrown while another stage was running: freed with the session
    Index::StagingSet set[2];     // stage n uses set[n & 1]
    hipStream_t upload = nullptr; // the uploads' stream (non-blocking: independent of the stream the kernels run on)
    unsigned long long* d_step_ctr = nullptr;  // TXQ_TRACE: what sparse_units_kernel did (entries, units, non-empty products, units that left a bit)
    std::vector<unsigned char> host_aux;  // a small stage is packed here and sent as one copy
    // where a stage's wall time goes (reported on stderr at session end when TXQ_TRACE is set)
    double t_validate = 0, t_upload = 0, t_device = 0;
    double t_grow = 0, t_plan = 0, t_wait = 0, t_alloc = 0;  // parts of t_upload: slot regions, units/tiles, waiting for the staging set, scratch
    const char* row_source = "none";  // where the dense steps of the last stage took M[k-mer] from
    // what the dense ops of the session amount to (TXQ_TRACE): predecessor visits and destination suffixes of the steps,
    // slots zeroed, entries reduced — the algorithmic bytes of dense_kernel follow from these and the mask width
    uint64_t n_step_pairs = 0, n_step_suffixes = 0, n_zero_slots = 0, n_reduce_entries = 0;
    size_t n_beside = 0;  // stages that ran on the other stream than their predecessor
    size_t n_stages = 0, bytes_uploaded = 0, n_dense_tiles = 0, n_levels = 0, n_unit_launches = 0, n_units = 0, n_dense_launches = 0;
    ~Session();
};

// The device code of a translation unit is loaded when one of its kernels is first needed — 14 ms for the executor's
// kernels, which a single `tetrex query` would pay inside its query time.  txq_init asks for them right away (the CLI
// calls it on a helper thread while the index file is parsed).
void preload_exec_kernels();
void preload_probe_kernels();
void preload_hibf_kernels();

hipStream_t take_spare_stream(int device);  // a non-blocking stream made at txq_init, or null (txq_api.hip)

int fail(int code, const char* fmt, ...);
int fail_hip(hipError_t e, const char* what);
int ensure(void** p, size_t* cap, size_t bytes);
int alloc_ibf(const txq_ibf_desc& d, uint64_t w0, uint64_t w1, IbfDev* out, uint64_t* bytes);

// txq_probe.hip
hipError_t launch_probe(const IbfDev& f, const uint64_t* kmers, size_t n, uint64_t* masks, uint64_t* alive, hipStream_t s);
hipError_t launch_probe_interleaved(const IbfDev& interleaved, const HibfNode& root, const void* children, uint32_t wpr_log2, const uint64_t* kmers,
                                    size_t n, uint64_t* masks, uint64_t* alive, hipStream_t s);
hipError_t launch_emplace(const IbfDev& f, const uint64_t* values, const uint32_t* bins_of, size_t n, hipStream_t s);

// txq_hibf.hip
int hibf_upload(Index& ix, const txq_index_desc& desc);
int hibf_probe(Index& ix, const Knobs& kn, const uint64_t* d_kmers, size_t n, uint64_t* d_masks, uint64_t* d_alive, hipStream_t s);
// layout-order rows of n k-mers: d_rows[n][v_words]
int hibf_probe_layout_order(Index& ix, const uint64_t* d_kmers, size_t n, uint64_t* d_rows, hipStream_t s);
// final masks of a layout-order session -> user-bin order: d_out[n][shard_words] (zeroed here)
int hibf_layout_to_user(const Index& ix, const uint64_t* d_rows, size_t n, uint64_t* d_out, hipStream_t s);

// txq_exec.hip
int run_programs(Index& ix, const void* blob, size_t blob_bytes, size_t n_programs, uint64_t* d_final, hipStream_t s);
int session_begin(Index& ix, size_t n_programs, Session** out);
int session_stage(Session& s, const void* blob, size_t bytes, const uint32_t* q_prog, const uint32_t* q_slot, size_t n_q,
                  uint8_t* alive, hipStream_t st);
int session_finish(Session& s, uint64_t* d_final, hipStream_t st);

}  // namespace txq

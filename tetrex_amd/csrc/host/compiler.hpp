// Host side: k-graph -> mask-DAG program (include/txq_program.h) — product code.
//
// This is OTFCollector::collect() (reference include/otf_collector.h:341-393) with the masks
// replaced by slot numbers: the host walks the k-graph in topological order, keeps the
// reference's state table (one state per node and (k-1)-symbol k-mer suffix, push/absorb
// :190-208, update_path :247-278, split_procedure :280-288) and EMITS the bit operations instead
// of performing them.  The k-mers a query needs are collected into a batch-wide, deduplicated
// table (the reference's kmer_cache_, :54,260-264), so the device probes each one once.
//
// Differences to the reference, all result-preserving on inputs where the reference's result
// is well defined:
//   * Ghost, Split and '$' nodes never touch a mask, so they are removed up front (epsilon
//     closure): a state leaving a residue node is handed directly to every residue/Match node it
//     can reach.  The reference walks a 20-way wildcard union through 19 Split and 19 Ghost
//     nodes, i.e. ~250 state visits per incoming state; here it is 20;
//   * states are never pruned (path_.none(), :383): a dead state only contributes zero masks;
//   * the state key also contains min(shift_count, k-1), so a path that has not yet seen k-1
//     symbols is never merged into one that has (the reference merges them when the leading
//     residues encode to 0 and then keeps whichever arrived first — implementation-defined,
//     SURVEY.md §7 "state-merge quirk").
#pragma once
#include "encoder.hpp"
#include "flat_map.hpp"
#include "kgraph.hpp"
#include "../../../include/txq_program.h"

#include <array>
#include <atomic>
#include <map>
#include <cstdint>
#include <functional>
#include <string>
#include <memory>
#include <unordered_map>
#include <vector>

namespace tetrex {

// What one query may cost.  Staged execution streams a query's ops to the device stage by stage (slots are recycled),
// so only what the host holds AT ONE TIME needs a bound: the states waiting in the tables (24 bytes each).  A query
// beyond it fails with a message; the reference would (slowly) answer, so the bound is generous.  One-shot
// compilation (ProgramBatch: the whole program in one blob) bounds the totals instead.
struct CompileLimits {
    size_t max_ops = (size_t)1 << 33;            // cumulative, per query: generous (minutes of work) but finite (`tetrex query --max-ops`)
    size_t max_states = (size_t)1 << 33;         // cumulative, per query
    size_t max_live_states = (size_t)96 << 20;   // waiting at the same time, per query (~2.3 GB)
    static CompileLimits one_shot() { return CompileLimits{(size_t)8 << 20, (size_t)8 << 20, (size_t)8 << 20}; }
};

// -a / -g of `tetrex query` (reference include/arg_parse.h:64,68): bypass catastrophic sub-graphs
// with Gap nodes, and filter across a gap with the d-gram index when one is loaded
// (DGramIndex, include/dGramIndex.h; gap_procedure / update_gapped, include/otf_collector.h:216-245,290-312).
struct GapOptions {
    bool augment = false;
    bool dgram_loaded = false;         // has_dibf_
    uint64_t min_gap = 0, max_gap = 0; // DGramIndex::getMinGap / getMaxGap (0 / 0 without -g)
};

// residue code of the d-gram alphabet (DGramTools::aa_to_num): Base code for 'A'..'Z', else 0
uint64_t dgram_residue_code(int symbol);
// the d-gram codes one sequence record contributes (DGramIndex::process_sequence)
void dgram_record_values(std::string_view seq, uint64_t min_gap, uint64_t max_gap, std::vector<uint64_t>& out);

// Dense DP steps (include/txq_program.h, version 3): where a state list saturates, the host keeps it as a
// block of A^(k-1) device slots and emits one op per residue set instead of one per state and residue.
struct DenseOptions {
    bool enabled = false;          // the executor runs dense ops (a flat-IBF device session)
    // (round 2, device-bound bench batch of 1000 motifs at k = 4: min_states / sparse_below 128 / 24: 30 ms; 64 / 24: 23 ms; 32 / 16: 18 ms;
    // round 3, the batches bound by the host's expansion — profiles/r3_dense_thresholds_ab.txt: 32 / 16, 16 / 8, 8 / 4, 4 / 2 give
    // 6.6 / 6.6 / 6.2 / 6.5 ms at 1000 motifs and 44.9 / 33.7 / 33.8 / 33.1 ms at 10 000; every other workload gains or stays)
    uint32_t min_states = 8;       // a list with at least this many full-length states becomes a block
    uint32_t sparse_below = 4;     // a block whose shape holds at most this many entries is enumerated again
    // ... while the run's pool of block memory is roomy (at least half of it left, and 256 blocks or more to begin with);
    // when it is short the blocks go to the lists that need them most (a wildcard's thousands of states, not a residue
    // class's dozen): the thresholds of round 2
    uint32_t short_min_states = 32, short_sparse_below = 16;
    int64_t pool_total = 0;        // what `pool` started with (0: unknown — never short)
    uint32_t cool_down = 1;        // released blocks kept out of circulation while new ones can be had (see can_take_blocks)
    uint32_t max_shape_per_state = 64;  // a list becomes a block only if its shape holds at most this many suffixes per state (fewer for wide masks: QueryExpansion::shape_limit)
    double host_ns_per_op = 12.0;       // what an enumerated state and residue costs the batch (wall time, all expansion threads)
    uint64_t slot_bytes = 0;       // bytes of one mask on the executing device (budgets; 0 = 128)
    uint64_t max_block_bytes = 1ull << 30;
    uint32_t max_blocks = 256;     // per query, live at the same time (fewer where 256 blocks exceed the dense slot space); `pool` is the real bound
    std::atomic<int64_t>* pool = nullptr;  // bytes all queries of a run may still take (null: unlimited)
    // What the run knows about the index (shared by its queries; null = kDense): do the masks of probed states stay full
    // (lists saturate: blocks pay) or thin out (states die within a few residues: enumerating and pruning pays)?  While
    // it is unknown, a query pauses once at its first list that could become a block and asks (QueryExpansion::observe).
    // kThin: states do not just thin out, they are down to a handful of bins after their first probe (fewer than
    // StagedOptions::thin_bits bits per probed state) — most die within a residue or two
    enum Evidence : int { kUnknown = 0, kDense = 1, kSparse = 2, kThin = 3 };
    std::atomic<int>* evidence = nullptr;
    // Tracked (sparse) blocks (include/txq_program.h): the executor keeps a live list per block and pushes steps from the
    // live entries only, so a block costs what its LIVING states cost.  Where the run has learned that states die out
    // (kThin) a query's lists then become blocks as soon as they hold min_states states, whatever their shape —
    // at k = 6 a list of a few thousand states inside 21^5 suffixes.  0: the executor cannot (an HIBF whose steps are
    // not fused, a test double without lists); -1 / +1 force it off / on for every query (tests, A/B).
    bool tracked_ok = false;
    int tracked_force = 0;
};
using DenseVec = std::vector<txq_dense_op>;
// slots of one dense block, A^(k-1), for this encoder — 0 when dense blocks cannot be used with it (k too
// large for a block to fit the limits, or dense steps switched off)
uint64_t dense_block_slots(const KmerEncoder& enc, const DenseOptions& opt);

// op.kmer of an op that ANDs with a d-gram mask: bit 31 set on the index into the d-gram table
// (local to one expansion; run_staged rebases both kinds into the stage's single table, d-grams last)
constexpr uint32_t kDgramFlag = 0x80000000u;

using OpVec = CachedVector<txq_op>;
using KmerVec = CachedVector<uint64_t>;

struct QueryProgram {
    OpVec ops;  // k-mer field indexes the BATCH table
    uint32_t n_slots = TXQ_SLOT_FIRST_FREE;
    uint64_t states = 0, probes = 0;  // statistics
};

// Resumable expansion of ONE query: emits ops node by node (topological order) and can pause
// between nodes so that the device can report which waiting states are already dead.
// k-mer value -> dense index, in first-seen order (one per stage and per expansion thread)
class KmerTable {
  public:
    // exact: every value gets one index (hash map).  Otherwise duplicates are only caught by a small
    // direct-mapped cache: a repeated k-mer may get a second entry and is then simply probed twice,
    // which costs the device less than an exact look-up costs the host (a wildcard position of a
    // peptide query yields 160 000 mostly distinct k-mers).
    explicit KmerTable(bool exact = true) : exact_(exact) {}
    uint32_t intern(uint64_t value) {
        if (exact_) {
            auto [slot, fresh] = index_.emplace(value, (uint32_t)values_.size());
            if (fresh) values_.push_back(value);
            return *slot;
        }
        // the cache starts small (most queries of a large batch need a few dozen k-mers) and grows with the table
        if (values_.size() >= ((size_t)2 << recent_bits_) && recent_bits_ < kRecentBits) {
            recent_bits_ = recent_bits_ + 2 > kRecentBits ? kRecentBits : recent_bits_ + 2;
            recent_.assign((size_t)1 << recent_bits_, 0);  // forgetting the old entries only costs a few repeats
        }
        if (recent_.empty()) recent_.assign((size_t)1 << recent_bits_, 0);
        uint64_t h = value * 0x9E3779B97F4A7C15ULL;
        uint32_t& id = recent_[h >> (64 - recent_bits_)];
        if (id < values_.size() && values_[id] == value) return id;  // entries left by an earlier stage fail this check
        id = (uint32_t)values_.size();
        values_.push_back(value);
        return id;
    }
    const KmerVec& values() const { return values_; }
    void clear() { index_.clear(); values_.clear(); }

  private:
    static constexpr unsigned kRecentBits = 13;
    unsigned recent_bits_ = 5;
    bool exact_;
    FlatMap index_;
    CachedVector<uint32_t> recent_;
    KmerVec values_;
};

class QueryExpansion {
  public:
    using Intern = KmerTable&;
    QueryExpansion(const KmerEncoder& enc, KGraph graph, CompileLimits limits, GapOptions gaps = {}, DenseOptions dense = {});
    // A query that is a plain string of residues (no operator): its k-graph is a chain and its expansion one state walking
    // along it — the ops are written out directly (no graph, no tables): what the constructor above and advance() emit for the
    // chain, op for op.  The common nucleotide query.
    QueryExpansion(const KmerEncoder& enc, std::string literal, CompileLimits limits);
    ~QueryExpansion();

    bool done() const { return literal_mode_ ? literal_done_ : cursor_ >= order_.size(); }
    // Expand whole nodes until the query is finished or `op_budget` ops were emitted by this call.
    // Ops are appended to `out`.  Throws std::runtime_error when a limit is exceeded.
    // verified_only: stop before the first item that would consume a state the device has not yet
    // reported on (see frontier_slots / prune).
    // Dense ops (DenseOptions::enabled) go to `dense`; their op-stream entries index it from its size at the call.
    void advance(size_t op_budget, Intern intern, OpVec& out, KmerTable* dgrams = nullptr, bool verified_only = false,
                 DenseVec* dense = nullptr);
    uint32_t n_slots() const { return high_water_; }
    // dense blocks the query addresses; 0 while it never went dense
    uint32_t n_dense_blocks() const { return (uint32_t)n_blocks_; }  // block ids 0 .. n-1 are in use
    bool tracked() const { return tracked_; }  // its blocks carry live lists (TXQ_PROGRAM_TRACKED_BIT)
    uint64_t dense_steps() const { return dense_steps_; }
    uint64_t pool_taken() const { return pool_taken_; }  // bytes of the run's dense pool this query holds
    uint64_t dense_block_slots() const { return dense_n_; }  // A^(k-1), 0 when dense blocks are off for this query
    // distinct non-constant slots of waiting states that were not asked about before (a waiting
    // state's mask only ever grows, so one answer per state is enough); marks them as asked
    void frontier_slots(std::vector<uint32_t>& out);
    // feedback is pointless where (almost) nothing dies: stop asking after enough evidence
    // (a query that runs dense steps has shown that its lists saturate instead of dying: it stops asking too)
    bool wants_feedback() const { return wants_evidence_ || (dense_steps_ == 0 && (asked_ < 2048 || pruned_ * 50 >= asked_)); }
    // Feedback of the stage, by slot: 0xFF = not asked, 0 = dead, b = 1 + floor(log2(bits set)) (txq_session_stage).  Adds the
    // bits of the asked states that have been probed at least once to *bits and their number to *states.
    void observe(const std::vector<uint8_t>& klass_by_slot, uint64_t* bits, uint64_t* states);
    // where a quarter or more of the states die, expanding an unconfirmed state is mostly wasted work:
    // such a query only expands what the device has confirmed alive (advance(..., verified_only))
    bool mostly_dying() const { return asked_ >= 64 && pruned_ * 4 >= asked_; }
    // drop every waiting state whose slot is listed as dead (dead[slot] != 0)
    void prune(const std::vector<uint8_t>& dead_by_slot);
    uint64_t states() const { return states_; }
    uint64_t probes() const { return probes_; }
    uint64_t pruned() const { return pruned_; }
    uint64_t total_ops() const { return total_ops_; }
    // rough cost of running this query to its end (waiting states x remaining items): orders the
    // tasks of a stage, largest first
    uint64_t weight() const { return literal_mode_ ? (literal_done_ ? 0 : literal_.size()) : (uint64_t)(order_.size() - cursor_) * (waiting_ + 1); }

  private:
    struct State { uint64_t kmer; uint32_t slot; uint8_t shift; uint8_t asked; uint8_t gapped = 0; uint8_t res1 = 0, res2 = 0; };
    using StateVec = CachedVector<State>;
    // by_key merges arrivals with equal keys (the collector's absorb).  Merging is an optimisation, not a
    // requirement: two unmerged states with one key just do the same work twice and OR the same bits into
    // RESULT.  Where a join's list shows that (almost) none of the arrivals made from it can merge —
    // sparse state sets, typical at k = 6 — the receiver gets no table and just appends (`append_only`):
    // its look-ups, growth and initialisation were a third of the expansion time there.
    // a block of the dense region as one list's (part of the) state set; shape[j] = codes that can occur at suffix
    // position j (0 = oldest): a superset, entries outside the real set are zero masks
    // phase: how many residues the block's states have seen — k-1 for ordinary (full-length) states; tracked programs also
    // keep the lists of SHORTER states as blocks (TXQ_DENSE_NOPROBE steps), one block per length
    struct DenseRef { uint32_t block; uint32_t owned; uint32_t phase; uint32_t shape[TXQ_DENSE_MAX_POSITIONS]; };
    struct NodeStates { StateVec items; FlatMap by_key; bool append_only = false; std::vector<DenseRef> dense; };
    static constexpr uint32_t kMergeSample = 4096;  // lists shorter than this are not worth the question
    bool merging_pays(const StateVec& list);
    uint64_t key_of(const State& s) const;
    static constexpr uint32_t kSearched = 4, kNoState = 0xFFFFFFFFu;  // lists shorter than kSearched have no merge table (arrive)
    uint32_t code_mask(int32_t node) const;
    bool literal_mode_ = false, literal_done_ = false;  // a plain string of residues: expanded without a graph (advance)
    std::string literal_;
    int32_t resume_item_ = KGraph::kNone;  // a fused class whose residues from resume_residue_ on are still to be rolled in (advance)
    uint32_t resume_residue_ = 0;
    static size_t merge_sample_threshold();
    const KmerEncoder& enc_;
    KGraph g_;
    CompileLimits limits_;
    GapOptions gaps_;
    // Derived, epsilon-free graph.  Items 0..n-1 are the k-graph's nodes (only residue and Match
    // nodes are ever visited), item n is the entry, items > n are JOINS: one per distinct set of
    // >= 2 residue/Match nodes reachable through Ghost/Split/'$' nodes.  A state leaving a residue
    // node goes to that node's join (or straight to its only target); the join merges equal
    // states ONCE and then fans them out to its targets.
    std::vector<int32_t> order_;      // residue, Match and join items, topologically sorted
    size_t cursor_ = 0;
    std::vector<int32_t> forward_;    // per item: the join or single target it hands states to, kNone = none
    std::vector<uint32_t> fan_first_; // per join (item - n - 1): CSR into fan_
    std::vector<int32_t> fan_;
    std::vector<uint8_t> dangling_;   // some path out of the item ends in a node without successor
    std::vector<uint8_t> single_source_;  // item fed by exactly one item: its arrivals cannot collide
    int32_t n_nodes_ = 0;
    std::vector<NodeStates> table_;
    // A join's merged list is read in place by the targets it alone feeds (instead of one copy per
    // target): input_of_[target] = that join while the target waits, readers_[join] = targets still to
    // read the list, open_joins_ = joins whose list is kept for readers.
    std::vector<int32_t> input_of_;
    std::vector<uint32_t> readers_;
    std::vector<int32_t> open_joins_;
    unsigned direct_key_bits_ = 0;    // FlatMap::want_direct for the merge tables (0: keys too wide)
    std::vector<uint32_t> refs_;
    // Freed slots are recycled oldest-first and only after the node item that freed them is
    // finished: immediate (LIFO) reuse would chain unrelated ops through write-after-read
    // hazards on the recycled slot and serialise the dependency levels of the schedule.
    std::vector<uint32_t> free_, parked_;
    size_t free_head_ = 0;
    uint32_t high_water_ = TXQ_SLOT_FIRST_FREE;
    uint64_t states_ = 0, probes_ = 0, pruned_ = 0, total_ops_ = 0, asked_ = 0, waiting_ = 0;
    // storage of consumed items, reused by the items that fill next (keeps the allocator out of the loop)
    std::vector<StateVec> spare_items_;
    std::vector<FlatMap> spare_maps_;
    std::vector<uint32_t> seen_;  // frontier_slots: slot -> epoch
    uint32_t seen_epoch_ = 0;

    // ---- dense blocks ----
    DenseOptions dense_;
    bool dense_ok_ = false;
    bool wants_evidence_ = false, evidence_asked_ = false;  // paused before the first list that could become a block
    bool tracked_ = false, tracked_decided_ = false;        // decided with the query's first block
    void decide_tracking();
    // per block: where its DENSE_ZERO sits in the stage's dense table, while that table is still being filled (see shape_zero)
    std::vector<uint32_t> zero_at_, zero_epoch_;
    uint32_t dense_epoch_ = 1;
    size_t dense_seen_ = 0;
    unsigned dense_pos_ = 0;      // k - 1
    uint32_t dense_a_ = 0;        // alphabet size A
    uint64_t dense_n_ = 0;        // A^(k-1)
    uint64_t n_blocks_ = 0, pool_taken_ = 0, dense_steps_ = 0;
    // A block id keeps its capacity (entries) for the life of the query; released ids wait in the list of their capacity.
    // Untracked blocks: A^(k-1) entries, full geometry.  Tracked blocks: laid out inside the geometry of the list they
    // belong to (static_shape_), capacity = the product of its sets rounded up to a power of two.
    using Geometry = std::array<uint32_t, TXQ_DENSE_MAX_POSITIONS>;
    struct FreeList { std::vector<uint32_t> ids; size_t head = 0; };
    std::map<uint32_t, FreeList> free_by_cap_;
    std::vector<uint32_t> block_refs_, block_cap_, parked_blocks_;
    std::vector<Geometry> block_geom_;
    // per item: the codes that can occur at each position of the (k-1)-suffix of a full-length state waiting there (a
    // superset, from one pass over the derived graph)
    std::vector<Geometry> static_shape_;
    void compute_static_shapes();
    Geometry geometry_of(int32_t item, unsigned phase) const;
    uint32_t capacity_of(const Geometry& g) const;
    DenseVec* dense_out_ = nullptr;
    static uint32_t dense_slot(uint32_t block, uint64_t index) { return TXQ_DENSE_SLOT_BIT | (block << TXQ_DENSE_BLOCK_SHIFT) | (uint32_t)index; }
    uint64_t dense_index(uint32_t block, uint64_t kmer) const;
    uint64_t index_of_codes(uint32_t block, const unsigned* code) const;
    uint64_t shape_entries(const DenseRef& r) const;
    bool can_take_blocks(const std::vector<uint32_t>& caps);
    std::vector<uint32_t> caps_scratch_;
    uint32_t new_block(OpVec& out, const Geometry& geom);
    void release_block(uint32_t block);
    void emit_dense(OpVec& out, const txq_dense_op& d);
    DenseRef* owned_block(int32_t item, NodeStates& ns, OpVec& out, unsigned phase);
    static bool has_owned(const NodeStates& ns, unsigned phase) {
        for (const DenseRef& r : ns.dense)
            if (r.owned && r.phase == phase) return true;
        return false;
    }
    void densify(int32_t item, NodeStates& ns, OpVec& out, bool may_hold_duplicates);
    void shape_zero(const DenseRef& r);
    void shape_unzero(const DenseRef& r);
    uint64_t shape_limit() const;
    void materialise(int32_t item, OpVec& out, bool all);
    void dense_receivers(int32_t item, std::vector<int32_t>& out) const;
    std::vector<int32_t> receivers_scratch_;
    void dense_step(const DenseRef& src, uint32_t r_mask, int32_t receiver, OpVec& out);
    // the run's pool of block memory is short: less than half of it left, or never more than 256 blocks to begin with (full
    // blocks of A^(k-1) entries, or 16 MB where those are larger: at k = 6 blocks are tracked and laid out inside their geometry)
    bool pool_short() const {
        if (!dense_.pool || dense_.pool_total <= 0) return false;
        const int64_t block = std::min<int64_t>((int64_t)(dense_n_ * (dense_.slot_bytes ? dense_.slot_bytes : 128)), (int64_t)16 << 20);
        return dense_.pool->load(std::memory_order_relaxed) * 2 < dense_.pool_total || dense_.pool_total < 256 * block;
    }
    uint32_t min_states_now() const { return pool_short() ? std::max(dense_.min_states, dense_.short_min_states) : dense_.min_states; }
    bool small_enough(const DenseRef& r) const { return shape_entries(r) <= (pool_short() ? std::max(dense_.sparse_below, dense_.short_sparse_below) : dense_.sparse_below); }

    uint32_t fresh();
    void share(uint32_t s);
    void drop(uint32_t s);
    bool exclusive(uint32_t s) const;
    void arrive(int32_t to, State s, OpVec& out);
    void adopt_storage(NodeStates& ns);
    void hand_on(int32_t from, State s, OpVec& out);
    void emit(OpVec& out, uint32_t kmer, uint32_t dst, uint32_t a, uint32_t b);
};

// Reorders `ops` (one program's ops of one stage, in a valid sequential order) into dependency
// levels as defined in txq_program.h (version 2) and returns the end index of every level.
// Scratch vectors are reused across calls (sized to n_slots).
struct LevelScratch {
    struct Slot { uint32_t stamp, wr, rd, acc; };  // last full write / last read / last accumulation level of a slot, valid when stamp == epoch
    // a dense block: last dense write / dense read / ordinary write (of either kind) / ordinary read / ordinary FULL write of
    // one of its slots.  The slots of a block have no records of their own: an ordinary op on one of them is ordered
    // against the block (a block holds thousands of slots, an op on one of them would be a cache miss per record)
    struct Block { uint32_t stamp, dw, dr, sw, sr, fw; };
    std::vector<Slot> slot;
    std::vector<Block> block;
    uint32_t epoch = 0;
    std::vector<uint32_t> level_of, pos;
    OpVec sorted;
};
std::vector<uint32_t> schedule_levels(OpVec& ops, uint32_t n_slots, LevelScratch& scratch);
// Same, but leaves `ops` alone and writes the reordered ops to `dst` (room for ops.size()), adding
// `kmer_add` to every k-mer index and `dgram_add` to every (flag-stripped) d-gram index on the way.
// dense ops of the program (op.dst indexes `table`; `index_add` rebases it into the stage's table)
struct DenseSchedule { const txq_dense_op* table; uint32_t index_add; uint32_t n_blocks; };
std::vector<uint32_t> schedule_levels_into(const OpVec& ops, uint32_t n_slots, LevelScratch& scratch, txq_op* dst,
                                           uint32_t kmer_add, uint32_t dgram_add, const DenseSchedule* dense = nullptr);

// What executes a stage: the GPU session (device_index.cpp) or a test double.
struct StageExecutor {
    virtual ~StageExecutor() = default;
    // Runs the NEW ops of every program (blob in txq_program.h format, all programs present,
    // 8-byte aligned) and answers alive[i] = 0 when slot query_slot[i] of program query_program[i] has no bit set,
    // otherwise 1 + floor(log2(bits set)) (a plain 1 is fine for an executor that does not count: "alive, nearly empty").
    virtual void stage(const uint8_t* blob, size_t blob_bytes, const std::vector<uint32_t>& query_program,
                       const std::vector<uint32_t>& query_slot, std::vector<uint8_t>& alive) = 0;
};

struct StagedOptions {
    GapOptions gaps;
    int threads = 0;                         // expansion threads (0 = all hardware threads)
    size_t ops_per_query_per_stage = 4096;   // pause a query for feedback after this many new ops
    size_t ops_per_stage = 16u << 20;        // bound on one stage's blob (256 MiB of ops)
    size_t ops_per_task = 256u << 10;        // a feedback-free query's first budget; it quadruples with every further stage the query
                                             // needs (up to 16x), so a skewed batch is not held to many stages by its heaviest query
    size_t stage_target_ops = 4u << 20;      // with few queries left, each gets a larger share of this
    size_t wave_ops = 40000;                 // queries BEGIN in waves of about this many ops — a wave is at least as large as everything emitted before
                                             // it, so a large batch does not become many small stages —, so that the device runs wave n while the host
                                             // expands wave n+1; 0: everybody begins in the first stage (profiles/r3_wave_size_ab.txt, last part:
                                             // 1000 / 10 000 motifs 6.4 / 31 ms with one wave of 96 k growing by halves, 5.4 / 30.5 ms with 40 k doubling)
    bool verified_levels = true;             // queries that still ask for feedback only expand states confirmed alive
    CompileLimits limits;
    DenseOptions dense;                      // run_staged fills `pool` itself
    uint64_t dense_pool_bytes = 48ull << 30; // device memory all dense blocks of a run may take
    int dense_evidence = DenseOptions::kUnknown;  // what earlier runs learned about this index (StagedStats::dense_evidence); kUnknown: ask
    uint64_t feedback_bins = 0;              // bins the executor's answers count bits over (a column shard); 0 = all bins
    double dense_min_fill = 0.25;            // masks of probed states at least this full on average: lists saturate, blocks pay
    double thin_bits = 16;                   // ... fewer bits than this per probed state on average: states die out (DenseOptions::kThin), tracked blocks
};

struct StagedStats {
    size_t stages = 0;
    uint64_t ops = 0, kmers = 0, states = 0, pruned = 0, feedback_queries = 0, dense_ops = 0, tracked_queries = 0;
    double expand_seconds = 0, execute_seconds = 0;  // host expansion vs. StageExecutor::stage
    int dense_evidence = DenseOptions::kUnknown;     // what this run knew / learned about the index (for the next run on it)
    double observed_fill = -1;                       // mean fraction of bits set in the masks it looked at (-1: it did not look)
};

// Drives a batch of queries through staged execution.  status[i] != 0: query i could not be
// expanded (its program stays empty, result mask zero); messages[i] says why.
StagedStats run_staged(const KmerEncoder& enc, uint64_t bins, const std::vector<std::string>& regexes, StageExecutor& exec,
                       const StagedOptions& opt, std::vector<int>* status, std::vector<std::string>* messages);

// Column shards of the masks of n queries -> full-width masks: shard r holds words [word0[r], word0[r] + words[r]) of
// every mask, row-major [n][words[r]] (what txq_session_end returns for that shard).  The shards must tile
// [0, mask_words) exactly.  This is the whole "OR-reduce" of the bin-sharded index: the shards are disjoint.
std::vector<uint64_t> join_shard_masks(size_t n, uint64_t mask_words, const std::vector<uint64_t>& word0, const std::vector<uint64_t>& words,
                                       const std::vector<const uint64_t*>& shard_masks);

// A batch of queries sharing one k-mer table; serialises to the blob txq_run_programs takes.
class ProgramBatch {
  public:
    explicit ProgramBatch(const KmerEncoder& enc, CompileLimits limits = CompileLimits::one_shot()) : enc_(enc), limits_(limits) {}

    // Compile one k-graph; throws std::runtime_error when a limit is exceeded.
    size_t add(const KGraph& g);
    // A query that needs no device work (1-bin index): result mask is slot ONES.
    size_t add_passthrough();
    // A placeholder for a query that could not be compiled: no ops, result mask zero.
    size_t add_empty();

    size_t size() const { return programs_.size(); }
    size_t kmer_count() const { return table_.values().size(); }
    const KmerVec& kmers() const { return table_.values(); }
    const QueryProgram& program(size_t i) const { return programs_[i]; }
    std::vector<uint8_t> serialise() const;

  private:
    const KmerEncoder& enc_;
    CompileLimits limits_;
    KmerTable table_;
    std::vector<QueryProgram> programs_;
};

}  // namespace tetrex

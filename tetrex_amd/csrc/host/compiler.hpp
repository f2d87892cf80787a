// Host side: k-graph -> mask-DAG program (include/txq_program.h) — product code.
//
// This is OTFCollector::collect() (reference include/otf_collector.h:341-393) with the masks
// replaced by slot numbers: the host walks the k-graph in topological order, keeps the
// reference's state table (one state per node and (k-1)-symbol k-mer suffix, push/absorb
// :190-208, update_path :247-278, split_procedure :280-288) and EMITS the bit operations instead
// of performing them.  The k-mers a query needs are collected into a batch-wide, deduplicated
// table (the reference's kmer_cache_, :54,260-264), so the device probes each one once.
//
// Differences to the reference, both result-preserving on inputs where the reference's result
// is well defined:
//   * states are never pruned (path_.none(), :383): a dead state only contributes zero masks;
//   * the state key also contains min(shift_count, k-1), so a path that has not yet seen k-1
//     symbols is never merged into one that has (the reference merges them when the leading
//     residues encode to 0 and then keeps whichever arrived first — implementation-defined,
//     SURVEY.md §7 "state-merge quirk").
#pragma once
#include "encoder.hpp"
#include "kgraph.hpp"
#include "../../../include/txq_program.h"

#include <cstdint>
#include <functional>
#include <string>
#include <memory>
#include <unordered_map>
#include <vector>

namespace tetrex {

struct CompileLimits {
    size_t max_ops = 8u << 20;     // per query
    size_t max_states = 8u << 20;  // per query
};

struct QueryProgram {
    std::vector<txq_op> ops;  // k-mer field indexes the BATCH table
    uint32_t n_slots = TXQ_SLOT_FIRST_FREE;
    uint64_t states = 0, probes = 0;  // statistics
};

// Resumable expansion of ONE query: emits ops node by node (topological order) and can pause
// between nodes so that the device can report which waiting states are already dead.
class QueryExpansion {
  public:
    using Intern = std::function<uint32_t(uint64_t)>;  // k-mer value -> index in the current stage's table
    QueryExpansion(const KmerEncoder& enc, KGraph graph, CompileLimits limits);

    bool done() const { return cursor_ >= order_.size(); }
    // Expand whole nodes until the query is finished or `op_budget` ops were emitted by this call.
    // Ops are appended to `out`.  Throws std::runtime_error when a limit is exceeded.
    void advance(size_t op_budget, const Intern& intern, std::vector<txq_op>& out);
    uint32_t n_slots() const { return high_water_; }
    // distinct non-constant slots held by states that wait at unexpanded nodes
    void frontier_slots(std::vector<uint32_t>& out) const;
    // drop every waiting state whose slot is listed as dead (dead[slot] != 0)
    void prune(const std::vector<uint8_t>& dead_by_slot);
    uint64_t states() const { return states_; }
    uint64_t probes() const { return probes_; }
    uint64_t pruned() const { return pruned_; }
    uint64_t total_ops() const { return total_ops_; }

  private:
    struct State { uint64_t kmer; uint32_t slot; uint8_t shift; };
    struct NodeStates { std::vector<State> items; std::unordered_map<uint64_t, uint32_t> by_key; };
    const KmerEncoder& enc_;
    KGraph g_;
    CompileLimits limits_;
    std::vector<int32_t> order_;
    size_t cursor_ = 0;
    std::vector<NodeStates> table_;
    std::vector<uint32_t> refs_, free_;
    uint32_t high_water_ = TXQ_SLOT_FIRST_FREE;
    uint64_t states_ = 0, probes_ = 0, pruned_ = 0, total_ops_ = 0;

    uint32_t fresh();
    void share(uint32_t s);
    void drop(uint32_t s);
    bool exclusive(uint32_t s) const;
    void arrive(int32_t to, State s, std::vector<txq_op>& out);
    void emit(std::vector<txq_op>& out, uint32_t kmer, uint32_t dst, uint32_t a, uint32_t b);
};

// What executes a stage: the GPU session (device_index.cpp) or a test double.
struct StageExecutor {
    virtual ~StageExecutor() = default;
    // Runs the NEW ops of every program (blob in txq_program.h format, all programs present) and
    // answers alive[i] = slot query_slot[i] of program query_program[i] has a bit set.
    virtual void stage(const std::vector<uint8_t>& blob, const std::vector<uint32_t>& query_program,
                       const std::vector<uint32_t>& query_slot, std::vector<uint8_t>& alive) = 0;
};

struct StagedOptions {
    size_t ops_per_query_per_stage = 4096;   // pause a query for feedback after this many new ops
    size_t ops_per_stage = 4u << 20;         // bound on one stage's blob
    CompileLimits limits;
};

struct StagedStats {
    size_t stages = 0;
    uint64_t ops = 0, kmers = 0, states = 0, pruned = 0, feedback_queries = 0;
};

// Drives a batch of queries through staged execution.  status[i] != 0: query i could not be
// expanded (its program stays empty, result mask zero); messages[i] says why.
StagedStats run_staged(const KmerEncoder& enc, uint64_t bins, const std::vector<std::string>& regexes, StageExecutor& exec,
                       const StagedOptions& opt, std::vector<int>* status, std::vector<std::string>* messages);

// A batch of queries sharing one k-mer table; serialises to the blob txq_run_programs takes.
class ProgramBatch {
  public:
    explicit ProgramBatch(const KmerEncoder& enc, CompileLimits limits = {}) : enc_(enc), limits_(limits) {}

    // Compile one k-graph; throws std::runtime_error when a limit is exceeded.
    size_t add(const KGraph& g);
    // A query that needs no device work (1-bin index): result mask is slot ONES.
    size_t add_passthrough();
    // A placeholder for a query that could not be compiled: no ops, result mask zero.
    size_t add_empty();

    size_t size() const { return programs_.size(); }
    size_t kmer_count() const { return kmers_.size(); }
    const std::vector<uint64_t>& kmers() const { return kmers_; }
    const QueryProgram& program(size_t i) const { return programs_[i]; }
    std::vector<uint8_t> serialise() const;

  private:
    const KmerEncoder& enc_;
    CompileLimits limits_;
    std::vector<uint64_t> kmers_;
    std::unordered_map<uint64_t, uint32_t> kmer_index_;
    std::vector<QueryProgram> programs_;

    uint32_t intern(uint64_t value);
};

}  // namespace tetrex

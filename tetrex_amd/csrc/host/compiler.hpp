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
#include <string>
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

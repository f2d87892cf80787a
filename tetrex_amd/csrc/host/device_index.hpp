// Host side: an index resident on the GPU, and its construction/query through the C-ABI
// (include/txq.h).  This is the host mirror of the reference's TetrexIndex façade for the
// query path: spawn_agent() -> upload, query() -> probe / run_programs
// (reference include/index_base.h:104-107,145-148).  Product code; requires a gfx950 GPU.
#pragma once
#include "compiler.hpp"
#include "encoder.hpp"
#include "index_file.hpp"
#include "../../../include/txq.h"

#include <string>
#include <vector>

namespace tetrex {

// Throws std::runtime_error carrying txq_last_error() when a txq call fails.
void txq_check(int rc, const char* what);

// StageExecutor over a txq session: slot masks stay in HBM between stages.
class TxqStageExecutor final : public StageExecutor {
  public:
    TxqStageExecutor(txq_index* ix, size_t n_programs, txq_index* aux = nullptr);
    ~TxqStageExecutor() override;
    void stage(const uint8_t* blob, size_t blob_bytes, const std::vector<uint32_t>& query_program,
               const std::vector<uint32_t>& query_slot, std::vector<uint8_t>& alive) override;
    // copies every program's RESULT mask (n_programs x shard_words words) and ends the session
    void finish(uint64_t* masks);

  private:
    txq_session* session_ = nullptr;
};

// StageExecutor over the column shards of one index, each with its own txq session (and, with several GPUs, its own
// device — include/txq.h txq_init).  ONE frontier expansion feeds all shards: a stage's blob goes to every session at the
// same time (one host thread per shard), and a waiting state counts as alive when any shard says so (the reference's
// path_.none() over the whole mask, include/otf_collector.h:383).  This is the seam of run_collection /
// run_multiple_queries (reference include/query.h:250-290,329-346) for a bin-sharded index.
class ShardedStageExecutor final : public StageExecutor {
  public:
    ShardedStageExecutor(const std::vector<txq_index*>& shards, size_t n_programs, const std::vector<txq_index*>& aux = {});
    ~ShardedStageExecutor() override;
    void stage(const uint8_t* blob, size_t blob_bytes, const std::vector<uint32_t>& query_program,
               const std::vector<uint32_t>& query_slot, std::vector<uint8_t>& alive) override;
    // ends the sessions and joins the shards' RESULT masks: n_programs x mask_words words
    std::vector<uint64_t> finish();

  private:
    std::vector<txq_session*> sessions_;
    std::vector<txq_index_info> info_;
    size_t n_programs_ = 0;
};

// Whole queries on an uploaded index: staged expansion + device execution.
std::vector<uint64_t> run_queries(txq_index* ix, const KmerEncoder& enc, const std::vector<std::string>& regexes,
                                  std::vector<int>* status, std::vector<std::string>* messages, StagedStats* stats,
                                  const StagedOptions* options, txq_index* aux = nullptr, uint64_t* into = nullptr);
// ... on all column shards of an index: full-width masks (n x mask_words)
std::vector<uint64_t> run_queries_sharded(const std::vector<txq_index*>& shards, const KmerEncoder& enc, const std::vector<std::string>& regexes,
                                          std::vector<int>* status, std::vector<std::string>* messages, StagedStats* stats,
                                          const StagedOptions* options, const std::vector<txq_index*>& aux = {});

class DeviceIndex {
  public:
    DeviceIndex() = default;
    DeviceIndex(const DeviceIndex&) = delete;
    DeviceIndex& operator=(const DeviceIndex&) = delete;
    ~DeviceIndex();

    // txq_init alone (HIP start-up costs ~0.5 s): `tetrex query` runs it on a helper thread while the index file is mapped and parsed
    static void warm_up(const std::vector<int>& devices);
    // txq_init + txq_index_upload of a parsed index file: ONE shard (shard_rank of n_shards) on one device
    void upload(const IndexImage& image, int device = 0, int shard_rank = 0, int n_shards = 1);
    // ... ALL n_shards column shards, dealt round-robin over `devices` (`tetrex query --gpus N`); queries then run on
    // every shard at once and query_masks returns full-width masks
    void upload_sharded(const IndexImage& image, const std::vector<int>& devices, int n_shards);
    size_t n_shards_held() const { return shards_.size(); }
    // words per mask that query_masks returns (the shard's words for upload(), the full mask for upload_sharded())
    uint64_t result_words() const { return shards_.size() > 1 ? info_.mask_words : info_.shard_words; }
    const txq_index_info& info() const { return info_; }
    KmerEncoder encoder() const { return enc_; }
    uint64_t bins() const { return info_.user_bins; }

    // candidate-bin masks for a batch of queries: n x shard_words words
    // status[i] != 0: query i could not be compiled (its mask is zero); messages[i] says why
    // Staged execution (compiler.hpp run_staged): the frontier is expanded on the host and
    // streamed to the device piecewise, with dead-state feedback between stages.
    std::vector<uint64_t> query_masks(const std::vector<std::string>& regexes, std::vector<int>* status = nullptr,
                                      std::vector<std::string>* messages = nullptr, StagedStats* stats = nullptr,
                                      const StagedOptions* options = nullptr);
    // `tetrex query -g`: upload the d-gram index next to the main index (same device, same shard)
    void attach_dgram(const DgramImage& dgram);
    bool has_dgram() const { return aux_ != nullptr; }
    uint64_t dgram_min_gap() const { return dgram_min_; }
    uint64_t dgram_max_gap() const { return dgram_max_; }

  private:
    txq_index* upload_one(const IndexImage& image, int shard_rank, int n_shards);
    std::vector<txq_index*> shards_;      // upload_sharded: all shards (shards_[0] == ix_); upload: just ix_
    std::vector<txq_index*> aux_shards_;  // the d-gram index, sharded the same way
    txq_index* ix_ = nullptr;
    txq_index* aux_ = nullptr;
    uint64_t dgram_min_ = 0, dgram_max_ = 0;
    int shard_rank_ = 0, n_shards_ = 1;
    txq_index_info info_{};
    KmerEncoder enc_;
};

// ascending ids of the set bits (compute_set_bins, reference src/query.cpp:40-75)
std::vector<uint64_t> set_bins(const uint64_t* mask, uint64_t bins);

// IBFIndex::compute_bitcount (reference include/index_ibf.h:133-139)
uint64_t compute_bitcount(uint64_t n, float fpr);

struct BuildOptions {
    unsigned k = 6;
    float fpr = 0.05f;
    unsigned hash_count = 3;
    bool dna = false;
    bool hibf = true;       // the reference's default flavour
    unsigned reduction = 0; // 0 None, 1 murphy, 2 li
    bool dna_wraparound = true;  // reproduce include/nucleotide_decomposer.h:106-110
    int device = 0;
};
// `tetrex index`: FASTA files -> index image, bits set on the GPU (txq_emplace_device).
IndexImage build_index(const std::vector<std::string>& bin_files, const BuildOptions& opt, size_t* n_sequences = nullptr);
// `tetrex track`: FASTA files -> d-gram index (reference src/dGramIndex.cpp:22-38, include/dGramIndex.h:105-157)
DgramImage build_dgram_index(const std::vector<std::string>& bin_files, uint64_t min_gap, uint64_t max_gap, unsigned hash_count,
                             float fpr, int device = 0);

}  // namespace tetrex

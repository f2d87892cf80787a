// C entry point for whole-query execution on a GPU-resident index (libtetrex_query.so).
#include "../../../include/txh.h"
#include "device_index.hpp"

#include <string>

using namespace tetrex;

namespace {
thread_local std::string g_qerr;
thread_local uint64_t g_dense_ops = 0, g_tracked = 0;

}  // namespace

extern "C" {

const char* txe_last_error(void) { return g_qerr.c_str(); }
uint64_t txe_last_dense_ops(void) { return g_dense_ops; }
uint64_t txe_last_tracked_queries(void) { return g_tracked; }

int txe_query_masks(void* handle, int dna, unsigned k, unsigned reduction, const char* const* regex, size_t n,
                    size_t ops_per_query_per_stage, uint64_t* masks, int* status, uint64_t* stats6) {
    return txe_query_masks_gapped(handle, nullptr, nullptr, dna, k, reduction, regex, n, ops_per_query_per_stage, masks, status, stats6);
}

int txe_query_masks_text(void* handle, int dna, unsigned k, unsigned reduction, const char* text, size_t text_bytes, size_t n,
                         size_t ops_per_query_per_stage, uint64_t* masks, int* status, uint64_t* stats6) {
    // the lines of the text as C strings: one copy of the text with the separators turned into terminators
    std::string copy(text ? text : "", text ? text_bytes : 0);
    std::vector<const char*> lines;
    lines.reserve(n);
    size_t at = 0;
    while (lines.size() < n && at <= copy.size()) {
        lines.push_back(copy.data() + at);
        const size_t nl = copy.find('\n', at);
        if (nl == std::string::npos) { at = copy.size() + 1; break; }
        copy[nl] = '\0';
        at = nl + 1;
    }
    if (lines.size() != n || (at < copy.size())) {
        g_qerr = "the text does not hold the stated number of motifs (one per line)";
        return -1;
    }
    return txe_query_masks_gapped(handle, nullptr, nullptr, dna, k, reduction, lines.data(), n, ops_per_query_per_stage, masks, status, stats6);
}

int txe_query_masks_gapped(void* handle, void* aux_handle, const txh_gap_options* gaps, int dna, unsigned k, unsigned reduction,
                           const char* const* regex, size_t n, size_t ops_per_query_per_stage, uint64_t* masks, int* status,
                           uint64_t* stats6) {
    try {
        txq_index* ix = static_cast<txq_index*>(handle);
        txq_index* aux = static_cast<txq_index*>(aux_handle);
        const KmerEncoder enc(dna ? Molecule::DNA : Molecule::Peptide, k, (Alphabet)reduction);
        std::vector<std::string> rx(regex, regex + n);
        StagedOptions opt;
        if (ops_per_query_per_stage) opt.ops_per_query_per_stage = ops_per_query_per_stage;
        if (gaps) opt.gaps = GapOptions{gaps->augment != 0, gaps->dgram_loaded != 0 && aux != nullptr, gaps->min_gap, gaps->max_gap};
        std::vector<int> st;
        std::vector<std::string> why;
        StagedStats s;
        (void)run_queries(ix, enc, rx, &st, &why, &s, &opt, aux, masks);  // (straight into the caller's array)
        g_dense_ops = s.dense_ops;
        g_tracked = s.tracked_queries;
        int failures = 0;
        for (size_t i = 0; i < n; ++i) {
            if (status) status[i] = st[i];
            if (st[i]) { ++failures; g_qerr = "query " + std::to_string(i) + ": " + why[i]; }
        }
        if (stats6) {
            stats6[0] = s.stages; stats6[1] = s.ops; stats6[2] = s.kmers; stats6[3] = s.states; stats6[4] = s.pruned; stats6[5] = s.feedback_queries;
            stats6[6] = (uint64_t)(s.expand_seconds * 1e6); stats6[7] = (uint64_t)(s.execute_seconds * 1e6);
        }
        return failures;
    } catch (const std::exception& e) {
        g_qerr = e.what();
        return -1;
    }
}

int txe_query_masks_sharded(void* const* handles, void* const* aux_handles, size_t n_shards, const txh_gap_options* gaps, int dna,
                            unsigned k, unsigned reduction, const char* const* regex, size_t n, size_t ops_per_query_per_stage,
                            uint64_t* masks, int* status, uint64_t* stats6) {
    try {
        std::vector<txq_index*> shards, aux;
        for (size_t r = 0; r < n_shards; ++r) shards.push_back(static_cast<txq_index*>(handles[r]));
        if (aux_handles)
            for (size_t r = 0; r < n_shards; ++r) aux.push_back(static_cast<txq_index*>(aux_handles[r]));
        const KmerEncoder enc(dna ? Molecule::DNA : Molecule::Peptide, k, (Alphabet)reduction);
        std::vector<std::string> rx(regex, regex + n);
        StagedOptions opt;
        if (ops_per_query_per_stage) opt.ops_per_query_per_stage = ops_per_query_per_stage;
        if (gaps) opt.gaps = GapOptions{gaps->augment != 0, gaps->dgram_loaded != 0 && !aux.empty(), gaps->min_gap, gaps->max_gap};
        std::vector<int> st;
        std::vector<std::string> why;
        StagedStats s;
        const std::vector<uint64_t> out = run_queries_sharded(shards, enc, rx, &st, &why, &s, &opt, aux);
        std::copy(out.begin(), out.end(), masks);
        g_dense_ops = s.dense_ops;
        g_tracked = s.tracked_queries;
        int failures = 0;
        for (size_t i = 0; i < n; ++i) {
            if (status) status[i] = st[i];
            if (st[i]) { ++failures; g_qerr = "query " + std::to_string(i) + ": " + why[i]; }
        }
        if (stats6) {
            stats6[0] = s.stages; stats6[1] = s.ops; stats6[2] = s.kmers; stats6[3] = s.states; stats6[4] = s.pruned; stats6[5] = s.feedback_queries;
            stats6[6] = (uint64_t)(s.expand_seconds * 1e6); stats6[7] = (uint64_t)(s.execute_seconds * 1e6);
        }
        return failures;
    } catch (const std::exception& e) {
        g_qerr = e.what();
        return -1;
    }
}

}  // extern "C"

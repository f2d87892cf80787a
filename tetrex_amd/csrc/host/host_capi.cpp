// C entry points of the host front-end (include/txh.h).
#include "../../../include/txh.h"
#include "compiler.hpp"
#include "encoder.hpp"
#include "index_file.hpp"
#include "kgraph.hpp"
#include "matcher.hpp"
#include "regex_front.hpp"

#include <cstring>
#include <memory>
#include <string>

using namespace tetrex;

namespace {
thread_local std::string g_err;
int fail(const std::string& m, int code = -1) { g_err = m; return code; }
int put(const std::string& s, char* out, size_t cap) {
    if (s.size() + 1 > cap) return fail("output buffer too small", -2);
    std::memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}
KmerEncoder encoder(int dna, unsigned k, unsigned reduction) {
    return KmerEncoder(dna ? Molecule::DNA : Molecule::Peptide, k, (Alphabet)reduction);
}
}  // namespace

struct txh_blob {
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> stats;
};

struct txh_index {
    IndexImage image;
    std::vector<uint8_t> file;
};

extern "C" {

const char* txh_last_error(void) { return g_err.c_str(); }

int txh_translate(const char* regex, char* out, size_t cap) { return put(translate(regex), out, cap); }

int txh_preprocess(const char* regex, int dna, unsigned k, unsigned reduction, char* preprocessed, size_t cap1,
                   char* postfix, size_t cap2) {
    try {
        KmerEncoder enc = encoder(dna, k, reduction);
        std::string pre;
        std::string post = preprocess_query(regex, enc, &pre);
        int r = put(pre, preprocessed, cap1);
        if (r < 0) return r;
        return put(post, postfix, cap2);
    } catch (const std::exception& e) { return fail(e.what()); }
}

int txh_kgraph(const char* postfix, unsigned k, int reduced, int32_t* labels, int32_t* next_a, int32_t* next_b, int32_t cap) {
    try {
        KGraph g = build_kgraph(postfix, k, reduced != 0);
        if (g.size() > cap) return fail("graph larger than output buffers", -2);
        std::memcpy(labels, g.label.data(), g.size() * 4);
        std::memcpy(next_a, g.next_a.data(), g.size() * 4);
        std::memcpy(next_b, g.next_b.data(), g.size() * 4);
        return g.size();
    } catch (const std::exception& e) { return fail(e.what()); }
}

int txh_kgraph_fused(const char* postfix, unsigned k, int32_t* labels, int32_t* next_a, int32_t* next_b, int32_t cap,
                     char* members, size_t members_cap) {
    try {
        KGraph g = build_kgraph(postfix, k, false, false, true);
        if (g.size() > cap) return fail("graph larger than output buffers", -2);
        std::memcpy(labels, g.label.data(), g.size() * 4);
        std::memcpy(next_a, g.next_a.data(), g.size() * 4);
        std::memcpy(next_b, g.next_b.data(), g.size() * 4);
        std::string lines;
        for (int32_t v = 0; v < g.size(); ++v) {
            if (g.takes_residue(v)) g.for_each_residue(v, [&](unsigned char c) { lines.push_back((char)c); });
            lines.push_back('\n');
        }
        const int r = put(lines, members, members_cap);
        return r < 0 ? r : g.size();
    } catch (const std::exception& e) { return fail(e.what()); }
}

int txh_kgraph_dot(const char* postfix, unsigned k, int reduced, int augment, char* out, size_t cap) {
    try {
        KGraph g = build_kgraph(postfix, k, reduced != 0);
        if (augment) g.augment();
        return put(g.to_graphviz(), out, cap);
    } catch (const std::exception& e) { return fail(e.what()); }
}

int txh_compile_batch(const char* const* regex, size_t n, int dna, unsigned k, unsigned reduction, uint64_t bins,
                      txh_blob** out, int* status) {
    try {
        KmerEncoder enc = encoder(dna, k, reduction);
        ProgramBatch batch(enc);
        auto blob = std::make_unique<txh_blob>();
        int failures = 0;
        for (size_t i = 0; i < n; ++i) {
            int st = 0;
            try {
                if (bins <= 1) batch.add_passthrough();
                else {
                    const std::string postfix = preprocess_query(regex[i], enc);
                    batch.add(build_kgraph(postfix, k, enc.alphabet() != Alphabet::Base, false, true));
                }
            } catch (const std::exception& e) {
                g_err = std::string("query ") + std::to_string(i) + ": " + e.what();
                st = -1;
                ++failures;
                // keep program indexes aligned with query indexes: an empty program (zero mask)
                batch.add_empty();
            }
            if (status) status[i] = st;
        }
        blob->bytes = batch.serialise();
        for (size_t i = 0; i < batch.size(); ++i) {
            const QueryProgram& p = batch.program(i);
            blob->stats.insert(blob->stats.end(), {(uint64_t)p.ops.size(), (uint64_t)p.n_slots, p.states, p.probes});
        }
        *out = blob.release();
        return failures;
    } catch (const std::exception& e) { return fail(e.what()); }
}

namespace {
struct CallbackExecutor final : StageExecutor {
    txh_stage_fn fn;
    void* user;
    void stage(const uint8_t* blob, size_t blob_bytes, const std::vector<uint32_t>& qp, const std::vector<uint32_t>& qs,
               std::vector<uint8_t>& alive) override {
        alive.assign(qp.size(), 1);
        if (fn(user, blob, blob_bytes, qp.data(), qs.data(), qp.size(), alive.data()) != 0)
            throw std::runtime_error("stage executor callback failed");
    }
};
}  // namespace

int txh_run_staged(const char* const* regex, size_t n, int dna, unsigned k, unsigned reduction, uint64_t bins,
                   size_t ops_per_query_per_stage, size_t ops_per_stage, const txh_gap_options* gaps, txh_stage_fn fn,
                   void* user, int* status, uint64_t* stats6) {
    return txh_run_staged_dense(regex, n, dna, k, reduction, bins, ops_per_query_per_stage, ops_per_stage, gaps, nullptr, fn, user,
                                status, stats6);
}

int txh_run_staged_dense(const char* const* regex, size_t n, int dna, unsigned k, unsigned reduction, uint64_t bins,
                         size_t ops_per_query_per_stage, size_t ops_per_stage, const txh_gap_options* gaps,
                         const txh_dense_options* dense, txh_stage_fn fn, void* user, int* status, uint64_t* stats6) {
    try {
        KmerEncoder enc = encoder(dna, k, reduction);
        std::vector<std::string> rx(regex, regex + n);
        CallbackExecutor exec;
        exec.fn = fn;
        exec.user = user;
        StagedOptions opt;
        // an explicit per-query budget is taken literally (no adaptive growth, no waiting for verified states)
        if (ops_per_query_per_stage) { opt.ops_per_query_per_stage = ops_per_query_per_stage; opt.stage_target_ops = 0; opt.verified_levels = false; }
        if (ops_per_stage) opt.ops_per_stage = ops_per_stage;
        // this entry point drives test doubles that run ops in an interpreter: keep a runaway query finite
        opt.limits.max_ops = (size_t)64 << 20;
        if (gaps) opt.gaps = GapOptions{gaps->augment != 0, gaps->dgram_loaded != 0, gaps->min_gap, gaps->max_gap};
        if (dense && dense->enabled) {
            opt.dense.enabled = true;
            if (dense->min_states) opt.dense.min_states = dense->min_states;
            if (dense->sparse_below) opt.dense.sparse_below = dense->sparse_below - 1;  // 1 = never enumerate again
            if (dense->max_blocks) opt.dense.max_blocks = dense->max_blocks;
            opt.dense.slot_bytes = dense->slot_bytes;
            if (dense->pool_bytes) opt.dense_pool_bytes = dense->pool_bytes;
            opt.dense.tracked_ok = dense->tracked != 0;     // the executor keeps live lists (tracked programs, include/txq_program.h)
            opt.dense.tracked_force = dense->tracked > 1 ? 1 : 0;  // 2: every query, whatever the run learns about the index
        }
        std::vector<int> st;
        std::vector<std::string> why;
        const StagedStats s = run_staged(enc, bins, rx, exec, opt, &st, &why);
        int failures = 0;
        for (size_t i = 0; i < n; ++i) {
            if (status) status[i] = st[i];
            if (st[i]) { ++failures; g_err = "query " + std::to_string(i) + ": " + why[i]; }
        }
        if (stats6) {
            stats6[0] = s.stages; stats6[1] = s.ops; stats6[2] = s.kmers; stats6[3] = s.states; stats6[4] = s.pruned; stats6[5] = s.feedback_queries;
            stats6[6] = (uint64_t)(s.expand_seconds * 1e6); stats6[7] = (uint64_t)(s.execute_seconds * 1e6);
        }
        return failures;
    } catch (const std::exception& e) { return fail(e.what()); }
}

int txh_join_shard_masks(size_t n, uint64_t mask_words, size_t n_shards, const uint64_t* word0, const uint64_t* words,
                         const uint64_t* const* shard_masks, uint64_t* out) {
    try {
        const std::vector<uint64_t> full = join_shard_masks(n, mask_words, std::vector<uint64_t>(word0, word0 + n_shards),
                                                            std::vector<uint64_t>(words, words + n_shards),
                                                            std::vector<const uint64_t*>(shard_masks, shard_masks + n_shards));
        std::memcpy(out, full.data(), full.size() * 8);
        return 0;
    } catch (const std::exception& e) { return fail(e.what()); }
}

int64_t txh_dgram_values(const char* seq, size_t len, uint64_t min_gap, uint64_t max_gap, uint64_t* out, size_t cap) {
    std::vector<uint64_t> v;
    dgram_record_values(std::string_view(seq, len), min_gap, max_gap, v);
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return (int64_t)v.size();
}

const void* txh_blob_data(const txh_blob* b, size_t* bytes) {
    if (bytes) *bytes = b->bytes.size();
    return b->bytes.data();
}
int txh_blob_stats(const txh_blob* b, uint64_t* stats4, size_t n) {
    if (n * 4 != b->stats.size()) return fail("stats size mismatch");
    std::memcpy(stats4, b->stats.data(), b->stats.size() * 8);
    return 0;
}
void txh_blob_free(txh_blob* b) { delete b; }

int64_t txh_regex_find_all(const char* pattern, int posix, const char* text, size_t len, uint64_t* out, size_t cap) {
    try {
        const Matcher m(pattern, posix ? Matcher::Semantics::LeftmostLongest : Matcher::Semantics::LeftmostFirst);
        Matcher::Cache cache;
        size_t n = 0;
        m.find_all(std::string_view(text, len), cache, [&](size_t s, size_t l) {
            if (2 * n + 1 < cap) { out[2 * n] = s; out[2 * n + 1] = l; }
            ++n;
        });
        return (int64_t)n;
    } catch (const std::exception& e) { return fail(e.what()); }
}

int64_t txh_regex_required_literal(const char* pattern, int posix, char* out, size_t cap) {
    try {
        const Matcher m(pattern, posix ? Matcher::Semantics::LeftmostLongest : Matcher::Semantics::LeftmostFirst);
        const std::string& lit = m.required_literal();
        for (size_t i = 0; i < lit.size() && i < cap; ++i) out[i] = lit[i];
        return (int64_t)lit.size();
    } catch (const std::exception& e) { return fail(e.what()); }
}

int64_t txh_record_values(int dna, unsigned k, unsigned reduction, const char* seq, size_t len, int wraparound,
                          uint64_t* out, size_t cap) {
    KmerEncoder enc = encoder(dna, k, reduction);
    std::vector<uint64_t> v;
    enc.record_values(std::string_view(seq, len), wraparound != 0, v);
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return (int64_t)v.size();
}

int txh_index_parse(const void* bytes, size_t n, txh_index** out) {
    try {
        std::vector<uint8_t> v((const uint8_t*)bytes, (const uint8_t*)bytes + n);
        auto ix = std::make_unique<txh_index>();
        ix->image = parse_index(v);
        *out = ix.release();
        return 0;
    } catch (const std::exception& e) { return fail(e.what()); }
}

int txh_index_load(const char* path, txh_index** out) {
    try {
        auto ix = std::make_unique<txh_index>();
        ix->image = read_index_file(path);
        *out = ix.release();
        return 0;
    } catch (const std::exception& e) { return fail(e.what()); }
}

int txh_index_from_ibf(unsigned k, int dna, unsigned reduction, unsigned hash_count, uint64_t bins, uint64_t bin_size,
                       const uint64_t* words, const char* paths, txh_index** out) {
    try {
        auto ix = std::make_unique<txh_index>();
        IndexImage& im = ix->image;
        im.k = (uint8_t)k;
        im.molecule = dna ? "na" : "aa";
        im.reduction = (uint8_t)reduction;
        im.hash_count = (uint8_t)hash_count;
        im.ibf.shape(bins, bin_size, hash_count);
        std::memcpy(im.ibf.words.data(), words, im.ibf.words.size() * 8);
        std::string all = paths ? paths : "";
        size_t at = 0;
        while (at <= all.size() && im.bin_paths.size() < bins) {
            const size_t nl = all.find('\n', at);
            im.bin_paths.push_back(all.substr(at, nl == std::string::npos ? std::string::npos : nl - at));
            if (nl == std::string::npos) break;
            at = nl + 1;
        }
        if (im.bin_paths.size() != bins) return fail("need exactly one path per bin");
        if (!im.ibf.consistent()) return fail("inconsistent IBF shape");
        *out = ix.release();
        return 0;
    } catch (const std::exception& e) { return fail(e.what()); }
}

static std::string json_str(const std::string& s) {
    std::string o = "\"";
    for (char c : s) {
        if (c == '"' || c == '\\') { o += '\\'; o += c; }
        else if ((unsigned char)c < 0x20) o += ' ';
        else o += c;
    }
    return o + "\"";
}

int txh_index_describe(const txh_index* ix, char* json, size_t cap) {
    const IndexImage& im = ix->image;
    std::string j = "{\"k\":" + std::to_string(im.k) + ",\"molecule\":" + json_str(im.molecule) + ",\"is_hibf\":" + (im.is_hibf ? "true" : "false") +
                    ",\"reduction\":" + std::to_string(im.reduction) + ",\"hash_count\":" + std::to_string(im.hash_count) +
                    ",\"bins\":" + std::to_string(im.bin_count()) + ",\"format\":" + json_str(im.format) + ",\"ibfs\":[";
    auto one = [](const IbfImage& f) {
        return "{\"bins\":" + std::to_string(f.bins) + ",\"tech_bins\":" + std::to_string(f.tech_bins) + ",\"bin_size\":" + std::to_string(f.bin_size) +
               ",\"hash_shift\":" + std::to_string(f.hash_shift) + ",\"bin_words\":" + std::to_string(f.bin_words) + ",\"hash_funs\":" + std::to_string(f.hash_funs) + "}";
    };
    if (im.is_hibf) for (size_t i = 0; i < im.hibf.ibfs.size(); ++i) j += (i ? "," : "") + one(im.hibf.ibfs[i]);
    else j += one(im.ibf);
    j += "],\"paths\":[";
    for (size_t i = 0; i < im.bin_paths.size(); ++i) j += (i ? "," : "") + json_str(im.bin_paths[i]);
    j += "]}";
    return put(j, json, cap);
}

static const IbfImage* pick(const txh_index* ix, uint64_t id) {
    const IndexImage& im = ix->image;
    if (!im.is_hibf) return id == 0 ? &im.ibf : nullptr;
    return id < im.hibf.ibfs.size() ? &im.hibf.ibfs[id] : nullptr;
}

int64_t txh_index_words(const txh_index* ix, uint64_t ibf_id, uint64_t* out, size_t cap) {
    const IbfImage* f = pick(ix, ibf_id);
    if (!f) return fail("IBF id out of range");
    if (out && cap >= f->word_count()) std::memcpy(out, f->word_data(), f->word_count() * 8);
    return (int64_t)f->word_count();
}

int64_t txh_index_maps(const txh_index* ix, uint64_t ibf_id, uint64_t* next_ibf_id, uint64_t* tb_to_user, size_t cap) {
    const IndexImage& im = ix->image;
    if (!im.is_hibf || ibf_id >= im.hibf.ibfs.size()) return fail("not an HIBF / IBF id out of range");
    const auto& nx = im.hibf.next_ibf_id[ibf_id];
    const auto& tb = im.hibf.tb_to_user_bin[ibf_id];
    if (cap >= nx.size()) {
        std::memcpy(next_ibf_id, nx.data(), nx.size() * 8);
        std::memcpy(tb_to_user, tb.data(), tb.size() * 8);
    }
    return (int64_t)nx.size();
}

const void* txh_index_serialise(txh_index* ix, size_t* bytes) {
    try {
        ix->file = serialise_index(ix->image);
        if (bytes) *bytes = ix->file.size();
        return ix->file.data();
    } catch (const std::exception& e) { fail(e.what()); return nullptr; }
}

void txh_index_free(txh_index* ix) { delete ix; }

}  // extern "C"

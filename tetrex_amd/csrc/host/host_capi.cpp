// C entry points of the host front-end (include/txh.h).
#include "../../../include/txh.h"
#include "compiler.hpp"
#include "encoder.hpp"
#include "kgraph.hpp"
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
                    batch.add(build_kgraph(postfix, k, enc.alphabet() != Alphabet::Base));
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

int64_t txh_record_values(int dna, unsigned k, unsigned reduction, const char* seq, size_t len, int wraparound,
                          uint64_t* out, size_t cap) {
    KmerEncoder enc = encoder(dna, k, reduction);
    std::vector<uint64_t> v;
    enc.record_values(std::string_view(seq, len), wraparound != 0, v);
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return (int64_t)v.size();
}

}  // extern "C"

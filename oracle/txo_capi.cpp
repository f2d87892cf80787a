// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see oracle/txo_ibf.hpp header).
// C entry points of the CPU oracle (liboracle.so), loaded with ctypes by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
#include "txo_collector.hpp"
#include <cmath>
#include <cstring>
#include <thread>
#include <memory>

namespace txo {
uint64_t compute_bitcount(uint64_t n, float fpr) {
    // include/index_ibf.h:133-139: std::log(float) is evaluated in float, then promoted.
    double num = -static_cast<double>(n) * std::log(fpr);
    double den = std::pow(std::log(2), 2);
    return static_cast<uint64_t>(std::ceil(num / den));
}
}  // namespace txo

using namespace txo;

struct txo_index {
    bool is_hibf = false;
    Ibf ibf;
    Hibf hibf;
    Encoder enc;
    IndexView view;
    void refresh_view() {
        view.enc = enc;
        if (is_hibf) {
            view.bins = hibf.user_bins;
            view.probe = [this](uint64_t v, uint64_t* out) { hibf.query(v, out); };
        } else {
            view.bins = ibf.bins;
            view.probe = [this](uint64_t v, uint64_t* out) { ibf.bulk_contains(v, out); };
        }
    }
};

static thread_local std::string g_err;
static int fail(const std::exception& e) { g_err = e.what(); return -1; }
static int copy_out(const std::string& s, char* out, size_t cap) {
    if (s.size() + 1 > cap) { g_err = "output buffer too small"; return -2; }
    std::memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}

extern "C" {

const char* txo_last_error() { return g_err.c_str(); }

uint64_t txo_compute_bitcount(uint64_t n, float fpr) { return compute_bitcount(n, fpr); }

// rows hit by `value` for hash functions 0..h-1 of an IBF with `bin_size` rows
int txo_hash_rows(uint64_t value, uint64_t bin_size, unsigned h, uint64_t* rows) {
    try {
        Ibf f(64, 1, 1);
        f.bin_size = bin_size; f.hash_shift = clz64(bin_size);
        if (h > 5) throw std::invalid_argument("h > 5");
        for (unsigned i = 0; i < h; ++i) rows[i] = f.row_of(value, i);
        return 0;
    } catch (const std::exception& e) { return fail(e); }
}

txo_index* txo_ibf_new(uint64_t bins, uint64_t bin_size, unsigned h, int dna, unsigned k, unsigned reduction) {
    try {
        auto ix = std::make_unique<txo_index>();
        ix->ibf.init(bins, bin_size, h);
        ix->enc = Encoder(dna != 0, (uint8_t)k, (uint8_t)reduction);
        ix->refresh_view();
        return ix.release();
    } catch (const std::exception& e) { fail(e); return nullptr; }
}

// adopt a row-major [bin_size][bin_words] matrix (copied)
int txo_ibf_set_words(txo_index* ix, const uint64_t* words, uint64_t n_words) {
    if (!ix || ix->is_hibf || n_words != ix->ibf.data.size()) { g_err = "word count mismatch"; return -1; }
    std::memcpy(ix->ibf.data.data(), words, n_words * 8);
    return 0;
}
const uint64_t* txo_ibf_words(const txo_index* ix, uint64_t* n_words) {
    if (n_words) *n_words = ix->ibf.data.size();
    return ix->ibf.data.data();
}
void txo_ibf_shape(const txo_index* ix, uint64_t* out6) {
    const Ibf& f = ix->ibf;
    out6[0] = f.bins; out6[1] = f.tech_bins; out6[2] = f.bin_size; out6[3] = f.hash_shift; out6[4] = f.bin_words; out6[5] = f.hash_funs;
}
int txo_ibf_emplace(txo_index* ix, const uint64_t* values, uint64_t n, uint64_t bin) {
    if (!ix || ix->is_hibf || bin >= ix->ibf.bins) { g_err = "bad emplace"; return -1; }
    for (uint64_t i = 0; i < n; ++i) ix->ibf.emplace(values[i], bin);
    return 0;
}
// values[i] goes to bins_of[i]
int txo_ibf_emplace_pairs(txo_index* ix, const uint64_t* values, const uint32_t* bins_of, uint64_t n) {
    if (!ix || ix->is_hibf) { g_err = "bad emplace"; return -1; }
    for (uint64_t i = 0; i < n; ++i) {
        if (bins_of[i] >= ix->ibf.bins) { g_err = "bin out of range"; return -1; }
        ix->ibf.emplace(values[i], bins_of[i]);
    }
    return 0;
}

// bulk_contains over a batch: out is n x bin_words (IBF) / n x ceil(user_bins/64) (HIBF).
// threads <= 1 is the faithful single-threaded reference behaviour.
int txo_probe(const txo_index* ix, const uint64_t* values, uint64_t n, uint64_t* out, int threads) {
    try {
        const uint64_t W = ix->view.words();
        auto body = [&](uint64_t lo, uint64_t hi) {
            for (uint64_t i = lo; i < hi; ++i) ix->view.probe(values[i], out + i * W);
        };
        if (threads <= 1) { body(0, n); return 0; }
        std::vector<std::thread> pool;
        uint64_t step = (n + threads - 1) / threads;
        for (int t = 0; t < threads; ++t) {
            uint64_t lo = std::min<uint64_t>(n, t * step), hi = std::min<uint64_t>(n, lo + step);
            if (lo < hi) pool.emplace_back(body, lo, hi);
        }
        for (auto& t : pool) t.join();
        return 0;
    } catch (const std::exception& e) { return fail(e); }
}

// ---- HIBF ------------------------------------------------------------------------
txo_index* txo_hibf_new(uint64_t user_bins, int dna, unsigned k, unsigned reduction) {
    auto ix = std::make_unique<txo_index>();
    ix->is_hibf = true;
    ix->hibf.user_bins = user_bins;
    ix->enc = Encoder(dna != 0, (uint8_t)k, (uint8_t)reduction);
    ix->refresh_view();
    return ix.release();
}
// append one IBF; next_ibf_id / tb_to_user have `bins` entries (merged = ~0 in tb_to_user)
int txo_hibf_add_ibf(txo_index* ix, uint64_t bins, uint64_t bin_size, unsigned h, const uint64_t* words,
                     const uint64_t* next_ibf_id, const uint64_t* tb_to_user) {
    try {
        Ibf f(bins, bin_size, h);
        if (words) std::memcpy(f.data.data(), words, f.data.size() * 8);
        ix->hibf.ibf.push_back(std::move(f));
        ix->hibf.next_ibf_id.emplace_back(next_ibf_id, next_ibf_id + bins);
        ix->hibf.tb_to_user.emplace_back(tb_to_user, tb_to_user + bins);
        return (int)ix->hibf.ibf.size() - 1;
    } catch (const std::exception& e) { return fail(e); }
}
int txo_hibf_emplace(txo_index* ix, uint64_t ibf_id, const uint64_t* values, uint64_t n, uint64_t tb) {
    if (!ix->is_hibf || ibf_id >= ix->hibf.ibf.size() || tb >= ix->hibf.ibf[ibf_id].bins) { g_err = "bad hibf emplace"; return -1; }
    for (uint64_t i = 0; i < n; ++i) ix->hibf.ibf[ibf_id].emplace(values[i], tb);
    return 0;
}
const uint64_t* txo_hibf_words(const txo_index* ix, uint64_t ibf_id, uint64_t* n_words) {
    if (n_words) *n_words = ix->hibf.ibf[ibf_id].data.size();
    return ix->hibf.ibf[ibf_id].data.data();
}
void txo_index_free(txo_index* ix) { delete ix; }
uint64_t txo_index_bins(const txo_index* ix) { return ix->view.bins; }

// ---- encoders --------------------------------------------------------------------
// returns the number of values (also when cap is too small, nothing is written past cap)
int64_t txo_decompose(int dna, unsigned k, unsigned reduction, const char* seq, uint64_t len, int quirk,
                      uint64_t* out, uint64_t cap) {
    Encoder e(dna != 0, (uint8_t)k, (uint8_t)reduction);
    auto v = e.decompose_record(std::string_view(seq, len), quirk != 0);
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return (int64_t)v.size();
}
// rolls every symbol of `seq` into *fwd; canon[i] receives the value update_kmer returns
void txo_update_kmers(int dna, unsigned k, unsigned reduction, const char* seq, uint64_t len,
                      uint64_t* fwd_io, uint64_t* fwd_out, uint64_t* canon_out) {
    Encoder e(dna != 0, (uint8_t)k, (uint8_t)reduction);
    uint64_t f = *fwd_io;
    for (uint64_t i = 0; i < len; ++i) {
        uint64_t c = e.update_kmer((unsigned char)seq[i], f);
        if (fwd_out) fwd_out[i] = f;
        if (canon_out) canon_out[i] = c;
    }
    *fwd_io = f;
}
void txo_encoder_tables(unsigned reduction, uint8_t* aamap256, char* redmap256) {
    Encoder e(false, 4, (uint8_t)reduction);
    std::memcpy(aamap256, e.aamap.data(), 256);
    std::memcpy(redmap256, e.redmap.data(), 256);
}

// ---- regex front-end ---------------------------------------------------------------
int txo_translate(const char* rx, char* out, size_t cap) { return copy_out(translate(rx), out, cap); }
int txo_trim_regex(const char* rx, char* out, size_t cap) { return copy_out(trim_regex(rx), out, cap); }
int txo_reduce_alphabet(const char* rx, unsigned reduction, char* out, size_t cap) {
    Encoder e(false, 4, (uint8_t)reduction);
    return copy_out(reduce_alphabet(rx, e.redmap), out, cap);
}

// ---- k-graph -----------------------------------------------------------------------
// labels[n], succ[2n] (first, second; -1 = none), ranks[n]; arcs[2*n_arcs] (source, target).
// Returns node count, or <0.  *n_arcs receives the arc count.
int txo_kgraph(const char* postfix, unsigned k, int reduced, int* labels, int* succ, int* ranks, int cap_nodes,
               int* arcs, int cap_arcs, int* n_arcs) {
    try {
        KGraphBuilder kb((uint8_t)k, reduced != 0);
        kb.build(postfix);
        int n = kb.g.node_count(), m = (int)kb.g.arc.size();
        if (n_arcs) *n_arcs = m;
        if (n > cap_nodes || m > cap_arcs) { g_err = "graph larger than output buffers"; return -2; }
        auto r = kb.g.ranks();
        for (int i = 0; i < n; ++i) {
            labels[i] = kb.g.label[i];
            succ[2 * i] = kb.g.succ[i].first; succ[2 * i + 1] = kb.g.succ[i].second;
            ranks[i] = r[i];
        }
        for (int a = 0; a < m; ++a) { arcs[2 * a] = kb.g.arc[a].first; arcs[2 * a + 1] = kb.g.arc[a].second; }
        return n;
    } catch (const std::exception& e) { return fail(e); }
}

// ---- d-gram index ("tetrex track") ---------------------------------------------------
// codes one sequence record contributes (DGramIndex::process_sequence); returns the count
int64_t txo_dgram_codes(const char* seq, uint64_t len, uint64_t min_gap, uint64_t max_gap, uint64_t* out, uint64_t cap) {
    std::vector<uint64_t> v;
    dgram_codes(std::string_view(seq, len), min_gap, max_gap, v);
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return (int64_t)v.size();
}

// query with -a (augment) and optionally -g (dgram: a flat-IBF txo_index over d-gram codes built
// for gaps min_gap..max_gap).  stats5: probes, states, quirk_merges, dgram_probes, gap_nodes.
int txo_query_aug(txo_index* ix, const char* regex, int augment, txo_index* dgram, uint64_t min_gap, uint64_t max_gap,
                  uint64_t* mask, uint64_t* stats5) {
    try {
        DGramView dv;
        if (dgram) {
            if (dgram->is_hibf || dgram->ibf.bins != ix->view.bins) throw std::invalid_argument("d-gram index must be a flat IBF over the same bins");
            dv.loaded = true; dv.min_gap = min_gap; dv.max_gap = max_gap;
            dv.probe = [dgram](uint64_t v, uint64_t* out) { dgram->ibf.bulk_contains(v, out); };
        }
        QueryResult q = run_query(ix->view, regex, augment != 0, &dv);
        std::memcpy(mask, q.mask.data(), q.mask.size() * 8);
        if (stats5) {
            stats5[0] = q.stats.probes; stats5[1] = q.stats.states; stats5[2] = q.stats.quirk_merges;
            stats5[3] = q.stats.dgram_probes; stats5[4] = q.stats.gap_nodes;
        }
        return 0;
    } catch (const std::exception& e) { return fail(e); }
}

// ---- whole query -------------------------------------------------------------------
// mask: ceil(bins/64) words.  stats3: probes, states, quirk_merges.
int txo_query(txo_index* ix, const char* regex, uint64_t* mask, uint64_t* stats3) {
    try {
        QueryResult q = run_query(ix->view, regex);
        std::memcpy(mask, q.mask.data(), q.mask.size() * 8);
        if (stats3) { stats3[0] = q.stats.probes; stats3[1] = q.stats.states; stats3[2] = q.stats.quirk_merges; }
        return 0;
    } catch (const std::exception& e) { return fail(e); }
}

// The same with states keyed by what they are (Collector::well_defined) — NOT the reference's merge rule; see txo_collector.hpp.
// For the queries where the reference's own result is implementation-defined (quirk_merges > 0).
int txo_query_well_defined(txo_index* ix, const char* regex, int augment, uint64_t* mask, uint64_t* stats3) {
    try {
        QueryResult q = run_query(ix->view, regex, augment != 0, nullptr, true);
        std::memcpy(mask, q.mask.data(), q.mask.size() * 8);
        if (stats3) { stats3[0] = q.stats.probes; stats3[1] = q.stats.states; stats3[2] = q.stats.quirk_merges; }
        return 0;
    } catch (const std::exception& e) { return fail(e); }
}

}  // extern "C"

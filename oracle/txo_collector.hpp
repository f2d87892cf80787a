// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see oracle/txo_ibf.hpp header).
// CPU oracle: the on-the-fly frontier collector, restating
//   include/otf_collector.h:162-214  extract_hash / create_selection_bitmask / push / absorb / scrub
//   include/otf_collector.h:247-288  update_path (+kmer_cache_), split_procedure
//   include/otf_collector.h:341-393  collect()
//   include/query.h:226-247          process_query (preprocess -> k-graph -> collector -> collect)
//   include/query.h:257-272          run_collection: hit_vector(all ones) &= collect(); 1-bin index skips filtering
//   src/query.cpp:40-75              compute_set_bins
// States of one rank are kept in a robin_hood map in the reference and popped via begin(),
// i.e. in an implementation-defined order.  The result mask does not depend on that order
// except when two states with equal (k-1)-suffix bits but different shift_count_ meet
// (first arrival wins, otf_collector.h:190-208; SURVEY.md §7 "state-merge quirk").  This
// restatement pops in insertion order and reports when such a collision happened.
#pragma once
#include "txo_ibf.hpp"
#include "txo_encode.hpp"
#include "txo_regex.hpp"
#include "txo_kgraph.hpp"
#include <functional>
#include <array>
#include <map>
#include <unordered_map>

namespace txo {

using Mask = std::vector<uint64_t>;

inline bool mask_none(const Mask& m) { for (uint64_t w : m) if (w) return false; return true; }

struct IndexView {
    uint64_t bins = 0;
    Encoder enc;
    std::function<void(uint64_t, uint64_t*)> probe;  // value -> bins-bit mask (W words)
    uint64_t words() const { return (bins + 63) / 64; }
};

// DGramIndex as the collector sees it (include/dGramIndex.h:63-103,213-283): a second IBF over
// gapped 3+3 residue codes and the gap range it was built for.  A default-constructed DGramIndex
// (no -g) has min_gap = max_gap = 0 and is never queried (has_dibf_ is false).
struct DGramView {
    bool loaded = false;
    uint64_t min_gap = 0, max_gap = 0;
    std::function<void(uint64_t, uint64_t*)> probe;
};

// DGramTools::aa_to_num (include/dGramIndex.h:24-60): Base residue code for 'A'..'Z', else 0.
inline uint64_t aa_to_num(int c) {
    static const uint8_t base[26] = {0, 2, 1, 2, 3, 4, 5, 6, 7, 9, 8, 9, 10, 11, 20, 12, 13, 14, 15, 16, 20, 17, 18, 20, 19, 3};
    return (c >= 'A' && c <= 'Z') ? base[c - 'A'] : 0;
}

// DGramIndex::process_sequence (include/dGramIndex.h:159-211): the codes one record contributes.
inline void dgram_codes(std::string_view seq, uint64_t min_gap, uint64_t max_gap, std::vector<uint64_t>& out) {
    static const std::string alphabet = "ACDEFGHIKLMNPQRSTVWY";
    auto idx = [&](char c) -> int { size_t p = alphabet.find(c); return p == std::string::npos ? -1 : (int)p; };
    if (seq.size() < min_gap + 7) return;
    for (size_t i = 2; i + min_gap + 3 < seq.size(); ++i) {
        const int a1 = idx(seq[i - 2]), a2 = idx(seq[i - 1]), a3 = idx(seq[i]);
        if (a1 < 0 || a2 < 0 || a3 < 0) continue;
        for (uint64_t gap = min_gap; gap <= max_gap; ++gap) {
            const size_t j = i + gap + 1;
            if (j + 2 >= seq.size()) break;
            const int b1 = idx(seq[j]), b2 = idx(seq[j + 1]), b3 = idx(seq[j + 2]);
            if (b1 < 0 || b2 < 0 || b3 < 0) continue;
            out.push_back(gap * 64000000ULL + a1 * 3200000ULL + a2 * 160000ULL + a3 * 8000ULL + b1 * 400ULL + b2 * 20ULL + b3);
        }
    }
}

struct CollectStats {
    uint64_t probes = 0;          // index look-ups actually issued (distinct forward k-mers)
    uint64_t states = 0;          // states popped
    uint64_t quirk_merges = 0;    // merges of states with different shift_count_ / gapped flag
    uint64_t dgram_probes = 0;    // d-gram index look-ups
    uint64_t gap_nodes = 0;       // Gap nodes added by augment()
};

struct Collector {
    struct Item { int node; uint8_t shift; uint64_t kmer; Mask path; bool gapped = false; int res1 = 0, res2 = 0; };
    struct Bucket { std::vector<Item> items; std::unordered_map<uint64_t, size_t> at; std::map<std::array<uint64_t, 3>, size_t> at_strict; size_t head = 0; };

    Graph g;  // own copy: augment() adds Gap / guard nodes
    const IndexView& ix;
    DGramView dgram;                 // default: not loaded
    std::vector<uint64_t> gap_of;    // gap_map_: gap length of a Gap node
    std::vector<int> rank;
    std::vector<Bucket> table;
    uint64_t submask = 0;
    std::unordered_map<uint64_t, Mask> cache;  // kmer_cache_, keyed by FORWARD k-mer
    CollectStats stats;
    // NOT the reference's behaviour.  The reference merges two states whenever the bits of their (k-1)-symbol suffix are
    // equal, also when one of them has seen fewer than k-1 symbols (leading residues that encode to 0) or is collecting a
    // d-gram, and keeps whichever arrived first — an implementation-defined result (robin_hood iteration order; SURVEY.md
    // §7 "state-merge quirk"), counted in stats.quirk_merges.  well_defined = true keys states by what they ARE (suffix +
    // how many symbols they have seen; for d-gram states the partial code + residues seen), so such states never merge:
    // the semantics the product implements.  Tests use it to check the product on exactly those queries where the
    // reference itself has no defined answer; everywhere else (quirk_merges == 0) both modes give the same masks.
    bool well_defined = false;

    Collector(const Graph& graph, const IndexView& index) : g(graph), ix(index) {
        for (unsigned c = ix.enc.k - 1; c > 0; --c) submask = (submask << ix.enc.lshift) | ix.enc.rmask;
        rank = g.ranks();
        table.resize(g.node_count());
        gap_of.assign(g.node_count(), 0);
    }

    // ---- -a / --augment (include/otf_collector.h:395-493) -----------------------------------
    int add_node(int label, uint64_t gap = 0) {
        int n = g.add_node(label);
        gap_of.push_back(gap);
        return n;
    }
    void add_gap(const Catsite& c, uint64_t gap) {
        int n = add_node(kGap, gap);
        g.connect(c.site, n);
        g.connect(n, c.downstream);
    }
    // merge_catsites (:439-464): adjacent catastrophic regions (by topological rank) fuse, their
    // gap sets add pairwise
    void merge_catsites(std::vector<Catsite>& cats) const {
        std::sort(cats.begin(), cats.end(), [&](const Catsite& a, const Catsite& b) { return rank[a.first] < rank[b.first]; });
        std::vector<Catsite> merged;
        bool done = false;
        for (const Catsite& c : cats) {
            if (merged.empty() || rank[c.first] - 1 != rank[merged.back().last]) { merged.push_back(c); continue; }
            Catsite& m = merged.back();
            m.last = c.last;
            std::set<uint64_t> sum;
            for (uint64_t x : m.gaps) for (uint64_t y : c.gaps) sum.insert(x + y);
            m.gaps = sum;
            done = true;
        }
        if (done) cats = merged;
    }
    // augment (:466-493).  The reference iterates the gap set in robin_hood order; a Split keeps only
    // its first and its LAST successor (update_arc_map), so with > 2 gaps the surviving pair is
    // implementation-defined there.  Here: ascending order.
    void augment(std::vector<Catsite> cats) {
        merge_catsites(cats);
        for (Catsite c : cats) {
            const std::set<uint64_t> gaps = c.gaps;
            c.downstream = g.succ[c.last].first;
            if (c.downstream < 0) throw std::out_of_range("catsite without downstream node");
            if (gaps.size() == 1) add_gap(c, *gaps.begin());
            else {
                int sp = add_node(kSplit), gh = add_node(kGhost);
                g.connect(c.site, sp);
                g.connect(gh, c.downstream);
                c.site = sp;
                c.downstream = gh;
                for (uint64_t gp : gaps) add_gap(c, gp);
            }
        }
        table.clear();
        table.resize(g.node_count());
        rank = g.ranks();
    }

    // gap_procedure (:290-312)
    void gap_step(int id, Item& top) {
        const uint64_t gap = gap_of[id];
        int next = g.succ[id].first;
        if (next < 0) throw std::out_of_range("arc_map_.at(): node without successor");
        if (top.shift < 3 || gap < dgram.min_gap || gap > dgram.max_gap) {
            push(Item{next, 0, 0, std::move(top.path), false, 0, 0});
            return;
        }
        const uint64_t a1 = (top.kmer >> 10) & 31, a2 = (top.kmer >> 5) & 31, a3 = top.kmer & 31;
        const uint64_t dg = gap * 64000000ULL + a1 * 3200000ULL + a2 * 160000ULL + a3 * 8000ULL;
        push(Item{next, 0, dg, std::move(top.path), true, 0, 0});
    }
    // update_gapped (:216-245): the three residues after a gap complete the d-gram
    void update_gapped(Item& s, int symbol) {
        if (s.shift == 0) { s.kmer += 400ULL * aa_to_num(symbol); s.res1 = symbol; ++s.shift; }
        else if (s.shift == 1) { s.kmer += 20ULL * aa_to_num(symbol); s.res2 = symbol; ++s.shift; }
        else if (s.shift == 2) {
            const uint64_t dg = s.kmer + aa_to_num(symbol);
            if (dgram.loaded) {
                Mask m(ix.words());
                dgram.probe(dg, m.data());
                ++stats.dgram_probes;
                for (size_t w = 0; w < s.path.size(); ++w) s.path[w] &= m[w];
            }
            s.kmer = 0;
            ix.enc.update_kmer(s.res1, s.kmer);
            ix.enc.update_kmer(s.res2, s.kmer);
            ix.enc.update_kmer(symbol, s.kmer);
            ++s.shift;
            s.gapped = false;
            s.res1 = s.res2 = 0;
        }
    }

    void push(Item&& it) {
        uint64_t key = it.kmer & submask;
        Bucket& b = table[rank[it.node]];
        if (well_defined) {
            const unsigned k = ix.enc.k;
            const std::array<uint64_t, 3> key3 = it.gapped ? std::array<uint64_t, 3>{it.kmer, 1, it.shift}
                                                           : std::array<uint64_t, 3>{key, 0, (uint64_t)(it.shift < k - 1 ? it.shift : k - 1)};
            auto f = b.at_strict.find(key3);
            if (f == b.at_strict.end()) { b.at_strict.emplace(key3, b.items.size()); b.items.push_back(std::move(it)); }
            else {
                Item& dst = b.items[f->second];
                if (dst.shift < it.shift) dst.shift = it.shift;  // k-1 and k symbols seen behave alike from here on
                for (size_t w = 0; w < dst.path.size(); ++w) dst.path[w] |= it.path[w];
            }
            return;
        }
        auto f = b.at.find(key);
        if (f == b.at.end()) { b.at.emplace(key, b.items.size()); b.items.push_back(std::move(it)); }
        else {
            Item& dst = b.items[f->second];
            // (a state behind a Gap node carries its partial d-gram code in kmer: two such states with DIFFERENT codes that agree in
            // the code's low bits meet in one table slot — the first one's d-gram is kept for both, which of the two it is follows
            // from the order of the walk: counted with the other ill-defined merges)
            if (dst.shift != it.shift || dst.gapped != it.gapped || dst.res1 != it.res1 || dst.res2 != it.res2 || (it.gapped && dst.kmer != it.kmer))
                ++stats.quirk_merges;
            for (size_t w = 0; w < dst.path.size(); ++w) dst.path[w] |= it.path[w];
        }
    }

    void update_path(Item& s, int symbol) {
        const unsigned k = ix.enc.k;
        if (s.shift < k - 1) { ix.enc.update_kmer(symbol, s.kmer); ++s.shift; return; }
        if (s.shift == k - 1 || s.shift == k) {
            uint64_t canon = ix.enc.update_kmer(symbol, s.kmer);
            auto f = cache.find(s.kmer);
            if (f == cache.end()) {
                Mask m(ix.words());
                ix.probe(canon, m.data());
                ++stats.probes;
                f = cache.emplace(s.kmer, std::move(m)).first;
            }
            for (size_t w = 0; w < s.path.size(); ++w) s.path[w] &= f->second[w];
            if (s.shift == k - 1) ++s.shift;
        }
    }

    Mask collect() {
        const uint64_t W = ix.words();
        Mask result(W, 0), ones(W, ~0ULL);
        if (ix.bins & 63) ones[W - 1] = (1ULL << (ix.bins & 63)) - 1;
        push(Item{0, 0, 0, ones, false, 0, 0});
        for (int r = 0; r < g.node_count(); ++r) {
            Bucket& b = table[r];
            while (b.head < b.items.size()) {
                Item top = std::move(b.items[b.head]);
                b.at.erase(top.kmer & submask);
                ++b.head;
                ++stats.states;
                int id = top.node, sym = g.label[id];
                auto succ = [&](bool second) {
                    int t = second ? g.succ[id].second : g.succ[id].first;
                    if (t < 0) throw std::out_of_range("arc_map_.at(): node without successor");
                    return t;
                };
                switch (sym) {
                    case kMatch:
                        for (size_t w = 0; w < W; ++w) result[w] |= top.path[w];
                        break;
                    case 36:  // '$'
                    case kGhost:
                        push(Item{succ(false), top.shift, top.kmer, std::move(top.path), top.gapped, top.res1, top.res2});
                        break;
                    case kSplit: {
                        int a = succ(false), c = succ(true);
                        push(Item{a, top.shift, top.kmer, top.path, top.gapped, top.res1, top.res2});
                        push(Item{c, top.shift, top.kmer, std::move(top.path), top.gapped, top.res1, top.res2});
                        break;
                    }
                    case kGap: gap_step(id, top); break;
                    default:
                        if (top.gapped) update_gapped(top, sym);
                        else update_path(top, sym);
                        if (mask_none(top.path)) break;
                        push(Item{succ(false), top.shift, top.kmer, std::move(top.path), top.gapped, top.res1, top.res2});
                        break;
                }
            }
            b.items.clear(); b.at.clear(); b.head = 0;
        }
        return result;
    }
};

struct QueryResult {
    Mask mask;
    std::string preprocessed, postfix;
    CollectStats stats;
    int nodes = 0;
};

// preprocess_query + process_query + the `hit_vector &= ...` of run_collection.
inline QueryResult run_query(const IndexView& ix, const std::string& regex, bool augment = false, const DGramView* dgram = nullptr,
                             bool well_defined = false) {
    QueryResult q;
    std::string rx = regex;
    if (!ix.enc.dna) {
        if (ix.enc.reduction > 0) rx = reduce_alphabet(rx, ix.enc.redmap);
        rx = trim_regex(rx);
    }
    q.preprocessed = rx;
    q.postfix = translate(rx);
    const uint64_t W = ix.words();
    if (ix.bins <= 1) {  // include/query.h:265-272: no filtering for a 1-bin index
        q.mask.assign(W, 0);
        q.mask[0] = 1;
        return q;
    }
    KGraphBuilder kb(ix.enc.k, ix.enc.reduction != kBase);
    kb.build(q.postfix);
    q.nodes = kb.g.node_count();
    Collector c(kb.g, ix);
    c.well_defined = well_defined;
    if (dgram) c.dgram = *dgram;
    if (augment && !kb.cats.empty()) {  // include/query.h:243
        const int before = c.g.node_count();
        c.augment(kb.cats);
        for (int n = before; n < c.g.node_count(); ++n) c.stats.gap_nodes += c.g.label[n] == kGap;
    }
    q.mask = c.collect();
    q.stats = c.stats;
    return q;
}

// compute_set_bins: ascending ids of the set bits; a 1-bin library always yields {0}.
inline std::vector<uint64_t> set_bins(const Mask& m, uint64_t bins) {
    std::vector<uint64_t> out;
    if (bins == 1) { out.push_back(0); return out; }
    for (size_t w = 0; w < m.size(); ++w) {
        uint64_t v = m[w];
        while (v) { out.push_back(w * 64 + (unsigned)__builtin_ctzll(v)); v &= v - 1; }
    }
    return out;
}

}  // namespace txo

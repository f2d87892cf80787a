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

struct CollectStats {
    uint64_t probes = 0;          // index look-ups actually issued (distinct forward k-mers)
    uint64_t states = 0;          // states popped
    uint64_t quirk_merges = 0;    // merges of states with different shift_count_
};

struct Collector {
    struct Item { int node; uint8_t shift; uint64_t kmer; Mask path; };
    struct Bucket { std::vector<Item> items; std::unordered_map<uint64_t, size_t> at; size_t head = 0; };

    const Graph& g;
    const IndexView& ix;
    std::vector<int> rank;
    std::vector<Bucket> table;
    uint64_t submask = 0;
    std::unordered_map<uint64_t, Mask> cache;  // kmer_cache_, keyed by FORWARD k-mer
    CollectStats stats;

    Collector(const Graph& graph, const IndexView& index) : g(graph), ix(index) {
        for (unsigned c = ix.enc.k - 1; c > 0; --c) submask = (submask << ix.enc.lshift) | ix.enc.rmask;
        rank = g.ranks();
        table.resize(g.node_count());
    }

    void push(Item&& it) {
        uint64_t key = it.kmer & submask;
        Bucket& b = table[rank[it.node]];
        auto f = b.at.find(key);
        if (f == b.at.end()) { b.at.emplace(key, b.items.size()); b.items.push_back(std::move(it)); }
        else {
            Item& dst = b.items[f->second];
            if (dst.shift != it.shift) ++stats.quirk_merges;
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
        push(Item{0, 0, 0, ones});
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
                        push(Item{succ(false), top.shift, top.kmer, std::move(top.path)});
                        break;
                    case kSplit: {
                        int a = succ(false), c = succ(true);
                        push(Item{a, top.shift, top.kmer, top.path});
                        push(Item{c, top.shift, top.kmer, std::move(top.path)});
                        break;
                    }
                    case kGap: throw std::runtime_error("Gap nodes (-a/-g) are not part of this oracle yet");
                    default:
                        update_path(top, sym);
                        if (mask_none(top.path)) break;
                        push(Item{succ(false), top.shift, top.kmer, std::move(top.path)});
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
inline QueryResult run_query(const IndexView& ix, const std::string& regex) {
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

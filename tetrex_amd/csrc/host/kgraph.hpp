// Host side: the k-graph (NFA with loops unrolled to depth k-1) of one query — product code.
// Behavioural contract = the reference's builders driven by the postfix text:
//   construct_kgraph          src/construct_nfa.cpp:265-335 (+ procedures :78-262, copy_subgraph :4-76)
//   construct_reduced_kgraph  src/construct_reduced_nfa.cpp:313-383 (+ :79-310)
//   update_arc_map            src/construction_tools.cpp:136-158 (two successor slots per node)
//   parse_quant               src/construction_tools.cpp:4-18
// Node labels: a byte value for a residue node, or one of the markers below
// (include/construction_tools.h:40-46).
#pragma once
#include <cstdint>
#include <set>
#include <string>
#include <utility>
#include <vector>

namespace tetrex {

// A concatenation whose right operand has so many paths that -a/--augment bypasses it with Gap
// nodes (Catsite, reference include/construction_tools.h:147-185; detect_bad_graphs,
// src/construction_tools.cpp:161-180).
struct CatSite {
    int32_t site = -1;        // node before the high-complexity sub-graph
    int32_t first = -1;       // its entry
    int32_t last = -1;        // its exit
    std::set<uint64_t> gaps;  // the sub-graph's possible path lengths
};

struct KGraph {
    static constexpr int32_t kMatch = 256, kGhost = 257, kSplit = 258, kGap = 259;
    static constexpr int32_t kNone = -1;
    // Labels from kClass up: a FUSED residue class (build_kgraph with fuse_classes) — one node standing for a union of
    // single residues (`[LIVM]`, `.`, `(A|G)`), which the reference's builders spell as a tree of Split / Ghost nodes over
    // one node per residue (58 nodes for a wildcard).  The strings the graph accepts are the same; only the expansion
    // (QueryExpansion) reads such graphs — drawing and -a keep the reference's node-for-node graph.
    static constexpr int32_t kClass = 260;
    struct ResidueClass { uint32_t first, count, letters; };  // class_bytes[first, first + count); letters: bit (c - 'A')

    std::vector<int32_t> label;   // per node
    std::vector<int32_t> next_a;  // first successor slot (kNone = no arc yet)
    std::vector<int32_t> next_b;  // second successor slot (== next_a unless the node is a Split)
    std::vector<int32_t> arc_src, arc_dst;
    std::vector<uint64_t> gap;       // gap length of a kGap node (0 elsewhere)
    std::vector<CatSite> catsites;   // filled by build_kgraph
    std::vector<ResidueClass> classes;  // label - kClass indexes this (entries no label points at are leftovers of fusing)
    std::string class_bytes;            // the members of the classes, in the order the query names them, each once

    bool is_class(int32_t v) const { return label[v] >= kClass; }
    // a node that consumes one residue of the text: a residue node or a fused class
    bool takes_residue(int32_t v) const { return label[v] >= kClass || (label[v] < 256 && label[v] != '$'); }
    // the residues a node stands for, in the query's order: fn(byte)
    template <class Fn>
    void for_each_residue(int32_t v, Fn&& fn) const {
        if (label[v] < kClass) { fn((unsigned char)label[v]); return; }
        const ResidueClass& c = classes[(size_t)(label[v] - kClass)];
        for (uint32_t i = 0; i < c.count; ++i) fn((unsigned char)class_bytes[c.first + i]);
    }
    // the same as an array: (bytes, count); `one` receives a residue node's single byte
    std::pair<const unsigned char*, uint32_t> residues(int32_t v, unsigned char* one) const {
        if (label[v] < kClass) { *one = (unsigned char)label[v]; return {one, 1u}; }
        const ResidueClass& c = classes[(size_t)(label[v] - kClass)];
        return {reinterpret_cast<const unsigned char*>(class_bytes.data()) + c.first, c.count};
    }
    uint32_t residue_count(int32_t v) const { return label[v] < kClass ? 1u : classes[(size_t)(label[v] - kClass)].count; }

    int32_t size() const { return (int32_t)label.size(); }
    int32_t add(int32_t lab);
    void link(int32_t from, int32_t to);
    // nodes in an order in which every arc goes forward (node 0, the start ghost, first)
    std::vector<int32_t> topological_order() const;
    // the ranks lemon::topologicalSort gives (reference include/otf_collector.h:328-339): DFS from
    // the highest node id down, newest out-arc first; only augment() depends on this numbering
    std::vector<int32_t> reference_ranks() const;
    // -a/--augment (reference include/otf_collector.h:395-493): bypass every (merged) catsite with
    // Gap nodes, one per possible length; returns the number of Gap nodes added
    size_t augment();
    // -d/--draw: Graphviz text of the graph (reference print_graph, src/construction_tools.cpp:42-94:
    // point for the start node, "Ø" splits, "•" ghosts, doublecircle match, "GAP" gaps, residue letters)
    std::string to_graphviz() const;
};

// Builds the k-graph of `postfix` for k-mer size k.  reduced_alphabet selects the reduced
// builder (symbols are buffered and identical reduced letters of a union collapse).
// Throws std::runtime_error where the reference would run into undefined behaviour
// (stack underflow, empty symbol buffer) instead of reproducing it.
// path_stats: keep the reference's path statistics (Subgraph::paths / lengths) and record the concatenations that -a would
// bypass (catsites) — only KGraph::augment() needs them; a graph built without them cannot be augmented
// fuse_classes: unions of single residues become ONE node each (KGraph::kClass; ignored for reduced alphabets, whose
// builder buffers symbols, and with path_stats, whose catsites are defined on the reference's node-for-node graph)
KGraph build_kgraph(const std::string& postfix, unsigned k, bool reduced_alphabet, bool path_stats = true, bool fuse_classes = false);

}  // namespace tetrex

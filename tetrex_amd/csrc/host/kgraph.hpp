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

    std::vector<int32_t> label;   // per node
    std::vector<int32_t> next_a;  // first successor slot (kNone = no arc yet)
    std::vector<int32_t> next_b;  // second successor slot (== next_a unless the node is a Split)
    std::vector<int32_t> arc_src, arc_dst;
    std::vector<uint64_t> gap;       // gap length of a kGap node (0 elsewhere)
    std::vector<CatSite> catsites;   // filled by build_kgraph

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
KGraph build_kgraph(const std::string& postfix, unsigned k, bool reduced_alphabet, bool path_stats = true);

}  // namespace tetrex

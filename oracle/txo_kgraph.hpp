// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see oracle/txo_ibf.hpp header).
// CPU oracle: k-graph (unrolled NFA) construction, restating
//   src/construct_nfa.cpp:4-335          copy_subgraph, *_procedure, construct_kgraph
//   src/construct_reduced_nfa.cpp:79-383 reduced-alphabet variant (symbol buffer, twin nodes)
//   src/construction_tools.cpp:4-18      parse_quant
//   src/construction_tools.cpp:136-158   update_arc_map (first/second successor slots)
//   include/construction_tools.h:40-46   node labels Match/Ghost/Split/Gap
//   include/otf_collector.h:328-339      determine_top_sort (lemon::topologicalSort ranks)
// The reference builds on lemon::SmartDigraph (ABSENT: lib/lemon is an empty submodule);
// the subset of its behaviour the builder relies on is restated in `Graph`:
// dense ids in creation order, NodeIt/ArcIt iterate in DESCENDING id order, OutArcIt
// visits the most recently added arc first, DFS-based topologicalSort.
#pragma once
#include <cstdint>
#include <set>
#include <string>
#include <vector>
#include <utility>
#include <stdexcept>
#include <cctype>

namespace txo {

enum : int { kMatch = 256, kGhost = 257, kSplit = 258, kGap = 259 };

struct Graph {
    std::vector<int> label;                    // lmap_t
    std::vector<std::pair<int, int>> arc;      // (source, target) by arc id
    std::vector<std::vector<int>> out, in;     // arc ids, oldest first
    std::vector<std::pair<int, int>> succ;     // amap_t: (first, second); -1 = no entry

    int add_node(int lab) {
        label.push_back(lab);
        out.emplace_back(); in.emplace_back();
        succ.emplace_back(-1, -1);
        return (int)label.size() - 1;
    }
    int node_count() const { return (int)label.size(); }

    // update_arc_map: non-split sources overwrite both slots with the newest target,
    // a split fills `first` on its first arc and `second` on every later one.
    int connect(int s, int t) {
        arc.emplace_back(s, t);
        int id = (int)arc.size() - 1;
        out[s].push_back(id); in[t].push_back(id);
        int lab = label[s];
        if (lab < kSplit || lab == kGap) succ[s] = {t, t};
        else if (succ[s].first < 0) succ[s].first = t;
        else succ[s].second = t;
        return id;
    }

    void reach(int from, bool forward, std::vector<char>& seen) const {
        std::vector<int> st{from};
        seen[from] = 1;
        while (!st.empty()) {
            int u = st.back(); st.pop_back();
            for (int a : (forward ? out[u] : in[u])) {
                int v = forward ? arc[a].second : arc[a].first;
                if (!seen[v]) { seen[v] = 1; st.push_back(v); }
            }
        }
    }

    // lemon::topologicalSort as used by determine_top_sort: DFS roots in descending id
    // order, out-arcs newest first, a node leaving the DFS gets index --n.  rank == index
    // (the index-0 node is dropped by the bool-valued write map and keeps rank 0).
    std::vector<int> ranks() const {
        int n = node_count(), next = n;
        std::vector<int> rank(n, 0), it(n, 0);
        std::vector<char> seen(n, 0);
        for (int r = n - 1; r >= 0; --r) {
            if (seen[r]) continue;
            std::vector<int> st{r};
            seen[r] = 1;
            while (!st.empty()) {
                int u = st.back();
                if (it[u] < (int)out[u].size()) {
                    int a = out[u][out[u].size() - 1 - it[u]++];
                    int v = arc[a].second;
                    if (!seen[v]) { seen[v] = 1; st.push_back(v); }
                } else { rank[u] = --next; st.pop_back(); }
            }
        }
        return rank;
    }
};

// Subgraph of include/construction_tools.h:71-142: entry/exit plus the bookkeeping the
// "catastrophic sub-graph" detector needs (number of paths, set of path lengths).
struct Sub {
    int start = -1, end = -1;
    bool twin = false;  // un-materialised single symbol (reduced builder)
    uint64_t paths = 1;
    std::set<uint64_t> lengths;
    static Sub leaf(int n, bool twin) { Sub s; s.start = s.end = n; s.twin = twin; s.lengths = {1}; return s; }
    void meta_from(const Sub& r) { paths = r.paths; lengths = r.lengths; }
};

// Catsite (include/construction_tools.h:147-185): a concatenation whose right operand has so many
// paths that -a/--augment replaces it by Gap nodes.  Ids instead of lemon nodes.
struct Catsite {
    int site = -1;        // cleavage_site_: node before the high-complexity sub-graph
    int first = -1;       // cleavage_start_
    int last = -1;        // cleavage_end_
    int downstream = -1;  // filled by complete()
    std::set<uint64_t> gaps;
};

struct KGraphBuilder {
    Graph g;
    uint8_t k;
    bool reduced;
    std::vector<Sub> st;
    std::vector<int> buf;  // reduced builder's symbol stack (buffer_t)
    std::vector<Catsite> cats;

    KGraphBuilder(uint8_t ksize, bool reduced_alphabet) : k(ksize), reduced(reduced_alphabet) {}

    static std::pair<size_t, size_t> parse_quant(const std::string& p, size_t at) {
        size_t comma = p.find(',', at), end = p.find('}', at);
        if (end == std::string::npos) throw std::runtime_error("quantifier without '}'");
        if (comma == std::string::npos || comma > end) return {(size_t)std::stoi(p.substr(at + 1, end - at)), 0};
        return {(size_t)std::stoi(p.substr(at + 1, comma - at)), (size_t)std::stoi(p.substr(comma + 1, end - comma - 1))};
    }

    Sub pop() {
        if (st.empty()) throw std::runtime_error("k-graph stack underflow");
        Sub s = st.back(); st.pop_back(); return s;
    }
    int buf_top() const {
        if (buf.empty()) throw std::runtime_error("reduced builder: empty symbol buffer (undefined in the reference)");
        return buf.back();
    }
    // twin_procedure: materialise a pending symbol into a node.
    void materialise(Sub& s) {
        if (!s.twin) return;
        int sym = buf_top(); buf.pop_back();
        int n = g.add_node(sym);
        s = Sub::leaf(n, false);
    }
    // In the reduced builder twin_test() is `start == end`, which is also true for a
    // materialised single node (construct_reduced_nfa.cpp:91-94).  `single(s)` restates that.
    bool twin_test(const Sub& s) const { return s.twin || s.start == s.end; }
    void materialise_if_twin(Sub& s) {
        if (!reduced) return;
        if (s.twin) { materialise(s); return; }
        if (s.start == s.end) {  // a real single node re-materialised from the buffer
            int sym = buf_top(); buf.pop_back();
            int n = g.add_node(sym);
            s = Sub::leaf(n, false);
        }
    }

    Sub copy(const Sub& s) {
        if (s.twin || s.start == s.end) {
            int lab = reduced ? buf_top() : g.label[s.start];
            int n = g.add_node(lab);
            Sub c = Sub::leaf(n, false);
            c.meta_from(s);
            return c;
        }
        int n0 = g.node_count(), a0 = (int)g.arc.size();
        std::vector<char> f(n0, 0), b(n0, 0);
        g.reach(s.start, true, f);
        g.reach(s.end, false, b);
        std::vector<int> twin_of(n0, -1);
        for (int v = n0 - 1; v >= 0; --v)  // NodeIt order
            if (f[v] && b[v]) twin_of[v] = g.add_node(g.label[v]);
        for (int a = a0 - 1; a >= 0; --a) {  // ArcIt order
            auto [u, v] = g.arc[a];
            if (twin_of[u] >= 0 && twin_of[v] >= 0) g.connect(twin_of[u], twin_of[v]);
        }
        Sub c;
        c.start = twin_of[s.start]; c.end = twin_of[s.end];
        c.meta_from(s);
        return c;
    }

    void op_symbol(int sym) {
        if (reduced) { buf.push_back(sym); st.push_back(Sub::leaf(-1, true)); }
        else { int n = g.add_node(sym); st.push_back(Sub::leaf(n, false)); }
    }
    void op_concat() {
        Sub b = pop(), a = pop();
        materialise_if_twin(b); materialise_if_twin(a);
        g.connect(a.end, b.start);
        Sub c;
        c.start = a.start; c.end = b.end;
        c.paths = a.paths * b.paths;  // concatInfo
        for (uint64_t x : a.lengths) for (uint64_t y : b.lengths) c.lengths.insert(x + y);
        // detect_bad_graphs (src/construction_tools.cpp:161-180)
        if (b.paths >= 15 || (c.paths >= 690000u && b.start != b.end)) {
            Catsite cs;
            cs.site = a.end; cs.first = b.start; cs.last = b.end; cs.gaps = b.lengths;
            cats.push_back(cs);
        }
        st.push_back(c);
    }
    void op_union() {
        Sub b = pop(), a = pop();
        if (reduced) {
            bool redundant = buf.size() >= 2 && buf[buf.size() - 1] == buf[buf.size() - 2];
            if (twin_test(b) && twin_test(a) && redundant) {  // identical reduced letters collapse
                int sym = buf.back(); buf.pop_back(); buf.pop_back();
                op_symbol(sym);
                return;
            }
            materialise_if_twin(a); materialise_if_twin(b);
        }
        int sp = g.add_node(kSplit);
        g.connect(sp, a.start); g.connect(sp, b.start);
        int gh = g.add_node(kGhost);
        g.connect(a.end, gh); g.connect(b.end, gh);
        Sub c;
        c.start = sp; c.end = gh;
        c.paths = a.paths + b.paths;  // unionInfo
        c.lengths = a.lengths;
        c.lengths.insert(b.lengths.begin(), b.lengths.end());
        st.push_back(c);
    }
    void op_optional() {
        Sub a = pop();
        materialise_if_twin(a);
        int sp = g.add_node(kSplit);
        g.connect(sp, a.start);
        int gh = g.add_node(kGhost);
        g.connect(sp, gh); g.connect(a.end, gh);
        Sub c;
        c.start = sp; c.end = gh;
        c.paths = a.paths + 1;  // optionInfo
        c.lengths = a.lengths;
        c.lengths.insert(0);
        st.push_back(c);
    }
    // `kk` arrives as `const uint8_t&` in the reference, so (max+1) is truncated mod 256.
    void op_kleene(uint8_t kk) {
        Sub a = pop();
        materialise_if_twin(a);
        int sp = g.add_node(kSplit);
        g.connect(sp, a.start);
        int gh = g.add_node(kGhost);
        g.connect(sp, gh);
        int back = a.end;
        for (uint8_t i = 1; (int)i < (int)kk - 1; ++i) {
            int isp = g.add_node(kSplit);
            g.connect(isp, gh);
            Sub c = copy(a);
            g.connect(back, isp);
            g.connect(isp, c.start);
            if ((int)i == (int)kk - 2) { g.connect(c.end, gh); break; }
            back = c.end;
        }
        Sub r;
        r.start = sp; r.end = gh;
        r.paths = a.paths * kk;  // kleeneInfo: lengths i*l for i < repeats
        for (uint64_t i = 0; i < kk; ++i) for (uint64_t l : a.lengths) r.lengths.insert(i * l);
        st.push_back(r);
    }
    void op_plus() {
        Sub a = pop();
        materialise_if_twin(a);
        int gh = g.add_node(kGhost);
        int back = a.end;
        for (uint8_t i = 1; (int)i < (int)k - 1; ++i) {
            int isp = g.add_node(kSplit);
            Sub c = copy(a);
            g.connect(back, isp);
            g.connect(isp, gh);
            g.connect(isp, c.start);
            if ((int)i == (int)k - 2) { g.connect(c.end, gh); break; }
            back = c.end;
        }
        Sub r;  // plus_procedure pushes a bare Subgraph{start, ghost}: paths 1, no lengths
        r.start = a.start; r.end = gh;
        st.push_back(r);
    }
    bool op_quant(size_t lo, size_t hi) {
        bool skip = false;
        if (lo == 0) {
            op_kleene((uint8_t)(hi + 1));
            if (st.size() != 1) { op_concat(); skip = true; }
            return skip;
        }
        if (st.empty()) throw std::runtime_error("k-graph stack underflow");
        Sub a = st.back();
        int sym = reduced ? buf_top() : 0;
        bool a_twin = reduced && twin_test(a);
        if (st.size() != 1) {
            op_concat();
            if (a_twin) op_symbol(sym);
            skip = true;
        }
        size_t extra = hi == 0 ? 0 : hi - lo;
        for (size_t i = 1; i < lo; ++i) {
            if (!reduced) st.push_back(copy(a));
            op_concat();
        }
        for (size_t i = 0; i < extra; ++i) {
            st.push_back(copy(a));
            op_optional();
            op_concat();
        }
        return skip;
    }

    void build(const std::string& postfix) {
        int start = g.add_node(kGhost);
        bool skip = false;
        for (size_t i = 0; i < postfix.size(); ++i) {
            int c = (unsigned char)postfix[i];
            if (std::isdigit(c)) continue;
            switch (c) {
                case '-': if (skip) { skip = false; continue; } op_concat(); break;
                case '|': op_union(); break;
                case '?': op_optional(); break;
                case '*': op_kleene(k); break;
                case '+': op_plus(); break;
                case '{': {
                    auto mm = parse_quant(postfix, i);
                    if (mm.first == 0 && mm.second == 1) { op_optional(); break; }
                    skip = op_quant(mm.first, mm.second);
                    break;
                }
                case '}': case ',': break;
                default: op_symbol((char)c); break;
            }
        }
        if (st.empty()) throw std::runtime_error("empty k-graph");
        // In the reduced builder a pattern that is a single pending symbol is never
        // materialised by the reference (it wires garbage ids); make that loud.
        if (st.back().twin) throw std::runtime_error("reduced builder: single-symbol pattern (undefined in the reference)");
        Sub top = st.back();
        g.connect(start, top.start);
        int m = g.add_node(kMatch);
        g.connect(top.end, m);
        st.pop_back();
        if (!st.empty()) throw std::runtime_error("k-graph stack not empty at end of postfix");
    }
};

}  // namespace txo

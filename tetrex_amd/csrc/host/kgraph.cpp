#include "kgraph.hpp"

#include <algorithm>
#include <cctype>
#include <stdexcept>

namespace tetrex {

int32_t KGraph::add(int32_t lab) {
    label.push_back(lab);
    gap.push_back(0);
    next_a.push_back(kNone);
    next_b.push_back(kNone);
    return size() - 1;
}

void KGraph::link(int32_t from, int32_t to) {
    arc_src.push_back(from);
    arc_dst.push_back(to);
    const int32_t lab = label[from];
    if (lab != kSplit) {                       // one successor: the newest arc wins both slots
        next_a[from] = next_b[from] = to;
    } else if (next_a[from] == kNone) {
        next_a[from] = to;                     // a split fills the first slot once ...
    } else {
        next_b[from] = to;                     // ... and the second slot with every later arc
    }
}

std::vector<int32_t> KGraph::topological_order() const {
    // Kahn's algorithm over the arc list; any topological order gives the same masks.
    const int32_t n = size();
    std::vector<int32_t> indeg(n, 0), head(n + 1, 0), adj(arc_src.size());
    for (size_t a = 0; a < arc_src.size(); ++a) { ++indeg[arc_dst[a]]; ++head[arc_src[a] + 1]; }
    for (int32_t i = 0; i < n; ++i) head[i + 1] += head[i];
    std::vector<int32_t> fill(head.begin(), head.end() - 1);
    for (size_t a = 0; a < arc_src.size(); ++a) adj[fill[arc_src[a]]++] = arc_dst[a];
    std::vector<int32_t> order;
    order.reserve(n);
    for (int32_t i = 0; i < n; ++i)
        if (indeg[i] == 0) order.push_back(i);
    for (size_t at = 0; at < order.size(); ++at) {
        const int32_t u = order[at];
        for (int32_t e = head[u]; e < head[u + 1]; ++e)
            if (--indeg[adj[e]] == 0) order.push_back(adj[e]);
    }
    if ((int32_t)order.size() != n) throw std::runtime_error("k-graph has a cycle");
    return order;
}

std::vector<int32_t> KGraph::reference_ranks() const {
    const int32_t n = size();
    std::vector<std::vector<int32_t>> out(n);
    for (size_t a = 0; a < arc_src.size(); ++a) out[arc_src[a]].push_back(arc_dst[a]);
    std::vector<int32_t> rank(n, 0), cursor(n, 0);
    std::vector<uint8_t> seen(n, 0);
    int32_t next = n;
    for (int32_t root = n - 1; root >= 0; --root) {
        if (seen[root]) continue;
        std::vector<int32_t> stack{root};
        seen[root] = 1;
        while (!stack.empty()) {
            const int32_t u = stack.back();
            if (cursor[u] < (int32_t)out[u].size()) {
                const int32_t v = out[u][out[u].size() - 1 - cursor[u]++];  // newest arc first
                if (!seen[v]) { seen[v] = 1; stack.push_back(v); }
            } else {
                rank[u] = --next;
                stack.pop_back();
            }
        }
    }
    return rank;
}

size_t KGraph::augment() {
    if (catsites.empty()) return 0;
    const std::vector<int32_t> rank = reference_ranks();
    std::vector<CatSite> cats = catsites;
    std::sort(cats.begin(), cats.end(), [&](const CatSite& a, const CatSite& b) { return rank[a.first] < rank[b.first]; });
    // neighbouring regions (consecutive ranks) fuse; their gap sets add pairwise
    std::vector<CatSite> merged;
    for (const CatSite& c : cats) {
        if (merged.empty() || rank[c.first] - 1 != rank[merged.back().last]) { merged.push_back(c); continue; }
        CatSite& m = merged.back();
        m.last = c.last;
        std::set<uint64_t> sum;
        for (uint64_t x : m.gaps) for (uint64_t y : c.gaps) sum.insert(x + y);
        m.gaps.swap(sum);
    }
    size_t added = 0;
    auto bypass = [&](int32_t from, int32_t to, uint64_t length) {
        const int32_t g = add(kGap);
        gap[g] = length;
        link(from, g);
        link(g, to);
        ++added;
    };
    for (const CatSite& c : merged) {
        const int32_t downstream = next_a[c.last];
        if (downstream == kNone) throw std::runtime_error("catastrophic region without a successor");
        if (c.gaps.size() == 1) { bypass(c.site, downstream, *c.gaps.begin()); continue; }
        // several lengths: a guard Split/Ghost pair.  A Split has two successor slots and every arc
        // after the first lands in the second (update_arc_map), so only the first and the last
        // length survive — in the reference in hash-set order, here in ascending order.
        const int32_t fork = add(kSplit), join = add(kGhost);
        link(c.site, fork);
        link(join, downstream);
        for (uint64_t length : c.gaps) bypass(fork, join, length);
    }
    return added;
}

std::string KGraph::to_graphviz() const {
    std::string out = "digraph kGraph\n{\n\trankdir=\"LR\";\n";
    for (int32_t v = size() - 1; v >= 0; --v) {  // the reference lists nodes from the newest down
        out += "\t" + std::to_string(v);
        if (v == 0) out += " [shape=point label=\"\"];\n";
        else if (label[v] == kSplit) out += " [label=\"\xC3\x98\"];\n";
        else if (label[v] == kGhost) out += " [label=\"\xE2\x80\xA2\"];\n";
        else if (label[v] == kMatch) out += " [shape=doublecircle label=\"\"];\n";
        else if (label[v] == kGap) out += " [label=\"GAP\"];\n";
        else out += std::string(" [label=\"") + (char)label[v] + "\"];\n";
    }
    // arcs that are still reachable through the successor slots (an augmented graph hides the bypassed ones)
    std::vector<uint8_t> seen(size(), 0);
    std::vector<int32_t> todo{0};
    seen[0] = 1;
    while (!todo.empty()) {
        const int32_t u = todo.back();
        todo.pop_back();
        int32_t targets[2] = {next_a[u], next_b[u] != next_a[u] ? next_b[u] : kNone};
        for (int32_t t : targets) {
            if (t == kNone) continue;
            out += "\t" + std::to_string(u) + "->" + std::to_string(t) + ";\n";
            if (!seen[t]) { seen[t] = 1; todo.push_back(t); }
        }
    }
    out += "}";
    return out;
}

namespace {

// A partially built piece of the graph: entry/exit node, or a symbol that the reduced builder
// has not turned into a node yet.
struct Fragment {
    int32_t entry = KGraph::kNone, exit = KGraph::kNone;
    bool pending = false;
    uint64_t paths = 1;          // Subgraph::paths (include/construction_tools.h:76)
    std::set<uint64_t> lengths;  // Subgraph::lengths
    bool single() const { return pending || entry == exit; }
    // with_stats: the path statistics (paths, lengths) are kept — only KGraph::augment() (-a) reads what comes of them
    static Fragment residue(int32_t node, bool pending, bool with_stats) {
        Fragment f;
        f.entry = f.exit = node;
        f.pending = pending;
        if (with_stats) f.lengths = {1};
        return f;
    }
    static Fragment span(int32_t entry, int32_t exit) {
        Fragment f;
        f.entry = entry;
        f.exit = exit;
        return f;
    }
};

class Builder {
  public:
    Builder(unsigned k, bool reduced, bool path_stats, bool fuse) : k_(k), reduced_(reduced), stats_(path_stats), fuse_(fuse && !reduced && !path_stats) {}

    KGraph finish(const std::string& postfix) {
        {  // a node per residue and two per operator at most (quantifiers that duplicate sub-graphs grow the vectors later)
            const size_t guess = postfix.size() + 2;
            g_.label.reserve(guess); g_.gap.reserve(guess); g_.next_a.reserve(guess); g_.next_b.reserve(guess);
            g_.arc_src.reserve(guess * 2); g_.arc_dst.reserve(guess * 2);
            if (fuse_) { g_.classes.reserve(16); g_.class_bytes.reserve(postfix.size()); }
            stack_.reserve(16);
        }
        const int32_t start = g_.add(KGraph::kGhost);
        bool skip_concat = false;
        for (size_t i = 0; i < postfix.size(); ++i) {
            const unsigned char c = (unsigned char)postfix[i];
            if (std::isdigit(c)) continue;
            switch (c) {
                case '-':
                    if (skip_concat) skip_concat = false;
                    else concat();
                    break;
                case '|': alternate(); break;
                case '?': optional(); break;
                case '*': star((uint8_t)k_); break;
                case '+': plus(); break;
                case '{': {
                    size_t lo, hi;
                    counts(postfix, i, lo, hi);
                    if (lo == 0 && hi == 1) optional();
                    else skip_concat = counted(lo, hi);
                    break;
                }
                case '}': case ',': break;
                default:
                    if (fuse_ && c >= 'A' && c <= 'Z') {
                        // `XY|Z|...`: a letter, then letters each followed by '|' — the union of a fresh operand with single
                        // residues, i.e. a class (a wildcard is spelt as 39 such characters): one node, made in one go
                        size_t j = i + 1;
                        uint32_t n = 1;
                        while (j + 1 < postfix.size() && postfix[j] >= 'A' && postfix[j] <= 'Z' && postfix[j + 1] == '|') { j += 2; ++n; }
                        if (n > 1) {
                            class_node(postfix, i, j);
                            i = j - 1;
                            break;
                        }
                    }
                    symbol(c);
                    break;
            }
        }
        if (stack_.empty()) throw std::runtime_error("empty query: nothing to search for");
        if (stack_.back().pending) throw std::runtime_error("single-residue query in a reduced alphabet is undefined in the reference");
        const Fragment whole = stack_.back();
        stack_.pop_back();
        if (!stack_.empty()) throw std::runtime_error("malformed postfix: operands left over");
        g_.link(start, whole.entry);
        const int32_t match = g_.add(KGraph::kMatch);
        g_.link(whole.exit, match);
        return std::move(g_);
    }

  private:
    KGraph g_;
    unsigned k_;
    bool reduced_;
    bool stats_;  // keep path statistics (for -a)
    bool fuse_;   // unions of single residues become one class node
    std::vector<Fragment> stack_;
    std::vector<int32_t> symbols_;  // reduced builder: residues waiting to become nodes

    static void counts(const std::string& p, size_t at, size_t& lo, size_t& hi) {
        const size_t close = p.find('}', at), comma = p.find(',', at);
        if (close == std::string::npos) throw std::runtime_error("quantifier without '}'");
        if (comma == std::string::npos || comma > close) {
            lo = (size_t)std::stoi(p.substr(at + 1, close - at));
            hi = 0;
        } else {
            lo = (size_t)std::stoi(p.substr(at + 1, comma - at));
            hi = (size_t)std::stoi(p.substr(comma + 1, close - comma - 1));
        }
    }

    Fragment take() {
        if (stack_.empty()) throw std::runtime_error("malformed query: operator without operand");
        Fragment f = stack_.back();
        stack_.pop_back();
        return f;
    }
    int32_t waiting_symbol() const {
        if (symbols_.empty()) throw std::runtime_error("reduced alphabet: no buffered residue (undefined in the reference)");
        return symbols_.back();
    }
    // the reduced builder turns ANY single-node fragment into a fresh node labelled with the
    // newest buffered residue (twin_test is `start == end`, construct_reduced_nfa.cpp:91-111)
    void realise(Fragment& f) {
        if (!reduced_ || !f.single()) return;
        const int32_t s = waiting_symbol();
        symbols_.pop_back();
        const int32_t n = g_.add(s);
        f = Fragment::residue(n, false, stats_);
    }

    // the class of postfix[from], postfix[from + 1], postfix[from + 3], ... (to = one past its last '|')
    void class_node(const std::string& postfix, size_t from, size_t to) {
        char members[32];
        uint32_t n = 0, letters = 0;
        auto put = [&](char c) {
            if (!((letters >> (c - 'A')) & 1u)) { letters |= 1u << (c - 'A'); members[n++] = c; }
        };
        put(postfix[from]);
        for (size_t j = from + 1; j < to; j += 2) put(postfix[j]);
        g_.classes.push_back(KGraph::ResidueClass{(uint32_t)g_.class_bytes.size(), n, letters});
        g_.class_bytes.append(members, n);
        stack_.push_back(Fragment::residue(g_.add(KGraph::kClass + (int32_t)g_.classes.size() - 1), false, stats_));
    }

    void symbol(int32_t s) {
        if (reduced_) {
            symbols_.push_back(s);
            stack_.push_back(Fragment::residue(KGraph::kNone, true, stats_));
        } else {
            stack_.push_back(Fragment::residue(g_.add(s), false, stats_));
        }
    }

    void concat() {
        Fragment right = take(), left = take();
        realise(right);
        realise(left);
        g_.link(left.exit, right.entry);
        Fragment both = Fragment::span(left.entry, right.exit);
        if (stats_) {
            both.paths = left.paths * right.paths;
            for (uint64_t x : left.lengths) for (uint64_t y : right.lengths) both.lengths.insert(x + y);
            if (right.paths >= 15 || (both.paths >= 690000u && right.entry != right.exit))
                g_.catsites.push_back(CatSite{left.exit, right.entry, right.exit, right.lengths});
        }
        stack_.push_back(both);
    }

    // A single-node fragment on the stack is a node nothing is linked to yet (operators replace what they take by spans),
    // and everything the graph has gained since `left` was pushed belongs to `right`: when both are letters or classes,
    // `right` is the graph's newest node and the two operands' class bytes are the tail of class_bytes.
    bool fusable(const Fragment& f) const {
        if (f.pending || f.entry != f.exit) return false;
        const int32_t lab = g_.label[f.entry];
        return lab >= KGraph::kClass || (lab >= 'A' && lab <= 'Z');
    }
    bool fuse(const Fragment& left, const Fragment& right) {
        if (!fusable(left) || !fusable(right) || right.entry != g_.size() - 1 || left.entry == right.entry) return false;
        char members[32];
        uint32_t n = 0, letters = 0;
        size_t cut = g_.class_bytes.size(), owned = 0;
        for (const int32_t v : {left.entry, right.entry})
            if (g_.label[v] >= KGraph::kClass) {
                const KGraph::ResidueClass& c = g_.classes[(size_t)(g_.label[v] - KGraph::kClass)];
                cut = std::min<size_t>(cut, c.first);
                owned += c.count;
            }
        if (g_.class_bytes.size() - cut != owned) return false;  // (never seen: somebody else's bytes would be cut off)
        auto take_in = [&](int32_t v) {
            g_.for_each_residue(v, [&](unsigned char c) {
                if (!((letters >> (c - 'A')) & 1u)) { letters |= 1u << (c - 'A'); members[n++] = (char)c; }
            });
        };
        take_in(left.entry);
        take_in(right.entry);
        g_.class_bytes.resize(cut);
        g_.classes.push_back(KGraph::ResidueClass{(uint32_t)cut, n, letters});
        g_.class_bytes.append(members, n);
        g_.label[left.entry] = KGraph::kClass + (int32_t)g_.classes.size() - 1;
        g_.label.pop_back();  // `right`, the newest node, never had an arc
        g_.gap.pop_back();
        g_.next_a.pop_back();
        g_.next_b.pop_back();
        return true;
    }

    void alternate() {
        Fragment right = take(), left = take();
        if (fuse_ && fuse(left, right)) { stack_.push_back(left); return; }
        if (reduced_) {
            const size_t n = symbols_.size();
            if (left.single() && right.single() && n >= 2 && symbols_[n - 1] == symbols_[n - 2]) {
                const int32_t s = symbols_.back();  // both branches are the same reduced letter
                symbols_.pop_back();
                symbols_.pop_back();
                symbol(s);
                return;
            }
            realise(left);
            realise(right);
        }
        const int32_t fork = g_.add(KGraph::kSplit);
        g_.link(fork, left.entry);
        g_.link(fork, right.entry);
        const int32_t join = g_.add(KGraph::kGhost);
        g_.link(left.exit, join);
        g_.link(right.exit, join);
        Fragment either = Fragment::span(fork, join);
        if (stats_) {
            either.paths = left.paths + right.paths;
            either.lengths = left.lengths;
            either.lengths.insert(right.lengths.begin(), right.lengths.end());
        }
        stack_.push_back(either);
    }

    void optional() {
        Fragment body = take();
        realise(body);
        const int32_t fork = g_.add(KGraph::kSplit);
        g_.link(fork, body.entry);
        const int32_t join = g_.add(KGraph::kGhost);
        g_.link(fork, join);
        g_.link(body.exit, join);
        Fragment maybe = Fragment::span(fork, join);
        if (stats_) {
            maybe.paths = body.paths + 1;
            maybe.lengths = body.lengths;
            maybe.lengths.insert(0);
        }
        stack_.push_back(maybe);
    }

    // duplicate the sub-graph between f.entry and f.exit (nodes on some entry->exit path)
    Fragment duplicate(const Fragment& f) {
        if (f.single()) {
            int32_t lab = reduced_ ? waiting_symbol() : g_.label[f.entry];
            if (lab >= KGraph::kClass) {  // a class of its own for the copy: its bytes must be the tail of class_bytes while it can still fuse
                const KGraph::ResidueClass c = g_.classes[(size_t)(lab - KGraph::kClass)];
                g_.classes.push_back(KGraph::ResidueClass{(uint32_t)g_.class_bytes.size(), c.count, c.letters});
                g_.class_bytes.append(g_.class_bytes, c.first, c.count);
                lab = KGraph::kClass + (int32_t)g_.classes.size() - 1;
            }
            Fragment copy = Fragment::residue(g_.add(lab), false, stats_);
            copy.paths = f.paths;
            copy.lengths = f.lengths;
            return copy;
        }
        const int32_t n0 = g_.size();
        const size_t a0 = g_.arc_src.size();
        std::vector<uint8_t> mark(n0, 0);
        flood(f.entry, true, mark, 1);
        flood(f.exit, false, mark, 2);
        std::vector<int32_t> image(n0, KGraph::kNone);
        for (int32_t v = n0 - 1; v >= 0; --v)         // newest node first, as lemon's NodeIt
            if (mark[v] == 3) image[v] = g_.add(g_.label[v]);
        for (size_t a = a0; a-- > 0;) {                // newest arc first, as lemon's ArcIt
            const int32_t u = g_.arc_src[a], v = g_.arc_dst[a];
            if (image[u] != KGraph::kNone && image[v] != KGraph::kNone) g_.link(image[u], image[v]);
        }
        Fragment copy = Fragment::span(image[f.entry], image[f.exit]);
        copy.paths = f.paths;
        copy.lengths = f.lengths;
        return copy;
    }
    void flood(int32_t from, bool forward, std::vector<uint8_t>& mark, uint8_t bit) const {
        // adjacency (CSR) of the current arc list in the requested direction
        const int32_t n = g_.size();
        const std::vector<int32_t>& tail = forward ? g_.arc_src : g_.arc_dst;
        const std::vector<int32_t>& head = forward ? g_.arc_dst : g_.arc_src;
        std::vector<int32_t> first(n + 1, 0), nbr(tail.size());
        for (int32_t u : tail) ++first[u + 1];
        for (int32_t i = 0; i < n; ++i) first[i + 1] += first[i];
        std::vector<int32_t> fill(first.begin(), first.end() - 1);
        for (size_t a = 0; a < tail.size(); ++a) nbr[fill[tail[a]]++] = head[a];
        std::vector<int32_t> todo{from};
        mark[from] |= bit;
        while (!todo.empty()) {
            const int32_t u = todo.back();
            todo.pop_back();
            for (int32_t e = first[u]; e < first[u + 1]; ++e) {
                const int32_t t = nbr[e];
                if (!(mark[t] & bit)) { mark[t] |= bit; todo.push_back(t); }
            }
        }
    }

    // X* unrolled to at most depth-1 repetitions; depth arrives as uint8_t in the reference
    // (`const uint8_t& k`), so {0,n} with n+1 > 255 wraps exactly like there.
    void star(uint8_t depth) {
        Fragment body = take();
        realise(body);
        const int32_t fork = g_.add(KGraph::kSplit);
        g_.link(fork, body.entry);
        const int32_t join = g_.add(KGraph::kGhost);
        g_.link(fork, join);
        int32_t tail = body.exit;
        for (int i = 1; i < (int)depth - 1; ++i) {
            const int32_t again = g_.add(KGraph::kSplit);
            g_.link(again, join);
            const Fragment copy = duplicate(body);
            g_.link(tail, again);
            g_.link(again, copy.entry);
            if (i == (int)depth - 2) { g_.link(copy.exit, join); break; }
            tail = copy.exit;
        }
        Fragment loop = Fragment::span(fork, join);
        if (stats_) {
            loop.paths = body.paths * depth;
            for (uint64_t i = 0; i < depth; ++i) for (uint64_t l : body.lengths) loop.lengths.insert(i * l);
        }
        stack_.push_back(loop);
    }

    void plus() {
        Fragment body = take();
        realise(body);
        const int32_t join = g_.add(KGraph::kGhost);
        int32_t tail = body.exit;
        for (int i = 1; i < (int)k_ - 1; ++i) {
            const int32_t again = g_.add(KGraph::kSplit);
            const Fragment copy = duplicate(body);
            g_.link(tail, again);
            g_.link(again, join);
            g_.link(again, copy.entry);
            if (i == (int)k_ - 2) { g_.link(copy.exit, join); break; }
            tail = copy.exit;
        }
        stack_.push_back(Fragment::span(body.entry, join));  // the reference records no path statistics for '+'
    }

    // X{lo} / X{lo,hi}; returns true when the following '-' of the postfix has been consumed
    bool counted(size_t lo, size_t hi) {
        bool consumed = false;
        if (lo == 0) {
            star((uint8_t)(hi + 1));
            if (stack_.size() != 1) { concat(); consumed = true; }
            return consumed;
        }
        if (stack_.empty()) throw std::runtime_error("malformed query: quantifier without operand");
        const Fragment body = stack_.back();
        const int32_t s = reduced_ ? waiting_symbol() : 0;
        if (stack_.size() != 1) {
            concat();
            if (reduced_ && body.single()) symbol(s);
            consumed = true;
        }
        const size_t extra = hi == 0 ? 0 : hi - lo;
        for (size_t i = 1; i < lo; ++i) {
            if (!reduced_) stack_.push_back(duplicate(body));
            concat();
        }
        for (size_t i = 0; i < extra; ++i) {
            stack_.push_back(duplicate(body));
            optional();
            concat();
        }
        return consumed;
    }
};

}  // namespace

KGraph build_kgraph(const std::string& postfix, unsigned k, bool reduced_alphabet, bool path_stats, bool fuse_classes) {
    return Builder(k, reduced_alphabet, path_stats, fuse_classes).finish(postfix);
}

}  // namespace tetrex

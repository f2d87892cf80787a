#include "matcher.hpp"

#include <algorithm>
#include <cstring>
#include <functional>
#include <stdexcept>

namespace tetrex {

namespace {

enum NodeType : uint8_t { nEmpty, nSet, nCat, nAlt, nStar, nPlus, nQuest, nRepeat, nBegin, nEnd };
struct Node { NodeType type; int a = -1, b = -1; int lo = 0, hi = 0; uint32_t set = 0; };  // hi < 0: unbounded

struct Parser {
    const std::string& src;
    size_t at = 0;
    std::vector<Node> nodes;
    std::vector<std::array<uint64_t, 4>>& sets;

    [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("regex: ") + what + " at offset " + std::to_string(at)); }
    int make(Node n) { nodes.push_back(n); return (int)nodes.size() - 1; }
    uint32_t intern(const std::array<uint64_t, 4>& s) {
        for (size_t i = 0; i < sets.size(); ++i) if (sets[i] == s) return (uint32_t)i;
        sets.push_back(s);
        return (uint32_t)sets.size() - 1;
    }
    static void add(std::array<uint64_t, 4>& s, unsigned c) { s[c >> 6] |= 1ULL << (c & 63); }

    int alt() {
        int left = cat();
        while (at < src.size() && src[at] == '|') {
            ++at;
            Node n{nAlt};
            n.a = left;
            n.b = cat();
            left = make(n);
        }
        return left;
    }
    int cat() {
        int left = -1;
        while (at < src.size() && src[at] != '|' && src[at] != ')') {
            const int r = repeat();
            if (left < 0) left = r;
            else { Node n{nCat}; n.a = left; n.b = r; left = make(n); }
        }
        return left < 0 ? make(Node{nEmpty}) : left;
    }
    int repeat() {
        int a = atom();
        for (;;) {
            if (at >= src.size()) return a;
            const char c = src[at];
            if (c == '*' || c == '+' || c == '?') {
                ++at;
                Node n{c == '*' ? nStar : c == '+' ? nPlus : nQuest};
                n.a = a;
                a = make(n);
            } else if (c == '{') {
                size_t p = at + 1;
                auto number = [&](int& v) {
                    if (p >= src.size() || src[p] < '0' || src[p] > '9') return false;
                    long x = 0;
                    while (p < src.size() && src[p] >= '0' && src[p] <= '9') { x = x * 10 + (src[p++] - '0'); if (x > 1000) fail("repetition count above 1000"); }
                    v = (int)x;
                    return true;
                };
                Node n{nRepeat};
                n.a = a;
                if (!number(n.lo)) return a;  // a literal '{' (RE2 treats a malformed repetition literally)
                n.hi = n.lo;
                if (p < src.size() && src[p] == ',') {
                    ++p;
                    if (!number(n.hi)) n.hi = -1;
                }
                if (p >= src.size() || src[p] != '}') return a;
                if (n.hi >= 0 && n.hi < n.lo) fail("bad repetition range");
                at = p + 1;
                a = make(n);
            } else return a;
        }
    }
    int atom() {
        if (at >= src.size()) fail("missing operand");
        const unsigned char c = (unsigned char)src[at++];
        std::array<uint64_t, 4> s{};
        switch (c) {
            case '(': {
                if (at + 1 < src.size() && src[at] == '?' && src[at + 1] == ':') at += 2;
                const int inner = alt();
                if (at >= src.size() || src[at] != ')') fail("missing )");
                ++at;
                return inner;
            }
            case '^': return make(Node{nBegin});
            case '$': return make(Node{nEnd});
            case '.':
                for (unsigned b = 0; b < 256; ++b) if (b != '\n') add(s, b);
                break;
            case '[': {
                bool negate = false;
                if (at < src.size() && src[at] == '^') { negate = true; ++at; }
                bool first = true;
                for (;; first = false) {
                    if (at >= src.size()) fail("missing ]");
                    unsigned lo = (unsigned char)src[at++];
                    if (lo == ']' && !first) break;
                    if (lo == '\\' && at < src.size()) lo = (unsigned char)src[at++];
                    unsigned hi = lo;
                    if (at + 1 < src.size() && src[at] == '-' && src[at + 1] != ']') {
                        hi = (unsigned char)src[at + 1];
                        at += 2;
                        if (hi == '\\' && at < src.size()) hi = (unsigned char)src[at++];
                        if (hi < lo) fail("bad character range");
                    }
                    for (unsigned b = lo; b <= hi; ++b) add(s, b);
                }
                if (negate) for (auto& w : s) w = ~w;
                break;
            }
            case '\\':
                if (at >= src.size()) fail("trailing backslash");
                add(s, (unsigned char)src[at++]);
                break;
            case '*': case '+': case '?': fail("repetition operator without operand");
            case ')': fail("unmatched )");
            default: add(s, c);
        }
        Node n{nSet};
        n.set = intern(s);
        return make(n);
    }
};

}  // namespace

Matcher::Matcher(const std::string& pattern, Semantics semantics) : semantics_(semantics) {
    Parser ps{pattern, 0, {}, sets_};
    std::array<uint64_t, 4> any;
    any.fill(~0ULL);
    const uint32_t any_set = ps.intern(any);
    const int root = ps.alt();
    if (ps.at != pattern.size()) ps.fail("unmatched )");
    // byte equivalence classes: bytes that no set of the pattern tells apart
    {
        std::map<std::vector<uint8_t>, uint8_t> seen;
        for (unsigned b = 0; b < 256; ++b) {
            std::vector<uint8_t> sig(sets_.size());
            for (size_t i = 0; i < sets_.size(); ++i) sig[i] = (sets_[i][b >> 6] >> (b & 63)) & 1;
            auto it = seen.emplace(sig, (uint8_t)seen.size()).first;
            class_of_[b] = it->second;
        }
        n_classes_ = (uint32_t)seen.size();
        set_has_class_.assign(sets_.size(), std::vector<uint8_t>(n_classes_, 0));
        for (unsigned b = 0; b < 256; ++b)
            for (size_t i = 0; i < sets_.size(); ++i)
                if ((sets_[i][b >> 6] >> (b & 63)) & 1) set_has_class_[i][class_of_[b]] = 1;
    }
    // continuation-style Thompson construction: compile(node, next) returns the entry of code that runs `node` and goes on at `next`
    auto build = [&](bool reverse, Prog& prog) {
        std::vector<Inst>& code = prog.inst;
        auto emit = [&](Inst i) { code.push_back(i); if (code.size() > 2000000) throw std::runtime_error("regex: pattern too large"); return (uint32_t)code.size() - 1; };
        std::function<uint32_t(int, uint32_t)> compile = [&](int id, uint32_t next) -> uint32_t {
            const Node n = ps.nodes[id];
            switch (n.type) {
                case nEmpty: return next;
                case nSet: return emit(Inst{kChar, n.set, next});
                case nBegin: return emit(Inst{(uint8_t)(reverse ? kEnd : kBegin), next, 0});
                case nEnd: return emit(Inst{(uint8_t)(reverse ? kBegin : kEnd), next, 0});
                case nCat: return reverse ? compile(n.b, compile(n.a, next)) : compile(n.a, compile(n.b, next));
                case nAlt: { const uint32_t a = compile(n.a, next), b = compile(n.b, next); return emit(Inst{kSplit, a, b}); }
                case nQuest: { const uint32_t a = compile(n.a, next); return emit(Inst{kSplit, a, next}); }
                case nStar: { const uint32_t loop = emit(Inst{kSplit, 0, next}); code[loop].x = compile(n.a, loop); return loop; }
                case nPlus: { const uint32_t loop = emit(Inst{kSplit, 0, next}); const uint32_t body = compile(n.a, loop); code[loop].x = body; return body; }
                case nRepeat: {
                    uint32_t tail = next;
                    if (n.hi < 0) { const uint32_t loop = emit(Inst{kSplit, 0, next}); code[loop].x = compile(n.a, loop); tail = loop; }
                    else for (int i = n.lo; i < n.hi; ++i) { const uint32_t a = compile(n.a, tail); tail = emit(Inst{kSplit, a, next}); }  // (a(a(..)?)?)?
                    for (int i = 0; i < n.lo; ++i) tail = compile(n.a, tail);
                    return tail;
                }
                default: return next;
            }
        };
        const uint32_t match = emit(Inst{kMatch, 0, 0});
        prog.start = compile(root, match);
        const uint32_t loop = emit(Inst{kSplit, prog.start, 0});  // the pattern first, then one more byte and again
        code[loop].y = emit(Inst{kChar, any_set, loop});
        prog.unanchored = loop;
    };
    build(false, fwd_);
    build(true, rev_);
    for (const Inst& in : fwd_.inst) has_begin_ = has_begin_ || in.op == kBegin;
    // The longest run of single-byte factors on the pattern's spine — the factors every match goes through in order: the
    // concatenation at the top, and what sits inside `+` / {m,..} with m >= 1 (at least one round) — is a string every
    // match contains.  Alternations, optional parts and byte sets end a run and contribute nothing.
    {
        std::string run;
        auto single = [&](const Node& n, unsigned char* byte) {
            const std::array<uint64_t, 4>& st = sets_[n.set];
            int bits = 0;
            for (int w = 0; w < 4; ++w) bits += __builtin_popcountll(st[w]);
            if (bits != 1) return false;
            for (int w = 0; w < 4; ++w) if (st[w]) *byte = (unsigned char)(w * 64 + __builtin_ctzll(st[w]));
            return true;
        };
        std::function<void(int)> walk = [&](int id) {
            const Node n = ps.nodes[id];
            unsigned char b = 0;
            switch (n.type) {
                case nCat: walk(n.a); walk(n.b); return;
                case nSet:
                    if (single(n, &b)) { run.push_back((char)b); if (run.size() > literal_.size()) literal_ = run; }
                    else run.clear();
                    return;
                case nPlus: run.clear(); walk(n.a); run.clear(); return;
                case nRepeat: run.clear(); if (n.lo >= 1) walk(n.a); run.clear(); return;
                case nEmpty: return;
                default: run.clear(); return;  // nAlt, nStar, nQuest, anchors
            }
        };
        walk(root);
    }
    // ... and the most selective run of single-byte factors, whatever their sets (selectivity: the product of |set| / 256 over the
    // run — the run whose product is smallest; '.' contributes nothing but keeps the run together)
    {
        std::vector<uint32_t> cur;
        double best = 1.0;
        auto weight = [&](uint32_t set) {
            int bits = 0;
            for (int w = 0; w < 4; ++w) bits += __builtin_popcountll(sets_[set][w]);
            return bits >= 200 ? 1.0 : bits / 24.0;  // (a class of residues: out of some twenty letters; '.' and negated sets: no help)
        };
        auto close = [&]() {
            double prod = 1.0;
            for (uint32_t x : cur) prod *= std::min(1.0, weight(x));
            if (!cur.empty() && prod < best) {
                best = prod;
                run_.clear();
                for (uint32_t x : cur) run_.push_back(sets_[x]);
            }
            cur.clear();
        };
        std::function<void(int)> walk = [&](int id) {
            const Node n = ps.nodes[id];
            switch (n.type) {
                case nCat: walk(n.a); walk(n.b); return;
                case nSet: cur.push_back(n.set); if (cur.size() >= 64) close(); return;
                case nPlus: close(); walk(n.a); close(); return;
                case nRepeat: close(); if (n.lo >= 1) walk(n.a); close(); return;
                case nEmpty: return;
                default: close(); return;
            }
        };
        walk(root);
        close();
        // leading / trailing factors that select nothing are dropped; a run that selects less than one position in 400 is not kept
        while (!run_.empty() && [&] { int b = 0; for (int w = 0; w < 4; ++w) b += __builtin_popcountll(run_.back()[w]); return b >= 200; }()) run_.pop_back();
        while (!run_.empty() && [&] { int b = 0; for (int w = 0; w < 4; ++w) b += __builtin_popcountll(run_.front()[w]); return b >= 200; }()) run_.erase(run_.begin());
        if (best > 1.0 / 400) run_.clear();
    }
}

bool Matcher::may_match(std::string_view text) const {
    if (literal_.empty()) return true;
    if (literal_.size() == 1) return std::memchr(text.data(), (unsigned char)literal_[0], text.size()) != nullptr;
    return memmem(text.data(), text.size(), literal_.data(), literal_.size()) != nullptr;
}

// epsilon closure of `seeds` in priority order: the Char and Match instructions reachable without consuming a byte
void Matcher::closure(const Prog& p, const std::vector<uint32_t>& seeds, bool at_begin, bool at_end, std::vector<uint32_t>& out, Cache& c) const {
    if (c.mark.size() < p.inst.size()) c.mark.assign(p.inst.size(), 0);
    if (++c.epoch == 0) { std::fill(c.mark.begin(), c.mark.end(), 0); c.epoch = 1; }
    out.clear();
    for (uint32_t seed : seeds) {
        c.stack.clear();
        c.stack.push_back(seed);
        while (!c.stack.empty()) {
            const uint32_t pc = c.stack.back();
            c.stack.pop_back();
            if (c.mark[pc] == c.epoch) continue;
            c.mark[pc] = c.epoch;
            const Inst& in = p.inst[pc];
            switch (in.op) {
                case kChar: case kMatch: out.push_back(pc); break;
                case kJmp: c.stack.push_back(in.x); break;
                case kSplit: c.stack.push_back(in.y); c.stack.push_back(in.x); break;  // x is taken first
                case kBegin: if (at_begin) c.stack.push_back(in.x); break;
                case kEnd: if (at_end) c.stack.push_back(in.x); break;
            }
        }
    }
}

uint32_t Matcher::dfa_state(const Prog& p, Cache::Dfa& d, std::vector<uint32_t>& seeds, bool at_begin, Cache& c) const {
    std::vector<uint32_t> set, with_end;
    closure(p, seeds, at_begin, false, set, c);
    closure(p, seeds, at_begin, true, with_end, c);
    uint8_t flags = 0;
    for (uint32_t pc : set) if (p.inst[pc].op == kMatch) flags |= 1;
    for (uint32_t pc : with_end) if (p.inst[pc].op == kMatch) flags |= 2;
    std::sort(set.begin(), set.end());
    if (set.empty() && !flags) return kDead;
    std::vector<uint32_t> key = set;
    key.push_back(0xFFFFFFF0u | flags);  // the same instructions with a different "accepts at the end" are different states
    auto it = d.ids.find(key);
    if (it != d.ids.end()) return it->second;
    const uint32_t id = (uint32_t)d.sets.size();
    d.ids.emplace(std::move(key), id);
    d.sets.push_back(std::move(set));
    d.flags.push_back(flags);
    d.next.resize((size_t)(id + 1) * n_classes_, kUnknown);
    return id;
}

uint32_t Matcher::dfa_step(const Prog& p, Cache::Dfa& d, uint32_t state, uint32_t cls, Cache& c) const {
    std::vector<uint32_t> seeds;
    for (uint32_t pc : d.sets[state]) {
        const Inst& in = p.inst[pc];
        if (in.op == kChar && set_has_class_[in.x][cls]) seeds.push_back(in.y);
    }
    const uint32_t to = seeds.empty() ? kDead : dfa_state(p, d, seeds, false, c);
    d.next[(size_t)state * n_classes_ + cls] = to;
    return to;
}

void Matcher::dfa_init(const Prog& p, bool unanchored, Cache::Dfa& d, Cache& c) const {
    std::vector<uint32_t> seeds{unanchored ? p.unanchored : p.start};
    d.start_begin = dfa_state(p, d, seeds, true, c);
    d.start_mid = dfa_state(p, d, seeds, false, c);
    d.ready = true;
}

void Matcher::match_starts(std::string_view text, Cache& c) const {
    Cache::Dfa& d = c.rev;
    if (d.sets.size() > 20000) d = Cache::Dfa{};  // a pathological pattern: start over rather than grow without bound
    if (!d.ready) {
        dfa_init(rev_, true, d, c);
        // the resting state: nothing of the pattern in flight.  Bytes that keep it there are skipped in a tight loop.
        c.rest_stays.assign(256, 0);  // by byte: one table look-up per skipped byte
        if (d.start_mid != kDead && d.flags[d.start_mid] == 0) {
            std::vector<uint8_t> by_class(n_classes_);
            for (uint32_t k = 0; k < n_classes_; ++k) by_class[k] = dfa_step(rev_, d, d.start_mid, k, c) == d.start_mid;
            for (unsigned b = 0; b < 256; ++b) c.rest_stays[b] = by_class[class_of_[b]];
        }
        // a motif that ends in one fixed residue leaves the resting state on that byte only: memrchr finds it
        c.single_leaver = -1;
        int leavers = 0;
        for (unsigned b = 0; b < 256; ++b) if (!c.rest_stays[b]) { ++leavers; c.single_leaver = (int)b; }
        if (leavers != 1) c.single_leaver = -1;
    }
    c.starts.clear();
    const size_t n = text.size();
    const unsigned char* t = reinterpret_cast<const unsigned char*>(text.data());
    uint32_t state = d.start_begin;  // the reversed scan begins at the END of the text: that is where '$' holds
    if (state == kDead) return;
    if ((d.flags[state] & 1) || (n == 0 && (d.flags[state] & 2))) c.starts.push_back(n);
    const uint32_t rest = d.start_mid;
    const uint8_t* stays = c.rest_stays.data();
    for (size_t i = 1; i <= n; ++i) {  // i bytes consumed from the end; the text position reached is n - i
        if (state == rest) {  // nothing of the pattern in flight: look for the next byte that starts something
            const unsigned char* p = t + (n - i);  // the byte about to be consumed; the scan runs towards t
            if (c.single_leaver >= 0) {
                p = static_cast<const unsigned char*>(memrchr(t, c.single_leaver, (size_t)(p - t) + 1));
                if (!p) break;
            } else {
                while (p - t >= 4 && (stays[p[0]] & stays[p[-1]] & stays[p[-2]] & stays[p[-3]])) p -= 4;
                while (p >= t && stays[*p]) --p;
                if (p < t) break;  // ran off the beginning while resting (`rest` cannot accept at the end either: its flags are 0)
            }
            i = n - (size_t)(p - t);
        }
        const uint32_t cls = class_of_[t[n - i]];
        uint32_t to = d.next[(size_t)state * n_classes_ + cls];
        if (to == kUnknown) to = dfa_step(rev_, d, state, cls, c);
        if (to == kDead) return;  // cannot happen for an unanchored program; defensive
        state = to;
        const uint8_t f = d.flags[state];
        if (f && ((f & 1) || (i == n && (f & 2)))) c.starts.push_back(n - i);
    }
}

size_t Matcher::match_end(std::string_view text, size_t start, Cache& c, bool at_begin, bool* matched) const {
    if (semantics_ == Semantics::LeftmostFirst) return pike_end(text, start, c, at_begin, matched);
    Cache::Dfa& d = c.fwd;
    if (d.sets.size() > 20000) d = Cache::Dfa{};
    if (!d.ready) dfa_init(fwd_, false, d, c);
    const size_t n = text.size();
    const unsigned char* t = reinterpret_cast<const unsigned char*>(text.data());
    uint32_t state = at_begin ? d.start_begin : d.start_mid;
    size_t last = start;
    bool any = false;
    for (size_t pos = start; state != kDead; ++pos) {
        const uint8_t f = d.flags[state];
        if ((f & 1) || (pos == n && (f & 2))) { last = pos; any = true; }
        if (pos == n) break;
        const uint32_t cls = class_of_[t[pos]];
        uint32_t to = d.next[(size_t)state * n_classes_ + cls];
        if (to == kUnknown) to = dfa_step(fwd_, d, state, cls, c);
        state = to;
    }
    if (matched) *matched = any;
    return last;
}

// Pike VM anchored at `start`: threads in priority order (alternatives left to right, quantifiers greedy); the match of the
// highest-priority thread that reaches Match wins, and cuts off everything of lower priority (RE2 default / Perl semantics).
size_t Matcher::pike_end(std::string_view text, size_t start, Cache& c, bool at_begin, bool* matched) const {
    const Prog& p = fwd_;
    const size_t n = text.size();
    std::vector<uint32_t>& cur = c.clist;
    std::vector<uint32_t>& nxt = c.nlist;
    std::vector<uint32_t> seeds{p.start};
    closure(p, seeds, at_begin, start == n, cur, c);
    size_t last = start;
    bool any = false;
    for (size_t pos = start;; ++pos) {
        seeds.clear();
        for (uint32_t pc : cur) {
            const Inst& in = p.inst[pc];
            if (in.op == kMatch) { last = pos; any = true; break; }  // lower-priority threads are cut off
            if (pos < n && ((sets_[in.x][(unsigned char)text[pos] >> 6] >> ((unsigned char)text[pos] & 63)) & 1)) seeds.push_back(in.y);
        }
        if (pos >= n || seeds.empty()) break;
        closure(p, seeds, false, pos + 1 == n, nxt, c);
        cur.swap(nxt);
    }
    if (matched) *matched = any;
    return last;
}

bool Matcher::contains(std::string_view text, Cache& c) const {
    if (!may_match(text)) return false;
    Cache::Dfa& d = c.fwd_un;
    if (d.sets.size() > 20000) d = Cache::Dfa{};
    if (!d.ready) dfa_init(fwd_, true, d, c);
    const size_t n = text.size();
    const unsigned char* t = reinterpret_cast<const unsigned char*>(text.data());
    uint32_t state = d.start_begin;
    for (size_t pos = 0; state != kDead; ++pos) {
        const uint8_t f = d.flags[state];
        if ((f & 1) || (pos == n && (f & 2))) return true;
        if (pos == n) break;
        const uint32_t cls = class_of_[t[pos]];
        uint32_t to = d.next[(size_t)state * n_classes_ + cls];
        if (to == kUnknown) to = dfa_step(fwd_, d, state, cls, c);
        state = to;
    }
    return false;
}

}  // namespace tetrex

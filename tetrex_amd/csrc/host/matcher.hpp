// Host side: the regular-expression matcher of the verification stage — product code.
//
// The reference verifies candidate bins with RE2 (absent here): `RE2::FindAndConsume` in a loop, default RE2 syntax
// (leftmost-first, Perl-like) for DNA (include/query.h:103) and RE2::POSIX (leftmost-longest) for peptides
// (include/query.h:148); src/query.cpp:194-237 prints every successive non-overlapping match.  RE2's property that
// matters operationally is linear time without recursion: a chromosome-length record with a `+` motif must neither
// overflow the stack nor take exponential time, which a backtracking matcher (std::regex) does.
//
// This matcher keeps that property with two automata built from one parse of the pattern:
//   * match STARTS: the reversed pattern, unanchored, as a lazily built DFA run over the text from its end — every
//     position where it accepts is a position where a match of the pattern starts (one pass per record; while the
//     DFA rests in its start state the scan only looks for a byte that leaves it);
//   * match END for a given start: the forward pattern anchored there — as a DFA run until it dies (the last accept
//     is the longest match: POSIX), or as a Pike VM with ordered threads (the first-priority match: RE2 default).
// find_all then walks the starts left to right exactly like the FindAndConsume loop.
// Grammar: literals, '.', [sets] / [^sets] with ranges, ( ), |, * + ? {m} {m,} {m,n}, ^ $ (text begin / end), \x escapes.
#pragma once
#include <array>
#include <cstdint>
#include <map>
#include <string>
#include <string_view>
#include <vector>

namespace tetrex {

class Matcher {
  public:
    enum class Semantics { LeftmostLongest, LeftmostFirst };
    Matcher(const std::string& pattern, Semantics semantics);  // throws std::runtime_error on a syntax error

    // Lazily built DFA states; one per thread that matches (the Matcher itself is immutable and shared).
    class Cache {
      public:
        Cache() = default;
      private:
        friend class Matcher;
        struct Dfa {
            std::map<std::vector<uint32_t>, uint32_t> ids;
            std::vector<std::vector<uint32_t>> sets;  // per state: its NFA instructions
            std::vector<uint32_t> next;                // [state * n_classes + class], kUnknown until built
            std::vector<uint8_t> flags;                // bit 0: accepts here, bit 1: accepts if the scan ends here
            uint32_t start_begin = 0, start_mid = 0;   // start states at / after the scan's first position
            bool ready = false;
        } rev, fwd, fwd_un;  // reversed unanchored (match starts), forward anchored (match end), forward unanchored (contains)
        std::vector<uint8_t> rest_stays;  // rev: bytes on which the resting start state stays put
        int single_leaver = -1;           // the one byte that leaves it, if there is exactly one
        std::vector<size_t> starts;
        std::vector<uint32_t> stack, clist, nlist;
        std::vector<uint32_t> mark;
        uint32_t epoch = 0;
    };

    // successive non-overlapping matches, each searched in what the previous one left (the FindAndConsume loop of
    // src/query.cpp:206-216); fn(start, length).  FindAndConsume hands RE2 the REMAINING text as the whole text, so `^`
    // holds again right behind a (non-empty) match: `^M.K` finds MAK twice in MAKMAK.  The starts are collected with `^` at
    // the record's beginning only; where the pattern has a `^`, the position a match ended at is tried under that rule first.
    template <class Fn>
    void find_all(std::string_view text, Cache& cache, Fn&& fn) const {
        if (!may_match(text)) return;  // a string every match contains is not in the text (memmem: far cheaper than the automata)
        match_starts(text, cache);
        size_t pos = 0;
        bool fresh_begin = false;  // `pos` is where a non-empty match ended: the beginning of what FindAndConsume searches next
        size_t i = cache.starts.size();  // starts are collected right to left
        for (;;) {
            size_t s, e;
            bool found = false;
            if (has_begin_ && fresh_begin && pos <= text.size()) {
                e = match_end(text, pos, cache, true, &found);
                s = pos;
            }
            if (!found) {
                while (i > 0 && cache.starts[i - 1] < pos) --i;
                if (i == 0) return;
                s = cache.starts[--i];
                e = match_end(text, s, cache, s == 0 || (fresh_begin && s == pos));
            }
            fn(s, e - s);
            fresh_begin = e > s;
            pos = e > s ? e : s + 1;
        }
    }
    // does the pattern match anywhere in the text?
    bool contains(std::string_view text, Cache& cache) const;
    // The prefilter in front of the automata: false = the text lacks a string that every match of the pattern contains
    // (required_literal(): the longest run of plain bytes on the pattern's spine; empty when there is none).
    bool may_match(std::string_view text) const;
    const std::string& required_literal() const { return literal_; }
    // The same idea for motifs made of residue classes: the most selective RUN of consecutive single-byte factors on the
    // pattern's spine (literals, [classes], '.'): every match contains, at consecutive positions, one byte of each of the run's
    // sets.  `[LIVM]-x-[DE]-A-[ST]` has no literal worth searching for, but its run occurs at one position in 10^4.
    // (A literal is a run of one-byte sets.)  Empty: nothing usable.
    const std::vector<std::array<uint64_t, 4>>& required_run() const { return run_; }

  private:
    struct Inst { uint8_t op; uint32_t x, y; };  // Char: x = set index, y unused; Split: x preferred over y; Jmp: x
    enum : uint8_t { kChar, kSplit, kJmp, kBegin, kEnd, kMatch };
    static constexpr uint32_t kUnknown = 0xFFFFFFFFu, kDead = 0xFFFFFFFEu;
    struct Prog {
        std::vector<Inst> inst;
        uint32_t start = 0;       // anchored entry
        uint32_t unanchored = 0;  // entry behind a leading any-byte loop
    };
    Semantics semantics_;
    bool has_begin_ = false;                     // the pattern has a `^`
    std::string literal_;                        // a string every match contains (may be empty)
    std::vector<std::array<uint64_t, 4>> run_;   // the byte sets of the most selective run of single-byte factors (may be empty)
    std::vector<std::array<uint64_t, 4>> sets_;  // byte sets of the pattern
    std::array<uint8_t, 256> class_of_{};        // byte -> equivalence class
    std::vector<std::vector<uint8_t>> set_has_class_;  // [set][class]
    uint32_t n_classes_ = 1;
    Prog fwd_, rev_;

    void match_starts(std::string_view text, Cache& cache) const;
    // end of the match that starts at `start` (at_begin: `^` holds there); *matched = is there one at all
    size_t match_end(std::string_view text, size_t start, Cache& cache, bool at_begin, bool* matched = nullptr) const;
    size_t pike_end(std::string_view text, size_t start, Cache& cache, bool at_begin, bool* matched) const;
    // DFA plumbing
    void dfa_init(const Prog& p, bool unanchored, Cache::Dfa& d, Cache& c) const;
    uint32_t dfa_state(const Prog& p, Cache::Dfa& d, std::vector<uint32_t>& seeds, bool at_begin, Cache& c) const;
    uint32_t dfa_step(const Prog& p, Cache::Dfa& d, uint32_t state, uint32_t cls, Cache& c) const;
    void closure(const Prog& p, const std::vector<uint32_t>& seeds, bool at_begin, bool at_end, std::vector<uint32_t>& out, Cache& c) const;
};

}  // namespace tetrex

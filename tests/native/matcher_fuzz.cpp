// Differential test and throughput of the verification matcher (host/matcher.hpp) — native, no GPU, no Python:
//   matcher_fuzz fuzz <seed> <cases>   random patterns of the supported grammar x random texts, every match of the
//                                      FindAndConsume loop against std::regex: leftmost-first against ECMAScript
//                                      regex_search; leftmost-longest against a brute-force oracle made of
//                                      std::regex_match calls (leftmost start that matches at all, then the longest
//                                      end: libstdc++'s own "extended" search is not reliably longest).  Loops whose
//                                      body can match the empty string are not generated: there ECMAScript's
//                                      empty-iteration rule and RE2's thread order legitimately differ.
//                                      Prints mismatches, exit 1 if any
//   matcher_fuzz speed <MB>            MB/s per thread on random protein text (Swissprot-shaped bins) per motif
// std::regex is the differential partner only: it backtracks (recursion depth grows with the input), which is why the
// product does not use it.
//   g++ -O2 -std=c++20 -o /tmp/matcher_fuzz tests/native/matcher_fuzz.cpp tetrex_amd/csrc/host/matcher.cpp
#include "../../tetrex_amd/csrc/host/matcher.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <regex>
#include <string>
#include <vector>

using tetrex::Matcher;

static std::mt19937_64 rng;
static int pick(int n) { return (int)(rng() % (uint64_t)n); }

struct Gen { std::string s; bool nullable; };

static Gen gen(const std::string& alphabet, int depth) {
    const int r = pick(depth > 0 ? 10 : 4);
    auto letter = [&]() { return std::string(1, alphabet[pick((int)alphabet.size())]); };
    switch (r) {
        case 0: case 1: return {letter(), false};
        case 2: return {".", false};
        case 3: {
            std::string s = pick(4) == 0 ? "[^" : "[";
            const int n = 1 + pick(3);
            for (int i = 0; i < n; ++i) s += letter();
            return {s + "]", false};
        }
        case 4: case 5: { const Gen a = gen(alphabet, depth - 1), b = gen(alphabet, depth - 1); return {a.s + b.s, a.nullable && b.nullable}; }
        case 6: { const Gen a = gen(alphabet, depth - 1), b = gen(alphabet, depth - 1); return {"(" + a.s + "|" + b.s + ")", a.nullable || b.nullable}; }
        case 7: {
            const char op = "*+?"[pick(3)];
            Gen a = gen(alphabet, depth - 1);
            while (op != '?' && a.nullable) a = gen(alphabet, depth - 1);  // no loop over a body that can match nothing
            return {"(" + a.s + ")" + op, op != '+'};
        }
        case 8: {
            const int lo = pick(3), hi = lo + pick(3);
            Gen a = gen(alphabet, depth - 1);
            while (a.nullable) a = gen(alphabet, depth - 1);
            char buf[32];
            const bool exact = pick(3) == 0;
            if (exact) std::snprintf(buf, sizeof buf, "{%d}", lo + 1);
            else std::snprintf(buf, sizeof buf, "{%d,%d}", lo, hi + (hi == 0));
            return {"(" + a.s + ")" + buf, !exact && lo == 0};
        }
        default: { const Gen a = gen(alphabet, depth - 1), b = gen(alphabet, depth - 1), c = gen(alphabet, depth - 1); return {a.s + b.s + c.s, a.nullable && b.nullable && c.nullable}; }
    }
}

// POSIX oracle: the leftmost position where anything matches, and the longest match from there; then on from its end
static std::vector<std::pair<size_t, size_t>> longest_by_brute_force(const std::string& text, const std::regex& rx) {
    // (FindAndConsume searches the REMAINING text as if it were the whole text: `^` holds again where a non-empty match ended)
    std::vector<std::pair<size_t, size_t>> out;
    size_t pos = 0;
    bool fresh = false;
    while (pos <= text.size()) {
        bool found = false;
        for (size_t s = pos; s <= text.size() && !found; ++s)
            for (size_t e = text.size() + 1; e-- > s;)
                if (std::regex_match(text.begin() + (std::ptrdiff_t)s, text.begin() + (std::ptrdiff_t)e, rx,
                                     (s && !(fresh && s == pos) ? std::regex_constants::match_not_bol | std::regex_constants::match_prev_avail : std::regex_constants::match_default) |
                                         (e < text.size() ? std::regex_constants::match_not_eol : std::regex_constants::match_default))) {
                    out.emplace_back(s, e - s);
                    fresh = e > s;
                    pos = e > s ? e : s + 1;
                    found = true;
                    break;
                }
        if (!found) break;
    }
    return out;
}

static std::vector<std::pair<size_t, size_t>> with_std(const std::string& text, const std::regex& rx) {
    std::vector<std::pair<size_t, size_t>> out;
    size_t pos = 0;
    bool fresh = false;  // pos is where a non-empty match ended: the beginning of the text as far as the next search is concerned
    std::smatch m;
    while (pos <= text.size()) {
        if (!std::regex_search(text.begin() + (std::ptrdiff_t)pos, text.end(), m, rx,
                               pos && !fresh ? std::regex_constants::match_prev_avail : std::regex_constants::match_default)) break;
        const size_t start = pos + (size_t)m.position(0), len = (size_t)m.length(0);
        out.emplace_back(start, len);
        fresh = len != 0;
        pos = start + (len ? len : 1);
    }
    return out;
}

int main(int argc, char** argv) {
    if (argc >= 3 && !std::strcmp(argv[1], "speed")) {
        const size_t bytes = (size_t)std::atoll(argv[2]) << 20;
        rng.seed(7);
        const char* aa = "ACDEFGHIKLMNPQRSTVWY";
        std::string text(bytes, 'A');
        for (char& c : text) c = aa[pick(20)];
        for (size_t at = 1000; at + 8 < bytes; at += bytes / 7) std::memcpy(&text[at], "LMAEGLYN", 8);
        for (const char* rx : {"(LMA(E|Q)GLYN)", "(A.C.E.GH)", "(W.{2}[LIVM]D[VFY][LIVM]{3}D.PPGT[GS]D)", "(C.{2,4}C.{3}[LIVMFYWC].{8}H.{3,5}H)", "(K[RK]{2,3}DE)", "(N[^P][ST][^P])"}) {
            const Matcher m(rx, Matcher::Semantics::LeftmostLongest);
            Matcher::Cache cache;
            size_t n = 0;
            m.find_all(std::string_view(text).substr(0, 1 << 20), cache, [&](size_t, size_t) {});  // build the automaton
            const auto t0 = std::chrono::steady_clock::now();
            // records of ~360 residues, like Swissprot entries
            for (size_t at = 0; at < bytes; at += 360) m.find_all(std::string_view(text).substr(at, std::min<size_t>(360, bytes - at)), cache, [&](size_t, size_t) { ++n; });
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("%-48s %8.1f MB/s  %zu matches\n", rx, bytes / dt / 1e6, n);
        }
        return 0;
    }
    const uint64_t seed = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1;
    const int cases = argc > 3 ? std::atoi(argv[3]) : 2000;
    rng.seed(seed);
    int bad = 0, compared = 0;
    for (int c = 0; c < cases && bad < 10; ++c) {
        const bool dna = pick(2) == 0;
        const std::string alphabet = dna ? "ACGT" : "ACDEKL";
        std::string pattern = gen(alphabet, 3).s;
        if (pick(8) == 0) pattern = "^" + pattern;
        if (pick(8) == 0) pattern += "$";
        for (int posix = 0; posix < 2; ++posix) {
            std::regex rx;
            try {
                rx = std::regex("(" + pattern + ")", std::regex::ECMAScript);
            } catch (const std::regex_error&) { continue; }
            const Matcher m("(" + pattern + ")", posix ? Matcher::Semantics::LeftmostLongest : Matcher::Semantics::LeftmostFirst);
            Matcher::Cache cache;
            for (int t = 0; t < 6; ++t) {
                std::string text(pick(posix ? 22 : 40), 'A');  // the POSIX oracle is O(n^2) regex_match calls
                for (char& ch : text) ch = alphabet[pick((int)alphabet.size())];
                std::vector<std::pair<size_t, size_t>> got;
                m.find_all(text, cache, [&](size_t s, size_t n) { got.emplace_back(s, n); });
                const auto want = posix ? longest_by_brute_force(text, rx) : with_std(text, rx);
                ++compared;
                if (got != want) {
                    ++bad;
                    std::printf("MISMATCH %s pattern (%s) text %s\n  matcher:", posix ? "posix" : "perl", pattern.c_str(), text.c_str());
                    for (auto& g : got) std::printf(" [%zu,+%zu]", g.first, g.second);
                    std::printf("\n  std::regex:");
                    for (auto& g : want) std::printf(" [%zu,+%zu]", g.first, g.second);
                    std::printf("\n");
                }
            }
        }
    }
    std::printf("%d comparisons, %d mismatches\n", compared, bad);
    return bad ? 1 : 0;
}

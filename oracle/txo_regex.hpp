// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see oracle/txo_ibf.hpp header).
// CPU oracle: regex front-end, restating
//   include/utils.h:96-310   RegexLexer (tokens, [..] classes, {m}/{m,n} quantifiers, escapes)
//   include/utils.h:312-468  PostfixConverter (explicit concat '-', shunting-yard)
//   src/utils.cpp:3-15       translate() (lexer errors are printed, result is "")
//   src/query.cpp:78-155     matchUninformative / trimRegEx / reduce_query_alphabet
//   include/query.h:80-94    preprocess_query
// Pinned by the translate() golden vectors of SURVEY.md §8c (tests/golden/translate.json).
#pragma once
#include <string>
#include <vector>
#include <algorithm>
#include <stdexcept>
#include <cctype>

namespace txo {

enum class Tk { Chr, Dot, Star, Plus, Quest, Pipe, LPar, RPar, Quant, Exact, Class, Cat, End };

struct Tok {
    Tk t;
    std::string text;
    int lo = 0, hi = 0;
    bool neg = false;
    std::vector<char> set;
};

inline std::vector<Tok> lex(const std::string& in) {
    std::vector<Tok> out;
    size_t p = 0;
    auto number = [&]() {
        int n = 0;
        while (p < in.size() && std::isdigit((unsigned char)in[p])) n = n * 10 + (in[p++] - '0');
        return n;
    };
    while (p < in.size()) {
        char c = in[p];
        switch (c) {
            case '.': out.push_back({Tk::Dot, "."}); ++p; break;
            case '*': out.push_back({Tk::Star, "*"}); ++p; break;
            case '+': out.push_back({Tk::Plus, "+"}); ++p; break;
            case '?': out.push_back({Tk::Quest, "?"}); ++p; break;
            case '|': out.push_back({Tk::Pipe, "|"}); ++p; break;
            case '(': out.push_back({Tk::LPar, "("}); ++p; break;
            case ')': out.push_back({Tk::RPar, ")"}); ++p; break;
            case '[': {
                ++p;
                if (p >= in.size()) throw std::runtime_error("Invalid character class: unexpected end of input");
                Tok t{Tk::Class, ""};
                if (in[p] == '^') { t.neg = true; ++p; }
                while (p < in.size() && in[p] != ']') {
                    char cur = in[p];
                    if (cur == '\\') {
                        ++p;
                        if (p >= in.size()) throw std::runtime_error("Invalid escape in character class");
                        char e = in[p];
                        if (e == 'n') e = '\n'; else if (e == 't') e = '\t'; else if (e == 'r') e = '\r';
                        t.set.push_back(e);
                        ++p;
                    } else { t.set.push_back(cur); ++p; }
                }
                if (p >= in.size() || in[p] != ']') throw std::runtime_error("Invalid character class: missing closing ']'");
                ++p;
                if (t.set.empty()) throw std::runtime_error("Empty character class");
                out.push_back(t);
                break;
            }
            case '{': {
                ++p;
                if (p >= in.size() || !std::isdigit((unsigned char)in[p])) throw std::runtime_error("Invalid quantifier: expected number after '{'");
                int lo = number();
                if (p >= in.size()) throw std::runtime_error("Invalid quantifier: unexpected end of input");
                if (in[p] == '}') { ++p; Tok t{Tk::Exact, ""}; t.lo = t.hi = lo; out.push_back(t); }
                else if (in[p] == ',') {
                    ++p;
                    if (p >= in.size()) throw std::runtime_error("Invalid quantifier: unexpected end after ','");
                    if (in[p] == '}') throw std::runtime_error("Open-ended quantifiers {m,} not supported");
                    if (!std::isdigit((unsigned char)in[p])) throw std::runtime_error("Invalid quantifier: expected number after ','");
                    int hi = number();
                    if (p >= in.size() || in[p] != '}') throw std::runtime_error("Invalid quantifier: expected '}' after max value");
                    ++p;
                    if (lo > hi) throw std::runtime_error("Invalid quantifier: min > max");
                    Tok t{Tk::Quant, ""}; t.lo = lo; t.hi = hi; out.push_back(t);
                } else throw std::runtime_error("Invalid quantifier: expected ',' or '}' after min value");
                break;
            }
            case '\\':
                ++p;
                if (p >= in.size()) throw std::runtime_error("Invalid escape: end of input after '\\'");
                out.push_back({Tk::Chr, std::string(1, in[p])}); ++p; break;
            default: out.push_back({Tk::Chr, std::string(1, c)}); ++p; break;
        }
    }
    out.push_back({Tk::End, ""});
    return out;
}

inline bool is_atom(Tk t) { return t == Tk::Chr || t == Tk::Dot || t == Tk::Class; }
inline bool is_rep(Tk t) { return t == Tk::Star || t == Tk::Plus || t == Tk::Quest || t == Tk::Quant || t == Tk::Exact; }
inline int prec(Tk t) { return t == Tk::Pipe ? 1 : t == Tk::Cat ? 2 : is_rep(t) ? 3 : 0; }

// utils.h:360-408 tokenToPostfix.  Unions are emitted as "ab|c|d|": first letter, then
// (letter '|') pairs.  '.' is the fixed 20-way amino-acid union of utils.h:365.
inline std::string emit(const Tok& t) {
    static const char aa20[] = "ACDEFGHIKLMNPQRSTVWY";  // utils.h:58-79 (sorted)
    switch (t.t) {
        case Tk::Chr: return t.text;
        case Tk::Dot: return "FQ|L|T|K|P|A|Y|R|N|H|G|E|C|I|V|D|W|S|M|";
        case Tk::Class: {
            std::vector<char> m = t.set;
            if (t.neg) {
                std::sort(m.begin(), m.end());
                std::vector<char> d;
                std::set_difference(aa20, aa20 + 20, m.begin(), m.end(), std::back_inserter(d));
                m = d;
            }
            std::string r(1, m.at(0));  // reference indexes [0] unchecked; .at() makes the empty case loud
            for (size_t i = 1; i < m.size(); ++i) { r += m[i]; r += '|'; }
            return r;
        }
        case Tk::Star: return "*";
        case Tk::Plus: return "+";
        case Tk::Quest: return "?";
        case Tk::Pipe: return "|";
        case Tk::Cat: return "-";
        case Tk::Exact: return "{" + std::to_string(t.lo) + "}";
        case Tk::Quant: return "{" + std::to_string(t.lo) + "," + std::to_string(t.hi) + "}";
        default: return "";
    }
}

// infixToPostfix; throws on lexer errors (translate() below swallows them like src/utils.cpp).
inline std::string to_postfix(const std::string& rx) {
    std::vector<Tok> toks = lex(rx), seq;
    for (size_t i = 0; i < toks.size(); ++i) {
        if (i > 0) {
            Tk pv = toks[i - 1].t, cu = toks[i].t;
            bool after = is_atom(pv) || pv == Tk::RPar || is_rep(pv);
            bool before = is_atom(cu) || cu == Tk::LPar;
            if (after && before) seq.push_back({Tk::Cat, ""});
        }
        seq.push_back(toks[i]);
    }
    std::string out;
    std::vector<Tok> ops;
    for (const Tok& t : seq) {
        if (is_atom(t.t)) out += emit(t);
        else if (t.t == Tk::LPar) ops.push_back(t);
        else if (t.t == Tk::RPar) {
            while (!ops.empty() && ops.back().t != Tk::LPar) { out += emit(ops.back()); ops.pop_back(); }
            if (!ops.empty()) ops.pop_back();
        } else if (t.t == Tk::End) break;
        else {  // operator
            while (!ops.empty() && ops.back().t != Tk::LPar && prec(ops.back().t) >= prec(t.t)) {
                out += emit(ops.back()); ops.pop_back();
            }
            ops.push_back(t);
        }
    }
    while (!ops.empty()) { out += emit(ops.back()); ops.pop_back(); }
    return out;
}

inline std::string translate(const std::string& rx) {
    try { return to_postfix(rx); } catch (const std::exception&) { return std::string(); }
}

// src/query.cpp:78-120
inline size_t match_uninformative(const std::string& s, size_t pos) {
    if (s[pos] == '^' || s[pos] == '$') return 1;
    if (s[pos] == '.') {
        if (pos + 1 < s.size() && (s[pos + 1] == '*' || s[pos + 1] == '+')) return 2;
        if (pos + 1 < s.size() && s[pos + 1] == '{') {
            size_t e = s.find('}', pos + 2);
            if (e != std::string::npos) return e - pos + 1;
        }
        return 1;
    }
    if (s[pos] == '[') {
        size_t e = s.find(']', pos + 1);
        if (e != std::string::npos) {
            std::string in = s.substr(pos + 1, e - pos - 1);
            if (!in.empty() && (in[0] == '^' || in == ".")) return e - pos + 1;
            if (in.find('-') != std::string::npos) return e - pos + 1;
        }
    }
    return 0;
}

// src/query.cpp:122-141.  Note the back trim probes the token test at the LAST character
// (end-1), exactly as the reference does, so only single-character tails ('$', '.') match.
inline std::string trim_regex(const std::string& rx) {
    size_t b = 0, e = rx.size();
    while (b < e) { size_t n = match_uninformative(rx, b); if (!n) break; b += n; }
    while (e > b) { size_t n = match_uninformative(rx, e - 1); if (!n) break; e -= n; }
    return rx.substr(b, e - b);
}

// src/query.cpp:145-155
inline std::string reduce_alphabet(std::string rx, const std::array<char, 256>& redmap) {
    for (char& c : rx) if (std::isalpha((unsigned char)c)) c = redmap[(unsigned char)c];
    return rx;
}

}  // namespace txo

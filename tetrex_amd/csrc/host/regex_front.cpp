#include "regex_front.hpp"
#include "encoder.hpp"

#include <algorithm>
#include <cctype>
#include <stdexcept>
#include <vector>

namespace tetrex {
namespace {

enum Kind : uint8_t { Literal, AnyChar, CharSet, Open, Close, Alternate, Concat, Repeat, Stop };

// what a token contributes to the postfix string: text[at, at + len) of the scanner's pool (one string for all tokens:
// a Lexeme with a std::string of its own cost an allocation per wildcard and per quantifier)
struct Lexeme {
    Kind kind;
    uint32_t at = 0, len = 0;
};

constexpr char kAmino[] = "ACDEFGHIKLMNPQRSTVWY";                          // include/utils.h:58-79
constexpr char kAnyUnion[] = "FQ|L|T|K|P|A|Y|R|N|H|G|E|C|I|V|D|W|S|M|";    // include/utils.h:365

class Scanner {
  public:
    explicit Scanner(const std::string& src) : s_(src) {
        pool_.reserve(src.size() * 2 + sizeof kAnyUnion + 16);
        pool_.append(kAnyUnion);  // [0, 39): every wildcard points here
        pool_.append("|-");       // [39], [40]
    }
    const std::string& pool() const { return pool_; }
    static Lexeme concat() { return {Concat, (uint32_t)sizeof kAnyUnion, 1}; }

    std::vector<Lexeme> run() {
        std::vector<Lexeme> out;
        out.reserve(s_.size() + 1);
        while (at_ < s_.size()) {
            const char c = s_[at_];
            switch (c) {
                case '.': out.push_back({AnyChar, 0, (uint32_t)sizeof kAnyUnion - 1}); ++at_; break;
                case '*': case '+': case '?': out.push_back(one(Repeat, c)); ++at_; break;
                case '|': out.push_back({Alternate, (uint32_t)sizeof kAnyUnion - 1, 1}); ++at_; break;
                case '(': out.push_back({Open, 0, 0}); ++at_; break;
                case ')': out.push_back({Close, 0, 0}); ++at_; break;
                case '[': out.push_back(char_set()); break;
                case '{': out.push_back(counted()); break;
                case '\\':
                    if (++at_ >= s_.size()) throw std::runtime_error("Invalid escape: end of input after '\\'");
                    out.push_back(one(Literal, s_[at_++]));
                    break;
                default: out.push_back(one(Literal, c)); ++at_; break;
            }
        }
        out.push_back({Stop, 0, 0});
        return out;
    }

  private:
    const std::string& s_;
    std::string pool_;
    size_t at_ = 0;

    Lexeme one(Kind kind, char c) {
        pool_.push_back(c);
        return {kind, (uint32_t)pool_.size() - 1, 1};
    }
    Lexeme text(Kind kind, const std::string& t) {
        pool_.append(t);
        return {kind, (uint32_t)(pool_.size() - t.size()), (uint32_t)t.size()};
    }
    bool digit() const { return at_ < s_.size() && std::isdigit((unsigned char)s_[at_]); }
    int number() {
        int v = 0;
        while (digit()) v = v * 10 + (s_[at_++] - '0');
        return v;
    }

    Lexeme char_set() {
        ++at_;  // '['
        if (at_ >= s_.size()) throw std::runtime_error("Invalid character class: unexpected end of input");
        bool negated = false;
        if (s_[at_] == '^') { negated = true; ++at_; }
        std::vector<char> members;
        while (at_ < s_.size() && s_[at_] != ']') {
            char c = s_[at_];
            if (c == '\\') {
                if (++at_ >= s_.size()) throw std::runtime_error("Invalid escape in character class");
                c = s_[at_];
                if (c == 'n') c = '\n';
                else if (c == 't') c = '\t';
                else if (c == 'r') c = '\r';
            }
            members.push_back(c);
            ++at_;
        }
        if (at_ >= s_.size()) throw std::runtime_error("Invalid character class: missing closing ']'");
        ++at_;  // ']'
        if (members.empty()) throw std::runtime_error("Empty character class");
        if (negated) {
            std::sort(members.begin(), members.end());
            std::vector<char> rest;
            std::set_difference(kAmino, kAmino + 20, members.begin(), members.end(), std::back_inserter(rest));
            if (rest.empty()) throw std::runtime_error("Negated character class excludes every residue");
            members.swap(rest);
        }
        // the union of the members: "AB|C|..."
        const uint32_t at = (uint32_t)pool_.size();
        pool_.push_back(members[0]);
        for (size_t i = 1; i < members.size(); ++i) {
            pool_.push_back(members[i]);
            pool_.push_back('|');
        }
        return {CharSet, at, (uint32_t)pool_.size() - at};
    }

    Lexeme counted() {
        ++at_;  // '{'
        if (!digit()) throw std::runtime_error("Invalid quantifier: expected number after '{'");
        const int lo = number();
        if (at_ >= s_.size()) throw std::runtime_error("Invalid quantifier: unexpected end of input");
        if (s_[at_] == '}') {
            ++at_;
            return text(Repeat, "{" + std::to_string(lo) + "}");
        }
        if (s_[at_] != ',') throw std::runtime_error("Invalid quantifier: expected ',' or '}' after min value");
        if (++at_ >= s_.size()) throw std::runtime_error("Invalid quantifier: unexpected end after ','");
        if (s_[at_] == '}') throw std::runtime_error("Open-ended quantifiers {m,} not supported");
        if (!digit()) throw std::runtime_error("Invalid quantifier: expected number after ','");
        const int hi = number();
        if (at_ >= s_.size() || s_[at_] != '}') throw std::runtime_error("Invalid quantifier: expected '}' after max value");
        ++at_;
        if (lo > hi) throw std::runtime_error("Invalid quantifier: min > max");
        return text(Repeat, "{" + std::to_string(lo) + "," + std::to_string(hi) + "}");
    }
};

inline bool operand(Kind k) { return k == Literal || k == AnyChar || k == CharSet; }
inline int binding(Kind k) { return k == Alternate ? 1 : k == Concat ? 2 : k == Repeat ? 3 : 0; }

}  // namespace

std::string regex_to_postfix(const std::string& regex) {
    Scanner scanner(regex);
    const std::vector<Lexeme> toks = scanner.run();
    const std::string& pool = scanner.pool();
    std::string out;
    out.reserve(pool.size());
    auto write = [&](const Lexeme& t) { out.append(pool, t.at, t.len); };
    std::vector<Lexeme> pending;  // operator stack
    pending.reserve(16);
    auto apply = [&](const Lexeme& op) {
        while (!pending.empty() && pending.back().kind != Open && binding(pending.back().kind) >= binding(op.kind)) {
            write(pending.back());
            pending.pop_back();
        }
        pending.push_back(op);
    };
    for (size_t i = 0; i < toks.size(); ++i) {
        const Lexeme& t = toks[i];
        if (i > 0) {  // implicit concatenation between an operand/')'/repeat and an operand/'('
            const Kind prev = toks[i - 1].kind;
            if ((operand(prev) || prev == Close || prev == Repeat) && (operand(t.kind) || t.kind == Open)) apply(Scanner::concat());
        }
        if (operand(t.kind)) write(t);
        else if (t.kind == Open) pending.push_back(t);
        else if (t.kind == Close) {
            while (!pending.empty() && pending.back().kind != Open) {
                write(pending.back());
                pending.pop_back();
            }
            if (!pending.empty()) pending.pop_back();
        } else if (t.kind == Stop) break;
        else apply(t);
    }
    while (!pending.empty()) {
        write(pending.back());  // an unmatched '(' contributes nothing, as in the reference
        pending.pop_back();
    }
    return out;
}

std::string translate(const std::string& regex, std::string* error) {
    try {
        return regex_to_postfix(regex);
    } catch (const std::exception& e) {
        if (error) *error = e.what();
        return std::string();
    }
}

namespace {
// length of an "uninformative" token starting at pos, or 0 (src/query.cpp:78-120)
size_t uninformative_at(const std::string& s, size_t pos) {
    const char c = s[pos];
    if (c == '^' || c == '$') return 1;
    if (c == '.') {
        if (pos + 1 < s.size()) {
            const char n = s[pos + 1];
            if (n == '*' || n == '+') return 2;
            if (n == '{') {
                const size_t close = s.find('}', pos + 2);
                if (close != std::string::npos) return close - pos + 1;
            }
        }
        return 1;
    }
    if (c == '[') {
        const size_t close = s.find(']', pos + 1);
        if (close != std::string::npos) {
            const std::string body = s.substr(pos + 1, close - pos - 1);
            if ((!body.empty() && (body[0] == '^' || body == ".")) || body.find('-') != std::string::npos) return close - pos + 1;
        }
    }
    return 0;
}
}  // namespace

std::string trim_uninformative(const std::string& regex) {
    size_t lo = 0, hi = regex.size();
    for (size_t n; lo < hi && (n = uninformative_at(regex, lo)) != 0;) lo += n;
    // the reference tests the LAST character of the remaining text (src/query.cpp:134-138)
    for (size_t n; hi > lo && (n = uninformative_at(regex, hi - 1)) != 0;) hi -= std::min(n, hi - lo);
    return regex.substr(lo, hi - lo);
}

std::string reduce_query_alphabet(const std::string& regex, const std::array<char, 256>& table) {
    std::string out = regex;
    for (char& c : out)
        if (std::isalpha((unsigned char)c)) c = table[(unsigned char)c];
    return out;
}

std::string preprocess_query(const std::string& regex, const KmerEncoder& enc, std::string* preprocessed) {
    std::string rx = regex;
    if (enc.molecule() == Molecule::Peptide) {
        if (enc.alphabet() != Alphabet::Base) rx = reduce_query_alphabet(rx, enc.reduce_table());
        rx = trim_uninformative(rx);
    }
    if (preprocessed) *preprocessed = rx;
    return translate(rx);
}

}  // namespace tetrex

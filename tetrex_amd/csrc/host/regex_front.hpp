// Host side: regex front-end of `tetrex query` (product code).
// Same observable behaviour as the reference's
//   translate()              src/utils.cpp:3-15  (RegexLexer + PostfixConverter, include/utils.h:96-468)
//   trimRegEx()              src/query.cpp:122-141
//   reduce_query_alphabet()  src/query.cpp:145-155
//   preprocess_query()       include/query.h:80-94
// The postfix text is the interface between this stage and the k-graph builder, exactly as in
// the reference ('-' = concatenation, unions as "ab|c|", quantifiers as "{m}" / "{m,n}").
#pragma once
#include <array>
#include <string>

namespace tetrex {

class KmerEncoder;

// infix -> postfix; lexer errors throw std::runtime_error with the reference's messages
std::string regex_to_postfix(const std::string& regex);
// translate(): like the reference, a lexer error yields "" (the message goes to `error` if given)
std::string translate(const std::string& regex, std::string* error = nullptr);
std::string trim_uninformative(const std::string& regex);
std::string reduce_query_alphabet(const std::string& regex, const std::array<char, 256>& table);
// preprocess_query for the index's molecule/alphabet; `preprocessed` receives the regex that
// was translated (reduced + trimmed for peptides)
std::string preprocess_query(const std::string& regex, const KmerEncoder& enc, std::string* preprocessed = nullptr);

}  // namespace tetrex

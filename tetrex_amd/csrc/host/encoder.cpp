#include "encoder.hpp"

namespace tetrex {

namespace {
// residue -> 5-bit code, letters A..Z (include/peptide_decomposer.h:59-151)
constexpr uint8_t kBaseCodes[26] = {0, 2, 1, 2, 3, 4, 5, 6, 7, 9, 8, 9, 10, 11, 20, 12, 13, 14, 15, 16, 20, 17, 18, 20, 19, 3};
constexpr uint8_t kMurphyCodes[26] = {0, 1, 2, 1, 1, 3, 4, 5, 6, 6, 7, 6, 6, 1, 7, 8, 1, 7, 9, 9, 2, 6, 3, 9, 3, 1};
constexpr uint8_t kLiCodes[26] = {0, 1, 2, 1, 1, 3, 4, 5, 6, 7, 8, 7, 7, 5, 8, 9, 1, 8, 0, 0, 2, 6, 3, 0, 3, 1};
// residue -> representative letter of its reduced class, letters A..Z
// (include/peptide_decomposer.h:155-213; the non-Murphy branch is the Li table)
constexpr char kMurphyLetters[27] = "ABCBBFGHIIKIIBKPBKSSCIFSFB";
constexpr char kLiLetters[27] = "ABCBBFGHIJKJJHKPBKAACIFAFB";
}  // namespace

KmerEncoder::KmerEncoder(Molecule mol, unsigned k, Alphabet alphabet) : mol_(mol), alphabet_(alphabet), k_(k) {
    bits_ = mol == Molecule::DNA ? 2u : 5u;
    const unsigned total = bits_ * k_;
    kmer_mask_ = (k_ >= 32 || total >= 64) ? ~0ULL : ((1ULL << total) - 1ULL);
    const unsigned sfx = k_ ? bits_ * (k_ - 1) : 0;
    suffix_mask_ = sfx >= 64 ? ~0ULL : ((1ULL << sfx) - 1ULL);
    if (mol == Molecule::Peptide) {
        const uint8_t* codes = alphabet == Alphabet::Murphy ? kMurphyCodes : alphabet == Alphabet::Li ? kLiCodes : kBaseCodes;
        const char* letters = alphabet == Alphabet::Murphy ? kMurphyLetters : kLiLetters;
        for (int i = 0; i < 26; ++i) {
            aa_code_[(unsigned)('A' + i)] = codes[i];
            reduce_[(unsigned)('A' + i)] = letters[i];
        }
    }
}

uint64_t KmerEncoder::canonical(uint64_t forward) const {
    // reverse complement under A0 C1 T2 G3: complement = code ^ 2
    uint64_t rc = 0, f = forward;
    for (unsigned i = 0; i < k_; ++i) {
        rc = (rc << 2) | ((f & 3u) ^ 2u);
        f >>= 2;
    }
    return forward <= rc ? forward : rc;
}

void KmerEncoder::record_values(std::string_view seq, bool wraparound, std::vector<uint64_t>& out) const {
    if (seq.size() < k_ || k_ == 0) return;
    uint64_t fwd = 0;
    for (unsigned i = 0; i < k_; ++i) fwd = (fwd << bits_) | code((unsigned char)seq[i]);
    if (mol_ == Molecule::Peptide) {
        out.push_back(fwd);
        for (size_t i = k_; i < seq.size(); ++i) {
            fwd = ((fwd << 5) & kmer_mask_) | aa_code_[(unsigned char)seq[i]];
            out.push_back(fwd);
        }
        return;
    }
    out.push_back(canonical(fwd));
    for (size_t i = wraparound ? 0 : k_; i < seq.size(); ++i) {
        fwd = ((fwd << 2) & kmer_mask_) | (((unsigned char)seq[i] >> 1) & 3u);
        out.push_back(canonical(fwd));
    }
}

}  // namespace tetrex

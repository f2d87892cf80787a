// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see oracle/txo_ibf.hpp header).
// CPU oracle: k-mer encoders, restating
//   include/nucleotide_decomposer.h:44-121  (2-bit DNA, canonical k-mers, record decomposition)
//   include/peptide_decomposer.h:59-213,262-299 (5-bit residues, Murphy/Li reductions)
//   include/molecule_decomposer.h:76-90     (lshift_/rmask_ = 2/3 for na, 5/31 for aa)
// Pinned by golden vectors captured from those headers in
// tests/golden/encoders.json and by the fixture test/data/ibf_idx.ibf.
#pragma once
#include <cstdint>
#include <string>
#include <string_view>
#include <vector>
#include <array>

namespace txo {

enum Reduction : uint8_t { kBase = 0, kMurphy = 1, kLi = 2 };

// reverse complement of a 2-bit packed k-mer under the reference's code A0 C1 T2 G3
// (complement = code ^ 2).  nucleotide_decomposer.h:44-79 does this with pshufb
// nibble look-ups; the scalar loop below is the same function.
inline uint64_t dna_revcomp(uint64_t kmer, unsigned k) {
    uint64_t r = 0;
    for (unsigned i = 0; i < k; ++i) {
        r = (r << 2) | ((kmer & 3) ^ 2);
        kmer >>= 2;
    }
    return r;
}

struct Encoder {
    bool dna = true;
    uint8_t k = 0;
    uint8_t reduction = kBase;
    uint8_t lshift = 2;   // bits per symbol
    uint8_t rmask = 3;    // symbol mask
    uint64_t selection_mask = 0;
    std::array<uint8_t, 256> aamap{};  // peptide_decomposer.h:59-151
    std::array<char, 256> redmap{};    // peptide_decomposer.h:155-213 (zero for unset entries)

    Encoder() = default;
    Encoder(bool is_dna, uint8_t ksize, uint8_t red) : dna(is_dna), k(ksize), reduction(red) {
        lshift = dna ? 2 : 5;
        rmask = dna ? 3 : 31;
        unsigned bits = lshift * k;
        // nucleotide_decomposer.h:36 / peptide_decomposer.h:49: (k >= 32) ? ~0 : (1<<bits)-1
        selection_mask = (k >= 32) ? ~0ULL : ((bits >= 64) ? ~0ULL : ((1ULL << bits) - 1ULL));
        if (!dna) { fill_aamap(); fill_redmap(); }
    }

    uint8_t code(int symbol) const {
        return dna ? (uint8_t)((symbol >> 1) & 3) : aamap[(uint8_t)symbol];
    }

    // update_kmer: rolls `symbol` into `fwd` (stored back) and returns the value probed
    // (canonical min(fwd, revcomp) for DNA, fwd for peptides).
    uint64_t update_kmer(int symbol, uint64_t& fwd) const {
        uint64_t f = ((fwd << lshift) & selection_mask) | code(symbol);
        fwd = f;
        if (!dna) return f;
        uint64_t r = dna_revcomp(f, k);
        return f <= r ? f : r;
    }

    // decompose_record: the values emplaced for one sequence record, in order.
    // quirk=true restates nucleotide_decomposer.h:100-110 literally: the first k-mer is
    // emplaced, then ALL symbols (including the first k again) are rolled over it, which
    // inserts k-1 wrap-around k-mers plus the first k-mer a second time.  quirk=false is
    // the intended decomposition (what the legacy fixture test/data/ibf_idx.ibf contains).
    // The peptide path (peptide_decomposer.h:283-290) has no such quirk.
    std::vector<uint64_t> decompose_record(std::string_view seq, bool quirk) const {
        std::vector<uint64_t> out;
        if (seq.size() < k) return out;
        uint64_t fwd = 0;
        for (unsigned i = 0; i < k; ++i) fwd = (fwd << lshift) | code((unsigned char)seq[i]);
        if (!dna) {
            out.push_back(fwd);
            for (size_t i = k; i < seq.size(); ++i) {
                fwd = ((fwd << 5) & selection_mask) | aamap[(uint8_t)seq[i]];
                out.push_back(fwd);
            }
            return out;
        }
        unsigned left_shift = 2u * k - 2u;
        uint64_t rev = dna_revcomp(fwd, k);
        out.push_back(fwd <= rev ? fwd : rev);
        for (size_t i = quirk ? 0 : k; i < seq.size(); ++i) {
            uint64_t fb = ((unsigned char)seq[i] >> 1) & 3;
            uint64_t cb = (fb ^ 2) << left_shift;
            fwd = ((fwd << 2) & selection_mask) | fb;
            rev = ((rev >> 2) & selection_mask) | cb;
            out.push_back(fwd <= rev ? fwd : rev);
        }
        return out;
    }

  private:
    void set(const char* letters, const uint8_t* codes) {
        for (size_t i = 0; letters[i]; ++i) aamap[(uint8_t)letters[i]] = codes[i];
    }
    void fill_aamap() {
        static const char L[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZ";
        static const uint8_t base[26]   = {0, 2, 1, 2, 3, 4, 5, 6, 7, 9, 8, 9, 10, 11, 20, 12, 13, 14, 15, 16, 20, 17, 18, 20, 19, 3};
        static const uint8_t murphy[26] = {0, 1, 2, 1, 1, 3, 4, 5, 6, 6, 7, 6, 6, 1, 7, 8, 1, 7, 9, 9, 2, 6, 3, 9, 3, 1};
        static const uint8_t li[26]     = {0, 1, 2, 1, 1, 3, 4, 5, 6, 7, 8, 7, 7, 5, 8, 9, 1, 8, 0, 0, 2, 6, 3, 0, 3, 1};
        set(L, reduction == kMurphy ? murphy : reduction == kLi ? li : base);
    }
    void fill_redmap() {
        // create_r2r_maps: Murphy table when reduction==Murphy, otherwise the Li table
        // (also for Base — benign, only consulted when reduction>0).
        static const char from[] = "ARNDCYEQGHILKMFPSTWVUOBZJX";
        static const char murphy[] = "AKBBCFBBGHIIKIFPSSFICKBBIS";
        static const char li[]     = "AKHBCFBBGHIJKJFPAAFICKBBJA";
        const char* to = reduction == kMurphy ? murphy : li;
        for (size_t i = 0; from[i]; ++i) redmap[(uint8_t)from[i]] = to[i];
    }
};

}  // namespace txo

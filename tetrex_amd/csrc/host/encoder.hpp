// Host side of the TetRex query path: k-mer encoders (product code; no oracle dependency).
// Mirrors the reference's MoleculeDecomposer interface for this path:
//   NucleotideDecomposer::update_kmer / decompose_record   include/nucleotide_decomposer.h:99-121
//   PeptideDecomposer::update_kmer / decompose_record       include/peptide_decomposer.h:283-299
//   aamap_ / redmap_ tables                                 include/peptide_decomposer.h:59-213
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <string_view>
#include <vector>

namespace tetrex {

enum class Molecule : uint8_t { DNA, Peptide };
enum class Alphabet : uint8_t { Base = 0, Murphy = 1, Li = 2 };

class KmerEncoder {
  public:
    KmerEncoder() = default;
    KmerEncoder(Molecule mol, unsigned k, Alphabet alphabet);

    Molecule molecule() const { return mol_; }
    unsigned k() const { return k_; }
    Alphabet alphabet() const { return alphabet_; }
    unsigned bits_per_symbol() const { return bits_; }
    uint64_t symbol_mask() const { return (1ULL << bits_) - 1ULL; }
    // residue codes are 0 .. alphabet_size()-1 (Base 21: the 20 residues + the catch-all code 20; Murphy / Li 10; DNA 4)
    unsigned alphabet_size() const { return mol_ == Molecule::DNA ? 4u : alphabet_ == Alphabet::Base ? 21u : 10u; }
    uint64_t kmer_mask() const { return kmer_mask_; }
    // mask selecting the (k-1)-symbol suffix of a forward k-mer (collector state key)
    uint64_t suffix_mask() const { return suffix_mask_; }

    uint8_t code(unsigned char symbol) const { return mol_ == Molecule::DNA ? (uint8_t)((symbol >> 1) & 3u) : aa_code_[symbol]; }
    char reduce(unsigned char residue) const { return reduce_[residue]; }
    const std::array<uint8_t, 256>& aa_table() const { return aa_code_; }
    const std::array<char, 256>& reduce_table() const { return reduce_; }

    // roll one symbol into the forward k-mer; returns the value to probe
    uint64_t roll(unsigned char symbol, uint64_t& forward) const {
        forward = ((forward << bits_) & kmer_mask_) | code(symbol);
        return mol_ == Molecule::DNA ? canonical(forward) : forward;
    }
    uint64_t canonical(uint64_t forward) const;

    // values inserted for one record (index construction).  `wraparound` reproduces the
    // reference's DNA behaviour of rolling the first k symbols a second time
    // (include/nucleotide_decomposer.h:106-110); it has no effect on peptides.
    void record_values(std::string_view seq, bool wraparound, std::vector<uint64_t>& out) const;

  private:
    Molecule mol_ = Molecule::DNA;
    Alphabet alphabet_ = Alphabet::Base;
    unsigned k_ = 0, bits_ = 2;
    uint64_t kmer_mask_ = 0, suffix_mask_ = 0;
    std::array<uint8_t, 256> aa_code_{};
    std::array<char, 256> reduce_{};
};

}  // namespace tetrex

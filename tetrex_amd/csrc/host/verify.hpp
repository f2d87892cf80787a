// Host side: verification of candidate bins — product code, CPU/IO stage downstream of the GPU
// (SURVEY.md §8f item 1).  Mirrors the observable behaviour of the reference's
//   iter_disk_search          include/query.h:97-188
//   verify_fasta_hit          src/query.cpp:194-237   "binpath\t>name\tmatch\tstart,end"
//   reverse_verify_fasta_hit  src/query.cpp:167-191   "binpath\t>name\tmatch\tREVERSE STRAND HIT"
//   verify_reduced_fasta_hit  src/query.cpp:240-315   (sequence mapped through the reduction first)
//   verify_fasta_set          src/query.cpp:318-339   (-c conjunction)
// The reference matches with RE2 (absent here); this build has its own linear-time matcher (matcher.hpp):
// leftmost-longest for peptides (RE2::POSIX), leftmost-first for DNA (RE2 default syntax).
#pragma once
#include "encoder.hpp"

#include <ostream>
#include <string>
#include <vector>

namespace tetrex {

struct VerifyOptions {
    int threads = 1;
};

// Scan the FASTA files of `bins` for `regex`; rows go to `out` in bin order (DNA reverse-strand
// rows go to `reverse_out`, which the reference always sends to stdout).  Returns matches found.
size_t verify_bins(const std::vector<uint64_t>& bins, const std::vector<std::string>& bin_paths, const std::string& regex,
                   const KmerEncoder& enc, std::ostream& out, std::ostream& reverse_out, const VerifyOptions& opt);

// A batch of queries (-f), verified BIN-MAJOR: the reference verifies motif by motif (include/query.h:329-346 over
// :126-138), so a batch re-opens, re-inflates and re-parses a FASTA bin once per motif that selected it.  Here every
// candidate bin is read ONCE and all the motifs that selected it run over its records (OpenMP over the bins, like the
// reference's loop over one motif's bins); a thread keeps its lazily built automata from bin to bin.  masks[q] = the
// candidate-bin mask of query q (mask_words words; nullptr: the query is skipped).  forward[q] / reverse[q] receive exactly
// the rows verify_bins(set_bins(masks[q]), ...) writes to `out` / `reverse_out` — same bytes, same order.
// Returns the matches found.
size_t verify_batch(const std::vector<const uint64_t*>& masks, uint64_t bins, const std::vector<std::string>& bin_paths,
                    const std::vector<std::string>& regexes, const KmerEncoder& enc, std::vector<std::string>* forward,
                    std::vector<std::string>* reverse, const VerifyOptions& opt);

// -c: records that match EVERY query
size_t verify_conjunction(const std::vector<uint64_t>& bins, const std::vector<std::string>& bin_paths,
                          const std::vector<std::string>& queries, std::ostream& out, const VerifyOptions& opt);

}  // namespace tetrex

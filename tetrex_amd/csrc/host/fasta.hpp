// Host side: minimal gz/plain FASTA/FASTQ record reader (zlib) — product code.
// Same record semantics as the klib kseq reader the reference vendors (include/kseq.h):
// name = first word after '>' or '@', comment = rest of the header line, sequence = following
// lines concatenated until a line starts with '>', '@' or '+'.
#pragma once
#include <functional>
#include <string>

namespace tetrex {

struct FastaRecord {
    std::string name, comment, seq;
};

// Calls `fn` for every record of `path`; returns the record count.
// Throws std::runtime_error if the file cannot be opened.
size_t for_each_record(const std::string& path, const std::function<void(const FastaRecord&)>& fn);

}  // namespace tetrex

// Host side: minimal gz/plain FASTA/FASTQ record reader (zlib) — product code.
// Same record semantics as the klib kseq reader the reference vendors (include/kseq.h):
// name = first word after '>' or '@', comment = rest of the header line, sequence = following
// lines concatenated until a line starts with '>', '@' or '+'.
#pragma once
#include <array>
#include <cstdint>
#include <functional>
#include <string>
#include <string_view>
#include <vector>

namespace tetrex {

struct FastaRecord {
    std::string name, comment, seq;
};

// Calls `fn` for every record of `path`; returns the record count.
// Throws std::runtime_error if the file cannot be opened.
size_t for_each_record(const std::string& path, const std::function<void(const FastaRecord&)>& fn);

// All records of a file at once: the sequences back to back in ONE buffer, a '\n' behind each (no sequence holds one), so
// that a string can be searched for in the whole bin with one memmem and mapped back to its record.  Same record semantics
// as for_each_record.  A plain file is read with one read() and split with memchr; a gzip file is inflated through zlib.
struct RecordSet {
    std::string text;                 // seq_0 '\n' seq_1 '\n' ...
    std::vector<size_t> start;        // start[i] = offset of record i in `text`; start[n] = text.size()
    std::vector<std::string> names;   // first word after '>' / '@'
    std::string raw;                  // (scratch: the file as read; kept so that loading bin after bin allocates nothing)
    size_t size() const { return names.size(); }
    std::string_view seq(size_t i) const { return std::string_view(text).substr(start[i], start[i + 1] - start[i] - 1); }
    size_t record_at(size_t offset) const;  // the record that holds text[offset]
};
void load_records(const std::string& path, RecordSet& out);  // throws std::runtime_error if the file cannot be opened

// The records of `set`-shaped text (`text`: set.text itself, or a buffer laid out like it) in which `needle` (2 bytes at least)
// occurs, ascending, into `records`; false — and nothing useful in `records` — as soon as more than `limit` records hold it.
// (32 positions per step where the CPU has AVX2: first and last byte compared at once, the rest only where both match.)
bool records_with(const RecordSet& set, const std::string& text, const std::string& needle, size_t limit, std::vector<size_t>& records);
// The same for a RUN of byte sets (Matcher::required_run): the records in which, at consecutive positions, one byte of each set
// occurs.  The two most selective sets are tested at 32 positions per step (AVX2, nibble look-ups), the rest where both hold.
bool records_with_run(const RecordSet& set, const std::string& text, const std::vector<std::array<uint64_t, 4>>& run, size_t limit, std::vector<size_t>& records);

}  // namespace tetrex

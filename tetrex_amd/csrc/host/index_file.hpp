// Host side: the on-disk TetRex index (".ibf") — product code.
// Layout = cereal BinaryOutputArchive of TetrexIndex (reference include/index_base.h:160-165,
// store_ibf/load_ibf :181-195, load_params :197-202): native little endian, no tags; string and
// vector = u64 count + payload; std::array<arithmetic> = raw bytes; bool/u8 = 1 byte; float = 4.
//
//   u8 k_ | str molecule_ | u8 is_hibf | vec<str> acid_libs_ | u8 reduction_ |
//   IBFIndex  (include/index_ibf.h:152-156):  u64 bin_count_ | u64 bin_size_ (always 0, shadowed member)
//                                             | u8 hash_count_ | vec<str> tech_bins_ | <hibf IBF>
//   HIBFIndex (include/index_hibf.h:154-158): u64 bin_count_ | f32 fpr_ | u8 hash_count_ | vec<str> user_bins_ | <hibf HIBF>
//   MoleculeDecomposer (include/molecule_decomposer.h:118-122): u8 ksize_ | u8 lshift_ | u8 rmask_ |
//       Nucleotide (include/nucleotide_decomposer.h:123-127): u8 k_ | u8 reduction_ | u8 left_shift_ | u64 selection_mask_
//    or Peptide    (include/peptide_decomposer.h:301-305):    u8 ksize_ | u8 reduction_ | u8 alphabet_size_ | u64 selection_mask_
//                                                             | u8[256] aamap_ | char[256] redmap_
//
// The nested <hibf IBF>/<hibf HIBF> blocks are serialised by seqan::hibf, whose sources are not
// in the reference tree, so their exact field order is UNPINNED.  The reader therefore tries
// the known/plausible variants and accepts the one that satisfies every structural invariant
// and ends exactly at EOF; the writer emits one documented variant (see index_file.cpp).
// A second reader handles the legacy container of the reference's own test fixture
// (test/data/ibf_idx.ibf, seqan3/sdsl era), which IS pinned byte for byte.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace tetrex {

struct IbfImage {
    uint64_t bins = 0, tech_bins = 0, bin_size = 0, hash_shift = 0, bin_words = 0, hash_funs = 0;
    std::vector<uint64_t> words;  // row-major [bin_size][bin_words]
    // An image read with read_index_file() does not copy the bit matrix out of the file: `mapped` then points at the
    // bin_size * bin_words words inside the file mapping (which the IndexImage keeps alive) and `words` stays empty.
    // The pointer has the file's alignment — any byte offset — so it is only ever handed to memcpy-like consumers.
    const uint8_t* mapped = nullptr;
    const uint64_t* word_data() const { return mapped ? reinterpret_cast<const uint64_t*>(mapped) : words.data(); }
    size_t word_count() const { return mapped ? (size_t)(bin_size * bin_words) : words.size(); }
    void shape(uint64_t bin_count, uint64_t rows, uint64_t h);  // fills the scalars, zeroes the words
    bool consistent() const;
};

struct HibfImage {
    uint64_t user_bins = 0;
    std::vector<IbfImage> ibfs;                         // [0] = root
    std::vector<std::vector<uint64_t>> next_ibf_id;     // [ibf][technical bin]
    std::vector<std::vector<uint64_t>> tb_to_user_bin;  // merged = UINT64_MAX
};

struct IndexImage {
    uint8_t k = 0;
    std::string molecule;  // "na" | "aa"
    bool is_hibf = false;
    uint8_t reduction = 0;  // 0 Base, 1 Murphy, 2 Li
    uint8_t hash_count = 0;
    float fpr = 0.05f;
    std::vector<std::string> bin_paths;  // one FASTA path per (user) bin
    IbfImage ibf;    // !is_hibf
    HibfImage hibf;  // is_hibf
    std::string format;  // which on-disk variant was recognised / will be written
    std::shared_ptr<void> mapping;  // the mapped file behind every IbfImage::mapped of this image

    uint64_t bin_count() const { return is_hibf ? hibf.user_bins : ibf.bins; }
};

// The d-gram index of `tetrex track` / `tetrex query -g` (reference include/dGramIndex.h:305-309):
//   u64 min_gap | u64 max_gap | u64 pad | u8 hash_count | f32 fpr | vec<str> bins | <hibf IBF> | u64 bin_count | bit_vector hits
struct DgramImage {
    uint64_t min_gap = 3, max_gap = 21, pad = 1;
    uint8_t hash_count = 3;
    float fpr = 0.05f;
    std::vector<std::string> bin_paths;
    IbfImage ibf;
    std::string format;
};
DgramImage parse_dgram_index(const std::vector<uint8_t>& bytes);
DgramImage read_dgram_index_file(const std::string& path);
std::vector<uint8_t> serialise_dgram_index(const DgramImage& d);
void write_dgram_index_file(const std::string& path, const DgramImage& d);

// Throws std::runtime_error with a descriptive message on malformed input.
// read_index_file maps the file and leaves the bit matrices in the mapping (IbfImage::mapped): a 787 MB index used to
// cost 0.56 s of reading and copying before the upload could start; the upload now copies straight out of the page cache.
IndexImage read_index_file(const std::string& path);
IndexImage parse_index(const std::vector<uint8_t>& bytes);
std::vector<uint8_t> serialise_index(const IndexImage& ix);
void write_index_file(const std::string& path, const IndexImage& ix);
// index_params peek (k, molecule, is_hibf) — load_params of the reference
void peek_index_params(const std::vector<uint8_t>& bytes, uint8_t& k, std::string& molecule, bool& is_hibf);

}  // namespace tetrex

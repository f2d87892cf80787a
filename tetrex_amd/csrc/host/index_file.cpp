#include "index_file.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "encoder.hpp"

#include <cstring>
#include <fstream>
#include <stdexcept>

namespace tetrex {

void IbfImage::shape(uint64_t bin_count, uint64_t rows, uint64_t h) {
    bins = bin_count;
    bin_words = (bin_count + 63) / 64;
    tech_bins = bin_words * 64;
    bin_size = rows;
    hash_shift = rows ? (uint64_t)__builtin_clzll(rows) : 64;
    hash_funs = h;
    words.assign(rows * bin_words, 0);
}

bool IbfImage::consistent() const {
    if (mapped && !words.empty()) return false;
    return bins >= 1 && bin_size >= 1 && bin_words == (bins + 63) / 64 && tech_bins == bin_words * 64 &&
           hash_shift == (uint64_t)__builtin_clzll(bin_size) && hash_funs >= 1 && hash_funs <= 5 &&
           word_count() == bin_size * bin_words;
}

namespace {

struct ParseError : std::runtime_error { using std::runtime_error::runtime_error; };

class In {
  public:
    In(const uint8_t* p, size_t n, bool zero_copy = false) : zero_copy(zero_copy), p_(p), n_(n) {}
    const bool zero_copy;  // large arrays stay where they are (the caller keeps the buffer alive)
    const uint8_t* view(size_t n) {
        if (left() < n) throw ParseError("unexpected end of file");
        const uint8_t* v = p_ + at_;
        at_ += n;
        return v;
    }
    size_t at() const { return at_; }
    size_t left() const { return n_ - at_; }
    template <class T> T get() {
        if (left() < sizeof(T)) throw ParseError("unexpected end of file");
        T v;
        std::memcpy(&v, p_ + at_, sizeof(T));
        at_ += sizeof(T);
        return v;
    }
    void bytes(void* dst, size_t n) {
        if (left() < n) throw ParseError("unexpected end of file");
        std::memcpy(dst, p_ + at_, n);
        at_ += n;
    }
    std::string str() {
        const uint64_t n = get<uint64_t>();
        if (n > left()) throw ParseError("string length beyond end of file");
        std::string s((const char*)p_ + at_, (size_t)n);
        at_ += n;
        return s;
    }
    std::vector<std::string> strs() {
        const uint64_t n = get<uint64_t>();
        if (n > left() / 8) throw ParseError("string count beyond end of file");
        std::vector<std::string> v((size_t)n);
        for (auto& s : v) s = str();
        return v;
    }
    std::vector<uint64_t> u64s() {
        const uint64_t n = get<uint64_t>();
        if (n > left() / 8) throw ParseError("vector length beyond end of file");
        std::vector<uint64_t> v((size_t)n);
        bytes(v.data(), v.size() * 8);
        return v;
    }

  private:
    const uint8_t* p_;
    size_t n_, at_ = 0;
};

// One way seqan::hibf may have laid out its objects.
struct Variant {
    bool ibf_version;   // u32 version in front of the IBF scalars
    int bit_vector;     // 0 {bits, words}  1 {n_words, words, bits}  2 {bits, n_words, words}
    bool pad512;        // words rounded up to 512-bit blocks
    int occupancy;      // 0 none, 1 {vec<u64>, u8} after the data, 2 before the data
    bool hibf_version;  // u32 version in front of the HIBF fields
    bool hibf_prev;     // vec<{u64,u64}> prev_ibf_id between next_ibf_id and the user-bin map
    std::string name() const {
        return std::string("hibf-layout{ibfver=") + (ibf_version ? "1" : "0") + ",bv=" + std::to_string(bit_vector) +
               (pad512 ? ",pad512" : "") + ",occ=" + std::to_string(occupancy) + ",hibfver=" + (hibf_version ? "1" : "0") +
               ",prev=" + (hibf_prev ? "1" : "0") + "}";
    }
};

void read_bit_vector(In& in, const Variant& v, IbfImage& f) {
    const uint64_t want_bits = f.tech_bins * f.bin_size;
    const uint64_t exact = (want_bits + 63) / 64;
    const uint64_t stored = v.pad512 ? ((exact + 7) / 8) * 8 : exact;
    auto take = [&](uint64_t n_words) {
        if (n_words != stored) throw ParseError("bit-vector word count mismatch");
        if (n_words > in.left() / 8) throw ParseError("bit vector beyond end of file");
        if (in.zero_copy) {  // the first `exact` words are the matrix; padding words follow
            f.words.clear();
            f.mapped = in.view((size_t)n_words * 8);
            return;
        }
        std::vector<uint64_t> w((size_t)n_words);
        in.bytes(w.data(), w.size() * 8);
        w.resize((size_t)exact);
        f.words.swap(w);
    };
    if (v.bit_vector == 0) {
        if (in.get<uint64_t>() != want_bits) throw ParseError("bit-vector size mismatch");
        take(stored);
    } else if (v.bit_vector == 1) {
        take(in.get<uint64_t>());
        if (in.get<uint64_t>() != want_bits) throw ParseError("bit-vector size mismatch");
    } else {
        if (in.get<uint64_t>() != want_bits) throw ParseError("bit-vector size mismatch");
        take(in.get<uint64_t>());
    }
}

void read_occupancy(In& in, const IbfImage& f) {
    const std::vector<uint64_t> occ = in.u64s();
    if (occ.size() != f.bins && occ.size() != f.tech_bins && !occ.empty()) throw ParseError("occupancy length mismatch");
    if (in.get<uint8_t>() > 1) throw ParseError("track_occupancy is not a bool");
}

IbfImage read_hibf_ibf(In& in, const Variant& v) {
    IbfImage f;
    if (v.ibf_version) {
        const uint32_t ver = in.get<uint32_t>();
        if (ver == 0 || ver > 16) throw ParseError("implausible IBF version");
    }
    f.bins = in.get<uint64_t>();
    f.tech_bins = in.get<uint64_t>();
    f.bin_size = in.get<uint64_t>();
    f.hash_shift = in.get<uint64_t>();
    f.bin_words = in.get<uint64_t>();
    f.hash_funs = in.get<uint64_t>();
    if (f.bins == 0 || f.bin_size == 0 || f.bin_words != (f.bins + 63) / 64 || f.tech_bins != f.bin_words * 64 ||
        f.hash_shift != (uint64_t)__builtin_clzll(f.bin_size) || f.hash_funs < 1 || f.hash_funs > 5)
        throw ParseError("IBF scalars violate their invariants");
    if (f.bin_size > (1ULL << 58) / f.tech_bins) throw ParseError("IBF too large");
    if (v.occupancy == 2) read_occupancy(in, f);
    read_bit_vector(in, v, f);
    if (v.occupancy == 1) read_occupancy(in, f);
    return f;
}

HibfImage read_hibf(In& in, const Variant& v) {
    HibfImage h;
    if (v.hibf_version) {
        const uint32_t ver = in.get<uint32_t>();
        if (ver == 0 || ver > 16) throw ParseError("implausible HIBF version");
    }
    h.user_bins = in.get<uint64_t>();
    const uint64_t n = in.get<uint64_t>();
    if (n == 0 || n > in.left() / 48) throw ParseError("implausible IBF count");
    for (uint64_t i = 0; i < n; ++i) h.ibfs.push_back(read_hibf_ibf(in, v));
    auto nested = [&](std::vector<std::vector<uint64_t>>& out) {
        const uint64_t cnt = in.get<uint64_t>();
        if (cnt != n) throw ParseError("HIBF map count mismatch");
        for (uint64_t i = 0; i < n; ++i) {
            out.push_back(in.u64s());
            // hibf sizes these by technical bins; trailing entries beyond `bins` are unused
            if (out.back().size() != h.ibfs[i].bins && out.back().size() != h.ibfs[i].tech_bins) throw ParseError("HIBF map length mismatch");
            out.back().resize((size_t)h.ibfs[i].bins);
        }
    };
    nested(h.next_ibf_id);
    if (v.hibf_prev) {
        const uint64_t cnt = in.get<uint64_t>();
        if (cnt != n) throw ParseError("prev_ibf_id count mismatch");
        for (uint64_t i = 0; i < 2 * n; ++i) in.get<uint64_t>();
    }
    nested(h.tb_to_user_bin);
    for (uint64_t i = 0; i < n; ++i)
        for (uint64_t b = 0; b < h.ibfs[i].bins; ++b) {
            const uint64_t ub = h.tb_to_user_bin[i][b];
            if (ub == UINT64_MAX) { if (h.next_ibf_id[i][b] >= n) throw ParseError("merged bin without child"); }
            else if (ub >= h.user_bins) throw ParseError("user bin id out of range");
        }
    return h;
}

void read_decomposer(In& in, const IndexImage& ix) {
    const uint8_t ksize = in.get<uint8_t>(), lshift = in.get<uint8_t>(), rmask = in.get<uint8_t>();
    const bool dna = ix.molecule == "na";
    if (ksize != ix.k || lshift != (dna ? 2 : 5) || rmask != (dna ? 3 : 31)) throw ParseError("decomposer header mismatch");
    const uint8_t k2 = in.get<uint8_t>();
    const uint8_t red = in.get<uint8_t>();
    in.get<uint8_t>();  // left_shift_ (na) / alphabet_size_ (aa)
    in.get<uint64_t>(); // selection_mask_
    if (k2 != ix.k || red != ix.reduction) throw ParseError("decomposer block mismatch");
    if (!dna) {
        uint8_t maps[512];
        in.bytes(maps, sizeof maps);
    }
}

IndexImage parse_current(const uint8_t* p, size_t n, const Variant& v, bool zero_copy = false) {
    In in(p, n, zero_copy);
    IndexImage ix;
    ix.k = in.get<uint8_t>();
    ix.molecule = in.str();
    if (ix.molecule != "na" && ix.molecule != "aa") throw ParseError("molecule is neither \"na\" nor \"aa\"");
    const uint8_t hibf = in.get<uint8_t>();
    if (hibf > 1) throw ParseError("is_hibf flag is not a bool");
    ix.is_hibf = hibf != 0;
    std::vector<std::string> libs = in.strs();
    ix.reduction = in.get<uint8_t>();
    if (ix.reduction > 2 || ix.k == 0) throw ParseError("bad reduction / k");
    const uint64_t bin_count = in.get<uint64_t>();
    if (ix.is_hibf) ix.fpr = in.get<float>();
    else in.get<uint64_t>();  // IBFIndex::bin_size_ (always 0)
    ix.hash_count = in.get<uint8_t>();
    ix.bin_paths = in.strs();
    if (ix.bin_paths.size() != bin_count || libs.size() != bin_count) throw ParseError("bin path count mismatch");
    if (ix.is_hibf) {
        ix.hibf = read_hibf(in, v);
        if (ix.hibf.user_bins != bin_count) throw ParseError("HIBF user bin count mismatch");
    } else {
        ix.ibf = read_hibf_ibf(in, v);
        if (ix.ibf.bins != bin_count || ix.ibf.hash_funs != ix.hash_count) throw ParseError("IBF shape mismatch");
    }
    read_decomposer(in, ix);
    if (in.left() != 0) throw ParseError("trailing bytes after the decomposer block");
    ix.format = "cereal/" + v.name();
    return ix;
}

// The reference's own fixture (test/data/ibf_idx.ibf): seqan3-era container, sdsl bit vector.
IndexImage parse_legacy(const uint8_t* p, size_t n) {
    In in(p, n);
    IndexImage ix;
    const uint64_t bin_count = in.get<uint64_t>();
    const uint64_t bin_size = in.get<uint64_t>();
    ix.hash_count = in.get<uint8_t>();
    IbfImage& f = ix.ibf;
    f.bins = in.get<uint64_t>(); f.tech_bins = in.get<uint64_t>(); f.bin_size = in.get<uint64_t>();
    f.hash_shift = in.get<uint64_t>(); f.bin_words = in.get<uint64_t>(); f.hash_funs = in.get<uint64_t>();
    if (f.bins != bin_count || f.bin_size != bin_size || f.hash_funs != ix.hash_count || f.bins == 0 || f.bin_size == 0 ||
        f.bin_words != (f.bins + 63) / 64 || f.tech_bins != f.bin_words * 64 || f.hash_shift != (uint64_t)__builtin_clzll(f.bin_size) ||
        f.hash_funs < 1 || f.hash_funs > 5)
        throw ParseError("legacy header violates the IBF invariants");
    in.get<uint8_t>();  // sdsl int_vector width
    in.get<float>();    // growth factor
    const uint64_t bits = in.get<uint64_t>();
    if (bits != f.tech_bins * f.bin_size) throw ParseError("legacy bit count mismatch");
    f.words.resize((size_t)((bits + 63) / 64));
    in.bytes(f.words.data(), f.words.size() * 8);
    ix.k = in.get<uint8_t>();
    ix.molecule = in.str();
    if (ix.molecule != "na" && ix.molecule != "aa") throw ParseError("legacy molecule tag");
    ix.bin_paths = in.strs();
    if (ix.bin_paths.size() != bin_count) throw ParseError("legacy path count mismatch");
    ix.format = "legacy-seqan3";
    return ix;  // trailing reduction name etc. are not needed on the query path
}

DgramImage parse_dgram(const uint8_t* p, size_t n, const Variant& v) {
    In in(p, n);
    DgramImage d;
    d.min_gap = in.get<uint64_t>();
    d.max_gap = in.get<uint64_t>();
    d.pad = in.get<uint64_t>();
    d.hash_count = in.get<uint8_t>();
    d.fpr = in.get<float>();
    if (d.min_gap > d.max_gap || d.max_gap > (1u << 20) || d.hash_count < 1 || d.hash_count > 5) throw ParseError("implausible d-gram header");
    d.bin_paths = in.strs();
    d.ibf = read_hibf_ibf(in, v);
    const uint64_t bc = in.get<uint64_t>();
    if (bc != d.bin_paths.size() || bc != d.ibf.bins || d.ibf.hash_funs != d.hash_count) throw ParseError("d-gram bin count mismatch");
    // trailing `hits_` bit vector of bc bits, same container layout as the filter's data
    const uint64_t words = (bc + 63) / 64, stored = v.pad512 ? ((words + 7) / 8) * 8 : words;
    if (v.bit_vector == 0) { if (in.get<uint64_t>() != bc) throw ParseError("hits size"); }
    else if (v.bit_vector == 1) { if (in.get<uint64_t>() != stored) throw ParseError("hits words"); }
    else { if (in.get<uint64_t>() != bc || in.get<uint64_t>() != stored) throw ParseError("hits size"); }
    for (uint64_t i = 0; i < stored; ++i) in.get<uint64_t>();
    if (v.bit_vector == 1 && in.get<uint64_t>() != bc) throw ParseError("hits size");
    if (in.left() != 0) throw ParseError("trailing bytes");
    d.format = "cereal/" + v.name();
    return d;
}

}  // namespace

DgramImage parse_dgram_index(const std::vector<uint8_t>& bytes) {
    std::string first_error;
    for (int ibfver = 1; ibfver >= 0; --ibfver)
        for (int bv = 0; bv < 3; ++bv)
            for (int pad = 0; pad < 2; ++pad)
                for (int occ = 0; occ < 3; ++occ) {
                    const Variant v{ibfver != 0, bv, pad != 0, occ, false, false};
                    try {
                        return parse_dgram(bytes.data(), bytes.size(), v);
                    } catch (const ParseError& e) {
                        if (first_error.empty()) first_error = e.what();
                    }
                }
    throw std::runtime_error("not a TetRex d-gram index (no known layout variant fits): " + first_error);
}

// the whole file in one sized read (an index is hundreds of MB: no byte-wise stream iteration)
static std::vector<uint8_t> read_whole_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("Filepath " + path + " not valid");
    const std::streamoff size = f.tellg();
    if (size < 0) throw std::runtime_error("Filepath " + path + " not valid");
    std::vector<uint8_t> bytes((size_t)size);
    f.seekg(0);
    if (size && !f.read(reinterpret_cast<char*>(bytes.data()), size)) throw std::runtime_error("could not read " + path);
    return bytes;
}

DgramImage read_dgram_index_file(const std::string& path) {
    return parse_dgram_index(read_whole_file(path));
}

static IndexImage parse_any(const uint8_t* data, size_t size, bool zero_copy) {
    std::string first_error;
    for (int ibfver = 1; ibfver >= 0; --ibfver)
        for (int bv = 0; bv < 3; ++bv)
            for (int pad = 0; pad < 2; ++pad)
                for (int occ = 0; occ < 3; ++occ)
                    for (int hv = 1; hv >= 0; --hv)
                        for (int prev = 1; prev >= 0; --prev) {
                            const Variant v{ibfver != 0, bv, pad != 0, occ, hv != 0, prev != 0};
                            try {
                                return parse_current(data, size, v, zero_copy);
                            } catch (const ParseError& e) {
                                if (first_error.empty()) first_error = e.what();
                            }
                        }
    try {
        return parse_legacy(data, size);
    } catch (const ParseError&) {
    }
    throw std::runtime_error("not a TetRex index (no known layout variant fits): " + first_error);
}

IndexImage parse_index(const std::vector<uint8_t>& bytes) { return parse_any(bytes.data(), bytes.size(), false); }

// The file is mapped, not read: the scalars and tables are parsed out of the mapping, the bit matrices stay in it
// (IbfImage::mapped) and go from the page cache straight into the upload's copies.
IndexImage read_index_file(const std::string& path) {
    struct Mapping {
        void* p = MAP_FAILED;
        size_t n = 0;
        ~Mapping() { if (p != MAP_FAILED) munmap(p, n); }
    };
    const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) throw std::runtime_error("Filepath " + path + " not valid");
    struct stat st{};
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); throw std::runtime_error("Filepath " + path + " not valid"); }
    if (st.st_size == 0) { ::close(fd); throw std::runtime_error("not a TetRex index: the file is empty"); }
    auto m = std::make_shared<Mapping>();
    m->n = (size_t)st.st_size;
    m->p = mmap(nullptr, m->n, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m->p == MAP_FAILED) return parse_index(read_whole_file(path));  // e.g. a file system without mmap: read it
    (void)madvise(m->p, m->n, MADV_WILLNEED);  // start the read-ahead now; the upload touches the pages next
    IndexImage image = parse_any(static_cast<const uint8_t*>(m->p), m->n, true);
    image.mapping = m;
    return image;
}

void peek_index_params(const std::vector<uint8_t>& bytes, uint8_t& k, std::string& molecule, bool& is_hibf) {
    In in(bytes.data(), bytes.size());
    k = in.get<uint8_t>();
    molecule = in.str();
    is_hibf = in.get<uint8_t>() != 0;
}

namespace {
class Out {
  public:
    std::vector<uint8_t> buf;
    template <class T> void put(T v) {
        const size_t at = buf.size();
        buf.resize(at + sizeof(T));
        std::memcpy(buf.data() + at, &v, sizeof(T));
    }
    void raw(const void* p, size_t n) {
        const size_t at = buf.size();
        buf.resize(at + n);
        if (n) std::memcpy(buf.data() + at, p, n);
    }
    void str(const std::string& s) { put<uint64_t>(s.size()); raw(s.data(), s.size()); }
    void strs(const std::vector<std::string>& v) { put<uint64_t>(v.size()); for (auto& s : v) str(s); }
    void u64s(const std::vector<uint64_t>& v) { put<uint64_t>(v.size()); raw(v.data(), v.size() * 8); }
};

// Written variant: u32 version = 1 | six scalars | bit_vector {u64 size_in_bits | words}.
void write_ibf(Out& o, const IbfImage& f) {
    if (!f.consistent()) throw std::runtime_error("inconsistent IBF image");
    o.put<uint32_t>(1);
    o.put(f.bins); o.put(f.tech_bins); o.put(f.bin_size); o.put(f.hash_shift); o.put(f.bin_words); o.put(f.hash_funs);
    o.put<uint64_t>(f.tech_bins * f.bin_size);
    o.raw(f.word_data(), f.word_count() * 8);
}
}  // namespace

std::vector<uint8_t> serialise_index(const IndexImage& ix) {
    const bool dna = ix.molecule == "na";
    if (!dna && ix.molecule != "aa") throw std::runtime_error("molecule must be \"na\" or \"aa\"");
    Out o;
    o.put<uint8_t>(ix.k);
    o.str(ix.molecule);
    o.put<uint8_t>(ix.is_hibf ? 1 : 0);
    o.strs(ix.bin_paths);
    o.put<uint8_t>(ix.reduction);
    o.put<uint64_t>(ix.bin_count());
    if (ix.is_hibf) o.put<float>(ix.fpr);
    else o.put<uint64_t>(0);
    o.put<uint8_t>(ix.hash_count);
    o.strs(ix.bin_paths);
    if (ix.is_hibf) {
        const HibfImage& h = ix.hibf;
        if (h.ibfs.empty() || h.next_ibf_id.size() != h.ibfs.size() || h.tb_to_user_bin.size() != h.ibfs.size())
            throw std::runtime_error("inconsistent HIBF image");
        o.put<uint32_t>(1);
        o.put<uint64_t>(h.user_bins);
        o.put<uint64_t>(h.ibfs.size());
        for (const auto& f : h.ibfs) write_ibf(o, f);
        o.put<uint64_t>(h.ibfs.size());
        for (const auto& v : h.next_ibf_id) o.u64s(v);
        o.put<uint64_t>(h.ibfs.size());
        for (const auto& v : h.tb_to_user_bin) o.u64s(v);
    } else {
        write_ibf(o, ix.ibf);
    }
    // decomposer block
    const KmerEncoder enc(dna ? Molecule::DNA : Molecule::Peptide, ix.k, (Alphabet)ix.reduction);
    o.put<uint8_t>(ix.k);
    o.put<uint8_t>(dna ? 2 : 5);
    o.put<uint8_t>(dna ? 3 : 31);
    o.put<uint8_t>(ix.k);
    o.put<uint8_t>(ix.reduction);
    if (dna) {
        o.put<uint8_t>((uint8_t)(2 * ix.k - 2));
        o.put<uint64_t>(enc.kmer_mask());
    } else {
        o.put<uint8_t>(ix.reduction == 0 ? 20 : 10);
        o.put<uint64_t>(enc.kmer_mask());
        o.raw(enc.aa_table().data(), 256);
        o.raw(enc.reduce_table().data(), 256);
    }
    return std::move(o.buf);
}

std::vector<uint8_t> serialise_dgram_index(const DgramImage& d) {
    Out o;
    o.put<uint64_t>(d.min_gap);
    o.put<uint64_t>(d.max_gap);
    o.put<uint64_t>(d.pad);
    o.put<uint8_t>(d.hash_count);
    o.put<float>(d.fpr);
    o.strs(d.bin_paths);
    write_ibf(o, d.ibf);
    o.put<uint64_t>(d.ibf.bins);
    o.put<uint64_t>(d.ibf.bins);  // hits_: bit_vector {size in bits | words}, all set (include/dGramIndex.h:96)
    for (uint64_t w = 0; w < (d.ibf.bins + 63) / 64; ++w) {
        const uint64_t left = d.ibf.bins - w * 64;
        o.put<uint64_t>(left >= 64 ? ~0ULL : ((1ULL << left) - 1ULL));
    }
    return std::move(o.buf);
}

void write_dgram_index_file(const std::string& path, const DgramImage& d) {
    const std::vector<uint8_t> bytes = serialise_dgram_index(d);
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f) throw std::runtime_error("cannot open " + path + " for writing");
    f.write((const char*)bytes.data(), (std::streamsize)bytes.size());
    if (!f) throw std::runtime_error("short write to " + path);
}

void write_index_file(const std::string& path, const IndexImage& ix) {
    const std::vector<uint8_t> bytes = serialise_index(ix);
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f) throw std::runtime_error("cannot open " + path + " for writing");
    f.write((const char*)bytes.data(), (std::streamsize)bytes.size());
    if (!f) throw std::runtime_error("short write to " + path);
}

}  // namespace tetrex

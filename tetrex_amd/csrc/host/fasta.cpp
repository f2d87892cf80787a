#include "fasta.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace tetrex {

namespace {
class Lines {
  public:
    explicit Lines(gzFile f) : f_(f) {}
    // next line without its terminator; false at EOF
    bool next(std::string& line) {
        line.clear();
        bool any = false;
        for (;;) {
            if (at_ == len_) {
                len_ = gzread(f_, buf_, sizeof buf_);
                at_ = 0;
                if (len_ <= 0) { len_ = 0; return any; }
            }
            any = true;
            const char* p = buf_ + at_;
            const char* e = buf_ + len_;
            const char* q = p;
            while (q < e && *q != '\n') ++q;
            line.append(p, q - p);
            at_ = (int)(q - buf_);
            if (q < e) {  // found '\n'
                ++at_;
                if (!line.empty() && line.back() == '\r') line.pop_back();
                return true;
            }
        }
    }

  private:
    gzFile f_;
    char buf_[1 << 16];
    int at_ = 0, len_ = 0;
};
}  // namespace

size_t for_each_record(const std::string& path, const std::function<void(const FastaRecord&)>& fn) {
    gzFile f = gzopen(path.c_str(), "r");
    if (!f) throw std::runtime_error("File not found. Did you move/rename an indexed file? (" + path + ")");
    Lines lines(f);
    std::string line;
    FastaRecord rec;
    bool open = false, in_quality = false;
    size_t count = 0;
    auto flush = [&]() {
        if (open) { fn(rec); ++count; }
        open = false;
    };
    while (lines.next(line)) {
        if (line.empty()) continue;
        const char c = line[0];
        if (in_quality) {  // FASTQ quality line(s): skip until as long as the sequence
            in_quality = false;
            continue;
        }
        if (c == '>' || c == '@') {
            flush();
            rec = FastaRecord{};
            const size_t sp = line.find_first_of(" \t");
            rec.name = line.substr(1, sp == std::string::npos ? std::string::npos : sp - 1);
            if (sp != std::string::npos) rec.comment = line.substr(sp + 1);
            open = true;
        } else if (c == '+') {
            in_quality = true;
        } else if (open) {
            rec.seq += line;
        }
    }
    flush();
    gzclose(f);
    return count;
}

namespace {
// candidate positions of `needle` in [s, s + n): called with every position whose first and last byte match
template <class Fn>
void scan_plain(const char* s, size_t n, const std::string& needle, Fn&& at) {
    const size_t m = needle.size();
    if (n < m) return;
    const char first = needle[0], last = needle[m - 1];
    for (size_t i = 0; i + m <= n; ++i)
        if (s[i] == first && s[i + m - 1] == last && !at(i)) return;
}
#if defined(__x86_64__)
template <class Fn>
__attribute__((target("avx2"))) void scan_avx2(const char* s, size_t n, const std::string& needle, Fn&& at) {
    const size_t m = needle.size();
    if (n < m) return;
    const __m256i first = _mm256_set1_epi8(needle[0]), last = _mm256_set1_epi8(needle[m - 1]);
    size_t i = 0;
    for (; i + m - 1 + 32 <= n; i += 32) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i + m - 1));
        uint32_t mask = (uint32_t)_mm256_movemask_epi8(_mm256_and_si256(_mm256_cmpeq_epi8(a, first), _mm256_cmpeq_epi8(b, last)));
        while (mask) {
            const size_t at_i = i + (size_t)__builtin_ctz(mask);
            mask &= mask - 1;
            if (!at(at_i)) return;
        }
    }
    for (; i + m <= n; ++i)
        if (s[i] == needle[0] && s[i + m - 1] == needle[m - 1] && !at(i)) return;
}
#endif
}  // namespace

bool records_with(const RecordSet& set, const std::string& text, const std::string& needle, size_t limit, std::vector<size_t>& records) {
    records.clear();
    const size_t m = needle.size();
    if (m < 2 || text.size() != set.text.size()) return false;
    bool ok = true;
    size_t last_record = (size_t)-1, skip_below = 0;  // (positions below skip_below lie in a record that is listed already)
    auto at = [&](size_t i) {
        if (i < skip_below) return true;
        if (m > 2 && std::memcmp(text.data() + i + 1, needle.data() + 1, m - 2) != 0) return true;
        const size_t r = set.record_at(i);
        if (r != last_record) {
            records.push_back(r);
            last_record = r;
            if (records.size() > limit) { ok = false; return false; }
        }
        skip_below = set.start[r + 1];
        return true;
    };
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2")) scan_avx2(text.data(), text.size(), needle, at);
    else
#endif
        scan_plain(text.data(), text.size(), needle, at);
    return ok;
}

namespace {
inline bool in_set(const std::array<uint64_t, 4>& st, unsigned char c) { return (st[c >> 6] >> (c & 63)) & 1ULL; }
inline int set_size(const std::array<uint64_t, 4>& st) {
    int b = 0;
    for (int w = 0; w < 4; ++w) b += __builtin_popcountll(st[w]);
    return b;
}
#if defined(__x86_64__)
// membership of 32 bytes in a byte set by two nibble look-ups: lo[c & 15] holds one bit per HIGH nibble value that occurs in
// the set together with that low nibble, hi[c >> 4] the bit of that high nibble; usable when the set's bytes have at most 8
// different high nibbles (letters: 4, 5, 6, 7)
struct NibbleSet {
    __m256i lo, hi;
    bool ok;
    int single = -1;  // the set is this one byte: a plain compare does
};
__attribute__((target("avx2"))) NibbleSet nibble_set(const std::array<uint64_t, 4>& st) {
    unsigned char lo[16] = {0}, hi[16] = {0};
    int bit_of_hi[16];
    int used = 0;
    for (int h = 0; h < 16; ++h) {
        bit_of_hi[h] = -1;
        bool any = false;
        for (int l = 0; l < 16 && !any; ++l) any = in_set(st, (unsigned char)(h * 16 + l));
        if (any) { if (used == 8) return NibbleSet{_mm256_setzero_si256(), _mm256_setzero_si256(), false, -1}; bit_of_hi[h] = used++; }
    }
    for (int h = 0; h < 16; ++h) {
        if (bit_of_hi[h] < 0) continue;
        hi[h] = (unsigned char)(1u << bit_of_hi[h]);
        for (int l = 0; l < 16; ++l)
            if (in_set(st, (unsigned char)(h * 16 + l))) lo[l] |= (unsigned char)(1u << bit_of_hi[h]);
    }
    const __m128i l128 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(lo)), h128 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(hi));
    NibbleSet out{_mm256_broadcastsi128_si256(l128), _mm256_broadcastsi128_si256(h128), true};
    if (set_size(st) == 1)
        for (int c = 0; c < 256; ++c) if (in_set(st, (unsigned char)c)) out.single = c;
    return out;
}
__attribute__((target("avx2"))) inline __m256i member(const NibbleSet& s, __m256i bytes) {
    if (s.single >= 0) return _mm256_cmpeq_epi8(bytes, _mm256_set1_epi8((char)s.single));
    const __m256i low = _mm256_and_si256(bytes, _mm256_set1_epi8(0x0F));
    const __m256i high = _mm256_and_si256(_mm256_srli_epi16(bytes, 4), _mm256_set1_epi8(0x0F));
    const __m256i both = _mm256_and_si256(_mm256_shuffle_epi8(s.lo, low), _mm256_shuffle_epi8(s.hi, high));
    return _mm256_xor_si256(_mm256_cmpeq_epi8(both, _mm256_setzero_si256()), _mm256_set1_epi8(-1));  // non-zero <=> member
}
// candidate positions i (run start) in [0, n - m]: bytes at i + a and i + b are in the sets a and b
template <class Fn>
__attribute__((target("avx2"))) void scan_run_avx2(const char* s, size_t n, size_t m, size_t a, size_t b, const NibbleSet& sa, const NibbleSet& sb, Fn&& at) {
    if (n < m) return;
    size_t i = 0;
    for (; i + m - 1 + 32 <= n; i += 32) {
        const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i + a));
        const __m256i y = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i + b));
        uint32_t mask = (uint32_t)_mm256_movemask_epi8(_mm256_and_si256(member(sa, x), member(sb, y)));
        while (mask) {
            const size_t at_i = i + (size_t)__builtin_ctz(mask);
            mask &= mask - 1;
            if (!at(at_i)) return;
        }
    }
    for (; i + m <= n; ++i)
        if (!at(i)) return;  // (the tail: `at` checks every position of the run itself)
}
#endif
}  // namespace

bool records_with_run(const RecordSet& set, const std::string& text, const std::vector<std::array<uint64_t, 4>>& run, size_t limit, std::vector<size_t>& records) {
    records.clear();
    const size_t m = run.size();
    if (m == 0 || text.size() != set.text.size()) return false;
    // the two most selective positions of the run
    size_t a = 0, b = 0;
    {
        int best_a = 1 << 30, best_b = 1 << 30;
        for (size_t j = 0; j < m; ++j) {
            const int z = set_size(run[j]);
            if (z < best_a) { best_b = best_a; b = a; best_a = z; a = j; }
            else if (z < best_b) { best_b = z; b = j; }
        }
        if (m == 1) b = a;
        if (a > b) std::swap(a, b);
    }
    bool ok = true;
    size_t last_record = (size_t)-1, skip_below = 0;  // (positions below skip_below lie in a record that is listed already)
    const unsigned char* t = reinterpret_cast<const unsigned char*>(text.data());
    auto at = [&](size_t i) {
        if (i < skip_below) return true;
        for (size_t j = 0; j < m; ++j)
            if (!in_set(run[j], t[i + j])) return true;  // (the '\n' between records is in no residue class: a run never straddles two)
        const size_t r = set.record_at(i);
        if (r != last_record) {
            records.push_back(r);
            last_record = r;
            if (records.size() > limit) { ok = false; return false; }
        }
        skip_below = set.start[r + 1];
        return true;
    };
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2")) {
        const NibbleSet sa = nibble_set(run[a]), sb = nibble_set(run[b]);
        if (sa.ok && sb.ok) {
            scan_run_avx2(text.data(), text.size(), m, a, b, sa, sb, at);
            return ok;
        }
    }
#endif
    for (size_t i = 0; i + m <= text.size(); ++i)
        if (in_set(run[a], t[i + a]) && in_set(run[b], t[i + b]) && !at(i)) break;
    return ok;
}

size_t RecordSet::record_at(size_t offset) const {
    size_t lo = 0, hi = names.size();  // the last record whose start is <= offset
    while (hi - lo > 1) {
        const size_t mid = (lo + hi) / 2;
        if (start[mid] <= offset) lo = mid; else hi = mid;
    }
    return lo;
}

// plain files from this size on are mapped instead of read (TETREX_MAP_FROM_MB: A/B knob; 0 = map every file)
static size_t map_from() {
    static const size_t v = [] {
        const char* e = std::getenv("TETREX_MAP_FROM_MB");
        return (size_t)(e ? std::max(0LL, std::atoll(e)) : 64) << 20;
    }();
    return v;
}

void load_records(const std::string& path, RecordSet& out) {
    out.text.clear();
    out.start.clear();
    out.names.clear();
    // the whole file in memory: plain files with one read, gzip files through zlib (gzread is transparent for plain data, but
    // its 64 KB buffer and the byte-wise line search above cost more than the matching they feed)
    std::string& raw = out.raw;
    raw.clear();
    const char* data = nullptr;  // the file's bytes: a mapping of a plain file (no copy at all), or `raw` for an inflated one
    size_t data_size = 0;
    void* mapping = nullptr;
    size_t mapping_size = 0;
    struct Unmap { void*& m; size_t& n; ~Unmap() { if (m) ::munmap(m, n); } } unmap{mapping, mapping_size};
    {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("File not found. Did you move/rename an indexed file? (" + path + ")");
        unsigned char magic[2] = {0, 0};
        const ssize_t got = ::pread(fd, magic, 2, 0);
        const bool gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        struct stat st;
        if (!gz) {
            if (::fstat(fd, &st) == 0 && st.st_size > 0) {
                // Small files are READ into this thread's buffer (kept from bin to bin): mapping and unmapping a few hundred
                // kilobytes per bin costs page faults and, with many verification threads in one address space, a TLB
                // shoot-down per file — more threads made a batch slower.  Large files are mapped (no second copy).
                void* m = (size_t)st.st_size < map_from() ? MAP_FAILED : ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) {
                    mapping = m;
                    mapping_size = (size_t)st.st_size;
                    data = static_cast<const char*>(m);
                    data_size = mapping_size;
                } else {  // (a file that cannot be mapped — a pipe, an odd file system —: read it)
                    raw.resize((size_t)st.st_size);
                    size_t used = 0;
                    for (ssize_t n; used < raw.size() && (n = ::pread(fd, raw.data() + used, raw.size() - used, (off_t)used)) > 0;) used += (size_t)n;
                    raw.resize(used);
                    data = raw.data();
                    data_size = raw.size();
                }
            }
            ::close(fd);
        } else {
            ::close(fd);
            gzFile g = gzopen(path.c_str(), "r");
            if (!g) throw std::runtime_error("File not found. Did you move/rename an indexed file? (" + path + ")");
            gzbuffer(g, 1 << 20);
            size_t used = 0;
            raw.resize(raw.capacity());
            for (;;) {
                if (raw.size() - used < (1u << 20)) raw.resize(raw.size() + (4u << 20));
                const int n = gzread(g, raw.data() + used, (unsigned)std::min<size_t>(raw.size() - used, 1u << 30));
                if (n <= 0) break;
                used += (size_t)n;
            }
            gzclose(g);
            raw.resize(used);
            data = raw.data();
            data_size = raw.size();
        }
    }
    out.text.reserve(data_size);
    bool open = false, in_quality = false;
    const char* p = data;
    const char* const e = p + data_size;
    auto close_record = [&]() {
        if (open) out.text.push_back('\n');
        open = false;
    };
    while (p < e) {
        const char* nl = static_cast<const char*>(std::memchr(p, '\n', (size_t)(e - p)));
        const char* line_end = nl ? nl : e;
        const char* next = nl ? nl + 1 : e;
        if (line_end > p && line_end[-1] == '\r') --line_end;
        if (line_end > p) {
            const char c = *p;
            if (in_quality) in_quality = false;  // FASTQ quality line: skipped
            else if (c == '>' || c == '@') {
                close_record();
                const char* sp = p + 1;
                while (sp < line_end && *sp != ' ' && *sp != '\t') ++sp;
                out.names.emplace_back(p + 1, sp);
                out.start.push_back(out.text.size());
                open = true;
            } else if (c == '+') in_quality = true;
            else if (open) out.text.append(p, line_end);
        }
        p = next;
    }
    close_record();
    out.start.push_back(out.text.size());
}

}  // namespace tetrex

#include "fasta.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <zlib.h>

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

size_t RecordSet::record_at(size_t offset) const {
    size_t lo = 0, hi = names.size();  // the last record whose start is <= offset
    while (hi - lo > 1) {
        const size_t mid = (lo + hi) / 2;
        if (start[mid] <= offset) lo = mid; else hi = mid;
    }
    return lo;
}

void load_records(const std::string& path, RecordSet& out) {
    out.text.clear();
    out.start.clear();
    out.names.clear();
    // the whole file in memory: plain files with one read, gzip files through zlib (gzread is transparent for plain data, but
    // its 64 KB buffer and the byte-wise line search above cost more than the matching they feed)
    std::string raw;
    {
        std::FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("File not found. Did you move/rename an indexed file? (" + path + ")");
        unsigned char magic[2] = {0, 0};
        const size_t got = std::fread(magic, 1, 2, f);
        const bool gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        if (!gz) {
            std::fseek(f, 0, SEEK_END);
            const long size = std::ftell(f);
            std::fseek(f, 0, SEEK_SET);
            raw.resize(size > 0 ? (size_t)size : 0);
            const size_t n = raw.empty() ? 0 : std::fread(raw.data(), 1, raw.size(), f);
            raw.resize(n);
            std::fclose(f);
        } else {
            std::fclose(f);
            gzFile g = gzopen(path.c_str(), "r");
            if (!g) throw std::runtime_error("File not found. Did you move/rename an indexed file? (" + path + ")");
            gzbuffer(g, 1 << 20);
            size_t used = 0;
            for (;;) {
                if (raw.size() - used < (1u << 20)) raw.resize(raw.size() + (4u << 20));
                const int n = gzread(g, raw.data() + used, (unsigned)std::min<size_t>(raw.size() - used, 1u << 30));
                if (n <= 0) break;
                used += (size_t)n;
            }
            gzclose(g);
            raw.resize(used);
        }
    }
    out.text.reserve(raw.size());
    bool open = false, in_quality = false;
    const char* p = raw.data();
    const char* const e = p + raw.size();
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

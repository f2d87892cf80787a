#include "fasta.hpp"

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

}  // namespace tetrex

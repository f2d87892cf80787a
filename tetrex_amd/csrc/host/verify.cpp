#include "verify.hpp"
#include "fasta.hpp"
#include "matcher.hpp"
#include "regex_front.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <sstream>
#include <stdexcept>

namespace tetrex {

namespace {

char complement(char c) {
    switch (c) {  // comp_tab of src/query.cpp:7-16 restricted to the IUPAC letters
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 'U': return 'A';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a'; case 'u': return 'a';
        case 'M': return 'K'; case 'K': return 'M'; case 'R': return 'Y'; case 'Y': return 'R';
        case 'V': return 'B'; case 'B': return 'V'; case 'H': return 'D'; case 'D': return 'H';
        default: return c;
    }
}

}  // namespace

size_t verify_bins(const std::vector<uint64_t>& bins, const std::vector<std::string>& bin_paths, const std::string& regex,
                   const KmerEncoder& enc, std::ostream& out, std::ostream& reverse_out, const VerifyOptions& opt) {
    const bool dna = enc.molecule() == Molecule::DNA;
    const bool reduced = !dna && enc.alphabet() != Alphabet::Base;
    std::string pattern = regex;
    if (reduced) pattern = reduce_query_alphabet(pattern, enc.reduce_table());
    pattern = "(" + pattern + ")";  // as the reference hands it to RE2 (include/query.h:103,148)
    // RE2 default syntax (leftmost-first) for DNA, RE2::POSIX (leftmost-longest) for peptides
    const Matcher rx(pattern, dna ? Matcher::Semantics::LeftmostFirst : Matcher::Semantics::LeftmostLongest);
    std::vector<std::string> fwd(bins.size()), rev(bins.size());
    std::vector<size_t> found(bins.size(), 0);
    std::string error;
#pragma omp parallel for schedule(dynamic) num_threads(opt.threads > 0 ? opt.threads : 1)
    for (size_t i = 0; i < bins.size(); ++i) {
        try {
            const std::string& path = bin_paths.at(bins[i]);
            std::ostringstream f, r;
            Matcher::Cache cache;  // lazily built automaton states: per thread
            for_each_record(path, [&](const FastaRecord& rec) {
                std::string seq = rec.seq;
                if (reduced) for (char& c : seq) c = enc.reduce((unsigned char)c);
                rx.find_all(seq, cache, [&](size_t s, size_t n) {
                    f << path << "\t>" << rec.name << "\t" << seq.substr(s, n) << "\t" << s << "," << s + n << "\n";
                    ++found[i];
                });
                if (dna) {
                    std::string rc(seq.rbegin(), seq.rend());
                    for (char& c : rc) c = complement(c);
                    rx.find_all(rc, cache, [&](size_t s, size_t n) {
                        r << path << "\t>" << rec.name << "\t" << rc.substr(s, n) << "\tREVERSE STRAND HIT\n";
                        ++found[i];
                    });
                }
            });
            fwd[i] = f.str();
            rev[i] = r.str();
        } catch (const std::exception& e) {
#pragma omp critical
            error = e.what();
        }
    }
    if (!error.empty()) throw std::runtime_error(error);
    size_t total = 0;
    for (size_t i = 0; i < bins.size(); ++i) {
        out << fwd[i];
        reverse_out << rev[i];
        total += found[i];
    }
    return total;
}

size_t verify_batch(const std::vector<const uint64_t*>& masks, uint64_t bins, const std::vector<std::string>& bin_paths,
                    const std::vector<std::string>& regexes, const KmerEncoder& enc, std::vector<std::string>* forward,
                    std::vector<std::string>* reverse, const VerifyOptions& opt) {
    const bool dna = enc.molecule() == Molecule::DNA;
    const bool reduced = !dna && enc.alphabet() != Alphabet::Base;
    const size_t nq = regexes.size();
    forward->assign(nq, std::string());
    reverse->assign(nq, std::string());
    std::vector<std::unique_ptr<Matcher>> rx(nq);
    for (size_t q = 0; q < nq; ++q) {
        if (!masks[q]) continue;
        std::string pattern = regexes[q];
        if (reduced) pattern = reduce_query_alphabet(pattern, enc.reduce_table());
        rx[q] = std::make_unique<Matcher>("(" + pattern + ")", dna ? Matcher::Semantics::LeftmostFirst : Matcher::Semantics::LeftmostLongest);
    }
    // bin -> the queries that selected it (ascending), for the bins anybody selected
    std::vector<std::vector<uint32_t>> wanted(bins);
    for (size_t q = 0; q < nq; ++q) {
        if (!masks[q]) continue;
        for (uint64_t w = 0; w * 64 < bins; ++w)
            for (uint64_t x = masks[q][w]; x; x &= x - 1) {
                const uint64_t b = w * 64 + (uint64_t)__builtin_ctzll(x);
                if (b < bins) wanted[b].push_back((uint32_t)q);
            }
    }
    std::vector<uint64_t> todo;
    for (uint64_t b = 0; b < bins; ++b)
        if (!wanted[b].empty()) todo.push_back(b);
    // per bin: the rows of each of its queries (bins are joined per query in ascending order afterwards)
    struct Rows { std::string fwd, rev; };
    std::vector<std::vector<Rows>> rows(todo.size());
    std::vector<size_t> found(todo.size(), 0);
    std::string error;
    // TETREX_TRACE: where the time goes (summed over the threads) and how much the required literals save
    const bool trace = std::getenv("TETREX_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_load = 0, t_literal = 0, t_match = 0;
    size_t n_pairs = 0, n_listed_pairs = 0, n_records_scanned = 0, n_records_total = 0;
    const double t_begin = now();
#pragma omp parallel num_threads(opt.threads > 0 ? opt.threads : 1)
    {
        std::vector<std::unique_ptr<Matcher::Cache>> caches(nq);  // this thread's automata, built on first use, kept from bin to bin
        RecordSet recs;                                            // the bin's records: sequences back to back, '\n' behind each
        std::string rc;                                            // ... and their reverse complements, laid out the same way (DNA)
        std::vector<size_t> hits;                                  // records that hold a motif's required literal
#pragma omp for schedule(dynamic)
        for (size_t i = 0; i < todo.size(); ++i) {
            try {
                const std::string& path = bin_paths.at(todo[i]);
                const std::vector<uint32_t>& qs = wanted[todo[i]];
                std::vector<Rows>& mine = rows[i];
                mine.resize(qs.size());
                const double t0 = trace ? now() : 0;
                load_records(path, recs);  // ONE read of the bin for all its motifs
                if (reduced)
                    for (char& c : recs.text) if (c != '\n') c = enc.reduce((unsigned char)c);
                if (dna) {
                    rc.resize(recs.text.size());
                    for (size_t r = 0; r < recs.size(); ++r) {
                        const size_t lo = recs.start[r], hi = recs.start[r + 1] - 1;  // [lo, hi): the sequence; text[hi] = '\n'
                        for (size_t p = lo; p < hi; ++p) rc[lo + (hi - 1 - p)] = complement(recs.text[p]);
                        rc[hi] = '\n';
                    }
                }
                // the records of `text` that can hold a match of m: all of them, or — where m has a required literal that is rare
                // enough — the ones the literal occurs in, found in one pass over the whole bin (no sequence holds the '\n' between
                // records, so an occurrence never straddles two)
                auto candidates = [&](const Matcher& m, const std::string& text) {
                    // the run of residue classes first (a literal is such a run; a motif of classes has no literal worth the search);
                    // not rare — in more than half of the records —: the per-record prefilter does as well
                    if (!m.required_run().empty()) return records_with_run(recs, text, m.required_run(), recs.size() / 2, hits);
                    const std::string& lit = m.required_literal();
                    return lit.size() >= 2 && records_with(recs, text, lit, recs.size() / 2, hits);
                };
                double my_literal = 0, my_match = 0;
                size_t my_listed = 0, my_scanned = 0;
                const double t1 = trace ? now() : 0;
                for (size_t j = 0; j < qs.size(); ++j) {
                    const uint32_t q = qs[j];
                    const Matcher& m = *rx[q];
                    if (!caches[q]) caches[q] = std::make_unique<Matcher::Cache>();
                    auto scan = [&](const std::string& text, bool reverse_strand) {
                        const double ta = trace ? now() : 0;
                        const bool listed = candidates(m, text);
                        const double tb = trace ? now() : 0;
                        my_literal += tb - ta;
                        const size_t n_scan = listed ? hits.size() : recs.size();
                        my_listed += listed;
                        my_scanned += n_scan;
                        for (size_t at = 0; at < n_scan; ++at) {
                            const size_t r = listed ? hits[at] : at;
                            const std::string_view seq(text.data() + recs.start[r], recs.start[r + 1] - recs.start[r] - 1);
                            m.find_all(seq, *caches[q], [&](size_t s, size_t n) {
                                std::string& o = reverse_strand ? mine[j].rev : mine[j].fwd;
                                o += path; o += "\t>"; o += recs.names[r]; o += '\t'; o.append(seq.data() + s, n);
                                if (reverse_strand) o += "\tREVERSE STRAND HIT\n";
                                else { o += '\t'; o += std::to_string(s); o += ','; o += std::to_string(s + n); o += '\n'; }
                                ++found[i];
                            });
                        }
                    };
                    scan(recs.text, false);
                    if (dna) scan(rc, true);
                }
                if (trace) {
                    my_match = now() - t1 - my_literal;
#pragma omp critical
                    {
                        t_load += t1 - t0; t_literal += my_literal; t_match += my_match;
                        n_pairs += qs.size(); n_listed_pairs += my_listed; n_records_scanned += my_scanned; n_records_total += recs.size() * qs.size();
                    }
                }
            } catch (const std::exception& e) {
#pragma omp critical
                error = e.what();
            }
        }
    }
    if (!error.empty()) throw std::runtime_error(error);
    if (trace)
        std::fprintf(stderr, "[tetrex] verify_batch: %zu bins, %zu (motif, bin) pairs in %.3f s; thread time: load %.3f s, literal search %.3f s, automata + rows %.3f s; "
                             "%zu pairs narrowed by their literal, %zu of %zu records given to the automata\n",
                     todo.size(), n_pairs, now() - t_begin, t_load, t_literal, t_match, n_listed_pairs, n_records_scanned, n_records_total);
    size_t total = 0;
    for (size_t i = 0; i < todo.size(); ++i) {
        const std::vector<uint32_t>& qs = wanted[todo[i]];
        for (size_t j = 0; j < qs.size(); ++j) {
            (*forward)[qs[j]] += rows[i][j].fwd;
            (*reverse)[qs[j]] += rows[i][j].rev;
        }
        total += found[i];
    }
    return total;
}

size_t verify_conjunction(const std::vector<uint64_t>& bins, const std::vector<std::string>& bin_paths,
                          const std::vector<std::string>& queries, std::ostream& out, const VerifyOptions& opt) {
    std::vector<Matcher> rxs;
    for (const auto& q : queries) rxs.emplace_back("(" + q + ")", Matcher::Semantics::LeftmostFirst);
    std::vector<std::string> rows(bins.size());
    std::vector<size_t> found(bins.size(), 0);
    std::string error;
#pragma omp parallel for schedule(dynamic) num_threads(opt.threads > 0 ? opt.threads : 1)
    for (size_t i = 0; i < bins.size(); ++i) {
        try {
            const std::string& path = bin_paths.at(bins[i]);
            std::ostringstream o;
            std::vector<Matcher::Cache> caches(rxs.size());
            for_each_record(path, [&](const FastaRecord& rec) {
                for (size_t q = 0; q < rxs.size(); ++q)
                    if (!rxs[q].contains(rec.seq, caches[q])) return;
                o << path << "\t>" << rec.name << "\tN --> ";
                for (const auto& q : queries) o << q << " --> ";
                o << "C\n";
                ++found[i];
            });
            rows[i] = o.str();
        } catch (const std::exception& e) {
#pragma omp critical
            error = e.what();
        }
    }
    if (!error.empty()) throw std::runtime_error(error);
    size_t total = 0;
    for (size_t i = 0; i < bins.size(); ++i) { out << rows[i]; total += found[i]; }
    return total;
}

}  // namespace tetrex

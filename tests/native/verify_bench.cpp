// Native harness (no GPU): host/verify.cpp verify_batch — a -f batch verified bin-major — on 1024 synthetic FASTA bins of 200 000
// residues with random candidate masks of 11 bins per motif (the shape of bench.py's end_to_end.with_verification leg), timed,
// and compared row for row with the motif-by-motif verify_bins.  Build and run:
//   g++ -O2 -std=c++20 -fopenmp -o /tmp/verify_bench tests/native/verify_bench.cpp tetrex_amd/csrc/host/{verify,fasta,matcher,regex_front,encoder}.cpp -lz
//   /tmp/verify_bench <threads> [motifs] [plain]
#include "../../tetrex_amd/csrc/host/verify.hpp"
#include "../../tetrex_amd/csrc/host/fasta.hpp"
#include <chrono>
#include <cstdio>
#include <fstream>
#include <cstdlib>
#include <random>
#include <string>
#include <sstream>
using namespace tetrex;
int main(int argc, char** argv) {
    const int bins = getenv("VERIFY_BENCH_BINS") ? atoi(getenv("VERIFY_BENCH_BINS")) : 1024;   // (a multiple of 64)
    const int per_bin = getenv("VERIFY_BENCH_PER_BIN") ? atoi(getenv("VERIFY_BENCH_PER_BIN")) : 200000;
    const int nq = argc > 2 ? atoi(argv[2]) : 200;
    const std::string dir = std::string("/tmp/tetrex_verify_bench_") + std::to_string(bins) + "_" + std::to_string(per_bin);
    const int threads = argc > 1 ? atoi(argv[1]) : 1;
    const bool plain = argc > 3;  // the bench leg's shape only (long motifs, 11 candidate bins each): timing; otherwise also short motifs that DO match
    std::mt19937_64 rng(11);
    (void)!system(("mkdir -p " + dir).c_str());
    const char* aa = "ACDEFGHIKLMNPQRSTVWY";
    std::vector<std::string> paths;
    for (int b = 0; b < bins; ++b) {
        char name[160]; snprintf(name, sizeof name, "%s/bin%04d.fa", dir.c_str(), b);
        paths.push_back(name);
        std::ifstream test(name);
        if (test.good()) continue;
        std::ofstream f(name);
        for (int i = 0, r = 0; i < per_bin; i += 360, ++r) {
            f << ">sp|" << b << "_" << r << "\n";
            for (int j = 0; j < 360 && i + j < per_bin; ++j) f << aa[rng() % 20];
            f << "\n";
        }
    }
    // motifs like bench.py's generator (5 % wildcards, 30 % classes, 2 % ranges, 8-14 long)
    std::vector<std::string> motifs;
    for (int q = 0; q < nq; ++q) {
        std::string m;
        const int len = 8 + rng() % 7;
        for (int i = 0; i < len; ++i) {
            const double r = (rng() % 10000) / 10000.0;
            if (i == 0 || i == len - 1) { m += aa[rng() % 20]; continue; }
            if (r < 0.05) m += '.';
            else if (r < 0.35) { m += '['; const int c = 2 + rng() % 4; for (int j = 0; j < c; ++j) m += aa[rng() % 20]; m += ']'; }
            else if (r < 0.37) m += ".{1,3}";
            else m += aa[rng() % 20];
        }
        motifs.push_back(!plain && q % 5 == 0 ? std::string(1, aa[q % 20]) + aa[(q / 5) % 20] + (q % 2 ? std::string(1, aa[(q * 7) % 20]) + ".K" : std::string("[DE]") + aa[(q * 3) % 20]) : m);
    }
    const uint64_t W = bins / 64;
    std::vector<uint64_t> masks((size_t)nq * W, 0);
    std::vector<const uint64_t*> mp(nq);
    for (int q = 0; q < nq; ++q) {
        for (int j = 0; j < 11; ++j) { const int b = rng() % bins; masks[q * W + b / 64] |= 1ULL << (b % 64); }
        if (!plain && q % 3 == 0) for (int b = 0; b < 64; ++b) masks[q * W] |= 1ULL << b;  // (more bins: some short motifs do match)
        mp[q] = masks.data() + q * W;
    }
    KmerEncoder enc(Molecule::Peptide, 6, Alphabet::Base);
    std::vector<std::string> fwd, rev;
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        const size_t found = verify_batch(mp, bins, paths, motifs, enc, &fwd, &rev, VerifyOptions{threads});
        printf("verify_batch: %.3f s, %zu matches\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), found);
    }
    {   // the same rows as the motif-by-motif path
        size_t bad = 0, nonempty = 0;
        for (int q = 0; q < nq; ++q) {
            std::vector<uint64_t> hit;
            for (int b = 0; b < bins; ++b) if ((masks[q * W + b / 64] >> (b % 64)) & 1) hit.push_back(b);
            std::ostringstream f, r;
            verify_bins(hit, paths, motifs[q], enc, f, r, VerifyOptions{threads});
            bad += f.str() != fwd[q] || r.str() != rev[q];
            nonempty += !fwd[q].empty();
        }
        printf("compared with verify_bins: %zu of %d motifs differ, %zu have rows\n", bad, nq, nonempty);
    }
    // parts
    auto t0 = std::chrono::steady_clock::now();
    RecordSet rs; size_t bytes = 0;
    for (auto& p : paths) { load_records(p, rs); bytes += rs.text.size(); }
    printf("load_records of all bins: %.3f s (%.1f MB)\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), bytes / 1e6);
}

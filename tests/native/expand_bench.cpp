// Native timing harness for the staged expansion (no GPU, no Python): expands the motifs of a text file
// against an executor that keeps every state alive and prints the time of five repetitions.
//   g++ -O3 -march=native -std=c++20 -pthread -o /tmp/expand_bench tests/native/expand_bench.cpp \
//       tetrex_amd/csrc/host/{encoder,regex_front,kgraph,compiler,staged}.cpp && /tmp/expand_bench motifs.txt 16
#include "../../tetrex_amd/csrc/host/compiler.hpp"
#include <chrono>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>
using namespace tetrex;
struct Fake final : StageExecutor {
    void stage(const uint8_t*, size_t, const std::vector<uint32_t>& qp, const std::vector<uint32_t>&, std::vector<uint8_t>& alive) override { alive.assign(qp.size(), 1); }
};
int main(int argc, char** argv) {
    std::vector<std::string> motifs; std::ifstream in(argv[1]); std::string l; while (std::getline(in, l)) if (!l.empty()) motifs.push_back(l);
    KmerEncoder enc(getenv("EB_DNA") ? Molecule::DNA : Molecule::Peptide, getenv("EB_K") ? (unsigned)atoi(getenv("EB_K")) : 4u, Alphabet::Base);  // EB_DNA: nucleotide motifs
    StagedOptions opt; opt.threads = argc > 2 ? atoi(argv[2]) : 1;
    if (getenv("EB_DENSE")) { opt.dense.enabled = true; opt.dense.slot_bytes = 128; }
    if (getenv("EB_TRACKED")) opt.dense.tracked_ok = true;  // the executor is assumed to keep live lists (tracked blocks)
    if (argc > 3) opt.ops_per_task = (size_t)atol(argv[3]);
    if (argc > 4) opt.ops_per_stage = (size_t)atol(argv[4]);
    for (int rep = 0; rep < 5; ++rep) {
        Fake exec; std::vector<int> st; std::vector<std::string> why;
        auto t0 = std::chrono::steady_clock::now();
        StagedStats s = run_staged(enc, 1024, motifs, exec, opt, &st, &why);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("threads %d: %.3f s total, expand %.3f s, ops %llu kmers %llu states %llu stages %zu\n", opt.threads, dt, s.expand_seconds, (unsigned long long)s.ops, (unsigned long long)s.kmers, (unsigned long long)s.states, s.stages);
    }
}

// ThreadSanitizer driver for the staged expansion (run by tools/tsan_cpu_test.sh): expands a batch of
// wildcard-rich motifs on 8 threads against an executor that declares every 7th frontier state dead,
// so the thread pool, the block cache and the parallel frontier/prune passes all run under TSan.
#include "../../tetrex_amd/csrc/host/compiler.hpp"

#include <cstdio>
#include <string>
#include <vector>

using namespace tetrex;

struct FakeExecutor final : StageExecutor {
    size_t stages = 0, bytes = 0;
    size_t kill_every = 7;  // 0: every state stays alive (queries stop asking and are expanded while a stage "executes")
    void stage(const uint8_t* blob, size_t blob_bytes, const std::vector<uint32_t>& qp, const std::vector<uint32_t>&,
               std::vector<uint8_t>& alive) override {
        alive.assign(qp.size(), 1);
        if (kill_every) for (size_t i = 0; i < alive.size(); i += kill_every) alive[i] = 0;
        volatile uint64_t sink = 0;  // pretend to be busy, so the overlapped expansion really overlaps
        for (size_t i = 0; i < blob_bytes; i += 4096) sink = sink + blob[i];
        ++stages;
        bytes += blob_bytes;
    }
};

int main() {
    const char* residues = "ACDEFGHIKLMNPQRSTVWY";
    std::vector<std::string> motifs;
    uint64_t x = 12345;
    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (int i = 0; i < 96; ++i) {
        std::string m;
        const int len = 5 + (int)(next() % 8);
        for (int j = 0; j < len; ++j) {
            const uint64_t r = next() % 10;
            if (r == 0) m += '.';
            else if (r == 1) m += ".{1,3}";
            else if (r == 2) { m += '['; for (int c = 0; c < 3; ++c) m += residues[next() % 20]; m += ']'; }
            else m += residues[next() % 20];
        }
        motifs.push_back(m);
    }
    KmerEncoder enc(Molecule::Peptide, 4, Alphabet::Base);
    FakeExecutor exec;
    StagedOptions opt;
    opt.threads = 8;
    opt.ops_per_query_per_stage = 512;
    opt.ops_per_task = 4096;
    opt.stage_target_ops = 1 << 16;
    std::vector<int> status;
    std::vector<std::string> why;
    const StagedStats st = run_staged(enc, 1024, motifs, exec, opt, &status, &why);
    size_t failed = 0;
    for (int s : status) failed += s != 0;
    std::printf("stages %zu ops %llu states %llu pruned %llu failed %zu blob bytes %zu\n", st.stages, (unsigned long long)st.ops,
                (unsigned long long)st.states, (unsigned long long)st.pruned, failed, exec.bytes);
    if (!(st.stages > 1 && st.pruned > 0)) return 1;
    FakeExecutor keep;
    keep.kill_every = 0;
    const StagedStats st2 = run_staged(enc, 1024, motifs, keep, opt, &status, &why);
    std::printf("all alive: stages %zu ops %llu states %llu pruned %llu\n", st2.stages, (unsigned long long)st2.ops, (unsigned long long)st2.states,
                (unsigned long long)st2.pruned);
    if (!(st2.stages > 1 && st2.pruned == 0)) return 1;
    // dense DP steps: the block budget is one atomic pool shared by the expansion threads; a small pool makes some
    // queries fall back to enumerated states while others hold blocks
    FakeExecutor dense_exec;
    dense_exec.kill_every = 0;
    StagedOptions dopt = opt;
    dopt.dense.enabled = true;
    dopt.dense_evidence = DenseOptions::kDense;  // the executor above answers a plain 1 (alive): nothing to learn fills from
    dopt.dense.slot_bytes = 128;
    dopt.dense_pool_bytes = (uint64_t)40 * 9261 * 128;  // room for 40 blocks among 96 queries
    const StagedStats st3 = run_staged(enc, 1024, motifs, dense_exec, dopt, &status, &why);
    failed = 0;
    for (int s : status) failed += s != 0;
    std::printf("dense: stages %zu ops %llu dense ops %llu failed %zu\n", st3.stages, (unsigned long long)st3.ops, (unsigned long long)st3.dense_ops, failed);
    return st3.dense_ops > 0 && failed == 0 && st3.ops < st2.ops ? 0 : 1;
}

// `tetrex` command line — product code.  Keeps the reference's surface for the query path:
//   tetrex query [-d] [-v] [-f] [-c] [-a] [-t N] [-o dest] [-g dibf] <index.ibf> <regex|->
//     (include/arg_parse.h:57-71, src/main.cpp:36-59, src/query.cpp:477-498)
//   tetrex index [-k K] [-p fpr] [-c hashes] [-t N] [-n] [-i] [-r murphy|li] <name> <libs...>
//     (include/arg_parse.h:10-38, src/index_base.cpp:73-117)
//   tetrex inspect <index.ibf>   (src/inspect_idx.cpp)
// The candidate-bin masks come from the GPU (libtxq.so); there is no CPU probe path.
#include "device_index.hpp"
#include "index_file.hpp"
#include "kgraph.hpp"
#include "regex_front.hpp"
#include "verify.hpp"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <future>
#include <iomanip>
#include <iostream>
#include <sstream>

using namespace tetrex;

namespace {

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Args {
    std::vector<std::string> pos;
    std::vector<std::pair<std::string, std::string>> opts;  // canonical long name -> value ("" for flags)
    bool has(const std::string& n) const {
        for (auto& o : opts) if (o.first == n) return true;
        return false;
    }
    std::string get(const std::string& n, const std::string& dflt) const {
        for (auto& o : opts) if (o.first == n) return o.second;
        return dflt;
    }
};

struct OptSpec { char s; const char* l; bool value; };

Args parse(int argc, char** argv, int from, const std::vector<OptSpec>& spec) {
    Args a;
    bool only_pos = false;
    for (int i = from; i < argc; ++i) {
        std::string t = argv[i];
        if (only_pos || t.size() < 2 || t[0] != '-' || t == "-") { a.pos.push_back(t); continue; }
        if (t == "--") { only_pos = true; continue; }
        const OptSpec* hit = nullptr;
        std::string inline_value;
        bool has_inline = false;
        if (t[1] == '-') {
            std::string name = t.substr(2);
            const size_t eq = name.find('=');
            if (eq != std::string::npos) { inline_value = name.substr(eq + 1); name = name.substr(0, eq); has_inline = true; }
            for (auto& s : spec) if (name == s.l) hit = &s;
        } else {
            for (auto& s : spec) if (t[1] == s.s) hit = &s;
            if (hit && t.size() > 2) {
                if (hit->value) { inline_value = t.substr(2); has_inline = true; }
                else {  // bundled flags: -vf
                    for (size_t j = 1; j < t.size(); ++j) {
                        const OptSpec* f = nullptr;
                        for (auto& s : spec) if (t[j] == s.s && !s.value) f = &s;
                        if (!f) throw std::runtime_error("Unknown option " + t);
                        a.opts.emplace_back(f->l, "");
                    }
                    continue;
                }
            }
        }
        if (!hit) throw std::runtime_error("Unknown option " + t);
        if (!hit->value) { a.opts.emplace_back(hit->l, ""); continue; }
        if (!has_inline) {
            if (i + 1 >= argc) throw std::runtime_error(std::string("Missing value for option --") + hit->l);
            inline_value = argv[++i];
        }
        a.opts.emplace_back(hit->l, inline_value);
    }
    return a;
}

std::vector<std::string> split(const std::string& s, char d) {
    std::vector<std::string> out;
    std::stringstream ss(s);
    for (std::string t; std::getline(ss, t, d);) out.push_back(t);
    return out;
}

size_t popcount_mask(const uint64_t* m, uint64_t words) {
    size_t n = 0;
    for (uint64_t w = 0; w < words; ++w) n += (size_t)__builtin_popcountll(m[w]);
    return n;
}

int cmd_query(int argc, char** argv) {
    const std::vector<OptSpec> spec = {{'d', "draw", false}, {'v', "verbose", false}, {'f', "file", false}, {'c', "conj", false},
                                       {'a', "augment", false}, {'t', "threads", true}, {'o', "output", true}, {'g', "gibf", true},
                                       {'D', "device", true}, {'S', "stats", false}, {'G', "gpus", true}, {'R', "shards", true}, {'M', "max-ops", true}};
    Args a;
    try {
        a = parse(argc, argv, 2, spec);
        if (a.pos.size() != 2) throw std::runtime_error("expected <index> <regex>");
    } catch (const std::exception& e) {
        std::cerr << "[Error TetRex Query module " << e.what() << "\n";
        return 0;  // the reference returns normally after a parser error (src/main.cpp:42-46)
    }
    const int threads = std::max(1, std::atoi(a.get("threads", "1").c_str()));
    bool verbose = a.has("verbose");
    const bool from_file = a.has("file"), conj = a.has("conj");
    std::string dest = a.get("output", "-");
    std::string input = a.pos[1];
    if (input == "-") std::cin >> input;

    const bool trace = std::getenv("TETREX_TRACE") != nullptr;
    const double t_start = now();
    // -D d[,d..] / --gpus N (not in the reference): the index's bins are cut into column shards, one per device (--shards R:
    // that many shards, dealt round-robin over the devices); one frontier expansion drives all shards, masks are joined on the host
    DeviceIndex dev;
    std::vector<int> devices;
    for (const std::string& d : split(a.get("device", "0"), ',')) devices.push_back(std::atoi(d.c_str()));
    if (a.has("gpus")) {
        devices.clear();
        for (int d = 0; d < std::max(1, std::atoi(a.get("gpus", "1").c_str())); ++d) devices.push_back(d);
    }
    const int n_shards = a.has("shards") ? std::max(1, std::atoi(a.get("shards", "1").c_str())) : (int)devices.size();
    // HIP start-up (~0.5 s) runs beside the mapping and parsing of the index file
    std::future<void> hip_ready = std::async(std::launch::async, [&devices]() { DeviceIndex::warm_up(devices); });
    IndexImage image;
    try {
        image = read_index_file(a.pos[0]);
    } catch (const std::exception& e) {
        try { hip_ready.get(); } catch (...) {}
        std::cerr << "Filepath to (H)IBF Index not valid" << std::endl;
        std::cerr << e.what() << '\n';
        return 0;
    }
    const double t_read = now();
    hip_ready.get();
    const double t_hip = now();
    if (n_shards > 1 || devices.size() > 1) dev.upload_sharded(image, devices, n_shards);
    else dev.upload(image, devices[0]);
    if (trace) std::cerr << "[tetrex] index map+parse " << (t_read - t_start) << " s, waited for HIP " << (t_hip - t_read) << " s, upload " << (now() - t_hip) << " s" << std::endl;
    if (a.has("gibf")) dev.attach_dgram(read_dgram_index_file(a.get("gibf", "")));  // include/query.h:259-264
    StagedOptions sopt;
    sopt.gaps.augment = a.has("augment");
    if (a.has("max-ops")) {  // -M / --max-ops (not in the reference): mask operations one query may expand to before it is given up (default 2^33)
        const long long m = std::atoll(a.get("max-ops", "0").c_str());
        if (m > 0) sopt.limits.max_ops = sopt.limits.max_states = (size_t)m;
    }
    const KmerEncoder enc = dev.encoder();
    const uint64_t bins = dev.bins(), W = dev.result_words();
    const VerifyOptions vopt{threads};
    // -S/--stats (not in the reference): one JSON line on stderr about the candidate-mask stage
    auto print_stats = [&](const StagedStats& st, size_t queries, double seconds) {
        if (!a.has("stats")) return;
        std::cerr << "{\"queries\": " << queries << ", \"mask_seconds\": " << seconds << ", \"stages\": " << st.stages
                  << ", \"ops\": " << st.ops << ", \"kmer_probes\": " << st.kmers << ", \"states\": " << st.states
                  << ", \"pruned_states\": " << st.pruned << ", \"dense_ops\": " << st.dense_ops << ", \"expand_seconds\": " << st.expand_seconds
                  << ", \"execute_seconds\": " << st.execute_seconds << ", \"bins\": " << bins << "}" << std::endl;
    };

    auto run_one = [&](const std::string& rx, const uint64_t* mask, const std::string& destination, bool log_file_mode, double t1) {
        const size_t narrowed = popcount_mask(mask, W);
        if (verbose) std::cerr << "Narrowed Search to " << narrowed << " possible bins" << std::endl;
        if (log_file_mode) std::cerr << "Bin Count: " << narrowed << "\t";
        if (narrowed) {
            try {
                const std::vector<uint64_t> hit = set_bins(mask, bins);
                if (destination == "-") verify_bins(hit, image.bin_paths, rx, enc, std::cout, std::cout, vopt);
                else {
                    std::ofstream f(destination);
                    if (!f) throw std::runtime_error("Failed to open output file: " + destination);
                    verify_bins(hit, image.bin_paths, rx, enc, f, std::cout, vopt);
                }
            } catch (const std::exception& e) {
                std::cerr << e.what() << '\n';
            }
        }
        if (verbose || log_file_mode) std::cerr << "Query Time: " << (now() - t1) << std::endl;
    };

    if (bins <= 1)
        std::cerr << "[WARNING] Index contains only 1 bin. Unable to accelerate search using the TetRex algorithm. Performing Linear Scan" << std::endl;

    if (from_file) {  // TSV: id <tab> motif; results go to <id>.tsv (src/query.cpp:342-363, include/query.h:329-339)
        verbose = false;
        std::ifstream in(input);
        if (!in) throw std::runtime_error("Could not open file: " + input);
        std::vector<std::string> ids, motifs;
        for (std::string line; std::getline(in, line);) {
            if (line.empty()) continue;
            const std::vector<std::string> f = split(line, '\t');
            if (f.size() >= 2) { ids.push_back(f[0]); motifs.push_back(f[1]); }
        }
        const double t0 = now();
        std::vector<int> status;
        std::vector<std::string> why;
        StagedStats st;
        const std::vector<uint64_t> masks = dev.query_masks(motifs, &status, &why, &st, &sopt);
        print_stats(st, motifs.size(), now() - t0);
        const double batch = (now() - t0) / std::max<size_t>(1, motifs.size());
        const double t_verify = now();
        int failed = 0;
        // The batch is verified BIN-MAJOR (verify_batch): every candidate bin is read once and all the motifs that selected it run
        // over its records; the reference — and TETREX_VERIFY_PER_MOTIF=1 here, for comparison — verifies motif by motif
        // (include/query.h:329-346), re-reading a bin once per motif.  Files and rows are the same either way.
        std::vector<std::string> fwd, rev;
        bool bin_major = !std::getenv("TETREX_VERIFY_PER_MOTIF");
        if (bin_major) {
            std::vector<const uint64_t*> mptr(motifs.size(), nullptr);
            for (size_t i = 0; i < motifs.size(); ++i)
                if (!status[i]) mptr[i] = masks.data() + i * W;
            try {
                verify_batch(mptr, bins, image.bin_paths, motifs, enc, &fwd, &rev, vopt);
            } catch (const std::exception& e) {  // (a bin that cannot be read: motif by motif, so that the others still get their results)
                std::cerr << e.what() << '\n';
                bin_major = false;
            }
        }
        const double per_motif = bin_major ? (now() - t_verify) / std::max<size_t>(1, motifs.size()) : 0.0;
        for (size_t i = 0; i < motifs.size(); ++i) {
            std::cerr << ids[i] << "\t";
            if (status[i]) {  // its mask is incomplete: verifying the bins it happens to hold would silently lose matches
                std::cerr << "[Error] query not searchable, no result written: " << why[i] << std::endl;
                ++failed;
                continue;
            }
            if (!bin_major) { run_one(motifs[i], masks.data() + i * W, ids[i] + ".tsv", true, now() - batch); continue; }
            const size_t narrowed = popcount_mask(masks.data() + i * W, W);
            std::cerr << "Bin Count: " << narrowed << "\t";
            if (narrowed) {
                std::ofstream f(ids[i] + ".tsv");
                if (!f) std::cerr << "Failed to open output file: " << ids[i] << ".tsv" << '\n';
                else f << fwd[i];
                std::cout << rev[i];
            }
            std::cerr << "Query Time: " << (batch + per_motif) << std::endl;  // (the batch's time, shared out evenly)
        }
        // -S: the whole batch as the reference times a query — from after the index is loaded to the last output byte
        // (include/query.h:256,287-289): candidate masks + verification of the candidate bins
        if (a.has("stats"))
            std::cerr << "{\"batch_seconds\": " << (now() - t0) << ", \"verify_seconds\": " << (now() - t_verify) << ", \"threads\": " << vopt.threads
                      << ", \"refused\": " << failed << "}" << std::endl;
        return failed ? 1 : 0;
    }
    if (conj) {
        const std::vector<std::string> queries = split(input, ':');
        if (queries.size() == 1) { std::cerr << "Did you use the correct delimiter (:)?" << std::endl; return 0; }
        const double t1 = now();
        StagedStats st;
        std::vector<int> status;
        std::vector<std::string> why;
        const std::vector<uint64_t> masks = dev.query_masks(queries, &status, &why, &st, &sopt);
        print_stats(st, queries.size(), now() - t1);
        for (size_t q = 0; q < queries.size(); ++q)
            if (status[q]) {  // ANDing a partial mask would drop true candidate bins without a word
                std::cerr << "[Error] query not searchable: " << queries[q] << ": " << why[q] << std::endl;
                return 1;
            }
        std::vector<uint64_t> all(masks.begin(), masks.begin() + W);
        for (size_t q = 1; q < queries.size(); ++q)
            for (uint64_t w = 0; w < W; ++w) all[w] &= masks[q * W + w];
        if (verbose) std::cerr << "Narrowed Search to " << popcount_mask(all.data(), W) << " possible bins" << std::endl;
        if (popcount_mask(all.data(), W)) verify_conjunction(set_bins(all.data(), bins), image.bin_paths, queries, std::cout, vopt);
        if (verbose) std::cerr << "Query Time: " << (now() - t1) << std::endl;
        return 0;
    }
    if (a.has("draw")) {  // include/query.h:244: kgraph_visualizer.gv in the working directory
        try {
            KGraph g = build_kgraph(preprocess_query(input, enc), enc.k(), enc.alphabet() != Alphabet::Base);
            if (sopt.gaps.augment) g.augment();
            std::ofstream("kgraph_visualizer.gv") << g.to_graphviz();
        } catch (const std::exception& e) {
            std::cerr << "[WARNING] could not draw the k-graph: " << e.what() << std::endl;
        }
    }
    const double t1 = now();
    std::vector<int> status;
    std::vector<std::string> why;
    StagedStats st;
    const std::vector<uint64_t> masks = dev.query_masks({input}, &status, &why, &st, &sopt);
    print_stats(st, 1, now() - t1);
    if (status[0]) { std::cerr << "[Error] query not searchable: " << why[0] << std::endl; return 1; }
    run_one(input, masks.data(), dest, false, t1);
    return 0;
}

int cmd_index(int argc, char** argv) {
    const std::vector<OptSpec> spec = {{'k', "ksize", true}, {'p', "fpr", true}, {'c', "hash_count", true}, {'t', "threads", true},
                                       {'n', "nucleic_acid", false}, {'i', "ibf", false}, {'r', "reduce", true}, {'D', "device", true},
                                       {'W', "no-wraparound", false}};
    Args a;
    try {
        a = parse(argc, argv, 2, spec);
        if (a.pos.size() < 2) throw std::runtime_error("expected <name> <libraries...>");
    } catch (const std::exception& e) {
        std::cerr << "[Indexing Parser Error] " << e.what() << "\n";
        return 0;
    }
    BuildOptions opt;
    opt.k = (unsigned)std::atoi(a.get("ksize", "6").c_str());
    opt.fpr = std::strtof(a.get("fpr", "0.05").c_str(), nullptr);
    opt.hash_count = (unsigned)std::atoi(a.get("hash_count", "3").c_str());
    opt.dna = a.has("nucleic_acid");
    opt.hibf = !a.has("ibf");
    opt.dna_wraparound = !a.has("no-wraparound");
    opt.device = std::atoi(a.get("device", "0").c_str());
    const std::string red = a.get("reduce", "None");
    if (red == "murphy") opt.reduction = 1;
    else if (red == "li") opt.reduction = 2;
    else if (red != "None") { std::cerr << "[Indexing Parser Error] reduce must be murphy or li\n"; return 0; }
    if (!opt.dna && opt.k > 12) { std::cerr << "[Indexing Parser Error] Max kmer size for amino acids is 12" << "\n"; return 0; }
    std::vector<std::string> files;
    for (size_t i = 1; i < a.pos.size(); ++i) {
        const std::filesystem::path p = a.pos[i];
        if (p.extension() == ".lst") {
            std::ifstream in(p);
            if (!in) throw std::runtime_error("Could not open file " + p.string() + " for reading.");
            for (std::string line; std::getline(in, line);) files.push_back(line);
        } else files.push_back(std::filesystem::absolute(p).string());
    }
    size_t seqs = 0;
    const IndexImage img = build_index(files, opt, &seqs);
    std::cerr << "Indexed " << seqs << " sequences across " << files.size() << " bins." << std::endl;
    if (files.size() == 1)
        std::cerr << "[WARNING] The indexed reference library was not split into bins. The TetRex runtime will be significantly slower." << std::endl;
    std::cerr << "Writing to disk... ";
    write_index_file(a.pos[0] + ".ibf", img);
    std::cerr << "DONE" << std::endl;
    return 0;
}

// tetrex track [-l lower] [-u upper] [-n] [-i] <name> <libs...>   (include/arg_parse.h:94-122, src/dGramIndex.cpp:22-38)
int cmd_track(int argc, char** argv) {
    const std::vector<OptSpec> spec = {{'l', "lower", true}, {'u', "upper", true}, {'n', "nucleic_acid", false}, {'i', "ibf", false},
                                       {'D', "device", true}};
    Args a;
    try {
        a = parse(argc, argv, 2, spec);
        if (a.pos.size() < 2) throw std::runtime_error("expected <name> <libraries...>");
    } catch (const std::exception& e) {
        std::cerr << "[Error TetRex Dgram Indexing module " << e.what() << "\n";
        return 0;
    }
    std::vector<std::string> files;
    for (size_t i = 1; i < a.pos.size(); ++i) {
        const std::filesystem::path p = a.pos[i];
        if (p.extension() == ".lst") {
            std::ifstream in(p);
            if (!in) throw std::runtime_error("Could not open file " + p.string() + " for reading.");
            for (std::string line; std::getline(in, line);) files.push_back(line);
        } else files.push_back(std::filesystem::absolute(p).string());
    }
    const DgramImage d = build_dgram_index(files, std::strtoull(a.get("lower", "3").c_str(), nullptr, 10),
                                           std::strtoull(a.get("upper", "21").c_str(), nullptr, 10), 3, 0.05f,
                                           std::atoi(a.get("device", "0").c_str()));
    write_dgram_index_file(a.pos[0], d);  // the reference writes the name as given, no extension
    return 0;
}

int cmd_inspect(int argc, char** argv) {
    if (argc != 3) { std::cerr << "[Error TetRex Index Inspection module expected <index>\n"; return 0; }
    std::cerr << "Reading Index from Disk... ";
    const double t1 = now();
    const IndexImage ix = read_index_file(argv[2]);
    std::cerr << "DONE in " << (now() - t1) << "s" << std::endl;
    const bool dna = ix.molecule == "na";
    if (ix.is_hibf) {
        std::cout << "INDEX TYPE: HIBF" << std::endl;
        std::cout << "FALSE POSITIVE RATE: " << std::fixed << std::setprecision(2) << ix.fpr << std::endl;
    } else {
        std::cout << "INDEX TYPE: IBF" << std::endl;
        std::cout << "BIN COUNT (BFs): " << ix.ibf.bins << std::endl;
        std::cout << "BIN SIZE (bits): " << ix.ibf.bin_size << std::endl;
    }
    std::cout << "HASH COUNT (hash functions): " << unsigned(ix.hash_count) << std::endl;
    std::cout << "KMER LENGTH (" << (dna ? "bases" : "residues") << "): " << unsigned(ix.k) << std::endl;
    std::cout << "MOLECULE TYPE (alphabet): " << (dna ? "Nucleic Acid" : "Amino Acid") << " [REDUCTION=";
    if (dna && ix.is_hibf) std::cout << "NONE";
    else std::cout << unsigned(ix.reduction);  // the reference streams the uint8_t itself
    std::cout << "]" << std::endl;
    std::cout << "ACID LIBRARY (filepaths):" << std::endl;
    for (const auto& p : ix.bin_paths) std::cout << "\t- " << p << std::endl;
    std::cerr << "DONE" << std::endl;
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    try {
        if (argc < 2) { std::cerr << "[Error] usage: tetrex {index|query|inspect} ...\n"; return -1; }
        const std::string sub = argv[1];
        if (sub == "query") return cmd_query(argc, argv);
        if (sub == "index") return cmd_index(argc, argv);
        if (sub == "inspect") return cmd_inspect(argc, argv);
        if (sub == "track") return cmd_track(argc, argv);
        std::cerr << "[Error] unknown sub-command " << sub << "\n";
        return -1;
    } catch (const std::exception& e) {
        std::cerr << "[Error] " << e.what() << std::endl;
        return 1;
    }
}

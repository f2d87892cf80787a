#include "device_index.hpp"
#include "fasta.hpp"
#include "kgraph.hpp"
#include "regex_front.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <future>
#include <stdexcept>

namespace tetrex {

void txq_check(int rc, const char* what) {
    if (rc != TXQ_OK) throw std::runtime_error(std::string(what) + ": " + txq_last_error());
}

static void ensure_devices(const std::vector<int>& devices) {
    static std::vector<int> bound;
    if (bound == devices) return;
    txq_check(txq_init((int)devices.size(), devices.data()), "txq_init");
    bound = devices;
}
static void ensure_device(int device) { ensure_devices(std::vector<int>{device}); }

void DeviceIndex::warm_up(const std::vector<int>& devices) { ensure_devices(devices); }

static txq_ibf_desc describe(const IbfImage& f) {
    return txq_ibf_desc{f.bins, f.tech_bins, f.bin_size, f.hash_shift, f.bin_words, f.hash_funs, f.word_data()};  // possibly a view into the mapped index file
}

DeviceIndex::~DeviceIndex() {
    for (txq_index* a : aux_shards_) txq_index_free(a);
    for (txq_index* s : shards_) txq_index_free(s);
}

void DeviceIndex::attach_dgram(const DgramImage& dgram) {
    if (!ix_) throw std::runtime_error("index not uploaded");
    if (dgram.ibf.bins != info_.user_bins) throw std::runtime_error("the d-gram index was built over a different number of bins");
    for (txq_index* a : aux_shards_) txq_index_free(a);
    aux_shards_.clear();
    aux_ = nullptr;
    txq_ibf_desc d = describe(dgram.ibf);
    txq_index_desc desc{1, &d, nullptr, nullptr, dgram.ibf.bins};
    for (size_t r = 0; r < shards_.size(); ++r) {  // same shard, same device as the main index's shard
        txq_index* a = nullptr;
        txq_check(txq_index_upload(&desc, shards_.size() > 1 ? (int)r : shard_rank_, n_shards_, &a), "txq_index_upload(d-gram)");
        aux_shards_.push_back(a);
    }
    aux_ = aux_shards_[0];
    dgram_min_ = dgram.min_gap;
    dgram_max_ = dgram.max_gap;
}

txq_index* DeviceIndex::upload_one(const IndexImage& image, int shard_rank, int n_shards) {
    txq_index* ix = nullptr;
    if (!image.is_hibf) {
        txq_ibf_desc d = describe(image.ibf);
        txq_index_desc desc{1, &d, nullptr, nullptr, image.ibf.bins};
        txq_check(txq_index_upload(&desc, shard_rank, n_shards, &ix), "txq_index_upload");
    } else {
        const HibfImage& h = image.hibf;
        std::vector<txq_ibf_desc> ds;
        std::vector<const uint64_t*> nx, tb;
        for (size_t i = 0; i < h.ibfs.size(); ++i) {
            ds.push_back(describe(h.ibfs[i]));
            nx.push_back(h.next_ibf_id[i].data());
            tb.push_back(h.tb_to_user_bin[i].data());
        }
        txq_index_desc desc{ds.size(), ds.data(), nx.data(), tb.data(), h.user_bins};
        // (a general tree is sharded by sub-trees — full-width masks, ORed —, a regular two-level one by mask columns: the library decides)
        txq_check(txq_index_upload_subtrees(&desc, shard_rank, n_shards, &ix), "txq_index_upload_subtrees");
    }
    return ix;
}

void DeviceIndex::upload(const IndexImage& image, int device, int shard_rank, int n_shards) {
    ensure_device(device);
    for (txq_index* a : aux_shards_) txq_index_free(a);
    for (txq_index* s : shards_) txq_index_free(s);
    aux_shards_.clear();
    shards_.clear();
    aux_ = ix_ = nullptr;
    shard_rank_ = shard_rank;
    n_shards_ = n_shards;
    enc_ = KmerEncoder(image.molecule == "na" ? Molecule::DNA : Molecule::Peptide, image.k, (Alphabet)image.reduction);
    // the library deals shards over its devices by rank; with one device every rank lands on it
    ix_ = upload_one(image, shard_rank, n_shards);
    shards_.push_back(ix_);
    txq_check(txq_index_get_info(ix_, &info_), "txq_index_get_info");
}

void DeviceIndex::upload_sharded(const IndexImage& image, const std::vector<int>& devices, int n_shards) {
    if (devices.empty() || n_shards < 1) throw std::runtime_error("upload_sharded needs at least one device and one shard");
    ensure_devices(devices);
    for (txq_index* a : aux_shards_) txq_index_free(a);
    for (txq_index* s : shards_) txq_index_free(s);
    aux_shards_.clear();
    shards_.clear();
    aux_ = ix_ = nullptr;
    shard_rank_ = 0;
    n_shards_ = n_shards;
    enc_ = KmerEncoder(image.molecule == "na" ? Molecule::DNA : Molecule::Peptide, image.k, (Alphabet)image.reduction);
    for (int r = 0; r < n_shards; ++r) shards_.push_back(upload_one(image, r, n_shards));
    ix_ = shards_[0];
    txq_check(txq_index_get_info(ix_, &info_), "txq_index_get_info");
}

TxqStageExecutor::TxqStageExecutor(txq_index* ix, size_t n_programs, txq_index* aux) {
    txq_check(txq_session_begin(ix, n_programs, &session_), "txq_session_begin");
    if (aux) {
        const int rc = txq_session_set_aux_index(session_, aux);
        if (rc != TXQ_OK) {
            txq_session_end(session_, nullptr);
            session_ = nullptr;
            txq_check(rc, "txq_session_set_aux_index");
        }
    }
}
TxqStageExecutor::~TxqStageExecutor() {
    if (session_) txq_session_end(session_, nullptr);
}
void TxqStageExecutor::stage(const uint8_t* blob, size_t blob_bytes, const std::vector<uint32_t>& qp, const std::vector<uint32_t>& qs,
                             std::vector<uint8_t>& alive) {
    alive.assign(qp.size(), 1);
    txq_check(txq_session_stage(session_, blob, blob_bytes, qp.data(), qs.data(), qp.size(), alive.data()), "txq_session_stage");
}
void TxqStageExecutor::finish(uint64_t* masks) {
    txq_session* s = session_;
    session_ = nullptr;
    txq_check(txq_session_end(s, masks), "txq_session_end");
}

ShardedStageExecutor::ShardedStageExecutor(const std::vector<txq_index*>& shards, size_t n_programs, const std::vector<txq_index*>& aux)
    : n_programs_(n_programs) {
    if (shards.empty()) throw std::runtime_error("no shards");
    if (!aux.empty() && aux.size() != shards.size()) throw std::runtime_error("the d-gram index must be sharded like the main index");
    try {
        for (size_t r = 0; r < shards.size(); ++r) {
            txq_index_info info{};
            txq_check(txq_index_get_info(shards[r], &info), "txq_index_get_info");
            info_.push_back(info);
            txq_session* s = nullptr;
            txq_check(txq_session_begin(shards[r], n_programs, &s), "txq_session_begin");
            sessions_.push_back(s);
            if (!aux.empty()) txq_check(txq_session_set_aux_index(s, aux[r]), "txq_session_set_aux_index");
        }
    } catch (...) {
        for (txq_session* s : sessions_) txq_session_end(s, nullptr);
        throw;
    }
}
ShardedStageExecutor::~ShardedStageExecutor() {
    for (txq_session* s : sessions_)
        if (s) txq_session_end(s, nullptr);
}
void ShardedStageExecutor::stage(const uint8_t* blob, size_t blob_bytes, const std::vector<uint32_t>& qp, const std::vector<uint32_t>& qs,
                                 std::vector<uint8_t>& alive) {
    const size_t R = sessions_.size(), nq = qp.size();
    std::vector<std::vector<uint8_t>> answers(R, std::vector<uint8_t>(nq, 0));
    std::vector<std::string> errors(R);
    auto run = [&](size_t r) {  // txq_last_error() is per thread: fetch it on the thread that made the call
        if (txq_session_stage(sessions_[r], blob, blob_bytes, qp.data(), qs.data(), nq, answers[r].data()) != TXQ_OK)
            errors[r] = std::string("txq_session_stage (shard ") + std::to_string(r) + "): " + txq_last_error();
    };
    std::vector<std::future<void>> others;
    for (size_t r = 1; r < R; ++r) others.push_back(std::async(std::launch::async, run, r));
    run(0);
    for (auto& f : others) f.get();
    for (const std::string& e : errors)
        if (!e.empty()) throw std::runtime_error(e);
    // an answer is 0 (no bit) or 1 + floor(log2(bits set in the shard's columns)): the shards' counts add up
    alive.assign(nq, 0);
    for (size_t i = 0; i < nq; ++i) {
        uint64_t bits = 0;
        for (size_t r = 0; r < R; ++r) {
            const uint8_t b = answers[r][i];
            bits += b == 0 ? 0 : b == 1 ? 1 : (3ULL << (b - 2));  // the middle of [2^(b-1), 2^b)
        }
        alive[i] = bits ? (uint8_t)(64 - __builtin_clzll(bits)) : 0;
    }
}
std::vector<uint64_t> ShardedStageExecutor::finish() {
    const size_t R = sessions_.size();
    std::vector<std::vector<uint64_t>> part(R);
    std::vector<uint64_t> word0(R), words(R);
    std::vector<const uint64_t*> ptr(R);
    std::string error;
    for (size_t r = 0; r < R; ++r) {
        part[r].resize(n_programs_ * info_[r].shard_words + 1);
        txq_session* s = sessions_[r];
        sessions_[r] = nullptr;
        if (txq_session_end(s, part[r].data()) != TXQ_OK && error.empty()) error = std::string("txq_session_end: ") + txq_last_error();
        word0[r] = info_[r].shard_word0;
        words[r] = info_[r].shard_words;
        ptr[r] = part[r].data();
    }
    if (!error.empty()) throw std::runtime_error(error);
    if (info_[0].join_or) {  // sub-tree shards of a general HIBF: full-width masks, ORed (a split bin may straddle shards)
        const uint64_t mw = info_[0].mask_words;
        std::vector<uint64_t> full(n_programs_ * mw, 0);
        for (size_t r = 0; r < R; ++r) {
            if (!info_[r].join_or || info_[r].shard_words != mw) throw std::runtime_error("shards of different kinds in one join");
            for (size_t i = 0; i < n_programs_ * mw; ++i) full[i] |= part[r][i];
        }
        return full;
    }
    return join_shard_masks(n_programs_, info_[0].mask_words, word0, words, ptr);
}

std::vector<uint64_t> run_queries_sharded(const std::vector<txq_index*>& shards, const KmerEncoder& enc, const std::vector<std::string>& regexes,
                                          std::vector<int>* status, std::vector<std::string>* messages, StagedStats* stats,
                                          const StagedOptions* options, const std::vector<txq_index*>& aux) {
    if (shards.empty()) throw std::runtime_error("no shards");
    txq_index_info info{};
    txq_check(txq_index_get_info(shards[0], &info), "txq_index_get_info");
    if (status) status->assign(regexes.size(), 0);
    if (messages) messages->assign(regexes.size(), std::string());
    if (regexes.empty()) return std::vector<uint64_t>();
    ShardedStageExecutor exec(shards, regexes.size(), aux);
    StagedOptions opt = options ? *options : StagedOptions{};
    opt.dense.enabled = true;
    opt.dense.tracked_ok = true;
    opt.dense.slot_bytes = 0;
    for (txq_index* s : shards) {  // dense steps only where every shard can run them; budgets by the widest shard
        txq_index_info i{};
        txq_check(txq_index_get_info(s, &i), "txq_index_get_info");
        if (i.shard_words) opt.dense.enabled = opt.dense.enabled && txq_index_supports_dense(s) != 0;
        if (i.shard_words) opt.dense.tracked_ok = opt.dense.tracked_ok && txq_index_supports_dense(s) == 2;
        opt.dense.slot_bytes = std::max<uint64_t>(opt.dense.slot_bytes, i.shard_words * 8);
    }
    uint64_t tag = 0;
    (void)txq_index_get_tag(shards[0], &tag);
    if (opt.dense_evidence == DenseOptions::kUnknown) opt.dense_evidence = (int)(tag & 3);  // what earlier runs learned about the index
    const StagedStats st = run_staged(enc, info.user_bins, regexes, exec, opt, status, messages);
    if (st.dense_evidence != DenseOptions::kUnknown) (void)txq_index_set_tag(shards[0], (tag & ~(uint64_t)3) | (uint64_t)st.dense_evidence);
    if (stats) *stats = st;
    return exec.finish();
}

std::vector<uint64_t> run_queries(txq_index* ix, const KmerEncoder& enc, const std::vector<std::string>& regexes,
                                  std::vector<int>* status, std::vector<std::string>* messages, StagedStats* stats,
                                  const StagedOptions* options, txq_index* aux, uint64_t* into) {
    txq_index_info info{};
    txq_check(txq_index_get_info(ix, &info), "txq_index_get_info");
    // into: the caller's own n x shard_words words (a binding's array): the masks go there and nothing is returned — 10 000 masks
    // of 8192 bins are 10 MB that would otherwise be zeroed, filled and copied once more
    std::vector<uint64_t> masks(into ? 0 : regexes.size() * info.shard_words);
    if (status) status->assign(regexes.size(), 0);
    if (messages) messages->assign(regexes.size(), std::string());
    if (regexes.empty()) return masks;
    const bool trace = std::getenv("TETREX_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    TxqStageExecutor exec(ix, regexes.size(), aux);
    if (trace)
        std::fprintf(stderr, "[tetrex] session begin %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    StagedOptions opt = options ? *options : StagedOptions{};
    // saturated state lists run as dense DP steps on the device where the index allows it (TETREX_DENSE=0 switches them off)
    opt.dense.enabled = txq_index_supports_dense(ix) != 0;
    opt.dense.tracked_ok = txq_index_supports_dense(ix) == 2;  // fused steps: the session keeps live lists (tracked programs)
    opt.dense.slot_bytes = info.shard_words * 8;
    opt.feedback_bins = std::min<uint64_t>(info.user_bins, info.shard_words * 64);
    {   // dense blocks may take three quarters of what the device has left (and of what the index kept from earlier sessions),
        // not a constant that ignores a 64 GB index next to them
        uint64_t free_b = 0, kept_b = 0;
        if (txq_index_memory(ix, &free_b, &kept_b) == TXQ_OK) opt.dense_pool_bytes = std::min<uint64_t>(opt.dense_pool_bytes, (free_b + kept_b) / 4 * 3);
    }
    uint64_t tag = 0;
    (void)txq_index_get_tag(ix, &tag);
    if (opt.dense_evidence == DenseOptions::kUnknown) opt.dense_evidence = (int)(tag & 3);  // what earlier runs learned about the index
    const auto t0 = std::chrono::steady_clock::now();
    const StagedStats st = run_staged(enc, info.user_bins, regexes, exec, opt, status, messages);
    if (st.dense_evidence != DenseOptions::kUnknown) (void)txq_index_set_tag(ix, (tag & ~(uint64_t)3) | (uint64_t)st.dense_evidence);
    if (stats) *stats = st;
    const auto t1 = std::chrono::steady_clock::now();
    exec.finish(into ? into : masks.data());  // waits for the device: a stage without feedback questions returns as soon as it is launched
    if (trace)
        std::fprintf(stderr, "[tetrex] run_staged %.2f ms, finish (device drain + result copy) %.2f ms\n",
                     std::chrono::duration<double, std::milli>(t1 - t0).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    return masks;
}

std::vector<uint64_t> DeviceIndex::query_masks(const std::vector<std::string>& regexes, std::vector<int>* status,
                                               std::vector<std::string>* messages, StagedStats* stats, const StagedOptions* options) {
    if (!ix_) throw std::runtime_error("index not uploaded");
    StagedOptions opt = options ? *options : StagedOptions{};
    if (aux_) {
        opt.gaps.dgram_loaded = true;
        opt.gaps.min_gap = dgram_min_;
        opt.gaps.max_gap = dgram_max_;
    }
    if (shards_.size() > 1) return run_queries_sharded(shards_, enc_, regexes, status, messages, stats, &opt, aux_shards_);
    return run_queries(ix_, enc_, regexes, status, messages, stats, &opt, aux_);
}

std::vector<uint64_t> set_bins(const uint64_t* mask, uint64_t bins) {
    std::vector<uint64_t> out;
    if (bins == 1) { out.push_back(0); return out; }  // a 1-bin library is always scanned
    const uint64_t words = (bins + 63) / 64;
    for (uint64_t w = 0; w < words; ++w)
        for (uint64_t v = mask[w]; v; v &= v - 1) out.push_back(w * 64 + (unsigned)__builtin_ctzll(v));
    return out;
}

uint64_t compute_bitcount(uint64_t n, float fpr) {
    const double num = -static_cast<double>(n) * std::log(fpr);  // float log, as in the reference
    const double den = std::pow(std::log(2), 2);
    return static_cast<uint64_t>(std::ceil(num / den));
}

namespace {

struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(size_t bytes) { txq_check(txq_malloc(&p, bytes), "txq_malloc"); }
    ~DevBuf() { txq_free(p); }
};

// Build one flat IBF on the device from per-bin value lists and copy its words back.
IbfImage build_flat(const std::vector<const std::vector<uint64_t>*>& per_bin, uint64_t rows, unsigned h) {
    IbfImage img;
    img.shape(per_bin.size(), rows, h);
    txq_index* ix = nullptr;
    txq_check(txq_index_create_ibf(img.bins, rows, h, 0, 1, &ix), "txq_index_create_ibf");
    try {
        std::vector<uint64_t> vals;
        std::vector<uint32_t> bins;
        auto flush = [&]() {
            if (vals.empty()) return;
            DevBuf dv(vals.size() * 8), db(bins.size() * 4);
            txq_check(txq_memcpy_h2d(dv.p, vals.data(), vals.size() * 8), "h2d");
            txq_check(txq_memcpy_h2d(db.p, bins.data(), bins.size() * 4), "h2d");
            txq_check(txq_emplace_device(ix, (const uint64_t*)dv.p, (const uint32_t*)db.p, vals.size(), nullptr), "txq_emplace_device");
            txq_check(txq_synchronize(), "txq_synchronize");
            vals.clear();
            bins.clear();
        };
        for (size_t b = 0; b < per_bin.size(); ++b) {
            for (uint64_t v : *per_bin[b]) { vals.push_back(v); bins.push_back((uint32_t)b); }
            if (vals.size() >= (1u << 24)) flush();
        }
        flush();
        txq_check(txq_index_download_words(ix, img.words.data(), img.words.size()), "txq_index_download_words");
    } catch (...) {
        txq_index_free(ix);
        throw;
    }
    txq_index_free(ix);
    return img;
}

}  // namespace

IndexImage build_index(const std::vector<std::string>& bin_files, const BuildOptions& opt, size_t* n_sequences) {
    if (bin_files.empty()) throw std::runtime_error("no input libraries");
    if (!opt.dna && opt.k > 12) throw std::runtime_error("Max kmer size for amino acids is 12");
    if (opt.dna && opt.k > 32) throw std::runtime_error("Max kmer size for nucleic acids is 32");
    ensure_device(opt.device);
    const KmerEncoder enc(opt.dna ? Molecule::DNA : Molecule::Peptide, opt.k, (Alphabet)opt.reduction);
    std::vector<std::vector<uint64_t>> values(bin_files.size());
    size_t seqs = 0;
    for (size_t b = 0; b < bin_files.size(); ++b)
        for_each_record(bin_files[b], [&](const FastaRecord& r) {
            if (r.seq.size() < opt.k) return;  // "RECORD TOO SHORT"
            ++seqs;
            enc.record_values(r.seq, opt.dna_wraparound, values[b]);
        });
    if (n_sequences) *n_sequences = seqs;

    IndexImage img;
    img.k = (uint8_t)opt.k;
    img.molecule = opt.dna ? "na" : "aa";
    img.reduction = (uint8_t)opt.reduction;
    img.hash_count = (uint8_t)opt.hash_count;
    img.fpr = opt.fpr;
    img.bin_paths = bin_files;
    auto largest = [](const std::vector<const std::vector<uint64_t>*>& v) {
        size_t m = 0;
        for (auto* p : v) m = std::max(m, p->size());
        return m;
    };
    if (!opt.hibf) {
        // IBFIndex::init_ibf: size every bin like the largest one, occurrences not deduplicated
        std::vector<const std::vector<uint64_t>*> per_bin;
        for (auto& v : values) per_bin.push_back(&v);
        const uint64_t rows = std::max<uint64_t>(1, compute_bitcount(largest(per_bin), opt.fpr));
        img.ibf = build_flat(per_bin, rows, opt.hash_count);
        img.format = "ibf";
        return img;
    }
    // HIBF with this project's own two-level layout (the reference delegates the layout to
    // seqan::hibf's sketch-based algorithm, which is index construction, not query): user bins are
    // dealt in order over at most t_max = 64*ceil(sqrt(B)/64) merged technical bins of the root, each
    // pointing at a child IBF with one technical bin per user bin.  B <= t_max: a single level.
    // A child holds a multiple of 64 user bins, so its rows are whole words of the result mask, and all children
    // have the rows of the largest user bin (as the flat IBF sizes its bins): the device keeps such a tree's children
    // side by side and probes them like one flat IBF (csrc/txq_hibf.hip: regular trees, uniform children).
    img.is_hibf = true;
    HibfImage& h = img.hibf;
    const uint64_t B = bin_files.size();
    h.user_bins = B;
    const uint64_t tmax = 64 * (((uint64_t)std::ceil(std::sqrt((double)B)) + 63) / 64);
    if (B <= tmax) {
        std::vector<const std::vector<uint64_t>*> per_bin;
        for (auto& v : values) per_bin.push_back(&v);
        h.ibfs.push_back(build_flat(per_bin, std::max<uint64_t>(1, compute_bitcount(largest(per_bin), opt.fpr)), opt.hash_count));
        h.next_ibf_id.emplace_back(B, 0);
        h.tb_to_user_bin.emplace_back();
        for (uint64_t b = 0; b < B; ++b) h.tb_to_user_bin.back().push_back(b);
    } else {
        const uint64_t per_child = 64 * (((B + tmax - 1) / tmax + 63) / 64), n_child = (B + per_child - 1) / per_child;
        std::vector<const std::vector<uint64_t>*> all_bins;
        for (auto& v : values) all_bins.push_back(&v);
        const uint64_t child_rows = std::max<uint64_t>(1, compute_bitcount(largest(all_bins), opt.fpr));
        std::vector<std::vector<uint64_t>> merged(n_child);
        h.ibfs.resize(1 + n_child);
        h.next_ibf_id.resize(1 + n_child);
        h.tb_to_user_bin.resize(1 + n_child);
        for (uint64_t c = 0; c < n_child; ++c) {
            const uint64_t lo = c * per_child, hi = std::min(B, lo + per_child);
            std::vector<const std::vector<uint64_t>*> per_bin;
            for (uint64_t b = lo; b < hi; ++b) {
                per_bin.push_back(&values[b]);
                merged[c].insert(merged[c].end(), values[b].begin(), values[b].end());
                h.tb_to_user_bin[1 + c].push_back(b);
            }
            h.next_ibf_id[1 + c].assign(hi - lo, 0);
            h.ibfs[1 + c] = build_flat(per_bin, child_rows, opt.hash_count);
            h.next_ibf_id[0].push_back(1 + c);
            h.tb_to_user_bin[0].push_back(UINT64_MAX);
        }
        std::vector<const std::vector<uint64_t>*> per_bin;
        for (auto& v : merged) per_bin.push_back(&v);
        h.ibfs[0] = build_flat(per_bin, std::max<uint64_t>(1, compute_bitcount(largest(per_bin), opt.fpr)), opt.hash_count);
    }
    img.format = "hibf";
    return img;
}

DgramImage build_dgram_index(const std::vector<std::string>& bin_files, uint64_t min_gap, uint64_t max_gap, unsigned hash_count,
                             float fpr, int device) {
    if (bin_files.empty()) throw std::runtime_error("no input libraries");
    if (min_gap > max_gap) throw std::runtime_error("lower gap bound above the upper bound");
    ensure_device(device);
    std::vector<std::vector<uint64_t>> values(bin_files.size());
    for (size_t b = 0; b < bin_files.size(); ++b)
        for_each_record(bin_files[b], [&](const FastaRecord& r) { dgram_record_values(r.seq, min_gap, max_gap, values[b]); });
    std::vector<const std::vector<uint64_t>*> per_bin;
    size_t most = 0;
    for (auto& v : values) { per_bin.push_back(&v); most = std::max(most, v.size()); }
    DgramImage d;
    d.min_gap = min_gap;
    d.max_gap = max_gap;
    d.hash_count = (uint8_t)hash_count;
    d.fpr = fpr;
    d.bin_paths = bin_files;
    d.ibf = build_flat(per_bin, std::max<uint64_t>(1, compute_bitcount(most, fpr)), hash_count);
    d.format = "dgram";
    return d;
}

}  // namespace tetrex

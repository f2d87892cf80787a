#include "compiler.hpp"
#include "regex_front.hpp"

#include <cstring>
#include <stdexcept>

namespace tetrex {

namespace {
constexpr uint32_t kPinned = 0x7FFFFFFF;  // reference count of the constant slots

std::vector<uint8_t> make_blob(const std::vector<uint64_t>& kmers, const std::vector<txq_program>& programs,
                               const std::vector<const std::vector<txq_op>*>& ops_of) {
    size_t n_ops = 0;
    for (const auto* v : ops_of) n_ops += v ? v->size() : 0;
    if (n_ops > 0xFFFFFFFFu) throw std::runtime_error("batch has more than 2^32 operations");
    txq_blob_header h{};
    h.magic = TXQ_PROGRAM_MAGIC;
    h.version = TXQ_PROGRAM_VERSION;
    h.n_programs = (uint32_t)programs.size();
    h.n_kmers = (uint32_t)kmers.size();
    h.n_ops = (uint32_t)n_ops;
    h.kmers_offset = sizeof(txq_blob_header);
    h.programs_offset = h.kmers_offset + kmers.size() * sizeof(uint64_t);
    h.ops_offset = h.programs_offset + programs.size() * sizeof(txq_program);
    std::vector<uint8_t> blob(h.ops_offset + n_ops * sizeof(txq_op));
    std::memcpy(blob.data(), &h, sizeof h);
    if (!kmers.empty()) std::memcpy(blob.data() + h.kmers_offset, kmers.data(), kmers.size() * 8);
    if (!programs.empty()) std::memcpy(blob.data() + h.programs_offset, programs.data(), programs.size() * sizeof(txq_program));
    uint8_t* at = blob.data() + h.ops_offset;
    for (const auto* v : ops_of) {
        if (!v || v->empty()) continue;
        std::memcpy(at, v->data(), v->size() * sizeof(txq_op));
        at += v->size() * sizeof(txq_op);
    }
    return blob;
}
}  // namespace

// ---- QueryExpansion -----------------------------------------------------------------------

QueryExpansion::QueryExpansion(const KmerEncoder& enc, KGraph graph, CompileLimits limits)
    : enc_(enc), g_(std::move(graph)), limits_(limits) {
    if (enc_.k() < 2) throw std::runtime_error("k must be at least 2");
    order_ = g_.topological_order();
    table_.resize(g_.size());
    refs_.assign(TXQ_SLOT_FIRST_FREE, kPinned);
    std::vector<txq_op> none;
    arrive(0, State{0, TXQ_SLOT_ONES, 0}, none);
}

uint32_t QueryExpansion::fresh() {
    uint32_t s;
    if (!free_.empty()) { s = free_.back(); free_.pop_back(); }
    else { s = (uint32_t)refs_.size(); refs_.push_back(0); }
    refs_[s] = 1;
    if (s + 1 > high_water_) high_water_ = s + 1;
    return s;
}
void QueryExpansion::share(uint32_t s) { if (refs_[s] != kPinned) ++refs_[s]; }
void QueryExpansion::drop(uint32_t s) {
    if (refs_[s] == kPinned) return;
    if (--refs_[s] == 0) free_.push_back(s);
}
bool QueryExpansion::exclusive(uint32_t s) const { return refs_[s] == 1; }

void QueryExpansion::emit(std::vector<txq_op>& out, uint32_t kmer, uint32_t dst, uint32_t a, uint32_t b) {
    if (++total_ops_ > limits_.max_ops) throw std::runtime_error("query expands to too many mask operations");
    out.push_back(txq_op{kmer, dst, a, b});
}

// hand a state (owning one reference to its slot) to node `to`
void QueryExpansion::arrive(int32_t to, State s, std::vector<txq_op>& out) {
    if (to == KGraph::kNone) throw std::runtime_error("k-graph node without successor (the reference fails here too)");
    const unsigned k = enc_.k(), bits = enc_.bits_per_symbol();
    NodeStates& ns = table_[to];
    // length-prefixed key: the symbols seen so far (at most the k-1 newest) with a marker bit just
    // above them, so paths of different length < k-1 never share a key
    const unsigned phase = s.shift < k - 1 ? s.shift : k - 1;
    const uint64_t key = (s.kmer & enc_.suffix_mask()) | (1ULL << (phase * bits));
    auto [it, inserted] = ns.by_key.emplace(key, (uint32_t)ns.items.size());
    if (inserted) {
        ns.items.push_back(s);
        if (++states_ > limits_.max_states) throw std::runtime_error("query expands to too many states");
        return;
    }
    State& have = ns.items[it->second];
    if (have.shift < s.shift) have.shift = s.shift;  // k-1 and k behave alike from here on
    if (have.slot == s.slot) { drop(s.slot); return; }
    // absorb: have.path |= s.path
    if (exclusive(have.slot)) {
        emit(out, TXQ_NO_KMER, have.slot, have.slot, s.slot);
        drop(s.slot);
    } else if (exclusive(s.slot)) {
        emit(out, TXQ_NO_KMER, s.slot, s.slot, have.slot);
        drop(have.slot);
        have.slot = s.slot;
    } else {
        const uint32_t d = fresh();
        emit(out, TXQ_NO_KMER, d, have.slot, s.slot);
        drop(have.slot);
        drop(s.slot);
        have.slot = d;
    }
}

void QueryExpansion::advance(size_t op_budget, const Intern& intern, std::vector<txq_op>& out) {
    const unsigned k = enc_.k();
    const size_t start = out.size();
    while (cursor_ < order_.size() && out.size() - start < op_budget) {
        const int32_t node = order_[cursor_++];
        NodeStates ns;
        ns.items.swap(table_[node].items);
        table_[node].by_key.clear();
        const int32_t lab = g_.label[node];
        for (State s : ns.items) {
            switch (lab) {
                case KGraph::kMatch:
                    emit(out, TXQ_NO_KMER, TXQ_SLOT_RESULT, s.slot, TXQ_SLOT_RESULT);
                    drop(s.slot);
                    break;
                case '$':  // passes through untouched (include/otf_collector.h:364-368)
                case KGraph::kGhost:
                    arrive(g_.next_a[node], s, out);
                    break;
                case KGraph::kSplit:
                    share(s.slot);
                    arrive(g_.next_a[node], s, out);
                    arrive(g_.next_b[node], s, out);
                    break;
                case KGraph::kGap:
                    throw std::runtime_error("gap nodes (-a/-g) are not supported yet");
                default: {
                    const uint64_t probe = enc_.roll((unsigned char)lab, s.kmer);
                    if (s.shift < k - 1) {
                        ++s.shift;
                    } else {
                        const uint32_t id = intern(probe);
                        ++probes_;
                        if (exclusive(s.slot)) {
                            emit(out, id, s.slot, s.slot, TXQ_SLOT_ZERO);
                        } else {
                            const uint32_t d = fresh();
                            emit(out, id, d, s.slot, TXQ_SLOT_ZERO);
                            drop(s.slot);
                            s.slot = d;
                        }
                        s.shift = (uint8_t)k;
                    }
                    arrive(g_.next_a[node], s, out);
                    break;
                }
            }
        }
    }
}

void QueryExpansion::frontier_slots(std::vector<uint32_t>& out) const {
    std::vector<uint8_t> seen(refs_.size(), 0);
    for (size_t c = cursor_; c < order_.size(); ++c)
        for (const State& s : table_[order_[c]].items)
            if (s.slot >= TXQ_SLOT_FIRST_FREE && !seen[s.slot]) { seen[s.slot] = 1; out.push_back(s.slot); }
}

void QueryExpansion::prune(const std::vector<uint8_t>& dead) {
    for (size_t c = cursor_; c < order_.size(); ++c) {
        NodeStates& ns = table_[order_[c]];
        bool any = false;
        for (const State& s : ns.items)
            if (s.slot < dead.size() && dead[s.slot]) { any = true; break; }
        if (!any) continue;
        std::vector<State> keep;
        for (const State& s : ns.items) {
            if (s.slot < dead.size() && dead[s.slot]) { drop(s.slot); ++pruned_; }
            else keep.push_back(s);
        }
        ns.items.swap(keep);
        ns.by_key.clear();
        const unsigned k = enc_.k(), bits = enc_.bits_per_symbol();
        for (uint32_t i = 0; i < ns.items.size(); ++i) {
            const State& s = ns.items[i];
            const unsigned phase = s.shift < k - 1 ? s.shift : k - 1;
            ns.by_key.emplace((s.kmer & enc_.suffix_mask()) | (1ULL << (phase * bits)), i);
        }
    }
}

// ---- staged driver ------------------------------------------------------------------------

StagedStats run_staged(const KmerEncoder& enc, uint64_t bins, const std::vector<std::string>& regexes, StageExecutor& exec,
                       const StagedOptions& opt, std::vector<int>* status, std::vector<std::string>* messages) {
    const size_t n = regexes.size();
    if (status) status->assign(n, 0);
    if (messages) messages->assign(n, std::string());
    std::vector<std::unique_ptr<QueryExpansion>> q(n);
    std::vector<uint8_t> passthrough(n, 0);
    auto failed = [&](size_t i, const char* why) {
        q[i].reset();
        if (status) (*status)[i] = -1;
        if (messages) (*messages)[i] = why;
    };
    for (size_t i = 0; i < n; ++i) {
        try {
            if (bins <= 1) { passthrough[i] = 1; continue; }  // include/query.h:265-272
            const std::string postfix = preprocess_query(regexes[i], enc);
            q[i] = std::make_unique<QueryExpansion>(enc, build_kgraph(postfix, enc.k(), enc.alphabet() != Alphabet::Base), opt.limits);
        } catch (const std::exception& e) { failed(i, e.what()); }
    }
    StagedStats st;
    std::vector<std::vector<txq_op>> ops(n);
    std::vector<uint32_t> slots(n, TXQ_SLOT_FIRST_FREE);
    bool first = true;
    for (;;) {
        std::vector<uint64_t> kmers;
        std::unordered_map<uint64_t, uint32_t> index;
        const QueryExpansion::Intern intern = [&](uint64_t v) {
            auto [it, fresh] = index.emplace(v, (uint32_t)kmers.size());
            if (fresh) kmers.push_back(v);
            return it->second;
        };
        size_t total = 0;
        bool pending = false;
        for (size_t i = 0; i < n; ++i) {
            ops[i].clear();
            if (first && passthrough[i]) ops[i].push_back(txq_op{TXQ_NO_KMER, TXQ_SLOT_RESULT, TXQ_SLOT_ONES, TXQ_SLOT_RESULT});
            if (!q[i] || q[i]->done()) continue;
            if (total < opt.ops_per_stage) {
                try {
                    q[i]->advance(opt.ops_per_query_per_stage, intern, ops[i]);
                } catch (const std::exception& e) {
                    // ops already emitted in earlier stages only ever feed RESULT through a Match
                    // op, so an abandoned query is neutralised by not emitting anything further
                    ops[i].clear();
                    failed(i, e.what());
                    continue;
                }
                total += ops[i].size();
                slots[i] = q[i]->n_slots();
            }
            if (!q[i]->done()) pending = true;
        }
        if (!first && total == 0 && !pending) break;
        std::vector<txq_program> programs(n);
        std::vector<const std::vector<txq_op>*> ops_of(n);
        uint32_t at = 0;
        for (size_t i = 0; i < n; ++i) {
            programs[i] = txq_program{at, (uint32_t)ops[i].size(), slots[i], 0};
            at += (uint32_t)ops[i].size();
            ops_of[i] = &ops[i];
        }
        std::vector<uint32_t> qp, qs;
        for (size_t i = 0; i < n; ++i) {
            if (!q[i] || q[i]->done()) continue;
            const size_t before = qs.size();
            q[i]->frontier_slots(qs);
            qp.insert(qp.end(), qs.size() - before, (uint32_t)i);
        }
        std::vector<uint8_t> alive(qp.size(), 1);
        exec.stage(make_blob(kmers, programs, ops_of), qp, qs, alive);
        ++st.stages;
        st.ops += total;
        st.kmers += kmers.size();
        st.feedback_queries += qp.size();
        // prune dead frontier states
        for (size_t a = 0; a < qp.size();) {
            const uint32_t p = qp[a];
            std::vector<uint8_t> dead(q[p]->n_slots(), 0);
            bool any = false;
            for (; a < qp.size() && qp[a] == p; ++a)
                if (!alive[a]) { dead[qs[a]] = 1; any = true; }
            if (any) q[p]->prune(dead);
        }
        first = false;
        if (!pending) break;
    }
    for (size_t i = 0; i < n; ++i)
        if (q[i]) { st.states += q[i]->states(); st.pruned += q[i]->pruned(); }
    return st;
}

// ---- one-shot batches ----------------------------------------------------------------------

uint32_t ProgramBatch::intern(uint64_t value) {
    auto it = kmer_index_.find(value);
    if (it != kmer_index_.end()) return it->second;
    if (kmers_.size() >= 0xFFFFFFFEu) throw std::runtime_error("k-mer table overflow");
    const uint32_t id = (uint32_t)kmers_.size();
    kmers_.push_back(value);
    kmer_index_.emplace(value, id);
    return id;
}

size_t ProgramBatch::add_passthrough() {
    QueryProgram p;
    p.ops.push_back(txq_op{TXQ_NO_KMER, TXQ_SLOT_RESULT, TXQ_SLOT_ONES, TXQ_SLOT_RESULT});
    programs_.push_back(std::move(p));
    return programs_.size() - 1;
}

size_t ProgramBatch::add_empty() {
    programs_.push_back(QueryProgram{});
    return programs_.size() - 1;
}

size_t ProgramBatch::add(const KGraph& g) {
    QueryExpansion x(enc_, g, limits_);
    QueryProgram prog;
    x.advance(SIZE_MAX, [this](uint64_t v) { return intern(v); }, prog.ops);
    prog.n_slots = x.n_slots();
    prog.states = x.states();
    prog.probes = x.probes();
    programs_.push_back(std::move(prog));
    return programs_.size() - 1;
}

std::vector<uint8_t> ProgramBatch::serialise() const {
    std::vector<txq_program> pr(programs_.size());
    std::vector<const std::vector<txq_op>*> ops_of(programs_.size());
    uint32_t first = 0;
    for (size_t i = 0; i < programs_.size(); ++i) {
        pr[i] = txq_program{first, (uint32_t)programs_[i].ops.size(), programs_[i].n_slots, 0};
        first += (uint32_t)programs_[i].ops.size();
        ops_of[i] = &programs_[i].ops;
    }
    return make_blob(kmers_, pr, ops_of);
}

}  // namespace tetrex

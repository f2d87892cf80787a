#include "compiler.hpp"

#include <cstring>
#include <stdexcept>

namespace tetrex {

namespace {

// Slot allocator with reference counts; slots 0..2 are the constants of txq_program.h.
class Slots {
  public:
    Slots() : refs_(TXQ_SLOT_FIRST_FREE, kPinned) {}
    static constexpr uint32_t kPinned = 0x7FFFFFFF;

    uint32_t fresh() {
        uint32_t s;
        if (!free_.empty()) { s = free_.back(); free_.pop_back(); }
        else { s = (uint32_t)refs_.size(); refs_.push_back(0); }
        refs_[s] = 1;
        return s;
    }
    void share(uint32_t s) { if (refs_[s] != kPinned) ++refs_[s]; }
    void drop(uint32_t s) {
        if (refs_[s] == kPinned) return;
        if (--refs_[s] == 0) free_.push_back(s);
    }
    bool exclusive(uint32_t s) const { return refs_[s] == 1; }
    uint32_t high_water() const { return (uint32_t)refs_.size(); }

  private:
    std::vector<uint32_t> refs_, free_;
};

struct State {
    uint64_t kmer;   // forward k-mer so far
    uint32_t slot;   // mask of the bins still compatible with this path
    uint8_t shift;   // symbols seen, saturating at k (shift_count_ of the reference)
};

struct NodeStates {
    std::vector<State> items;
    std::unordered_map<uint64_t, uint32_t> by_key;
};

}  // namespace

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
    const unsigned k = enc_.k();
    if (k < 2) throw std::runtime_error("k must be at least 2");
    const uint64_t suffix = enc_.suffix_mask();
    const unsigned bits = enc_.bits_per_symbol();
    QueryProgram prog;
    Slots slots;
    std::vector<NodeStates> table(g.size());
    const std::vector<int32_t> order = g.topological_order();

    auto emit = [&](uint32_t kmer, uint32_t dst, uint32_t a, uint32_t b) {
        if (prog.ops.size() >= limits_.max_ops) throw std::runtime_error("query expands to too many mask operations");
        prog.ops.push_back(txq_op{kmer, dst, a, b});
    };
    // hand a state (owning one reference to its slot) to node `to`
    auto arrive = [&](int32_t to, State s) {
        if (to == KGraph::kNone) throw std::runtime_error("k-graph node without successor (the reference fails here too)");
        NodeStates& ns = table[to];
        // length-prefixed key: the symbols seen so far (at most the k-1 newest) with a marker
        // bit just above them, so paths of different length < k-1 never share a key
        const unsigned phase = s.shift < k - 1 ? s.shift : k - 1;
        const uint64_t key = (s.kmer & suffix) | (1ULL << (phase * bits));
        auto [it, inserted] = ns.by_key.emplace(key, (uint32_t)ns.items.size());
        if (inserted) {
            ns.items.push_back(s);
            if (++prog.states > limits_.max_states) throw std::runtime_error("query expands to too many states");
            return;
        }
        State& have = ns.items[it->second];
        if (have.shift < s.shift) have.shift = s.shift;  // k-1 and k behave alike from here on
        if (have.slot == s.slot) { slots.drop(s.slot); return; }
        // absorb: have.path |= s.path
        if (slots.exclusive(have.slot)) {
            emit(TXQ_NO_KMER, have.slot, have.slot, s.slot);
            slots.drop(s.slot);
        } else if (slots.exclusive(s.slot)) {
            emit(TXQ_NO_KMER, s.slot, s.slot, have.slot);
            slots.drop(have.slot);
            have.slot = s.slot;
        } else {
            const uint32_t d = slots.fresh();
            emit(TXQ_NO_KMER, d, have.slot, s.slot);
            slots.drop(have.slot);
            slots.drop(s.slot);
            have.slot = d;
        }
    };

    arrive(0, State{0, TXQ_SLOT_ONES, 0});
    for (int32_t node : order) {
        NodeStates& ns = table[node];
        const int32_t lab = g.label[node];
        for (size_t i = 0; i < ns.items.size(); ++i) {
            State s = ns.items[i];
            switch (lab) {
                case KGraph::kMatch:
                    emit(TXQ_NO_KMER, TXQ_SLOT_RESULT, s.slot, TXQ_SLOT_RESULT);
                    slots.drop(s.slot);
                    break;
                case '$':  // passes through untouched (include/otf_collector.h:364-368)
                case KGraph::kGhost:
                    arrive(g.next_a[node], s);
                    break;
                case KGraph::kSplit:
                    slots.share(s.slot);
                    arrive(g.next_a[node], s);
                    arrive(g.next_b[node], s);
                    break;
                case KGraph::kGap:
                    throw std::runtime_error("gap nodes (-a/-g) are not supported yet");
                default: {
                    const uint64_t probe = enc_.roll((unsigned char)lab, s.kmer);
                    if (s.shift < k - 1) {
                        ++s.shift;
                    } else {
                        const uint32_t id = intern(probe);
                        ++prog.probes;
                        if (slots.exclusive(s.slot)) {
                            emit(id, s.slot, s.slot, TXQ_SLOT_ZERO);
                        } else {
                            const uint32_t d = slots.fresh();
                            emit(id, d, s.slot, TXQ_SLOT_ZERO);
                            slots.drop(s.slot);
                            s.slot = d;
                        }
                        s.shift = (uint8_t)k;
                    }
                    arrive(g.next_a[node], s);
                    break;
                }
            }
        }
        NodeStates().items.swap(ns.items);
        ns.by_key.clear();
    }
    prog.n_slots = slots.high_water();
    programs_.push_back(std::move(prog));
    return programs_.size() - 1;
}

std::vector<uint8_t> ProgramBatch::serialise() const {
    size_t n_ops = 0;
    for (const auto& p : programs_) n_ops += p.ops.size();
    if (n_ops > 0xFFFFFFFFu) throw std::runtime_error("batch has more than 2^32 operations");
    txq_blob_header h{};
    h.magic = TXQ_PROGRAM_MAGIC;
    h.version = TXQ_PROGRAM_VERSION;
    h.n_programs = (uint32_t)programs_.size();
    h.n_kmers = (uint32_t)kmers_.size();
    h.n_ops = (uint32_t)n_ops;
    h.kmers_offset = sizeof(txq_blob_header);
    h.programs_offset = h.kmers_offset + kmers_.size() * sizeof(uint64_t);
    h.ops_offset = h.programs_offset + programs_.size() * sizeof(txq_program);
    std::vector<uint8_t> blob(h.ops_offset + n_ops * sizeof(txq_op));
    std::memcpy(blob.data(), &h, sizeof h);
    if (!kmers_.empty()) std::memcpy(blob.data() + h.kmers_offset, kmers_.data(), kmers_.size() * 8);
    txq_program* pr = reinterpret_cast<txq_program*>(blob.data() + h.programs_offset);
    txq_op* ops = reinterpret_cast<txq_op*>(blob.data() + h.ops_offset);
    uint32_t first = 0;
    for (size_t i = 0; i < programs_.size(); ++i) {
        const QueryProgram& p = programs_[i];
        pr[i] = txq_program{first, (uint32_t)p.ops.size(), p.n_slots, 0};
        if (!p.ops.empty()) std::memcpy(ops + first, p.ops.data(), p.ops.size() * sizeof(txq_op));
        first += (uint32_t)p.ops.size();
    }
    return blob;
}

}  // namespace tetrex

#include "compiler.hpp"
#include "regex_front.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>

namespace tetrex {

namespace {
constexpr uint32_t kPinned = 0x7FFFFFFF;  // reference count of the constant slots

using ParallelFor = std::function<void(size_t, const std::function<void(size_t, int)>&)>;

std::vector<uint8_t> make_blob(const std::vector<const KmerVec*>& kmer_tables, const std::vector<txq_program>& programs,
                               const std::vector<const OpVec*>& ops_of,
                               const std::vector<const std::vector<uint32_t>*>& levels_of, const ParallelFor* par = nullptr,
                               size_t n_aux_kmers = 0) {
    size_t n_kmers = 0;
    for (const auto* t : kmer_tables) n_kmers += t->size();
    if (n_kmers > 0xFFFFFFFEu) throw std::runtime_error("k-mer table overflow");
    size_t n_ops = 0, n_levels = 0;
    for (const auto* v : ops_of) n_ops += v ? v->size() : 0;
    for (const auto* v : levels_of) n_levels += v ? v->size() : 0;
    if (n_ops > 0xFFFFFFFFu || n_levels > 0xFFFFFFFFu) throw std::runtime_error("batch has more than 2^32 operations");
    txq_blob_header_v2 h{};
    h.magic = TXQ_PROGRAM_MAGIC;
    h.version = TXQ_PROGRAM_VERSION_LEVELS;
    h.n_programs = (uint32_t)programs.size();
    h.n_kmers = (uint32_t)n_kmers;
    h.n_ops = (uint32_t)n_ops;
    h.n_levels = (uint32_t)n_levels;
    h.n_aux_kmers = n_aux_kmers;
    h.kmers_offset = sizeof(txq_blob_header_v2);
    h.programs_offset = h.kmers_offset + n_kmers * sizeof(uint64_t);
    h.ops_offset = h.programs_offset + programs.size() * sizeof(txq_program_v2);
    h.levels_offset = h.ops_offset + n_ops * sizeof(txq_op);
    std::vector<uint8_t> blob(h.levels_offset + ((n_levels * 4 + 7) & ~(size_t)7));
    std::memcpy(blob.data(), &h, sizeof h);
    {
        uint8_t* at = blob.data() + h.kmers_offset;
        for (const auto* t : kmer_tables) {
            if (!t->empty()) std::memcpy(at, t->data(), t->size() * 8);
            at += t->size() * 8;
        }
    }
    txq_program_v2* pr = reinterpret_cast<txq_program_v2*>(blob.data() + h.programs_offset);
    uint32_t* lv = reinterpret_cast<uint32_t*>(blob.data() + h.levels_offset);
    uint32_t first_level = 0;
    for (size_t i = 0; i < programs.size(); ++i) {
        const uint32_t nl = levels_of[i] ? (uint32_t)levels_of[i]->size() : 0;
        pr[i] = txq_program_v2{programs[i].first_op, programs[i].n_ops, programs[i].n_slots, first_level, nl, 0};
        first_level += nl;
    }
    uint8_t* ops_at = blob.data() + h.ops_offset;
    auto copy_program = [&](size_t i, int) {
        if (pr[i].n_levels) std::memcpy(lv + pr[i].first_level, levels_of[i]->data(), (size_t)pr[i].n_levels * 4);
        const auto* v = ops_of[i];
        if (v && !v->empty()) std::memcpy(ops_at + (size_t)pr[i].first_op * sizeof(txq_op), v->data(), v->size() * sizeof(txq_op));
    };
    if (par) (*par)(programs.size(), copy_program);
    else for (size_t i = 0; i < programs.size(); ++i) copy_program(i, 0);
    return blob;
}
}  // namespace

// ---- QueryExpansion -----------------------------------------------------------------------

static bool is_epsilon(int32_t label) { return label == KGraph::kGhost || label == KGraph::kSplit || label == '$'; }

uint64_t dgram_residue_code(int symbol) {
    static const uint8_t base[26] = {0, 2, 1, 2, 3, 4, 5, 6, 7, 9, 8, 9, 10, 11, 20, 12, 13, 14, 15, 16, 20, 17, 18, 20, 19, 3};
    return (symbol >= 'A' && symbol <= 'Z') ? base[symbol - 'A'] : 0;
}

void dgram_record_values(std::string_view seq, uint64_t min_gap, uint64_t max_gap, std::vector<uint64_t>& out) {
    // only the 20 standard residues take part (include/dGramIndex.h:128-135,159-211)
    auto code = [](char c) -> int {
        static const char* alphabet = "ACDEFGHIKLMNPQRSTVWY";
        const char* p = c ? std::strchr(alphabet, c) : nullptr;
        return p ? (int)(p - alphabet) : -1;
    };
    if (seq.size() < min_gap + 7) return;
    for (size_t i = 2; i + min_gap + 3 < seq.size(); ++i) {
        const int a1 = code(seq[i - 2]), a2 = code(seq[i - 1]), a3 = code(seq[i]);
        if (a1 < 0 || a2 < 0 || a3 < 0) continue;
        for (uint64_t gap = min_gap; gap <= max_gap; ++gap) {
            const size_t j = i + gap + 1;
            if (j + 2 >= seq.size()) break;
            const int b1 = code(seq[j]), b2 = code(seq[j + 1]), b3 = code(seq[j + 2]);
            if (b1 < 0 || b2 < 0 || b3 < 0) continue;
            out.push_back(gap * 64000000ULL + (uint64_t)a1 * 3200000ULL + (uint64_t)a2 * 160000ULL + (uint64_t)a3 * 8000ULL +
                          (uint64_t)b1 * 400ULL + (uint64_t)b2 * 20ULL + (uint64_t)b3);
        }
    }
}

QueryExpansion::~QueryExpansion() = default;

uint64_t dense_block_slots(const KmerEncoder& enc, const DenseOptions& opt) {
    const unsigned pos = enc.k() - 1;
    if (!opt.enabled || enc.k() < 2 || pos > TXQ_DENSE_MAX_POSITIONS || opt.max_blocks == 0) return 0;
    const uint64_t slot_bytes = opt.slot_bytes ? opt.slot_bytes : 128;
    uint64_t n = 1;
    for (unsigned j = 1; j <= pos; ++j) {
        n *= enc.alphabet_size();
        if (n > (1u << 22) || n * slot_bytes > opt.max_block_bytes) return 0;
    }
    return n * 2 < TXQ_DENSE_SLOT_BIT ? n : 0;  // room for at least two blocks
}

QueryExpansion::QueryExpansion(const KmerEncoder& enc, KGraph graph, CompileLimits limits, GapOptions gaps, DenseOptions dense)
    : enc_(enc), g_(std::move(graph)), limits_(limits), gaps_(gaps), dense_(dense) {
    if (enc_.k() < 2) throw std::runtime_error("k must be at least 2");
    // dense blocks: the d-gram filter needs three residues of history per state, which a (k-1)-suffix block does not keep
    dense_pos_ = enc_.k() - 1;
    dense_a_ = enc_.alphabet_size();
    if (const uint64_t n = gaps_.dgram_loaded ? 0 : tetrex::dense_block_slots(enc_, dense_)) {
        dense_ok_ = true;
        dense_n_ = n;
        if (dense_.max_blocks > TXQ_DENSE_MAX_BLOCKS) dense_.max_blocks = TXQ_DENSE_MAX_BLOCKS;  // block ids are 8 bits of a dense slot number
    }
    if (gaps_.augment) g_.augment();
    const int32_t n = n_nodes_ = g_.size();
    const std::vector<int32_t> topo = g_.topological_order();
    // closure[v] of epsilon nodes, successors first: sorted, duplicate-free lists in ONE arena (a 20-way wildcard union is
    // a tree of 19 Split nodes over 20 residues: per-node vectors were a third of this constructor's time)
    std::vector<int32_t> arena;
    arena.reserve((size_t)n * 8);
    std::vector<uint32_t> c_off(n, 0), c_len(n, 0);
    std::vector<uint8_t> hole(n, 0);
    // the target list behind successor slot t: {t} itself for a residue / Match node, closure[t] for an epsilon node
    auto list_of = [&](int32_t t, const int32_t** p, uint32_t* len, int32_t* self, uint8_t& dang) {
        *len = 0;
        if (t == KGraph::kNone) { dang = 1; return; }
        if (!is_epsilon(g_.label[t])) { *self = t; *p = self; *len = 1; return; }
        *p = arena.data() + c_off[t];
        *len = c_len[t];
        dang |= hole[t];
    };
    for (size_t i = topo.size(); i-- > 0;) {
        const int32_t v = topo[i];
        if (!is_epsilon(g_.label[v])) continue;
        const int32_t *pa = nullptr, *pb = nullptr;
        uint32_t la = 0, lb = 0;
        int32_t sa = 0, sb = 0;
        list_of(g_.next_a[v], &pa, &la, &sa, hole[v]);
        if (g_.label[v] == KGraph::kSplit) list_of(g_.next_b[v], &pb, &lb, &sb, hole[v]);
        const size_t at = arena.size();
        if (arena.capacity() < at + la + lb) {  // (pa / pb may point into the arena: re-derive them after growing)
            const size_t oa = pa && pa != &sa ? (size_t)(pa - arena.data()) : 0, ob = pb && pb != &sb ? (size_t)(pb - arena.data()) : 0;
            arena.reserve(std::max(arena.capacity() * 2, at + la + lb));
            if (pa && pa != &sa) pa = arena.data() + oa;
            if (pb && pb != &sb) pb = arena.data() + ob;
        }
        arena.resize(at + la + lb);
        int32_t* out = arena.data() + at;  // merge of two sorted lists, duplicates dropped
        uint32_t x = 0, y = 0, m = 0;
        while (x < la || y < lb) {
            int32_t w;
            if (y >= lb || (x < la && pa[x] <= pb[y])) { w = pa[x]; if (y < lb && pb[y] == w) ++y; ++x; }
            else w = pb[y++];
            out[m++] = w;
        }
        arena.resize(at + m);
        c_off[v] = (uint32_t)at;
        c_len[v] = m;
    }
    // per source item (residue nodes and the entry n): its target set -> join or single target.  Equal sets share a join:
    // found through a hash of the list (open addressing), verified by comparing the lists.
    forward_.assign(n + 1, KGraph::kNone);
    dangling_.assign(n + 1, 0);
    fan_first_.assign(1, 0);
    size_t table_size = 16;
    while (table_size < (size_t)n * 2 + 2) table_size <<= 1;
    std::vector<int32_t> join_slot(table_size, -1);  // -> index of the join (0-based)
    size_t n_joins = 0;
    for (int32_t u = 0; u <= n; ++u) {
        const int32_t* t = nullptr;
        uint32_t len = 0;
        int32_t self = 0;
        if (u == n) list_of(0, &t, &len, &self, dangling_[u]);
        else if (!is_epsilon(g_.label[u]) && g_.label[u] != KGraph::kMatch) list_of(g_.next_a[u], &t, &len, &self, dangling_[u]);  // residues and gaps
        else continue;
        if (len == 0) continue;
        if (len == 1) { forward_[u] = t[0]; continue; }
        uint64_t hsh = 0xcbf29ce484222325ULL;
        for (uint32_t i = 0; i < len; ++i) hsh = (hsh ^ (uint64_t)(uint32_t)t[i]) * 0x100000001b3ULL;
        size_t at = (size_t)(hsh * 0x9E3779B97F4A7C15ULL >> 20) & (table_size - 1);
        int32_t found = -1;
        for (;; at = (at + 1) & (table_size - 1)) {
            const int32_t j = join_slot[at];
            if (j < 0) break;
            const uint32_t lo = fan_first_[j], hi = fan_first_[j + 1];
            if (hi - lo == len && std::equal(t, t + len, fan_.begin() + lo)) { found = j; break; }
        }
        if (found < 0) {
            found = (int32_t)n_joins++;
            join_slot[at] = found;
            fan_.insert(fan_.end(), t, t + len);
            fan_first_.push_back((uint32_t)fan_.size());
        }
        forward_[u] = n + 1 + found;
    }
    const int32_t items = n + 1 + (int32_t)n_joins;
    forward_.resize(items, KGraph::kNone);
    dangling_.resize(items, 0);
    // topological order of the derived graph (Kahn): edges u -> forward_[u], join -> its targets
    std::vector<int32_t> indeg(items, 0);
    auto targets_of = [&](int32_t item, auto&& fn) {
        if (item > n) { for (uint32_t i = fan_first_[item - n - 1]; i < fan_first_[item - n]; ++i) fn(fan_[i]); }
        else if (forward_[item] != KGraph::kNone) fn(forward_[item]);
    };
    std::vector<uint8_t> live(items, 0);
    for (int32_t v = 0; v < n; ++v) live[v] = !is_epsilon(g_.label[v]);
    for (int32_t v = n; v < items; ++v) live[v] = 1;
    for (int32_t v = 0; v < items; ++v)
        if (live[v]) targets_of(v, [&](int32_t t) { ++indeg[t]; });
    // A residue node's states have pairwise different keys, and rolling one residue into all of them
    // can make keys collide; a join's states are already merged.  So only items fed by exactly one
    // JOIN receive collision-free arrivals.
    single_source_.assign(items, 0);
    {
        std::vector<int32_t> from_joins(items, 0);
        for (int32_t j = n + 1; j < items; ++j) targets_of(j, [&](int32_t t) { ++from_joins[t]; });
        for (int32_t v = 0; v < items; ++v) single_source_[v] = indeg[v] == 1 && from_joins[v] == 1;
    }
    std::vector<int32_t> ready;
    for (int32_t v = 0; v < items; ++v)
        if (live[v] && indeg[v] == 0) ready.push_back(v);
    for (size_t at = 0; at < ready.size(); ++at) {
        const int32_t v = ready[at];
        if (v != n) order_.push_back(v);
        targets_of(v, [&](int32_t t) { if (--indeg[t] == 0) ready.push_back(t); });
    }
    table_.resize(items);
    // state keys are (k-1) symbols plus the length marker bit: small enough for a directly indexed merge table?
    direct_key_bits_ = (enc_.k() - 1) * enc_.bits_per_symbol() + 1 <= 20 ? (enc_.k() - 1) * enc_.bits_per_symbol() + 1 : 0;
    input_of_.assign(items, KGraph::kNone);
    readers_.assign(n_joins, 0);
    refs_.assign(TXQ_SLOT_FIRST_FREE, kPinned);
    if (dense_ok_) compute_static_shapes();
    OpVec none;
    hand_on(n, State{0, TXQ_SLOT_ONES, 0, 0, 0, 0, 0}, none);
}

// Which codes can stand at each position of the (k-1)-suffix of a full-length state waiting at an item: the residues of
// the nodes k-1 .. 1 steps back along any path.  One pass in topological order: a residue node shifts the sets of its
// input by one position and puts its own code last; joins, Match and '$' pass them on; the entry and Gap nodes (a state
// restarts its k-mer there) contribute nothing — a state that leaves them is not full-length before it has passed k-1
// residue nodes, each of which then puts its code where it belongs.
void QueryExpansion::compute_static_shapes() {
    const size_t items = table_.size();
    static_shape_.assign(items, Geometry{});
    auto feed = [&](int32_t to, const Geometry& out) {
        if (to == KGraph::kNone) return;
        for (unsigned j = 0; j < dense_pos_; ++j) static_shape_[to][j] |= out[j];
    };
    auto forward_all = [&](int32_t item, const Geometry& out) {
        if (item > n_nodes_) { for (uint32_t i = fan_first_[item - n_nodes_ - 1]; i < fan_first_[item - n_nodes_]; ++i) feed(fan_[i], out); }
        else feed(forward_[item], out);
    };
    forward_all(n_nodes_, Geometry{});
    for (const int32_t item : order_) {
        Geometry out{};
        if (item > n_nodes_) out = static_shape_[item];
        else {
            const int32_t lab = g_.label[item];
            if (lab == KGraph::kMatch) continue;
            if (lab == KGraph::kGap) out = Geometry{};
            else if (g_.takes_residue(item)) {
                for (unsigned j = 0; j + 1 < dense_pos_; ++j) out[j] = static_shape_[item][j + 1];
                out[dense_pos_ - 1] = code_mask(item);
            } else out = static_shape_[item];
        }
        forward_all(item, out);
    }
}

// the geometry of the blocks of the list at `item`: tracked programs lay a block out inside the list's static shape,
// untracked ones over the whole alphabet (entry index = the suffix as a number in base A, as the step kernel of
// untracked blocks computes it)
QueryExpansion::Geometry QueryExpansion::geometry_of(int32_t item, unsigned phase) const {
    Geometry g{};
    if (tracked_) {
        g = static_shape_[item];
        for (unsigned j = 0; j < dense_pos_; ++j)
            if (!g[j] || j + phase < dense_pos_) g[j] = 1u;  // a position the states have not filled yet holds code 0, like their k-mer values
    } else
        for (unsigned j = 0; j < dense_pos_; ++j) g[j] = dense_a_ >= 32 ? 0xFFFFFFFFu : ((1u << dense_a_) - 1u);
    return g;
}

uint32_t QueryExpansion::capacity_of(const Geometry& g) const {
    if (!tracked_) return (uint32_t)dense_n_;
    uint64_t n = 1;
    for (unsigned j = 0; j < dense_pos_; ++j) n *= (uint64_t)__builtin_popcount(g[j]);
    uint64_t cap = 64;
    while (cap < n) cap <<= 1;
    return (uint32_t)std::min<uint64_t>(cap, (uint64_t)1 << TXQ_DENSE_BLOCK_SHIFT);  // (n <= A^(k-1) <= 2^22)
}

// the codes of the residues a residue node or fused class stands for
uint32_t QueryExpansion::code_mask(int32_t node) const {
    uint32_t m = 0;
    g_.for_each_residue(node, [&](unsigned char c) { m |= 1u << enc_.code(c); });
    return m;
}

// a state leaves item `from` (a residue node or the entry): to its join or only target
void QueryExpansion::hand_on(int32_t from, State s, OpVec& out) {
    if (dangling_[from]) throw std::runtime_error("k-graph node without successor (the reference fails here too)");
    if (forward_[from] == KGraph::kNone) { drop(s.slot); return; }
    arrive(forward_[from], s, out);
}

uint32_t QueryExpansion::fresh() {
    uint32_t s;
    if (free_head_ < free_.size()) {
        s = free_[free_head_++];
        if (free_head_ == free_.size()) { free_.clear(); free_head_ = 0; }
    } else { s = (uint32_t)refs_.size(); refs_.push_back(0); }
    refs_[s] = 1;
    if (s + 1 > high_water_) high_water_ = s + 1;
    return s;
}
void QueryExpansion::share(uint32_t s) { if (refs_[s] != kPinned) ++refs_[s]; }
void QueryExpansion::drop(uint32_t s) {
    if (refs_[s] == kPinned) return;
    if (--refs_[s] == 0) parked_.push_back(s);
}
bool QueryExpansion::exclusive(uint32_t s) const { return refs_[s] == 1; }

void QueryExpansion::emit(OpVec& out, uint32_t kmer, uint32_t dst, uint32_t a, uint32_t b) {
    if (++total_ops_ > limits_.max_ops) throw std::runtime_error("query expands to too many mask operations");
    out.push_back(txq_op{kmer, dst, a, b});
}

// length-prefixed key: the symbols seen so far (at most the k-1 newest) with a marker bit just above them, so paths of
// different length < k-1 never share a key; a state collecting the residues after a gap is identified by its partial d-gram
// and how many residues it has seen (bits 60-62 keep it apart from every ordinary state)
uint64_t QueryExpansion::key_of(const State& s) const {
    const unsigned k = enc_.k(), bits = enc_.bits_per_symbol();
    const unsigned phase = s.shift < k - 1 ? s.shift : k - 1;
    return s.gapped ? (s.kmer | ((uint64_t)(4 + s.shift) << 60)) : ((s.kmer & enc_.suffix_mask()) | (1ULL << (phase * bits)));
}

// hand a state (owning one reference to its slot) to node `to`
void QueryExpansion::arrive(int32_t to, State s, OpVec& out) {
    NodeStates& ns = table_[to];
    if (ns.items.capacity() == 0) adopt_storage(ns);
    if (waiting_ >= limits_.max_live_states) throw std::runtime_error("query holds too many states at the same time");
    if (single_source_[to] || ns.append_only) {  // nothing to merge with, or merging does not pay: no table look-up
        s.asked = 0;
        ns.items.push_back(s);
        ++waiting_;
        if (++states_ > limits_.max_states) throw std::runtime_error("query expands to too many states");
        return;
    }
    const uint64_t key = key_of(s);
    // A list of a few states is searched, not hashed: the merge table (an allocation per node) is made when the list reaches
    // kSearched states — along a run of literals every list holds one state, and the tables were a third of its expansion.
    uint32_t at = kNoState;
    if (ns.by_key.size() == 0 && ns.items.size() < kSearched) {
        for (uint32_t i = 0; i < ns.items.size(); ++i)
            if (key_of(ns.items[i]) == key) { at = i; break; }
        if (at == kNoState && ns.items.size() + 1 == kSearched) {  // the list outgrows the search with this state: everybody into the table
            ns.by_key.want_direct(direct_key_bits_);
            for (uint32_t i = 0; i < ns.items.size(); ++i) ns.by_key.emplace(key_of(ns.items[i]), i);
            ns.by_key.emplace(key, (uint32_t)ns.items.size());
        }
    } else {
        ns.by_key.want_direct(direct_key_bits_);
        auto [where, inserted] = ns.by_key.emplace(key, (uint32_t)ns.items.size());
        if (!inserted) at = *where;
    }
    if (at == kNoState) {
        s.asked = 0;
        ns.items.push_back(s);
        ++waiting_;
        if (++states_ > limits_.max_states) throw std::runtime_error("query expands to too many states");
        return;
    }
    State& have = ns.items[at];
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

// Will the arrivals that a union of residue nodes makes out of `list` merge at the node after it?  Two
// of them merge exactly when their states agree in everything but the oldest symbol of the key, so the
// answer is in the list itself: a sample taken BY KEY (one sixteenth of the key space, so both partners
// of a pair are in or out together) counts how many states have such a partner.
// lists shorter than this are not worth the question (TETREX_MERGE_SAMPLE lowers it so that small tests reach the code)
size_t QueryExpansion::merge_sample_threshold() {
    static const size_t v = [] {
        const char* e = std::getenv("TETREX_MERGE_SAMPLE");
        return e && std::atoll(e) > 0 ? (size_t)std::atoll(e) : (size_t)kMergeSample;
    }();
    return v;
}

bool QueryExpansion::merging_pays(const StateVec& list) {
    const unsigned k = enc_.k(), bits = enc_.bits_per_symbol();
    const uint64_t rest_mask = enc_.suffix_mask() >> bits;
    if (!rest_mask) return true;
    FlatMap seen;
    if (!spare_maps_.empty()) { std::swap(seen, spare_maps_.back()); spare_maps_.pop_back(); }
    uint32_t sampled = 0, partners = 0;
    for (const State& s : list) {
        if (s.gapped || s.shift < k - 1) { sampled = 0; break; }  // not the uniform case: keep merging
        const uint64_t rest = s.kmer & rest_mask;
        if ((rest * 0x9E3779B97F4A7C15ULL) >> 60) continue;
        ++sampled;
        if (!seen.emplace(rest, 0).second) ++partners;
    }
    seen.clear();
    if (seen.capacity() && spare_maps_.size() < 64) { spare_maps_.emplace_back(); std::swap(spare_maps_.back(), seen); }
    const uint32_t enough = merge_sample_threshold() < kMergeSample ? 1 : 64;
    return sampled < enough || partners * 4 >= sampled;  // a quarter or more of the states would be absorbed
}

void QueryExpansion::adopt_storage(NodeStates& ns) {
    if (!spare_items_.empty()) { ns.items.swap(spare_items_.back()); spare_items_.pop_back(); }
    if (ns.by_key.capacity() == 0 && !spare_maps_.empty()) { std::swap(ns.by_key, spare_maps_.back()); spare_maps_.pop_back(); }
}

// ---- dense blocks (include/txq_program.h, version 3) -----------------------------------------

// entry of a suffix in a block: the ranks of its codes within the block's geometry, as a mixed-radix number
uint64_t QueryExpansion::index_of_codes(uint32_t block, const unsigned* code) const {
    const Geometry& g = block_geom_[block];
    uint64_t idx = 0;
    for (unsigned j = 0; j < dense_pos_; ++j) {
        if (!((g[j] >> code[j]) & 1u)) throw std::logic_error("state outside the geometry of its dense block");
        idx = idx * (uint64_t)__builtin_popcount(g[j]) + (uint64_t)__builtin_popcount(g[j] & ((1u << code[j]) - 1u));
    }
    return idx;
}
uint64_t QueryExpansion::dense_index(uint32_t block, uint64_t kmer) const {
    const unsigned bits = enc_.bits_per_symbol();
    const uint64_t sym = enc_.symbol_mask();
    unsigned code[TXQ_DENSE_MAX_POSITIONS];
    for (unsigned j = 0; j < dense_pos_; ++j) code[j] = (unsigned)((kmer >> (bits * (dense_pos_ - 1 - j))) & sym);
    return index_of_codes(block, code);
}

uint64_t QueryExpansion::shape_entries(const DenseRef& r) const {
    uint64_t n = 1;
    for (unsigned j = 0; j < dense_pos_; ++j) n *= (uint64_t)__builtin_popcount(r.shape[j]);
    return n;
}

// makes sure that the next new_block() calls for blocks of these capacities succeed (blocks come from the free list of
// their capacity or get a new id)
bool QueryExpansion::can_take_blocks(const std::vector<uint32_t>& caps) {
    if (caps.empty()) return true;
    // The most recently released blocks stay out of circulation while new ones can be had: a recycled block must be zeroed
    // AFTER its last reader, which puts its DENSE_ZERO one level behind the step that read it and the next step one more
    // level behind that — a chain of steps ping-ponging between two blocks needs two levels per step instead of one.
    std::map<uint32_t, size_t> want;
    for (uint32_t c : caps) ++want[c];
    size_t more_ids = 0;
    int64_t bytes = 0;
    bool cooling_will_do = true;
    for (const auto& [cap, n] : want) {
        const auto it = free_by_cap_.find(cap);
        const size_t in_list = it == free_by_cap_.end() ? 0 : it->second.ids.size() - it->second.head;
        const size_t available = in_list > dense_.cool_down ? in_list - dense_.cool_down : 0;
        if (n > available) {
            more_ids += n - available;
            bytes += (int64_t)((n - available) * (uint64_t)cap * (dense_.slot_bytes ? dense_.slot_bytes : 128));
        }
        cooling_will_do = cooling_will_do && n <= in_list;
    }
    if (!more_ids) return true;
    if (n_blocks_ + more_ids > dense_.max_blocks) return cooling_will_do;  // at the cap: the cooling ones will do, if there are enough
    if (dense_.pool) {
        if (dense_.pool->fetch_sub(bytes, std::memory_order_relaxed) - bytes < 0) {
            dense_.pool->fetch_add(bytes, std::memory_order_relaxed);
            return cooling_will_do;  // no memory for new blocks
        }
        pool_taken_ += (uint64_t)bytes;  // the stage driver hands them back once the device has run the query's last ops
    }
    for (const auto& [cap, n] : want) {
        FreeList& fl = free_by_cap_[cap];
        const size_t in_list = fl.ids.size() - fl.head;
        const size_t available = in_list > dense_.cool_down ? in_list - dense_.cool_down : 0;
        for (size_t i = available; i < n; ++i) {  // new blocks go to the FRONT: they are taken before the released ones that are still cooling
            fl.ids.insert(fl.ids.begin() + (std::ptrdiff_t)fl.head, (uint32_t)n_blocks_++);
            block_refs_.push_back(0);
            block_cap_.push_back(cap);
            block_geom_.push_back(Geometry{});
        }
    }
    return true;
}

uint32_t QueryExpansion::new_block(OpVec& out, const Geometry& geom) {
    const uint32_t cap = capacity_of(geom);
    FreeList& fl = free_by_cap_[cap];
    if (fl.head >= fl.ids.size()) throw std::logic_error("dense block taken without a reservation");
    const uint32_t b = fl.ids[fl.head++];
    if (fl.head == fl.ids.size()) { fl.ids.clear(); fl.head = 0; }
    block_refs_[b] = 1;
    block_geom_[b] = geom;
    txq_dense_op z{};
    z.kind = TXQ_DENSE_ZERO;
    z.dst = dense_slot(b, 0);
    if (tracked_) {  // (re)creates the block inside this geometry
        z.src = cap;
        z.r_mask = 1;
        for (unsigned j = 0; j < dense_pos_; ++j) z.shape[j] = geom[j];
    }
    emit_dense(out, z);
    if (zero_at_.size() <= b) { zero_at_.resize(b + 1, 0); zero_epoch_.resize(b + 1, 0); }
    zero_at_[b] = (uint32_t)(dense_out_->size() - 1);
    zero_epoch_[b] = dense_epoch_;
    return b;
}

// The list that owns block r is about to be consumed: its shape is final, and nothing ever reads (or accumulates into)
// an entry outside it.  If the block's DENSE_ZERO is still in the stage's table, it is told to zero that shape instead
// of all A^(k-1) slots (31.8 M slots = 4 GB per 1000-motif batch otherwise, a sixth of the dense kernel's bytes).
void QueryExpansion::shape_zero(const DenseRef& r) {
    if (tracked_) return;  // a tracked ZERO carries the block's geometry, and clears what is listed
    if (!dense_out_ || r.block >= zero_at_.size() || zero_epoch_[r.block] != dense_epoch_ || zero_at_[r.block] >= dense_out_->size()) return;
    txq_dense_op& z = (*dense_out_)[zero_at_[r.block]];
    if (z.kind != TXQ_DENSE_ZERO || z.dst != dense_slot(r.block, 0)) return;
    for (unsigned j = 0; j < dense_pos_; ++j) z.shape[j] = r.shape[j];
    z.r_mask = 1;  // "the shape is given"
}

// the block will receive more states in a later stage, when its ZERO is out of reach: the ZERO clears the whole block again
void QueryExpansion::shape_unzero(const DenseRef& r) {
    if (tracked_) return;
    if (!dense_out_ || r.block >= zero_at_.size() || zero_epoch_[r.block] != dense_epoch_ || zero_at_[r.block] >= dense_out_->size()) return;
    txq_dense_op& z = (*dense_out_)[zero_at_[r.block]];
    if (z.kind != TXQ_DENSE_ZERO || z.dst != dense_slot(r.block, 0)) return;
    z.r_mask = 0;  // "everything"
}

void QueryExpansion::release_block(uint32_t block) {
    if (--block_refs_[block] == 0) parked_blocks_.push_back(block);  // reusable once the current item is finished
}

// Do this query's blocks carry live lists?  Decided once, with its first block: where the run knows that states thin out on
// this index and the executor keeps lists.
void QueryExpansion::decide_tracking() {
    if (tracked_decided_) return;
    tracked_decided_ = true;
    if (!dense_.tracked_ok || dense_.tracked_force < 0) return;
    // ... and wherever an untracked block (A^(k-1) entries) would be tens of megabytes: 21^5 masks of 128 bytes are half a
    // gigabyte, a tracked block of the same list a few hundred kilobytes — pushing from a full list costs little more than
    // pulling, allocating (and zeroing) gigabytes per query costs seconds
    const bool huge = dense_n_ * (dense_.slot_bytes ? dense_.slot_bytes : 128) >= ((uint64_t)64 << 20);
    tracked_ = dense_.tracked_force > 0 || huge || (dense_.evidence && dense_.evidence->load(std::memory_order_relaxed) == DenseOptions::kThin);
}

void QueryExpansion::emit_dense(OpVec& out, const txq_dense_op& d0) {
    if (!dense_out_) throw std::logic_error("dense op without a dense table");
    txq_dense_op d = d0;
    if (tracked_) d.reserved |= TXQ_DENSE_TRACKED;
    dense_out_->push_back(d);
    dense_seen_ = dense_out_->size();
    emit(out, TXQ_DENSE_OP, (uint32_t)(dense_out_->size() - 1), 0, 0);
    if (d.kind == TXQ_DENSE_STEP) ++dense_steps_;
}

// the block this list accumulates into (created on first use; the caller has reserved it)
QueryExpansion::DenseRef* QueryExpansion::owned_block(int32_t item, NodeStates& ns, OpVec& out, unsigned phase) {
    for (DenseRef& r : ns.dense)
        if (r.owned && r.phase == phase) return &r;
    DenseRef r{};
    r.block = new_block(out, geometry_of(item, phase));
    r.owned = 1;
    r.phase = phase;
    ns.dense.push_back(r);
    return &ns.dense.back();
}

// How many suffixes of its shape a list may have per state it holds and still become a block (0: never).  A step visits
// every suffix of the shape, alive or not, at (h + 1 ~ 4 loads) x the mask's bytes over ~4 TB/s per predecessor on the
// device — 0.13 ns for 1024 bins, 8 ns for 65536; an enumerated state costs that as well, plus ~12 ns of the batch's
// host time per residue.  So narrow masks pay for a block while the shape is up to 64 times the list (at k = 4 that is
// any list worth the question; at k = 6, 21^5 suffixes, only lists of tens of thousands of states), 8 KiB masks only
// when the list fills half of its shape.  Where the run has learned that states thin out on this index (most entries
// of a shape are dead a few residues on, and every later step keeps visiting them) the bar is four times higher.
uint64_t QueryExpansion::shape_limit() const {
    const double t_dev = (double)(dense_.slot_bytes ? dense_.slot_bytes : 128) / 1000.0;
    double per_state = (dense_.host_ns_per_op + t_dev) / t_dev;
    if (dense_.evidence && dense_.evidence->load(std::memory_order_relaxed) >= DenseOptions::kSparse) per_state /= 4;
    if (per_state < 1) return 0;
    return std::min<uint64_t>((uint64_t)per_state, dense_.max_shape_per_state);
}

// a list with many full-length states becomes (part of) a block: one scatter op per state now instead of
// one op per state and residue at every later step
void QueryExpansion::densify(int32_t item, NodeStates& ns, OpVec& out, bool may_hold_duplicates) {
    const uint32_t min_states = min_states_now();
    if (!dense_ok_ || ns.items.size() < min_states) return;
    decide_tracking();
    const unsigned k = enc_.k(), bits = enc_.bits_per_symbol();
    const uint64_t sym = enc_.symbol_mask();
    // full-length states (k-1 residues and more) go to the list's ordinary block; a tracked program also makes blocks of the
    // states that are still filling their first k-mer, one per length
    for (unsigned phase = tracked_ ? 1 : k - 1; phase <= k - 1; ++phase) {
        auto of_phase = [&](const State& s) { return !s.gapped && (phase == k - 1 ? s.shift >= k - 1 : s.shift == phase); };
        size_t full = 0;
        uint32_t shape[TXQ_DENSE_MAX_POSITIONS] = {};
        for (const State& s : ns.items) {
            if (!of_phase(s)) continue;
            ++full;
            for (unsigned j = 0; j < dense_pos_; ++j) shape[j] |= 1u << ((s.kmer >> (bits * (dense_pos_ - 1 - j))) & sym);
        }
        if (full < min_states) continue;
        // the shape (the product of the per-position code sets) against the states it holds: shape_limit() — a tracked block
        // costs what its living entries cost, whatever the shape
        uint64_t product = 1;
        for (unsigned j = 0; j < dense_pos_; ++j) product *= (uint64_t)__builtin_popcount(shape[j]);
        if (!tracked_) {
            const uint64_t limit = shape_limit();
            if (limit == 0 || product > limit * full) continue;
        }
        if (!has_owned(ns, phase)) {
            caps_scratch_.assign(1, capacity_of(geometry_of(item, phase)));
            if (!can_take_blocks(caps_scratch_)) continue;
        }
        DenseRef* own = owned_block(item, ns, out, phase);
        // States that all carry ONE mask and fill their shape exactly — the states behind a run of wildcards that have not been
        // probed yet (they share ONES) — are spread by a single FILL.  (An append-only list may hold a key twice: no counting there.)
        bool uniform = !may_hold_duplicates && product == full;
        uint32_t the_slot = 0;
        if (uniform) {
            bool first = true;
            for (const State& s : ns.items) {
                if (!of_phase(s)) continue;
                if (first) { the_slot = s.slot; first = false; }
                else if (s.slot != the_slot) { uniform = false; break; }
            }
        }
        for (unsigned j = 0; j < dense_pos_; ++j)
            if (shape[j] & ~block_geom_[own->block][j]) throw std::logic_error("state list outside the geometry of its dense block");
        if (uniform) {
            txq_dense_op f{};
            f.kind = TXQ_DENSE_FILL;
            f.dst = dense_slot(own->block, 0);
            f.src = the_slot;
            for (unsigned j = 0; j < dense_pos_; ++j) f.shape[j] = shape[j];
            emit_dense(out, f);
        }
        size_t kept = 0;
        for (size_t i = 0; i < ns.items.size(); ++i) {
            const State s = ns.items[i];
            if (!of_phase(s)) { ns.items[kept++] = s; continue; }
            if (!uniform) {
                const uint32_t e = dense_slot(own->block, dense_index(own->block, s.kmer));
                emit(out, TXQ_NO_KMER, e, e, s.slot);  // block[e] |= state (an append-only list may hold one key twice)
            }
            drop(s.slot);
        }
        for (unsigned j = 0; j < dense_pos_; ++j) own->shape[j] |= shape[j];
        ns.items.resize(kept);
    }
}

// blocks of `item` whose shape has shrunk to a few entries (or all of them, when no block can be had for
// their successors) go back to enumerated states: one copy op per entry of the shape
void QueryExpansion::materialise(int32_t item, OpVec& out, bool all) {
    NodeStates& ns = table_[item];
    const unsigned k = enc_.k(), bits = enc_.bits_per_symbol();
    std::vector<DenseRef> keep, take;
    for (const DenseRef& r : ns.dense) (all || small_enough(r) ? take : keep).push_back(r);
    if (take.empty()) return;
    ns.dense.swap(keep);
    for (const DenseRef& r : take) {
        // odometer over shape[0] x .. x shape[k-2]
        unsigned code[TXQ_DENSE_MAX_POSITIONS];
        bool empty = false;
        for (unsigned j = 0; j < dense_pos_; ++j) {
            if (!r.shape[j]) { empty = true; break; }
            code[j] = (unsigned)__builtin_ctz(r.shape[j]);
        }
        while (!empty) {
            uint64_t kmer = 0;
            for (unsigned j = 0; j < dense_pos_; ++j) kmer = (kmer << bits) | code[j];
            const uint64_t idx = index_of_codes(r.block, code);
            const uint32_t d = fresh();
            emit(out, TXQ_NO_KMER, d, dense_slot(r.block, idx), TXQ_SLOT_ZERO);
            arrive(item, State{kmer, d, (uint8_t)(r.phase < dense_pos_ ? r.phase : k), 0, 0, 0, 0}, out);
            unsigned j = dense_pos_;
            for (;;) {
                if (j == 0) { empty = true; break; }
                --j;
                const uint32_t higher = r.shape[j] & ~((2u << code[j]) - 1u);
                if (higher) { code[j] = (unsigned)__builtin_ctz(higher); break; }
                code[j] = (unsigned)__builtin_ctz(r.shape[j]);
            }
        }
        release_block(r.block);
    }
}

// one collector round for all states of `src` and the residues of r_mask, accumulated into the receiver's block
void QueryExpansion::dense_step(const DenseRef& src, uint32_t r_mask, int32_t receiver, OpVec& out) {
    if (receiver == KGraph::kNone || !r_mask) return;
    NodeStates& rs = table_[receiver];
    DenseRef* own = owned_block(receiver, rs, out, std::min<unsigned>(src.phase + 1, dense_pos_));
    txq_dense_op d{};
    d.kind = TXQ_DENSE_STEP;
    if (src.phase < dense_pos_) d.reserved = TXQ_DENSE_NOPROBE;  // the states are still filling their first k-mer: nothing to look up
    d.dst = dense_slot(own->block, 0);
    d.src = dense_slot(src.block, 0);
    d.r_mask = r_mask;
    for (unsigned j = 0; j < dense_pos_; ++j) d.shape[j] = src.shape[j];
    emit_dense(out, d);
    for (unsigned j = 0; j + 1 < dense_pos_; ++j) own->shape[j] |= src.shape[j + 1];
    own->shape[dense_pos_ - 1] |= r_mask;
    for (unsigned j = 0; j < dense_pos_; ++j)
        if (own->shape[j] & ~block_geom_[own->block][j]) throw std::logic_error("dense step leaves the geometry of its destination block");
}

// the lists that dense steps out of `item` accumulate into
void QueryExpansion::dense_receivers(int32_t item, std::vector<int32_t>& out) const {
    out.clear();
    auto add = [&](int32_t r) {
        if (r != KGraph::kNone && std::find(out.begin(), out.end(), r) == out.end()) out.push_back(r);
    };
    if (item > n_nodes_) {
        for (uint32_t i = fan_first_[item - n_nodes_ - 1]; i < fan_first_[item - n_nodes_]; ++i) {
            const int32_t t = fan_[i];
            if (g_.takes_residue(t)) add(forward_[t]);
        }
    } else if (g_.takes_residue(item)) add(forward_[item]);
}

QueryExpansion::QueryExpansion(const KmerEncoder& enc, std::string literal, CompileLimits limits)
    : literal_mode_(true), literal_(std::move(literal)), enc_(enc), limits_(limits) {
    if (enc_.k() < 2) throw std::runtime_error("k must be at least 2");
    if (literal_.empty()) throw std::runtime_error("empty query: nothing to search for");
    refs_.assign(TXQ_SLOT_FIRST_FREE, kPinned);
}

void QueryExpansion::advance(size_t op_budget, Intern intern, OpVec& out, KmerTable* dgrams, bool verified_only, DenseVec* dense) {
    const unsigned k = enc_.k();
    if (literal_mode_) {
        // one state from the start to the Match node: the first k - 1 residues fill its k-mer, every residue after them is a probe
        // ANDed into its mask (a slot of its own from the first probe on), the Match node ORs the mask into RESULT
        if (literal_done_) return;
        uint64_t kmer = 0;
        uint32_t slot = TXQ_SLOT_ONES;
        unsigned shift = 0;
        for (const char c : literal_) {
            const uint64_t probe = enc_.roll((unsigned char)c, kmer);
            if (shift < k - 1) { ++shift; continue; }
            const uint32_t id = intern.intern(probe);
            ++probes_;
            if (slot == TXQ_SLOT_ONES) { slot = fresh(); emit(out, id, slot, TXQ_SLOT_ONES, TXQ_SLOT_ZERO); }
            else emit(out, id, slot, slot, TXQ_SLOT_ZERO);
        }
        emit(out, TXQ_NO_KMER, TXQ_SLOT_RESULT, slot, TXQ_SLOT_RESULT);
        drop(slot);
        states_ += literal_.size() + 1;
        literal_done_ = true;
        return;
    }
    const size_t start = out.size();
    dense_out_ = dense;
    if (dense && dense->size() < dense_seen_) {  // the stage driver has shipped the table: earlier ZERO ops are out of reach
        ++dense_epoch_;
        dense_seen_ = dense->size();
    }
    const bool go_dense = dense_ok_ && dense != nullptr;
    if (evidence_asked_) wants_evidence_ = false;  // the pause lasted one stage: its answers are in (observe), or somebody else's are
    while (cursor_ < order_.size() && out.size() - start < op_budget) {
        if (verified_only) {
            // Expand only what the device has confirmed alive: an item whose input still holds a state
            // created since the last feedback waits for the next stage.  This is the reference's
            // immediate `none()` pruning at the granularity of one DP level per stage — without it a
            // sparse index makes the host expand (and the device probe) whole sub-trees of dead states.
            const int32_t next = order_[cursor_];
            const StateVec& waiting = input_of_[next] != KGraph::kNone ? table_[input_of_[next]].items : table_[next].items;
            bool unverified = false;
            for (const State& s : waiting)
                if (!s.asked && s.slot >= TXQ_SLOT_FIRST_FREE) { unverified = true; break; }
            if (unverified) break;
        }
        if (!parked_.empty()) {  // slots freed by the previous item become reusable now
            free_.insert(free_.end(), parked_.begin(), parked_.end());
            parked_.clear();
        }
        if (!parked_blocks_.empty()) {
            for (uint32_t b : parked_blocks_) free_by_cap_[block_cap_[b]].ids.push_back(b);
            parked_blocks_.clear();
        }
        // a fused class that stopped between two of its residues when the budget ran out goes on where it stopped
        const bool resuming = resume_item_ != KGraph::kNone;
        bool densify_here = false;
        if (dense_ok_ && !resuming) {
            // Dense part of this item's input.  First, shapes that have shrunk to a few entries are enumerated again (so is
            // everything when this call has nowhere to put dense ops).  Then the blocks the item will need are reserved: one
            // per list its steps accumulate into that has none yet, plus its own if its enumerated states are about to become
            // a block.  If they cannot be had (budget), the rest of the dense input is enumerated as well.
            const int32_t next = order_[cursor_];
            NodeStates& cur = table_[next];
            for (const DenseRef& d : cur.dense)
                if (d.owned) shape_zero(d);
            if (!cur.dense.empty()) materialise(next, out, !go_dense);
            bool may_densify = go_dense && input_of_[next] == KGraph::kNone && cur.items.size() >= min_states_now();
            if (may_densify && dense_.evidence) {
                const int ev = dense_.evidence->load(std::memory_order_relaxed);
                if (ev != DenseOptions::kUnknown) wants_evidence_ = false;
                if (ev == DenseOptions::kUnknown && !evidence_asked_) {
                    // the first list of this query that could become a block, and nobody knows yet how states fare on this
                    // index: stop here for this stage and ask (the waiting states are this stage's questions) — if the list
                    // has something to tell: states that have been probed at least once (a list right behind leading
                    // wildcards or residue classes holds none, its masks are still all ones)
                    size_t probed = 0;
                    for (const State& s : cur.items) probed += !s.gapped && s.shift >= k;
                    if (probed >= 16) {
                        evidence_asked_ = wants_evidence_ = true;
                        // (the item waits for the next stage, where its own states may still join its block — with residues its
                        // block's ZERO, shaped a few lines up and shipped with this stage, would not have cleared)
                        for (const DenseRef& d : cur.dense)
                            if (d.owned) shape_unzero(d);
                        break;
                    }
                }
                // (still unknown after asking: as if they saturate; what the run has learned sets the bar in densify())
            }
            if (!cur.dense.empty() || may_densify) {
                if (go_dense) decide_tracking();
                // the lengths (phases) of the states this item will hold as blocks: those it holds already, and those densify() will make
                uint32_t phases = 0, fresh_phases = 0;
                for (const DenseRef& d : cur.dense) phases |= 1u << d.phase;
                if (may_densify) {
                    size_t count[TXQ_DENSE_MAX_POSITIONS + 1] = {};
                    for (const State& s : cur.items)
                        if (!s.gapped) ++count[s.shift < dense_pos_ ? s.shift : dense_pos_];
                    for (unsigned p = tracked_ ? 1 : dense_pos_; p <= dense_pos_; ++p)
                        if (count[p] >= min_states_now()) { phases |= 1u << p; if (!has_owned(cur, p)) fresh_phases |= 1u << p; }
                }
                dense_receivers(next, receivers_scratch_);
                caps_scratch_.clear();
                for (int32_t r : receivers_scratch_)
                    for (unsigned p = 1; p <= dense_pos_; ++p) {
                        if (!((phases >> p) & 1u)) continue;
                        const unsigned q = std::min(p + 1, dense_pos_);
                        if (p < dense_pos_ && q == dense_pos_ && ((phases >> dense_pos_) & 1u)) continue;  // (counted with phase k-1 itself)
                        if (!has_owned(table_[r], q)) caps_scratch_.push_back(capacity_of(geometry_of(r, q)));
                    }
                for (unsigned p = 1; p <= dense_pos_; ++p)
                    if ((fresh_phases >> p) & 1u) caps_scratch_.push_back(capacity_of(geometry_of(next, p)));
                const bool ok = go_dense && can_take_blocks(caps_scratch_);
                densify_here = ok && may_densify;
                if (!ok && !cur.dense.empty()) materialise(next, out, true);
            }
        }
        const int32_t item = order_[cursor_++];
        NodeStates ns;
        ns.items.swap(table_[item].items);
        ns.dense.swap(table_[item].dense);
        if (!resuming) waiting_ -= ns.items.size();
        if (densify_here) densify(item, ns, out, table_[item].append_only);
        for (const DenseRef& d : ns.dense)  // (densify has just added the list's own states to the shape)
            if (d.owned) shape_zero(d);
        table_[item].append_only = false;
        if (table_[item].by_key.capacity()) {
            table_[item].by_key.clear();
            if (table_[item].by_key.capacity() && spare_maps_.size() < 64) { spare_maps_.emplace_back(); std::swap(spare_maps_.back(), table_[item].by_key); }
        }
        // the consumed vector's storage goes back to the pool when this item is finished
        struct Recycle {
            std::vector<StateVec>& pool; StateVec& v;
            ~Recycle() { if (v.capacity() && pool.size() < 64) { v.clear(); pool.emplace_back(); pool.back().swap(v); } }
        } recycle{spare_items_, ns.items};
        if (item > n_nodes_) {  // join: equal states were merged on arrival; fan out
            const uint32_t lo = fan_first_[item - n_nodes_ - 1], hi = fan_first_[item - n_nodes_];
            // A target fed by this join alone reads the join's list in place when its turn comes
            // (no copy per target); a target with other sources gets its copies now, to merge them.
            uint32_t readers = 0;
            for (uint32_t i = lo; i < hi; ++i) readers += single_source_[fan_[i]];
            for (const State& s : ns.items) {
                if (refs_[s.slot] != kPinned) refs_[s.slot] += hi - lo - 1;
                for (uint32_t i = lo; i < hi; ++i)
                    if (!single_source_[fan_[i]]) arrive(fan_[i], s, out);
            }
            if (!ns.dense.empty()) {
                // One step per receiver for ALL the residue nodes this join feeds — also those with other sources: a block
                // arriving at a residue node merges with nothing there, it only needs the node's residue and receiver, and the
                // receiver is processed after the node in any case.  (Handing such nodes a reference instead made every one of
                // the 20 nodes of a wildcard inside x(m,n) do its own step into the same block, one dependency level each:
                // 165 levels for the bench batch instead of ~60.)  Match / Gap nodes take a reference and deal with it on their turn.
                struct Group { int32_t receiver; uint32_t r_mask; };
                std::vector<Group> groups;
                for (uint32_t i = lo; i < hi; ++i) {
                    const int32_t t = fan_[i];
                    if (g_.takes_residue(t)) {
                        if (dangling_[t]) throw std::runtime_error("k-graph node without successor (the reference fails here too)");
                        if (forward_[t] == KGraph::kNone) continue;
                        const uint32_t bit = code_mask(t);
                        bool found = false;
                        for (Group& g : groups)
                            if (g.receiver == forward_[t]) { g.r_mask |= bit; found = true; }
                        if (!found) groups.push_back(Group{forward_[t], bit});
                    } else {
                        for (DenseRef r : ns.dense) {
                            r.owned = 0;
                            ++block_refs_[r.block];
                            table_[t].dense.push_back(r);
                        }
                    }
                }
                for (const Group& g : groups)
                    for (const DenseRef& r : ns.dense) dense_step(r, g.r_mask, g.receiver, out);
                for (const DenseRef& r : ns.dense) release_block(r.block);
                ns.dense.clear();
            }
            if (readers && !ns.items.empty()) {
                if (ns.items.size() >= merge_sample_threshold() && !merging_pays(ns.items))
                    for (uint32_t i = lo; i < hi; ++i) {  // the readers' receivers will just append
                        const int32_t reader = fan_[i], recv = single_source_[reader] ? forward_[reader] : KGraph::kNone;
                        if (recv != KGraph::kNone && !single_source_[recv] && table_[recv].items.empty()) table_[recv].append_only = true;
                    }
                for (uint32_t i = lo; i < hi; ++i)
                    if (single_source_[fan_[i]]) input_of_[fan_[i]] = item;
                readers_[item - n_nodes_ - 1] = readers;
                waiting_ += (uint64_t)ns.items.size() * readers;
                if (waiting_ > limits_.max_live_states) throw std::runtime_error("query holds too many states at the same time");
                table_[item].items.swap(ns.items);  // stays until the last reader is done
                open_joins_.push_back(item);
            }
            continue;
        }
        // the states to process: this item's own arrivals, or the list of the join that feeds it
        const StateVec* input = &ns.items;
        struct Release {  // the last reader of a join's list recycles it
            QueryExpansion& q; int32_t join;
            ~Release() {
                if (join == KGraph::kNone) return;
                if (--q.readers_[join - q.n_nodes_ - 1] != 0) return;
                StateVec& v = q.table_[join].items;
                if (v.capacity() && q.spare_items_.size() < 64) { v.clear(); q.spare_items_.emplace_back(); q.spare_items_.back().swap(v); }
                else StateVec().swap(v);
                for (size_t i = 0; i < q.open_joins_.size(); ++i)
                    if (q.open_joins_[i] == join) { q.open_joins_[i] = q.open_joins_.back(); q.open_joins_.pop_back(); break; }
            }
        } release{*this, input_of_[item]};
        const int32_t in_join = input_of_[item];
        if (in_join != KGraph::kNone) {
            input = &table_[in_join].items;
            input_of_[item] = KGraph::kNone;
            if (!resuming) {
                waiting_ -= input->size();
                states_ += input->size();
                if (states_ > limits_.max_states) throw std::runtime_error("query expands to too many states");
            }
        }
        const int32_t lab = g_.label[item];
        for (const DenseRef& r : ns.dense) {
            if (lab == KGraph::kMatch) {
                txq_dense_op d{};
                d.kind = TXQ_DENSE_REDUCE;
                d.dst = TXQ_SLOT_RESULT;
                d.src = dense_slot(r.block, 0);
                for (unsigned j = 0; j < dense_pos_; ++j) d.shape[j] = r.shape[j];
                emit_dense(out, d);
            } else if (lab == KGraph::kGap) {
                // gap_procedure without a d-gram index: every state restarts its k-mer, i.e. they all merge into one
                const uint32_t acc = fresh();
                emit(out, TXQ_NO_KMER, acc, TXQ_SLOT_ZERO, TXQ_SLOT_ZERO);
                txq_dense_op d{};
                d.kind = TXQ_DENSE_REDUCE;
                d.dst = acc;
                d.src = dense_slot(r.block, 0);
                for (unsigned j = 0; j < dense_pos_; ++j) d.shape[j] = r.shape[j];
                emit_dense(out, d);
                hand_on(item, State{0, acc, 0, 0, 0, 0, 0}, out);
            } else {
                if (dangling_[item]) throw std::runtime_error("k-graph node without successor (the reference fails here too)");
                dense_step(r, code_mask(item), forward_[item], out);
            }
            release_block(r.block);
        }
        ns.dense.clear();
        if (forward_[item] != KGraph::kNone && !single_source_[forward_[item]] && input->size() > 8) {
            // the merging table of the receiver grows once, not by repeated doubling and re-hashing
            NodeStates& next = table_[forward_[item]];
            if (next.items.capacity() == 0) adopt_storage(next);
            if (next.append_only) next.items.reserve(next.items.size() + input->size());
            else next.by_key.reserve(next.items.size() + input->size());
        }
        if (lab == KGraph::kMatch) {
            for (const State& s : *input) {
                emit(out, TXQ_NO_KMER, TXQ_SLOT_RESULT, s.slot, TXQ_SLOT_RESULT);
                drop(s.slot);
            }
            continue;
        }
        if (lab == KGraph::kGap) {  // gap_procedure: restart the k-mer, or start a d-gram
            const uint64_t gap = g_.gap[item];
            for (State s : *input) {
                if (s.shift < 3 || gap < gaps_.min_gap || gap > gaps_.max_gap) {
                    s.kmer = 0;
                    s.gapped = 0;
                } else {
                    s.kmer = gap * 64000000ULL + ((s.kmer >> 10) & 31) * 3200000ULL + ((s.kmer >> 5) & 31) * 160000ULL + (s.kmer & 31) * 8000ULL;
                    s.gapped = 1;
                }
                s.shift = 0;
                s.res1 = s.res2 = 0;
                hand_on(item, s, out);
            }
            continue;
        }
        constexpr size_t kBatch = 16;
        State pending[kBatch];
        size_t n_pending = 0;
        const int32_t receiver = forward_[item];
        const bool batching = receiver != KGraph::kNone && !dangling_[item] && !single_source_[receiver] && input->size() >= 4 * kBatch;
        auto flush = [&]() {
            const FlatMap& keys = table_[receiver].by_key;
            const unsigned bits = enc_.bits_per_symbol();
            for (size_t i = 0; i < n_pending; ++i) {
                const State& t = pending[i];
                const unsigned phase = t.shift < k - 1 ? t.shift : k - 1;
                keys.prefetch(t.gapped ? (t.kmer | ((uint64_t)(4 + t.shift) << 60)) : ((t.kmer & enc_.suffix_mask()) | (1ULL << (phase * bits))));
            }
            for (size_t i = 0; i < n_pending; ++i) arrive(receiver, pending[i], out);
            n_pending = 0;
        };
        // A fused class (KGraph::kClass) stands for n residue nodes that read one list: residue by residue, as those nodes would
        // in their turn — every state is handed on once per residue, so it takes n - 1 more references first (what the join in
        // front of the n nodes used to give it).  Will the arrivals merge at the receiver?  Asked of the list as a join asks.
        // Like the n nodes, the class can stop BETWEEN two residues when the call's budget is spent (a wildcard behind 20^3
        // states is 160 000 ops per residue: stages would overshoot their budget twenty-fold, and on a sparse index the states
        // the next stage's answers would have pruned are expanded): the list stays where it was and the next call goes on.
        unsigned char one_residue = 0;
        const auto [members, n_residues] = g_.residues(item, &one_residue);
        uint32_t r0 = 0;
        if (resuming) {
            r0 = resume_residue_;
            resume_item_ = KGraph::kNone;
            waiting_ -= (uint64_t)input->size() * (n_residues - r0);
        } else if (n_residues > 1) {
            for (const State& s : *input)
                if (refs_[s.slot] != kPinned) refs_[s.slot] += n_residues - 1;
            states_ += (uint64_t)input->size() * (n_residues - 1);
            if (states_ > limits_.max_states) throw std::runtime_error("query expands to too many states");
            if (receiver != KGraph::kNone && !single_source_[receiver] && table_[receiver].items.empty() &&
                input->size() >= merge_sample_threshold() && !merging_pays(*input))
                table_[receiver].append_only = true;
        }
        for (uint32_t r = r0; r < n_residues; ++r) {
        if (r > r0 && out.size() - start >= op_budget) {  // out of budget: the rest of the residues in the next call
            resume_item_ = item;
            resume_residue_ = r;
            --cursor_;
            waiting_ += (uint64_t)input->size() * (n_residues - r);
            if (in_join != KGraph::kNone) { input_of_[item] = in_join; release.join = KGraph::kNone; }
            else table_[item].items.swap(ns.items);
            break;
        }
        const unsigned char lab = members[r];
        for (State s : *input) {
            if (s.gapped) {  // update_gapped: three residues complete the d-gram
                if (s.shift == 0) { s.kmer += 400ULL * dgram_residue_code(lab); s.res1 = (uint8_t)lab; s.shift = 1; }
                else if (s.shift == 1) { s.kmer += 20ULL * dgram_residue_code(lab); s.res2 = (uint8_t)lab; s.shift = 2; }
                else {
                    if (gaps_.dgram_loaded) {
                        if (!dgrams) throw std::runtime_error("d-gram table missing");
                        const uint32_t id = kDgramFlag | dgrams->intern(s.kmer + dgram_residue_code(lab));
                        if (exclusive(s.slot)) emit(out, id, s.slot, s.slot, TXQ_SLOT_ZERO);
                        else {
                            const uint32_t d = fresh();
                            emit(out, id, d, s.slot, TXQ_SLOT_ZERO);
                            drop(s.slot);
                            s.slot = d;
                        }
                    }
                    s.kmer = 0;
                    enc_.roll(s.res1, s.kmer);
                    enc_.roll(s.res2, s.kmer);
                    enc_.roll((unsigned char)lab, s.kmer);
                    s.shift = 3 < k ? 3 : (uint8_t)k;
                    s.gapped = 0;
                    s.res1 = s.res2 = 0;
                }
                hand_on(item, s, out);
                continue;
            }
            const uint64_t probe = enc_.roll((unsigned char)lab, s.kmer);
            if (s.shift < k - 1) {
                ++s.shift;
            } else {
                const uint32_t id = intern.intern(probe);
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
            // Arrivals of one node all go to the same receiver; with millions of keys (k = 6 peptides)
            // its merge table lives in DRAM, so the table lines of a batch of arrivals are requested
            // before the first of them is merged.
            if (batching) {
                pending[n_pending++] = s;
                if (n_pending == kBatch) flush();
            } else {
                hand_on(item, s, out);
            }
        }
        if (n_pending) flush();
        }
    }
}

void QueryExpansion::frontier_slots(std::vector<uint32_t>& out) {
    if (seen_.size() < refs_.size()) seen_.resize(refs_.size(), 0);
    if (++seen_epoch_ == 0) { std::fill(seen_.begin(), seen_.end(), 0); seen_epoch_ = 1; }
    auto ask = [&](StateVec& items) {
        for (State& s : items) {
            if (s.asked) continue;
            s.asked = 1;
            if (s.slot >= TXQ_SLOT_FIRST_FREE && seen_[s.slot] != seen_epoch_) { seen_[s.slot] = seen_epoch_; out.push_back(s.slot); ++asked_; }
        }
    };
    for (size_t c = cursor_; c < order_.size(); ++c) ask(table_[order_[c]].items);
    for (int32_t j : open_joins_) ask(table_[j].items);  // lists that wait for their readers
}

void QueryExpansion::observe(const std::vector<uint8_t>& klass, uint64_t* bits, uint64_t* states) {
    wants_evidence_ = false;
    const unsigned k = enc_.k();
    auto look = [&](const StateVec& items) {
        for (const State& s : items) {
            if (s.slot >= klass.size() || klass[s.slot] == 0xFF || s.gapped || s.shift < k) continue;  // not asked / never probed
            const uint8_t b = klass[s.slot];
            // b = 1 + floor(log2(bits)): the middle of [2^(b-1), 2^b)
            *bits += b == 0 ? 0 : b == 1 ? 1 : (3ULL << (b - 2));
            ++*states;
        }
    };
    for (size_t c = cursor_; c < order_.size(); ++c) look(table_[order_[c]].items);
    for (int32_t j : open_joins_) look(table_[j].items);
}
void QueryExpansion::prune(const std::vector<uint8_t>& dead) {
    const unsigned k = enc_.k(), bits = enc_.bits_per_symbol();
    auto is_dead = [&](const State& s) { return s.slot < dead.size() && dead[s.slot]; };
    // `holders` = how many readers still hold a reference to every state of the list
    auto sweep = [&](StateVec& items, uint32_t holders) -> bool {
        bool any = false;
        for (const State& s : items)
            if (is_dead(s)) { any = true; break; }
        if (!any) return false;
        StateVec keep;
        for (const State& s : items) {
            if (is_dead(s)) {
                for (uint32_t h = 0; h < holders; ++h) drop(s.slot);
                ++pruned_;
                waiting_ -= holders;
            } else keep.push_back(s);
        }
        items.swap(keep);
        return true;
    };
    // a class that stopped between two residues (resume_item_) holds one reference per state and residue still to come
    const uint32_t class_holds = resume_item_ != KGraph::kNone ? g_.residue_count(resume_item_) - resume_residue_ : 0;
    for (size_t c = cursor_; c < order_.size(); ++c) {
        NodeStates& ns = table_[order_[c]];
        const bool stopped_here = order_[c] == resume_item_;
        if (!sweep(ns.items, stopped_here ? class_holds : 1)) continue;
        ns.by_key.clear();
        if (stopped_here || single_source_[order_[c]] || ns.append_only || ns.items.size() < kSearched) continue;  // (short lists are searched: arrive)
        for (uint32_t i = 0; i < ns.items.size(); ++i) {
            const State& s = ns.items[i];
            const unsigned phase = s.shift < k - 1 ? s.shift : k - 1;
            ns.by_key.emplace(s.gapped ? (s.kmer | ((uint64_t)(4 + s.shift) << 60))
                                       : ((s.kmer & enc_.suffix_mask()) | (1ULL << (phase * bits))), i);
        }
    }
    for (int32_t j : open_joins_) {
        const bool read_by_class = resume_item_ != KGraph::kNone && input_of_[resume_item_] == j;
        sweep(table_[j].items, readers_[j - n_nodes_ - 1] + (read_by_class ? class_holds - 1 : 0));
    }
}

// ---- level scheduling ---------------------------------------------------------------------

std::vector<uint32_t> schedule_levels_into(const OpVec& ops, uint32_t n_slots, LevelScratch& sc, txq_op* dst,
                                           uint32_t kmer_add, uint32_t dgram_add, const DenseSchedule* dn) {
    std::vector<uint32_t> ends;
    if (ops.empty()) return ends;
    if (sc.slot.size() < (size_t)n_slots) sc.slot.resize((size_t)n_slots, LevelScratch::Slot{0, 0, 0, 0});
    if (dn && sc.block.size() < TXQ_DENSE_MAX_BLOCKS) sc.block.resize(TXQ_DENSE_MAX_BLOCKS, LevelScratch::Block{0, 0, 0, 0, 0, 0});
    if (++sc.epoch == 0) {
        for (auto& s : sc.slot) s.stamp = 0;
        for (auto& b : sc.block) b.stamp = 0;
        sc.epoch = 1;
    }
    auto touch = [&](uint32_t s) -> LevelScratch::Slot& {  // ordinary slots only
        LevelScratch::Slot& x = sc.slot[s];
        if (x.stamp != sc.epoch) x = LevelScratch::Slot{sc.epoch, 0, 0, 0};
        return x;
    };
    // hazards between dense ops (whole blocks) and ordinary ops on single slots of a block are kept per block:
    // dw / dr = last level a dense op wrote / read the block, sw / sr = last level an ordinary op wrote / read a slot of it
    auto block_of = [&](uint32_t s) -> LevelScratch::Block* {
        if (!(s & TXQ_DENSE_SLOT_BIT)) return nullptr;
        if (!dn) throw std::logic_error("dense slot without a dense table");
        LevelScratch::Block& b = sc.block[(s & ~TXQ_DENSE_SLOT_BIT) >> TXQ_DENSE_BLOCK_SHIFT];
        if (b.stamp != sc.epoch) b = LevelScratch::Block{sc.epoch, 0, 0, 0, 0, 0};
        return &b;
    };
    // level of op = smallest level that respects every hazard against earlier ops (levels from 1)
    sc.level_of.resize(ops.size());
    uint32_t top = 0;
    for (size_t i = 0; i < ops.size(); ++i) {
        const txq_op& o = ops[i];
        uint32_t lvl = 1;
        auto after = [&](uint32_t l) { if (l + 1 > lvl) lvl = l + 1; };
        if (o.kmer == TXQ_DENSE_OP) {
            if (!dn) throw std::logic_error("dense op without a dense table");
            const txq_dense_op& d = dn->table[o.dst];
            if (d.kind == TXQ_DENSE_REDUCE) {
                LevelScratch::Block& x = *block_of(d.src);
                LevelScratch::Slot& sd = touch(d.dst);
                LevelScratch::Block* bd = block_of(d.dst);
                after(x.dw); after(x.sw);
                after(sd.wr); after(sd.rd);
                if (bd) { after(bd->dw); after(bd->dr); }
                if (sd.acc > lvl) lvl = sd.acc;  // an accumulation: shares its level with others into the same slot
                if (x.dr < lvl) x.dr = lvl;
                sd.acc = lvl;
                if (bd && bd->sw < lvl) bd->sw = lvl;
            } else {
                LevelScratch::Block& y = *block_of(d.dst);
                after(y.dw); after(y.dr); after(y.sw); after(y.sr);
                if (d.kind == TXQ_DENSE_STEP) {
                    LevelScratch::Block& x = *block_of(d.src);
                    after(x.dw); after(x.sw);
                    if (x.dr < lvl) x.dr = lvl;
                } else if (d.kind == TXQ_DENSE_FILL) {  // reads one ordinary slot
                    LevelScratch::Slot& x = touch(d.src);
                    after(x.wr); after(x.acc);
                    if (x.rd < lvl) x.rd = lvl;
                }
                y.dw = lvl;
            }
            sc.level_of[i] = lvl;
            if (lvl > top) top = lvl;
            continue;
        }
        // an operand is an ordinary slot (its own record) or a slot of a block (the block's record)
        struct Ref { LevelScratch::Slot* s; LevelScratch::Block* b; };
        auto ref = [&](uint32_t slot) -> Ref {
            if (slot & TXQ_DENSE_SLOT_BIT) return Ref{nullptr, block_of(slot)};
            return Ref{&touch(slot), nullptr};
        };
        auto after_writes_of = [&](const Ref& r) {  // RAW: full writes, accumulations and dense writes
            if (r.s) { after(r.s->wr); after(r.s->acc); }
            else { after(r.b->dw); after(r.b->sw); }
        };
        auto mark_read = [&](const Ref& r) {
            if (r.s) { if (r.s->rd < lvl) r.s->rd = lvl; }
            else if (r.b->sr < lvl) r.b->sr = lvl;
        };
        const Ref rd = ref(o.dst), ra = ref(o.a), rb = ref(o.b);
        LevelScratch::Block* bd = rd.b;
        const bool accumulate = o.kmer == TXQ_NO_KMER && (o.dst == o.a || o.dst == o.b);
        if (accumulate) {
            const Ref& src = o.dst == o.a ? rb : ra;
            after_writes_of(src);
            if (rd.s) {
                after(rd.s->wr); after(rd.s->rd);        // after the last full write and every earlier reader
                if (rd.s->acc > lvl) lvl = rd.s->acc;    // may share a level with other accumulations
            } else {                                     // ... of a block: dense writes / reads, full writes and readers of its slots,
                after(bd->dw); after(bd->dr); after(bd->fw); after(bd->sr);  // but not the other accumulations into it (sw)
            }
            mark_read(src);
            if (rd.s) rd.s->acc = lvl;
        } else {
            after_writes_of(ra);
            after_writes_of(rb);
            // WAR / WAW on dst; an in-place op (dst == a or b) reads its own old value, which is fine
            if (rd.s) { after(rd.s->wr); after(rd.s->acc); after(rd.s->rd); }
            else { after(bd->dw); after(bd->dr); after(bd->sw); after(bd->sr); }
            mark_read(ra);
            mark_read(rb);
            if (rd.s) rd.s->wr = lvl;
            else if (bd->fw < lvl) bd->fw = lvl;
        }
        if (bd && bd->sw < lvl) bd->sw = lvl;
        sc.level_of[i] = lvl;
        if (lvl > top) top = lvl;
    }
    // stable counting sort by level, straight into dst
    ends.assign(top, 0);
    for (uint32_t l : sc.level_of) ++ends[l - 1];
    sc.pos.assign(top, 0);
    for (uint32_t l = 1; l < top; ++l) sc.pos[l] = sc.pos[l - 1] + ends[l - 1];
    const uint32_t dense_add = dn ? dn->index_add : 0;
    for (size_t i = 0; i < ops.size(); ++i) {
        txq_op o = ops[i];
        if (o.kmer == TXQ_DENSE_OP) o.dst += dense_add;
        else if (o.kmer != TXQ_NO_KMER) o.kmer = (o.kmer & kDgramFlag) ? (o.kmer & ~kDgramFlag) + dgram_add : o.kmer + kmer_add;
        dst[sc.pos[sc.level_of[i] - 1]++] = o;
    }
    for (uint32_t l = 1; l < top; ++l) ends[l] += ends[l - 1];
    return ends;
}

std::vector<uint32_t> schedule_levels(OpVec& ops, uint32_t n_slots, LevelScratch& sc) {
    sc.sorted.resize(ops.size());
    // the one-shot path has no d-gram ops: indexes pass through unchanged
    std::vector<uint32_t> ends = schedule_levels_into(ops, n_slots, sc, sc.sorted.data(), 0, kDgramFlag);
    ops.swap(sc.sorted);
    return ends;
}

// ---- one-shot batches ----------------------------------------------------------------------

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
    x.advance(SIZE_MAX, table_, prog.ops);
    prog.n_slots = x.n_slots();
    prog.states = x.states();
    prog.probes = x.probes();
    programs_.push_back(std::move(prog));
    return programs_.size() - 1;
}

std::vector<uint8_t> ProgramBatch::serialise() const {
    std::vector<txq_program> pr(programs_.size());
    std::vector<OpVec> ops(programs_.size());
    std::vector<std::vector<uint32_t>> levels(programs_.size());
    std::vector<const OpVec*> ops_of(programs_.size());
    std::vector<const std::vector<uint32_t>*> levels_of(programs_.size());
    LevelScratch scratch;
    uint32_t first = 0;
    for (size_t i = 0; i < programs_.size(); ++i) {
        ops[i] = programs_[i].ops;
        levels[i] = schedule_levels(ops[i], programs_[i].n_slots, scratch);
        pr[i] = txq_program{first, (uint32_t)ops[i].size(), programs_[i].n_slots, 0};
        first += (uint32_t)ops[i].size();
        ops_of[i] = &ops[i];
        levels_of[i] = &levels[i];
    }
    return make_blob({&table_.values()}, pr, ops_of, levels_of);
}

}  // namespace tetrex

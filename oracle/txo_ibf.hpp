// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
// CPU oracle for the TetRex query hot path: (H)IBF arithmetic.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything under oracle/.  The product path (tetrex_amd/) never links it.
//
// The arithmetic restated here lives in the third-party library seqan/hibf
// (git submodule lib/hibf of the reference, .gitmodules:18-21, branch master,
// pinned SHA unknown, sources ABSENT from /root/reference).  It is restated
// from hibf's published algorithm and anchored on the reference's call sites:
//   include/index_ibf.h:35-37,92,96-97   ctor + emplace
//   include/index_ibf.h:143,148          containment_agent / bulk_contains
//   include/index_hibf.h:145,151         membership_agent / membership_for
// Pinned by the reference's own fixture test/data/ibf_idx.ibf (hash seeds 0-2,
// fastrange, row-major 64-bin interleave) — see tests/test_oracle_fixture.py.
// Unpinned (no fixture in the reference): seeds 3-4, HIBF descent semantics.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <algorithm>
#include <stdexcept>

namespace txo {

// hibf: interleaved_bloom_filter::hash_seeds
static constexpr uint64_t kSeeds[5] = {
    13572355802537770549ULL, 13043817825332782213ULL, 10650232656628343401ULL,
    16499269484942379435ULL, 4893150838803335377ULL};
static constexpr uint64_t kGolden = 11400714819323198485ULL;  // 2^64 / phi

inline unsigned clz64(uint64_t v) { return v ? (unsigned)__builtin_clzll(v) : 64u; }

struct Ibf {
    uint64_t bins = 0;        // user-visible bin count B
    uint64_t tech_bins = 0;   // 64 * ceil(B/64)
    uint64_t bin_size = 0;    // rows m
    uint64_t hash_shift = 0;  // countl_zero(bin_size)
    uint64_t bin_words = 0;   // W = tech_bins / 64
    uint64_t hash_funs = 0;   // h, 1..5
    std::vector<uint64_t> data;  // row-major [bin_size][bin_words]

    Ibf() = default;
    Ibf(uint64_t b, uint64_t m, uint64_t h) { init(b, m, h); }
    void init(uint64_t b, uint64_t m, uint64_t h) {
        if (b == 0 || m == 0 || h == 0 || h > 5) throw std::invalid_argument("bad IBF shape");
        bins = b;
        bin_words = (b + 63) / 64;
        tech_bins = bin_words * 64;
        bin_size = m;
        hash_shift = clz64(m);
        hash_funs = h;
        data.assign(m * bin_words, 0);
    }

    // hibf: interleaved_bloom_filter::hash_and_fit — returns the BIT offset of the row start.
    uint64_t hash_and_fit(uint64_t v, uint64_t seed) const {
        v *= seed;
        v ^= v >> hash_shift;
        v *= kGolden;
        v = (uint64_t)(((unsigned __int128)v * (unsigned __int128)bin_size) >> 64);
        return v * tech_bins;
    }
    uint64_t row_of(uint64_t v, unsigned i) const { return hash_and_fit(v, kSeeds[i]) / tech_bins; }

    // hibf: interleaved_bloom_filter::emplace(value, bin_index)
    void emplace(uint64_t v, uint64_t bin) {
        for (unsigned i = 0; i < hash_funs; ++i) {
            uint64_t bit = hash_and_fit(v, kSeeds[i]) + bin;
            data[bit >> 6] |= 1ULL << (bit & 63);
        }
    }

    // hibf: containment_agent_type::bulk_contains(value) -> bit_vector of `bins` bits
    // (bin_words words; bits >= bins are zero because they are never set).
    void bulk_contains(uint64_t v, uint64_t* out) const {
        uint64_t off[5];
        for (unsigned i = 0; i < hash_funs; ++i) off[i] = hash_and_fit(v, kSeeds[i]) >> 6;
        for (uint64_t w = 0; w < bin_words; ++w) {
            uint64_t acc = data[off[0] + w];
            for (unsigned i = 1; i < hash_funs; ++i) acc &= data[off[i] + w];
            out[w] = acc;
        }
    }
};

// hibf: hierarchical_interleaved_bloom_filter (ibf_vector, next_ibf_id,
// ibf_bin_to_user_bin_id with merged bins marked by a negative id / ~0).
struct Hibf {
    static constexpr uint64_t kMerged = ~0ULL;
    uint64_t user_bins = 0;
    std::vector<Ibf> ibf;                            // ibf[0] = root
    std::vector<std::vector<uint64_t>> next_ibf_id;  // [ibf][tb]
    std::vector<std::vector<uint64_t>> tb_to_user;   // [ibf][tb], kMerged for merged bins

    // hibf: membership_agent_type::membership_for_impl restated for ONE value and
    // threshold 1 (how TetRex calls it, include/index_hibf.h:142-147): walk the
    // technical bins left to right, summing counts over a run mapped to the same
    // user bin; a merged bin with a hit recurses; a run with a hit emits its id.
    void descend(uint64_t v, uint64_t idx, std::vector<uint64_t>& tmp, std::vector<uint64_t>& ids) const {
        const Ibf& f = ibf[idx];
        std::vector<uint64_t> hit(f.bin_words);
        f.bulk_contains(v, hit.data());
        unsigned sum = 0;
        const auto& map = tb_to_user[idx];
        for (uint64_t b = 0; b < f.bins; ++b) {
            sum += (unsigned)((hit[b >> 6] >> (b & 63)) & 1);
            uint64_t ub = map[b];
            if (ub == kMerged) {
                if (sum >= 1) descend(v, next_ibf_id[idx][b], tmp, ids);
                sum = 0;
            } else if (b + 1 == f.bins || ub != map[b + 1]) {
                if (sum >= 1) ids.push_back(ub);
                sum = 0;
            }
        }
    }
    // membership_for({v}, 1): sorted user-bin ids
    std::vector<uint64_t> membership_for(uint64_t v) const {
        std::vector<uint64_t> tmp, ids;
        descend(v, 0, tmp, ids);
        std::sort(ids.begin(), ids.end());
        return ids;
    }
    // HIBFIndex::query + populate_bitvector (include/index_hibf.h:132-147): fresh
    // user_bins-bit vector with one bit per returned id.
    void query(uint64_t v, uint64_t* out) const {
        uint64_t W = (user_bins + 63) / 64;
        for (uint64_t w = 0; w < W; ++w) out[w] = 0;
        for (uint64_t id : membership_for(v)) out[id >> 6] |= 1ULL << (id & 63);
    }
};

// IBFIndex::compute_bitcount (include/index_ibf.h:133-139): m = ceil(-n ln p / ln^2 2),
// p is a float in the reference (promoted to double inside std::log).
uint64_t compute_bitcount(uint64_t n, float fpr);

}  // namespace txo

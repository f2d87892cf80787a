// HIBF on gfx950: level-synchronous descent replacing
// seqan::hibf membership_agent::membership_for({kmer}, 1) + HIBFIndex::populate_bitvector
// (reference include/index_hibf.h:132-147).
//
// The reference recurses per k-mer, allocating a counting agent per visited IBF and a fresh
// user_bins-bit vector per k-mer.  Here the whole batch descends one tree level per kernel:
// a frontier of (k-mer, IBF) work items is expanded into the next level's frontier through one
// wave-aggregated atomic counter, and user-bin hits are OR-ed straight into the batch's mask
// matrix.  For one k-mer and threshold 1 "sum of counts over a run of technical bins >= 1" is
// "any bit of the run is set", so every set technical bin can be handled independently.
// No host round trip between levels: each level kernel reads its item count from HBM.
#include "txq_internal.hpp"
#include <cstdlib>
#include <cstring>
#include <deque>
#include <utility>

namespace txq {

struct HibfView {
    const IbfDev* ibf;
    const uint64_t* next;
    const uint64_t* tb_user;
    const uint64_t* map_off;
    const uint64_t* merged;      // per IBF word: bit b set <=> technical bin 64w+b is a merged bin
    const uint64_t* descend;     // same layout: the merged bins whose sub-tree holds user bins of THIS shard's mask columns
    const uint64_t* merged_off;  // [n_ibf] offset of IBF i's words in `merged` / `descend`
    // fused kernel: everything about the child IBF behind merged technical bin e in ONE 32-byte
    // record, nodes[e] (no next_ibf_id -> descriptor chain); nodes[total technical bins] is the root
    const struct HibfNode* nodes;
    uint32_t root_entry;
    // LAYOUT rows of trees with split user bins (txq_internal.hpp VSplit): per row word the bits that are not their bin's
    // representative, and for such a bit the representative's position in the row (null: no split bins)
    const uint64_t* nonrep = nullptr;
    const uint32_t* rep_pos = nullptr;
};


// loads through a pointer that was itself read from memory: tell the compiler it is global memory
// (otherwise it emits flat loads, which also wait on the LDS counter)
__device__ __forceinline__ uint64_t gload(const uint64_t* p) {
    return *(const __attribute__((address_space(1))) uint64_t*)p;
}
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ulonglong2 gload2(const uint64_t* p) {  // p is 16-byte aligned
    const u64x2 q = *(const __attribute__((address_space(1))) u64x2*)p;
    return ulonglong2{q.x, q.y};
}

// exclusive prefix sum of `v` over the 64 lanes of a wave; *total receives the wave sum
__device__ __forceinline__ uint32_t wave_exclusive_scan(uint32_t v, uint32_t* total) {
    const int lane = threadIdx.x & 63;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    *total = __shfl(incl, 63);
    return incl - v;
}

// G lanes per work item, one 64-bit word of the IBF row per lane (G = pow2 >= widest row, <= 64;
// rows wider than 64 words take `w_iters` sweeps).  Children are appended to the next frontier
// with ONE atomicAdd per wave (prefix sum of the per-lane child counts).
template <int G>
__global__ __launch_bounds__(256) void hibf_level_kernel(HibfView t, const uint64_t* __restrict__ kmers,
                                                         const WorkItem* __restrict__ in, const uint32_t* __restrict__ in_count,
                                                         uint32_t n_level0, WorkItem* __restrict__ out, uint32_t* __restrict__ out_count,
                                                         uint32_t out_cap, uint32_t* __restrict__ overflow,
                                                         uint64_t* __restrict__ masks, uint32_t w_out, uint32_t word0, uint32_t w_iters) {
    const uint32_t count = in ? *in_count : n_level0;
    const uint32_t sub = threadIdx.x % G;
    const size_t group = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const size_t n_groups = ((size_t)gridDim.x * blockDim.x) / G;
    const size_t rounds = ((size_t)count + n_groups - 1) / n_groups;  // uniform over the grid
    for (size_t r = 0; r < rounds; ++r) {
        const size_t item = r * n_groups + group;
        const bool live = item < count;
        uint32_t kidx = 0, id = 0;
        if (live) {
            kidx = in ? in[item].kmer : (uint32_t)item;
            id = in ? in[item].ibf : 0u;
        }
        const IbfDev f = t.ibf[id];
        const uint64_t off = t.map_off[id];
        const uint64_t moff = t.merged_off[id];
        const uint64_t v = live ? kmers[kidx] : 0;
        uint64_t row[5];
#pragma unroll
        for (uint32_t j = 0; j < 5; ++j) row[j] = (j == 0 || j < f.hash_funs) ? hash_row(v, kSeeds[j], f.hash_shift, f.bin_size) : row[j ? j - 1 : 0];
        for (uint32_t wi = 0; wi < w_iters; ++wi) {
            const uint32_t w = wi * G + sub;
            uint64_t acc = 0;
            if (live && w < f.shard_words) {
                // all loads issue together: a hash function the IBF does not have repeats the last real row
                acc = ~0ULL;
#pragma unroll
                for (uint32_t j = 0; j < 5; ++j) acc &= gload(f.words + row[j] * f.stride + w);
            }
            uint64_t kids = acc ? (acc & t.descend[moff + w]) : 0;
            uint64_t hits = acc ? (acc & ~t.merged[moff + w]) : 0;
            // children -> next frontier, one atomic per wave
            uint32_t total;
            const uint32_t mine = (uint32_t)__builtin_popcountll(kids);
            uint32_t at = wave_exclusive_scan(mine, &total);
            if (total) {
                uint32_t base = 0;
                if ((threadIdx.x & 63) == 63) base = atomicAdd(out_count, total);
                base = __shfl(base, 63);
                at += base;
                while (kids) {
                    const uint32_t tb = w * 64u + (uint32_t)__builtin_ctzll(kids);
                    kids &= kids - 1;
                    if (at < out_cap) out[at] = WorkItem{kidx, (uint32_t)t.next[off + tb]};
                    else *overflow = 1u;
                    ++at;
                }
            }
            // user bins -> result mask
            if (f.ident_word != kNoIdent) {  // the row word is a mask word
                const uint64_t word = (uint64_t)f.ident_word + w;
                if (hits && word >= word0 && word < (uint64_t)word0 + w_out)
                    atomicOr((unsigned long long*)(masks + (size_t)kidx * w_out + (word - word0)), hits);
                hits = 0;
            }
            while (hits) {
                const uint32_t tb = w * 64u + (uint32_t)__builtin_ctzll(hits);
                hits &= hits - 1;
                if (tb >= f.bins) break;  // never set; guards the map look-up
                const uint64_t ub = t.tb_user[off + tb];
                const uint64_t word = ub >> 6;
                if (word >= word0 && word < (uint64_t)word0 + w_out)
                    atomicOr((unsigned long long*)(masks + (size_t)kidx * w_out + (word - word0)), 1ULL << (ub & 63));
            }
        }
    }
}

constexpr uint32_t kRootEntry = 0xFFFFFFFFu;  // stack entry of the root IBF (every other entry is a technical-bin index)

// value of `v` in lane `src` (wave-uniform src)
__device__ __forceinline__ uint64_t read_lane(uint64_t v, int src) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

// orders this wave's LDS traffic: everything before is visible to every lane after
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Whole descent of ONE k-mer per wave, no frontier in HBM: the wave keeps the k-mer's result row
// (w_out words) and a stack of IBF ids in LDS, pops up to 64/G IBFs per round (G lanes each, FOUR
// row words = two 16-byte loads per lane and hash function), pushes the children found and ORs
// user-bin hits into the LDS row; the finished row is written once, coalesced.  HBM traffic per
// k-mer is the row itself (w_out * 8 B) instead of a zero fill plus one read-modify-write per hit
// word; the tree (tens of MB) is served from L2 / Infinity Cache, which makes the kernel
// instruction-bound: hence few lanes per IBF (a 256-bin IBF is one lane) and only as many hash
// evaluations as the tree's IBFs have (h_max, wave-uniform).  A k-mer visits every IBF at most
// once (the IBFs form a tree), so a stack of n_ibf entries cannot overflow.
// LDS per wave: w_out * 8 + stack_lds * 4 bytes (dynamic); launch: 64 * waves threads per block.
// Only the first stack_lds entries of the stack live in LDS (a k-mer of a real tree has a few dozen IBFs pending, the bound is
// EVERY IBF of the tree): entries beyond them go into the k-mer's own output row in HBM, which is this wave's alone until it
// writes the finished row over it (the host checks that the row has the room: stack_cap - stack_lds <= 2 * w_out entries) —
// the LDS the never-used tail of the stack took is worth one to four more resident waves per CU.
// LAYOUT: the row is the tree's layout-order row (txq_internal.hpp VChunk) — t.nodes are then the records whose ident_word is
// the IBF's first word IN THAT ROW (Index::d_vnodes), t.descend = t.merged, and an IBF's ANDed row words go into its segment
// as they are, merged bins' bits included (one lane owns a word: plain LDS stores; no technical-bin -> user-bin mapping).
template <int G, bool LAYOUT = false>
__global__ __launch_bounds__(256) void hibf_fused_kernel(HibfView t, const uint64_t* __restrict__ kmers, size_t n,
                                                         uint64_t* __restrict__ masks, uint32_t w_out, uint32_t word0,
                                                         uint32_t w_iters, uint32_t stack_cap, uint32_t wave_words,
                                                         uint32_t h_max, uint64_t* __restrict__ alive, uint32_t stack_lds, uint32_t row_lds_words) {
    extern __shared__ uint64_t lds[];
    const uint32_t lane = threadIdx.x & 63;
    // LAYOUT with row_lds_words == 0: no row in LDS at all — every word of a layout-order row has ONE writer (the lane that ANDs
    // it), so the wave clears the k-mer's row in HBM and the lanes store their non-zero words straight into it (the bits of
    // split user bins move with global atomics); the LDS then only holds the stacks and the waves are bounded by registers
    const bool direct = LAYOUT && row_lds_words == 0;
    uint64_t* row = lds + (size_t)(threadIdx.x >> 6) * wave_words;
    uint32_t* stack = reinterpret_cast<uint32_t*>(row + row_lds_words);
    const uint32_t sub = lane % G, group = lane / G;
    constexpr uint32_t kPerRound = 64 / G;
    const size_t waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t first = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint64_t v_next = first < n ? kmers[first] : 0;
    for (size_t i = first; i < n; i += waves) {
        uint64_t* const out = masks + i * (size_t)w_out;
        const bool wide = (w_out & 1u) == 0 && (reinterpret_cast<uintptr_t>(masks) & 15u) == 0;  // (16-byte aligned rows: a lane stores two words at once)
        if (direct) {
            if (wide) {
                for (uint32_t j = lane * 2; j < w_out; j += 128) *reinterpret_cast<u64x2*>(out + j) = u64x2{0, 0};
            } else {
                for (uint32_t j = lane; j < w_out; j += 64) out[j] = 0;
            }
        } else {
            for (uint32_t j = lane; j < w_out; j += 64) row[j] = 0;
        }
        if (lane == 0) stack[0] = t.root_entry;
        uint32_t* spill = reinterpret_cast<uint32_t*>(out);  // stack entries stack_lds.. (see above; none when the row is written directly)
        // the k-mer is the same in every lane: make that visible, so that the seed products are
        // computed once per k-mer on the scalar unit instead of per IBF on the vector unit
        const uint64_t v = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v_next >> 32)) << 32) |
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v_next);
        if (i + waves < n) v_next = kmers[i + waves];  // lands while this k-mer descends
        uint64_t seeded[5];
#pragma unroll
        for (uint32_t j = 0; j < 5; ++j) seeded[j] = v * kSeeds[j];
        uint32_t count = 1;  // wave-uniform
        while (count) {
            wave_sync();
            const uint32_t take = count < kPerRound ? count : kPerRound;
            count -= take;
            const bool live = group < take;
            // a stack entry is a technical-bin index; everything about the IBF behind it is one record
            const uint32_t at = count + group;
            const HibfNode nd = t.nodes[live ? (at < stack_lds ? stack[at] : spill[at - stack_lds]) : t.root_entry];
            const uint32_t stride = nd.stride(), words_per_row = nd.words_per_row();
            uint64_t r[5];
#pragma unroll
            for (uint32_t j = 0; j < 5; ++j) {
                if (j >= h_max) { r[j] = 0; continue; }  // uniform: no IBF of this tree has more
                // a hash function this IBF does not have repeats its last real row (AND is idempotent)
                r[j] = (j == 0 || j < nd.hash_funs()) ? hash_row_seeded32(seeded[j], nd.hash_shift(), nd.bin_size) : r[j ? j - 1 : 0];
            }
            // most rounds only meet leaves: then nothing is loaded or done for merged bins
            const bool any_merged = __ballot(live && nd.has_merged()) != 0;
            uint64_t moved[4] = {0, 0, 0, 0};  // LAYOUT, split user bins: bits of this lane's words that belong at their bin's representative
            for (uint32_t wi = 0; wi < w_iters; ++wi) {
                const uint32_t w0 = (wi * G + sub) * 4u;
                uint64_t acc[4] = {0, 0, 0, 0}, mg[4] = {0, 0, 0, 0}, dn[4] = {0, 0, 0, 0};
                if constexpr (LAYOUT) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) moved[q] = 0;
                }
                if (live && w0 < words_per_row) {  // every load of the round issues before the first is used
                    const uint64_t* words = (const uint64_t*)nd.words;
                    if constexpr (LAYOUT) {
                        if (t.nonrep)  // (uniform) split user bins: which of these words' bits are not their bin's representative
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (w0 + (uint32_t)q < words_per_row) moved[q] = gload(t.nonrep + nd.ident_word + w0 + (uint32_t)q);
                    }
                    if (stride == 1) {
                        uint64_t x = ~0ULL;
#pragma unroll
                        for (uint32_t j = 0; j < 5; ++j)
                            if (j < h_max) x &= gload(words + r[j]);
                        acc[0] = x;
                    } else {  // rows are padded to an even number of words, the padding is zero
                        const bool upper = w0 + 2 < stride;
                        ulonglong2 lo{~0ULL, ~0ULL}, hi{~0ULL, ~0ULL};
#pragma unroll
                        for (uint32_t j = 0; j < 5; ++j) {
                            if (j >= h_max) continue;
                            const uint64_t* p = words + r[j] * stride + w0;
                            const ulonglong2 a = gload2(p);
                            lo.x &= a.x; lo.y &= a.y;
                            if (upper) { const ulonglong2 b = gload2(p + 2); hi.x &= b.x; hi.y &= b.y; }
                        }
                        acc[0] = lo.x; acc[1] = lo.y;
                        if (upper) { acc[2] = hi.x; acc[3] = hi.y; }
                    }
                    if (nd.has_merged()) {  // merged-bin masks are padded to 4 words per IBF
                        const uint64_t moff = nd.moff;
                        const ulonglong2 m0 = gload2(t.merged + moff + w0), m1 = gload2(t.merged + moff + w0 + 2);
                        const ulonglong2 d0 = gload2(t.descend + moff + w0), d1 = gload2(t.descend + moff + w0 + 2);
                        mg[0] = m0.x; mg[1] = m0.y; mg[2] = m1.x; mg[3] = m1.y;
                        dn[0] = d0.x; dn[1] = d0.y; dn[2] = d1.x; dn[3] = d1.y;
                    }
                }
                // Children and mapped user bins are expanded by the whole wave, one 64-bit word per step:
                // lane L takes bit L, so a word costs the same whether 1 or 64 of its bits are set.
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t first_tb = nd.off + (w0 + (uint32_t)q) * 64u;
                    if (any_merged) {
                        const uint64_t kids = acc[q] & dn[q];  // children outside this shard's columns are not visited
                        uint64_t owners = __ballot(kids != 0);
                        while (owners) {  // wave-uniform
                            const int src = __builtin_ctzll(owners);
                            owners &= owners - 1;
                            const uint64_t word = read_lane(kids, src);
                            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)first_tb, src);
                            if ((word >> lane) & 1) {
                                const uint32_t pos = count + (uint32_t)__builtin_popcountll(word & ((1ULL << lane) - 1));
                                if (pos < stack_lds) stack[pos] = base + lane;
                                else if (pos < stack_cap) spill[pos - stack_lds] = base + lane;
                            }
                            count += (uint32_t)__builtin_popcountll(word);
                            if (count > stack_cap) count = stack_cap;  // unreachable for a tree; keeps the indexes in range
                        }
                    }
                    const uint32_t w = w0 + (uint32_t)q;
                    if constexpr (LAYOUT) {
                        moved[q] &= acc[q];  // (the bits that move to their representative once the round's words are in the row)
                        if (live && w < words_per_row) {
                            const uint64_t mine = acc[q] & ~moved[q];
                            if (!direct) row[nd.ident_word + w] = mine;
                            else if (mine) out[nd.ident_word + w] = mine;
                        }
                        continue;
                    }
                    uint64_t hits = acc[q] & ~mg[q];
                    if (hits && nd.ident_word != kNoIdent) {  // the row word is a mask word
                        const uint64_t word = (uint64_t)nd.ident_word + w;
                        if (word >= word0 && word < (uint64_t)word0 + w_out) atomicOr((unsigned long long*)(row + (word - word0)), hits);
                        hits = 0;
                    }
                    // padding bits are never set; this guards the map look-up
                    if (w * 64u + 63u >= nd.bins) hits &= nd.bins > w * 64u ? (~0ULL >> (63u - ((nd.bins - 1u) & 63u))) : 0;
                    uint64_t owners = __ballot(hits != 0);
                    while (owners) {
                        const int src = __builtin_ctzll(owners);
                        owners &= owners - 1;
                        const uint64_t word = read_lane(hits, src);
                        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)first_tb, src);
                        if ((word >> lane) & 1) {
                            const uint64_t ub = gload(t.tb_user + base + lane);
                            const uint64_t mw = ub >> 6;
                            if (mw >= word0 && mw < (uint64_t)word0 + w_out) atomicOr((unsigned long long*)(row + (mw - word0)), 1ULL << (ub & 63));
                        }
                    }
                }
            }
            if constexpr (LAYOUT) {
                // Split user bins (txq_internal.hpp VSplit): a bin is its representative — the bits of its other parts were held
                // back from the row above (their mask came with the rows' loads) and go to the representative now that all of the
                // round's words are in the row (plain stores, every IBF by its own lanes): LDS atomics for the rare word that has any.
                // (w_iters == 1 wherever this kernel runs in layout order: an IBF of more than 256 * G technical bins would need
                // a second pass here.)
                if (t.nonrep) {  // (uniform)
                    wave_sync();
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t j = nd.ident_word + sub * 4u + (uint32_t)q;
                        for (uint64_t m = moved[q]; m; m &= m - 1) {
                            const uint32_t to = t.rep_pos[(size_t)j * 64 + (uint32_t)__builtin_ctzll(m)];
                            if (direct) atomicOr((unsigned long long*)(out + (to >> 6)), 1ULL << (to & 63));
                            else atomicOr((unsigned long long*)(row + (to >> 6)), 1ULL << (to & 63));
                        }
                    }
                }
            }
        }
        wave_sync();
        uint64_t any = 0;
        if (!direct) {
            if (wide) {
                for (uint32_t j = lane * 2; j < w_out; j += 128) {
                    const u64x2 x{row[j], row[j + 1]};
                    any |= x.x | x.y;
                    __builtin_nontemporal_store(x, reinterpret_cast<u64x2*>(out + j));
                }
            } else {
                for (uint32_t j = lane; j < w_out; j += 64) {
                    const uint64_t x = row[j];
                    any |= x;
                    __builtin_nontemporal_store(x, out + j);
                }
            }
        }
        if (alive) {
            const uint64_t some = __ballot(any != 0);
            if (lane == 0 && some) atomicOr((unsigned long long*)(alive + (i >> 6)), 1ULL << (i & 63));
        }
        wave_sync();  // the next k-mer reuses the row
    }
}

template <int G, bool LAYOUT = false>
static hipError_t launch_fused(unsigned grid, unsigned threads, size_t lds_bytes, hipStream_t s, HibfView t, const uint64_t* kmers, size_t n,
                               uint64_t* masks, uint32_t w_out, uint32_t word0, uint32_t w_iters, uint32_t stack_cap, uint32_t wave_words,
                               uint32_t h_max, uint64_t* alive, uint32_t stack_lds, uint32_t row_lds_words) {
    if (lds_bytes > (48u << 10)) {  // (more dynamic LDS than the default limit: ask for it)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hibf_fused_kernel<G, LAYOUT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hibf_fused_kernel<G, LAYOUT><<<grid, threads, lds_bytes, s>>>(t, kmers, n, masks, w_out, word0, w_iters, stack_cap, wave_words, h_max, alive, stack_lds, row_lds_words);
    return hipGetLastError();
}

// Small trees (every IBF at most 256 technical bins, at most 32 IBFs, at most 16 mask words — the
// Swissprot-HIBF shape): ONE LANE per k-mer instead of one wave.  A lane keeps its k-mer's row and its
// stack of technical-bin indexes in LDS columns ([word][lane], [entry][lane]: conflict-free), visits its IBFs
// one after the other (a k-mer of such a tree visits two or three) and the workgroup writes its 256 rows
// out together, coalesced.  The tree (about a MB) is L2-resident; 64 independent descents per wave
// supply the memory-level parallelism that the wave-per-k-mer kernel gets from 64 IBFs per round.
constexpr uint32_t kSmallThreads = 256, kSmallStack = 32, kSmallPitch = kSmallThreads + 1;

__global__ __launch_bounds__(256) void hibf_small_kernel(HibfView t, const uint64_t* __restrict__ kmers, size_t n,
                                                         uint64_t* __restrict__ masks, uint32_t w_out, uint32_t word0,
                                                         uint32_t h_max, uint64_t* __restrict__ alive) {
    extern __shared__ uint64_t lds[];
    uint64_t* rows = lds;                                                          // [w_out][kSmallPitch]
    uint16_t* stacks = reinterpret_cast<uint16_t*>(rows + (size_t)w_out * kSmallPitch);  // [kSmallStack][kSmallThreads]; a technical-bin index fits 16 bits here
    const uint32_t tid = threadIdx.x;
    for (size_t base = (size_t)blockIdx.x * kSmallThreads; base < n; base += (size_t)gridDim.x * kSmallThreads) {
        const size_t i = base + tid;
        const bool live = i < n;
        const uint64_t v = live ? kmers[i] : 0;
        for (uint32_t w = 0; w < w_out; ++w) rows[w * kSmallPitch + tid] = 0;
        stacks[tid] = (uint16_t)t.root_entry;
        uint32_t count = live ? 1u : 0u;
        while (count) {  // lanes leave the loop one by one
            const HibfNode nd = t.nodes[stacks[--count * kSmallThreads + tid]];
            const uint32_t stride = nd.stride(), words_per_row = nd.words_per_row();
            const uint64_t* words = (const uint64_t*)nd.words;
            uint64_t acc[4] = {~0ULL, stride > 1 ? ~0ULL : 0ULL, stride > 2 ? ~0ULL : 0ULL, stride > 2 ? ~0ULL : 0ULL};
#pragma unroll
            for (uint32_t j = 0; j < 5; ++j) {
                if (j >= h_max) continue;
                const uint32_t jj = j < nd.hash_funs() ? j : nd.hash_funs() - 1;  // a missing hash function repeats the last real row
                const uint64_t r = hash_row_seeded32(v * kSeeds[jj], nd.hash_shift(), nd.bin_size);
                if (stride == 1) {
                    acc[0] &= gload(words + r);
                } else {  // rows are padded to an even number of words, the padding is zero
                    const ulonglong2 a = gload2(words + r * stride);
                    acc[0] &= a.x;
                    acc[1] &= a.y;
                    if (stride > 2) {
                        const ulonglong2 b = gload2(words + r * stride + 2);
                        acc[2] &= b.x;
                        acc[3] &= b.y;
                    }
                }
            }
            uint64_t mg[4] = {0, 0, 0, 0}, dn[4] = {0, 0, 0, 0};
            if (nd.has_merged()) {  // merged-bin masks are padded to 4 words per IBF
                const ulonglong2 m0 = gload2(t.merged + nd.moff), m1 = gload2(t.merged + nd.moff + 2);
                const ulonglong2 d0 = gload2(t.descend + nd.moff), d1 = gload2(t.descend + nd.moff + 2);
                mg[0] = m0.x; mg[1] = m0.y; mg[2] = m1.x; mg[3] = m1.y;
                dn[0] = d0.x; dn[1] = d0.y; dn[2] = d1.x; dn[3] = d1.y;
            }
#pragma unroll
            for (uint32_t w = 0; w < 4; ++w) {
                if (w >= words_per_row) continue;
                uint64_t kids = acc[w] & dn[w];  // children outside this shard's columns are not visited
                while (kids) {
                    const uint32_t tb = w * 64u + (uint32_t)__builtin_ctzll(kids);
                    kids &= kids - 1;
                    if (count < kSmallStack) stacks[count++ * kSmallThreads + tid] = (uint16_t)(nd.off + tb);
                }
                uint64_t hits = acc[w] & ~mg[w];
                if (!hits) continue;
                if (nd.ident_word != kNoIdent) {  // the row word is a mask word
                    const uint64_t word = (uint64_t)nd.ident_word + w;
                    if (word >= word0 && word < (uint64_t)word0 + w_out) rows[(word - word0) * kSmallPitch + tid] |= hits;
                    continue;
                }
                // padding bits are never set; this guards the map look-up
                if (w * 64u + 63u >= nd.bins) hits &= nd.bins > w * 64u ? (~0ULL >> (63u - ((nd.bins - 1u) & 63u))) : 0;
                while (hits) {
                    const uint32_t tb = w * 64u + (uint32_t)__builtin_ctzll(hits);
                    hits &= hits - 1;
                    const uint64_t ub = gload(t.tb_user + nd.off + tb);
                    const uint64_t mw = ub >> 6;
                    if (mw >= word0 && mw < (uint64_t)word0 + w_out) rows[(mw - word0) * kSmallPitch + tid] |= 1ULL << (ub & 63);
                }
            }
        }
        if (alive) {  // base is a multiple of 256: every wave owns whole alive words
            uint64_t any = 0;
            for (uint32_t w = 0; w < w_out; ++w) any |= rows[w * kSmallPitch + tid];
            const uint64_t bits = __ballot(live && any != 0);
            if ((tid & 63) == 0 && base + (tid & ~63u) < n) alive[(base + tid) >> 6] = bits;
        }
        __syncthreads();
        // the workgroup's rows are one contiguous piece of the output: element e = row e / w_out, word e % w_out
        const size_t rows_here = n - base < kSmallThreads ? n - base : kSmallThreads;
        const size_t elems = rows_here * w_out;
        uint64_t* out = masks + base * w_out;
        for (size_t e = tid; e < elems; e += kSmallThreads) {
            const uint32_t r = (uint32_t)(e / w_out), w = (uint32_t)(e % w_out);
            __builtin_nontemporal_store(rows[w * kSmallPitch + r], out + e);
        }
        __syncthreads();  // the next chunk reuses the rows
    }
}


// ---- child-stationary descent of regular two-level trees ------------------------------------------------
// The k-mer-stationary kernels above make every IBF row a cold cache-line fill: a k-mer of the 65536-bin tree reads
// ~270 random 32-byte rows out of a 31 MB tree that an XCD's 4 MB L2 mostly misses (round 1: 14 of 21 ms per 4 M
// k-mers).  Here the CHILDREN stay put instead.  The shard's children are cut into groups whose matrices fit an
// XCD's L2 (~2 MB: 32 children of the 65536-bin tree); workgroup b works on group b % 8 — the dispatcher deals
// workgroups round-robin over the 8 XCDs, so every XCD keeps probing the same 2 MB (a speed assumption only) —
// and streams a tile of k-mers past it.  A wave step covers 128 mask words = 64 lanes x 16 B: lane l owns 16
// bytes of child l / lanes_per_child, gathers that piece of the child's h rows (L2 hits), ANDs, and the wave
// writes the k-mer's 1-KiB row segment with one coalesced non-temporal store — a child that the root row rules
// out costs nothing but the zeros.  No LDS row, no atomics: the row segments of different groups are disjoint.
// Pass 1 (hibf_root_kernel) probes the root once per k-mer and leaves its row (one bit per child) in HBM.

// row r of every child side by side: out[r][c * wpr + w] = child c's word w of row r (Index::interleaved)
__global__ __launch_bounds__(256) void interleave_children_kernel(const ChildRec* __restrict__ ch, uint32_t n_children, uint32_t wpr, uint64_t rows,
                                                                  uint64_t* __restrict__ out, uint32_t out_stride) {
    const uint64_t per_row = (uint64_t)n_children * wpr, total = rows * per_row;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / per_row;
        const uint32_t w = (uint32_t)(i % per_row), c = w / wpr;
        out[r * out_stride + w] = gload((const uint64_t*)ch[c].words + r * wpr + w % wpr);
    }
}

// child_rows != null (all children of one size): also the k-mer's row indexes in a child, child_hf per k-mer — they
// are the same in every child, so the children kernel reads them instead of hashing again for each of its groups
__global__ __launch_bounds__(256) void hibf_root_kernel(HibfNode root, const uint64_t* __restrict__ kmers, size_t n,
                                                        uint64_t* __restrict__ cm, uint32_t row_words, uint32_t* __restrict__ child_rows,
                                                        uint32_t child_bin_size, uint32_t child_shift, uint32_t child_hf) {
    const uint64_t* words = (const uint64_t*)root.words;
    const uint32_t stride = root.stride(), hf = root.hash_funs();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t v = kmers[i];
        if (child_rows)
            for (uint32_t j = 0; j < child_hf; ++j) child_rows[i * child_hf + j] = (uint32_t)hash_row_seeded32(v * kSeeds[j], child_shift, child_bin_size);
        uint32_t r[5];
#pragma unroll
        for (uint32_t j = 0; j < 5; ++j) r[j] = j < hf ? (uint32_t)hash_row_seeded32(v * kSeeds[j], root.hash_shift(), root.bin_size) : 0;
        if (stride == 1) {
            uint64_t x = ~0ULL;
#pragma unroll
            for (uint32_t j = 0; j < 5; ++j) if (j < hf) x &= gload(words + r[j]);
            cm[i * row_words] = x;
            continue;
        }
        for (uint32_t w = 0; w < row_words; w += 2) {
            ulonglong2 x{~0ULL, ~0ULL};
#pragma unroll
            for (uint32_t j = 0; j < 5; ++j)
                if (j < hf) { const ulonglong2 a = gload2(words + (size_t)r[j] * stride + w); x.x &= a.x; x.y &= a.y; }
            cm[i * row_words + w] = x.x;
            if (w + 1 < row_words) cm[i * row_words + w + 1] = x.y;
        }
    }
}

typedef uint32_t hu32x4 __attribute__((ext_vector_type(4)));

// 16-byte store of a row segment with a chosen cache policy (gfx950 cache-control bits; MI355X_MICROARCH.md, "stores of each
// flavour": plain / nt stores keep the line in the XCD's L2, sc1 stores drop it).  The row stream is 8 KB per k-mer against
// 2 MB of child matrices per XCD that must stay in L2, so the default is a store that does not stay.  flavour is wave-uniform.
__device__ __forceinline__ void store_row16(hu32x4* p, hu32x4 v, int flavour) {
    switch (flavour) {
        case 1: *p = v; break;
        case 2: asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); break;
        case 3: asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); break;
        case 4: asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory"); break;
        default: __builtin_nontemporal_store(v, p); break;
    }
}

// U k-mers in flight per wave: U * h 16-byte gathers per lane before the first AND.
// UNIFORM: all children have the same rows / hash shift / hash count (the synthetic trees, and any index whose bins
// are equally full): the row indexes are then the same in every lane and the hash runs on the scalar unit — with
// per-lane hashing the 64-bit multiplies were 40 % of the kernel's time (1 M k-mers x 8 groups x 22 quarter-rate
// multiplies).  Otherwise only the seed products are scalar.
template <int U, bool UNIFORM>
__global__ __launch_bounds__(256) void hibf_children_kernel(const ChildRec* __restrict__ recs, uint32_t n_children, uint32_t lanes_per_child,
                                                            const uint64_t* __restrict__ kmers, size_t n, const uint64_t* __restrict__ cm,
                                                            uint32_t cm_words, uint64_t* __restrict__ masks, uint32_t w_out,
                                                            uint32_t n_steps, uint32_t steps_per_group, uint32_t groups_per_phase,
                                                            uint32_t n_tiles, uint32_t tile, uint32_t h_max, uint64_t* __restrict__ alive,
                                                            int store_flavour, const uint32_t* __restrict__ child_rows) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), n_waves = blockDim.x >> 6;
    const uint32_t gp = blockIdx.x % groups_per_phase, rest = blockIdx.x / groups_per_phase;
    const uint32_t group = (rest / n_tiles) * groups_per_phase + gp;
    const size_t first = (size_t)(rest % n_tiles) * tile;
    const size_t last = first + tile < n ? first + tile : n;
    const uint32_t children_per_step = 64u / lanes_per_child;
    for (uint32_t s = 0; s < steps_per_group; ++s) {
        const uint32_t step = group * steps_per_group + s;  // wave-uniform
        if (step >= n_steps) break;
        const uint32_t child = step * children_per_step + lane / lanes_per_child;
        const bool valid = child < n_children;
        const ChildRec rec = recs[valid ? child : 0];
        uint32_t shift = rec.packed & 0xFFu, hf = (rec.packed >> 8) & 0xFu, bin_size = rec.bin_size;
        const uint32_t tb = rec.packed >> 12;
        if (UNIFORM) {
            shift = (uint32_t)__builtin_amdgcn_readfirstlane((int)shift);
            hf = (uint32_t)__builtin_amdgcn_readfirstlane((int)hf);
            bin_size = (uint32_t)__builtin_amdgcn_readfirstlane((int)bin_size);
        }
        const uint32_t stride = lanes_per_child * 2u;
        const uint64_t* words = (const uint64_t*)rec.words + (lane % lanes_per_child) * 2u;
        uint64_t* out = masks + (size_t)step * 128u + lane * 2u;
        const uint64_t* cmw = cm + (tb >> 6);
        for (size_t i0 = first + (size_t)wave * U; i0 < last; i0 += (size_t)n_waves * U) {
            hu32x4 x[U][5];
            bool hit[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t i = i0 + u < last ? i0 + u : last - 1;  // a tail step repeats the last k-mer (and stores it again)
                uint64_t seeded[5] = {0, 0, 0, 0, 0};
                if (!UNIFORM) {
                    const uint64_t v = kmers[i];  // wave-uniform: the seed products are scalar
#pragma unroll
                    for (uint32_t j = 0; j < 5; ++j) seeded[j] = j < h_max ? v * kSeeds[j] : 0;
                }
#ifdef TXQ_EXPERIMENTS
                hit[u] = valid && ((cmw[i * cm_words] >> (tb & 63u)) & 1u) && !(store_flavour & 16);  // bit 4: timing experiment, no row gathers (wrong masks)
#else
                hit[u] = valid && ((cmw[i * cm_words] >> (tb & 63u)) & 1u);
#endif
#pragma unroll
                for (uint32_t j = 0; j < 5; ++j) {
                    if (j >= h_max) continue;
                    // a hash function this child does not have repeats its last real row (AND is idempotent)
                    uint64_t sv = seeded[j];
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q) if (q < j && hf == q + 1) sv = seeded[q];
                    // UNIFORM: h_max == hf, the rows were hashed once per k-mer by the root pass (a scalar load here)
                    const uint32_t r = UNIFORM ? child_rows[i * h_max + j] : (uint32_t)hash_row_seeded32(sv, shift, bin_size);
                    if (hit[u]) x[u][j] = *(const __attribute__((address_space(1))) hu32x4*)(words + (size_t)r * stride);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                hu32x4 acc{0u, 0u, 0u, 0u};
                if (hit[u]) {
                    acc = x[u][0];
#pragma unroll
                    for (uint32_t j = 1; j < 5; ++j) if (j < h_max) acc &= x[u][j];
                }
                const size_t i = i0 + u < last ? i0 + u : last - 1;
#ifdef TXQ_EXPERIMENTS
                if (valid && (!(store_flavour & 32) || (acc.x == 0x12345u && acc.y == 0x54321u)))  // bit 5: timing experiment, (almost) no stores (wrong masks)
#else
                if (valid)
#endif
                    store_row16(reinterpret_cast<hu32x4*>(out + i * w_out), acc, store_flavour & 15);
                if (alive) {
                    const bool some = __ballot((acc.x | acc.y | acc.z | acc.w) != 0u) != 0;
                    if (some && lane == 0) atomicOr((unsigned long long*)(alive + (i >> 6)), 1ULL << (i & 63));
                }
            }
        }
    }
}

// the frontier count of a level can exceed the capacity only through a bug; clamp for the reader
__global__ void hibf_clamp_kernel(uint32_t* count, uint32_t cap) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && *count > cap) *count = cap;
}

// alive[i/64] bit i%64 = any word of mask row i is non-zero
__global__ __launch_bounds__(256) void mask_alive_kernel(const uint64_t* __restrict__ masks, size_t n, uint32_t w,
                                                         uint64_t* __restrict__ alive) {
    const size_t n_pad = (n + 63) & ~(size_t)63;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += stride) {
        uint64_t any = 0;
        if (i < n)
            for (uint32_t j = 0; j < w; ++j) any |= masks[i * w + j];
        const uint64_t bits = __ballot(any != 0);
        if ((threadIdx.x & 63) == 0) alive[i >> 6] = bits;
    }
}

// ---- layout order (txq_internal.hpp VChunk): the rows of a general HIBF, level by level, child-stationary ----------
// One launch per level of the tree.  A workgroup = a tile of k-mers x one GROUP of the level's chunks (consecutive
// chunks whose IBFs' rows fit an XCD's L2: workgroup b takes group b % 8 of its phase, and the dispatcher deals
// workgroups round-robin over the 8 XCDs, so an XCD keeps probing the same few MB — a speed assumption only).
// A lane owns one 16-byte chunk of the row (two words of one IBF) for the whole tile: its IBF's parameters stay in
// registers, the k-mers stream past.  Per k-mer: the gate (the bit of the parent's row that leads to this IBF, written
// by the previous level's launch), then — only where it is set — h row gathers, AND, and one coalesced store (a wave
// covers 1 KiB of the row).  Chunks of IBFs that were not reached get zeros: every byte of the row is written once.
// CW: words per chunk (2: 16-byte lanes; 1: 8-byte lanes, for trees of narrow IBFs whose rows would double in width if every
// IBF were padded to 16 bytes).  H: the most hash functions of any IBF of the tree.  The lanes of a wave are shared out
// between chunks and k-mers: a level with few chunks (the root: often one) takes as many k-mers per wave step instead.
template <int CW, int H>
__global__ __launch_bounds__(256) void hibf_layout_level_kernel(const VChunk* __restrict__ chunks, const uint32_t* __restrict__ group_first,
                                                                uint32_t n_groups, const uint64_t* __restrict__ kmers, size_t n,
                                                                uint64_t* __restrict__ rows, uint32_t v_words, uint32_t n_tiles, uint32_t tile, uint32_t experiment) {
    constexpr int U = 4;  // k-mers in flight per lane: U gates, then up to U * h row gathers
    (void)experiment;  // (TXQ_EXPERIMENTS builds: 16 no row gathers, 32 no stores, 64 no gate loads — timing only, wrong rows)
    const uint32_t gsel = blockIdx.x % 8u, t = (blockIdx.x / 8u) % n_tiles, phase = blockIdx.x / (8u * n_tiles);
    const uint32_t g = phase * 8u + gsel;
    if (g >= n_groups) return;
    const uint32_t c0 = group_first[g], c1 = group_first[g + 1], n_c = c1 - c0;
    const size_t k0 = (size_t)t * tile, k1 = k0 + tile < n ? k0 + tile : n;
    uint32_t cw = 1;  // chunk lanes per wave; the other lane bits take different k-mers
    while (cw < 64u && cw < n_c) cw <<= 1;
    const uint32_t kpw = 64u / cw, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t csub = lane % cw, ksub = lane / cw, k_step = (blockDim.x >> 6) * kpw;
    for (uint32_t cb = 0; cb < n_c; cb += cw) {
        const uint32_t c = c0 + cb + csub;
        if (cb + csub >= n_c) continue;
        const VChunk rec = chunks[c];
        const uint64_t* words = (const uint64_t*)rec.words;
        const uint32_t stride = rec.packed & 0xFFFFFu, shift = (rec.packed >> 20) & 63u, hf = (rec.packed >> 26) & 7u;
        const bool single = CW == 2 && ((rec.packed >> 29) & 1u);
        for (size_t i0 = k0 + wave * kpw + ksub; i0 < k1; i0 += (size_t)k_step * U) {
            uint64_t v[U];
            bool pass[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t i = i0 + (size_t)u * k_step;
                pass[u] = i < k1;
                v[u] = pass[u] ? kmers[i] : 0;
#ifdef TXQ_EXPERIMENTS
                if (experiment & 64u) { if (rec.gate_word != kNoGate) pass[u] = pass[u] && ((v[u] >> 3) & 1ULL); continue; }
#endif
                if (pass[u] && rec.gate_word != kNoGate) pass[u] = (gload(rows + i * v_words + rec.gate_word) >> rec.gate_bit) & 1ULL;
            }
            ulonglong2 x[U][H];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < H; ++j) {
                    x[u][j] = ulonglong2{~0ULL, ~0ULL};
                    if ((uint32_t)j >= hf || !pass[u]) continue;
#ifdef TXQ_EXPERIMENTS
                    if (experiment & 16u) continue;
#endif
                    const uint64_t r = hash_row_seeded32(v[u] * kSeeds[j], shift, rec.bin_size);
                    if (CW == 1 || single) x[u][j].x = gload(words + r * stride + rec.col);
                    else x[u][j] = gload2(words + r * stride + rec.col);
                }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t i = i0 + (size_t)u * k_step;
                if (i >= k1) break;
                uint64_t a = 0, b = 0;
                if (pass[u] && hf) {
                    a = b = ~0ULL;
#pragma unroll
                    for (int j = 0; j < H; ++j) { a &= x[u][j].x; b &= x[u][j].y; }
                    if (single) b = 0;
                }
                uint64_t* out = rows + i * v_words + (size_t)(c - 0) * CW;
#ifdef TXQ_EXPERIMENTS
                if ((experiment & 32u) && (a | b) != 0x123456789ULL) continue;
#endif
                if (CW == 1) __builtin_nontemporal_store(a, out);
                else store_row16(reinterpret_cast<hu32x4*>(out), hu32x4{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)}, 0);
            }
        }
    }
}

// final masks of a layout-order session -> user-bin order (split bins: several technical bins, one user bin: ORed)
__global__ __launch_bounds__(256) void hibf_layout_to_user_kernel(const uint64_t* __restrict__ rows, size_t n, uint32_t v_words, const uint64_t* __restrict__ leaf,
                                                                  const uint32_t* __restrict__ vuser, uint64_t* __restrict__ out, uint32_t w_out) {
    const size_t total = n * (size_t)v_words;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / v_words;
        const uint32_t w = (uint32_t)(i % v_words);
        for (uint64_t x = rows[i] & leaf[w]; x; x &= x - 1) {
            const uint32_t ub = vuser[(size_t)w * 64 + (uint32_t)__builtin_ctzll(x)];
            atomicOr((unsigned long long*)(out + p * w_out + (ub >> 6)), 1ULL << (ub & 63u));
        }
    }
}

#define TXQ_HIP(call)                                        \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return fail_hip(e_, #call);    \
    } while (0)

// Layout order for a tree that is not regular (txq_internal.hpp VChunk): the rows of all IBFs, levels ascending, each IBF
// padded to whole 16-byte chunks; per chunk its record, per IBF its ancestors, which bits are user bins, and the user
// bin behind every bit.  Single shard only (a column shard of the USER bins does not cut the layout-order row in one piece).
// The side matrix of an IBF with split user bins (txq_internal.hpp VSplit): one thread per row copies the bits of the IBF's
// non-representative parts (`entries`) to their places in the side row (`pos`: word * 64 + bit).
__global__ __launch_bounds__(256) void build_side_matrix_kernel(const uint64_t* __restrict__ words, uint32_t stride, uint64_t rows, const VSplit* __restrict__ entries,
                                                                const uint32_t* __restrict__ pos, uint32_t n_entries, uint64_t* __restrict__ side, uint32_t side_stride) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint64_t* row = words + r * stride;
    uint64_t* out = side + r * side_stride;
    uint64_t acc = 0;
    uint32_t word = 0;
    for (uint32_t e = 0; e < n_entries; ++e) {  // (the entries are in side order)
        const uint32_t p = pos[e];
        if ((p >> 6) != word) { out[word] = acc; acc = 0; word = p >> 6; }
        const VSplit sp = entries[e];
        acc |= ((row[sp.part_word] >> sp.part_bit) & 1ULL) << (p & 63);
    }
    out[word] = acc;
}

static int build_layout_order(Index& ix, const txq_index_desc& desc, const std::vector<int>& level, const std::vector<uint64_t>& next,
                              const std::vector<uint64_t>& tbu, const std::vector<uint64_t>& off, const std::vector<HibfNode>* nodes) {
    const uint64_t n = desc.n_ibf;
    if (ix.shard_words != ix.mask_words || ix.shard_word0 != 0 || ix.depth > kMaxVDepth + 1 || desc.user_bins >= kNoGate) return TXQ_OK;
    for (const IbfDev& f : ix.ibf)
        if ((f.bin_size >> 32) || f.stride >= (1u << 20) || f.hash_funs > 5) return TXQ_OK;
    // IBFs by level (BFS order within a level), their segments in the row
    std::vector<std::vector<uint64_t>> by_level(ix.depth);
    for (uint64_t i = 0; i < n; ++i) by_level[level[i]].push_back(i);
    std::vector<uint64_t> seg(n, 0), parent(n, UINT64_MAX), parent_tb(n, 0);
    for (uint64_t i = 0; i < n; ++i)
        for (uint64_t b = 0; b < desc.ibf[i].bins; ++b)
            if (tbu[off[i] + b] == TXQ_MERGED_BIN) { parent[next[off[i] + b]] = i; parent_tb[next[off[i] + b]] = b; }
    // 16-byte chunks (every IBF padded to an even number of words) unless that widens the row by more than 30 % — trees of
    // many one-word IBFs —: then 8-byte chunks
    uint64_t exact = 0, padded2 = 0;
    for (uint64_t i = 0; i < n; ++i) { exact += desc.ibf[i].bin_words; padded2 += (desc.ibf[i].bin_words + 1) & ~(uint64_t)1; }
    const uint64_t cwords = padded2 * 10 > exact * 13 ? 1 : 2;
    uint64_t words = 0;
    for (auto& lv : by_level)
        for (uint64_t i : lv) { seg[i] = words; words += (desc.ibf[i].bin_words + cwords - 1) / cwords * cwords; }
    const bool pad_word = (words & 1) != 0;  // (slot masks of an even number of words: one word that belongs to no IBF)
    if (pad_word) ++words;
    if (words >= (1u << 26)) return TXQ_OK;
    std::vector<VChunk> chunks;
    std::vector<uint32_t> chunk0(n, 0);  // an IBF's first chunk
    std::vector<VPath> paths(n);
    std::vector<uint64_t> leaf(words, 0);
    std::vector<uint32_t> vuser(words * 64, kNoGate);
    std::vector<uint32_t> groups;  // per level: first chunk of each group, then the level's end
    ix.vlevels.clear();
    auto packed_of = [](const IbfDev& f) { return f.stride | (f.hash_shift << 20) | (f.hash_funs << 26) | ((uint32_t)(f.stride == 1) << 29); };
    for (auto& lv : by_level) {
        VLevel L;
        L.first_chunk = (uint32_t)chunks.size();
        uint64_t group_bytes = 0;
        for (uint64_t i : lv) {
            const IbfDev& f = ix.ibf[i];
            const uint64_t bytes = f.bin_size * (uint64_t)f.stride * 8;
            if (L.group_first.empty() || group_bytes + bytes > ((uint64_t)2 << 20)) { L.group_first.push_back((uint32_t)chunks.size()); group_bytes = 0; }
            group_bytes += bytes;
            const uint64_t padded = (desc.ibf[i].bin_words + cwords - 1) / cwords * cwords;
            chunk0[i] = (uint32_t)chunks.size();
            for (uint64_t c = 0; c < padded; c += cwords) {
                VChunk r{};
                r.words = (uint64_t)(uintptr_t)f.words;
                r.bin_size = (uint32_t)f.bin_size;
                r.packed = packed_of(f);
                r.col = (uint32_t)c;
                r.gate_word = parent[i] == UINT64_MAX ? kNoGate : (uint32_t)(seg[parent[i]] + (parent_tb[i] >> 6));
                r.gate_bit = (uint32_t)(parent_tb[i] & 63);
                r.ibf = (uint32_t)i;
                chunks.push_back(r);
            }
            for (uint64_t b = 0; b < desc.ibf[i].bins; ++b)
                if (tbu[off[i] + b] != TXQ_MERGED_BIN) {
                    leaf[seg[i] + (b >> 6)] |= 1ULL << (b & 63);
                    vuser[(seg[i] + (b >> 6)) * 64 + (b & 63)] = (uint32_t)tbu[off[i] + b];
                }
            VPath& p = paths[i];
            p.depth = 0;
            std::vector<uint64_t> chain;  // i's ancestors, nearest first
            for (uint64_t a = i; parent[a] != UINT64_MAX; a = parent[a]) chain.push_back(a);
            for (size_t at = chain.size(); at-- > 0;) {  // root first
                const uint64_t child = chain[at], a = parent[child];
                const IbfDev& fa = ix.ibf[a];
                auto& slot = p.anc[p.depth++];
                slot.words = (uint64_t)(uintptr_t)fa.words;
                slot.bin_size = (uint32_t)fa.bin_size;
                slot.packed = packed_of(fa);
                slot.word = (uint32_t)(parent_tb[child] >> 6);
                slot.bit = (uint32_t)(parent_tb[child] & 63);
            }
        }
        if (pad_word && &lv == &by_level.back()) {  // the padding word: a chunk without hash functions — always zero
            VChunk r{};
            r.words = (uint64_t)(uintptr_t)ix.ibf[0].words;
            r.bin_size = 1;
            r.packed = 1;  // stride 1, no hash function
            r.gate_word = kNoGate;
            chunks.push_back(r);
        }
        L.n_chunks = (uint32_t)chunks.size() - L.first_chunk;
        L.group_first.push_back((uint32_t)chunks.size());
        ix.vlevels.push_back(L);
    }
    for (VLevel& L : ix.vlevels) {  // the groups of all levels in one device array; group_first becomes offsets into it
        const uint32_t at = (uint32_t)groups.size();
        groups.insert(groups.end(), L.group_first.begin(), L.group_first.end());
        const uint32_t ng = (uint32_t)L.group_first.size() - 1;
        L.group_first.assign({at, ng});
    }
    if (nodes) {  // the fused kernel's node records with every IBF's place in the layout-order row (hibf_fused_kernel<G, LAYOUT>)
        std::vector<HibfNode> vn(*nodes);
        for (uint64_t i = 0; i < n; ++i)
            for (uint64_t b = 0; b < desc.ibf[i].bins; ++b)
                if (tbu[off[i] + b] == TXQ_MERGED_BIN) vn[off[i] + b].ident_word = (uint32_t)seg[next[off[i] + b]];
        vn[off[n]].ident_word = (uint32_t)seg[0];
        TXQ_HIP(hipMalloc((void**)&ix.d_vnodes, vn.size() * sizeof(HibfNode)));
        TXQ_HIP(hipMemcpy(ix.d_vnodes, vn.data(), vn.size() * sizeof(HibfNode), hipMemcpyHostToDevice));
        ix.device_bytes += vn.size() * sizeof(HibfNode);
    }
    TXQ_HIP(hipMalloc((void**)&ix.d_vchunks, chunks.size() * sizeof(VChunk)));
    TXQ_HIP(hipMalloc((void**)&ix.d_vpaths, paths.size() * sizeof(VPath)));
    TXQ_HIP(hipMalloc((void**)&ix.d_vleaf, leaf.size() * 8));
    TXQ_HIP(hipMalloc((void**)&ix.d_vuser, vuser.size() * 4));
    TXQ_HIP(hipMalloc((void**)&ix.d_vgroups, groups.size() * 4));
    TXQ_HIP(hipMemcpy(ix.d_vchunks, chunks.data(), chunks.size() * sizeof(VChunk), hipMemcpyHostToDevice));
    TXQ_HIP(hipMemcpy(ix.d_vpaths, paths.data(), paths.size() * sizeof(VPath), hipMemcpyHostToDevice));
    TXQ_HIP(hipMemcpy(ix.d_vleaf, leaf.data(), leaf.size() * 8, hipMemcpyHostToDevice));
    TXQ_HIP(hipMemcpy(ix.d_vuser, vuser.data(), vuser.size() * 4, hipMemcpyHostToDevice));
    TXQ_HIP(hipMemcpy(ix.d_vgroups, groups.data(), groups.size() * 4, hipMemcpyHostToDevice));
    {   // split user bins (txq_internal.hpp VSplit): per IBF the technical bins by user bin; the lowest part represents the bin
        std::vector<uint64_t> nonrep(words, 0);
        std::vector<uint32_t> rep_pos;
        std::vector<std::vector<VSplit>> per_chunk(chunks.size());
        bool any = false;
        std::vector<std::pair<uint64_t, uint64_t>> bins_of;  // (user bin, technical bin) of one IBF
        for (uint64_t i = 0; i < n; ++i) {
            bins_of.clear();
            for (uint64_t b = 0; b < desc.ibf[i].bins; ++b)
                if (tbu[off[i] + b] != TXQ_MERGED_BIN) bins_of.emplace_back(tbu[off[i] + b], b);
            std::sort(bins_of.begin(), bins_of.end());
            for (size_t at = 0; at < bins_of.size();) {
                size_t end = at + 1;
                while (end < bins_of.size() && bins_of[end].first == bins_of[at].first) ++end;
                if (end - at > 1) {
                    if (!any) { any = true; rep_pos.assign(words * 64, kNoGate); }
                    const uint64_t rep = bins_of[at].second;  // (sorted: the lowest technical bin)
                    const uint32_t chunk = chunk0[i] + (uint32_t)((rep >> 6) / cwords);
                    const uint16_t rep_bit = (uint16_t)(rep - (uint64_t)((rep >> 6) / cwords) * cwords * 64);
                    for (size_t j = at + 1; j < end; ++j) {
                        const uint64_t part = bins_of[j].second;
                        nonrep[seg[i] + (part >> 6)] |= 1ULL << (part & 63);
                        rep_pos[(seg[i] + (part >> 6)) * 64 + (part & 63)] = (uint32_t)((seg[i] + (rep >> 6)) * 64 + (rep & 63));
                        per_chunk[chunk].push_back(VSplit{(uint32_t)(part >> 6), rep_bit, (uint16_t)(part & 63)});
                    }
                }
                at = end;
            }
        }
        if (any) {
            std::vector<VSplitRange> ranges(chunks.size());
            std::vector<VSplit> flat;
            std::vector<uint32_t> side_pos;             // per entry: its bit in its IBF's side row (word * 64 + bit)
            std::vector<uint64_t> side_off(n + 1, 0);   // per IBF: first word of its side matrix in d_vside
            std::vector<uint32_t> side_stride(n, 0);
            for (uint64_t i = 0; i < n; ++i) {
                uint32_t word = 0, used = 0, last_word = 0;
                bool has = false;
                const uint64_t padded = (desc.ibf[i].bin_words + cwords - 1) / cwords * cwords;
                for (uint32_t c = chunk0[i]; c < chunk0[i] + padded / cwords; ++c) {
                    std::stable_sort(per_chunk[c].begin(), per_chunk[c].end(), [](const VSplit& x, const VSplit& y) { return x.rep_bit < y.rep_bit; });
                    const uint32_t cnt = (uint32_t)per_chunk[c].size();
                    ranges[c] = VSplitRange{(uint32_t)flat.size(), cnt, {0, 0, 0, 0}, 0, 0, 0};
                    if (!cnt) continue;
                    has = true;
                    // a chunk's parts (at most 127) are consecutive side bits from bit0 of one word on, into the next word if need be
                    if (used && used + cnt > 64) { ++word; used = 0; }
                    ranges[c].bit0 = used;
                    ranges[c].side = word;  // (the word for now; the pointer once the matrices have their place)
                    for (uint32_t e = 0; e < cnt; ++e) {
                        ranges[c].reps[per_chunk[c][e].rep_bit >> 5] |= 1u << (per_chunk[c][e].rep_bit & 31);
                        side_pos.push_back(word * 64 + used + e);
                    }
                    used += cnt;
                    last_word = word + (used - 1) / 64;
                    while (used >= 64) { used -= 64; ++word; }
                    flat.insert(flat.end(), per_chunk[c].begin(), per_chunk[c].end());
                }
                side_stride[i] = has ? last_word + 1 : 0;
                side_off[i + 1] = side_off[i] + (uint64_t)side_stride[i] * ix.ibf[i].bin_size;
            }
            TXQ_HIP(hipMalloc((void**)&ix.d_vside, std::max<uint64_t>(side_off[n], 1) * 8));
            TXQ_HIP(hipMemset(ix.d_vside, 0, std::max<uint64_t>(side_off[n], 1) * 8));
            for (uint64_t i = 0; i < n; ++i) {
                const uint64_t padded = (desc.ibf[i].bin_words + cwords - 1) / cwords * cwords;
                for (uint32_t c = chunk0[i]; c < chunk0[i] + padded / cwords; ++c) {
                    ranges[c].side_stride = side_stride[i];
                    ranges[c].side = (uint64_t)(uintptr_t)(ix.d_vside + side_off[i] + ranges[c].side);
                }
            }
            // a chunk that holds representatives says so in its record (VChunk::packed bit 30): the others never look at their range
            for (size_t c = 0; c < chunks.size(); ++c)
                if (ranges[c].count) chunks[c].packed |= 1u << 30;
            TXQ_HIP(hipMemcpy(ix.d_vchunks, chunks.data(), chunks.size() * sizeof(VChunk), hipMemcpyHostToDevice));
            TXQ_HIP(hipMalloc((void**)&ix.d_vnonrep, nonrep.size() * 8));
            TXQ_HIP(hipMalloc((void**)&ix.d_vrep, rep_pos.size() * 4));
            TXQ_HIP(hipMalloc((void**)&ix.d_vsplit_range, ranges.size() * sizeof(VSplitRange)));
            TXQ_HIP(hipMalloc((void**)&ix.d_vsplits, flat.size() * sizeof(VSplit)));
            TXQ_HIP(hipMemcpy(ix.d_vnonrep, nonrep.data(), nonrep.size() * 8, hipMemcpyHostToDevice));
            TXQ_HIP(hipMemcpy(ix.d_vrep, rep_pos.data(), rep_pos.size() * 4, hipMemcpyHostToDevice));
            TXQ_HIP(hipMemcpy(ix.d_vsplit_range, ranges.data(), ranges.size() * sizeof(VSplitRange), hipMemcpyHostToDevice));
            TXQ_HIP(hipMemcpy(ix.d_vsplits, flat.data(), flat.size() * sizeof(VSplit), hipMemcpyHostToDevice));
            {   // the side matrices: every IBF's entries are consecutive in `flat` (its chunks are)
                uint32_t* d_pos = nullptr;
                TXQ_HIP(hipMalloc((void**)&d_pos, std::max<size_t>(side_pos.size(), 1) * 4));
                TXQ_HIP(hipMemcpy(d_pos, side_pos.data(), side_pos.size() * 4, hipMemcpyHostToDevice));
                for (uint64_t i = 0; i < n; ++i) {
                    if (!side_stride[i]) continue;
                    const uint64_t padded = (desc.ibf[i].bin_words + cwords - 1) / cwords * cwords;
                    const uint32_t e0 = ranges[chunk0[i]].first;
                    const VSplitRange& last = ranges[chunk0[i] + padded / cwords - 1];
                    const uint32_t e1 = last.first + last.count;
                    const IbfDev& f = ix.ibf[i];
                    build_side_matrix_kernel<<<(unsigned)((f.bin_size + 255) / 256), 256>>>(f.words, f.stride, f.bin_size, ix.d_vsplits + e0, d_pos + e0, e1 - e0,
                                                                                       ix.d_vside + side_off[i], side_stride[i]);
                }
                hipError_t e = hipDeviceSynchronize();
                (void)hipFree(d_pos);
                if (e != hipSuccess) return fail_hip(e, "building the side matrices of split bins");
            }
            ix.device_bytes += side_off[n] * 8;
            ix.device_bytes += nonrep.size() * 8 + rep_pos.size() * 4 + ranges.size() * sizeof(VSplitRange) + flat.size() * sizeof(VSplit);
            // the ONES of a layout-order session: a split bin is its representative
            for (size_t w = 0; w < leaf.size(); ++w) leaf[w] &= ~nonrep[w];
            TXQ_HIP(hipMemcpy(ix.d_vleaf, leaf.data(), leaf.size() * 8, hipMemcpyHostToDevice));
        }
    }
    ix.v_words = (uint32_t)words;
    ix.v_inner_words = 0;
    for (uint64_t i = 0; i < n; ++i) {
        bool inner = false;
        for (uint64_t b = 0; b < desc.ibf[i].bins && !inner; ++b) inner = tbu[off[i] + b] == TXQ_MERGED_BIN;
        if (inner) ix.v_inner_words += (uint32_t)((desc.ibf[i].bin_words + cwords - 1) / cwords * cwords);
    }
    ix.v_chunk_words = (uint32_t)cwords;
    ix.n_vchunks = (uint32_t)chunks.size();
    ix.v_depth = ix.depth - 1;
    ix.tree_hash_max = 1;
    for (const IbfDev& f : ix.ibf) ix.tree_hash_max = std::max(ix.tree_hash_max, f.hash_funs);
    ix.device_bytes += chunks.size() * sizeof(VChunk) + paths.size() * sizeof(VPath) + leaf.size() * 8 + vuser.size() * 4;
    return TXQ_OK;
}

// Waves per workgroup of hibf_fused_kernel: its waves share nothing (a wave's row and stack are its own piece of the LDS, only
// wave-level synchronisation), so the block size is free — taken so that the CU's 160 KB of LDS hold the most waves (a 22 KB
// layout-order row of the 65 536-bin trees: 7 single-wave blocks against 3 blocks of two; an 11 KB user-order row: 14 against 12).
// Entries of a wave's IBF stack kept in LDS (TXQ_HIBF_STACK_LDS, default 128; even: the stack follows the row's 8-byte words);
// the whole stack when the output row could not take the rest.
static uint32_t fused_stack_lds(const Knobs& kn, uint32_t stack_cap, uint32_t w_out) {
    uint32_t in_lds = (uint32_t)std::max(2LL, kn.hibf_stack_lds) & ~1u;
    if (in_lds >= stack_cap || (uint64_t)stack_cap - in_lds > 2ull * w_out) in_lds = (stack_cap + 1) & ~1u;
    return in_lds;
}

static unsigned fused_waves_per_block(size_t wave_bytes) {
    constexpr size_t kLdsPerCu = 160u << 10, kGranule = 1280;  // (allocation granule: the larger of the documented ones — a safe count)
    unsigned best = 1;
    size_t best_resident = 0;
    for (unsigned w = 4; w >= 1; w >>= 1) {
        if (wave_bytes * w > (64u << 10)) continue;
        const size_t block = (wave_bytes * w + kGranule - 1) / kGranule * kGranule;
        const size_t resident = std::min<size_t>(kLdsPerCu / block * w, 32);  // (a CU runs at most 8 waves per SIMD: short rows keep blocks of four)
        if (resident > best_resident) { best_resident = resident; best = w; }  // (ties: the larger block, met first)
    }
    return best;
}

// The layout-order rows with ONE wave per k-mer (hibf_fused_kernel<G, LAYOUT>): the wave walks the IBFs the k-mer reaches,
// keeps the row in LDS and writes it once, coalesced — a k-mer of the 65 536-bin trees reaches a few hundred of the row's
// 1237 chunks, where the level kernels below visit every chunk of every level for every k-mer.  Where a row (plus the stack)
// does not fit a wave's share of the LDS, or TXQ_HIBF_LAYOUT_FUSED=0, false: the level kernels.
static bool layout_order_fused(Index& ix, const uint64_t* d_kmers, size_t n, uint64_t* d_rows, hipStream_t s, int* rc) {
    const Knobs kn = knobs();
    if (!ix.d_vnodes || !kn.hibf_layout_fused) return false;
    const uint32_t w_out = ix.v_words, stack_cap = (uint32_t)ix.ibf.size();
    // (TXQ_HIBF_LAYOUT_DIRECT=0: the row in LDS, written out when the k-mer is done — the A/B and what the tests compare with)
    const bool direct = kn.hibf_layout_direct;
    const uint32_t stack_lds = direct ? ((stack_cap + 1) & ~1u) : fused_stack_lds(kn, stack_cap, w_out);
    const uint32_t row_lds_words = direct ? 0 : w_out;
    const size_t wave_words = (size_t)row_lds_words + stack_lds / 2, wave_bytes = wave_words * 8;
    if (wave_bytes > (64u << 10)) return false;
    uint32_t h_max = 1;
    for (const IbfDev& f : ix.ibf) if (f.hash_funs > h_max) h_max = f.hash_funs;
    HibfView t{ix.d_ibf, ix.d_next, ix.d_tb_user, ix.d_map_off, ix.d_merged, ix.d_merged, ix.d_merged_off, (const HibfNode*)ix.d_vnodes, (uint32_t)ix.hibf_total_tbs};
    t.nonrep = ix.d_vnonrep;
    t.rep_pos = ix.d_vrep;
    const unsigned waves = fused_waves_per_block(wave_bytes);
    const size_t want_waves = kn.hibf_waves > 0 ? (size_t)kn.hibf_waves : (size_t)256 * 64;
    const size_t total_waves = n < want_waves ? n : want_waves;
    const unsigned grid = (unsigned)((total_waves + waves - 1) / waves);
    const uint32_t quads = (ix.max_stride + 3) / 4;  // a lane owns four row words
    int g = 1;
    while (g < 64 && (uint32_t)g < quads) g <<= 1;
    const uint32_t w_iters = (quads + (uint32_t)g - 1) / (uint32_t)g;
    if (w_iters > 1 && ix.d_vnonrep) return false;  // (split bins are unified for one pass of words per lane: the level kernels then)
    hipError_t e;
#define TXQ_FUSED(G) e = launch_fused<G, true>(grid, waves * 64, wave_bytes * waves, s, t, d_kmers, n, d_rows, w_out, 0u, w_iters, stack_cap, (uint32_t)wave_words, h_max, nullptr, stack_lds, row_lds_words)
    switch (g) {
        case 1: TXQ_FUSED(1); break;
        case 2: TXQ_FUSED(2); break;
        case 4: TXQ_FUSED(4); break;
        case 8: TXQ_FUSED(8); break;
        case 16: TXQ_FUSED(16); break;
        case 32: TXQ_FUSED(32); break;
        default: TXQ_FUSED(64); break;
    }
#undef TXQ_FUSED
    *rc = e == hipSuccess ? TXQ_OK : fail_hip(e, "layout-order fused kernel launch");
    return true;
}

// Rows of plain k-mers as the kernels above write them hold every technical bin's own bit; a split user bin becomes its
// representative here (txq_internal.hpp VSplit): one thread per row word, the rare word with a set non-representative bit moves it.
__global__ __launch_bounds__(256) void unify_split_rows_kernel(uint64_t* __restrict__ rows, size_t n_words, uint32_t w_out,
                                                               const uint64_t* __restrict__ nonrep, const uint32_t* __restrict__ rep_pos) {
    for (size_t at = (size_t)blockIdx.x * blockDim.x + threadIdx.x; at < n_words; at += (size_t)gridDim.x * blockDim.x) {
        const uint32_t j = (uint32_t)(at % w_out);
        const uint64_t nr = nonrep[j];
        if (!nr) continue;
        uint64_t moved = rows[at] & nr;
        if (!moved) continue;
        uint64_t* const row = rows + (at - j);
        atomicAnd((unsigned long long*)(row + j), ~moved);  // (another thread may be setting a representative in this very word)
        for (; moved; moved &= moved - 1) {
            const uint32_t to = rep_pos[(size_t)j * 64 + (uint32_t)__builtin_ctzll(moved)];
            atomicOr((unsigned long long*)(row + (to >> 6)), 1ULL << (to & 63));
        }
    }
}

static int unify_split_rows(const Index& ix, uint64_t* d_rows, size_t n, hipStream_t s) {
    if (!ix.d_vnonrep || !n) return TXQ_OK;
    const size_t n_words = n * (size_t)ix.v_words;
    const size_t blocks = std::min<size_t>((n_words + 255) / 256, (size_t)256 * 32);
    unify_split_rows_kernel<<<(unsigned)blocks, 256, 0, s>>>(d_rows, n_words, ix.v_words, ix.d_vnonrep, ix.d_vrep);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TXQ_OK : fail_hip(e, "split-bin kernel launch");
}

int hibf_probe_layout_order(Index& ix, const uint64_t* d_kmers_all, size_t n_all, uint64_t* d_rows_all, hipStream_t s) {
    if (!ix.d_vchunks) return fail(TXQ_ERR_STATE, "the index has no layout order");
    if (!n_all) return TXQ_OK;
    {
        int rc = TXQ_OK;
        if (layout_order_fused(ix, d_kmers_all, n_all, d_rows_all, s, &rc)) return rc;  // (split bins: unified as the rows are written)
    }
    const uint32_t tile = 2048;
    // A level reads the gates the level above it wrote (the words of the IBFs that have children: v_inner_words of a row):
    // a very large batch goes through the levels in pieces whose gates (about 100 MB) are still in the Infinity Cache when
    // the next level asks for them — but never in pieces so small that a level's launch could not fill the device
    // (measured on the 65 536-bin trees of tests/perf_hibf_ragged.py: 1 M k-mers in one piece 28 ms, in pieces of 128 k 47 ms — the
    // launches' tails cost more than the gates' cache misses; pieces are for batches of many millions)
    size_t piece = std::max<size_t>((size_t)4 << 20, (((size_t)96 << 20) / ((size_t)std::max(ix.v_inner_words, 1u) * 8)) / tile * tile);
    for (size_t off = 0; off < n_all; off += piece) {
        const size_t n = std::min(piece, n_all - off);
        const uint64_t* d_kmers = d_kmers_all + off;
        uint64_t* d_rows = d_rows_all + off * ix.v_words;
        const uint32_t n_tiles = (uint32_t)((n + tile - 1) / tile);
        for (const VLevel& L : ix.vlevels) {
            const uint32_t at = L.group_first[0], ng = L.group_first[1];
            const uint32_t phases = (ng + 7) / 8;
            if ((uint64_t)phases * n_tiles * 8 >= ((uint64_t)1 << 31)) return fail(TXQ_ERR_ARG, "too many k-mers for one layout-order probe");
#define TXQ_LEVEL(CW, H) hibf_layout_level_kernel<CW, H><<<phases * n_tiles * 8, 256, 0, s>>>(ix.d_vchunks, ix.d_vgroups + at, ng, d_kmers, n, d_rows, ix.v_words, n_tiles, tile, (uint32_t)knobs().hibf_store & 112u)
            if (ix.v_chunk_words == 1) {
                switch (ix.tree_hash_max) { case 1: TXQ_LEVEL(1, 1); break; case 2: TXQ_LEVEL(1, 2); break; case 3: TXQ_LEVEL(1, 3); break; case 4: TXQ_LEVEL(1, 4); break; default: TXQ_LEVEL(1, 5); break; }
            } else {
                switch (ix.tree_hash_max) { case 1: TXQ_LEVEL(2, 1); break; case 2: TXQ_LEVEL(2, 2); break; case 3: TXQ_LEVEL(2, 3); break; case 4: TXQ_LEVEL(2, 4); break; default: TXQ_LEVEL(2, 5); break; }
            }
#undef TXQ_LEVEL
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail_hip(e, "layout-order level kernel launch");
        }
    }
    return unify_split_rows(ix, d_rows_all, n_all, s);
}

int hibf_layout_to_user(const Index& ix, const uint64_t* d_rows, size_t n, uint64_t* d_out, hipStream_t s) {
    if (!n || !ix.shard_words) return TXQ_OK;
    TXQ_HIP(hipMemsetAsync(d_out, 0, n * ix.shard_words * 8, s));
    size_t blocks = (n * ix.v_words + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hibf_layout_to_user_kernel<<<(unsigned)blocks, 256, 0, s>>>(d_rows, n, ix.v_words, ix.d_vleaf, ix.d_vuser, d_out, (uint32_t)ix.shard_words);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "layout-to-user kernel launch");
    return TXQ_OK;
}

int hibf_upload(Index& ix, const txq_index_desc& desc) {
    const uint64_t n = desc.n_ibf;
    if (n >> 31) return fail(TXQ_ERR_ARG, "too many IBFs");
    // validate the tree on the host before anything reaches the GPU
    std::vector<uint64_t> off(n + 1, 0);
    for (uint64_t i = 0; i < n; ++i) {
        if (!desc.next_ibf_id[i] || !desc.tb_to_user_bin[i]) return fail(TXQ_ERR_ARG, "HIBF map %llu is null", (unsigned long long)i);
        off[i + 1] = off[i] + desc.ibf[i].bins;
    }
    std::vector<uint64_t> next(off[n]), tbu(off[n]);
    std::vector<int> level(n, -1);
    std::deque<uint64_t> q{0};
    level[0] = 0;
    std::vector<uint64_t> width(1, 1);
    while (!q.empty()) {
        const uint64_t i = q.front();
        q.pop_front();
        for (uint64_t b = 0; b < desc.ibf[i].bins; ++b) {
            const uint64_t ub = desc.tb_to_user_bin[i][b];
            uint64_t nx = desc.next_ibf_id[i][b];
            if (ub == TXQ_MERGED_BIN) {
                if (nx >= n || nx == i) return fail(TXQ_ERR_ARG, "IBF %llu bin %llu: bad child %llu", (unsigned long long)i, (unsigned long long)b, (unsigned long long)nx);
                if (level[nx] >= 0) return fail(TXQ_ERR_ARG, "IBF %llu has two parents: not a tree", (unsigned long long)nx);
                level[nx] = level[i] + 1;
                if ((size_t)level[nx] >= width.size()) width.push_back(0);
                ++width[level[nx]];
                q.push_back(nx);
            } else {
                if (ub >= desc.user_bins) return fail(TXQ_ERR_ARG, "IBF %llu bin %llu: user bin %llu out of range", (unsigned long long)i, (unsigned long long)b, (unsigned long long)ub);
                nx = 0;
            }
            next[off[i] + b] = nx;
            tbu[off[i] + b] = ub;
        }
    }
    for (uint64_t i = 0; i < n; ++i)
        if (level[i] < 0) return fail(TXQ_ERR_ARG, "IBF %llu is unreachable from the root", (unsigned long long)i);
    ix.depth = (uint32_t)width.size();
    ix.hibf_total_tbs = off[n];
    ix.max_level_width = 1;
    for (uint64_t w : width) if (w > ix.max_level_width) ix.max_level_width = w;

    ix.ibf.reserve(n);
    ix.max_stride = 1;
    for (uint64_t i = 0; i < n; ++i) {
        IbfDev f;
        uint64_t bytes;
        // every IBF of the tree is kept whole; only the user-bin mask columns are sharded
        if (int rc = alloc_ibf(desc.ibf[i], 0, desc.ibf[i].bin_words, &f, &bytes)) return rc;
        {   // identity-mapped leaf?
            const uint64_t base = tbu[off[i]];
            bool ident = base != TXQ_MERGED_BIN && base % 64 == 0 && (base >> 6) + desc.ibf[i].bin_words < kNoIdent;
            for (uint64_t b = 0; ident && b < desc.ibf[i].bins; ++b) ident = tbu[off[i] + b] == base + b;
            if (ident) f.ident_word = (uint32_t)(base >> 6);
        }
        ix.ibf.push_back(f);
        ix.device_bytes += bytes;
        if (f.stride > ix.max_stride) ix.max_stride = f.stride;
    }
    TXQ_HIP(hipMalloc((void**)&ix.d_ibf, n * sizeof(IbfDev)));
    TXQ_HIP(hipMalloc((void**)&ix.d_next, (off[n] ? off[n] : 1) * 8));
    TXQ_HIP(hipMalloc((void**)&ix.d_tb_user, (off[n] ? off[n] : 1) * 8));
    TXQ_HIP(hipMalloc((void**)&ix.d_map_off, n * 8));
    // merged-bin bitmasks, one 64-bit word per row word of every IBF
    std::vector<uint64_t> moff(n + 1, 0);
    // (padded to 4 words per IBF: the fused kernel reads them as two 16-byte loads)
    for (uint64_t i = 0; i < n; ++i) moff[i + 1] = moff[i] + ((desc.ibf[i].bin_words + 3) & ~(uint64_t)3);
    std::vector<uint64_t> merged(moff[n], 0);
    for (uint64_t i = 0; i < n; ++i)
        for (uint64_t b = 0; b < desc.ibf[i].bins; ++b)
            if (desc.tb_to_user_bin[i][b] == TXQ_MERGED_BIN) merged[moff[i] + (b >> 6)] |= 1ULL << (b & 63);
    // Sub-tree pruning for column shards: a merged bin is only descended into when its sub-tree
    // holds a user bin whose mask word belongs to this shard (span[i] = mask-word range under IBF i).
    std::vector<std::pair<uint64_t, uint64_t>> span(n, {UINT64_MAX, 0});
    {
        std::vector<uint64_t> order;  // parents before children (BFS order from the tree walk above)
        order.reserve(n);
        std::deque<uint64_t> bfs{0};
        while (!bfs.empty()) {
            const uint64_t i = bfs.front();
            bfs.pop_front();
            order.push_back(i);
            for (uint64_t b = 0; b < desc.ibf[i].bins; ++b)
                if (tbu[off[i] + b] == TXQ_MERGED_BIN) bfs.push_back(next[off[i] + b]);
        }
        for (size_t at = order.size(); at-- > 0;) {
            const uint64_t i = order[at];
            for (uint64_t b = 0; b < desc.ibf[i].bins; ++b) {
                const uint64_t ub = tbu[off[i] + b];
                std::pair<uint64_t, uint64_t> r = ub == TXQ_MERGED_BIN ? span[next[off[i] + b]] : std::make_pair(ub >> 6, ub >> 6);
                if (r.first < span[i].first) span[i].first = r.first;
                if (r.first != UINT64_MAX && r.second > span[i].second) span[i].second = r.second;
            }
        }
    }
    std::vector<uint64_t> descend(moff[n], 0);
    const uint64_t shard_lo = ix.shard_word0, shard_hi = ix.shard_word0 + ix.shard_words;  // [lo, hi)
    for (uint64_t i = 0; i < n; ++i)
        for (uint64_t b = 0; b < desc.ibf[i].bins; ++b) {
            if (tbu[off[i] + b] != TXQ_MERGED_BIN) continue;
            const auto& r = span[next[off[i] + b]];
            if (r.first != UINT64_MAX && r.first < shard_hi && r.second >= shard_lo) descend[moff[i] + (b >> 6)] |= 1ULL << (b & 63);
        }
    TXQ_HIP(hipMalloc((void**)&ix.d_descend, (moff[n] ? moff[n] : 1) * 8));
    TXQ_HIP(hipMemcpy(ix.d_descend, descend.data(), moff[n] * 8, hipMemcpyHostToDevice));
    TXQ_HIP(hipMalloc((void**)&ix.d_merged, (moff[n] ? moff[n] : 1) * 8));
    TXQ_HIP(hipMalloc((void**)&ix.d_merged_off, n * 8));
    TXQ_HIP(hipMemcpy(ix.d_merged, merged.data(), moff[n] * 8, hipMemcpyHostToDevice));
    TXQ_HIP(hipMemcpy(ix.d_merged_off, moff.data(), n * 8, hipMemcpyHostToDevice));
    TXQ_HIP(hipMemcpy(ix.d_ibf, ix.ibf.data(), n * sizeof(IbfDev), hipMemcpyHostToDevice));
    TXQ_HIP(hipMemcpy(ix.d_next, next.data(), off[n] * 8, hipMemcpyHostToDevice));
    TXQ_HIP(hipMemcpy(ix.d_tb_user, tbu.data(), off[n] * 8, hipMemcpyHostToDevice));
    TXQ_HIP(hipMemcpy(ix.d_map_off, off.data(), n * 8, hipMemcpyHostToDevice));
    ix.device_bytes += n * sizeof(IbfDev) + off[n] * 16 + n * 8;
    // node records by technical bin for the fused kernel (skipped for trees too large for it)
    bool compact = off[n] < kRootEntry && moff[n] < 0xFFFFFFFFull && off[n] <= ((size_t)256 << 20) / sizeof(HibfNode);
    for (const IbfDev& f : ix.ibf) compact = compact && !(f.bin_size >> 32) && f.stride < (1u << 20) && f.hash_shift < 64 && f.hash_funs < 8;
    std::vector<HibfNode> nodes;
    if (compact) {
        nodes.resize(off[n] + 1);
        std::memset(nodes.data(), 0, nodes.size() * sizeof(HibfNode));
        auto node_of = [&](uint64_t i) {
            const IbfDev& f = ix.ibf[i];
            bool has_merged = false;
            for (uint64_t w = moff[i]; w < moff[i + 1]; ++w) has_merged = has_merged || merged[w] != 0;
            HibfNode nd{};
            nd.words = (uint64_t)(uintptr_t)f.words;
            nd.bin_size = (uint32_t)f.bin_size;
            nd.packed = f.stride | (f.hash_shift << 20) | (f.hash_funs << 26) | ((uint32_t)has_merged << 29);
            nd.off = (uint32_t)off[i];
            nd.moff = (uint32_t)moff[i];
            nd.ident_word = f.ident_word;
            nd.bins = f.bins;
            return nd;
        };
        for (uint64_t i = 0; i < n; ++i)
            for (uint64_t b = 0; b < desc.ibf[i].bins; ++b)
                if (tbu[off[i] + b] == TXQ_MERGED_BIN) nodes[off[i] + b] = node_of(next[off[i] + b]);
        nodes[off[n]] = node_of(0);
        TXQ_HIP(hipMalloc((void**)&ix.d_nodes, nodes.size() * sizeof(HibfNode)));
        TXQ_HIP(hipMemcpy(ix.d_nodes, nodes.data(), nodes.size() * sizeof(HibfNode), hipMemcpyHostToDevice));
        ix.device_bytes += nodes.size() * sizeof(HibfNode);
    }

    // Regular two-level tree?  (root: only merged bins; every child: a leaf whose technical bins are an aligned run of
    // user bins, all children of one power-of-two row width, tiling the mask columns in order; this shard's
    // column range starts and ends on child boundaries)  -> ChildRec table for the child-stationary descent.
    if (compact && ix.depth == 2 && n >= 2 && desc.ibf[0].bins < (1u << 20)) {
        const uint32_t wpr = (uint32_t)desc.ibf[1].bin_words;
        bool regular = wpr >= 1 && (wpr & (wpr - 1)) == 0 && wpr <= 128;
        std::vector<uint64_t> by_column(n - 1, UINT64_MAX), root_tb(n, 0);
        for (uint64_t b = 0; regular && b < desc.ibf[0].bins; ++b) {
            if (tbu[off[0] + b] != TXQ_MERGED_BIN) { regular = false; break; }
            root_tb[next[off[0] + b]] = b;
        }
        for (uint64_t i = 1; regular && i < n; ++i) {
            const IbfDev& f = ix.ibf[i];
            regular = f.ident_word != kNoIdent && desc.ibf[i].bin_words == wpr && f.stride == wpr && f.ident_word % wpr == 0 &&
                      f.ident_word / wpr < n - 1 && by_column[f.ident_word / wpr] == UINT64_MAX && f.hash_funs <= 5;
            if (regular) by_column[f.ident_word / wpr] = i;
        }
        regular = regular && ix.mask_words == (uint64_t)wpr * (n - 1) && ix.shard_word0 % wpr == 0 && ix.shard_words % wpr == 0 && ix.shard_words > 0;
        if (regular) {
            std::vector<ChildRec> recs;
            uint64_t bytes = 0;
            for (uint64_t c = ix.shard_word0 / wpr; c < (ix.shard_word0 + ix.shard_words) / wpr; ++c) {
                const IbfDev& f = ix.ibf[by_column[c]];
                recs.push_back(ChildRec{(uint64_t)(uintptr_t)f.words, (uint32_t)f.bin_size, f.hash_shift | (f.hash_funs << 8) | ((uint32_t)root_tb[by_column[c]] << 12)});
                bytes += f.bin_size * (uint64_t)f.stride * 8;
            }
            TXQ_HIP(hipMalloc((void**)&ix.d_children, recs.size() * sizeof(ChildRec)));
            TXQ_HIP(hipMemcpy(ix.d_children, recs.data(), recs.size() * sizeof(ChildRec), hipMemcpyHostToDevice));
            ix.n_children = (uint32_t)recs.size();
            ix.children_uniform = true;
            for (const ChildRec& c : recs) ix.children_uniform = ix.children_uniform && c.bin_size == recs[0].bin_size && (c.packed & 0xFFFu) == (recs[0].packed & 0xFFFu);
            ix.child_row_words = wpr;
            ix.children_bytes = bytes;
            const IbfDev& root = ix.ibf[0];
            ix.root_node = HibfNode{};
            ix.root_node.words = (uint64_t)(uintptr_t)root.words;
            ix.root_node.bin_size = (uint32_t)root.bin_size;
            ix.root_node.packed = root.stride | (root.hash_shift << 20) | (root.hash_funs << 26);
            ix.root_node.bins = root.bins;
            ix.tree_hash_max = 1;
            for (const IbfDev& f : ix.ibf) ix.tree_hash_max = std::max(ix.tree_hash_max, f.hash_funs);
            if (ix.children_uniform && root.bins <= 64 && ix.shard_words <= 32 && knobs().hibf_interleave) {  // (TXQ_HIBF_INTERLEAVE=0: never)
                const IbfDev& c0 = ix.ibf[by_column[ix.shard_word0 / wpr]];
                IbfDev f{};
                f.bin_size = c0.bin_size;
                f.hash_shift = c0.hash_shift;
                f.hash_funs = c0.hash_funs;
                f.shard_words = (uint32_t)ix.shard_words;
                f.stride = f.shard_words <= 1 ? 1u : ((f.shard_words + 1u) & ~1u);
                f.bins = (uint32_t)ix.shard_words * 64u;
                f.ident_word = kNoIdent;
                const size_t nbytes = (size_t)f.bin_size * f.stride * 8;
                TXQ_HIP(hipMalloc((void**)&f.words, nbytes));
                ix.interleaved = f;  // released with the index from here on
                TXQ_HIP(hipMemset(f.words, 0, nbytes));
                const uint64_t total = f.bin_size * ix.shard_words;
                const unsigned grid = (unsigned)std::min<uint64_t>((total + 255) / 256, 256 * 64);
                interleave_children_kernel<<<grid ? grid : 1, 256, 0, nullptr>>>((const ChildRec*)ix.d_children, ix.n_children, wpr, f.bin_size, f.words, f.stride);
                TXQ_HIP(hipGetLastError());
                TXQ_HIP(hipDeviceSynchronize());
                ix.device_bytes += nbytes;
            }
            ix.device_bytes += recs.size() * sizeof(ChildRec);
        }
    }
    // any other tree: sessions work in layout order
    if (!ix.d_children)
        if (int rc = build_layout_order(ix, desc, level, next, tbu, off, compact ? &nodes : nullptr)) return rc;
    return TXQ_OK;
}

template <int G>
static hipError_t launch_level(unsigned grid, hipStream_t s, HibfView t, const uint64_t* kmers, const WorkItem* in,
                               const uint32_t* in_count, uint32_t n0, WorkItem* out, uint32_t* out_count, uint32_t cap,
                               uint32_t* overflow, uint64_t* masks, uint32_t w_out, uint32_t word0, uint32_t w_iters) {
    hibf_level_kernel<G><<<grid, 256, 0, s>>>(t, kmers, in, in_count, n0, out, out_count, cap, overflow, masks, w_out, word0, w_iters);
    return hipGetLastError();
}

// fused descent (hibf_fused_kernel) when a k-mer's row and IBF stack fit the LDS; returns false if not
// (TXQ_HIBF_LEVELS=1 forces the level-synchronous path, for A/B runs and for testing both)
static bool hibf_probe_fused(Index& ix, const Knobs& kn, const uint64_t* d_kmers, size_t n, uint64_t* d_masks, uint64_t* d_alive, hipStream_t s, int* rc) {
    const uint32_t w_out = (uint32_t)ix.shard_words;
    const uint32_t stack_cap = (uint32_t)ix.ibf.size();
    const uint32_t stack_lds = fused_stack_lds(kn, stack_cap, w_out);
    const size_t wave_words = (size_t)w_out + stack_lds / 2;
    const size_t wave_bytes = wave_words * 8;
    const size_t lds_budget = 64u << 10;
    if (!w_out || wave_bytes > lds_budget || !ix.d_nodes || kn.hibf_levels) return false;
    uint32_t h_max = 1;
    for (const IbfDev& f : ix.ibf) if (f.hash_funs > h_max) h_max = f.hash_funs;
    const HibfView t{ix.d_ibf, ix.d_next, ix.d_tb_user, ix.d_map_off, ix.d_merged, ix.d_descend, ix.d_merged_off, (const HibfNode*)ix.d_nodes, (uint32_t)ix.hibf_total_tbs};
    *rc = TXQ_OK;
    // regular two-level trees: the children stay put in L2, the k-mers stream past (TXQ_HIBF_STATIONARY=0: A/B against the kernels below)
    {
        // (narrow masks, <= 16 words, are better off with one lane per k-mer: hibf_small_kernel)
        if (ix.d_children && ix.n_children && ix.child_row_words >= 2 && w_out > 16 && kn.hibf_stationary) {
            const uint32_t wpr = ix.child_row_words, lpc = wpr / 2, cps = 64 / lpc;
            const uint32_t n_steps = (ix.n_children + cps - 1) / cps;
            // a group = whole wave steps whose children fit ~2 MB (half an XCD's L2); at least 8 groups when there are 8 steps
            const uint64_t per_step = ix.children_bytes / n_steps + 1;
            uint32_t spg = (uint32_t)std::max<uint64_t>(1, ((uint64_t)2 << 20) / per_step);
            if (n_steps >= 8 && (n_steps + spg - 1) / spg < 8) spg = n_steps / 8;
            if (kn.hibf_steps_per_group) spg = (uint32_t)kn.hibf_steps_per_group;
            const uint32_t n_groups = (n_steps + spg - 1) / spg;
            const uint32_t gpp = n_groups < 8 ? n_groups : 8;
            const uint32_t phases = (n_groups + gpp - 1) / gpp;
            uint32_t tile = 2048;
            if (kn.hibf_tile) tile = (uint32_t)std::max(64, kn.hibf_tile);
            const size_t n_tiles = (n + tile - 1) / tile;
            if ((size_t)phases * n_tiles * gpp < ((size_t)1 << 31) && n_tiles < ((size_t)1 << 31)) {
                const IbfDev& root = ix.ibf[0];
                const uint32_t cm_words = root.stride;
                if (int e = ensure((void**)&ix.scratch_cm, &ix.cap_cm, n * (size_t)cm_words * 8)) { *rc = e; return true; }
                HibfNode rn{};
                rn.words = (uint64_t)(uintptr_t)root.words;
                rn.bin_size = (uint32_t)root.bin_size;
                rn.packed = root.stride | (root.hash_shift << 20) | (root.hash_funs << 26);
                rn.bins = root.bins;
                size_t rb = (n + 255) / 256;
                if (rb > 256 * 32) rb = 256 * 32;
                const bool uniform = ix.children_uniform && !kn.hibf_lane_hash;  // (A/B: per-lane hashing on a uniform tree)
                uint32_t* d_child_rows = nullptr;
                const IbfDev& c0 = ix.ibf[1];  // uniform: every child looks like this one
                if (uniform) {
                    if (int e = ensure((void**)&ix.scratch_crows, &ix.cap_crows, n * (size_t)c0.hash_funs * 4)) { *rc = e; return true; }
                    d_child_rows = ix.scratch_crows;
                    h_max = c0.hash_funs;
                }
                hibf_root_kernel<<<(unsigned)rb, 256, 0, s>>>(rn, d_kmers, n, ix.scratch_cm, cm_words, d_child_rows, (uint32_t)c0.bin_size, c0.hash_shift, c0.hash_funs);
                if (d_alive) {
                    hipError_t e = hipMemsetAsync(d_alive, 0, ((n + 63) / 64) * 8, s);
                    if (e != hipSuccess) { *rc = fail_hip(e, "hipMemsetAsync(alive)"); return true; }
                }
                const unsigned grid = (unsigned)((size_t)phases * n_tiles * gpp);
                const int unroll = kn.hibf_unroll, store_flavour = kn.hibf_store;
#define TXQ_CHILDREN(U, UNI) hibf_children_kernel<U, UNI><<<grid, 256, 0, s>>>((const ChildRec*)ix.d_children, ix.n_children, lpc, d_kmers, n, ix.scratch_cm, \
                                                                                cm_words, d_masks, w_out, n_steps, spg, gpp, (uint32_t)n_tiles, tile, h_max, d_alive, store_flavour, d_child_rows)
                if (uniform) {
                    if (unroll == 1) TXQ_CHILDREN(1, true);
                    else if (unroll == 2) TXQ_CHILDREN(2, true);
                    else if (unroll == 3) TXQ_CHILDREN(3, true);
                    else if (unroll == 8) TXQ_CHILDREN(8, true);
                    else TXQ_CHILDREN(4, true);
                } else {
                    if (unroll == 1) TXQ_CHILDREN(1, false);
                    else if (unroll == 2) TXQ_CHILDREN(2, false);
                    else if (unroll == 8) TXQ_CHILDREN(8, false);
                    else TXQ_CHILDREN(4, false);
                }
#undef TXQ_CHILDREN
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) *rc = fail_hip(e, "hibf child-stationary kernel launch");
                return true;
            }
        }
    }
    // small trees: one lane per k-mer (TXQ_HIBF_SMALL=0 keeps them on the wave-per-k-mer kernel, for A/B runs)
    if (ix.max_stride <= 4 && ix.ibf.size() <= kSmallStack && ix.hibf_total_tbs < 0xFFFF && w_out <= 16 && kn.hibf_small) {
        const size_t lds_bytes = (size_t)w_out * kSmallPitch * 8 + (size_t)kSmallStack * kSmallThreads * 2;
        size_t blocks = (n + kSmallThreads - 1) / kSmallThreads;
        if (blocks > 256 * 8) blocks = 256 * 8;
        hibf_small_kernel<<<(unsigned)blocks, kSmallThreads, lds_bytes, s>>>(t, d_kmers, n, d_masks, w_out, (uint32_t)ix.shard_word0, h_max, d_alive);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) *rc = fail_hip(e, "hibf small-tree kernel launch");
        return true;
    }
    const unsigned waves = fused_waves_per_block(wave_bytes);
    // 64 waves per CU are launched: with an 8 KiB row the LDS keeps 14 of them resident and the rest queue up,
    // with the short rows of a column shard more are resident and the finer grain is worth 14 % (TXQ_HIBF_WAVES overrides)
    size_t want_waves = (size_t)256 * 64;
    if (kn.hibf_waves > 0) want_waves = (size_t)kn.hibf_waves;
    const size_t total_waves = n < want_waves ? n : want_waves;
    const unsigned grid = (unsigned)((total_waves + waves - 1) / waves);
    const uint32_t quads = (ix.max_stride + 3) / 4;  // a lane owns four row words
    int g = 1;
    while (g < 64 && (uint32_t)g < quads) g <<= 1;
    const uint32_t w_iters = (quads + (uint32_t)g - 1) / (uint32_t)g;
    if (d_alive) {
        hipError_t e = hipMemsetAsync(d_alive, 0, ((n + 63) / 64) * 8, s);
        if (e != hipSuccess) { *rc = fail_hip(e, "hipMemsetAsync(alive)"); return true; }
    }
    hipError_t e;
#define TXQ_FUSED(G) e = launch_fused<G>(grid, waves * 64, wave_bytes * waves, s, t, d_kmers, n, d_masks, w_out, (uint32_t)ix.shard_word0, w_iters, \
                                         stack_cap, (uint32_t)wave_words, h_max, d_alive, stack_lds, w_out)
    switch (g) {
        case 1: TXQ_FUSED(1); break;
        case 2: TXQ_FUSED(2); break;
        case 4: TXQ_FUSED(4); break;
        case 8: TXQ_FUSED(8); break;
        case 16: TXQ_FUSED(16); break;
        case 32: TXQ_FUSED(32); break;
        default: TXQ_FUSED(64); break;
    }
#undef TXQ_FUSED
    if (e != hipSuccess) *rc = fail_hip(e, "hibf fused kernel launch");
    return true;
}

void preload_hibf_kernels() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&hibf_clamp_kernel));
    (void)hipGetLastError();
}

int hibf_probe(Index& ix, const Knobs& kn, const uint64_t* d_kmers, size_t n, uint64_t* d_masks, uint64_t* d_alive, hipStream_t s) {
    const uint32_t w_out = (uint32_t)ix.shard_words;
    if (n == 0) return TXQ_OK;
    if (ix.probes_interleaved(kn)) {
        // a small regular tree of uniform children: its interleaved children are probed like a flat IBF (one row segment per
        // hash function), the root's word clears the words of the children the k-mer cannot be in (txq_probe.hip TreeRoot)
        uint32_t wpr_log2 = 0;
        while ((1u << wpr_log2) < ix.child_row_words) ++wpr_log2;
        hipError_t e = launch_probe_interleaved(ix.interleaved, ix.root_node, ix.d_children, wpr_log2, d_kmers, n, d_masks, d_alive, s);
        if (e != hipSuccess) return fail_hip(e, "interleaved probe launch");
        return TXQ_OK;
    }
    {
        int rc = TXQ_OK;
        if (hibf_probe_fused(ix, kn, d_kmers, n, d_masks, d_alive, s, &rc)) return rc;
    }
    if (w_out) TXQ_HIP(hipMemsetAsync(d_masks, 0, n * w_out * 8, s));
    // Frontier bound: a (k-mer, IBF) pair occurs at most once, so level l holds at most
    // chunk * (#IBFs on level l) items.  Choose the chunk so that this always fits.
    const size_t cap_items = (size_t)1 << 24;  // 128 MiB per frontier buffer
    size_t chunk = cap_items / ix.max_level_width;
    if (chunk == 0) chunk = 1;
    if (chunk > n) chunk = n;
    const size_t cap = chunk * ix.max_level_width;
    if (ix.depth > 1) {
        for (int i = 0; i < 2; ++i)
            if (int rc = ensure((void**)&ix.frontier[i], &ix.cap_frontier[i], cap * sizeof(WorkItem))) return rc;
    }
    if (int rc = ensure((void**)&ix.d_counts, &ix.cap_counts, ((size_t)ix.depth + 2) * 4)) return rc;
    uint32_t* overflow = ix.d_counts + ix.depth + 1;
    const HibfView t{ix.d_ibf, ix.d_next, ix.d_tb_user, ix.d_map_off, ix.d_merged, ix.d_descend, ix.d_merged_off, (const HibfNode*)ix.d_nodes, (uint32_t)ix.hibf_total_tbs};
    int g = 1;
    while (g < 64 && (uint32_t)g < ix.max_stride) g <<= 1;
    const uint32_t w_iters = (ix.max_stride + (uint32_t)g - 1) / (uint32_t)g;

    for (size_t off = 0; off < n; off += chunk) {
        const size_t m = n - off < chunk ? n - off : chunk;
        TXQ_HIP(hipMemsetAsync(ix.d_counts, 0, ((size_t)ix.depth + 2) * 4, s));
        for (uint32_t lvl = 0; lvl < ix.depth; ++lvl) {
            const WorkItem* in = lvl ? ix.frontier[(lvl - 1) & 1] : nullptr;
            const uint32_t* in_count = lvl ? ix.d_counts + (lvl - 1) : nullptr;
            WorkItem* out = ix.frontier[lvl & 1];
            // level 0 is sized by the batch; deeper levels are grid-stride over an unknown count
            size_t groups = lvl ? (size_t)2048 * 256 / g : m;
            size_t blocks = (groups * g + 255) / 256;
            if (blocks > 2048) blocks = 2048;
            if (blocks == 0) blocks = 1;
            hipError_t e;
#define TXQ_LVL(G) e = launch_level<G>((unsigned)blocks, s, t, d_kmers + off, in, in_count, (uint32_t)m, out, ix.d_counts + lvl, \
                                       (uint32_t)(ix.depth > 1 ? cap : 0), overflow, d_masks + off * w_out, w_out, (uint32_t)ix.shard_word0, w_iters)
            switch (g) {
                case 1: TXQ_LVL(1); break;
                case 2: TXQ_LVL(2); break;
                case 4: TXQ_LVL(4); break;
                case 8: TXQ_LVL(8); break;
                case 16: TXQ_LVL(16); break;
                case 32: TXQ_LVL(32); break;
                default: TXQ_LVL(64); break;
            }
#undef TXQ_LVL
            if (e != hipSuccess) return fail_hip(e, "hibf level kernel launch");
            hibf_clamp_kernel<<<1, 64, 0, s>>>(ix.d_counts + lvl, (uint32_t)cap);
        }
    }
    if (d_alive) {
        size_t blocks = ((n + 63) / 64 * 64 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        mask_alive_kernel<<<(unsigned)blocks, 256, 0, s>>>(d_masks, n, w_out, d_alive);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail_hip(e, "mask_alive kernel launch");
    }
    return TXQ_OK;
}

}  // namespace txq

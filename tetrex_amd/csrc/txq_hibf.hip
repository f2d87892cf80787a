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
#include <deque>

namespace txq {

struct HibfView {
    const IbfDev* ibf;
    const uint64_t* next;
    const uint64_t* tb_user;
    const uint64_t* map_off;
    const uint64_t* merged;      // per IBF word: bit b set <=> technical bin 64w+b is a merged bin
    const uint64_t* merged_off;  // [n_ibf] offset of IBF i's words in `merged`
};

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
        for (uint32_t j = 0; j < 5; ++j) row[j] = j < f.hash_funs ? hash_row(v, kSeeds[j], f.hash_shift, f.bin_size) : 0;
        for (uint32_t wi = 0; wi < w_iters; ++wi) {
            const uint32_t w = wi * G + sub;
            uint64_t acc = 0;
            if (live && w < f.shard_words) {
                acc = ~0ULL;
#pragma unroll
                for (uint32_t j = 0; j < 5; ++j)
                    if (j < f.hash_funs) acc &= f.words[row[j] * f.stride + w];
            }
            uint64_t kids = acc ? (acc & t.merged[moff + w]) : 0;
            uint64_t hits = acc & ~kids;
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

#define TXQ_HIP(call)                                        \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return fail_hip(e_, #call);    \
    } while (0)

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
    for (uint64_t i = 0; i < n; ++i) moff[i + 1] = moff[i] + desc.ibf[i].bin_words;
    std::vector<uint64_t> merged(moff[n], 0);
    for (uint64_t i = 0; i < n; ++i)
        for (uint64_t b = 0; b < desc.ibf[i].bins; ++b)
            if (desc.tb_to_user_bin[i][b] == TXQ_MERGED_BIN) merged[moff[i] + (b >> 6)] |= 1ULL << (b & 63);
    TXQ_HIP(hipMalloc((void**)&ix.d_merged, (moff[n] ? moff[n] : 1) * 8));
    TXQ_HIP(hipMalloc((void**)&ix.d_merged_off, n * 8));
    TXQ_HIP(hipMemcpy(ix.d_merged, merged.data(), moff[n] * 8, hipMemcpyHostToDevice));
    TXQ_HIP(hipMemcpy(ix.d_merged_off, moff.data(), n * 8, hipMemcpyHostToDevice));
    TXQ_HIP(hipMemcpy(ix.d_ibf, ix.ibf.data(), n * sizeof(IbfDev), hipMemcpyHostToDevice));
    TXQ_HIP(hipMemcpy(ix.d_next, next.data(), off[n] * 8, hipMemcpyHostToDevice));
    TXQ_HIP(hipMemcpy(ix.d_tb_user, tbu.data(), off[n] * 8, hipMemcpyHostToDevice));
    TXQ_HIP(hipMemcpy(ix.d_map_off, off.data(), n * 8, hipMemcpyHostToDevice));
    ix.device_bytes += n * sizeof(IbfDev) + off[n] * 16 + n * 8;
    return TXQ_OK;
}

template <int G>
static hipError_t launch_level(unsigned grid, hipStream_t s, HibfView t, const uint64_t* kmers, const WorkItem* in,
                               const uint32_t* in_count, uint32_t n0, WorkItem* out, uint32_t* out_count, uint32_t cap,
                               uint32_t* overflow, uint64_t* masks, uint32_t w_out, uint32_t word0, uint32_t w_iters) {
    hibf_level_kernel<G><<<grid, 256, 0, s>>>(t, kmers, in, in_count, n0, out, out_count, cap, overflow, masks, w_out, word0, w_iters);
    return hipGetLastError();
}

int hibf_probe(Index& ix, const uint64_t* d_kmers, size_t n, uint64_t* d_masks, uint64_t* d_alive, hipStream_t s) {
    const uint32_t w_out = (uint32_t)ix.shard_words;
    if (n == 0) return TXQ_OK;
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
    const HibfView t{ix.d_ibf, ix.d_next, ix.d_tb_user, ix.d_map_off, ix.d_merged, ix.d_merged_off};
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

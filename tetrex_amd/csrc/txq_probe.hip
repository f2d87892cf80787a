// Flat-IBF kernels for gfx950: batched bulk_contains ("probe") and device-side emplace.
//
// probe: k-mer -> h row indices (seqan::hibf hash_and_fit) -> gather h bin-wide bit rows ->
// 64-bit AND -> per-bin hit mask.  Replaces seqan::hibf containment_agent::bulk_contains as
// called from the reference at include/index_ibf.h:146-150.
//
// Mapping to the machine (wave = 64 lanes):
//   * a wave owns a tile of 64 consecutive k-mers; lane l hashes k-mer l ONCE (h row indices),
//   * a row is moved by LPK lanes x 16 B (global_load_dwordx4); LPK = pow2 >= stride/2, so a
//     128-byte row (1024 bins) takes 8 lanes and a wave gathers 8 k-mers x h rows per step,
//   * the row indices travel from the hashing lane to the gathering lane group by ds_bpermute
//     (__shfl), no LDS allocation, no redundant 64-bit multiplies,
//   * two steps are in flight at a time (2*h independent 16-byte gathers per lane before the
//     first AND, 94 VGPRs = 5 waves/SIMD); output rows of one step are contiguous (coalesced 1-KiB stores),
//   * `alive` (mask != 0, the collector's path_.none() test) falls out of one __ballot per step.
// Algorithmic HBM bytes per probe: h*W*8 (rows) + W*8 (mask) + 8 (k-mer); W = shard_words.
#include "txq_internal.hpp"
#include <cstdlib>

namespace txq {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 ld16(const uint64_t* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16_stream(uint64_t* p, u32x4 v) {
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
}
__device__ __forceinline__ bool nonzero(u32x4 v) { return (v.x | v.y | v.z | v.w) != 0u; }

__device__ __forceinline__ void store_chunk(const IbfDev& f, uint64_t* masks, size_t kidx, uint32_t c, u32x4 acc) {
    uint64_t* dst = masks + kidx * f.shard_words + 2u * c;
    if (2u * c + 1u < f.shard_words) st16_stream(dst, acc);
    else __builtin_nontemporal_store(((uint64_t)acc.y << 32) | acc.x, dst);  // odd tail word
}

// Where the matrix is the interleaved children of a small regular HIBF (Index::interleaved: row r of every child side by
// side), the root's rows decide which children's words survive: the hashing lane gathers the root word of its k-mer
// (at most 64 merged bins) and hands it to the gathering lanes with the row indices.
struct NoRoot { static constexpr bool kActive = false; };
struct TreeRoot {
    static constexpr bool kActive = true;
    HibfNode root;
    const ChildRec* children;  // in mask-column order; packed >> 12 = the child's merged bin in the root
    uint32_t wpr_log2;         // log2(mask words per child)
};

// LPK lanes per k-mer, H hash functions, U steps in flight.  Requires bin_size < 2^32 and an even
// stride.  U*H independent 16-byte gathers per lane are issued before the first AND.
template <int LPK, int H, int U, bool NT, class ROOT = NoRoot>
__global__ __launch_bounds__(256) void probe_kernel(IbfDev f, const uint64_t* __restrict__ kmers, size_t n,
                                                    uint64_t* __restrict__ masks, uint64_t* __restrict__ alive, ROOT R = ROOT{}) {
    constexpr int KPS = 64 / LPK;          // k-mers per step
    constexpr int UU = U < LPK ? U : LPK;  // a tile has LPK steps
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPK, grp = lane / LPK;
    const uint32_t chunks = f.stride >> 1;
    const size_t n_tiles = (n + 63) >> 6;
    const size_t n_waves = (size_t)gridDim.x * (blockDim.x >> 6);
    for (size_t tile = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < n_tiles; tile += n_waves) {
        const size_t base = tile << 6;
        const size_t mine = base + lane;
        const uint64_t v = mine < n ? __builtin_nontemporal_load(kmers + mine) : 0;
        uint32_t row[H];
#pragma unroll
        for (int i = 0; i < H; ++i) row[i] = (uint32_t)hash_row(v, kSeeds[i], f.hash_shift, f.bin_size);
        uint32_t root_lo = ~0u, root_hi = ~0u;  // the root's verdict on my k-mer: bit b = it may be in the child behind merged bin b
        if constexpr (ROOT::kActive) {
            const uint64_t* rw = (const uint64_t*)R.root.words;
            uint64_t x = ~0ULL;
            for (uint32_t i = 0; i < R.root.hash_funs(); ++i)
                x &= rw[hash_row_seeded(v * kSeeds[i], R.root.hash_shift(), R.root.bin_size) * R.root.stride()];
            root_lo = (uint32_t)x;
            root_hi = (uint32_t)(x >> 32);
        }
        bool my_alive = false;
        for (int s = 0; s < LPK; s += UU) {
            uint32_t r[UU][H];
#pragma unroll
            for (int u = 0; u < UU; ++u)
#pragma unroll
                for (int i = 0; i < H; ++i) r[u][i] = __shfl(row[i], (s + u) * KPS + grp);
            uint64_t verdict[UU];
            if constexpr (ROOT::kActive) {
#pragma unroll
                for (int u = 0; u < UU; ++u)
                    verdict[u] = ((uint64_t)(uint32_t)__shfl((int)root_hi, (s + u) * KPS + grp) << 32) | (uint32_t)__shfl((int)root_lo, (s + u) * KPS + grp);
            }
            bool nz[UU];
#pragma unroll
            for (int u = 0; u < UU; ++u) nz[u] = false;
            for (uint32_t c = sub; c < chunks; c += LPK) {
                uint32_t bin0 = 0, bin1 = 0;  // merged bins of the children behind words 2c and 2c + 1
                if constexpr (ROOT::kActive) {
                    bin0 = R.children[(2u * c) >> R.wpr_log2].packed >> 12;
                    bin1 = 2u * c + 1u < f.shard_words ? R.children[(2u * c + 1u) >> R.wpr_log2].packed >> 12 : bin0;
                }
                u32x4 x[UU][H];
#pragma unroll
                for (int u = 0; u < UU; ++u)
#pragma unroll
                    for (int i = 0; i < H; ++i) {
                        const uint64_t* p = f.words + (size_t)r[u][i] * f.stride + 2u * c;
                        x[u][i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)) : ld16(p);
                    }
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    u32x4 acc = x[u][0];
#pragma unroll
                    for (int i = 1; i < H; ++i) acc &= x[u][i];
                    if constexpr (ROOT::kActive) {
                        const uint32_t m0 = (verdict[u] >> bin0) & 1ULL ? ~0u : 0u, m1 = (verdict[u] >> bin1) & 1ULL ? ~0u : 0u;
                        acc.x &= m0; acc.y &= m0; acc.z &= m1; acc.w &= m1;
                    }
                    const size_t kidx = base + (s + u) * KPS + grp;
                    if (kidx < n) store_chunk(f, masks, kidx, c, acc);
                    nz[u] |= nonzero(acc);
                }
            }
            if (alive) {
                const uint64_t gm = LPK == 64 ? ~0ULL : ((1ULL << LPK) - 1ULL);
                const int my_step = lane / KPS, my_grp = lane % KPS;
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    const uint64_t b = __ballot(nz[u]);
                    if (my_step == s + u) my_alive = ((b >> (my_grp * LPK)) & gm) != 0;
                }
            }
        }
        if (alive) {
            const uint64_t bits = __ballot(my_alive && mine < n);
            if (lane == 0) alive[tile] = bits;
        }
    }
}

// <= 64 bins in this shard: one 8-byte word per row, one lane per k-mer.
template <int H>
__global__ __launch_bounds__(256) void probe_w1_kernel(IbfDev f, const uint64_t* __restrict__ kmers, size_t n,
                                                       uint64_t* __restrict__ masks, uint64_t* __restrict__ alive) {
    const size_t stride_threads = (size_t)gridDim.x * blockDim.x;
    const size_t n_pad = (n + 63) & ~(size_t)63;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += stride_threads) {
        uint64_t acc = 0;
        if (i < n) {
            const uint64_t v = kmers[i];
            uint64_t w[H];
#pragma unroll
            for (int j = 0; j < H; ++j) w[j] = f.words[hash_row(v, kSeeds[j], f.hash_shift, f.bin_size) * f.stride];
            acc = w[0];
#pragma unroll
            for (int j = 1; j < H; ++j) acc &= w[j];
            masks[i] = acc;
        }
        if (alive) {
            const uint64_t bits = __ballot(acc != 0);
            if ((threadIdx.x & 63) == 0) alive[i >> 6] = bits;
        }
    }
}

// Fallback for rows >= 2^32 (row index needs 64 bits): one wave per k-mer, lanes sweep the row.
template <int H>
__global__ __launch_bounds__(256) void probe_bigrows_kernel(IbfDev f, const uint64_t* __restrict__ kmers, size_t n,
                                                            uint64_t* __restrict__ masks, uint64_t* __restrict__ alive) {
    const int lane = threadIdx.x & 63;
    const size_t n_waves = (size_t)gridDim.x * (blockDim.x >> 6);
    for (size_t k = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); k < n; k += n_waves) {
        const uint64_t v = kmers[k];
        uint64_t row[H];
#pragma unroll
        for (int j = 0; j < H; ++j) row[j] = hash_row(v, kSeeds[j], f.hash_shift, f.bin_size);
        bool nz = false;
        for (uint32_t w = lane; w < f.shard_words; w += 64) {
            uint64_t acc = f.words[row[0] * f.stride + w];
#pragma unroll
            for (int j = 1; j < H; ++j) acc &= f.words[row[j] * f.stride + w];
            masks[k * f.shard_words + w] = acc;
            nz |= acc != 0;
        }
        if (alive) {
            const bool any = __ballot(nz) != 0;
            if (lane == 0 && any) atomicOr((unsigned long long*)(alive + (k >> 6)), 1ULL << (k & 63));
        }
    }
}

// emplace: value i -> bin bins_of[i]; sets h bits with atomicOr (idempotent, order-free).
__global__ __launch_bounds__(256) void emplace_kernel(IbfDev f, const uint64_t* __restrict__ values,
                                                      const uint32_t* __restrict__ bins_of, size_t n) {
    const size_t stride_threads = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride_threads) {
        const uint32_t bin = bins_of[i];
        const uint32_t w = bin >> 6;
        if (bin >= f.bins || w < f.word0 || w >= f.word0 + f.shard_words) continue;
        const uint64_t v = values[i];
        const unsigned long long bit = 1ULL << (bin & 63);
        for (uint32_t j = 0; j < f.hash_funs; ++j) {
            const uint64_t r = hash_row(v, kSeeds[j], f.hash_shift, f.bin_size);
            atomicOr((unsigned long long*)(f.words + r * f.stride + (w - f.word0)), bit);
        }
    }
}

// ---- launchers --------------------------------------------------------------------------

static inline unsigned grid_for(size_t work_items, unsigned per_block) {
    size_t blocks = (work_items + per_block - 1) / per_block;
    // Up to 256 blocks per CU before the kernels start to grid-stride: letting the dispatcher hand out
    // short blocks balances slightly better than 8 long ones per CU (1024-bin index: +0.6 % cache-resident,
    // within the noise on matrices that miss the Infinity Cache).  TXQ_PROBE_BLOCKS_PER_CU is the A/B knob.
    const size_t per_cu = (size_t)knobs().probe_blocks_per_cu;
    const size_t cap = 256u * (per_cu ? per_cu : 1);
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}

template <int LPK>
static hipError_t launch_lpk(const IbfDev& f, const uint64_t* k, size_t n, uint64_t* m, uint64_t* a, hipStream_t s) {
    const unsigned grid = grid_for((n + 63) / 64, 4);
    // experiment knobs (round 1 tuning): steps in flight and non-temporal row loads
    // (profiles/r1_probe_variants_ab.txt: 1/2/4/8 steps in flight differ by <= 2 % — occupancy already
    // supplies the memory-level parallelism, and 4 steps cost 124 VGPRs = half the waves per SIMD;
    // non-temporal ROW loads cost 20 % on a cache-resident matrix and gain nothing on an 8 GB one.
    // Default: 2 steps in flight, 5 waves/SIMD (94 VGPRs), plain row loads.)
    const int unroll = knobs().probe_unroll;
    const bool nt = knobs().probe_nt;
    if (f.hash_funs == 3 && LPK == 8 && (unroll != 2 || nt)) {
        if (unroll == 1 && !nt) probe_kernel<LPK, 3, 1, false><<<grid, 256, 0, s>>>(f, k, n, m, a);
        else if (unroll == 4 && !nt) probe_kernel<LPK, 3, 4, false><<<grid, 256, 0, s>>>(f, k, n, m, a);
        else if (unroll == 8 && !nt) probe_kernel<LPK, 3, 8, false><<<grid, 256, 0, s>>>(f, k, n, m, a);
        else if (unroll == 2 && nt) probe_kernel<LPK, 3, 2, true><<<grid, 256, 0, s>>>(f, k, n, m, a);
        else if (unroll == 4 && nt) probe_kernel<LPK, 3, 4, true><<<grid, 256, 0, s>>>(f, k, n, m, a);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    switch (f.hash_funs) {
        case 1: probe_kernel<LPK, 1, 2, false><<<grid, 256, 0, s>>>(f, k, n, m, a); break;
        case 2: probe_kernel<LPK, 2, 2, false><<<grid, 256, 0, s>>>(f, k, n, m, a); break;
        case 3: probe_kernel<LPK, 3, 2, false><<<grid, 256, 0, s>>>(f, k, n, m, a); break;
        case 4: probe_kernel<LPK, 4, 2, false><<<grid, 256, 0, s>>>(f, k, n, m, a); break;
        case 5: probe_kernel<LPK, 5, 2, false><<<grid, 256, 0, s>>>(f, k, n, m, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

#define TXQ_H_SWITCH(KERNEL, GRID)                                         \
    switch (f.hash_funs) {                                                 \
        case 1: KERNEL<1><<<GRID, 256, 0, s>>>(f, k, n, m, a); break;      \
        case 2: KERNEL<2><<<GRID, 256, 0, s>>>(f, k, n, m, a); break;      \
        case 3: KERNEL<3><<<GRID, 256, 0, s>>>(f, k, n, m, a); break;      \
        case 4: KERNEL<4><<<GRID, 256, 0, s>>>(f, k, n, m, a); break;      \
        case 5: KERNEL<5><<<GRID, 256, 0, s>>>(f, k, n, m, a); break;      \
        default: return hipErrorInvalidValue;                              \
    }

void preload_probe_kernels() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&emplace_kernel));
    (void)hipGetLastError();
}

hipError_t launch_probe(const IbfDev& f, const uint64_t* k, size_t n, uint64_t* m, uint64_t* a, hipStream_t s) {
    if (n == 0 || f.shard_words == 0) return hipSuccess;
    if (f.bin_size >> 32) {
        if (a) {
            hipError_t e = hipMemsetAsync(a, 0, ((n + 63) / 64) * 8, s);
            if (e != hipSuccess) return e;
        }
        const unsigned grid = grid_for(n, 4);
        TXQ_H_SWITCH(probe_bigrows_kernel, grid);
        return hipGetLastError();
    }
    if (f.stride == 1) {
        const unsigned grid = grid_for((n + 63) & ~(size_t)63, 256);
        TXQ_H_SWITCH(probe_w1_kernel, grid);
        return hipGetLastError();
    }
    const uint32_t chunks = f.stride >> 1;
    if (chunks <= 1) return launch_lpk<1>(f, k, n, m, a, s);
    if (chunks <= 2) return launch_lpk<2>(f, k, n, m, a, s);
    if (chunks <= 4) return launch_lpk<4>(f, k, n, m, a, s);
    if (chunks <= 8) return launch_lpk<8>(f, k, n, m, a, s);
    if (chunks <= 16) return launch_lpk<16>(f, k, n, m, a, s);
    if (chunks <= 32) return launch_lpk<32>(f, k, n, m, a, s);
    return launch_lpk<64>(f, k, n, m, a, s);
}

// the interleaved children of a small regular HIBF (even stride, rows < 2^32, root of at most 64 merged bins)
template <int LPK>
static hipError_t launch_tree_lpk(const IbfDev& f, const TreeRoot& root, const uint64_t* k, size_t n, uint64_t* m, uint64_t* a, hipStream_t s) {
    const unsigned grid = grid_for((n + 63) / 64, 4);
    switch (f.hash_funs) {
        case 1: probe_kernel<LPK, 1, 2, false, TreeRoot><<<grid, 256, 0, s>>>(f, k, n, m, a, root); break;
        case 2: probe_kernel<LPK, 2, 2, false, TreeRoot><<<grid, 256, 0, s>>>(f, k, n, m, a, root); break;
        case 3: probe_kernel<LPK, 3, 2, false, TreeRoot><<<grid, 256, 0, s>>>(f, k, n, m, a, root); break;
        case 4: probe_kernel<LPK, 4, 2, false, TreeRoot><<<grid, 256, 0, s>>>(f, k, n, m, a, root); break;
        case 5: probe_kernel<LPK, 5, 2, false, TreeRoot><<<grid, 256, 0, s>>>(f, k, n, m, a, root); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_probe_interleaved(const IbfDev& f, const HibfNode& root, const void* children, uint32_t wpr_log2, const uint64_t* k, size_t n,
                                    uint64_t* m, uint64_t* a, hipStream_t s) {
    if (n == 0 || f.shard_words == 0) return hipSuccess;
    if ((f.bin_size >> 32) || f.stride < 2 || (f.stride & 1) || root.bins > 64) return hipErrorInvalidValue;
    const TreeRoot r{root, (const ChildRec*)children, wpr_log2};
    const uint32_t chunks = f.stride >> 1;
    if (chunks <= 1) return launch_tree_lpk<1>(f, r, k, n, m, a, s);
    if (chunks <= 2) return launch_tree_lpk<2>(f, r, k, n, m, a, s);
    if (chunks <= 4) return launch_tree_lpk<4>(f, r, k, n, m, a, s);
    if (chunks <= 8) return launch_tree_lpk<8>(f, r, k, n, m, a, s);
    if (chunks <= 16) return launch_tree_lpk<16>(f, r, k, n, m, a, s);
    return hipErrorInvalidValue;
}

hipError_t launch_emplace(const IbfDev& f, const uint64_t* values, const uint32_t* bins_of, size_t n, hipStream_t s) {
    if (n == 0 || f.shard_words == 0) return hipSuccess;
    emplace_kernel<<<grid_for(n, 256), 256, 0, s>>>(f, values, bins_of, n);
    return hipGetLastError();
}

}  // namespace txq

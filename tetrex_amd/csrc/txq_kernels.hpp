// Device-side data structures and kernels of the MI355X TetRex query engine (gfx950 only).
//
// HBM layout of a flat IBF shard ("column shard"):
//   words[r * stride + w], r < bin_size, w < shard_words; stride = shard_words rounded up to an
//   even number of 64-bit words (16-byte lane accesses) unless shard_words == 1.  For the
//   1024-bin configuration a row is exactly one 128-byte line.
// The on-disk matrix is row-major over ALL technical bins (include/txq.h: txq_ibf_desc); the
// re-layout to a contiguous per-rank column slice happens once at upload.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace txq {

// seqan::hibf::interleaved_bloom_filter::hash_seeds / hash_and_fit constants (restated; the
// library is absent from the reference tree — see include/txq.h and DESIGN.md §Oracle).
__device__ __constant__ const uint64_t kSeeds[5] = {
    13572355802537770549ULL, 13043817825332782213ULL, 10650232656628343401ULL,
    16499269484942379435ULL, 4893150838803335377ULL};
static constexpr uint64_t kGolden = 11400714819323198485ULL;

struct IbfDev {
    uint64_t* words;      // device pointer, [bin_size][stride]
    uint64_t bin_size;    // rows
    uint32_t hash_shift;  // countl_zero(bin_size)
    uint32_t hash_funs;   // 1..5
    uint32_t stride;      // words per row in HBM
    uint32_t shard_words; // mask words this shard owns (<= stride)
    uint32_t word0;       // first full-mask word owned by this shard
    uint32_t bins;        // technical bins in use (unsharded)
    // HIBF leaves whose technical bin b is user bin 64*ident_word + b (no merged bins): a hit word
    // of the row IS a word of the result mask.  ident_word == kNoIdent otherwise.
    uint32_t ident_word;
    uint32_t reserved;
};
static constexpr uint32_t kNoIdent = 0xFFFFFFFFu;

// floor(x * n / 2^64) for n < 2^32 (every IBF with fewer than 4 G rows): two 32-bit multiplies
__device__ __forceinline__ uint64_t fastrange32(uint64_t x, uint32_t n) {
    return ((uint64_t)(uint32_t)(x >> 32) * (uint64_t)n + (uint64_t)__umulhi((uint32_t)x, n)) >> 32;
}
__device__ __forceinline__ uint64_t fastrange(uint64_t x, uint64_t n) {
    return (n >> 32) ? __umul64hi(x, n) : fastrange32(x, (uint32_t)n);
}

#if defined(TXQ_EXPERIMENTS) && defined(TXQ_CHEAP_HASH)
// TIMING EXPERIMENT ONLY (wrong rows): three 32-bit multiplies instead of eleven — what do the kernels gain if hashing is free?
__device__ __forceinline__ uint64_t hash_row_seeded(uint64_t v, uint32_t, uint64_t bin_size) {
    return __umulhi(((uint32_t)v ^ (uint32_t)(v >> 32)) * 0x9E3779B1u, (uint32_t)bin_size);
}
__device__ __forceinline__ uint64_t hash_row_seeded32(uint64_t v, uint32_t, uint32_t bin_size) { return hash_row_seeded(v, 0, bin_size); }
__device__ __forceinline__ uint64_t hash_row(uint64_t v, uint64_t seed, uint32_t, uint64_t bin_size) {
    return hash_row_seeded((uint64_t)(((uint32_t)v ^ ((uint32_t)(v >> 32) * 0x85EBCA6Bu)) * (uint32_t)seed), 0, bin_size);
}
#else
// second half of hash_row for a value that is already multiplied by its seed
__device__ __forceinline__ uint64_t hash_row_seeded(uint64_t v, uint32_t shift, uint64_t bin_size) {
    v ^= v >> shift;
    v *= kGolden;
    return fastrange(v, bin_size);
}
// the same when the caller knows that bin_size < 2^32 (saves evaluating both fastrange variants
// where bin_size differs from lane to lane)
__device__ __forceinline__ uint64_t hash_row_seeded32(uint64_t v, uint32_t shift, uint32_t bin_size) {
    v ^= v >> shift;
    v *= kGolden;
    return fastrange32(v, bin_size);
}

// row index of `v` under hash function i: fastrange of the mixed hash onto [0, bin_size)
__device__ __forceinline__ uint64_t hash_row(uint64_t v, uint64_t seed, uint32_t shift, uint64_t bin_size) {
    return hash_row_seeded(v * seed, shift, bin_size);
}
#endif

}  // namespace txq

// Mask-DAG executor for gfx950: the bit algebra of OTFCollector::collect()
// (reference include/otf_collector.h:341-393) for a whole batch of queries in two launches.
//
//   phase 1  probe every DISTINCT k-mer of the batch once (flat IBF: txq_probe.hip gather/AND;
//            HIBF: txq_hibf.hip descent)  ->  M[n_kmers][W] in HBM.  This is the reference's
//            kmer_cache_ made batch-wide.
//   phase 2  one lane group per program walks its op list; every lane owns fixed mask-word
//            columns, so the whole program needs no barrier and no cross-lane traffic: bins are
//            independent in every operation of the collector.  Slot masks live in a per-program
//            HBM scratch arena that stays L2-resident (programs touch a few KiB each).
// Format of the op list: include/txq_program.h.
#include "txq_internal.hpp"
#include "../../include/txq_program.h"
#include <cstring>

namespace txq {

// G lanes per program (pow2 >= W, <= 64); lane `sub` owns words sub, sub+G, ...
template <int G>
__global__ __launch_bounds__(256) void exec_kernel(const txq_program* __restrict__ progs, const txq_op* __restrict__ ops,
                                                   const uint64_t* __restrict__ slot_off, uint32_t n_programs,
                                                   const uint64_t* __restrict__ M, uint64_t* __restrict__ slots,
                                                   uint64_t* __restrict__ final_masks, uint32_t W, uint64_t user_bins,
                                                   uint64_t word0) {
    const uint32_t sub = threadIdx.x % G;
    const size_t group = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const size_t n_groups = ((size_t)gridDim.x * blockDim.x) / G;
    for (size_t p = group; p < n_programs; p += n_groups) {
        const txq_program pr = progs[p];
        uint64_t* S = slots + slot_off[p];  // [n_slots][W]
        for (uint32_t w = sub; w < W; w += G) {
            // ONES = hit_vector(bin_count, true): bits of this shard's word that are real bins
            const uint64_t first_bin = (word0 + w) * 64;
            uint64_t ones = 0;
            if (first_bin < user_bins) ones = (user_bins - first_bin >= 64) ? ~0ULL : ((1ULL << (user_bins - first_bin)) - 1ULL);
            S[(size_t)TXQ_SLOT_ZERO * W + w] = 0;
            S[(size_t)TXQ_SLOT_ONES * W + w] = ones;
            S[(size_t)TXQ_SLOT_RESULT * W + w] = 0;
        }
        const txq_op* op = ops + pr.first_op;
        for (uint32_t i = 0; i < pr.n_ops; ++i) {
            const txq_op o = op[i];
            for (uint32_t w = sub; w < W; w += G) {
                uint64_t x = S[(size_t)o.a * W + w];
                if (o.kmer != TXQ_NO_KMER) x &= M[(size_t)o.kmer * W + w];
                x |= S[(size_t)o.b * W + w];
                S[(size_t)o.dst * W + w] = x;
            }
        }
        for (uint32_t w = sub; w < W; w += G) final_masks[p * W + w] = S[(size_t)TXQ_SLOT_RESULT * W + w];
    }
}

#define TXQ_HIP(call)                                        \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return fail_hip(e_, #call);    \
    } while (0)

// Host-side validation: nothing malformed may reach the GPU (a stray slot or k-mer index would
// be an out-of-bounds access there).
static int validate_blob(const unsigned char* blob, size_t bytes, size_t n_programs, const txq_blob_header** hdr_out) {
    if (bytes < sizeof(txq_blob_header)) return fail(TXQ_ERR_PROGRAM, "blob shorter than its header");
    if ((uintptr_t)blob % 8) return fail(TXQ_ERR_PROGRAM, "blob must be 8-byte aligned");
    const txq_blob_header* h = (const txq_blob_header*)blob;
    if (h->magic != TXQ_PROGRAM_MAGIC || h->version != TXQ_PROGRAM_VERSION) return fail(TXQ_ERR_PROGRAM, "bad blob magic/version");
    if (h->n_programs != n_programs) return fail(TXQ_ERR_PROGRAM, "blob holds %u programs, caller says %zu", h->n_programs, n_programs);
    auto in_range = [&](uint64_t off, uint64_t count, uint64_t elem) {
        return off % 8 == 0 && off <= bytes && count <= (bytes - off) / elem;
    };
    if (!in_range(h->kmers_offset, h->n_kmers, 8) || !in_range(h->programs_offset, h->n_programs, sizeof(txq_program)) ||
        !in_range(h->ops_offset, h->n_ops, sizeof(txq_op)))
        return fail(TXQ_ERR_PROGRAM, "blob table outside the blob");
    const txq_program* pr = (const txq_program*)(blob + h->programs_offset);
    const txq_op* ops = (const txq_op*)(blob + h->ops_offset);
    for (uint32_t p = 0; p < h->n_programs; ++p) {
        if (pr[p].n_slots < TXQ_SLOT_FIRST_FREE) return fail(TXQ_ERR_PROGRAM, "program %u: n_slots < 3", p);
        if (pr[p].first_op > h->n_ops || pr[p].n_ops > h->n_ops - pr[p].first_op) return fail(TXQ_ERR_PROGRAM, "program %u: ops out of range", p);
        for (uint32_t i = 0; i < pr[p].n_ops; ++i) {
            const txq_op& o = ops[pr[p].first_op + i];
            if (o.dst >= pr[p].n_slots || o.a >= pr[p].n_slots || o.b >= pr[p].n_slots)
                return fail(TXQ_ERR_PROGRAM, "program %u op %u: slot out of range", p, i);
            if (o.dst == TXQ_SLOT_ZERO || o.dst == TXQ_SLOT_ONES) return fail(TXQ_ERR_PROGRAM, "program %u op %u: writes a constant slot", p, i);
            if (o.kmer != TXQ_NO_KMER && o.kmer >= h->n_kmers) return fail(TXQ_ERR_PROGRAM, "program %u op %u: k-mer index out of range", p, i);
        }
    }
    *hdr_out = h;
    return TXQ_OK;
}

int run_programs(Index& ix, const void* blob_v, size_t bytes, size_t n_programs, uint64_t* d_final, hipStream_t s) {
    const unsigned char* blob = (const unsigned char*)blob_v;
    const txq_blob_header* h = nullptr;
    if (int rc = validate_blob(blob, bytes, n_programs, &h)) return rc;
    const uint32_t W = (uint32_t)ix.shard_words;
    if (n_programs == 0 || W == 0) return TXQ_OK;

    // slot arena offsets (in words) per program, appended to the device copy of the blob
    const txq_program* pr = (const txq_program*)(blob + h->programs_offset);
    std::vector<uint64_t> slot_off(n_programs);
    uint64_t total = 0;
    for (size_t p = 0; p < n_programs; ++p) { slot_off[p] = total; total += (uint64_t)pr[p].n_slots * W; }

    const size_t off_bytes = (bytes + 7) & ~(size_t)7;
    if (int rc = ensure((void**)&ix.scratch_blob, &ix.cap_blob, off_bytes + n_programs * 8)) return rc;
    if (int rc = ensure((void**)&ix.scratch_slots, &ix.cap_slots, total * 8)) return rc;
    const size_t nk = h->n_kmers;
    if (int rc = ensure((void**)&ix.scratch_masks, &ix.cap_masks, (nk ? nk : 1) * (size_t)W * 8)) return rc;
    // The staging copies below read pageable host memory; they are synchronous with respect to
    // the host buffer, so `blob` may be reused by the caller as soon as this function returns.
    TXQ_HIP(hipMemcpyAsync(ix.scratch_blob, blob, bytes, hipMemcpyHostToDevice, s));
    TXQ_HIP(hipMemcpyAsync(ix.scratch_blob + off_bytes, slot_off.data(), n_programs * 8, hipMemcpyHostToDevice, s));
    TXQ_HIP(hipStreamSynchronize(s));  // slot_off is a local; the blob copy must also have left host memory

    const uint64_t* d_kmers = (const uint64_t*)(ix.scratch_blob + h->kmers_offset);
    if (nk) {
        if (ix.is_hibf) {
            if (int rc = hibf_probe(ix, d_kmers, nk, ix.scratch_masks, nullptr, s)) return rc;
        } else {
            hipError_t e = launch_probe(ix.ibf[0], d_kmers, nk, ix.scratch_masks, nullptr, s);
            if (e != hipSuccess) return fail_hip(e, "probe kernel launch");
        }
    }
    int g = 1;
    while (g < 64 && (uint32_t)g < W) g <<= 1;
    size_t blocks = (n_programs * g + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    const txq_program* d_pr = (const txq_program*)(ix.scratch_blob + h->programs_offset);
    const txq_op* d_ops = (const txq_op*)(ix.scratch_blob + h->ops_offset);
    const uint64_t* d_off = (const uint64_t*)(ix.scratch_blob + off_bytes);
#define TXQ_EXEC(G) exec_kernel<G><<<(unsigned)blocks, 256, 0, s>>>(d_pr, d_ops, d_off, (uint32_t)n_programs, ix.scratch_masks, \
                                                                    ix.scratch_slots, d_final, W, ix.user_bins, ix.shard_word0)
    switch (g) {
        case 1: TXQ_EXEC(1); break;
        case 2: TXQ_EXEC(2); break;
        case 4: TXQ_EXEC(4); break;
        case 8: TXQ_EXEC(8); break;
        case 16: TXQ_EXEC(16); break;
        case 32: TXQ_EXEC(32); break;
        default: TXQ_EXEC(64); break;
    }
#undef TXQ_EXEC
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "exec kernel launch");
    return TXQ_OK;
}

}  // namespace txq

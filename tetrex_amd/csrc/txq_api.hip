// C-ABI (include/txq.h) of the MI355X TetRex query engine: index residency in HBM and the
// entry points the reference's seam would bind.  No CPU fallback: without a GPU every call
// that needs one fails with TXQ_ERR_STATE.
#include "../../include/txq.h"
#include "txq_internal.hpp"

#include <algorithm>
#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

namespace txq {

static thread_local std::string g_err;
static std::vector<int> g_devices;  // txq_init: shard r of an index lives on g_devices[r % size]

static std::mutex g_spare_mutex;
static std::map<int, std::vector<hipStream_t>> g_spare_streams;  // per device: streams created at txq_init for the first sessions
hipStream_t take_spare_stream(int device) {
    std::lock_guard<std::mutex> lk(g_spare_mutex);
    auto it = g_spare_streams.find(device);
    if (it == g_spare_streams.end() || it->second.empty()) return nullptr;
    hipStream_t st = it->second.back();
    it->second.pop_back();
    return st;
}

static Knobs g_knobs;
static std::mutex g_knobs_mutex;
Knobs knobs() {
    std::lock_guard<std::mutex> lock(g_knobs_mutex);
    return g_knobs;
}
void read_knobs() {
    auto flag = [](const char* name) { return std::getenv(name) != nullptr; };
    auto is = [](const char* name, char c) { const char* e = std::getenv(name); return e && e[0] == c; };
    auto num = [](const char* name, long long otherwise) { const char* e = std::getenv(name); return e && *e ? std::atoll(e) : otherwise; };
    Knobs k;
    k.trace = flag("TXQ_TRACE");
    k.trace_stages = flag("TXQ_TRACE_STAGES");
    k.trace_sync = flag("TXQ_TRACE_SYNC");
    k.dense_tree = (int)num("TXQ_DENSE_TREE", -1);
    k.dense_unroll = (int)num("TXQ_DENSE_UNROLL", 3);
    k.dense_slices = (int)std::max(1LL, num("TXQ_DENSE_SLICES", 2));
    k.dense_tile_rounds = (int)std::max(1LL, num("TXQ_DENSE_TILE_ROUNDS", 4));
    k.dense_nt = (int)num("TXQ_DENSE_NT", 0) & 3;
    k.fuse_units = !is("TXQ_FUSE_UNITS", '0');
    k.one_stream = flag("TXQ_ONE_STREAM");
    k.sparse_steps = !is("TXQ_SPARSE_STEPS", '0');
    k.sparse_unroll = (int)num("TXQ_SPARSE_UNROLL", 3);
    k.sparse_units = (int)std::min(1536LL, std::max(64LL, num("TXQ_SPARSE_UNITS", 512)));
    k.kmer_table_mb = std::max(0LL, num("TXQ_KMER_TABLE_MB", 512));
    k.kmer_table_min = std::max(1LL, num("TXQ_KMER_TABLE_MIN", 16));
    k.hibf_interleave = !is("TXQ_HIBF_INTERLEAVE", '0');
    k.hibf_interleave_probe = !is("TXQ_HIBF_INTERLEAVE_PROBE", '0');
    k.hibf_levels = is("TXQ_HIBF_LEVELS", '1');
    k.hibf_stationary = !is("TXQ_HIBF_STATIONARY", '0');
    k.hibf_small = !is("TXQ_HIBF_SMALL", '0');
    k.hibf_lane_hash = flag("TXQ_HIBF_LANE_HASH");
    k.hibf_layout_order = !is("TXQ_HIBF_LAYOUT_ORDER", '0');
    k.hibf_layout_fused = !is("TXQ_HIBF_LAYOUT_FUSED", '0');
    k.final_pinned = !is("TXQ_FINAL_PINNED", '0');
    k.hibf_steps_per_group = (int)std::max(0LL, num("TXQ_HIBF_STEPS_PER_GROUP", 0));
    k.hibf_tile = (int)std::max(0LL, num("TXQ_HIBF_TILE", 0));
    k.hibf_unroll = (int)num("TXQ_HIBF_UNROLL", 1);
    k.hibf_store = (int)num("TXQ_HIBF_STORE_KIND", 0) & 3;  // which store instruction writes the rows
#ifdef TXQ_EXPERIMENTS
    // timing experiments of tools/ab_hibf*.sh (`make EXPERIMENTS=1`): bit 4 no row gathers, bit 5 (almost) no stores — WRONG masks,
    // which is why the product build does not contain them
    k.hibf_store |= (int)num("TXQ_HIBF_STORE", 0) & 112;  // (64: the layout-order level kernel without its gate loads)
#endif
    k.hibf_waves = std::max(0LL, num("TXQ_HIBF_WAVES", 0));
    k.hibf_stack_lds = std::max(2LL, num("TXQ_HIBF_STACK_LDS", 128));
    k.hibf_layout_direct = num("TXQ_HIBF_LAYOUT_DIRECT", 1) != 0;
    k.probe_blocks_per_cu = (int)std::max(1LL, num("TXQ_PROBE_BLOCKS_PER_CU", 256));
    k.probe_unroll = (int)num("TXQ_PROBE_UNROLL", 2);
    k.probe_nt = flag("TXQ_PROBE_NT");
    std::lock_guard<std::mutex> lock(g_knobs_mutex);
    g_knobs = k;
}

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
int fail_hip(hipError_t e, const char* what) {
    return fail(e == hipErrorOutOfMemory ? TXQ_ERR_NOMEM : TXQ_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}
// The HIP device is a per-thread setting: a thread other than the one that called txq_init (e.g. the host's
// stage-submission threads, one per shard) would otherwise talk to device 0, and the embedding application may
// have selected another device since the last call (torch.cuda.set_device): always ask, never trust a cached answer.
static int bind_device(int device) {
    int current = -1;
    if (hipGetDevice(&current) != hipSuccess || current != device) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
    }
    return TXQ_OK;
}
int require_init() {
    if (g_devices.empty()) return fail(TXQ_ERR_STATE, "txq_init has not been called (or found no GPU); this library has no CPU fallback");
    return bind_device(g_devices[0]);
}
// calls that work on an index run on the device that holds it
static int bind_index(const Index* ix) {
    if (g_devices.empty()) return fail(TXQ_ERR_STATE, "txq_init has not been called (or found no GPU); this library has no CPU fallback");
    return bind_device(ix ? ix->device : g_devices[0]);
}

#define TXQ_HIP(call)                                        \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return fail_hip(e_, #call);    \
    } while (0)

// shard r of R owns mask words [lo, hi): as even as possible, earlier shards get the remainder
static void shard_range(uint64_t words, int r, int R, uint64_t* lo, uint64_t* hi) {
    uint64_t base = words / R, rem = words % R;
    *lo = base * r + (r < (int)rem ? r : rem);
    *hi = *lo + base + (r < (int)rem ? 1 : 0);
}

static int validate_ibf(const txq_ibf_desc& d, bool need_words) {
    if (d.bins == 0 || d.bin_size == 0) return fail(TXQ_ERR_ARG, "IBF with zero bins or rows");
    if (d.hash_funs < 1 || d.hash_funs > 5) return fail(TXQ_ERR_ARG, "hash_funs %llu outside 1..5", (unsigned long long)d.hash_funs);
    if (d.bin_words != (d.bins + 63) / 64 || d.tech_bins != d.bin_words * 64)
        return fail(TXQ_ERR_ARG, "inconsistent bins/tech_bins/bin_words (%llu/%llu/%llu)", (unsigned long long)d.bins,
                    (unsigned long long)d.tech_bins, (unsigned long long)d.bin_words);
    if (d.hash_shift != (uint64_t)__builtin_clzll(d.bin_size))
        return fail(TXQ_ERR_ARG, "hash_shift %llu != countl_zero(bin_size)", (unsigned long long)d.hash_shift);
    if (d.bin_words >> 31) return fail(TXQ_ERR_ARG, "more than 2^37 bins are not supported");
    if (need_words && !d.words) return fail(TXQ_ERR_ARG, "IBF descriptor without words");
    return TXQ_OK;
}

// Allocate one IBF (column slice [w0, w1) of its rows) in HBM; copies from `src` when given.
int alloc_ibf(const txq_ibf_desc& d, uint64_t w0, uint64_t w1, IbfDev* out, uint64_t* bytes) {
    IbfDev f{};
    f.bin_size = d.bin_size;
    f.hash_shift = (uint32_t)d.hash_shift;
    f.hash_funs = (uint32_t)d.hash_funs;
    f.bins = (uint32_t)d.bins;
    f.word0 = (uint32_t)w0;
    f.ident_word = kNoIdent;
    f.reserved = 0;
    f.shard_words = (uint32_t)(w1 - w0);
    f.stride = f.shard_words <= 1 ? 1u : ((f.shard_words + 1u) & ~1u);
    f.words = nullptr;
    *bytes = 0;
    if (f.shard_words) {
        const size_t nbytes = (size_t)d.bin_size * f.stride * 8;
        TXQ_HIP(hipMalloc((void**)&f.words, nbytes));
        *bytes = nbytes;
        hipError_t e = hipSuccess;
        if (f.stride != f.shard_words || !d.words) e = hipMemset(f.words, 0, nbytes);
        if (e == hipSuccess && d.words)
            e = hipMemcpy2D(f.words, (size_t)f.stride * 8, d.words + w0, (size_t)d.bin_words * 8, (size_t)f.shard_words * 8,
                            (size_t)d.bin_size, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(f.words);
            return fail_hip(e, "uploading IBF words");
        }
    }
    *out = f;
    return TXQ_OK;
}

void Index::release() {
    for (auto& f : ibf) if (f.words) (void)hipFree(f.words);
    ibf.clear();
    if (d_ibf) (void)hipFree(d_ibf);
    if (d_next) (void)hipFree(d_next);
    if (d_tb_user) (void)hipFree(d_tb_user);
    if (d_map_off) (void)hipFree(d_map_off);
    for (int i = 0; i < 2; ++i) {
        if (host_pipe.stream[i]) (void)hipStreamDestroy(host_pipe.stream[i]);
        if (host_pipe.done[i]) (void)hipEventDestroy(host_pipe.done[i]);
        if (host_pipe.d_kmers[i]) (void)hipFree(host_pipe.d_kmers[i]);
        if (host_pipe.d_masks[i]) (void)hipFree(host_pipe.d_masks[i]);
        if (host_pipe.bounce[i]) (void)hipHostFree(host_pipe.bounce[i]);
    }
    host_pipe = HostPipe{};
    for (const ArenaChunk& c : session_cache.chunks) (void)hipFree(c.p);
    for (const ArenaChunk& c : session_cache.block_chunks) (void)hipFree(c.p);
    for (StagingSet& t : session_cache.set)
        if (t.done) (void)hipEventDestroy(t.done);
    if (session_cache.upload) (void)hipStreamDestroy(session_cache.upload);
    if (session_cache.side) (void)hipStreamDestroy(session_cache.side);
    for (void* p : {(void*)session_cache.set[0].d_blob, (void*)session_cache.set[0].d_aux, (void*)session_cache.set[1].d_blob, (void*)session_cache.set[1].d_aux,
                    (void*)session_cache.set[0].d_masks, (void*)session_cache.set[1].d_masks})
        if (p) (void)hipFree(p);
    session_cache = SessionCache{};
    for (void* p : {(void*)d_vchunks, (void*)d_vpaths, (void*)d_vleaf, (void*)d_vuser, (void*)d_vgroups, (void*)d_vnodes, (void*)d_vnonrep, (void*)d_vrep, (void*)d_vsplit_range, (void*)d_vsplits, (void*)d_vside})
        if (p) (void)hipFree(p);
    d_vchunks = nullptr; d_vpaths = nullptr; d_vleaf = nullptr; d_vuser = nullptr; d_vgroups = nullptr; d_vnodes = nullptr;
    d_vnonrep = nullptr; d_vrep = nullptr; d_vsplit_range = nullptr; d_vsplits = nullptr; d_vside = nullptr;
    v_words = n_vchunks = 0;
    vlevels.clear();
    if (d_children) (void)hipFree(d_children);
    if (interleaved.words) (void)hipFree(interleaved.words);
    interleaved = IbfDev{};
    if (scratch_cm) (void)hipFree(scratch_cm);
    if (scratch_crows) (void)hipFree(scratch_crows);
    scratch_crows = nullptr; cap_crows = 0;
    d_children = nullptr; scratch_cm = nullptr; cap_cm = 0; n_children = 0;
    if (d_merged) (void)hipFree(d_merged);
    if (d_descend) (void)hipFree(d_descend);
    if (d_nodes) (void)hipFree(d_nodes);
    d_nodes = nullptr;
    d_descend = nullptr;
    if (d_merged_off) (void)hipFree(d_merged_off);
    d_merged = d_merged_off = nullptr;
    for (void* p : {(void*)scratch_kmers, (void*)scratch_masks, (void*)frontier[0], (void*)frontier[1], (void*)d_counts,
                    (void*)scratch_blob, (void*)scratch_slots, (void*)scratch_final, (void*)scratch_dense_kmers, (void*)scratch_dense_masks, (void*)kmer_table})
        if (p) (void)hipFree(p);
    kmer_table = nullptr; kmer_table_bits = 0;
    d_ibf = nullptr; d_next = d_tb_user = nullptr; d_map_off = nullptr;
    scratch_kmers = scratch_masks = nullptr; frontier[0] = frontier[1] = nullptr; d_counts = nullptr;
    scratch_blob = nullptr; scratch_slots = scratch_final = nullptr;
    if (host_final) { (void)hipHostFree(host_final); host_final = nullptr; cap_host_final = 0; }
    scratch_dense_kmers = scratch_dense_masks = nullptr; cap_dense_kmers = cap_dense_masks = 0;
}

int ensure(void** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes && *p) return TXQ_OK;
    if (*p) {  // kernels in flight may still use the buffer (a session's stages are not waited for one by one)
        (void)hipDeviceSynchronize();
        (void)hipFree(*p);
    }
    *p = nullptr; *cap = 0;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return fail_hip(e, "hipMalloc(scratch)");
    *cap = bytes;
    return TXQ_OK;
}

}  // namespace txq

using namespace txq;

struct txq_index : txq::Index {};
struct txq_session : txq::Session {};

extern "C" {

const char* txq_last_error(void) { return g_err.c_str(); }

int txq_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail_hip(e, "hipGetDeviceCount");
    return n;
}

int txq_init(int n_devices, const int* device_ids) {
    read_knobs();
    if (n_devices < 1 || n_devices > 64) return fail(TXQ_ERR_ARG, "n_devices must be 1..64 (got %d)", n_devices);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_devices.clear();
        return fail(TXQ_ERR_STATE, "no HIP device visible (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    }
    std::vector<int> devices;
    for (int i = 0; i < n_devices; ++i) {
        const int dev = device_ids ? device_ids[i] : i;
        if (dev < 0 || dev >= n) return fail(TXQ_ERR_ARG, "device %d out of range (have %d)", dev, n);
        hipDeviceProp_t prop;
        TXQ_HIP(hipGetDeviceProperties(&prop, dev));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(TXQ_ERR_STATE, "device %d is %s; this library is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
        devices.push_back(dev);
    }
    for (size_t i = devices.size(); i-- > 0;) {  // load the kernels now, not inside the first query (ends on devices[0])
        TXQ_HIP(hipSetDevice(devices[i]));
        preload_exec_kernels();
        preload_probe_kernels();
        preload_hibf_kernels();
        // A non-blocking stream is a hardware queue of its own: 9 ms to create.  A session needs two; the first session of a
        // process takes them from here instead of paying 18 ms inside its first query (later sessions on an index inherit its streams).
        std::lock_guard<std::mutex> lk(g_spare_mutex);
        while (g_spare_streams[devices[i]].size() < 2) {
            hipStream_t st = nullptr;
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
            g_spare_streams[devices[i]].push_back(st);
        }
    }
    g_devices = devices;
    return TXQ_OK;
}

int txq_shutdown(void) {
    g_devices.clear();
    return TXQ_OK;
}

static int check_desc(const txq_index_desc* desc, int shard_rank, int n_shards, txq_index** out, bool* hibf_out) {
    if (!desc || !out || !desc->ibf || desc->n_ibf == 0) return fail(TXQ_ERR_ARG, "null descriptor");
    if (n_shards < 1 || shard_rank < 0 || shard_rank >= n_shards) return fail(TXQ_ERR_ARG, "bad shard %d/%d", shard_rank, n_shards);
    const bool hibf = desc->next_ibf_id != nullptr || desc->tb_to_user_bin != nullptr || desc->n_ibf > 1;
    if (hibf && (!desc->next_ibf_id || !desc->tb_to_user_bin)) return fail(TXQ_ERR_ARG, "HIBF descriptor needs both maps");
    for (uint64_t i = 0; i < desc->n_ibf; ++i)
        if (int rc = validate_ibf(desc->ibf[i], true)) return rc;
    if (!hibf && desc->user_bins != desc->ibf[0].bins) return fail(TXQ_ERR_ARG, "flat IBF: user_bins must equal ibf[0].bins");
    if (desc->user_bins == 0) return fail(TXQ_ERR_ARG, "user_bins == 0");
    *hibf_out = hibf;
    return TXQ_OK;
}

// the upload itself: the index goes to the device of `device_rank`; it keeps mask columns `range_rank` of `range_shards`
static int upload_impl(const txq_index_desc* desc, bool hibf, int device_rank, int range_rank, int range_shards, txq_index** out) {
    const int shard_rank = range_rank, n_shards = range_shards;
    txq_index* ix = new (std::nothrow) txq_index();
    if (!ix) return fail(TXQ_ERR_NOMEM, "out of host memory");
    ix->device = g_devices[(size_t)device_rank % g_devices.size()];  // shards go round-robin over the devices of txq_init
    if (int rc = bind_device(ix->device)) { delete ix; return rc; }
    ix->is_hibf = hibf;
    ix->user_bins = desc->user_bins;
    ix->mask_words = (desc->user_bins + 63) / 64;
    ix->shard_rank = device_rank;
    ix->n_shards = range_shards;
    uint64_t lo, hi;
    shard_range(ix->mask_words, shard_rank, n_shards, &lo, &hi);
    ix->shard_word0 = lo;
    ix->shard_words = hi - lo;
    int rc = TXQ_OK;
    if (!hibf) {
        IbfDev f;
        uint64_t bytes;
        rc = alloc_ibf(desc->ibf[0], lo, hi, &f, &bytes);
        if (rc == TXQ_OK) { ix->ibf.push_back(f); ix->device_bytes += bytes; }
    } else {
        rc = hibf_upload(*ix, *desc);
    }
    if (rc != TXQ_OK) {
        ix->release();
        delete ix;
        return rc;
    }
    *out = ix;
    return TXQ_OK;
}

int txq_index_upload(const txq_index_desc* desc, int shard_rank, int n_shards, txq_index** out) {
    read_knobs();
    if (int rc = require_init()) return rc;
    bool hibf = false;
    if (int rc = check_desc(desc, shard_rank, n_shards, out, &hibf)) return rc;
    return upload_impl(desc, hibf, shard_rank, shard_rank, n_shards, out);
}

// Is the tree what hibf_upload recognises as a regular two-level one (root of merged bins only over leaf IBFs that each map an
// aligned run of user bins, all of one power-of-two row width, tiling the mask)?  Those shard by mask columns.
static bool regular_two_level(const txq_index_desc& d) {
    const uint64_t n = d.n_ibf;
    if (n < 2 || d.ibf[0].bins != n - 1) return false;
    const uint64_t wpr = d.ibf[1].bin_words;
    if (wpr < 1 || (wpr & (wpr - 1)) || wpr > 128 || (d.user_bins + 63) / 64 != wpr * (n - 1)) return false;
    std::vector<uint8_t> seen(n, 0), column(n - 1, 0);
    for (uint64_t b = 0; b < d.ibf[0].bins; ++b) {
        if (d.tb_to_user_bin[0][b] != TXQ_MERGED_BIN) return false;
        const uint64_t c = d.next_ibf_id[0][b];
        if (c == 0 || c >= n || seen[c]) return false;
        seen[c] = 1;
    }
    for (uint64_t i = 1; i < n; ++i) {
        if (d.ibf[i].bin_words != wpr) return false;
        const uint64_t base = d.tb_to_user_bin[i][0];
        if (base == TXQ_MERGED_BIN || base % (wpr * 64)) return false;
        for (uint64_t b = 0; b < d.ibf[i].bins; ++b)
            if (d.tb_to_user_bin[i][b] != base + b) return false;
        const uint64_t col = base / (wpr * 64);
        if (col >= n - 1 || column[col]) return false;
        column[col] = 1;
    }
    return true;
}

int txq_index_upload_subtrees(const txq_index_desc* desc, int shard_rank, int n_shards, txq_index** out) {
    read_knobs();
    if (int rc = require_init()) return rc;
    bool hibf = false;
    if (int rc = check_desc(desc, shard_rank, n_shards, out, &hibf)) return rc;
    if (!hibf || n_shards == 1) return upload_impl(desc, hibf, shard_rank, shard_rank, n_shards, out);
    const uint64_t n = desc->n_ibf;
    for (uint64_t i = 0; i < n; ++i)
        if (!desc->next_ibf_id[i] || !desc->tb_to_user_bin[i]) return fail(TXQ_ERR_ARG, "HIBF map %llu is null", (unsigned long long)i);
    if (regular_two_level(*desc)) return upload_impl(desc, hibf, shard_rank, shard_rank, n_shards, out);
    // the sub-trees under the root's merged bins and the row words under each (the whole tree is checked by hibf_upload below;
    // here only what the walk itself needs: children in range, no IBF reached twice)
    std::vector<int> owner_of_ibf(n, -1);  // which root bin's sub-tree an IBF belongs to (-1: the root)
    std::vector<uint64_t> weight(desc->ibf[0].bins, 0);
    std::vector<uint8_t> reached(n, 0);
    reached[0] = 1;
    for (uint64_t b = 0; b < desc->ibf[0].bins; ++b) {
        if (desc->tb_to_user_bin[0][b] != TXQ_MERGED_BIN) continue;
        std::vector<uint64_t> stack{desc->next_ibf_id[0][b]};
        while (!stack.empty()) {
            const uint64_t i = stack.back();
            stack.pop_back();
            if (i >= n || reached[i]) return fail(TXQ_ERR_ARG, "HIBF: bad child %llu (out of range or reached twice)", (unsigned long long)i);
            reached[i] = 1;
            owner_of_ibf[i] = (int)b;
            weight[b] += desc->ibf[i].bin_words;
            for (uint64_t c = 0; c < desc->ibf[i].bins; ++c)
                if (desc->tb_to_user_bin[i][c] == TXQ_MERGED_BIN) stack.push_back(desc->next_ibf_id[i][c]);
        }
    }
    for (uint64_t i = 0; i < n; ++i)
        if (!reached[i]) return fail(TXQ_ERR_ARG, "IBF %llu is unreachable from the root", (unsigned long long)i);
    // largest sub-tree first, to the shard that holds least (ties: the lower shard) — the same deal on every rank
    std::vector<uint64_t> order;
    for (uint64_t b = 0; b < desc->ibf[0].bins; ++b)
        if (desc->tb_to_user_bin[0][b] == TXQ_MERGED_BIN) order.push_back(b);
    std::stable_sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return weight[a] > weight[b]; });
    std::vector<uint64_t> load(n_shards, 0);
    std::vector<int> shard_of_bin(desc->ibf[0].bins, 0);  // (the root's own user bins: shard 0)
    for (uint64_t b : order) {
        int best = 0;
        for (int r = 1; r < n_shards; ++r)
            if (load[r] < load[best]) best = r;
        shard_of_bin[b] = best;
        load[best] += weight[b];
    }
    // this shard's tree: the root and its own sub-trees, renumbered in the original order; in its copy of the root the columns of
    // everybody else's technical bins are cleared and those bins become plain technical bins that never fire
    std::vector<uint64_t> new_id(n, UINT64_MAX), kept;
    for (uint64_t i = 0; i < n; ++i)
        if (i == 0 || shard_of_bin[owner_of_ibf[i]] == shard_rank) { new_id[i] = kept.size(); kept.push_back(i); }
    const txq_ibf_desc& root = desc->ibf[0];
    std::vector<uint64_t> keep_mask(root.bin_words, 0);
    for (uint64_t b = 0; b < root.bins; ++b)
        if (shard_of_bin[b] == shard_rank) keep_mask[b >> 6] |= 1ULL << (b & 63);
    std::vector<uint64_t> root_words((size_t)root.bin_size * root.bin_words);
    for (uint64_t r = 0; r < root.bin_size; ++r)
        for (uint64_t w = 0; w < root.bin_words; ++w) root_words[r * root.bin_words + w] = root.words[r * root.bin_words + w] & keep_mask[w];
    std::vector<txq_ibf_desc> ibfs;
    std::vector<std::vector<uint64_t>> next(kept.size()), user(kept.size());
    std::vector<const uint64_t*> next_p, user_p;
    for (size_t j = 0; j < kept.size(); ++j) {
        const uint64_t i = kept[j];
        txq_ibf_desc d = desc->ibf[i];
        if (i == 0) d.words = root_words.data();
        ibfs.push_back(d);
        next[j].assign(desc->ibf[i].bins, 0);
        user[j].assign(desc->ibf[i].bins, 0);
        for (uint64_t b = 0; b < desc->ibf[i].bins; ++b) {
            const uint64_t ub = desc->tb_to_user_bin[i][b];
            if (i == 0 && shard_of_bin[b] != shard_rank) { user[j][b] = 0; continue; }  // (cleared column: user bin 0 is never reported through it)
            user[j][b] = ub;
            if (ub == TXQ_MERGED_BIN) next[j][b] = new_id[desc->next_ibf_id[i][b]];
        }
    }
    for (size_t j = 0; j < kept.size(); ++j) { next_p.push_back(next[j].data()); user_p.push_back(user[j].data()); }
    const txq_index_desc pruned{kept.size(), ibfs.data(), next_p.data(), user_p.data(), desc->user_bins};
    if (int rc = upload_impl(&pruned, true, shard_rank, 0, 1, out)) return rc;
    (*out)->join_or = true;
    (*out)->n_shards = n_shards;
    return TXQ_OK;
}

int txq_index_create_ibf(uint64_t bins, uint64_t bin_size, uint64_t hash_funs, int shard_rank, int n_shards, txq_index** out) {
    if (int rc = require_init()) return rc;
    if (!out || n_shards < 1 || shard_rank < 0 || shard_rank >= n_shards) return fail(TXQ_ERR_ARG, "bad arguments");
    txq_ibf_desc d{};
    d.bins = bins;
    d.bin_words = (bins + 63) / 64;
    d.tech_bins = d.bin_words * 64;
    d.bin_size = bin_size;
    d.hash_shift = bin_size ? (uint64_t)__builtin_clzll(bin_size) : 0;
    d.hash_funs = hash_funs;
    d.words = nullptr;
    if (int rc = validate_ibf(d, false)) return rc;
    txq_index* ix = new (std::nothrow) txq_index();
    if (!ix) return fail(TXQ_ERR_NOMEM, "out of host memory");
    ix->device = g_devices[(size_t)shard_rank % g_devices.size()];
    if (int rc = bind_device(ix->device)) { delete ix; return rc; }
    ix->user_bins = bins;
    ix->mask_words = d.bin_words;
    ix->shard_rank = shard_rank;
    ix->n_shards = n_shards;
    uint64_t lo, hi;
    shard_range(ix->mask_words, shard_rank, n_shards, &lo, &hi);
    ix->shard_word0 = lo;
    ix->shard_words = hi - lo;
    IbfDev f;
    uint64_t bytes;
    int rc = alloc_ibf(d, lo, hi, &f, &bytes);
    if (rc != TXQ_OK) { delete ix; return rc; }
    ix->ibf.push_back(f);
    ix->device_bytes = bytes;
    *out = ix;
    return TXQ_OK;
}

int txq_index_get_info(const txq_index* ix, txq_index_info* info) {
    if (!ix || !info) return fail(TXQ_ERR_ARG, "null argument");
    info->user_bins = ix->user_bins;
    info->mask_words = ix->mask_words;
    info->shard_word0 = ix->shard_word0;
    info->shard_words = ix->shard_words;
    info->n_ibf = ix->ibf.size();
    info->device_bytes = ix->device_bytes;
    info->is_hibf = ix->is_hibf ? 1 : 0;
    info->device = ix->device;
    info->join_or = ix->join_or ? 1 : 0;
    info->shard_rank = ix->shard_rank;
    info->n_shards = ix->n_shards;
    info->reserved = 0;
    return TXQ_OK;
}

int txq_index_supports_dense(const txq_index* ix) {
    read_knobs();
    if (!ix || ix->ibf.empty() || ix->shard_words == 0) return 0;
    if (!ix->is_hibf) return (ix->ibf[0].bin_size >> 32) == 0 ? 2 : 0;
    return index_fuses_tree_steps(*ix, knobs()) || ix->layout_order(knobs()) ? 2 : 1;  // other HIBFs: steps run as k-mer batches through the descent
}

int txq_index_free(txq_index* ix) {
    if (!ix) return TXQ_OK;
    if (ix->open_sessions > 0)
        return fail(TXQ_ERR_STATE, "the index still has %d open session(s): end them first (txq_session_end)", ix->open_sessions);
    if (!g_devices.empty()) (void)bind_device(ix->device);
    ix->release();
    delete ix;
    return TXQ_OK;
}

int txq_index_memory(const txq_index* ix, uint64_t* free_bytes, uint64_t* kept_bytes) {
    if (!ix || !free_bytes || !kept_bytes) return fail(TXQ_ERR_ARG, "null argument");
    if (int rc = bind_index(ix)) return rc;
    size_t free_b = 0, total_b = 0;
    TXQ_HIP(hipMemGetInfo(&free_b, &total_b));
    uint64_t kept = 0;
    for (const Index::ArenaChunk& c : ix->session_cache.chunks) kept += (uint64_t)c.cap * 8;
    for (const Index::ArenaChunk& c : ix->session_cache.block_chunks) kept += (uint64_t)c.cap * 8;
    *free_bytes = free_b;
    *kept_bytes = kept;
    return TXQ_OK;
}

int txq_index_set_tag(txq_index* ix, uint64_t tag) {
    if (!ix) return fail(TXQ_ERR_ARG, "null argument");
    ix->user_tag = tag;
    return TXQ_OK;
}
int txq_index_get_tag(const txq_index* ix, uint64_t* tag) {
    if (!ix || !tag) return fail(TXQ_ERR_ARG, "null argument");
    *tag = ix->user_tag;
    return TXQ_OK;
}

int txq_index_download_words(const txq_index* ix, uint64_t* words, size_t n_words) {
    if (!ix) return fail(TXQ_ERR_ARG, "null argument");
    if (int rc = bind_index(ix)) return rc;
    if (!ix || !words) return fail(TXQ_ERR_ARG, "null argument");
    if (ix->is_hibf) return fail(TXQ_ERR_ARG, "download_words is for flat IBFs");
    const IbfDev& f = ix->ibf[0];
    if (n_words != (size_t)f.bin_size * f.shard_words) return fail(TXQ_ERR_ARG, "expected %zu words", (size_t)f.bin_size * f.shard_words);
    if (!f.shard_words) return TXQ_OK;
    TXQ_HIP(hipMemcpy2D(words, (size_t)f.shard_words * 8, f.words, (size_t)f.stride * 8, (size_t)f.shard_words * 8,
                        (size_t)f.bin_size, hipMemcpyDeviceToHost));
    return TXQ_OK;
}

int txq_probe_device(txq_index* ix, const uint64_t* d_kmers, size_t n, uint64_t* d_masks, uint64_t* d_alive, void* stream) {
    read_knobs();
    if (!ix) return fail(TXQ_ERR_ARG, "null argument");
    if (int rc = bind_index(ix)) return rc;
    if (!ix || (n && (!d_kmers || !d_masks))) return fail(TXQ_ERR_ARG, "null argument");
    if (n >> 32) return fail(TXQ_ERR_ARG, "at most 2^32-1 k-mers per call");
    hipStream_t s = (hipStream_t)stream;
    if (ix->is_hibf) return hibf_probe(*ix, knobs(), d_kmers, n, d_masks, d_alive, s);
    hipError_t e = launch_probe(ix->ibf[0], d_kmers, n, d_masks, d_alive, s);
    if (e != hipSuccess) return fail_hip(e, "probe kernel launch");
    return TXQ_OK;
}

// Host-buffer probe: chunks are pipelined over two streams — while chunk c's masks travel back
// (device -> pinned bounce buffer -> caller's memory, or straight into the caller's memory when
// that is pinned, e.g. from txq_host_alloc), chunk c+1 is probed.
int txq_probe(txq_index* ix, const uint64_t* kmers, size_t n, uint64_t* masks) {
    if (!ix) return fail(TXQ_ERR_ARG, "null argument");
    if (int rc = bind_index(ix)) return rc;
    if (!ix || (n && (!kmers || !masks))) return fail(TXQ_ERR_ARG, "null argument");
    const size_t W = ix->shard_words;
    if (W == 0 || n == 0) return TXQ_OK;
    // chunk: about 32 MiB of masks, at most 2^20 k-mers
    size_t chunk = ((size_t)32 << 20) / (W * 8);
    if (chunk > ((size_t)1 << 20)) chunk = (size_t)1 << 20;
    if (chunk < 1024) chunk = 1024;
    if (chunk > n) chunk = n;
    Index::HostPipe& hp = ix->host_pipe;
    for (int i = 0; i < 2; ++i) {
        if (!hp.stream[i]) TXQ_HIP(hipStreamCreateWithFlags(&hp.stream[i], hipStreamNonBlocking));
        if (!hp.done[i]) TXQ_HIP(hipEventCreateWithFlags(&hp.done[i], hipEventDisableTiming));
        if (int rc = ensure((void**)&hp.d_kmers[i], &hp.cap_kmers[i], chunk * 8)) return rc;
        if (int rc = ensure((void**)&hp.d_masks[i], &hp.cap_masks[i], chunk * W * 8)) return rc;
    }
    hipPointerAttribute_t attr{};
    const bool pinned_out = hipPointerGetAttributes(&attr, masks) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();  // an unregistered pointer is reported as an error: not one of ours
    if (!pinned_out) {
        for (int i = 0; i < 2; ++i) {
            if (hp.cap_bounce[i] < chunk * W * 8) {
                if (hp.bounce[i]) (void)hipHostFree(hp.bounce[i]);
                hp.bounce[i] = nullptr;
                hp.cap_bounce[i] = 0;
                TXQ_HIP(hipHostMalloc((void**)&hp.bounce[i], chunk * W * 8, hipHostMallocDefault));
                hp.cap_bounce[i] = chunk * W * 8;
            }
        }
    }
    size_t pending_off[2] = {0, 0}, pending_m[2] = {0, 0};
    auto drain = [&](int b) -> int {  // wait for buffer b's chunk and hand it to the caller
        if (!pending_m[b]) return TXQ_OK;
        TXQ_HIP(hipEventSynchronize(hp.done[b]));
        if (!pinned_out) std::memcpy(masks + pending_off[b] * W, hp.bounce[b], pending_m[b] * W * 8);
        pending_m[b] = 0;
        return TXQ_OK;
    };
    int b = 0;
    for (size_t off = 0; off < n; off += chunk, b ^= 1) {
        const size_t m = n - off < chunk ? n - off : chunk;
        if (int rc = drain(b)) return rc;  // this buffer's previous chunk
        // the HIBF descent has per-index scratch (frontiers): its chunks share one stream
        hipStream_t st = hp.stream[ix->is_hibf ? 0 : b];
        TXQ_HIP(hipMemcpyAsync(hp.d_kmers[b], kmers + off, m * 8, hipMemcpyHostToDevice, st));
        if (int rc = txq_probe_device(ix, hp.d_kmers[b], m, hp.d_masks[b], nullptr, st)) return rc;
        TXQ_HIP(hipMemcpyAsync(pinned_out ? (void*)(masks + off * W) : (void*)hp.bounce[b], hp.d_masks[b], m * W * 8, hipMemcpyDeviceToHost, st));
        TXQ_HIP(hipEventRecord(hp.done[b], st));
        pending_off[b] = off;
        pending_m[b] = m;
    }
    if (int rc = drain(b)) return rc;
    if (int rc = drain(b ^ 1)) return rc;
    return TXQ_OK;
}

int txq_host_alloc(void** ptr, size_t bytes) {
    if (int rc = require_init()) return rc;
    if (!ptr) return fail(TXQ_ERR_ARG, "null argument");
    TXQ_HIP(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return TXQ_OK;
}

int txq_host_free(void* ptr) {
    if (!ptr) return TXQ_OK;
    TXQ_HIP(hipHostFree(ptr));
    return TXQ_OK;
}

int txq_emplace_device(txq_index* ix, const uint64_t* d_values, const uint32_t* d_bins_of, size_t n, void* stream) {
    if (!ix) return fail(TXQ_ERR_ARG, "null argument");
    if (int rc = bind_index(ix)) return rc;
    if (!ix || (n && (!d_values || !d_bins_of))) return fail(TXQ_ERR_ARG, "null argument");
    if (ix->is_hibf) return fail(TXQ_ERR_ARG, "emplace is for flat IBFs");
    if (ix->open_sessions > 0)
        return fail(TXQ_ERR_STATE, "the index has %d open session(s): its bits cannot change under them (txq_session_end first)", ix->open_sessions);
    {   // what was derived from the old bits goes: the table of all k-mers' masks (dense steps would read stale rows), the host's
        // verdict on how states fare on the index (tag) stays — it is advisory
        std::lock_guard<std::mutex> lock(ix->table_mutex);
        if (ix->kmer_table) {
            TXQ_HIP(hipDeviceSynchronize());  // (kernels of ended sessions may still be reading it)
            (void)hipFree(ix->kmer_table);
            ix->kmer_table = nullptr;
            ix->kmer_table_bits = 0;
        }
        ix->kmer_table_refused = false;
    }
    hipError_t e = launch_emplace(ix->ibf[0], d_values, d_bins_of, n, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "emplace kernel launch");
    return TXQ_OK;
}

int txq_run_programs_device(txq_index* ix, const void* blob, size_t blob_bytes, size_t n_programs, uint64_t* d_final_masks, void* stream) {
    read_knobs();
    if (!ix) return fail(TXQ_ERR_ARG, "null argument");
    if (int rc = bind_index(ix)) return rc;
    if (!ix || !blob || (n_programs && !d_final_masks)) return fail(TXQ_ERR_ARG, "null argument");
    return run_programs(*ix, blob, blob_bytes, n_programs, d_final_masks, (hipStream_t)stream);
}

int txq_run_programs(txq_index* ix, const void* blob, size_t blob_bytes, size_t n_programs, uint64_t* final_masks) {
    read_knobs();
    if (!ix) return fail(TXQ_ERR_ARG, "null argument");
    if (int rc = bind_index(ix)) return rc;
    if (!ix || !blob || (n_programs && !final_masks)) return fail(TXQ_ERR_ARG, "null argument");
    const size_t bytes = n_programs * ix->shard_words * 8;
    if (bytes == 0) return TXQ_OK;
    if (int rc = ensure((void**)&ix->scratch_final, &ix->cap_final, bytes)) return rc;
    if (int rc = run_programs(*ix, blob, blob_bytes, n_programs, ix->scratch_final, nullptr)) return rc;
    TXQ_HIP(hipMemcpy(final_masks, ix->scratch_final, bytes, hipMemcpyDeviceToHost));
    return TXQ_OK;
}

int txq_session_begin(txq_index* ix, size_t n_programs, txq_session** out) {
    read_knobs();
    if (!ix) return fail(TXQ_ERR_ARG, "null argument");
    if (int rc = bind_index(ix)) return rc;
    if (!ix || !out) return fail(TXQ_ERR_ARG, "null argument");
    Session* s = nullptr;
    if (int rc = session_begin(*ix, n_programs, &s)) return rc;
    *out = static_cast<txq_session*>(s);
    return TXQ_OK;
}

int txq_session_set_aux_index(txq_session* s, txq_index* aux) {
    if (!s) return fail(TXQ_ERR_ARG, "null argument");
    if (aux) {
        if (aux->is_hibf) return fail(TXQ_ERR_ARG, "the auxiliary index must be a flat IBF");
        if (aux->user_bins != s->ix->user_bins || aux->shard_word0 != s->ix->shard_word0 || aux->shard_words != s->ix->shard_words ||
            aux->device != s->ix->device)
            return fail(TXQ_ERR_ARG, "the auxiliary index must cover the same bins and the same shard (on the same device) as the main index");
    }
    if (aux && s->vspace) {  // d-gram masks come in user-bin order: the session cannot work in layout order then
        if (s->n_stages) return fail(TXQ_ERR_STATE, "attach the auxiliary index before the session's first stage");
        s->vspace = false;
        s->W = (uint32_t)s->ix->shard_words;
    }
    if (s->aux) --s->aux->open_sessions;
    s->aux = aux;
    if (aux) ++aux->open_sessions;
    return TXQ_OK;
}

int txq_session_stage(txq_session* s, const void* blob, size_t blob_bytes, const uint32_t* query_program,
                      const uint32_t* query_slot, size_t n_queries, uint8_t* alive) {
    if (!s) return fail(TXQ_ERR_ARG, "null argument");
    if (int rc = bind_index(s->ix)) return rc;
    if ( !blob || (n_queries && (!query_program || !query_slot || !alive))) return fail(TXQ_ERR_ARG, "null argument");
    if (s->failed) return fail(TXQ_ERR_STATE, "an earlier stage of this session failed: end the session");
    const int rc = session_stage(*s, blob, blob_bytes, query_program, query_slot, n_queries, alive, nullptr);
    if (rc != TXQ_OK) {
        // Kernels of the stage may already be running (on either stream) and the stage's bookkeeping is half done: nothing of
        // this session may be in flight when its buffers change hands, and no further stage may build on it
        const std::string why = g_err;
        (void)hipDeviceSynchronize();
        s->failed = true;
        g_err = why;
    }
    return rc;
}

int txq_session_end(txq_session* s, uint64_t* final_masks) {
    if (!s) return TXQ_OK;
    int rc = bind_index(s->ix);  // the calling thread may never have selected the device
    if (rc != TXQ_OK) { delete static_cast<Session*>(s); return rc; }
    const bool trace = s->kn.trace;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    double t_finish = t0, t_copy = t0;
    if (final_masks && s->failed) {
        // a stage of this session failed: its slots hold half-executed state, there are no final masks to hand out
        rc = fail(TXQ_ERR_STATE, "a stage of this session failed: it has no final masks");
    } else if (final_masks && s->n_programs && s->W) {
        Index& ix = *s->ix;
        const size_t bytes = s->n_programs * (size_t)ix.shard_words * 8;  // (a layout-order session hands out user-bin masks too)
        if (s->kn.final_pinned && !s->vspace && bytes <= ((size_t)64 << 20)) {  // (layout-order rows are first put back in user-bin order, with atomics: on the device)
            // The gather kernel writes the masks straight into pinned host memory (kept with the index): no copy engine is
            // involved.  In a fresh process (`tetrex query`) the device-to-host copy of a session's 25 KB of final masks took
            // 8 ms with the device idle — pageable or pinned destination, hipMemcpy or hipMemcpyAsync alike —, as much as the
            // whole batch on the device; the kernel's stores over the link take 0.02 ms (profiles/r4_cold_cli_final_masks.txt).
            if (ix.cap_host_final < bytes) {
                if (ix.host_final) (void)hipHostFree(ix.host_final);
                ix.host_final = nullptr;
                ix.cap_host_final = 0;
                const size_t cap = std::max<size_t>(bytes + bytes / 2, (size_t)1 << 20);
                hipError_t e = hipHostMalloc((void**)&ix.host_final, cap, hipHostMallocDefault);
                if (e != hipSuccess) rc = fail_hip(e, "pinned buffer for the final masks");
                else ix.cap_host_final = cap;
            }
            if (rc == TXQ_OK) rc = session_finish(*s, ix.host_final, nullptr);
            t_finish = now();
            if (rc == TXQ_OK) {
                hipError_t e = hipStreamSynchronize(nullptr);
                if (e != hipSuccess) rc = fail_hip(e, "waiting for the final masks");
                else std::memcpy(final_masks, ix.host_final, bytes);
            }
        } else {
            rc = ensure((void**)&ix.scratch_final, &ix.cap_final, bytes);
            if (rc == TXQ_OK) rc = session_finish(*s, ix.scratch_final, nullptr);
            t_finish = now();
            if (rc == TXQ_OK) {
                hipError_t e = hipMemcpy(final_masks, ix.scratch_final, bytes, hipMemcpyDeviceToHost);
                if (e != hipSuccess) rc = fail_hip(e, "copying final masks");
            }
        }
        t_copy = now();
    }
    (void)hipDeviceSynchronize();
    const double t_sync = now();
    delete static_cast<Session*>(s);
    if (trace)
        fprintf(stderr, "[txq] session end: last launches %.2f ms, masks to the host %.2f ms, device idle after %.2f ms, session released in %.2f ms\n",
                (t_finish - t0) * 1e3, (t_copy - t_finish) * 1e3, (t_sync - t_copy) * 1e3, (now() - t_sync) * 1e3);
    return rc;
}

int txq_malloc(void** dptr, size_t bytes) {
    if (int rc = require_init()) return rc;
    if (!dptr) return fail(TXQ_ERR_ARG, "null argument");
    TXQ_HIP(hipMalloc(dptr, bytes ? bytes : 8));
    return TXQ_OK;
}
int txq_free(void* dptr) {
    if (dptr) TXQ_HIP(hipFree(dptr));
    return TXQ_OK;
}
int txq_memcpy_h2d(void* dst, const void* src, size_t bytes) {
    if (int rc = require_init()) return rc;
    TXQ_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return TXQ_OK;
}
int txq_memcpy_d2h(void* dst, const void* src, size_t bytes) {
    if (int rc = require_init()) return rc;
    TXQ_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return TXQ_OK;
}
int txq_synchronize(void) {
    if (int rc = require_init()) return rc;
    TXQ_HIP(hipDeviceSynchronize());
    return TXQ_OK;
}

}  // extern "C"

/* txq.h — C-ABI of the MI355X (gfx950) TetRex query engine (libtxq.so).
 *
 * This is the drop-in boundary for the (H)IBF probe hot path of remyschwab/TetRex.  Every
 * entry point replaces one seam of the reference (paths relative to the reference root):
 *
 *   txq_index_upload        <- TetrexIndex::spawn_agent()            include/index_base.h:145
 *                              IBFIndex::spawn_agent / HIBFIndex::spawn_agent
 *                                                                     include/index_ibf.h:141-144, index_hibf.h:149-152
 *                              (the agent's raw pointer into the bit matrix becomes an HBM-resident copy)
 *   txq_probe / _device     <- TetrexIndex::query(uint64_t) -> bitvector  include/index_base.h:104-107
 *                              IBFIndex::query  -> bulk_contains      include/index_ibf.h:146-150
 *                              HIBFIndex::query -> membership_for(.,1) + populate_bitvector
 *                                                                     include/index_hibf.h:132-147
 *                              (batched: one call probes n k-mers; call sites include/otf_collector.h:262,273)
 *   txq_run_programs        <- OTFCollector::collect()               include/otf_collector.h:341-393
 *                              (the mask algebra of the collector — path_ &= hits, absorb |=, Match |= —
 *                               compiled by the host into mask-DAG programs, see txq_program.h)
 *   txq_emplace_device      <- interleaved_bloom_filter::emplace     include/index_ibf.h:94-98 (index build, "next")
 *
 * Conventions: plain C, no exceptions cross the boundary.  Every function returns 0 on success or a
 * negative txq_status; txq_last_error() gives the message for the calling thread.  Host buffers are
 * owned by the caller; device memory is owned by the library unless the function name ends in
 * _device (then pointers are HIP device pointers owned by the caller and `stream` is a hipStream_t,
 * NULL = default stream, and the call is asynchronous on that stream).  One txq_index may be used
 * from one host thread at a time; distinct indexes are independent.
 *
 * Bit layout (identical to seqan::hibf::bit_vector as used by the reference): mask word w, bit b
 * (LSB first) <-> bin 64*w + b.  Bits >= bins in the last word are zero.
 */
#ifndef TXQ_H
#define TXQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TXQ_VERSION 1

/* Environment variables.  Every one selects between code paths that give the SAME results (A/B measurements, tests that
 * run one input through every path); none is needed in production.  They are read at txq_init, txq_index_upload,
 * txq_session_begin, txq_run_programs* and txq_probe* — never while a stage runs — and a session keeps the values it began with.
 *   TXQ_TRACE, TXQ_TRACE_STAGES, TXQ_TRACE_SYNC      timers and per-stage notes on stderr
 *   TXQ_DENSE_TREE=0|1|2                              dense steps on a regular HIBF: generic descent | TreeRows | TreeRowsByLane
 *   TXQ_DENSE_UNROLL, TXQ_DENSE_SLICES, TXQ_DENSE_TILE_ROUNDS, TXQ_DENSE_NT   shape of a dense step's tiles, cache policy of its destination accesses
 *   TXQ_FUSE_UNITS=0, TXQ_ONE_STREAM                  one launch per kind and level; no second stream
 *   TXQ_FINAL_PINNED=0                                a session's final masks gathered on the device and copied, not written straight into pinned host memory
 *   TXQ_SPARSE_STEPS=0, TXQ_SPARSE_UNROLL=2|3, TXQ_SPARSE_UNITS=<n>
 *                                                     pushed steps of tracked blocks on narrow masks: rounds of entries instead of units; units in
 *                                                     flight per lane group; units per chunk (default 512)
 *   TXQ_KMER_TABLE_MB=<n>, TXQ_KMER_TABLE_MIN=<n>     most an index's table of all k-mers' masks may take (default 512; 0: none);
 *                                                     the session of fewest programs that builds it (default 16)
 *   TXQ_HIBF_INTERLEAVE=0, TXQ_HIBF_INTERLEAVE_PROBE=0, TXQ_HIBF_LEVELS=1, TXQ_HIBF_STATIONARY=0, TXQ_HIBF_SMALL=0,
 *   TXQ_HIBF_LAYOUT_ORDER=0, TXQ_HIBF_LAYOUT_FUSED=0, TXQ_HIBF_LANE_HASH, TXQ_HIBF_STEPS_PER_GROUP, TXQ_HIBF_TILE, TXQ_HIBF_UNROLL, TXQ_HIBF_STORE_KIND, TXQ_HIBF_WAVES, TXQ_HIBF_STACK_LDS, TXQ_HIBF_LAYOUT_DIRECT=0
 *                                                     which HIBF descent kernel runs, and its tiling
 *   TXQ_PROBE_BLOCKS_PER_CU, TXQ_PROBE_UNROLL, TXQ_PROBE_NT   grid and variant of the flat probe kernel
 * (tests/test_gpu_knobs.py runs a workload under each of them against the oracle.) */

typedef enum {
    TXQ_OK = 0,
    TXQ_ERR_ARG = -1,      /* invalid argument / inconsistent descriptor            */
    TXQ_ERR_HIP = -2,      /* a HIP runtime call failed (message has the HIP error)  */
    TXQ_ERR_NOMEM = -3,    /* host or device allocation failed                       */
    TXQ_ERR_STATE = -4,    /* txq_init not called / no GPU present                   */
    TXQ_ERR_OVERFLOW = -5, /* an internal work queue overflowed (HIBF frontier)      */
    TXQ_ERR_PROGRAM = -6   /* malformed mask-DAG program                             */
} txq_status;

typedef struct txq_index txq_index; /* opaque; owns device memory */

/* Host view of one interleaved Bloom filter, exactly the six scalars + the word array that
 * seqan::hibf::interleaved_bloom_filter serialises (SURVEY.md §8c ".ibf layout"):
 * words[r * bin_words + w] holds technical bins 64w..64w+63 of row r. */
typedef struct {
    uint64_t bins;       /* number of (technical) bins in use, B                  */
    uint64_t tech_bins;  /* 64 * ceil(B / 64)                                     */
    uint64_t bin_size;   /* rows m                                                */
    uint64_t hash_shift; /* countl_zero(bin_size)                                 */
    uint64_t bin_words;  /* tech_bins / 64                                        */
    uint64_t hash_funs;  /* h, 1..5                                               */
    const uint64_t* words;
} txq_ibf_desc;

#define TXQ_MERGED_BIN UINT64_MAX

/* Host view of a whole index.  n_ibf == 1 and NULL maps: a flat IBF (IBFIndex).  Otherwise an
 * HIBF: ibf[0] is the root, next_ibf_id[i][b] is the child IBF of merged technical bin b of
 * IBF i, tb_to_user_bin[i][b] its user bin or TXQ_MERGED_BIN; each map has ibf[i].bins entries. */
typedef struct {
    uint64_t n_ibf;
    const txq_ibf_desc* ibf;
    const uint64_t* const* next_ibf_id;
    const uint64_t* const* tb_to_user_bin;
    uint64_t user_bins; /* bits in a result mask (== ibf[0].bins for a flat IBF) */
} txq_index_desc;

typedef struct {
    uint64_t user_bins;    /* bits of a full (unsharded) mask                         */
    uint64_t mask_words;   /* words of a full mask = ceil(user_bins / 64)             */
    uint64_t shard_word0;  /* first mask word owned by this shard                     */
    uint64_t shard_words;  /* mask words owned by this shard (what txq_probe emits)   */
    uint64_t n_ibf;        /* 1 for a flat IBF                                        */
    uint64_t device_bytes; /* HBM held by the index                                   */
    int      is_hibf;
    int      device;       /* HIP device ordinal                                      */
    int      join_or;      /* 0: this shard owns the mask-word columns [shard_word0, shard_word0 + shard_words);
                              1: a SUB-TREE shard (txq_index_upload_subtrees): it emits full-width masks (shard_word0 = 0,
                              shard_words = mask_words) that hold only the user bins of its sub-trees — the shards' masks are ORed */
    int      shard_rank;   /* which shard of ... */
    int      n_shards;     /* ... how many this index was uploaded as                 */
    int      reserved;
} txq_index_info;

/* Bind this process to n_devices GPUs (device_ids == NULL: devices 0 .. n_devices-1).  One process per GPU
 * (n_devices == 1, the torch.distributed / RCCL deployment) and one process driving all GPUs of a node are both
 * supported: shard r of an index (txq_index_upload, txq_index_create_ibf) lives on device_ids[r % n_devices], and
 * every call on an index or session runs on the device that holds it, whichever thread makes it.
 * Fails with TXQ_ERR_STATE when no GPU is present: there is no CPU fallback anywhere in this library. */
int txq_init(int n_devices, const int* device_ids);
int txq_shutdown(void);
const char* txq_last_error(void);
int txq_device_count(void); /* >= 0, or a negative txq_status */

/* Copy an index into HBM (of device_ids[shard_rank % n_devices], see txq_init).  With n_shards > 1 only the
 * mask-word columns of shard `shard_rank` are kept (flat IBF: words [W*r/R, W*(r+1)/R) of every row, re-laid out
 * contiguously; HIBF: the whole tree is kept and only the user-bin mask columns are sharded). */
int txq_index_upload(const txq_index_desc* desc, int shard_rank, int n_shards, txq_index** out);
/* The same for an HIBF, sharded by SUB-TREES where the tree is a general one (as seqan::hibf's layout shapes it: reference
 * include/index_hibf.h:114-129): the root is replicated, its merged bins — the level-1 sub-trees — are dealt over the shards
 * by the row words under them (largest first, to the shard that holds least), the user bins that sit in the root itself go to
 * shard 0, and a shard keeps only its own sub-trees (in its copy of the root the technical bins of the others are cleared, so
 * nothing is ever descended into or reported for them).  Such a shard emits FULL-WIDTH masks (info.join_or = 1) with the user
 * bins of its sub-trees only; a split bin may straddle shards: the join is an OR.  Its sessions work in layout order like an
 * unsharded general tree's.  A regular two-level tree (what `tetrex index` writes) and a flat IBF are sharded by mask
 * columns exactly as txq_index_upload does (info.join_or = 0). */
int txq_index_upload_subtrees(const txq_index_desc* desc, int shard_rank, int n_shards, txq_index** out);
int txq_index_get_info(const txq_index* ix, txq_index_info* info);
int txq_index_free(txq_index* ix);

/* Do sessions on this index execute dense DP steps (include/txq_program.h version 3)?  0: no (flat IBFs with 2^32 rows
 * and more).  1: yes (any HIBF: a step runs as a batch of k-mers through the descent).  2: yes, fused into one kernel
 * (flat IBFs, regular two-level HIBFs) — such sessions also run TRACKED programs (sparse blocks with live lists). */
int txq_index_supports_dense(const txq_index* ix);

/* Device memory as a session on this index will find it: bytes free on the index's device, and bytes of slot storage the
 * index keeps from earlier sessions (the next session reuses them before it allocates).  Host layers size their budgets
 * for slot storage (dense blocks) by the sum instead of by a constant. */
int txq_index_memory(const txq_index* ix, uint64_t* free_bytes, uint64_t* kept_bytes);

/* One 64-bit value a host layer may keep with the index (0 after upload); libtetrex_query stores what its staged
 * expansion has learned about the index there. */
int txq_index_set_tag(txq_index* ix, uint64_t tag);
int txq_index_get_tag(const txq_index* ix, uint64_t* tag);

/* An empty (all-zero) flat IBF living only in HBM, for device-side construction. */
int txq_index_create_ibf(uint64_t bins, uint64_t bin_size, uint64_t hash_funs, int shard_rank, int n_shards,
                         txq_index** out);
/* Copy the shard's bit matrix back, row-major [bin_size][shard_words]. */
int txq_index_download_words(const txq_index* ix, uint64_t* words, size_t n_words);

/* Batched bulk_contains: masks[i * shard_words + w] for k-mer i.  Host buffers; synchronous.
 * Chunks are pipelined (probe of chunk c+1 overlaps the copy-back of chunk c); a `masks` buffer
 * from txq_host_alloc (pinned) receives the device copies directly, any other one goes through a
 * pinned bounce buffer. */
int txq_probe(txq_index* ix, const uint64_t* kmers, size_t n, uint64_t* masks);
/* Same on device-resident buffers, asynchronous on `stream`.  d_alive may be NULL; otherwise it
 * receives ceil(n/64) words, bit i set iff mask i has any bit set in this shard
 * (bitvector::none() of include/otf_collector.h:383, per shard). */
int txq_probe_device(txq_index* ix, const uint64_t* d_kmers, size_t n, uint64_t* d_masks, uint64_t* d_alive,
                     void* stream);

/* Device-side emplace: value i is inserted into bin bins_of[i] (flat IBF only; bins outside this
 * shard's columns are skipped). */
int txq_emplace_device(txq_index* ix, const uint64_t* d_values, const uint32_t* d_bins_of, size_t n, void* stream);

/* Mask-DAG programs (format: include/txq_program.h).  Runs n_programs programs found in
 * `blob`; final_masks receives n_programs x shard_words words (host buffer, synchronous). */
int txq_run_programs(txq_index* ix, const void* blob, size_t blob_bytes, size_t n_programs, uint64_t* final_masks);
int txq_run_programs_device(txq_index* ix, const void* blob, size_t blob_bytes, size_t n_programs,
                            uint64_t* d_final_masks, void* stream);

/* Staged execution: the host expands the frontier of a batch of queries piecewise and streams
 * each piece (a blob with the NEW ops of every program; slot contents persist in HBM between
 * stages).  After a stage the library answers n_queries feedback questions "has slot
 * query_slot[i] of program query_program[i] any bit set?" (bitvector::none() of
 * include/otf_collector.h:383) into alive[i]: 0 = none, otherwise 1 + floor(log2(number of bits set in this
 * shard's columns)) — alive or not is all the collector needs; how FULL the surviving masks are tells the host whether
 * state lists saturate on this index (dense DP steps pay) or thin out (pruning pays), so the host can prune dead
 * states before expanding them.  Every stage's blob must hold exactly n_programs programs (possibly with no
 * ops) and their current n_slots.  txq_session_end copies the RESULT slot of every program to
 * final_masks (n_programs x shard_words, host) and destroys the session; a NULL final_masks
 * just destroys it. */
typedef struct txq_session txq_session;
int txq_session_begin(txq_index* ix, size_t n_programs, txq_session** out);
/* Attach the d-gram index of `tetrex query -g` (reference include/otf_collector.h:235,
 * include/dGramIndex.h:279-283): a flat IBF over the same bins, uploaded with the same shard.  The
 * last n_aux_kmers entries of a stage's k-mer table (txq_program.h) are then probed on it. */
int txq_session_set_aux_index(txq_session* s, txq_index* aux);
int txq_session_stage(txq_session* s, const void* blob, size_t blob_bytes, const uint32_t* query_program,
                      const uint32_t* query_slot, size_t n_queries, uint8_t* alive);
int txq_session_end(txq_session* s, uint64_t* final_masks);

/* Plain device-memory helpers so that a host program needs no HIP headers. */
int txq_malloc(void** dptr, size_t bytes);
int txq_free(void* dptr);
int txq_memcpy_h2d(void* dst, const void* src, size_t bytes);
int txq_memcpy_d2h(void* dst, const void* src, size_t bytes);
int txq_synchronize(void);
/* Page-locked host memory (hipHostMalloc): fastest source/destination of the host-buffer entry points. */
int txq_host_alloc(void** ptr, size_t bytes);
int txq_host_free(void* ptr);

#ifdef __cplusplus
}
#endif
#endif /* TXQ_H */

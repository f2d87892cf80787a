/* txq_program.h — mask-DAG program blob consumed by txq_run_programs (include/txq.h).
 *
 * A program is the mask algebra OTFCollector::collect() performs for ONE query
 * (reference include/otf_collector.h:341-393), with the data-dependent parts removed:
 *   path_ &= hits            (update_path, :265,276)   ->  dst = a & M[kmer]
 *   absorb: path_ |= other   (:204-208)                ->  dst = a | b
 *   Match:  path_matrix |= path_ (:361-363)            ->  RESULT = a | RESULT
 * Every op has the single form   slot[dst] = (slot[a] & K) | slot[b]   with K = M[kmer], the
 * bin mask of k-mer table entry `kmer` (bulk_contains / membership_for), or all ones when
 * kmer == TXQ_NO_KMER.  Pruning of dead states (:383) does not change any result mask (a dead
 * state only contributes zeros), so programs may enumerate a superset of the states the
 * reference visits.
 *
 * Slots are W-word masks private to a program.  Reserved slots:
 *   0 TXQ_SLOT_ZERO    all zero, never written
 *   1 TXQ_SLOT_ONES    the initial hit_vector(bin_count, true) (bits >= user_bins are zero)
 *   2 TXQ_SLOT_RESULT  starts zero; its final value is the program's candidate-bin mask
 * Every other slot must be written before it is read.
 *
 * Blob = header | kmers[n_kmers] | programs[n_programs] | ops[n_ops], little endian, offsets in
 * bytes from the start of the blob, each table 8-byte aligned.  The k-mer table holds the VALUES
 * to probe (canonical k-mers for DNA), deduplicated by the host across the whole batch — the
 * device probes each distinct k-mer once (the reference's kmer_cache_, :54,260-264).
 */
#ifndef TXQ_PROGRAM_H
#define TXQ_PROGRAM_H
#include <stdint.h>

#define TXQ_PROGRAM_MAGIC 0x50515854u /* "TXQP" */
#define TXQ_PROGRAM_VERSION 1u   /* programs are executed strictly in op order                  */
#define TXQ_PROGRAM_VERSION_LEVELS 2u /* ops are grouped into dependency levels (see below)      */
#define TXQ_NO_KMER 0xFFFFFFFFu
#define TXQ_SLOT_ZERO 0u
#define TXQ_SLOT_ONES 1u
#define TXQ_SLOT_RESULT 2u
#define TXQ_SLOT_FIRST_FREE 3u

typedef struct {
    uint32_t magic;
    uint32_t version;
    uint32_t n_programs;
    uint32_t n_kmers;
    uint32_t n_ops;
    uint32_t reserved;
    uint64_t kmers_offset;
    uint64_t programs_offset;
    uint64_t ops_offset;
} txq_blob_header;

typedef struct {
    uint32_t first_op; /* index into ops[] */
    uint32_t n_ops;
    uint32_t n_slots;  /* >= 3 */
    uint32_t reserved;
} txq_program;

typedef struct {
    uint32_t kmer; /* index into kmers[] or TXQ_NO_KMER */
    uint32_t dst;
    uint32_t a;
    uint32_t b;
} txq_op;

/* Version 2 — level-scheduled programs.  The ops of a program are ordered by dependency LEVEL:
 * all ops of one level may execute concurrently, a level starts when the previous one is
 * complete.  Within a level
 *   - no op reads a slot that another op of the level writes,
 *   - no two ops write the same slot, EXCEPT accumulations  slot[dst] |= slot[x]  (kmer ==
 *     TXQ_NO_KMER and dst == a or dst == b), any number of which may target the same dst
 *     (the device applies them atomically).
 * Layout: txq_blob_header_v2 | kmers | programs (txq_program_v2) | ops | levels, where
 * levels[first_level .. first_level + n_levels) are the END op indices (relative to first_op,
 * ascending, the last equal to n_ops) of the program's levels.  n_levels == 0 with n_ops > 0
 * means "execute in op order" as in version 1. */
typedef struct {
    uint32_t magic;
    uint32_t version; /* TXQ_PROGRAM_VERSION_LEVELS */
    uint32_t n_programs;
    uint32_t n_kmers;
    uint32_t n_ops;
    uint32_t n_levels; /* total entries of the levels table */
    uint64_t kmers_offset;
    uint64_t programs_offset;
    uint64_t ops_offset;
    uint64_t levels_offset;
    uint64_t n_aux_kmers; /* the LAST n_aux_kmers entries of kmers[] are probed on the session's
                             auxiliary index (the d-gram index of `tetrex query -g`), the others on
                             the main index */
} txq_blob_header_v2;

typedef struct {
    uint32_t first_op;
    uint32_t n_ops;
    uint32_t n_slots;
    uint32_t first_level;
    uint32_t n_levels;
    uint32_t reserved;
} txq_program_v2;

#endif /* TXQ_PROGRAM_H */

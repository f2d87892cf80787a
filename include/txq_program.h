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

/* Version 4 — dense DP steps (blob version TXQ_PROGRAM_VERSION_DENSE; a superset of version 2; version 3 numbered
 * dense slots block * N + index and is no longer accepted).
 *
 * Where a query's state set saturates (a run of wildcards or residue classes makes every (k-1)-symbol
 * suffix a live state: 20^(k-1) states, and 20 times as many ops for the next wildcard) the host stops
 * enumerating states.  It keeps such a state set as a DENSE BLOCK: N = A^(k-1) consecutive slots of the
 * program's dense region (A = number of residue codes of the index's alphabet), the slot of suffix
 * (y1 .. y_{k-1}), y1 oldest, at index  sum_j code(y_j) * A^(k-1-j).  Absent states are zero masks.
 * One DENSE STEP then does the collector's whole update_path/absorb round for a set R of residues
 * (reference include/otf_collector.h:247-278,190-208) on the device:
 *     dst[(x1 .. x_{k-2}, r)] |= OR over a in shape[0] of  src[(a, x1 .. x_{k-2})] & M[kmer(a, x1 .. x_{k-2}, r)]
 * for every r in R and every (x1 .. x_{k-2}) in shape[1] x .. x shape[k-2]; M[.] is bulk_contains of the
 * (for DNA: canonical) k-mer, evaluated inside the kernel — these masks never touch HBM.
 *
 * Slots with TXQ_DENSE_SLOT_BIT set address the program's dense blocks: slot = BIT | block << 22 | index, at most 256
 * blocks of at most 2^22 entries.  Ordinary ops may read and write them (scattering enumerated states into a block,
 * copying block entries out).
 * A dense op sits in the op stream as  { kmer = TXQ_DENSE_OP, dst = index into the dense table, a = b = 0 }
 * and obeys the level rules with its blocks as operands (a step reads all of src, reads and writes all of dst).
 * On a flat IBF a step is one fused kernel; on an HIBF the predecessor k-mers are written out, descended as one
 * batch and combined (txq_exec.hip). */
#define TXQ_PROGRAM_VERSION_DENSE 4u
#define TXQ_DENSE_BLOCK_SHIFT 22u
#define TXQ_DENSE_INDEX_MASK 0x3FFFFFu
#define TXQ_DENSE_MAX_BLOCKS 256u
#define TXQ_DENSE_OP 0xFFFFFFFEu
#define TXQ_DENSE_SLOT_BIT 0x40000000u
#define TXQ_DENSE_MAX_POSITIONS 11u /* k - 1 <= 11 */

enum { TXQ_DENSE_ZERO = 0,   /* r_mask == 0: dst block := 0; r_mask != 0: only its entries inside shape[0] x .. x
                                shape[k-2] := 0 (the others are never read before the block is zeroed again)  */
       TXQ_DENSE_STEP = 1,   /* see above                                                             */
       TXQ_DENSE_REDUCE = 2, /* slot dst |= OR of the src entries inside shape[0] x .. x shape[k-2]   */
       TXQ_DENSE_FILL = 3    /* every dst entry inside shape[0] x .. x shape[k-2] |= slot src (an ORDINARY slot):
                                a product-shaped list of states that all carry one mask (the states behind a run
                                of wildcards that have not been probed yet) becomes a block with one op         */ };

/* Tracked (sparse) blocks.  A program whose txq_program_v2.reserved has TXQ_PROGRAM_TRACKED_BIT set keeps, next to
 * every block, the list of its entries that hold a bit ("live list").  The meaning of its dense ops is unchanged;
 * what changes is the work: a STEP is pushed from the live entries of src (rows gathered, ANDed, ORed into the
 * destination entry, which joins dst's list the first time it receives a bit), a REDUCE reads the live entries only,
 * a ZERO clears them — cost proportional to the states that are ALIVE, not to the shape.  This is what makes blocks
 * pay at k >= 6 (21^5 suffixes, of which an index of real sequences keeps a few thousand alive): the collector's
 * path_.none() pruning (reference include/otf_collector.h:383) happens on the device, entry by entry.
 * Every dense op of a tracked program carries TXQ_DENSE_TRACKED in `reserved`.
 *
 * A tracked block is also SMALLER than A^(k-1): it is laid out inside its GEOMETRY, per suffix position the set of codes
 * that can occur there at all (the host derives it from the k-graph: the residue classes of the motif around that
 * place), entry index = the mixed-radix number of the codes' ranks within those sets (oldest position most significant).
 * A tracked ZERO (re)creates its block: shape[] = the geometry, src = the block's capacity in entries (>= the product of
 * the sets' sizes; a block id keeps its capacity for as long as the program lives).  The other tracked ops find the
 * geometry of their blocks on the device; their shape[] is a hint.  At k = 6 a list behind `[LIVM]-x-x-[DE]-A` is a
 * block of 4*20*20*2*1 entries, not 21^5.  Untracked blocks have the full geometry (every code at every position). */
#define TXQ_PROGRAM_TRACKED_BIT 0x80000000u
#define TXQ_DENSE_TRACKED 1u
/* A tracked STEP with TXQ_DENSE_NOPROBE rolls its residues in WITHOUT a probe: dst[(x1 .. x_{k-2}, r)] |= src[(a, x1 .. x_{k-2})].
 * That is the collector's update_path for states that have not seen k - 1 residues yet (reference
 * include/otf_collector.h:247-259: the k-mer is still being filled, nothing is looked up): the lists behind the first
 * residues of a motif — 20^4 states behind `C-x(4)` at k = 6 — are blocks too (their leading suffix positions hold the
 * one code 0, as the k-mer value of such a state has zeros there; a list keeps states of different lengths in
 * different blocks, the product never merges them). */
#define TXQ_DENSE_NOPROBE 2u

typedef struct {
    uint32_t kind;
    uint32_t dst;     /* ZERO, STEP, FILL: first slot of the block (dense slot id); REDUCE: any writable slot */
    uint32_t src;     /* STEP, REDUCE: first slot of the block read; FILL: the ordinary slot whose mask is spread;
                         tracked ZERO: the block's capacity in entries                                            */
    uint32_t r_mask;  /* STEP: bit c set <=> residue code c is rolled in                                    */
    uint32_t shape[TXQ_DENSE_MAX_POSITIONS]; /* per suffix position (oldest first): codes worth visiting    */
    uint32_t reserved; /* bit 0: TXQ_DENSE_TRACKED; bit 1: TXQ_DENSE_NOPROBE */
} txq_dense_op; /* 64 bytes */

typedef struct {
    txq_blob_header_v2 v2;   /* version = TXQ_PROGRAM_VERSION_DENSE; txq_program_v2.reserved = the program's
                                dense blocks (ids 0 .. n-1 are in use; bit 31: TXQ_PROGRAM_TRACKED_BIT)                                            */
    uint64_t dense_offset;   /* txq_dense_op[n_dense]                                                       */
    uint32_t n_dense;
    uint32_t k;              /* k-mer length                                                                */
    uint32_t bits;           /* bits per residue code in a k-mer value (5 peptides, 2 DNA)                  */
    uint32_t alphabet;       /* A: residue codes are 0 .. A-1                                               */
    uint32_t canonical;      /* 1: probe min(forward, reverse complement) (DNA, code ^ 2 = complement)      */
    uint32_t reserved;
} txq_blob_header_v3;

#endif /* TXQ_PROGRAM_H */

/*
 * include/brx.h -- C-ABI of libbrx_hip.so, the MI355X-native replacement for Badread's per-read
 * hot path (badread/simulate.py:63-86 -> build_fragment :91, sequence_fragment :256,
 * qscore_model.get_qscores qscore_model.py:32, and the edlib.align calls they make).
 *
 * Badread has no plugin/FFI interface of its own (it is pure Python; its one native dependency is
 * the `edlib` wheel).  This header therefore declares the boundary a maintainer would bind with
 * ctypes from badread/simulate.py -- see INTEGRATION.md for the stub.  Rules:
 *   - plain C, plain pointers and sizes, no torch / HIP types in any signature;
 *   - every pointer named d_* is a DEVICE pointer owned by the caller (the Python host holds them
 *     as PyTorch-ROCm tensors); the library never frees caller memory and never allocates output;
 *   - scratch is one caller-provided device arena; if it is too small a call fails with
 *     BRX_E_SCRATCH and brx_scratch_needed() tells how much the same call wants;
 *   - no exceptions cross the ABI: int status + brx_last_error();
 *   - a context belongs to one device and one host thread; work is enqueued on the caller's
 *     HIP stream (pass torch.cuda.current_stream().cuda_stream, or NULL for the default stream).
 *
 * The same descriptor structs (with HOST pointers) parameterise the CPU oracle under oracle/,
 * which is test infrastructure only.
 */
#ifndef BRX_H
#define BRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status codes */
enum {
    BRX_OK = 0,
    BRX_E_ARG = -1,        /* bad argument                                                    */
    BRX_E_HIP = -2,        /* a HIP runtime call failed (message in brx_last_error)            */
    BRX_E_SCRATCH = -3,    /* scratch arena too small; see brx_scratch_needed()                */
    BRX_E_OUTPUT = -4,     /* output buffer too small; see brx_output_needed()                 */
    BRX_E_NOFRAG = -5,     /* a read failed 1000 times to draw a fragment (simulate.py:159-165) */
    BRX_E_STATE = -6,      /* reference / models / params not set                             */
    BRX_E_INTERNAL = -7    /* a kernel flagged an internal inconsistency                      */
};

/* per-read status bits written to brx_read_stats.status */
enum {
    BRX_RS_NOFRAG = 1u,       /* get_fragment gave up after 1000 tries                        */
    BRX_RS_TOO_MANY_SEGS = 2u,/* more than 4096 base segments in one read AND the batch's overflow lists used up (see brx_kernels.h) */
    BRX_RS_BAND = 4u,         /* alignment failed: score above its proven bound (bug) or a band of more than 7.3 M rows */
    BRX_RS_QMISS = 8u,        /* qscore fallback reached a 1-op cigar absent from the model   */
    BRX_RS_EMPTY = 16u        /* read trimmed to zero length: skipped, like simulate.py:70-71  */
};

/* base codes: 0..3 = A,C,G,T; 4 = N; 5..15 = other symbols of this reference (sym[] gives ASCII) */
#define BRX_CODE_N 4

/* ------------------------------------------------------------------ reference (misc.py:122-153) */
typedef struct {
    uint64_t base_off;     /* index of the contig's first base in the packed array             */
    uint32_t length;       /* bases                                                            */
    uint32_t flags;        /* bit0 circular, bit1 hairpin_left, bit2 hairpin_right             */
    uint32_t name_off;     /* into names[]                                                     */
    uint32_t name_len;
} brx_contig;

typedef struct {           /* run of non-ACGT bases: packed coordinates [start,end) hold `code` */
    uint64_t start, end;
    uint32_t code;
    uint32_t pad_;
} brx_exception;

typedef struct {
    const uint32_t *d_packed;        /* 2-bit bases, 16 per word, base g at bits 2*(g%16) of word g/16 */
    uint64_t n_bases;
    const brx_contig *d_contigs;     /* [n_contigs]                                              */
    uint32_t n_contigs;
    const brx_exception *d_exceptions; /* sorted by start, non-overlapping                        */
    uint32_t n_exceptions;
    const uint8_t *d_names;          /* concatenated contig names (ASCII)                        */
    uint32_t names_len;
    uint8_t sym[16];                 /* code -> ASCII                                            */
    uint8_t comp[16];                /* code -> complement code (misc.py:56-67)                  */
    const double *d_cum_weight;      /* [n_contigs] running sum of depth*length (simulate.py:118-121) */
    double total_weight;
} brx_reference;

/* ------------------------------------------------------------------ error model (error_model.py:86-229)
 * Flattened by the host from the `kmer,p;alt,p;...` file after align_kmers():
 *   row r = 2-bit value of the k-mer (first base most significant); alts of row r are
 *   [row_off[r], row_off[r+1]); an absent row (error_model.py:143) has an empty range.
 *   thr[a]   cumulative probability of alts row_off[r]..a as a 32-bit fixed-point threshold:
 *            alt a is chosen iff it is the first with draw < thr[a]; if none is and the row's last
 *            thr is 0xFFFFFFFF the last alt is chosen (rows with sum p >= 1), otherwise the
 *            remainder goes to add_one_random_change (error_model.py:151-158).
 *   desc[a]  offset o into pool: pool[o], pool[o+1] = bitmask of positions whose string differs
 *            from the k-mer base; pool[o+2 .. o+2+k) = per-position string lengths;
 *            then the alt's characters (base codes), in order.
 *   self_thr[r] = thr of alt 0 (the unchanged k-mer), 0 for an absent row: one compare rejects
 *            the ~93 % of draws that leave the k-mer unchanged (simulate.py:300).
 * pool[0..528) is a fixed preamble used for add_one_random_change results:
 *   pool[c] = c for c<16 (single chars), pool[16 + 2*(16x+y)] = x,y (all two-char strings).     */
#define BRX_POOL_PREAMBLE 528
typedef struct {
    int32_t k;                  /* k-mer size; 1 for the 'random' model                          */
    int32_t type;               /* 0 = 'random' (no table), 1 = table                            */
    uint32_t n_rows;            /* 4^k                                                           */
    uint32_t n_alts;
    uint32_t pool_len;
    uint32_t pad_;
    const uint32_t *d_row_off;  /* [n_rows+1]                                                    */
    const uint32_t *d_self_thr; /* [n_rows]                                                      */
    const uint32_t *d_thr;      /* [n_alts]                                                      */
    const uint32_t *d_desc;     /* [n_alts]                                                      */
    const uint8_t *d_pool;      /* [pool_len]                                                    */
    /* The same table laid out for ONE load per step of the lookup (round 4; required when type = 1).  The proposal rounds of
     * the mutate loop are chains of dependent L2 round trips: self_thr -> row_off -> five probes of a binary search -> desc ->
     * pool was ten deep; with these it is three (row entry -> a block of eight thresholds -> the alternative's descriptor).
     *   d_rowx[2 r], [2 r + 1] = self_thr[r], row_off[r] for r <= n_rows (self_thr[n_rows] = 0): a 16-byte load at row r
     *            also holds row_off[r + 1]; 16 bytes of padding behind the last pair.
     *   d_thr    must be readable 8 entries past n_alts (zeros): thresholds are scanned in blocks of eight.
     *   d_altx[4 a .. 4 a + 3] = desc[a]; diff | flags << 16; lengths of positions 0-7 in 4 bits each; 0.
     *            flags bit 0: a length above 15 or k > 8 -- the kernel then reads the lengths from the pool.
     * badread_amd.error_model.derive_lookup_tables builds both from the arrays above.                */
    const uint32_t *d_rowx;     /* [2 (n_rows + 1) + 4]                                          */
    const uint32_t *d_altx;     /* [4 n_alts]                                                    */
} brx_error_model;

/* ------------------------------------------------------------------ qscore model (qscore_model.py:178-287)
 * A cigar window of w non-D ops (w odd) is keyed as
 *   key = (w << 56) | sum_i op_i << (bits*i)   with op codes 0 '=',1 'X',2 'I' in 2 bits and the
 *   D-run length between op i and op i+1 in `gap_bits` bits placed after op i
 * (see brx_qs_key in badread_amd/csrc).  A run longer than the field holds is encoded as all-ones,
 * which no table row uses, so it misses and falls back exactly like the reference's string lookup
 * (qscore_model.py:278-286).  Open-addressing hash: slot = hash(key) & (hash_size-1), linear probe,
 * empty slot = key ~0.                                                                           */
typedef struct {
    int32_t k;                  /* kmer_size: largest non-D window length in the model (odd)      */
    int32_t gap_bits;           /* 3 or 4                                                        */
    uint32_t hash_size;         /* power of two                                                  */
    uint32_t n_rows;
    uint32_t n_entries;
    uint32_t pad_;
    const uint64_t *d_hash_key; /* [hash_size]                                                   */
    const uint32_t *d_hash_row; /* [hash_size]                                                   */
    const uint32_t *d_row_off;  /* [n_rows+1] into thr/score                                     */
    const uint32_t *d_thr;      /* [n_entries] cumulative 32-bit thresholds; first draw < thr wins, else last */
    const uint8_t *d_score;     /* [n_entries] phred score                                       */
} brx_qscore_model;

/* ------------------------------------------------------------------ simulation parameters
 * The derived fields argparse validation leaves on `args` (badread/__main__.py:239-313).         */
typedef struct {
    /* fragment lengths, fragment_lengths.py:47-64 */
    double frag_mean, frag_stdev, gamma_k, gamma_t;
    /* identities, identities.py:76-103.  mode 0: constant id_max; 1: id_max*beta(id_a,id_b);
       2: 1-10^(-normal(id_a,id_b)/10) */
    int32_t identity_mode;
    int32_t pad0_;
    double id_a, id_b, id_max;
    /* adapters, simulate.py:361-387: rates/amounts as fractions */
    double start_rate, start_amount, end_rate, end_amount;
    const uint8_t *d_start_adapter;   /* base codes */
    const uint8_t *d_end_adapter;
    uint32_t start_adapter_len, end_adapter_len;
    /* problems: fractions (simulate.py:101,172-173) */
    double junk_rate, random_rate, chimera_rate;
    /* glitches (simulate.py:459-482) */
    double glitch_rate, glitch_size, glitch_skip;
} brx_sim_params;

/* ------------------------------------------------------------------ per-read results */
typedef struct {
    uint32_t status;           /* BRX_RS_* bits                                                 */
    uint32_t frag_len;         /* error-free_length: len(fragment) before padding (simulate.py:74) */
    uint32_t seq_len;          /* length= : trimmed read length                                 */
    uint32_t n_cols;           /* columns of the final alignment (pads included)                */
    uint32_t n_match;          /* '=' columns: read_identity = n_match / n_cols (misc.py:228-240) */
    uint32_t padded_len;       /* bases of the read before trimming the pads (what get_qscores scored);
                                  edit distance = n_cols - n_match                                */
    uint32_t loop_count;       /* k-mer draws consumed by the mutate loop                       */
    uint32_t change_count;     /* positions changed                                             */
    uint32_t n_alignments;     /* in-loop identity alignments (simulate.py:325-346)             */
    uint32_t rec_len;          /* bytes of this read's FASTQ record (0 if skipped)              */
    uint64_t rec_off;          /* offset of the record in the output buffer                     */
    double target_identity;
    double qerr_sum;           /* sum over read bases of 10^(-q/10) (qscore_model.py:71-73)     */
} brx_read_stats;

typedef struct brx_ctx brx_ctx;

/* ------------------------------------------------------------------ entry points */
int brx_create(int device_id, brx_ctx **out);
void brx_destroy(brx_ctx *ctx);
const char *brx_last_error(const brx_ctx *ctx);   /* ctx == NULL: why the last brx_create failed */
const char *brx_version(void);

int brx_set_reference(brx_ctx *ctx, const brx_reference *ref);
int brx_set_error_model(brx_ctx *ctx, const brx_error_model *em);
int brx_set_qscore_model(brx_ctx *ctx, const brx_qscore_model *qm);
int brx_set_params(brx_ctx *ctx, const brx_sim_params *p);
int brx_set_scratch(brx_ctx *ctx, void *d_scratch, size_t bytes);
size_t brx_scratch_needed(const brx_ctx *ctx);
size_t brx_output_needed(const brx_ctx *ctx);

/* replaces the body of the `while total_size < target_size` loop (simulate.py:63-86) for reads
 * [first_read, first_read+n_reads): FASTQ records are written back-to-back in read order into
 * d_out; d_stats[i] describes read first_read+i.  *out_bytes receives the bytes written.
 * Synchronous with respect to the host on return (small size read-backs happen inside).         */
int brx_simulate_batch(brx_ctx *ctx, uint64_t seed, uint64_t first_read, uint32_t n_reads,
                       uint8_t *d_out, size_t out_cap, brx_read_stats *d_stats,
                       size_t *out_bytes, void *hip_stream);

/* replaces badread.simulate.sequence_fragment (simulate.py:256-358) for caller-supplied fragments:
 * fragment i is d_frags[frag_off[i] .. frag_off[i+1]) as base codes, with target identity
 * d_target[i]; read index first_read+i keys the random streams.  Output per fragment i:
 * sequence codes then phred+33 bytes, each seq_len long, at d_out + stats[i].rec_off.           */
int brx_sequence_fragments(brx_ctx *ctx, uint64_t seed, uint64_t first_read, uint32_t n_frags,
                           const uint8_t *d_frags, const uint64_t *d_frag_off,
                           const double *d_target, uint8_t *d_out, size_t out_cap,
                           brx_read_stats *d_stats, size_t *out_bytes, void *hip_stream);

/* replaces edlib.align(query, target, task='path') (simulate.py:330,340; qscore_model.py:37;
 * error_model.py:202) for a batch: pair i aligns query d_queries[q_off[i]..q_off[i+1]) against
 * target d_targets[t_off[i]..t_off[i+1]) (any byte alphabet; equality is byte equality).
 * k_hint[i] >= 0 is a proven upper bound on the distance (no band search), -1 = unknown (band
 * doubling from 64, like edlib).  Writes the edit distance to d_dist[i], the number of alignment
 * columns to d_ncols[i], the number of '=' columns to d_nmatch[i] and, if d_ops != NULL, the
 * per-column ops (0 '=',1 'X',2 'I',3 'D', forward order; 'I' = byte present in the query only)
 * at d_ops + ops_off[i] (capacity qlen+tlen).                                                    */
int brx_align_batch(brx_ctx *ctx, uint32_t n_pairs,
                    const uint8_t *d_queries, const uint64_t *d_q_off,
                    const uint8_t *d_targets, const uint64_t *d_t_off, const int32_t *d_k_hint,
                    int32_t *d_dist, uint32_t *d_ncols, uint32_t *d_nmatch,
                    uint8_t *d_ops, const uint64_t *d_ops_off, void *hip_stream);

/* Device time (ms) of each pipeline stage of the last brx_simulate_batch / brx_sequence_fragments
 * call, from HIP events recorded on the launch stream immediately before and after the stage's
 * kernels (host work between stages is not included):
 *   PLAN    k_plan_count, k_scan_plan, k_plan_fill (includes one small size read-back)
 *   BUILD   k_build (+ k_copy_frags), k_order
 *   MUTATE  k_mut_fill, k_mut_lanes, k_mutate_seg, k_mut_epilogue (+ k_mutate for overflow reads) -- or, under BRX_MUTATE_PASSES=1,
 *           all passes of {k_mut_apply, k_mut_post, k_pass_lists, k_win_lane, k_win_wave} including the host round trips between
 *           them; brx_last_mutate_passes() gives the count of launches / passes
 *   SCAN    k_scan_mut
 *   FINAL   the whole final stage: k_fin_join, then k_fin_align<...> for every band class (two streams) +
 *           k_fin_qscore, one set of launches per scratch chunk (and per phase: brx_last_window_misses)
 *   EMIT    k_recsize, k_scan_rec, k_emit, k_stats (includes one small size read-back)
 *   ALIGN1  average duration of ONE launch of k_fin_align<1,1,1> (the largest single kernel) over the chunks
 *   QSCORE  average duration of ONE launch of k_fin_qscore for the one- and two-word band classes over the chunks */
enum { BRX_STAGE_PLAN = 0, BRX_STAGE_BUILD = 1, BRX_STAGE_MUTATE = 2, BRX_STAGE_SCAN = 3,
       BRX_STAGE_FINAL = 4, BRX_STAGE_EMIT = 5, BRX_STAGE_ALIGN1 = 6, BRX_STAGE_QSCORE = 7, BRX_STAGE_COUNT = 8 };
int brx_last_stage_ms(const brx_ctx *ctx, float ms[BRX_STAGE_COUNT]);
/* Shader-clock cycles (s_memtime) each read of the last pipeline call spent in the two heavy
 * kernels, 8 x u64 per read, copied to HOST memory h_out (valid until the next call on ctx):
 *   [0] k_mutate_seg / k_mut_post total over all passes  [1] passes (alignments) of the read  [2] 1 if the final traceback left the stored window
 *   [3] k_fin_align total   [4] final alignment forward    [5] final traceback   [6] k_fin_qscore
 *   [7] words per lane (G) of the final alignment's band geometry in bits 0-15; bit 16: the read was aligned as one of four per
 *       wave (k_fin_quad), bit 17: one read per lane (k_fin_lanes) -- a read repeated in the second phase carries k_fin_align's word */
int brx_last_read_cycles(brx_ctx *ctx, uint64_t *h_out, uint32_t n_reads);
/* number of scratch chunks (sets of final-stage launches) and of mutate passes of the last call */
/* With BRX_PROFILE=1 in the environment at brx_create the mutate kernels time their phases (shader clock, per read,
 * summed over passes), 8 x u64 per read to HOST memory h_out:
 *   [0] proposals (draws, k-mer bytes, table lookups)  [1] applying survivors  [2] parking a window (join, copies)
 *   [3] in-place window alignment  [4] everything else  [5] / [6] forward / traceback share of [3]  [7] unused
 * Zeros without BRX_PROFILE (the default kernels do not contain the clock reads). */
int brx_last_phase_cycles(brx_ctx *ctx, uint64_t *h_out, uint32_t n_reads);
uint32_t brx_last_mutate_passes(const brx_ctx *ctx);
/* Per-kernel launch timing, opt-in (brx_set_kernel_timing(ctx, 1); bench.py and the profiling tools use it): every
 * launch of the kernels below is bracketed by two HIP events ON THE STREAM THE KERNEL IS LAUNCHED ON, and
 * brx_last_kernel_stats() returns, for the last brx_simulate_batch / brx_sequence_fragments call, the number of
 * launches of each kernel, the sum of their event durations and the bases of the reads each kernel handled
 * (fragment bases incl. pads: a read counts ONCE per kernel class however many passes it took part in).  The
 * average launch duration ms / launches is what a rocprofv3 kernel trace reports for the same kernel. */
enum { BRX_KERN_PLAN = 0,        /* k_plan_count + k_scan_plan + k_plan_fill                                   */
       BRX_KERN_BUILD = 1,       /* k_build                                                                    */
       BRX_KERN_MUTATE_SEG = 2,  /* k_mut_apply: the sequential half of the bulk passes, one read per lane (round 6)  */
       BRX_KERN_MUTATE_RUN = 3,  /* k_mutate_seg: reads run to completion with in-place window alignments        */
       BRX_KERN_WIN_LANE = 4,    /* k_win_lane: parked window alignments of a bulk pass, one window per lane    */
       BRX_KERN_WIN_WAVE = 5,    /* k_win_wave                                                                 */
       BRX_KERN_FIN_JOIN = 6,
       BRX_KERN_FIN_ALIGN1 = 7,  /* k_fin_align<1,1,1>                                                         */
       BRX_KERN_FIN_ALIGN2 = 8,  /* k_fin_align<2,2,2>                                                         */
       BRX_KERN_FIN_ALIGN4 = 9,  /* k_fin_align<4,4,4>                                                         */
       BRX_KERN_FIN_ALIGN16 = 10,/* k_fin_align<16,8,65535>: 8 or more band words per lane                  */
       BRX_KERN_FIN_QSCORE = 11,
       BRX_KERN_EMIT = 12,       /* k_recsize + k_scan_rec + k_emit + k_stats                                  */
       BRX_KERN_FIN_LANES = 13,  /* k_fin_lanes: final alignments with a band of up to four blocks, one read per lane */
       BRX_KERN_FIN_QUAD1 = 14,  /* k_fin_quad<1>: bands of up to 13 superblocks of one word, four reads per wave      */
       BRX_KERN_MUT_POST = 15,   /* k_mut_post + k_pass_lists + k_mut_epilogue: parking, proposals ahead, lists, epilogues of the bulk passes (round 6) */
       BRX_KERN_COUNT = 16 };
typedef struct {
    uint32_t launches;
    float ms;                  /* sum of the launches' event durations                                          */
    double bases;              /* fragment bases (RS.n) of the reads this kernel class handled in the call      */
} brx_kernel_stat;
int brx_set_kernel_timing(brx_ctx *ctx, int on);
int brx_last_kernel_stats(const brx_ctx *ctx, brx_kernel_stat out[BRX_KERN_COUNT]);
/* Traceback slabs (= persistent waves of the final align kernels, summed over band classes and the two sets) of the last
 * call: one per read of a class up to the class's wave limit; fewer when the scratch arena could not hold that many. */
uint32_t brx_last_final_launches(const brx_ctx *ctx);
/* Reads of the last call whose final traceback asked for a cell outside the windowed traceback store and were
 * aligned a second time with the full store (environment BRX_TB_WINDOW: window height in
 * sqrt(edit bound) units, default 2, 0 = always the full store).  Results do not depend on it. */
uint32_t brx_last_window_misses(const brx_ctx *ctx);

/* ------------------------------------------------------------------ model builders
 * The counting loops of make_error_model (error_model.py:31-83) and make_qscore_model (qscore_model.py:78-162) over a set
 * of read-to-reference alignments.  The host parses FASTA / FASTQ / PAF exactly as the reference does and ships, as device
 * arrays: the aligned slices of the reads and their qualities, the reference slices (reverse-complemented for '-' strand
 * alignments), and the CIGAR parts ('M' 0, 'I' 1, 'D' 2; reversed for '-' strand, alignment.py:61-64) with the prefix sums
 * of their column / read / reference offsets.  brx_model_count expands the CIGARs into the gapped column arrays
 * (alignment.py:101-132) and counts every sliding window into the caller's hash table (keys preset to ~0, counts to 0,
 * first to ~0):
 *   kind 0, error model:   key = ref k-mer (2k bits, first base most significant) | read k-mer length << 2k
 *                                | read k-mer (2 bits per base, first base least significant) << (2k + 5)
 *   kind 1, qscore model:  key = quality of the middle base (7 bits) | window size index (k_size - 1) / 2 << 7
 *                                | ops << 11, ops = 2 bits per read base (0 '=', 1 'X', 2 'I'), each but the first
 *                                preceded by 4 bits holding the (collapsed) length of the deletion run before it
 *   first[slot] = the earliest (alignment, [size index,] window start) that produced the key: Python dictionaries and its
 *   stable sort order equal counts by first insertion.
 * Windows a key cannot hold (error model: a read k-mer of more than 21 bases; qscore model: more than 52 key bits, a
 * window opening with deletion columns) are appended to `spill` for the host to count: kind 0 as (alignment << 32 | start column), kind 1 as
 * (alignment << 36 | window size index << 32 | start column).
 * Returns BRX_E_OUTPUT when the table is full.                                                                        */
typedef struct {
    uint32_t n_align, k, max_del, n_ksizes;
    uint64_t n_cols;
    const uint8_t *d_seq, *d_qual, *d_ref;
    const uint8_t *d_part_type;
    const uint32_t *d_part_len;
    const uint64_t *d_part_col, *d_part_read, *d_part_ref;
    const uint64_t *d_align_part_off, *d_align_col_off;
    uint8_t *d_rcol, *d_qcol, *d_fcol;
    uint64_t *d_keys; uint32_t *d_counts; uint64_t *d_first;
    uint64_t table_mask;
    uint32_t *d_flags;                 /* [4], preset to 0: [0] table full, [1] / [2] spilled windows of kind 0 / 1 */
    uint64_t *d_spill; uint32_t spill_cap;
} brx_model_job;
int brx_model_count(brx_ctx *ctx, int kind, const brx_model_job *job, void *hip_stream);

/* ---- output stage: the FASTQ bytes of a batch as gzip members, made on the device ----
 * Replaces the `| gzip` the reference's users put behind /root/reference/badread/simulate.py:73-82 (print of the
 * record text).  d_in: n_bytes of text in device memory.  d_block_off: n_blocks + 1 ascending byte offsets (device
 * memory; first 0, last n_bytes, no block longer than 128 MB) -- every block becomes one gzip member with its own
 * Huffman code -- or NULL (n_blocks 0): 64 KB blocks.  d_out: device buffer of at least brx_gzip_device_bound(n_bytes,
 * n_blocks) bytes (4-byte aligned); d_scratch: brx_gzip_device_scratch(n_bytes, n_blocks) bytes of device memory.
 * Writes the members back to back (dynamic Huffman coding of the bytes, no match search: gzip readers take them as one
 * stream); *out_bytes = their total size.  Synchronous on `hip_stream`.  BRX_E_OUTPUT / BRX_E_SCRATCH when a buffer is
 * too small. */
size_t brx_gzip_device_bound(size_t n_bytes, uint32_t n_blocks);
size_t brx_gzip_device_scratch(size_t n_bytes, uint32_t n_blocks);
int brx_gzip_device(brx_ctx *ctx, const void *d_in, size_t n_bytes, const uint64_t *d_block_off, uint32_t n_blocks, void *d_out,
                    size_t out_cap, void *d_scratch, size_t scratch_bytes, size_t *out_bytes, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* BRX_H */

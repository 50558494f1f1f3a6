/*
 * brx_host.h -- host-side (CPU) helpers of libbrx_host.so: the steps either side of the GPU hot path that the
 * reference does in Python and that dominate start-up on large genomes.  Plain C ABI,
 * no HIP, no torch; built with g++ + zlib by `python -m badread_amd.build`.
 *
 * brx_fasta_* replaces misc.load_fasta (/root/reference/badread/misc.py:122-153) followed by the Python packing
 * of badread_amd/reference.py: FASTA or FASTA.gz -> the arrays brx_reference (include/brx.h) points at.
 *   - header parsing exactly as the reference: name = first token after '>', depth=X / circular=true /
 *     hairpin_left=true / hairpin_right=true looked up in the lower-cased header, bad depth -> 1.0,
 *     a repeated name keeps its first position and takes the last sequence;
 *   - lines are stripped, blank lines skipped, '\r' and '\n' both end a line (text mode), bases upper-cased;
 *   - alphabet: A,C,G,T = 0..3, N = 4, then every other byte value that occurs (ascending), each followed by its
 *     IUPAC complement if new; more than 16 symbols is an error;
 *   - bases are packed 16 per 32-bit word across contig boundaries; bases outside ACGT are stored as 0 bits
 *     plus a sorted list of maximal same-symbol runs in packed coordinates.
 * The packed form can be saved next to the FASTA and reloaded (brx_fasta_save / brx_fasta_load): the sidecar
 * records the source's size and mtime and is ignored when they no longer match.
 *
 * brx_gzip_* is the host output stage: multi-threaded gzip of the FASTQ bytes (see below).
 */
#ifndef BRX_HOST_H
#define BRX_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "brx.h"        /* brx_contig, brx_exception */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct brx_fasta brx_fasta;

typedef struct {
    uint64_t n_bases;              /* sum of contig lengths                                          */
    uint64_t n_words;              /* packed words: (n_bases + 15) / 16 + 1                           */
    uint32_t n_contigs;
    uint32_t n_exceptions;
    uint32_t names_len;
    uint32_t n_symbols;            /* codes in use (5..16)                                            */
    const uint32_t *packed;        /* [n_words]                                                       */
    const brx_contig *contigs;     /* [n_contigs]                                                     */
    const brx_exception *exceptions; /* [n_exceptions], sorted, non-overlapping                       */
    const uint8_t *names;          /* [names_len] concatenated contig names                           */
    const double *depths;          /* [n_contigs]                                                     */
    uint8_t sym[16];               /* code -> ASCII                                                   */
    uint8_t comp[16];              /* code -> code of the complement                                  */
} brx_fasta_view;

/* 0 on success; on failure a message in err (truncated to err_cap) and a negative BRX_E_* code */
int brx_fasta_pack(const char *path, brx_fasta **out, char *err, size_t err_cap);
int brx_fasta_view_of(const brx_fasta *f, brx_fasta_view *view);
void brx_fasta_free(brx_fasta *f);

/* sidecar: save the packed form of `f` (made from `source_path`) to `sidecar_path`; load returns BRX_E_STATE when
 * the sidecar is missing, malformed, or was made from a file with another size or modification time */
int brx_fasta_save(const brx_fasta *f, const char *source_path, const char *sidecar_path, char *err, size_t err_cap);
int brx_fasta_load(const char *source_path, const char *sidecar_path, brx_fasta **out, char *err, size_t err_cap);

/* ---- output stage -------------------------------------------------------------
 * The reference prints FASTQ text (simulate.py:79-82) and its documentation pipes it through `gzip`, one core of
 * deflate (~30 MB/s) behind a simulator that now emits GB/s.  brx_gzip_parallel compresses a buffer as a sequence
 * of independent gzip members (RFC 1952 allows concatenation: `gzip -d`, zlib's gz* layer and Python's gzip module
 * read the result as one stream), `block_bytes` of input per member, `threads` members at a time.
 * Returns 0 and *out_bytes; BRX_E_OUTPUT with *out_bytes = a sufficient capacity when `cap` is too small. */
size_t brx_gzip_bound(size_t n_bytes, size_t block_bytes);
int brx_gzip_parallel(const uint8_t *in, size_t n_bytes, int level, int threads, size_t block_bytes,
                      uint8_t *out, size_t cap, size_t *out_bytes);

#ifdef __cplusplus
}
#endif
#endif /* BRX_HOST_H */

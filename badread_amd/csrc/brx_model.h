/*
 * brx_model.h -- the step BEFORE the hot path: counting the tables of an error model
 * and of a qscore model from reads aligned to a reference.
 *   make_error_model   /root/reference/badread/error_model.py:31-83
 *   make_qscore_model  /root/reference/badread/qscore_model.py:78-162
 *   align_sequences    /root/reference/badread/alignment.py:101-132 (CIGAR -> gapped read / quality / reference strings)
 * The reference walks every alignment in Python, one sliding window at a time, building strings and dictionaries.
 * Here the host ships the CIGAR parts with their column / read / reference offsets (prefix sums) and the kernels do the
 * rest: one thread per alignment COLUMN expands the CIGAR (k_mb_expand), one thread per window start slides the k-mer
 * window and counts into an open-addressing hash table in global memory (k_mb_error, k_mb_qscore).  Every key also
 * records the earliest (alignment, window) that produced it: Python's dictionaries iterate in first-insertion order and
 * its sort is stable, so that rank is what orders equal counts in the model file.
 * Memory-bound byte work: one read of the three column arrays per window start, one atomic per counted window.
 */
#ifndef BRX_MODEL_H
#define BRX_MODEL_H

#define BRX_MB_GAP 0u                          /* gap column (the reference's '-' / ' ') */
#define BRX_MB_EMPTY 0xFFFFFFFFFFFFFFFFull
#define BRX_MB_MAX_READ_KMER 21                /* read k-mer characters a 64-bit key holds (error model) */

/* flags[0] table full  [1] windows whose read k-mer was longer than a key holds (the host counts them itself)
   [2] qscore windows that start with a deletion column / runs the key cannot hold (likewise) */
struct BrxMbJob {
    uint32_t n_align; uint32_t k; uint32_t max_del; uint32_t n_ksizes;
    uint64_t n_cols;
    const uint8_t *seq, *qual, *ref;            /* read bases / qualities (per alignment slice), reference slices (strand applied) */
    const uint8_t *part_type;                   /* 0 'M', 1 'I', 2 'D' */
    const uint32_t *part_len;
    const uint64_t *part_col, *part_read, *part_ref;   /* first column / read offset / reference offset of every part (global) */
    const uint64_t *align_part_off;             /* [n_align + 1] into the part arrays   */
    const uint64_t *align_col_off;              /* [n_align + 1] into the column arrays */
    uint8_t *rcol, *qcol, *fcol;                /* [n_cols] gapped read / quality / reference */
    uint64_t *keys; uint32_t *counts; uint64_t *first;  /* hash table: key, count, earliest rank */
    uint64_t table_mask;
    uint32_t *flags;
    uint64_t *spill; uint32_t spill_cap;        /* (alignment << 32 | start column) of windows the keys cannot hold */
};

__device__ __forceinline__ uint32_t mb_find_align(const BrxMbJob &j, uint64_t col) {
    uint32_t lo = 0, hi = j.n_align;
    while (lo + 1 < hi) { const uint32_t mid = (lo + hi) >> 1; if (j.align_col_off[mid] <= col) lo = mid; else hi = mid; }
    return lo;
}

__global__ void __launch_bounds__(256) k_mb_expand(BrxMbJob j) {
    const uint64_t c = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (c >= j.n_cols) return;
    const uint32_t a = mb_find_align(j, c);
    uint64_t lo = j.align_part_off[a], hi = j.align_part_off[a + 1];
    while (lo + 1 < hi) { const uint64_t mid = (lo + hi) >> 1; if (j.part_col[mid] <= c) lo = mid; else hi = mid; }
    const uint64_t o = c - j.part_col[lo];
    const uint32_t t = j.part_type[lo];
    j.rcol[c] = t != 2u ? j.seq[j.part_read[lo] + o] : (uint8_t)BRX_MB_GAP;
    j.qcol[c] = t != 2u ? j.qual[j.part_read[lo] + o] : (uint8_t)BRX_MB_GAP;
    j.fcol[c] = t != 1u ? j.ref[j.part_ref[lo] + o] : (uint8_t)BRX_MB_GAP;
}

/* 'A' 'C' 'G' 'T' -> 0..3 and the test for them, as arithmetic (a chain of comparisons becomes a jump tree per column) */
__device__ __forceinline__ uint32_t mb_code(uint32_t ch) { return ((ch >> 1) ^ (ch >> 2)) & 3u; }
__device__ __forceinline__ bool mb_acgt(uint32_t ch) { return (ch >> 5) == 2u && ((0x0010008Au >> (ch & 31u)) & 1u) != 0u; }

__device__ inline void mb_count(const BrxMbJob &j, uint64_t key, uint64_t rank) {
    uint64_t slot = ((key * 0x9E3779B97F4A7C15ull) >> 20) & j.table_mask;
    for (uint64_t probe = 0; probe <= j.table_mask; ++probe) {
        unsigned long long cur = __hip_atomic_load((unsigned long long *)&j.keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == BRX_MB_EMPTY) {
            cur = atomicCAS((unsigned long long *)&j.keys[slot], (unsigned long long)BRX_MB_EMPTY, (unsigned long long)key);
            if (cur == BRX_MB_EMPTY) cur = key;
        }
        if (cur == key) {
            atomicAdd(&j.counts[slot], 1u);
            atomicMin((unsigned long long *)&j.first[slot], (unsigned long long)rank);
            return;
        }
        slot = (slot + 1) & j.table_mask;
    }
    atomicOr(&j.flags[0], 1u);
}

/* error_model.py:44-66: window [start, end) holds k reference bases; start is column 0 of the alignment or any later
   reference-base column; key = ref k-mer (2k bits) | read k-mer length (5 bits) | read k-mer (2 bits per base) */
__global__ void __launch_bounds__(256) k_mb_error(BrxMbJob j) {
    const uint64_t s = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (s >= j.n_cols) return;
    const uint32_t a = mb_find_align(j, s);
    const uint64_t c0 = j.align_col_off[a], c1 = j.align_col_off[a + 1];
    if (s != c0 && j.fcol[s] == BRX_MB_GAP) return;
    uint32_t refk = 0, nref = 0, nread = 0, first_read = 0, last_read = 0, first_ref = 0, last_ref = 0;
    uint64_t readk = 0;
    bool ok = true, too_long = false;
    uint64_t e = s;
    for (; e < c1 && nref < j.k; ++e) {
        const uint32_t f = j.fcol[e], r = j.rcol[e];
        const bool hf = f != BRX_MB_GAP, hr = r != BRX_MB_GAP;
        ok = ok && (!hf || mb_acgt(f)) && (!hr || mb_acgt(r));
        refk = hf ? ((refk << 2) | mb_code(f)) : refk;
        first_ref = (hf && nref == 0) ? f : first_ref;
        last_ref = hf ? f : last_ref;
        nref += hf ? 1u : 0u;
        const bool fits = nread < BRX_MB_MAX_READ_KMER;
        readk |= (hr && fits) ? ((uint64_t)mb_code(r) << (2 * (nread & 31u))) : 0ull;
        too_long = too_long || (hr && !fits);
        first_read = (hr && nread == 0) ? r : first_read;
        last_read = hr ? r : last_read;
        nread += hr ? 1u : 0u;
    }
    if (nref < j.k) return;                                     /* `end > len(aligned_ref_seq)`: no window left */
    if (!(nread > 1 && first_ref == first_read && last_ref == last_read && ok)) return;
    if (too_long) {
        const uint32_t at = atomicAdd(&j.flags[1], 1u);
        if (at < j.spill_cap) j.spill[at] = ((uint64_t)a << 32) | (uint64_t)(s - c0);
        return;
    }
    const uint64_t key = (uint64_t)refk | ((uint64_t)nread << (2 * j.k)) | (readk << (2 * j.k + 5));
    mb_count(j, key, ((uint64_t)a << 32) | (uint64_t)(s - c0));
}

/* qscore_model.py:103-146, one thread per (window start, odd window size): window [start, end) holds ks read bases; key =
   quality of the middle base (7 bits) | size index (4 bits) | ops, 2 bits each, with the deletion run after every op
   in 4 bits (runs are collapsed to max_del first, qscore_model.py:91-92,133) */
__global__ void __launch_bounds__(256) k_mb_qscore(BrxMbJob j) {
    const uint64_t tid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t s = tid / j.n_ksizes;
    const uint32_t ki = (uint32_t)(tid % j.n_ksizes), ks = 2u * ki + 1u;
    if (s >= j.n_cols) return;
    const uint32_t a = mb_find_align(j, s);
    const uint64_t c0 = j.align_col_off[a], c1 = j.align_col_off[a + 1];
    if (s != c0 && j.rcol[s] == BRX_MB_GAP) return;
    uint64_t ops = 0;
    uint32_t nread = 0, run = 0, q = 0;
    int shift = 0;
    bool odd = false;
    for (uint64_t e = s; e < c1 && nread < ks; ++e) {
        const uint32_t r = j.rcol[e], f = j.fcol[e];
        if (r == BRX_MB_GAP) { run += 1; continue; }
        if (nread == 0 && run) odd = true;                      /* the window starts with deletion columns: host */
        if (nread > 0) {
            const uint32_t code = run > j.max_del ? j.max_del : run;
            if (code > 15u) odd = true;
            ops |= (uint64_t)(code & 15u) << shift; shift += 4;
        }
        const uint32_t op = f == BRX_MB_GAP ? 2u : (r == f ? 0u : 1u);
        ops |= (uint64_t)op << shift; shift += 2;
        if (nread == (ks - 1) / 2) q = j.qcol[e] - 33u;
        run = 0; nread += 1;
    }
    if (nread < ks) return;
    if (odd || shift > 52 || q > 127u) {
        /* the host counts this window: the word names it completely (alignment, size index, start column), so the host never
           has to restate these rules */
        const uint32_t at = atomicAdd(&j.flags[2], 1u);
        if (at < j.spill_cap) j.spill[at] = ((uint64_t)a << 36) | ((uint64_t)ki << 32) | (uint64_t)(s - c0);
        return;
    }
    const uint64_t key = (uint64_t)q | ((uint64_t)ki << 7) | (ops << 11);
    mb_count(j, key, ((uint64_t)a << 40) | ((uint64_t)ki << 36) | (uint64_t)(s - c0));
}

#endif /* BRX_MODEL_H */

/*
 * oracle/myers_ref.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * CPU restatement of the only native arithmetic on Badread's simulate path: the unit-cost global
 * (NW) alignment that the reference obtains from the third-party `edlib` package (un-vendored,
 * un-pinned: /root/reference/requirements.txt:1, setup.py:95).  Call sites this stands in for:
 *   badread/simulate.py:330,340      in-loop identity re-estimation
 *   badread/qscore_model.py:37       final read-vs-fragment alignment
 *   badread/error_model.py:202       align_kmers at model-load time
 *   test/test_simulate.py:85         the reference's own acceptance test
 *
 * edlib's published algorithm (Myers 1999 bit-vector, Hyyro block formulation, Ukkonen band with
 * k-doubling) is restated here in two independent forms:
 *   orc_align_dp     plain O(n*m) integer DP matrix         (obviously correct, small inputs)
 *   orc_align_myers  64-bit block bit-vector, banded         (fast, any size)
 * Both return the SAME canonical optimal path, defined on the full DP matrix:
 *   walk back from the bottom-right cell; prefer UP ('I', consumes a query char) when
 *   D[i-1][j]+1 == D[i][j], else LEFT ('D', consumes a target char) when D[i][j-1]+1 == D[i][j],
 *   else DIAGONAL ('=' when the characters are equal, 'X' otherwise).
 * This is edlib's traceback priority for problems small enough that it does not switch to
 * Hirschberg.  PARITY NOTE: edlib itself is absent from this container, so tie-breaking against
 * real edlib is "parity unpinned" (SURVEY.md section 0.4); the reference's tests pin only the
 * score and the path where the optimum is unique, and those are checked in tests/.
 *
 * Op codes written to `ops` (forward order): 0 '=', 1 'X', 2 'I', 3 'D'.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

enum { OP_EQ = 0, OP_X = 1, OP_I = 2, OP_D = 3 };

static void reverse_ops(uint8_t *ops, int64_t n) {
    for (int64_t a = 0, b = n - 1; a < b; ++a, --b) { uint8_t t = ops[a]; ops[a] = ops[b]; ops[b] = t; }
}

/* ---------------------------------------------------------------- plain DP (small inputs) */
int64_t orc_align_dp(const uint8_t *q, int64_t n, const uint8_t *t, int64_t m,
                     uint8_t *ops, int64_t *n_ops) {
    int64_t W = m + 1;
    int32_t *D = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1) * (size_t)W);
    if (!D) return -2;
    for (int64_t j = 0; j <= m; ++j) D[j] = (int32_t)j;
    for (int64_t i = 1; i <= n; ++i) {
        D[i * W] = (int32_t)i;
        for (int64_t j = 1; j <= m; ++j) {
            int32_t best = D[(i - 1) * W + (j - 1)] + (q[i - 1] != t[j - 1]);
            int32_t up = D[(i - 1) * W + j] + 1;
            int32_t left = D[i * W + (j - 1)] + 1;
            if (up < best) best = up;
            if (left < best) best = left;
            D[i * W + j] = best;
        }
    }
    int64_t dist = D[n * W + m];
    if (ops) {
        int64_t i = n, j = m, k = 0;
        while (i > 0 || j > 0) {
            int32_t cur = D[i * W + j];
            if (i > 0 && D[(i - 1) * W + j] + 1 == cur) { ops[k++] = OP_I; --i; }
            else if (j > 0 && D[i * W + (j - 1)] + 1 == cur) { ops[k++] = OP_D; --j; }
            else { ops[k++] = (q[i - 1] == t[j - 1]) ? OP_EQ : OP_X; --i; --j; }
        }
        reverse_ops(ops, k);
        if (n_ops) *n_ops = k;
    }
    free(D);
    return dist;
}

/* ---------------------------------------------------------------- banded block Myers */
typedef struct { uint64_t P, M; } PM;

typedef struct {
    int64_t n, m;
    int64_t *col_off;   /* [m+2] offset of column j's first stored block */
    int32_t *col_blo;   /* [m+1] first block in band at column j */
    int32_t *col_bhi;   /* [m+1] last block in band at column j  */
    PM *pm;             /* stored vertical deltas, after the column update */
    int32_t *score;     /* stored score at the bottom row of each block    */
} Band;

static inline int popc64(uint64_t x) { return __builtin_popcountll(x); }

/* value of DP cell (i,j), i in 1..n, j in 1..m, from the stored band; INT32_MAX/2 if not stored */
static inline int32_t cell_value(const Band *B, int64_t i, int64_t j) {
    if (i == 0) return (int32_t)j;
    if (j == 0) return (int32_t)i;
    int32_t b = (int32_t)((i - 1) >> 6);
    if (b < B->col_blo[j] || b > B->col_bhi[j]) return INT32_MAX / 2;
    int64_t idx = B->col_off[j] + (b - B->col_blo[j]);
    int r = (int)((i - 1) & 63);
    uint64_t mask = (r == 63) ? 0ULL : (~0ULL << (r + 1));
    return B->score[idx] - popc64(B->pm[idx].P & mask) + popc64(B->pm[idx].M & mask);
}

/* one attempt with threshold k; returns distance if <= k, -1 if the band was too narrow */
static int64_t myers_attempt(const uint8_t *q, int64_t n, const uint8_t *t, int64_t m, int64_t k,
                             const uint64_t *peq, const int *symid, int64_t nb,
                             uint8_t *ops, int64_t *n_ops) {
    int64_t dend = n - m;
    int64_t adend = dend < 0 ? -dend : dend;
    if (adend > k) return -1;
    int64_t half = (k - adend) / 2;
    int64_t dlo = (dend < 0 ? dend : 0) - half;
    int64_t dhi = (dend > 0 ? dend : 0) + half;

    Band B; B.n = n; B.m = m;
    B.col_off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m + 2));
    B.col_blo = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m + 1));
    B.col_bhi = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m + 1));
    int64_t total = 0;
    for (int64_t j = 1; j <= m; ++j) {
        int64_t rlo = j + dlo; if (rlo < 1) rlo = 1;
        int64_t rhi = j + dhi; if (rhi > n) rhi = n;
        B.col_blo[j] = (int32_t)((rlo - 1) >> 6);
        B.col_bhi[j] = (int32_t)((rhi - 1) >> 6);
        B.col_off[j] = total;
        total += B.col_bhi[j] - B.col_blo[j] + 1;
    }
    B.col_off[m + 1] = total;
    B.pm = (PM *)malloc(sizeof(PM) * (size_t)(total ? total : 1));
    B.score = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total ? total : 1));

    uint64_t *P = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nb);
    uint64_t *M = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nb);
    int32_t *S = (int32_t *)malloc(sizeof(int32_t) * (size_t)nb);
    /* column 0: D[i][0] = i for the rows the band covers there */
    int64_t r0 = dhi < 1 ? 1 : dhi; if (r0 > n) r0 = n;
    int32_t bhi_prev = (int32_t)((r0 - 1) >> 6);
    for (int32_t b = 0; b <= bhi_prev; ++b) { P[b] = ~0ULL; M[b] = 0; S[b] = 64 * (b + 1); }

    for (int64_t j = 1; j <= m; ++j) {
        int32_t blo = B.col_blo[j], bhi = B.col_bhi[j];
        const uint64_t *peq_c = peq + (size_t)symid[t[j - 1]] * (size_t)nb;
        int hin = 1;            /* D[0][j]-D[0][j-1] = +1, and the same upper bound above the band */
        int prev_hout = 0;      /* hout of block b-1 in THIS column (0 if it was not computed)   */
        int prev_done = 0;
        for (int32_t b = blo; b <= bhi; ++b) {
            if (b > bhi_prev) {             /* block enters the band: cells assumed +1 per row */
                int32_t above_prev = (b == 0) ? (int32_t)(j - 1)
                                              : (prev_done ? S[b - 1] - prev_hout : S[b - 1]);
                P[b] = ~0ULL; M[b] = 0; S[b] = above_prev + 64;
            }
            uint64_t Eq = peq_c[b], Pv = P[b], Mv = M[b];
            uint64_t Xv = Eq | Mv;
            if (hin < 0) Eq |= 1ULL;
            uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            uint64_t Ph = Mv | ~(Xh | Pv);
            uint64_t Mh = Pv & Xh;
            int hout = (int)(Ph >> 63) - (int)(Mh >> 63);
            Ph <<= 1; Mh <<= 1;
            if (hin < 0) Mh |= 1ULL; else if (hin > 0) Ph |= 1ULL;
            P[b] = Mh | ~(Xv | Ph);
            M[b] = Ph & Xv;
            S[b] += hout;
            int64_t idx = B.col_off[j] + (b - blo);
            B.pm[idx].P = P[b]; B.pm[idx].M = M[b]; B.score[idx] = S[b];
            hin = hout; prev_hout = hout; prev_done = 1;
        }
        if (bhi > bhi_prev) bhi_prev = bhi;
    }
    int64_t dist;
    if (m == 0) dist = n;
    else dist = cell_value(&B, n, m);
    int64_t ret = (dist <= k) ? dist : -1;

    if (ret >= 0 && ops) {
        int64_t i = n, j = m, c = 0;
        int32_t cur = (int32_t)dist;
        while (i > 0 || j > 0) {
            if (i > 0 && j > 0) {
                int32_t up = cell_value(&B, i - 1, j);
                if (up + 1 == cur) { ops[c++] = OP_I; --i; cur = up; continue; }
                int32_t left = cell_value(&B, i, j - 1);
                if (left + 1 == cur) { ops[c++] = OP_D; --j; cur = left; continue; }
                int eq = (q[i - 1] == t[j - 1]);
                ops[c++] = eq ? OP_EQ : OP_X; --i; --j; cur -= !eq;
            } else if (i > 0) { ops[c++] = OP_I; --i; --cur; }
            else { ops[c++] = OP_D; --j; --cur; }
        }
        reverse_ops(ops, c);
        if (n_ops) *n_ops = c;
    }
    free(P); free(M); free(S);
    free(B.col_off); free(B.col_blo); free(B.col_bhi); free(B.pm); free(B.score);
    return ret;
}

/* k < 0: unbounded (k-doubling from 64, like edlib).  k >= 0: returns -1 if distance > k. */
int64_t orc_align_myers(const uint8_t *q, int64_t n, const uint8_t *t, int64_t m, int64_t k,
                        uint8_t *ops, int64_t *n_ops) {
    if (n == 0 || m == 0) {
        int64_t d = n + m;
        if (k >= 0 && d > k) return -1;
        if (ops) {
            for (int64_t x = 0; x < n; ++x) ops[x] = OP_I;
            for (int64_t x = 0; x < m; ++x) ops[x] = OP_D;
            if (n_ops) *n_ops = d;
        }
        return d;
    }
    int64_t nb = (n + 63) >> 6;
    int symid[256]; int nsym = 0;
    for (int x = 0; x < 256; ++x) symid[x] = -1;
    for (int64_t x = 0; x < n; ++x) if (symid[q[x]] < 0) symid[q[x]] = nsym++;
    int absent = nsym;                       /* target-only symbols share one all-zero row */
    for (int x = 0; x < 256; ++x) if (symid[x] < 0) symid[x] = absent;
    uint64_t *peq = (uint64_t *)calloc((size_t)(nsym + 1) * (size_t)nb, sizeof(uint64_t));
    for (int64_t x = 0; x < n; ++x) peq[(size_t)symid[q[x]] * (size_t)nb + (size_t)(x >> 6)] |= 1ULL << (x & 63);

    int64_t kk = (k >= 0) ? k : 64;
    int64_t ret;
    for (;;) {
        ret = myers_attempt(q, n, t, m, kk, peq, symid, nb, ops, n_ops);
        if (ret >= 0 || k >= 0) break;
        kk *= 2;
    }
    free(peq);
    return ret;
}

/* dispatcher used by the shim and by the oracle pipeline */
int64_t orc_align(const uint8_t *q, int64_t n, const uint8_t *t, int64_t m,
                  uint8_t *ops, int64_t *n_ops) {
    return orc_align_myers(q, n, t, m, -1, ops, n_ops);
}

/* run-length extended CIGAR ("12=1X3I...") from forward ops; returns length written (no NUL count) */
int64_t orc_ops_to_cigar(const uint8_t *ops, int64_t n_ops, char *out, int64_t cap) {
    static const char sym[4] = { '=', 'X', 'I', 'D' };
    int64_t w = 0, i = 0;
    while (i < n_ops) {
        int64_t j = i;
        while (j < n_ops && ops[j] == ops[i]) ++j;
        int wrote = snprintf(out + w, (size_t)(cap - w), "%lld%c", (long long)(j - i), sym[ops[i]]);
        if (wrote < 0 || w + wrote >= cap) return -1;
        w += wrote; i = j;
    }
    if (w < cap) out[w] = 0;
    return w;
}

/* matches / alignment columns -- badread/misc.py:228-240 identity_from_edlib_cigar */
double orc_identity_from_ops(const uint8_t *ops, int64_t n_ops) {
    if (n_ops == 0) return 0.0;
    int64_t matches = 0;
    for (int64_t i = 0; i < n_ops; ++i) matches += (ops[i] == OP_EQ);
    return (double)matches / (double)n_ops;
}

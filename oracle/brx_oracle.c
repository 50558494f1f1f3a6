/*
 * oracle/brx_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * Single-threaded CPU restatement of Badread's per-read hot path, one function per reference
 * function, written to be obviously correct rather than fast.  It consumes the same flattened
 * tables (include/brx.h, with HOST pointers) and the same counter-based random streams
 * (include/brx_spec.h) as the HIP kernels, so the GPU must reproduce it byte for byte; its
 * deterministic pieces are pinned against the unmodified reference by tests/golden/ fixtures
 * generated with oracle/make_golden.py.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the library built from this file.
 *
 * Reference functions restated (paths relative to /root/reference):
 *   plan_read          badread/simulate.py:91-115 build_fragment, :148-165 get_fragment,
 *                      :168-180 get_fragment_type, :183-246 get_real_fragment, :249-253 junk,
 *                      :361-387 adapters, :459-482 add_glitches,
 *                      badread/fragment_lengths.py:47-52, badread/identities.py:76-93
 *   build_fragment     string slicing in simulate.py:206-246 + misc.py:56-71 reverse_complement
 *   choose_alt         badread/error_model.py:135-176 add_errors_to_kmer / add_one_random_change
 *   mutate             badread/simulate.py:256-346 (loop of sequence_fragment)
 *   assign_qscores     badread/qscore_model.py:32-75 get_qscores, :273-287 get_qscore
 *   format_record      badread/simulate.py:73-82
 * Alignment is oracle/myers_ref.c (edlib stand-in).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "brx.h"
#include "brx_spec.h"

int64_t orc_align(const uint8_t *q, int64_t n, const uint8_t *t, int64_t m, uint8_t *ops, int64_t *n_ops);

enum { SEG_REF = 0, SEG_ADAPTER = 1, SEG_RANDOM = 2, SEG_JUNK = 3 };
enum { PC_JUNK = 0, PC_RANDOM = 1, PC_REAL = 2, PC_HAIRPIN = 3 };

typedef struct { uint32_t type, a, b; uint64_t start, len; } oseg;
typedef struct { uint32_t type, contig, strand; uint64_t start, end, left_over; } opiece;

typedef struct {
    oseg *segs; int n_segs, cap_segs;
    opiece *pieces; int n_pieces, cap_pieces;
    uint64_t frag_len;
    double target_identity;
    uint32_t status;
} oplan;

typedef struct {
    brx_reference ref;
    brx_error_model em;
    brx_qscore_model qm;
    brx_sim_params p;
} orc_ctx;

orc_ctx *orc_create(void) { return (orc_ctx *)calloc(1, sizeof(orc_ctx)); }
void orc_destroy(orc_ctx *c) { free(c); }
void orc_set_reference(orc_ctx *c, const brx_reference *r) { c->ref = *r; }
void orc_set_error_model(orc_ctx *c, const brx_error_model *m) { c->em = *m; }
void orc_set_qscore_model(orc_ctx *c, const brx_qscore_model *m) { c->qm = *m; }
void orc_set_params(orc_ctx *c, const brx_sim_params *p) { c->p = *p; }

/* ------------------------------------------------------------------ decision trace
 * Every primitive random decision of plan_read can be logged, so that tools/make_golden.py can feed
 * the SAME decisions to the reference's build_fragment (simulate.py:91-115) through its random
 * sources and compare the resulting fragment strings with ours. */
enum { TR_U = 0,        /* raw uniform behind random.random(): chance tests and get_fragment_type   */
       TR_LENGTH = 1,   /* FragmentLengths.get_fragment_length()                                    */
       TR_CONTIG = 2,   /* random.choices(ref_contigs, weights)                                     */
       TR_START = 3,    /* random.randint(0, len-1)                                                 */
       TR_JUNKLEN = 4,  /* random.randint(1, 5)                                                     */
       TR_JUNKUNIT = 5, /* get_random_sequence(repeat_length): 2 bits per base                      */
       TR_SERIAL = 6,   /* get_random_sequence(n) for a random fragment / glitch insert: our serial */
       TR_ADAPTLEN = 7, /* int(len(adapter) * beta)                                                 */
       TR_GEO = 8,      /* np.random.geometric                                                      */
       TR_IDENTITY = 9  /* Identities.get_identity()                                                */ };
typedef struct { int32_t *kinds; double *vals; int cap, n; } orc_trace;
static __thread orc_trace *g_trace = NULL;
static void tr(int kind, double v) {
    if (g_trace && g_trace->n < g_trace->cap) { g_trace->kinds[g_trace->n] = kind; g_trace->vals[g_trace->n] = v; }
    if (g_trace) g_trace->n += 1;
}
static double draw_u(brx_rng *g) { double u = brx_next_double(g); tr(TR_U, u); return u; }

/* ------------------------------------------------------------------ plan */
static void push_seg(oplan *pl, uint32_t type, uint32_t a, uint32_t b, uint64_t start, uint64_t len) {
    if (len == 0) return;
    if (pl->n_segs == pl->cap_segs) {
        pl->cap_segs = pl->cap_segs ? 2 * pl->cap_segs : 16;
        pl->segs = (oseg *)realloc(pl->segs, sizeof(oseg) * (size_t)pl->cap_segs);
    }
    oseg s = { type, a, b, start, len };
    pl->segs[pl->n_segs++] = s;
}

static void push_piece(oplan *pl, opiece pc) {
    if (pl->n_pieces == pl->cap_pieces) {
        pl->cap_pieces = pl->cap_pieces ? 2 * pl->cap_pieces : 4;
        pl->pieces = (opiece *)realloc(pl->pieces, sizeof(opiece) * (size_t)pl->cap_pieces);
    }
    pl->pieces[pl->n_pieces++] = pc;
}

/* fragment_lengths.py:47-52 */
static uint64_t draw_fragment_length(const brx_sim_params *p, brx_rng *g) {
    int64_t L;
    if (p->frag_stdev == 0.0) L = brx_round_half_even(p->frag_mean);
    else {
        L = brx_round_half_even(brx_std_gamma(g, p->gamma_k) * p->gamma_t);
        if (L < 1) L = 1;
    }
    tr(TR_LENGTH, (double)L);
    return (uint64_t)L;
}

/* simulate.py:183-246; returns 0 on failure (the '' return at :213) */
static int real_fragment(const orc_ctx *c, brx_rng *g, uint64_t length, oplan *base) {
    const brx_reference *r = &c->ref;
    uint32_t contig = 0;
    if (r->n_contigs > 1) {                                    /* random.choices, :189 */
        double x = brx_next_double(g) * r->total_weight;
        while (contig < r->n_contigs - 1 && !(r->d_cum_weight[contig] > x)) ++contig;
        tr(TR_CONTIG, (double)contig);
    }
    uint32_t strand = (draw_u(g) < 0.5) ? 0u : 1u;             /* :194 */
    const brx_contig *ct = &r->d_contigs[contig];
    int circular = ct->flags & 1u;
    int hairpin = strand == 0 ? ((ct->flags >> 2) & 1u) : ((ct->flags >> 1) & 1u);   /* :202 */
    uint64_t len_c = ct->length;
    opiece pc = { PC_REAL, contig, strand, 0, 0, 0 };
    if (length >= len_c && !circular && !hairpin) {            /* :206-208 */
        pc.start = 0; pc.end = len_c;
        push_piece(base, pc);
        push_seg(base, SEG_REF, contig, strand, 0, len_c);
        return 1;
    }
    if (length > len_c && circular) return 0;                  /* :212-213 */
    uint64_t start = brx_next_below(g, len_c);                 /* :215 */
    tr(TR_START, (double)start);
    uint64_t end = start + length;
    if (circular) {                                            /* :219-226 */
        pc.start = start; pc.end = end;
        push_piece(base, pc);
        if (end <= len_c) push_seg(base, SEG_REF, contig, strand, start, length);
        else {
            push_seg(base, SEG_REF, contig, strand, start, len_c - start);
            push_seg(base, SEG_REF, contig, strand, 0, end - len_c);
        }
        return 1;
    }
    if (end > len_c) {
        if (hairpin) {                                         /* :235-240 */
            uint64_t fwd = len_c - start;
            uint64_t left_over = length - fwd < fwd ? length - fwd : fwd;
            pc.type = PC_HAIRPIN; pc.start = start; pc.end = len_c; pc.left_over = left_over;
            push_piece(base, pc);
            push_seg(base, SEG_REF, contig, strand, start, fwd);
            push_seg(base, SEG_REF, contig, strand ^ 1u, 0, left_over);
            return 1;
        }
        end = len_c;                                           /* :243 */
    }
    pc.start = start; pc.end = end;
    push_piece(base, pc);
    push_seg(base, SEG_REF, contig, strand, start, end - start);
    return 1;
}

/* simulate.py:148-165; returns 0 if 1000 tries failed */
static int get_fragment(const orc_ctx *c, brx_rng *g, oplan *base, uint32_t *next_serial) {
    const brx_sim_params *p = &c->p;
    uint64_t length = draw_fragment_length(p, g);
    double u = draw_u(g);                                      /* :174 */
    if (u < p->junk_rate) {                                    /* :249-253 */
        uint32_t unit_len = 1u + (uint32_t)brx_next_below(g, 5);
        uint32_t unit = 0;
        for (uint32_t i = 0; i < unit_len; ++i) unit |= (uint32_t)brx_next_below(g, 4) << (2 * i);
        tr(TR_JUNKLEN, (double)unit_len); tr(TR_JUNKUNIT, (double)unit);
        opiece pc = { PC_JUNK, 0, 0, 0, 0, 0 };
        push_piece(base, pc);
        push_seg(base, SEG_JUNK, unit, unit_len, 0, length);
        return 1;
    }
    if (u < p->junk_rate + p->random_rate) {
        opiece pc = { PC_RANDOM, 0, 0, 0, 0, 0 };
        push_piece(base, pc);
        tr(TR_SERIAL, (double)*next_serial);
        push_seg(base, SEG_RANDOM, (*next_serial)++, 0, 0, length);
        return 1;
    }
    for (int attempt = 0; attempt < 1000; ++attempt)
        if (real_fragment(c, g, length, base)) return 1;
    return 0;
}

/* copy [a,b) of the concatenation of base segments into out, splitting segments as needed */
static void copy_range(const oplan *base, uint64_t a, uint64_t b, oplan *out) {
    uint64_t pos = 0;
    for (int s = 0; s < base->n_segs && pos < b; ++s) {
        const oseg *sg = &base->segs[s];
        uint64_t lo = pos, hi = pos + sg->len;
        pos = hi;
        if (hi <= a) continue;
        uint64_t x0 = a > lo ? a : lo, x1 = b < hi ? b : hi;
        push_seg(out, sg->type, sg->a, sg->b, sg->start + (x0 - lo), x1 - x0);
    }
}

static void plan_free(oplan *pl) { free(pl->segs); free(pl->pieces); memset(pl, 0, sizeof(*pl)); }

static void plan_read(const orc_ctx *c, uint64_t seed, uint64_t read, oplan *out) {
    const brx_sim_params *p = &c->p;
    brx_rng g;
    brx_rng_init(&g, seed, read, BRX_ST_PLAN);
    uint32_t next_serial = 2;            /* serials 0 and 1 are the two pads of sequence_fragment */
    oplan base; memset(&base, 0, sizeof(base));
    memset(out, 0, sizeof(*out));

    /* start adapter, simulate.py:361-370 */
    if (p->start_adapter_len > 0 && p->start_rate != 0.0 && p->start_amount != 0.0) {
        if (draw_u(&g) < p->start_rate) {
            if (p->start_amount == 1.0) push_seg(&base, SEG_ADAPTER, 0, 0, 0, p->start_adapter_len);
            else {
                double f = brx_beta(&g, 2.0 * p->start_amount, 2.0 - 2.0 * p->start_amount);
                uint64_t L = (uint64_t)((double)p->start_adapter_len * f);       /* :387 int() */
                tr(TR_ADAPTLEN, (double)L);
                push_seg(&base, SEG_ADAPTER, 0, 0, p->start_adapter_len - L, L); /* suffix, :368 */
            }
        }
    }
    int ok = get_fragment(c, &g, &base, &next_serial);
    while (ok && draw_u(&g) < p->chimera_rate) {                                /* :101-110 */
        if (draw_u(&g) < 0.25) push_seg(&base, SEG_ADAPTER, 1, 0, 0, p->end_adapter_len);
        if (draw_u(&g) < 0.25) push_seg(&base, SEG_ADAPTER, 0, 0, 0, p->start_adapter_len);
        ok = get_fragment(c, &g, &base, &next_serial);
    }
    if (!ok) { out->status |= BRX_RS_NOFRAG; out->pieces = base.pieces; out->n_pieces = base.n_pieces;
               base.pieces = NULL; plan_free(&base); return; }
    /* end adapter, simulate.py:373-381 */
    if (p->end_adapter_len > 0 && p->end_rate != 0.0 && p->end_amount != 0.0) {
        if (draw_u(&g) < p->end_rate) {
            if (p->end_amount == 1.0) push_seg(&base, SEG_ADAPTER, 1, 0, 0, p->end_adapter_len);
            else {
                double f = brx_beta(&g, 2.0 * p->end_amount, 2.0 - 2.0 * p->end_amount);
                uint64_t L = (uint64_t)((double)p->end_adapter_len * f);
                tr(TR_ADAPTLEN, (double)L);
                push_seg(&base, SEG_ADAPTER, 1, 0, 0, L);                        /* prefix, :380 */
            }
        }
    }
    uint64_t base_len = 0;
    for (int s = 0; s < base.n_segs; ++s) base_len += base.segs[s].len;

    /* glitches, simulate.py:459-482 */
    if (p->glitch_rate == 0.0) {
        for (int s = 0; s < base.n_segs; ++s)
            push_seg(out, base.segs[s].type, base.segs[s].a, base.segs[s].b, base.segs[s].start, base.segs[s].len);
    } else {
        double p_rate = p->glitch_rate > 1.0 ? 1.0 / p->glitch_rate : 1.0;
        double p_size = p->glitch_size > 1.0 ? 1.0 / p->glitch_size : 1.0;
        double p_skip = p->glitch_skip > 1.0 ? 1.0 / p->glitch_skip : 1.0;
        uint64_t i = 0;
        for (;;) {
            uint64_t dist = (uint64_t)brx_geometric(&g, p_rate);
            tr(TR_GEO, (double)dist);
            uint64_t e = i + dist < base_len ? i + dist : base_len;
            copy_range(&base, i, e, out);
            i += dist;
            if (i >= base_len) break;
            if (p->glitch_size > 0.0) {
                uint64_t sz = (uint64_t)brx_geometric(&g, p_size);
                tr(TR_GEO, (double)sz); tr(TR_SERIAL, (double)next_serial);
                push_seg(out, SEG_RANDOM, next_serial++, 0, 0, sz);
            }
            if (p->glitch_skip > 0.0) { uint64_t sk = (uint64_t)brx_geometric(&g, p_skip); tr(TR_GEO, (double)sk); i += sk; }
            if (i >= base_len) break;
        }
    }
    out->pieces = base.pieces; out->n_pieces = base.n_pieces; base.pieces = NULL;
    plan_free(&base);
    out->frag_len = 0;
    for (int s = 0; s < out->n_segs; ++s) out->frag_len += out->segs[s].len;

    /* identities.py:76-93 */
    if (p->identity_mode == 0) out->target_identity = p->id_max;
    else if (p->identity_mode == 1) out->target_identity = p->id_max * brx_beta(&g, p->id_a, p->id_b);
    else {
        for (;;) {
            double q = p->id_a + p->id_b * brx_normal(&g);
            double id = 1.0 - brx_exp((-q / 10.0) * 2.302585092994046);
            if (id >= 0.0 && id <= 100.0) { out->target_identity = id; break; }
        }
    }
    tr(TR_IDENTITY, out->target_identity);
}

/* the decisions of plan_read for one read, in order; returns their number (may exceed cap) */
int orc_plan_trace(const orc_ctx *c, uint64_t seed, uint64_t read, int32_t *kinds, double *vals, int cap) {
    orc_trace t = { kinds, vals, cap, 0 };
    oplan pl;
    g_trace = &t;
    plan_read(c, seed, read, &pl);
    g_trace = NULL;
    plan_free(&pl);
    return t.n;
}

/* brx_draw4 and the samplers of include/brx_spec.h, for known-answer and distribution tests */
void orc_draw4(uint64_t seed, uint64_t read, uint32_t stream, uint64_t index, uint32_t out[4]) {
    brx_draw4(seed, read, stream, index, out);
}
void orc_sample(int kind, double a, double b, uint64_t seed, uint64_t n, double *out) {
    for (uint64_t i = 0; i < n; ++i) {
        brx_rng g;
        brx_rng_init(&g, seed, i, BRX_ST_PLAN);
        if (kind == 0) out[i] = brx_std_gamma(&g, a) * b;
        else if (kind == 1) out[i] = brx_beta(&g, a, b);
        else if (kind == 2) out[i] = a + b * brx_normal(&g);
        else if (kind == 3) out[i] = (double)brx_geometric(&g, a);
        else if (kind == 4) out[i] = brx_next_double(&g);
        else if (kind == 5) out[i] = (double)brx_next_below(&g, (uint64_t)a);
        else if (kind == 6) out[i] = brx_log(a + (double)i * b);
        else if (kind == 7) out[i] = brx_exp(a + (double)i * b);
    }
}

/* ------------------------------------------------------------------ fragment bytes */
static uint8_t ref_code(const brx_reference *r, uint32_t contig, uint32_t strand, uint64_t pos) {
    const brx_contig *ct = &r->d_contigs[contig];
    uint64_t f = strand == 0 ? pos : (uint64_t)ct->length - 1 - pos;
    uint64_t gidx = ct->base_off + f;
    uint8_t code = (uint8_t)((r->d_packed[gidx >> 4] >> (2 * (gidx & 15))) & 3u);
    for (uint32_t e = 0; e < r->n_exceptions; ++e)         /* linear scan: obviously correct */
        if (gidx >= r->d_exceptions[e].start && gidx < r->d_exceptions[e].end) { code = (uint8_t)r->d_exceptions[e].code; break; }
    return strand == 0 ? code : r->comp[code];
}

static void fill_segments(const orc_ctx *c, uint64_t seed, uint64_t read, const oplan *pl, uint8_t *dst) {
    uint64_t w = 0;
    for (int s = 0; s < pl->n_segs; ++s) {
        const oseg *sg = &pl->segs[s];
        for (uint64_t x = 0; x < sg->len; ++x) {
            uint64_t pos = sg->start + x;
            uint8_t code;
            switch (sg->type) {
            case SEG_REF: code = ref_code(&c->ref, sg->a, sg->b, pos); break;
            case SEG_ADAPTER: code = (sg->a == 0 ? c->p.d_start_adapter : c->p.d_end_adapter)[pos]; break;
            case SEG_RANDOM: code = (uint8_t)brx_random_base(seed, read, sg->a, pos); break;
            default: code = (uint8_t)((sg->a >> (2 * (pos % sg->b))) & 3u); break;
            }
            dst[w++] = code;
        }
    }
}

/* ------------------------------------------------------------------ error model lookup */
/* result: per-position replacement words for positions i..i+k-1, 0 where unchanged.
 * word = 0x80000000 | len << 24 | pool offset of the characters.  Returns 0 if the k-mer is
 * unchanged (the `continue` at simulate.py:300-301). */
static int random_change(const uint8_t *kmer, int k, uint32_t w3, uint32_t *rep) {   /* error_model.py:163-176 */
    uint32_t type = w3 % 3u;
    uint32_t pos = (w3 / 3u) % (uint32_t)k;
    uint32_t rest = w3 / (3u * (uint32_t)k);
    uint32_t o = kmer[pos];
    for (int j = 0; j < k; ++j) rep[j] = 0;
    if (type == 0) {                                   /* substitution: a base different from o */
        uint32_t nb = o < 4 ? ((o + 1u + rest % 3u) & 3u) : (rest & 3u);
        rep[pos] = 0x80000000u | (1u << 24) | nb;
    } else if (type == 1) {                            /* insertion after (1) or before (0) */
        uint32_t after = rest & 1u, nb = (rest >> 1) & 3u;
        uint32_t x = after ? o : nb, y = after ? nb : o;
        rep[pos] = 0x80000000u | (2u << 24) | (16u + 2u * (16u * x + y));
    } else rep[pos] = 0x80000000u;                     /* deletion: length 0 */
    return 1;
}

static int choose_alt(const brx_error_model *em, const uint8_t *kmer, uint32_t w2, uint32_t w3, uint32_t *rep) {
    int k = em->k;
    if (em->type == 0) return random_change(kmer, k, w3, rep);      /* error_model.py:140-141 */
    uint32_t row = 0;
    for (int j = 0; j < k; ++j) {
        if (kmer[j] > 3) return random_change(kmer, k, w3, rep);    /* not in table, :143-144 */
        row = (row << 2) | kmer[j];
    }
    uint32_t a0 = em->d_row_off[row], a1 = em->d_row_off[row + 1];
    if (a0 == a1) return random_change(kmer, k, w3, rep);
    uint32_t a = a0;
    while (a < a1 && !(w2 < em->d_thr[a])) ++a;
    if (a == a1) {
        if (em->d_thr[a1 - 1] == 0xFFFFFFFFu) a = a1 - 1;
        else return random_change(kmer, k, w3, rep);                /* remainder, :151-158 */
    }
    uint32_t o = em->d_desc[a];
    uint32_t diff = (uint32_t)em->d_pool[o] | ((uint32_t)em->d_pool[o + 1] << 8);
    if (diff == 0) return 0;
    uint32_t coff = o + 2u + (uint32_t)k;
    for (int j = 0; j < k; ++j) {
        uint32_t len = em->d_pool[o + 2 + (uint32_t)j];
        rep[j] = ((diff >> j) & 1u) ? (0x80000000u | (len << 24) | coff) : 0u;
        coff += len;
    }
    return 1;
}

static inline uint32_t rep_len(uint32_t w) { return w ? ((w >> 24) & 0x7Fu) : 1u; }

/* join(new_fragment_bases[a:b]) */
static uint64_t join_range(const brx_error_model *em, const uint8_t *F, const uint32_t *repl,
                           uint64_t a, uint64_t b, uint8_t *out) {
    uint64_t w = 0;
    for (uint64_t p = a; p < b; ++p) {
        if (!repl[p]) { if (out) out[w] = F[p]; ++w; continue; }
        uint32_t len = (repl[p] >> 24) & 0x7Fu, off = repl[p] & 0x00FFFFFFu;
        if (out) for (uint32_t x = 0; x < len; ++x) {
            uint8_t ch;
            if (off < 16) ch = (uint8_t)off;
            else if (off < BRX_POOL_PREAMBLE) { uint32_t v = (off - 16) / 2; ch = (uint8_t)(x == 0 ? v / 16 : v % 16); }
            else ch = em->d_pool[off + x];
            out[w + x] = ch;
        }
        w += len;
    }
    return w;
}

static void count_ops(const uint8_t *ops, int64_t n, uint32_t *match, uint32_t *edits) {
    uint32_t m = 0;
    for (int64_t i = 0; i < n; ++i) m += (ops[i] == 0);
    *match = m; *edits = (uint32_t)n - m;
}

/* ------------------------------------------------------------------ sequence_fragment */
typedef struct {
    uint8_t *seq; uint8_t *qual; uint64_t seq_len;     /* trimmed */
} oread;

/* F: padded fragment codes, n = len(F).  simulate.py:256-358 */
static void sequence_padded(const orc_ctx *c, uint64_t seed, uint64_t read, const uint8_t *F, uint64_t n,
                            double target, brx_read_stats *st, oread *out) {
    const brx_error_model *em = &c->em;
    const brx_qscore_model *qm = &c->qm;
    int k = em->k;
    uint32_t *repl = (uint32_t *)calloc((size_t)n, sizeof(uint32_t));
    uint8_t *win = (uint8_t *)malloc(128 * 1000 + 16);
    uint8_t *wops = (uint8_t *)malloc(129 * 1000 + 16);
    double errors = 0.0;
    uint64_t change = 0, loops = 0, nalign = 0;
    uint64_t max_i = n - 1 - (uint64_t)k;                              /* :269 */
    double need = (double)n * (1.0 - target);                          /* :270 */
    uint32_t rep[16];
    for (;;) {
        if (need < 0.5) break;                                         /* :274 */
        loops += 1;
        if (loops > 100 * n) break;                                    /* :279 */
        if ((double)change > 0.9 * (double)n) break;                   /* :285 */
        double est = 1.0 - errors / (double)n;                         /* :290 */
        if (est <= target) break;
        uint32_t w[4];
        brx_draw4(seed, read, BRX_ST_MUT, loops - 1, w);
        uint64_t i = brx_mulhi64(((uint64_t)w[1] << 32) | w[0], max_i + 1);   /* :294 */
        if (!choose_alt(em, F + i, w[2], w[3], rep)) continue;          /* :296-301 */
        double scale = est * brx_sqrt(est);                            /* est ** 1.5, :321 */
        for (int j = 0; j < k; ++j) {
            if (!rep[j] || repl[i + (uint64_t)j]) continue;             /* :309 */
            repl[i + (uint64_t)j] = rep[j];
            change += 1;
            uint32_t len = (rep[j] >> 24) & 0x7Fu;
            double new_errors = (double)(len < 2 ? 1u : len - 1u);     /* :312-315 */
            errors += new_errors * scale;
            if (change % 25 == 0) {                                    /* :325 */
                uint64_t a = 0, b = n;
                if (n > 1000) {
                    uint32_t ww[4];
                    brx_draw4(seed, read, BRX_ST_WIN, nalign, ww);
                    a = brx_mulhi64(((uint64_t)ww[1] << 32) | ww[0], n - 1000 + 1);   /* :338 */
                    b = a + 1000;
                }
                nalign += 1;
                uint64_t tl = join_range(em, F, repl, a, b, NULL);
                uint8_t *tbuf = win, *obuf = wops;
                if (tl > 128 * 1000) { tbuf = (uint8_t *)malloc(tl + 16); obuf = (uint8_t *)malloc(tl + (b - a) + 16); }
                join_range(em, F, repl, a, b, tbuf);
                int64_t nops = 0;
                orc_align(F + a, (int64_t)(b - a), tbuf, (int64_t)tl, obuf, &nops);   /* :330,340 */
                uint32_t match, edits;
                count_ops(obuf, nops, &match, &edits);
                double id = nops ? (double)match / (double)nops : 0.0;
                if (n <= 1000) errors = (1.0 - id) * (double)n;        /* :333 */
                else {
                    double est_err = (1.0 - id) * (double)n;           /* :344 */
                    double weight = 1000.0 / (double)n;
                    errors = est_err * weight + errors * (1.0 - weight);
                }
                if (tbuf != win) { free(tbuf); free(obuf); }
            }
        }
    }
    st->loop_count = (uint32_t)loops; st->change_count = (uint32_t)change; st->n_alignments = (uint32_t)nalign;

    uint64_t start_trim = join_range(em, F, repl, 0, (uint64_t)k, NULL);          /* :348 */
    uint64_t end_trim = join_range(em, F, repl, n - (uint64_t)k, n, NULL);        /* :349 */
    uint64_t m = join_range(em, F, repl, 0, n, NULL);
    uint8_t *seq = (uint8_t *)malloc(m + 16);
    join_range(em, F, repl, 0, n, seq);

    /* get_qscores, qscore_model.py:32-75 */
    uint8_t *ops = (uint8_t *)malloc(m + n + 16);
    int64_t nops = 0;
    int64_t dist = orc_align(seq, (int64_t)m, F, (int64_t)n, ops, &nops);          /* :37 */
    uint32_t match, edits;
    count_ops(ops, nops, &match, &edits);
    st->n_cols = (uint32_t)nops; st->n_match = match; st->padded_len = (uint32_t)m; (void)dist;
    uint64_t *col_of = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(m + 1));
    { uint64_t s = 0; for (int64_t cidx = 0; cidx < nops; ++cidx) if (ops[cidx] != 3) col_of[s++] = (uint64_t)cidx; }
    uint8_t *qual = (uint8_t *)malloc(m + 16);
    int margin = (qm->k - 1) / 2;                                                  /* :42 */
    uint64_t qhist[256];
    memset(qhist, 0, sizeof(qhist));
    for (uint64_t s = 0; s < m; ++s) {
        uint64_t h = (uint64_t)margin;
        if (s < h) h = s;
        if (m - 1 - s < h) h = m - 1 - s;                                          /* :57-59 */
        /* window ops and interior D-run lengths, centre-out trimming on a miss (:278-286) */
        uint32_t score = 0; int found = 0;
        for (;;) {
            uint64_t c0 = col_of[s - h], c1 = col_of[s + h];
            uint64_t key = (uint64_t)(2 * h + 1) << 56;
            int shift = 0, idx = 0;
            uint32_t run = 0;
            for (uint64_t cc = c0; cc <= c1; ++cc) {
                if (ops[cc] == 3) { run += 1; continue; }
                if (idx > 0) {
                    uint32_t maxrun = (1u << qm->gap_bits) - 1u;
                    uint32_t code = run >= maxrun ? maxrun : run;
                    key |= (uint64_t)code << shift; shift += qm->gap_bits;
                }
                key |= (uint64_t)ops[cc] << shift; shift += 2;
                run = 0; idx += 1;
            }
            /* hash lookup */
            uint64_t hsh = key * 0x9E3779B97F4A7C15ull;
            uint32_t slot = (uint32_t)(hsh >> 32) & (qm->hash_size - 1);
            int64_t row = -1;
            for (;;) {
                uint64_t kk = qm->d_hash_key[slot];
                if (kk == key) { row = qm->d_hash_row[slot]; break; }
                if (kk == ~0ull) break;
                slot = (slot + 1) & (qm->hash_size - 1);
            }
            if (row >= 0) {
                uint32_t e0 = qm->d_row_off[row], e1 = qm->d_row_off[row + 1];
                uint32_t w4[4];
                brx_draw4(seed, read, BRX_ST_QS, s >> 2, w4);
                uint32_t u = w4[s & 3];
                uint32_t e = e0;
                while (e < e1 - 1 && !(u < qm->d_thr[e])) ++e;
                score = qm->d_score[e]; found = 1;
                break;
            }
            if (h == 0) break;
            h -= 1;
        }
        if (!found) { st->status |= BRX_RS_QMISS; score = 0; }
        qual[s] = (uint8_t)(score + 33);
        qhist[score & 255u] += 1;
    }
    /* sum of 10^(-q/10) over read bases (qscore_model.py:71-73), accumulated per score value in
       ascending score order so that a parallel implementation can reproduce the rounding */
    double qerr = 0.0;
    for (int q = 0; q < 256; ++q)
        if (qhist[q]) qerr += (double)qhist[q] * brx_exp(-(double)q / 10.0 * 2.302585092994046);
    st->qerr_sum = qerr;
    /* trim, simulate.py:355-356 */
    uint64_t lo = start_trim, hi = m >= end_trim ? m - end_trim : 0;
    if (end_trim == 0) hi = 0;                   /* seq[a:-0] is empty in Python */
    if (hi < lo) hi = lo;
    out->seq_len = hi - lo;
    out->seq = (uint8_t *)malloc(out->seq_len + 1);
    out->qual = (uint8_t *)malloc(out->seq_len + 1);
    memcpy(out->seq, seq + lo, out->seq_len);
    memcpy(out->qual, qual + lo, out->seq_len);
    st->seq_len = (uint32_t)out->seq_len;
    free(repl); free(win); free(wops); free(seq); free(ops); free(col_of); free(qual);
}

/* ------------------------------------------------------------------ record formatting */
static size_t format_header(const orc_ctx *c, uint64_t seed, uint64_t read, const oplan *pl,
                            const brx_read_stats *st, char *buf, size_t cap) {
    uint32_t w[4];
    brx_draw4(seed, read, BRX_ST_NAME, 0, w);                         /* uuid.UUID(int=getrandbits(128)), :77 */
    size_t o = (size_t)snprintf(buf, cap, "@%08x-%04x-%04x-%04x-%04x%08x ", w[0], w[1] >> 16, w[1] & 0xffffu,
                                w[2] >> 16, w[2] & 0xffffu, w[3]);
    for (int i = 0; i < pl->n_pieces; ++i) {
        const opiece *pc = &pl->pieces[i];
        if (i > 0) o += (size_t)snprintf(buf + o, cap - o, "chimera ");
        if (pc->type == PC_JUNK) o += (size_t)snprintf(buf + o, cap - o, "junk_seq ");
        else if (pc->type == PC_RANDOM) o += (size_t)snprintf(buf + o, cap - o, "random_seq ");
        else {
            const brx_contig *ct = &c->ref.d_contigs[pc->contig];
            memcpy(buf + o, c->ref.d_names + ct->name_off, ct->name_len); o += ct->name_len;
            o += (size_t)snprintf(buf + o, cap - o, ",%cstrand,", pc->strand ? '-' : '+');
            if (pc->type == PC_HAIRPIN)
                o += (size_t)snprintf(buf + o, cap - o, "%llu-%llu (hairpin) 0-%llu ", (unsigned long long)pc->start,
                                      (unsigned long long)pc->end, (unsigned long long)pc->left_over);
            else
                o += (size_t)snprintf(buf + o, cap - o, "%llu-%llu ", (unsigned long long)pc->start, (unsigned long long)pc->end);
        }
    }
    double ident = st->n_cols ? (double)st->n_match / (double)st->n_cols : 0.0;
    o += (size_t)snprintf(buf + o, cap - o, "length=%u error-free_length=%u read_identity=%.3f%%\n",
                          st->seq_len, st->frag_len, ident * 100.0);
    return o;
}

static size_t header_bound(const orc_ctx *c, const oplan *pl) {
    size_t b = 160;
    for (int i = 0; i < pl->n_pieces; ++i) {
        b += 96;
        if (pl->pieces[i].type >= PC_REAL) b += c->ref.d_contigs[pl->pieces[i].contig].name_len;
    }
    return b;
}

/* ------------------------------------------------------------------ public entry points */
/* returns bytes written, or -(bytes needed) if cap is too small */
int64_t orc_simulate_batch(const orc_ctx *c, uint64_t seed, uint64_t first_read, uint32_t n_reads,
                           uint8_t *out, int64_t cap, brx_read_stats *stats) {
    int64_t w = 0;
    int k = c->em.k;
    for (uint32_t r = 0; r < n_reads; ++r) {
        uint64_t read = first_read + r;
        brx_read_stats *st = &stats[r];
        memset(st, 0, sizeof(*st));
        oplan pl;
        plan_read(c, seed, read, &pl);
        st->status = pl.status;
        st->target_identity = pl.target_identity;
        st->rec_off = (uint64_t)w;
        if (pl.status & BRX_RS_NOFRAG) { plan_free(&pl); continue; }
        uint64_t L = pl.frag_len, n = L + 2 * (uint64_t)k;
        st->frag_len = (uint32_t)L;
        uint8_t *F = (uint8_t *)malloc(n + 16);
        for (int x = 0; x < k; ++x) {                                             /* simulate.py:260 */
            F[x] = (uint8_t)brx_random_base(seed, read, 0, (uint64_t)x);
            F[n - (uint64_t)k + (uint64_t)x] = (uint8_t)brx_random_base(seed, read, 1, (uint64_t)x);
        }
        fill_segments(c, seed, read, &pl, F + k);
        oread rd;
        sequence_padded(c, seed, read, F, n, pl.target_identity, st, &rd);
        if (rd.seq_len == 0) st->status |= BRX_RS_EMPTY;                           /* simulate.py:70-71 */
        else {
            size_t hb = header_bound(c, &pl);
            char *hdr = (char *)malloc(hb);
            size_t hl = format_header(c, seed, read, &pl, st, hdr, hb);
            int64_t need = (int64_t)hl + 2 * (int64_t)rd.seq_len + 4;
            st->rec_len = (uint32_t)need;
            if (out && w + need <= cap) {
                memcpy(out + w, hdr, hl);
                uint8_t *p = out + w + hl;
                for (uint64_t x = 0; x < rd.seq_len; ++x) p[x] = c->ref.sym[rd.seq[x]];
                p += rd.seq_len; *p++ = '\n'; *p++ = '+'; *p++ = '\n';
                memcpy(p, rd.qual, rd.seq_len); p += rd.seq_len; *p++ = '\n';
            }
            w += need;
            free(hdr);
        }
        free(rd.seq); free(rd.qual); free(F);
        plan_free(&pl);
    }
    return (out == NULL || w <= cap) ? w : -w;
}

/* sequence_fragment on caller-supplied fragments: output per fragment = seq codes then quals */
int64_t orc_sequence_fragments(const orc_ctx *c, uint64_t seed, uint64_t first_read, uint32_t n_frags,
                               const uint8_t *frags, const uint64_t *frag_off, const double *target,
                               uint8_t *out, int64_t cap, brx_read_stats *stats) {
    int64_t w = 0;
    int k = c->em.k;
    for (uint32_t r = 0; r < n_frags; ++r) {
        uint64_t read = first_read + r;
        brx_read_stats *st = &stats[r];
        memset(st, 0, sizeof(*st));
        uint64_t L = frag_off[r + 1] - frag_off[r], n = L + 2 * (uint64_t)k;
        st->frag_len = (uint32_t)L; st->target_identity = target[r]; st->rec_off = (uint64_t)w;
        uint8_t *F = (uint8_t *)malloc(n + 16);
        for (int x = 0; x < k; ++x) {
            F[x] = (uint8_t)brx_random_base(seed, read, 0, (uint64_t)x);
            F[n - (uint64_t)k + (uint64_t)x] = (uint8_t)brx_random_base(seed, read, 1, (uint64_t)x);
        }
        memcpy(F + k, frags + frag_off[r], L);
        oread rd;
        sequence_padded(c, seed, read, F, n, target[r], st, &rd);
        int64_t need = 2 * (int64_t)rd.seq_len;
        st->rec_len = (uint32_t)need;
        if (rd.seq_len == 0) st->status |= BRX_RS_EMPTY;
        if (out && w + need <= cap) {
            memcpy(out + w, rd.seq, rd.seq_len);
            memcpy(out + w + rd.seq_len, rd.qual, rd.seq_len);
        }
        w += need;
        free(rd.seq); free(rd.qual); free(F);
    }
    return (out == NULL || w <= cap) ? w : -w;
}

/* ---- fine-grained probes for golden tests against the reference's deterministic functions ---- */

/* plan only: fills flat arrays; segs: 5 x u64 each (type, a, b, start, len); pieces: 6 x u64 */
int orc_plan_probe(const orc_ctx *c, uint64_t seed, uint64_t read, uint64_t *segs, int seg_cap, int *n_segs,
                   uint64_t *pieces, int piece_cap, int *n_pieces, uint64_t *frag_len, double *target, uint32_t *status) {
    oplan pl;
    plan_read(c, seed, read, &pl);
    *n_segs = pl.n_segs; *n_pieces = pl.n_pieces; *frag_len = pl.frag_len; *target = pl.target_identity; *status = pl.status;
    for (int s = 0; s < pl.n_segs && s < seg_cap; ++s) {
        segs[5 * s] = pl.segs[s].type; segs[5 * s + 1] = pl.segs[s].a; segs[5 * s + 2] = pl.segs[s].b;
        segs[5 * s + 3] = pl.segs[s].start; segs[5 * s + 4] = pl.segs[s].len;
    }
    for (int s = 0; s < pl.n_pieces && s < piece_cap; ++s) {
        pieces[6 * s] = pl.pieces[s].type; pieces[6 * s + 1] = pl.pieces[s].contig; pieces[6 * s + 2] = pl.pieces[s].strand;
        pieces[6 * s + 3] = pl.pieces[s].start; pieces[6 * s + 4] = pl.pieces[s].end; pieces[6 * s + 5] = pl.pieces[s].left_over;
    }
    plan_free(&pl);
    return 0;
}

/* the unpadded fragment of one read as base codes (what build_fragment returns, simulate.py:115) */
int64_t orc_fragment_probe(const orc_ctx *c, uint64_t seed, uint64_t read, uint8_t *out, int64_t cap) {
    oplan pl;
    plan_read(c, seed, read, &pl);
    int64_t L = (int64_t)pl.frag_len;
    if (!(pl.status & BRX_RS_NOFRAG) && L <= cap) fill_segments(c, seed, read, &pl, out);
    plan_free(&pl);
    return L;
}

/* reference slice as codes: what seq[start:start+len] of the chosen strand string holds */
void orc_ref_slice(const orc_ctx *c, uint32_t contig, uint32_t strand, uint64_t start, uint64_t len, uint8_t *out) {
    for (uint64_t x = 0; x < len; ++x) out[x] = ref_code(&c->ref, contig, strand, start + x);
}

/* error_model.add_errors_to_kmer with explicit draws: writes per-position strings as codes,
 * lens[j] (255 = unchanged marker is not used; unchanged positions return the original base) */
int orc_choose_alt_probe(const orc_ctx *c, const uint8_t *kmer, uint32_t w2, uint32_t w3, uint8_t *lens, uint8_t *chars) {
    uint32_t rep[16];
    int k = c->em.k;
    int changed = choose_alt(&c->em, kmer, w2, w3, rep);
    uint32_t w = 0;
    for (int j = 0; j < k; ++j) {
        if (!changed || !rep[j]) { lens[j] = 1; chars[w++] = kmer[j]; continue; }
        uint32_t one[1] = { rep[j] };
        uint8_t dummyF[1] = { kmer[j] };
        lens[j] = (uint8_t)join_range(&c->em, dummyF, one, 0, 1, chars + w);
        w += lens[j];
    }
    return changed;
}

/* get_qscores given explicit ops (forward, 0 '=',1 'X',2 'I',3 'D'): returns per-base window keys
 * resolved after fallback as the table row index (or -1), for golden comparison with get_qscore */
int orc_qscore_rows_probe(const orc_ctx *c, const uint8_t *ops, int64_t nops, int64_t *rows, int32_t *used_h) {
    const brx_qscore_model *qm = &c->qm;
    int64_t m = 0;
    for (int64_t i = 0; i < nops; ++i) m += (ops[i] != 3);
    uint64_t *col_of = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(m + 1));
    { uint64_t s = 0; for (int64_t cidx = 0; cidx < nops; ++cidx) if (ops[cidx] != 3) col_of[s++] = (uint64_t)cidx; }
    int margin = (qm->k - 1) / 2;
    for (int64_t s = 0; s < m; ++s) {
        int64_t h = margin;
        if (s < h) h = s;
        if (m - 1 - s < h) h = m - 1 - s;
        rows[s] = -1; used_h[s] = -1;
        for (;;) {
            uint64_t c0 = col_of[s - h], c1 = col_of[s + h];
            uint64_t key = (uint64_t)(2 * h + 1) << 56;
            int shift = 0, idx = 0; uint32_t run = 0;
            for (uint64_t cc = c0; cc <= c1; ++cc) {
                if (ops[cc] == 3) { run += 1; continue; }
                if (idx > 0) {
                    uint32_t maxrun = (1u << qm->gap_bits) - 1u;
                    uint32_t code = run >= maxrun ? maxrun : run;
                    key |= (uint64_t)code << shift; shift += qm->gap_bits;
                }
                key |= (uint64_t)ops[cc] << shift; shift += 2; run = 0; idx += 1;
            }
            uint64_t hsh = key * 0x9E3779B97F4A7C15ull;
            uint32_t slot = (uint32_t)(hsh >> 32) & (qm->hash_size - 1);
            for (;;) {
                uint64_t kk = qm->d_hash_key[slot];
                if (kk == key) { rows[s] = qm->d_hash_row[slot]; break; }
                if (kk == ~0ull) break;
                slot = (slot + 1) & (qm->hash_size - 1);
            }
            if (rows[s] >= 0) { used_h[s] = (int32_t)h; break; }
            if (h == 0) break;
            h -= 1;
        }
    }
    free(col_of);
    return 0;
}

/*
 * brx_align.h -- wavefront-cooperative banded Myers/Hyyro bit-vector global alignment for gfx950.
 *
 * Replaces every edlib.align(..., task='path') on Badread's simulate path
 * (/root/reference/badread/simulate.py:330,340; qscore_model.py:37; error_model.py:202).
 *
 * One alignment per 64-lane wavefront.  The query is cut into 32-row words; lane l owns
 * "superblock" s = l (mod 64) = G consecutive words (R = 32*G rows).  Lanes run the column
 * recurrence systolically: superblock s works on column j at time step t = j + s, so the
 * horizontal delta leaving superblock s-1 for column j (computed at t-1) reaches superblock s
 * exactly when it needs it -- one cross-lane exchange per step, no carry look-ahead, no wasted
 * work.  Only superblocks inside the Ukkonen band |i-j| <= f(k) are computed; a lane that leaves
 * the band at the top re-enters 64 superblocks further down.  Each lane keeps the five equality
 * masks (A,C,G,T,N) of its words in registers for as long as it owns them, so the only per-step
 * memory traffic is one target byte in (4-byte prefetched) and 8*G bytes of traceback bits out,
 * written as [time step][band slot] rows so that a wave's stores are contiguous.
 *
 * Traceback (canonical path: up/'I' first, then left/'D', then diagonal, from the bottom-right
 * cell; identical to oracle/myers_ref.c) uses the stored vertical-plus (Pv) and horizontal-plus
 * (Ph) bit-vectors only: a cell moves up iff its Pv bit is set, else left iff its Ph bit is set,
 * else diagonally.  All 64 lanes speculate down the current diagonal at once (lane l looks at
 * cell (i-l, j-l)); a ballot finds the first lane that is not a diagonal move, so one round
 * retires a whole run of '='/'X' columns plus the indel that ends it.
 *
 * Exactness: cells outside the band are bounded above (+1 per row/column), cells whose true value
 * is <= k are exact, and the canonical path only visits such cells when k >= distance; a
 * traceback that would leave the band or ends with cost > k reports failure (the caller doubles
 * k, or flags BRX_RS_BAND when k was a proven bound).
 */
#ifndef BRX_ALIGN_H
#define BRX_ALIGN_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

/* pointer to global memory, for the stores whose address the compiler would otherwise treat as generic (flat) */
#ifndef BRX_GLOBAL
#define BRX_GLOBAL __attribute__((address_space(1)))
#endif

#define BRX_OP_EQ 0
#define BRX_OP_X 1
#define BRX_OP_I 2
#define BRX_OP_D 3

/* debug progress words (pinned host memory, system-scope stores); prog == nullptr in normal runs */
#define BRX_PROG(prog, slot, val)                                                                   \
    do { if ((prog) && (threadIdx.x & 63) == 0 && blockIdx.x < 8)                                    \
             __hip_atomic_store((prog) + 8 * blockIdx.x + (slot), (uint32_t)(val), __ATOMIC_RELAXED,  \
                                __HIP_MEMORY_SCOPE_SYSTEM); } while (0)

#define BRX_U1 8       /* columns per loop trip of the one-word-per-lane forward pass */
struct BrxGeom {
    int Q, T;          /* query rows, target columns                              */
    int dlo, dhi;      /* band of diagonals i-j                                   */
    int G, R;          /* words per lane, rows per superblock (32*G)              */
    int NS, NW;        /* superblocks, 32-row words                               */
    int WSp;           /* band slots per time step in the traceback store         */
    int WSrow, slot0;  /* slots of a whole store row and the first slot of THIS alignment in it: WSp and 0 when the store holds one
                          alignment; brx_quad.h lays the rows of four alignments side by side (4 x WSp, row x WSp)          */
    int K;             /* time skew between neighbouring superblocks: superblock s handles column j at time j + K*s (1) */
    int U;             /* columns per loop trip of the forward pass (BRX_U1 for G = 1, else 1): the windowed store is decided per trip */
    int t_end;         /* last traceback row = T + NS - 1 (G = 1: rounded up to whole trips) */
    int H;             /* windowed traceback store: only superblocks within H rows of the straight line
                          row = column * Q / T are written (BRX_H_ALL: every superblock of the band)    */
    uint32_t slope;    /* Q / T with 20 fractional bits (0 when the store is not windowed)               */
};
#define BRX_H_ALL (1 << 29)

__host__ __device__ inline uint32_t brx_isqrt(uint32_t v) {
    uint32_t r = 0;
    for (uint32_t bit = 1u << 15; bit; bit >>= 1) { const uint32_t c = r | bit; if ((uint64_t)c * c <= v) r = c; }
    return r;
}

/* Words per lane: 1..16 live in registers (brx_align_forward<G>); wider bands -- a read that is both very long and very
 * inaccurate: more than 57 344 band rows -- keep the lane's words in the per-wave state array of brx_align_forward_wide,
 * which takes any power of two.  4096 words per lane = a band of 7.3 M rows: beyond any fragment the planner can draw
 * from a 2^32-base contig at an identity the CLI accepts; past it brx_make_geom gives up (BRX_RS_BAND). */
#define BRX_GEOM_MAXG 4096
/* k must be >= |Q-T|.  Returns G = 0 if the band is wider than 64 lanes x BRX_GEOM_MAXG words can hold.
 *
 * hmul != 0 (windowed store): the forward pass still computes the whole Ukkonen band (the cell values, and therefore every
 * Pv/Ph bit, are unchanged), but only the superblocks that intersect rows [c(j) - H, c(j) + H] of column j,
 * c(j) = j*Q/T, are written to the traceback store.  The canonical path of an alignment with k edits strays
 * from that straight line like a random walk of ~k steps (measured: at most 1.5 sqrt(distance) rows on
 * nanopore2023 reads of 2-60 kb), so with H = hmul sqrt(k) + 24, hmul = 2 (BRX_TB_WINDOW; measured on configs[3]: 4 -> 2.84, 3 -> 2.92, 2 -> 2.98 Gbases/s with no miss, 1 -> 2.9 with 110 misses per 295 k reads), the traceback practically never asks
 * for a cell that was not stored (hmul < 0: H = 8, a test setting that makes most reads miss); when it does, the alignment reports failure and the caller repeats it with the full
 * store.  A traceback that succeeds read exactly the bits the full store would have held: same result. */
/* span / maxg: the band may cover `span` superblocks (56 of the 64 lanes of a wave; BRX_QUAD_SPAN of the 16 lanes of a row,
   brx_quad.h) of at most maxg words each */
__host__ __device__ inline BrxGeom brx_make_geom_span(int Q, int T, int k, int hmul, int span, int maxg) {
    BrxGeom g;
    g.Q = Q; g.T = T;
    int dend = Q - T;
    int adend = dend < 0 ? -dend : dend;
    if (k < adend) k = adend;
    int half = (k - adend) / 2;
    g.dlo = (dend < 0 ? dend : 0) - half;
    g.dhi = (dend > 0 ? dend : 0) + half;
    int bw = g.dhi - g.dlo + 1;
    int G = 1;
    while (G <= maxg && (long long)bw > (long long)span * 32ll * G) G *= 2;
    if (G > maxg) { g.G = 0; g.R = 0; g.NS = 0; g.NW = 0; g.WSp = 0; g.WSrow = 0; g.slot0 = 0; g.K = 1; g.U = 1; g.t_end = 0; g.H = BRX_H_ALL; g.slope = 0; return g; }
    g.G = G; g.R = 32 * G;
    g.NW = (Q + 31) / 32;
    g.NS = (Q + g.R - 1) / g.R;
    g.WSp = (bw + g.R - 2) / (g.R + 1) + 2;
    if (g.WSp > g.NS) g.WSp = g.NS;
    if (g.WSp < 1) g.WSp = 1;
    g.K = 1;
    g.U = G == 1 ? BRX_U1 : G == 2 ? 4 : G == 4 ? 2 : 1;           /* G * U = 8 words of {Pv, Ph} per lane and trip */
    g.t_end = (T + g.NS - 1 + g.U - 1) / g.U * g.U;                 /* whole trips: the last one may run past time T + NS - 1 */
    g.H = BRX_H_ALL; g.slope = 0;
    if (hmul != 0 && G <= 16 && T > 0 && (uint64_t)Q < ((uint64_t)T << 11)) {
        const int H = hmul > 0 ? hmul * (int)brx_isqrt((uint32_t)k) + 24 : 8;
        const int slots = (2 * H + g.R - 1) / g.R + 1;      /* superblocks that can meet the window in one store row */
        if (slots < g.WSp) { g.WSp = slots; g.H = H; g.slope = (uint32_t)(((uint64_t)Q << 20) / (uint64_t)T); }
    }
    g.WSrow = g.WSp; g.slot0 = 0;
    return g;
}
__host__ __device__ inline BrxGeom brx_make_geom(int Q, int T, int k, int hmul = 0) { return brx_make_geom_span(Q, T, k, hmul, 56, BRX_GEOM_MAXG); }

/* Is superblock s written for the column group represented by column jrep?  (jrep = the column itself when
 * K = 1; the column in the middle of its trip otherwise: brx_jrep.)  For a fixed store row the rows
 * R*s - c(jrep) grow by at least R per superblock, so at most (2H + R - 1)/R + 1 consecutive superblocks
 * pass -- they land in distinct slots s % WSp. */
/* The test itself, in the form the forward passes keep it (ONE definition for them and for the traceback -- ADVICE r4):
   keep_base = R s + H + R - 1 (per superblock), keep_lim = 2 H + R - 1, jrep as below. */
__host__ __device__ __forceinline__ bool brx_keep_trip(uint32_t slope, int keep_base, uint32_t keep_lim, int jrep) {
    return (uint32_t)(keep_base - (int)(uint32_t)(((uint64_t)(uint32_t)jrep * (uint64_t)slope) >> 20)) <= keep_lim;
}
__host__ __device__ __forceinline__ bool brx_stored(const BrxGeom &g, int s, int jrep) {
    return brx_keep_trip(g.slope, g.R * s + g.H + g.R - 1, (uint32_t)(2 * g.H + g.R - 1), jrep);
}
/* the column that stands for loop trip tau (columns U tau + 1 - s .. U tau + U - s) of superblock s: its middle one, never below 0 */
__host__ __device__ __forceinline__ int brx_jrep_trip(int U, int tau, int s) {
    const int jr = U * tau + U / 2 - s;
    return jr > 0 ? jr : 0;
}
/* the column that stands for column j of superblock s in the windowed-store test: the middle one of the loop trip that
   computes it (time j + s), never below 0 */
__host__ __device__ __forceinline__ int brx_jrep(const BrxGeom &g, int s, int j) {
    if (g.U == 1) return j;
    return brx_jrep_trip(g.U, (j + s - 1) / g.U, s);
}

/* 8-byte units of traceback storage an alignment needs */
__host__ __device__ inline uint64_t brx_tb_units(const BrxGeom &g) {
    return (uint64_t)(g.t_end + 1) * (uint64_t)g.WSrow * (uint64_t)g.G;
}
/* lanes holding more than BRX_REGPEQ_MAXG words read their equality masks from a table
   [5][NW] (u32) that the wave builds behind the traceback store */
#define BRX_REGPEQ_MAXG 0
__host__ __device__ inline uint64_t brx_peq_units(const BrxGeom &g) {
    return g.G > BRX_REGPEQ_MAXG ? ((uint64_t)5 * (uint64_t)g.NW * 4 + 7) / 8 + (uint64_t)64 * (uint64_t)g.G : 0;
}
__host__ __device__ inline uint64_t brx_align_units(const BrxGeom &g) { return brx_tb_units(g) + brx_peq_units(g); }

/* scratch units of a read's final alignment (query = mutated read, m bytes; target = fragment, n bytes; ub = the
 * proven bound on their distance): traceback store + the col_of[] array of the qscore stage.  *too_wide: the band
 * does not fit the aligner at all (BRX_RS_BAND). */
__host__ __device__ inline uint64_t brx_final_units(uint32_t m, uint32_t n, uint32_t ub, int hmul, bool *too_wide) {
    uint64_t units = 0;
    *too_wide = false;
    if (m) {
        const BrxGeom g = brx_make_geom((int)m, (int)n, (int)ub, hmul);
        if (g.G == 0) *too_wide = true; else units = brx_align_units(g);
    }
    return units + ((uint64_t)m * 4 + 7) / 8 + 2;
}

__device__ inline int brx_jfirst(const BrxGeom &g, int s) {
    int j = g.R * s - g.dhi + 1;
    return j < 1 ? 1 : j;
}
__device__ inline int brx_jlast(const BrxGeom &g, int s) {
    long long j = (long long)g.R * (s + 1) - g.dlo;
    return j > g.T ? g.T : (int)j;
}

/* equality mask of word w for symbol c, straight from the query bytes (symbols outside 0..4) */
__device__ inline uint32_t brx_eq_slow(const uint8_t *Qs, int Q, int w, uint32_t c) {
    uint32_t m = 0;
    int base = 32 * w;
    for (int r = 0; r < 32; ++r) {
        int idx = base + r;
        if (idx < Q && Qs[idx] == c) m |= 1u << r;
    }
    return m;
}

/* equality mask of word w for a symbol outside 0..4 (IUPAC codes other than N): rare, kept out of line */
__device__ __noinline__ uint32_t brx_eq_rare(const uint8_t *Qs, int Q, int w, uint32_t c) {
    return brx_eq_slow(Qs, Q, w, c);
}

/* lane i receives the value of lane (i - 1) mod 64: DPP wave_ror:1, two VALU cycles, no LDS traffic */
__device__ __forceinline__ int brx_from_lane_above(int v) {
    return __builtin_amdgcn_mov_dpp(v, 0x13C /* wave_ror:1 */, 0xF, 0xF, false);     /* every lane is written: no old value to preserve (update_dpp costs a move for it) */
}

#define BRX_RING_BYTES 2048         /* per-wave LDS window of target bytes (eight 256-byte chunks; the one-column-per-trip passes use two) */
/* One wave per workgroup, so one window per workgroup.  File scope keeps the LDS address space
 * visible to the compiler (ds_read_u8 / ds_write_b32); a generic or volatile pointer to it turns
 * every access into a flat load that waits on vmcnt -- exactly what the window is there to avoid. */
__shared__ uint32_t brx_ring32[BRX_RING_BYTES / 4];
/* ---------------------------------------------------------------------------------------------
 * forward pass: fills tb[(t*WSp + s%WSp)*G + g] = {Pv after column j, Ph before its shift}
 * Qs and Ts must be readable up to 16 bytes past their ends (buffers are padded by the caller) and
 * 4-byte aligned.
 *
 * Memory discipline of the loop (this is what bounds a step): the only steady-state memory
 * operations are the traceback STORES.  On gfx9 loads and stores retire through one in-order
 * counter (vmcnt), so a load inside the loop makes the wave wait for the write acknowledgements
 * of every store before it; r01a measured ~1700 cycles per step that way.  Target bytes therefore
 * come from a 512-byte LDS window that the wave refills 256 bytes at a time, split-phase (global
 * load issued 64 steps before the data is written to LDS), the per-step byte is read from LDS one
 * step ahead, and the carry between superblocks travels by DPP instead of ds_bpermute.  The
 * remaining global loads are the 16 bytes of bit planes per word that a lane reads when its superblock
 * enters the band (once per R columns per wave; brx_build_planes).
 * ------------------------------------------------------------------------------------------- */
__device__ __forceinline__ uint32_t brx_bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }   /* v_bfi_b32 */

/* sign-extended bit b of v: all ones or zero (v_bfe_i32) */
__device__ __forceinline__ uint32_t brx_bit_mask(uint32_t v, int b) { return (uint32_t)((int32_t)(v << (31 - b)) >> 31); }

/* The query word of a lane as bit planes instead of one equality mask per symbol: code bit 0, code bit 1, "row holds
   A/C/G/T" and "row holds N" (rows past the end of the query are in neither).  The equality mask of a column whose
   target symbol is A/C/G/T is then two three-input bit operations on the symbol's two code bits -- the five-way select
   over per-symbol masks cost 12 instructions per column, a quarter of the whole column update. */
struct BrxQPlanes { uint32_t lo, hi, acgt, n; };
/* k0 / k1: bit 0 / bit 1 of the target symbol, as masks */
__device__ __forceinline__ uint32_t brx_eq_acgt(const BrxQPlanes &p, uint32_t k0, uint32_t k1) {
    return ~(p.lo ^ k0) & ~(p.hi ^ k1) & p.acgt;
}

/* The planes of EVERY query word, built by the whole wave before the forward pass: 64 consecutive rows per ballot, 16 bytes per
   word ({lo, hi}, {acgt, n}) in the table area behind the traceback store (brx_peq_units: every alignment has it).  A lane
   whose superblock enters the band then loads 16 bytes per word; building the planes there, from 32 bytes with ~330
   instructions per word that the whole wave issues for ONE lane's benefit, was a quarter of the forward pass. */
__device__ inline void brx_build_planes(const uint8_t *__restrict__ Qs, const BrxGeom &g, uint2 *__restrict__ area) {
    const int lane = threadIdx.x & 63;
    for (int base = 0; base < g.Q; base += 256) {
        uint32_t code[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int r = base + 64 * u + lane; code[u] = r < g.Q ? (uint32_t)Qs[r] : 0u; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = base + 64 * u + lane < g.Q;
            const uint32_t c = code[u];
            const unsigned long long lo = __ballot(ok && (c & 1u)), hi = __ballot(ok && (c & 2u));
            const unsigned long long ac = __ballot(ok && c < 4u), nn = __ballot(ok && c == 4u);
            const int w = (base + 64 * u) / 32 + lane;
            if (lane < 2 && w < g.NW) {
                area[2 * w] = make_uint2((uint32_t)(lo >> (32 * lane)), (uint32_t)(hi >> (32 * lane)));
                area[2 * w + 1] = make_uint2((uint32_t)(ac >> (32 * lane)), (uint32_t)(nn >> (32 * lane)));
            }
        }
    }
}
__device__ __forceinline__ BrxQPlanes brx_load_planes(const uint2 *__restrict__ planes, int w) {
    const BRX_GLOBAL uint64_t *pl = (const BRX_GLOBAL uint64_t *)planes;      /* global, not flat: the loop's loads must not wait on LDS */
    const uint64_t a = pl[2 * w], b = pl[2 * w + 1];
    return BrxQPlanes{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
}

template <int G>
__device__ void brx_align_forward(const uint8_t *__restrict__ Qs, const uint8_t *__restrict__ Ts,
                                  const BrxGeom g, uint2 *__restrict__ tb, const uint2 *__restrict__ planes, uint32_t *prog = nullptr) {
    const int lane = threadIdx.x & 63;
    uint32_t *const ring32 = brx_ring32;
    constexpr int NEVER = 0x7FFFFFFF;
    /* A lane works on superblock s during time steps [tf, tl] (column j = t - s), then hops to s + 64.
     * The band is narrower than 62 superblocks (brx_make_geom), so whenever the lane above was active
     * in the previous step it held superblock s - 1 and worked on this same column: the carry word
     * needs no tag.  carry: 0 = lane above idle (top of the band: +1), 1/2/3 = hout -1/0/+1. */
    int s = lane;
    int tf = NEVER, tl = NEVER, slot = 0;
    uint32_t tspan = 0;                              /* step t is computed iff (uint32_t)(t - tf) <= tspan */
    if (s < g.NS) { tf = brx_jfirst(g, s) + s; tl = brx_jlast(g, s) + s; slot = s % g.WSp; tspan = tl >= tf ? (uint32_t)(tl - tf) : 0u; if (tl < tf) tf = NEVER; }
    const int slot_step = 64 % g.WSp;                /* a lane's next superblock is s + 64: its slot moves by this much (mod WSp) */
    uint32_t Pv[G], Mv[G];
    BrxQPlanes qp[G];
#pragma unroll
    for (int x = 0; x < G; ++x) { Pv[x] = 0xFFFFFFFFu; Mv[x] = 0; qp[x] = BrxQPlanes{0u, 0u, 0u, 0u}; }
    uint32_t carry = 2u;                             /* an idle lane hands on +1 */
    /* windowed traceback store (brx_stored), incrementally: acc = slope * column, exact in 64 bits */
    const uint32_t keep_lim = (uint32_t)(2 * g.H + g.R - 1);
    int keep_base = g.R * s + g.H + g.R - 1;
    int64_t acc = (int64_t)(1 - s) * (int64_t)g.slope;

    /* target window: chunk c = target bytes [256c, 256c+256) lives in half c & 1 of the first 512 ring bytes; `odd` remembers
       which halves hold a symbol outside A,C,G,T,N (IUPAC codes: slow equality path) */
    auto fetch_chunk = [&](int c) -> uint32_t {
        const int idx = 256 * c + 4 * lane;
        return (idx + 4 <= g.T + 16) ? *reinterpret_cast<const uint32_t *>(Ts + idx) : 0xFEFEFEFEu;
    };
    auto chunk_odd = [&](int c, uint32_t v) -> uint32_t {
        const int idx = 256 * c + 4 * lane;
        bool o = false;
#pragma unroll
        for (int b = 0; b < 4; ++b) o |= idx + b < g.T && ((v >> (8 * b)) & 0xFFu) > 4u;
        return __ballot(o) != 0ull ? 1u : 0u;
    };
    uint32_t pending = fetch_chunk(0);
    ring32[lane] = pending;
    uint32_t odd = chunk_odd(0, pending);
    pending = fetch_chunk(1);
    ring32[64 + lane] = pending;
    odd |= chunk_odd(1, pending) << 1;
    int s_top = 0;                                   /* first superblock still inside the band (wave-uniform) */
    int t_top = brx_jlast(g, 0) + 1;                 /* time step at which s_top leaves the band              */
    uint32_t cnext = reinterpret_cast<const uint8_t *>(ring32)[(uint32_t)(0 - s) & 511u];   /* column 1 - s */
    /* Superblocks enter and leave the band in order (brx_jfirst / brx_jlast grow with s), so the next time step at which one
       enters (e_s, e_t) or leaves (h_s, h_t) is scalar bookkeeping; a wave-wide minimum over the lanes' tf / tl at every
       event was three 6-step shuffle reductions per superblock. */
    int e_s = 0, h_s = 0;
    while (e_s < g.NS && brx_jlast(g, e_s) < brx_jfirst(g, e_s)) ++e_s;
    int next_entry = e_s < g.NS ? brx_jfirst(g, e_s) + e_s : NEVER;
    int next_hop = g.NS > 0 ? brx_jlast(g, 0) + 0 : NEVER;

    const size_t step_units = (size_t)g.WSp * (size_t)G;
    uint2 *dst = tb + ((size_t)1 * (size_t)g.WSp + (size_t)slot) * (size_t)G;
    for (int t = 1; t <= g.t_end; ++t, dst += step_units, acc += (int64_t)g.slope) {
        /* ---- refill of the target window, keyed on the newest column in use (scalar code) ---- */
        while (__builtin_expect(s_top < g.NS - 1 && t >= t_top, 0)) { s_top += 1; t_top = brx_jlast(g, s_top) + s_top + 1; }
        const int front = t - s_top - 1;             /* 0-based target index of the newest column */
        if (__builtin_expect((front & 63) == 0, 0)) {
            if ((front & 255) == 128) pending = fetch_chunk((front >> 8) + 1);
            else if ((front & 255) == 192) {
                const int c = (front >> 8) + 1;
                ring32[(c & 1) * 64 + lane] = pending;
                odd = (odd & ~(1u << (c & 1))) | (chunk_odd(c, pending) << (c & 1));
            }
        }

        /* ---- a superblock enters the band (one lane every R steps): build its equality masks ---- */
        if (__builtin_expect(t == next_entry, 0)) {
            if (t == tf) {
#pragma unroll
                for (int x = 0; x < G; ++x) {
                    Pv[x] = 0xFFFFFFFFu; Mv[x] = 0;          /* cells below the band grow by +1 per row */
                    const int w = s * G + x;
                    qp[x] = w < g.NW ? brx_load_planes(planes, w) : BrxQPlanes{0u, 0u, 0u, 0u};
                }
            }
            do { ++e_s; } while (e_s < g.NS && brx_jlast(g, e_s) < brx_jfirst(g, e_s));
            next_entry = e_s < g.NS ? brx_jfirst(g, e_s) + e_s : NEVER;
        }

        /* ---- the column update: straight-line VALU code, lanes outside the band discard the result ---- */
        const uint32_t nb = (uint32_t)brx_from_lane_above((int)carry);
        const uint32_t c = cnext;
        const bool act = (uint32_t)(t - tf) <= tspan;                          /* tf <= t <= tl */
        uint32_t hp = nb >> 1, hm = nb & 1u;                                   /* carry: bit 1 = hout +1, bit 0 = hout -1 */
        const uint32_t k0 = 0u - (c & 1u), k1 = 0u - ((c >> 1) & 1u), k4 = 0u - ((c >> 2) & 1u);
        const bool rare = __builtin_expect(odd != 0u, 0) && __ballot(act && c > 4u) != 0ull;
        const bool keep = (uint32_t)(keep_base - (int)(uint32_t)((uint64_t)acc >> 20)) <= keep_lim;   /* windowed traceback store */
        const bool put = act && keep;
#pragma unroll
        for (int x = 0; x < G; ++x) {
            /* Words past the last query row (only the last superblock has them) are computed like the others: their planes
               are empty, their stores land in slots nothing reads, and the carry they hand on leaves the matrix. */
            uint32_t Eq = brx_bfi(k4, qp[x].n, brx_eq_acgt(qp[x], k0, k1));
            if (rare) { if (act && c > 4u && s * G + x < g.NW) Eq = brx_eq_rare(Qs, g.Q, s * G + x, c); }
            const uint32_t pv0 = Pv[x], mv0 = Mv[x];
            const uint32_t Xv = Eq | mv0;
            const uint32_t Eq2 = Eq | hm;
            const uint32_t Xh = (((Eq2 & pv0) + pv0) ^ pv0) | Eq2;
            const uint32_t Ph = mv0 | ~(Xh | pv0);
            const uint32_t Mh = pv0 & Xh;
            const uint32_t PhS = (Ph << 1) | hp;
            const uint32_t MhS = (Mh << 1) | hm;
            const uint32_t pv = MhS | ~(Xv | PhS);
            const uint32_t mv = PhS & Xv;
            Pv[x] = act ? pv : pv0;
            Mv[x] = act ? mv : mv0;
            if (put) dst[x] = make_uint2(pv, Ph);
            hp = Ph >> 31;
            hm = Mh >> 31;
        }
        carry = act ? ((hp << 1) | hm) : 2u;

        /* ---- a superblock leaves the band: its lane takes superblock s + 64 ---- */
        if (__builtin_expect(t == next_hop, 0)) {
            if (t >= tl) {
                s += 64;
                tspan = 0;
                if (s < g.NS) {
                    tf = brx_jfirst(g, s) + s; tl = brx_jlast(g, s) + s;
                    if (tl >= tf) tspan = (uint32_t)(tl - tf); else tf = NEVER;
                    int nslot = slot + slot_step;
                    if (nslot >= g.WSp) nslot -= g.WSp;
                    dst += ((ptrdiff_t)nslot - (ptrdiff_t)slot) * (ptrdiff_t)G;
                    slot = nslot;
                } else { tf = NEVER; tl = NEVER; }
                keep_base = g.R * s + g.H + g.R - 1;
                acc = (int64_t)(t - s) * (int64_t)g.slope;     /* the loop header adds one step before the next trip */
            }
            ++h_s;
            next_hop = h_s < g.NS ? brx_jlast(g, h_s) + h_s : NEVER;
        }
        cnext = reinterpret_cast<const uint8_t *>(ring32)[(uint32_t)(t - s) & 511u];   /* column t + 1 - s */
    }
    (void)prog;
}

/* ---------------------------------------------------------------------------------------------
 * traceback.  ops_end: one past the last byte of the caller's ops area (written backwards), or
 * nullptr for counts only.  Returns true on success; *n_cols, *n_match valid for all lanes.
 * ------------------------------------------------------------------------------------------- */
__device__ __forceinline__ uint32_t brx_from_lane_below(uint32_t v) { return (uint32_t)__shfl_down((int)v, 1, 64); }   /* lane l <- lane l + 1 */

/* Chained rounds.  A round loads, for lane l, the traceback word of cell (i - l, j - l) and the two sequence bytes --
 * one memory round trip for 64 cells of the current diagonal.  The plain scheme used one round per run of diagonal
 * moves: every indel cost a round trip.  But after an 'I' (up) at lane b the path continues on cells one ROW up in the
 * SAME columns, and after a 'D' (left) on cells one COLUMN left in the same rows -- and those cells' traceback bits
 * are, word boundaries aside, already in registers: in the lane's own word (I) or in the word of the lane below (D),
 * which loaded the same column one row up.  So the round goes on: shift the query byte (I) or the word, its tag and the
 * target byte (D) down one lane and keep walking, until a lane's word no longer covers the cell it needs (tags
 * decide) or the lanes run out.  Rounds drop from (#indels + columns / 64) to about columns / 64 + word crossings. */
__device__ inline bool brx_align_traceback(const uint8_t *__restrict__ Qs, const uint8_t *__restrict__ Ts,
                                           const BrxGeom g, const uint2 *__restrict__ tb,
                                           uint8_t *ops_end, int *n_cols, int *n_match, uint32_t *prog = nullptr) {
    const int lane = threadIdx.x & 63;
    int i = g.Q, j = g.T;
    int pos = 0, nmatch = 0;
    bool ok = true;
    const int shiftR = 31 - __clz(g.R);          /* R is a power of two */
    long long guard = (long long)g.Q + (long long)g.T + 8;   /* every round retires >= 1 column or row */
    while (ok && i > 0 && j > 0) {
        if (--guard < 0) { ok = false; break; }
        BRX_PROG(prog, 5, (uint32_t)guard);
        /* ---- load: lane l <- cell (i - l, j - l) ---- */
        const int I0 = i, J0 = j;
        uint32_t w_pv = 0, w_ph = 0, qch = 0x100u, tch = 0x200u;
        int tag_word = -1, tag_col = -1, q_row = -1;
        {
            const int ci = I0 - lane, cj = J0 - lane;
            if (ci >= 1 && cj >= 1) {
                const int s = (ci - 1) >> shiftR;
                if (cj >= brx_jfirst(g, s) && cj <= brx_jlast(g, s) && brx_stored(g, s, brx_jrep(g, s, cj))) {
                    const int x = ((ci - 1) & (g.R - 1)) >> 5;
                    const uint2 v = tb[((size_t)(cj + g.K * s) * (size_t)g.WSrow + (size_t)(g.slot0 + s % g.WSp)) * (size_t)g.G + (size_t)x];
                    w_pv = v.x; w_ph = v.y; tag_word = (ci - 1) >> 5; tag_col = cj;
                }
                qch = Qs[ci - 1]; tch = Ts[cj - 1]; q_row = ci;
            }
        }
        /* ---- walk: lanes >= base hold the not yet consumed part of the diagonal, shifted nI rows / nD columns ---- */
        int base = 0, nI = 0, nD = 0;
        for (;;) {
            const int ci = I0 - lane - nI, cj = J0 - lane - nD;
            const bool mine = lane >= base && ci >= 1 && cj >= 1;
            bool cell = false;                                     /* inside the band and the stored window */
            if (mine) {
                const int s = (ci - 1) >> shiftR;
                cell = cj >= brx_jfirst(g, s) && cj <= brx_jlast(g, s) && brx_stored(g, s, brx_jrep(g, s, cj));
            }
            const bool inhand = cell && tag_word == ((ci - 1) >> 5) && tag_col == cj && q_row == ci;
            const int bit = (ci - 1) & 31;
            const bool up = inhand && ((w_pv >> bit) & 1u), left = inhand && ((w_ph >> bit) & 1u);
            const bool diag = inhand && !up && !left;
            const bool eq = qch == tch;
            const unsigned long long dm = __ballot(diag) >> base, em = __ballot(eq) >> base;
            const int room = 64 - base;
            int run = (~dm == 0ull) ? 64 : __ffsll((long long)~dm) - 1;
            if (run > room) run = room;
            if (lane >= base && lane < base + run && ops_end) ops_end[-(pos + (lane - base)) - 1] = eq ? BRX_OP_EQ : BRX_OP_X;
            const unsigned long long runmask = run == 64 ? ~0ull : ((1ull << run) - 1ull);
            nmatch += __popcll(em & runmask);
            pos += run; i -= run; j -= run; base += run;
            if (base >= 64 || i <= 0 || j <= 0) break;
            /* lane `base` is the first cell that is not a diagonal move */
            const unsigned long long cm = __ballot(cell), hm = __ballot(inhand), um = __ballot(up);
            if (!((cm >> base) & 1ull)) { ok = false; break; }              /* the path leaves the band / the stored window */
            if (!((hm >> base) & 1ull)) break;                             /* its word is not in registers: next round loads it */
            const bool isup = (um >> base) & 1ull;
            if (lane == 0 && ops_end) ops_end[-pos - 1] = isup ? BRX_OP_I : BRX_OP_D;
            pos += 1;
            if (isup) {
                i -= 1; nI += 1;
                const uint32_t q2 = brx_from_lane_below(qch); const int r2 = (int)brx_from_lane_below((uint32_t)q_row);
                if (lane < 63) { qch = q2; q_row = r2; } else q_row = -1;
            } else {
                j -= 1; nD += 1;
                const uint32_t a2 = brx_from_lane_below(w_pv), b2 = brx_from_lane_below(w_ph), t2 = brx_from_lane_below(tch);
                const int tw2 = (int)brx_from_lane_below((uint32_t)tag_word), tc2 = (int)brx_from_lane_below((uint32_t)tag_col);
                if (lane < 63) { w_pv = a2; w_ph = b2; tch = t2; tag_word = tw2; tag_col = tc2; } else { tag_word = -1; tag_col = -1; }
            }
        }
    }
    if (ok) {
        if (ops_end) {
            for (int x = lane; x < i; x += 64) ops_end[-(pos + x) - 1] = BRX_OP_I;
            for (int x = lane; x < j; x += 64) ops_end[-(pos + x) - 1] = BRX_OP_D;
        }
        pos += i + j;
    }
    *n_cols = pos; *n_match = nmatch;
    return ok;
}

/* dispatch on words-per-lane */
/* Wide bands (more than BRX_REGPEQ_MAXG words per lane; reads that are both very long and very
 * inaccurate): same systolic schedule, but the G words of a lane live in a per-wave state array
 * st[x*64 + lane] = {Pv, Mv} behind the Peq table instead of registers.  Rare, so simplicity wins. */
__device__ inline void brx_align_forward_wide(const uint8_t *__restrict__ Qs, const uint8_t *__restrict__ Ts,
                                              const BrxGeom g, uint2 *__restrict__ tb,
                                              const uint32_t *__restrict__ peq, uint2 *__restrict__ st) {
    const int lane = threadIdx.x & 63;
    const int G = g.G;
    int s = lane;
    bool has = s < g.NS;
    int jf = has ? brx_jfirst(g, s) : 0, jl = has ? brx_jlast(g, s) : -1;
    int slot = has ? s % g.WSp : 0;
    int hout_last = 0, s_last = -1;
    bool active_last = false;
    for (int t = 1; t <= g.t_end; ++t) {
        int packed = (s_last << 3) | (active_last ? 4 : 0) | (hout_last + 1);
        int nb = __shfl(packed, (lane + 63) & 63, 64);
        int j = t - s;
        bool active = has && j >= jf && j <= jl;
        if (active) {
            bool fresh = (j == jf);
            uint32_t c = Ts[j - 1];
            int hin = 1;
            if (s > 0 && (nb >> 3) == s - 1 && (nb & 4)) hin = (nb & 3) - 1;
            uint32_t hp = hin > 0 ? 1u : 0u, hm = hin < 0 ? 1u : 0u;
            uint2 *dst = tb + ((size_t)t * (size_t)g.WSp + (size_t)slot) * (size_t)G;
            for (int x = 0; x < G; ++x) {
                int w = s * G + x;
                if (w >= g.NW) break;
                uint2 pm = fresh ? make_uint2(0xFFFFFFFFu, 0u) : st[x * 64 + lane];
                uint32_t Eq = c < 5 ? peq[(size_t)c * (size_t)g.NW + (size_t)w] : brx_eq_slow(Qs, g.Q, w, c);
                uint32_t pv = pm.x, mv = pm.y;
                uint32_t Xv = Eq | mv;
                uint32_t Eq2 = Eq | hm;
                uint32_t Xh = (((Eq2 & pv) + pv) ^ pv) | Eq2;
                uint32_t Ph = mv | ~(Xh | pv);
                uint32_t Mh = pv & Xh;
                uint32_t op = Ph >> 31, om = Mh >> 31;
                uint32_t PhS = (Ph << 1) | hp;
                uint32_t MhS = (Mh << 1) | hm;
                pv = MhS | ~(Xv | PhS);
                mv = PhS & Xv;
                st[x * 64 + lane] = make_uint2(pv, mv);
                dst[x] = make_uint2(pv, Ph);
                hp = op; hm = om;
            }
            hout_last = (int)hp - (int)hm;
        }
        active_last = active;
        s_last = s;
        if (has && j >= jl) {
            s += 64;
            has = s < g.NS;
            if (has) { jf = brx_jfirst(g, s); jl = brx_jlast(g, s); slot = s % g.WSp; }
        }
    }
}

__device__ inline void brx_build_peq(const uint8_t *Qs, const BrxGeom &g, uint32_t *peq) {
    const int lane = threadIdx.x & 63;
    for (int w = lane; w < g.NW; w += 64) {
        uint32_t m[5] = {0, 0, 0, 0, 0};
        for (int r = 0; r < 32; ++r) {
            int idx = 32 * w + r;
            uint32_t code = idx < g.Q ? Qs[idx] : 0xFFu;
#pragma unroll
            for (int c = 0; c < 5; ++c) m[c] |= (uint32_t)(code == (uint32_t)c) << r;
        }
#pragma unroll
        for (int c = 0; c < 5; ++c) peq[(size_t)c * (size_t)g.NW + (size_t)w] = m[c];
    }
}

/* ---------------------------------------------------------------------------------------------
 * forward pass for G = 1: a lag of ONE column between neighbouring superblocks, U = 8 columns per loop trip.
 *
 * Superblock s handles column j at time t = j + s (K = 1), so the lane above has finished this very column one
 * micro-step earlier and its carry comes over by DPP in every micro-step.  A loop trip is eight micro-steps, t = 8 tau + 1
 * .. 8 tau + 8: the loop bookkeeping (time tests, ring refill, lane hops, pointer arithmetic, activity and window tests) is
 * paid once per eight columns as before, but a read of T columns and NS superblocks takes (T + NS - 1) / 8 trips instead of
 * T / 8 + NS - 1 -- the round-3 schedule (a lag of one TRIP, the eight carries in one word) spent a quarter of its trips on
 * fill and drain.  Traceback row of column j of superblock s is its time j + s = 8 tau + c + 1: the same for every lane of a
 * micro-step, so the stores stay slot-contiguous.
 *
 * Lanes work in WHOLE trips, and a trip's columns 8 tau + c + 1 - s differ from lane to lane.  Widening a band window to trip
 * boundaries computes up to seven columns more than the Ukkonen band on either side (harmless: upper bounds), but at the
 * top-left corner the extra columns of superblocks 1 .. (dhi + 6) / 32 would lie left of column 1: the first trips
 * (tau <= tau_pro, at most eight) take the rolled path, which skips columns below 1.
 * ------------------------------------------------------------------------------------------- */
__device__ __forceinline__ uint32_t brx_funnel_bytes(uint32_t hi, uint32_t lo, uint32_t nbytes) {   /* v_alignbyte_b32 */
    return (uint32_t)((((uint64_t)hi << 32) | (uint64_t)lo) >> (8u * (nbytes & 3u)));
}
template <int U, int G>
__device__ inline void brx_align_forward_u(const uint8_t *__restrict__ Qs, const uint8_t *__restrict__ Ts,
                                           const BrxGeom g, uint2 *__restrict__ tb, const uint2 *__restrict__ planes) {
    static_assert(U == 8 || U == 4 || U == 2, "columns per trip");
    static_assert(G * U <= 8, "G words per lane, U columns per trip");
    constexpr int LU = U == 8 ? 3 : U == 4 ? 2 : 1;  /* log2 U */
    const int lane = threadIdx.x & 63;
    uint32_t *const ring32 = brx_ring32;
    /* the store base is the same in every lane: say so (and that it is global memory), and the traceback stores take the
       scalar-base form -- row address in SGPRs + a 32-bit lane offset -- instead of a 64-bit vector add per column */
    const uint64_t tb_addr = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)tb >> 32)) << 32) |
                             (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint64_t)tb);
    constexpr int NEVER = 0x7FFFFFFF;
    int s = lane;
    int slot = lane % g.WSp;                         /* s % WSp, kept incrementally: s moves by 64 */
    const int slot_step = 64 % g.WSp;
    uint32_t slot8 = 0;                              /* byte offset of the lane's slot in a traceback row */
    int tf = NEVER, tl = NEVER;                     /* first / last loop trip of the lane's superblock: trip tau is computed
                                                       iff (uint32_t)(tau - tf) <= tspan                                  */
    uint32_t tspan = 0;
    /* windowed traceback store (brx_stored): superblock s is written in trip tau iff
       (uint32_t)(keep_base - ((slope * brx_jrep) >> 20)) <= keep_lim, brx_jrep = max(U tau + U / 2 - s, 0) */
    const uint32_t keep_lim = (uint32_t)(2 * g.H + g.R - 1);
    int keep_base = 0;
    auto window = [&]() {                            /* everything that depends on s */
        tf = NEVER; tl = NEVER; tspan = 0;
        if (s < g.NS) {
            const int jf = brx_jfirst(g, s), jl = brx_jlast(g, s);
            slot8 = 8u * (uint32_t)G * (uint32_t)slot;
            tl = (jl + s - 1) >> LU;
            if (jl >= jf) { tf = (jf + s - 1) >> LU; tspan = (uint32_t)(tl - tf); }   /* empty window: never active, but it still hops at tl */
        }
        keep_base = g.R * s + g.H + g.R - 1;
    };
    window();
    uint32_t Pv[G], Mv[G];
    BrxQPlanes qp[G];
#pragma unroll
    for (int x = 0; x < G; ++x) { Pv[x] = 0xFFFFFFFFu; Mv[x] = 0; qp[x] = BrxQPlanes{0u, 0u, 0u, 0u}; }
    constexpr uint32_t IDLE = 0x80000000u;          /* carry word: bit 31 = hout is +1, bit 0 = hout is -1.  An idle lane hands on +1
                                                       (the cells above the band grow by one per column) */
    uint32_t carry = IDLE;

    auto fetch_chunk = [&](int c) -> uint32_t {
        const int idx = 256 * c + 4 * lane;
        return (idx + 4 <= g.T + 16) ? *reinterpret_cast<const uint32_t *>(Ts + idx) : 0xFEFEFEFEu;
    };
    auto chunk_odd = [&](int c, uint32_t v) -> uint32_t {       /* any symbol other than A/C/G/T in the chunk (N and IUPAC codes) */
        const int idx = 256 * c + 4 * lane;
        bool o = false;
#pragma unroll
        for (int b = 0; b < 4; ++b) o |= idx + b < g.T && ((v >> (8 * b)) & 0xFFu) > 3u;
        return __ballot(o) != 0ull ? 1u : 0u;
    };
    uint32_t odd = 0, pending = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {                   /* chunks 0..2; chunk c lives in ring slot c & 7 */
        pending = fetch_chunk(c);
        ring32[(c & 7) * 64 + lane] = pending;
        odd |= chunk_odd(c, pending) << c;
    }
    int s_top = 0;                                  /* first superblock still inside the band (uniform) */
    int tl_top = (brx_jlast(g, 0) + 0 - 1) >> LU;   /* its last trip                                     */
    int next_mark = 64;                             /* next multiple of 64 the front byte crosses: refill events */
    /* entries and exits happen in superblock order: scalar bookkeeping.  Near the corners several superblocks can enter or
       leave in one trip (their windows are clamped to column 1 / column T and their times differ by one column) */
    auto tf_of = [&](int x) { return (brx_jfirst(g, x) + x - 1) >> LU; };
    auto tl_of = [&](int x) { return (brx_jlast(g, x) + x - 1) >> LU; };
    int e_s = 0, h_s = 0;
    while (e_s < g.NS && brx_jlast(g, e_s) < brx_jfirst(g, e_s)) ++e_s;
    int next_entry = e_s < g.NS ? tf_of(e_s) : NEVER;
    int next_hop = g.NS > 0 ? tl_of(0) : NEVER;
    const int tau_end = (g.T + g.NS - 1 - 1) >> LU;
    /* trips in which a lane of the first 64 superblocks would compute columns left of column 1 */
    int sv = (g.dhi + U - 2) / g.R;
    if (sv > g.NS - 1) sv = g.NS - 1;
    if (sv > 63) sv = 63;
    const int tau_pro = sv >= 1 ? tf_of(sv) : -1;
    const size_t wsp = (size_t)g.WSp;
    /* traceback rows U tau + 1 .. U tau + U (uniform addresses); a lane writes G words at byte slot8 of each */
    BRX_GLOBAL char *row0 = (BRX_GLOBAL char *)((BRX_GLOBAL uint64_t *)tb_addr + wsp * (size_t)G);
    const size_t row_bytes = 8 * wsp * (size_t)G;
    const size_t trip_bytes = (size_t)U * row_bytes;
    /* the lane's U target bytes of a trip: ring bytes U tau - s .. U tau - s + U - 1, an unaligned window of two or three words */
    auto ring_bytes = [&](int tau_, uint32_t *x0, uint32_t *x1) {
        const uint32_t b0 = (uint32_t)(U * tau_ - s);
        const uint32_t d = b0 >> 2;
        const uint32_t w0 = ring32[d & (BRX_RING_BYTES / 4 - 1)], w1 = ring32[(d + 1) & (BRX_RING_BYTES / 4 - 1)];
        *x0 = brx_funnel_bytes(w1, w0, b0);
        if constexpr (U == 8) { const uint32_t w2 = ring32[(d + 2) & (BRX_RING_BYTES / 4 - 1)]; *x1 = brx_funnel_bytes(w2, w1, b0); }
        else *x1 = 0u;
    };
    uint32_t xn0, xn1;
    ring_bytes(0, &xn0, &xn1);
    for (int tau = 0; tau <= tau_end; ++tau, row0 += trip_bytes) {
        /* ---- refill of the target window, keyed on the newest byte in use (scalar code) ---- */
        while (__builtin_expect(s_top < g.NS - 1 && tau > tl_top, 0)) { s_top += 1; tl_top = tl_of(s_top); }
        const int fb = U * tau + U - s_top;         /* one past the newest byte this trip reads */
        if (__builtin_expect(fb >= next_mark, 0)) {
            /* The band spans fewer than 62 superblocks, so the oldest byte still read is less than 80 bytes behind the
               front: when the front enters chunk m, chunk m + 2 takes the slot of the long dead chunk m - 6; its load was
               issued a quarter chunk earlier. */
            const int mark = next_mark;
            next_mark += 64;
            const int ph = (mark >> 6) & 3;
            if (ph == 3) pending = fetch_chunk((mark >> 8) + 3);
            else if (ph == 0) {
                const int c = (mark >> 8) + 2;
                ring32[(c & 7) * 64 + lane] = pending;
                odd = (odd & ~(1u << (c & 7))) | (chunk_odd(c, pending) << (c & 7));
            }
        }

        /* ---- superblocks enter the band: read their query planes ---- */
        if (__builtin_expect(tau == next_entry, 0)) {
            if (tau == tf) {
#pragma unroll
                for (int x = 0; x < G; ++x) {
                    Pv[x] = 0xFFFFFFFFu; Mv[x] = 0;               /* cells below the band grow by +1 per row */
                    const int w = s * G + x;
                    qp[x] = w < g.NW ? brx_load_planes(planes, w) : BrxQPlanes{0u, 0u, 0u, 0u};
                }
            }
            do { ++e_s; } while (e_s < g.NS && (brx_jlast(g, e_s) < brx_jfirst(g, e_s) || tf_of(e_s) <= tau));
            next_entry = e_s < g.NS ? tf_of(e_s) : NEVER;
        }

        /* ---- U column updates, straight-line ---- */
        const uint32_t x0 = xn0, x1 = xn1;
        const bool act = (uint32_t)(tau - tf) <= tspan;
        const bool keep = brx_keep_trip(g.slope, keep_base, keep_lim, brx_jrep_trip(U, tau, s));     /* one test per trip: what brx_stored(g, s, brx_jrep(g, s, j)) says for its columns */
        bool rare = tau <= tau_pro;
        if (__builtin_expect(odd != 0u, 0)) {
            bool lr = false;
#pragma unroll
            for (int c = 0; c < U; ++c) lr |= (((c < 4 ? x0 : x1) >> (8 * (c & 3))) & 0xFFu) > 3u;
            rare = rare || __ballot(lr && act) != 0ull;
        }
        uint32_t P[G], M[G];
#pragma unroll
        for (int x = 0; x < G; ++x) { P[x] = Pv[x]; M[x] = Mv[x]; }
        if (__builtin_expect(rare, 0)) {
            /* rolled: symbols outside A/C/G/T, and the corner trips whose columns may lie left of column 1 */
#pragma unroll 1
            for (int c = 0; c < U; ++c) {
                const uint32_t nb = (uint32_t)brx_from_lane_above((int)carry);
                const int j = U * tau + c + 1 - s;
                const bool real = act && j >= 1;
                uint32_t hm = nb & 1u, hp = nb >> 31;
                const uint32_t ch = ((c < 4 ? x0 : x1) >> (8 * (c & 3))) & 0xFFu;
                BRX_GLOBAL uint64_t *dst = (BRX_GLOBAL uint64_t *)(row0 + (size_t)c * row_bytes + slot8);
#pragma unroll
                for (int x = 0; x < G; ++x) {
                    uint32_t Eq = brx_eq_acgt(qp[x], 0u - (ch & 1u), 0u - ((ch >> 1) & 1u));
                    if (ch == 4u) Eq = qp[x].n;
                    if (real && ch > 4u && s * G + x < g.NW) {    /* inline, rolled: a call would impose the callee's registers */
                        uint32_t mq = 0;
#pragma unroll 1
                        for (int rr = 0; rr < 32; ++rr) { const int qi = 32 * (s * G + x) + rr; if (qi < g.Q && Qs[qi] == ch) mq |= 1u << rr; }
                        Eq = mq;
                    }
                    const uint32_t Xv = Eq | M[x];
                    const uint32_t Eq2 = Eq | hm;
                    const uint32_t Xh = (((Eq2 & P[x]) + P[x]) ^ P[x]) | Eq2;
                    const uint32_t Ph = M[x] | ~(Xh | P[x]);
                    const uint32_t Mh = P[x] & Xh;
                    const uint32_t PhS = (Ph << 1) | hp;
                    const uint32_t MhS = (Mh << 1) | hm;
                    const uint32_t Pn = MhS | ~(Xv | PhS), Mn = PhS & Xv;
                    if (real) { P[x] = Pn; M[x] = Mn; }
                    if (real && keep) dst[x] = ((uint64_t)Ph << 32) | (uint64_t)Pn;
                    hp = Ph >> 31; hm = Mh >> 31;
                }
                carry = real ? ((hp << 31) | hm) : IDLE;
            }
        } else {
            uint32_t pvs[U][G], phs[U][G];
#pragma unroll
            for (int c = 0; c < U; ++c) {
                const uint32_t nb = (uint32_t)brx_from_lane_above((int)carry);
                const uint32_t w = c < 4 ? x0 : x1;
                const uint32_t k0 = brx_bit_mask(w, 8 * (c & 3)), k1 = brx_bit_mask(w, 8 * (c & 3) + 1);
                uint32_t hm = nb & 1u;
                uint32_t hpw = nb;                                 /* hp in bit 31 */
                uint32_t Ph = 0, Mh = 0;
#pragma unroll
                for (int x = 0; x < G; ++x) {
                    const uint32_t Eq = brx_eq_acgt(qp[x], k0, k1);
                    const uint32_t Xv = Eq | M[x];
                    const uint32_t Eq2 = Eq | hm;
                    const uint32_t Xh = (((Eq2 & P[x]) + P[x]) ^ P[x]) | Eq2;
                    Ph = M[x] | ~(Xh | P[x]);
                    Mh = P[x] & Xh;
                    const uint32_t PhS = __builtin_amdgcn_alignbit(Ph, hpw, 31);     /* Ph << 1 | hp */
                    const uint32_t MhS = (Mh << 1) | hm;
                    P[x] = MhS | ~(Xv | PhS);
                    M[x] = PhS & Xv;
                    pvs[c][x] = P[x]; phs[c][x] = Ph;
                    hpw = Ph; hm = Mh >> 31;
                }
                carry = act ? ((Ph & 0x80000000u) | (Mh >> 31)) : IDLE;
            }
            if (act && keep) {                                     /* uint2 {pv, Ph} per word and column: stores with scalar row bases */
#pragma unroll
                for (int c = 0; c < U; ++c) {
                    BRX_GLOBAL uint64_t *dst = (BRX_GLOBAL uint64_t *)(row0 + (size_t)c * row_bytes + slot8);
#pragma unroll
                    for (int x = 0; x < G; ++x) dst[x] = ((uint64_t)phs[c][x] << 32) | (uint64_t)pvs[c][x];
                }
            }
        }
#pragma unroll
        for (int x = 0; x < G; ++x) { Pv[x] = act ? P[x] : Pv[x]; Mv[x] = act ? M[x] : Mv[x]; }

        /* ---- superblocks leave the band: their lanes take superblock s + 64 ---- */
        if (__builtin_expect(tau == next_hop, 0)) {
            if (tau >= tl) {
                s += 64; slot += slot_step; if (slot >= g.WSp) slot -= g.WSp;
                window();
            }
            do { ++h_s; } while (h_s < g.NS && tl_of(h_s) <= tau);
            next_hop = h_s < g.NS ? tl_of(h_s) : NEVER;
        }
        ring_bytes(tau + 1, &xn0, &xn1);                           /* bytes of the next trip */
    }
}

/* MAXG: the widest band geometry (32-bit words per lane) compiled with register-resident state.
 * Register use grows with it (7 VGPRs per word), so kernels that only ever meet narrow bands are
 * instantiated with a small MAXG and run at a higher occupancy; geometries above MAXG take the
 * slow memory-resident path below (correct, rare). */
template <int MAXG, int MING = 1>
__device__ inline void brx_align_forward_any(const uint8_t *Qs, const uint8_t *Ts, const BrxGeom &g, uint2 *tb,
                                             uint32_t *prog = nullptr) {
    if (g.G > MAXG || g.G < MING) {
        uint32_t *peq = reinterpret_cast<uint32_t *>(tb + brx_tb_units(g));
        uint2 *st = tb + brx_tb_units(g) + ((uint64_t)5 * (uint64_t)g.NW * 4 + 7) / 8;
        brx_build_peq(Qs, g, peq);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        brx_align_forward_wide(Qs, Ts, g, tb, peq, st);
        return;
    }
    uint2 *planes = tb + brx_tb_units(g);             /* the table area of brx_peq_units: 2 units per word here, 2.5 reserved */
    brx_build_planes(Qs, g, planes);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);                    /* the table is read back by other lanes of this wave */
    if constexpr (MING <= 1) { if (g.G == 1) { brx_align_forward_u<BRX_U1, 1>(Qs, Ts, g, tb, planes); return; } }
    if constexpr (MAXG >= 2 && MING <= 2) { if (g.G == 2) { brx_align_forward_u<4, 2>(Qs, Ts, g, tb, planes); return; } }
    if constexpr (MAXG >= 4 && MING <= 4) { if (g.G == 4) { brx_align_forward_u<2, 4>(Qs, Ts, g, tb, planes); return; } }
    if constexpr (MAXG >= 8 && MING <= 8) { if (g.G == 8) { brx_align_forward<8>(Qs, Ts, g, tb, planes, prog); return; } }
    if constexpr (MAXG >= 16 && MING <= 16) { if (g.G == 16) { brx_align_forward<16>(Qs, Ts, g, tb, planes, prog); return; } }
}

/* Full alignment with a given band bound k.  Returns false if the band was too narrow.
 * Handles empty inputs.  All lanes of the wave must call; results are wave-uniform. */
template <int MAXG = 16, int MING = 1>
__device__ inline bool brx_wave_align(const uint8_t *Qs, int Q, const uint8_t *Ts, int T, int k,
                                      uint2 *tb, uint64_t tb_cap_units, uint8_t *ops_end,
                                      int *n_cols, int *n_match, bool *no_space, uint32_t *prog = nullptr,
                                      uint64_t *clk = nullptr, int hmul = 0) {
    const int lane = threadIdx.x & 63;
    *no_space = false;
    if (Q == 0 || T == 0) {
        if (ops_end) {
            for (int x = lane; x < Q; x += 64) ops_end[-x - 1] = BRX_OP_I;
            for (int x = lane; x < T; x += 64) ops_end[-x - 1] = BRX_OP_D;
        }
        *n_cols = Q + T; *n_match = 0;
        return true;
    }
    BrxGeom g = brx_make_geom(Q, T, k, hmul);
    if (g.G == 0 || brx_align_units(g) > tb_cap_units) { *no_space = true; *n_cols = 0; *n_match = 0; return false; }
    BRX_PROG(prog, 3, 1);
    BRX_PROG(prog, 6, (uint32_t)g.t_end);
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
    brx_align_forward_any<MAXG, MING>(Qs, Ts, g, tb, prog);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);      /* stores of this wave visible to its own later loads */
    BRX_PROG(prog, 3, 2);
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    bool ok = brx_align_traceback(Qs, Ts, g, tb, ops_end, n_cols, n_match, prog);
    if (clk) { const uint64_t c2 = __builtin_amdgcn_s_memtime(); clk[0] += c1 - c0; clk[1] += c2 - c1; }
    BRX_PROG(prog, 3, 3);
    if (ok && (*n_cols - *n_match) > k) ok = false;
    return ok;
}

#endif /* BRX_ALIGN_H */

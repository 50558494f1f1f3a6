/*
 * brx_mutate.h -- the mutate loop of sequence_fragment (/root/reference/badread/simulate.py:272-346)
 * as a multi-pass pipeline, included by brx_kernels.h.
 *
 * The reference re-estimates the identity every 25 applied changes by aligning a 1000-base window
 * of the fragment against its mutated version (simulate.py:325-346).  Doing that alignment with a
 * whole wavefront (k_mutate, kept as the fallback) occupies a wave for ~1000 dependent column steps
 * with 3-4 of its 64 lanes busy -- r01 profiles: 80 % of the mutate stage.  Here the loop is cut at
 * every alignment:
 *
 *   k_mutate_seg   1 wave = 1 read, run to completion: the loop with every window aligned IN PLACE by the
 *                  wave-systolic aligner (the head set of a batch and the in-place tail of its bulk set).  It
 *                  takes a read over from the bulk passes in any state (MS).  k-mers come from the read's
 *                  2-bit codes (F2); the error model is read through its lookup-order tables (include/brx.h:
 *                  d_rowx, d_altx).  The BULK passes themselves are brx_passes.h (round 6): k_mut_apply (one
 *                  read per lane), k_mut_post (park / propose ahead), k_pass_lists.
 *   k_win_lane     1 LANE = 1 parked window (64 windows per wave): banded block Myers over the window
 *                  with the band state, the query planes and a 32-column window of the target planes in
 *                  REGISTERS (the lanes are skewed so that every band moves in the same loop trip; no
 *                  LDS), 2-bit move codes to global memory in a [trip][slot][lane] layout, then a per-lane
 *                  canonical traceback on those codes, 16 columns per memory round trip.
 *   k_win_wave     1 wave = 1 parked window, for windows the lane kernel does not take (non-ACGT
 *                  symbols, very wide bands, very long targets): the wave-systolic aligner.
 *
 * The host repeats {k_mut_apply, k_mut_post, k_pass_lists, k_win_lane, k_win_wave} until few reads are left.  Results are
 * identical to the sequential loop: proposals are pure functions of (seed, read, iteration), the
 * alignment result is applied exactly where the inline alignment was, and both aligners produce
 * the canonical path (distance, columns and matches are all that is used here).
 */
#ifndef BRX_MUTATE_H
#define BRX_MUTATE_H

#ifndef BRX_SEG_WPS
#define BRX_SEG_WPS 4                                    /* register budget of the mutate kernels: waves per SIMD (below) */
#endif
#define BRX_WIN_Q 1024                                   /* slot bytes reserved for the window of F           */
#define BRX_WIN_BYTES 5120                               /* byte part of a slot: [0,1024) query, then target  */
#define BRX_WIN_TMAX (BRX_WIN_BYTES - BRX_WIN_Q - 16)    /* longer joined windows go to the legacy kernel     */
#define BRX_LANE_TMAX 1536                               /* lane kernel: target columns held in LDS planes    */
/* behind the bytes: the same pair as 2-bit planes (32-bit words: query lo[32] hi[32], target lo[48] hi[48]), written by
   the parking wave for windows bound for the lane kernel -- 64 symbols per ballot in the wave that has just written
   them, instead of 64 windows x 32 dependent loads in front of every lane-kernel wave (measured: 0.59 of its 2.1 ms) */
#define BRX_WIN_PLANES BRX_WIN_BYTES
#define BRX_WIN_PLANE_WORDS (2 * 32 + 2 * (BRX_LANE_TMAX / 32))
#define BRX_WIN_STRIDE (BRX_WIN_BYTES + 4 * BRX_WIN_PLANE_WORDS)   /* slot bytes per read */
#define BRX_LANE_W 8                                     /* lane kernel: band blocks alive in one column      */
#define BRX_LANE_TBC 16                                  /* lane kernel: traceback columns fetched per round  */
#define BRX_LANE_QW 32                                   /* lane kernel: query plane words (windows of up to 1024 rows) */
#define BRX_LANE_TB_UNITS ((uint64_t)(BRX_LANE_TMAX + 34) * BRX_LANE_W * 64)  /* uint2 units of move codes per wave: [trip][slot][lane] */

struct MS {                       /* loop state of a parked read */
    double errors, est;
    uint64_t round_loops;         /* loop_count at the start of the 64-proposal round being applied */
    uint32_t change, nalign;
    uint32_t phase;               /* 0 not started, 1 waiting for an alignment, 2 done, 3 needs k_mutate */
    uint32_t surv_lane, j_next;   /* proposal (lane of the round) and k-mer position to resume at     */
    uint32_t win_a, win_b, tl, cost;
    uint32_t res_ncols, res_nmatch;
    uint32_t status, passes;
    uint32_t win_kind;            /* bulk passes (brx_passes.h): which aligner the parked window goes to -- MC_EASY | band class << 8, MC_HARD, MC_LEGACY */
    uint32_t pad_;
};

enum { MC_QUEUE = 0, MC_OUT = 1, MC_EASY = 2, MC_HARD = 3, MC_LEGACY = 4, MC_WORDS = 8 };
/* Windows bound for the lane kernel are listed by band width: a wave of k_win_lane computes as many band slots per column
 * as its WIDEST window needs, and the widest of 64 windows taken as they come is near the top of the range (7-8 slots) while
 * most need 3-5.  Class c = band blocks - 3 (<= 3 blocks: class 0; 8 blocks: class 5). */
#define BRX_LANE_CLASSES 6
/* the class counters live a cache line apart: device-scope atomics on ONE line retire at 11.4 ns each whatever the number of waves
   (tools/native/atomic_bench.hip, profiles/r06_atomic_bench.jsonl) */
#define BRX_CLS_STRIDE 32
__device__ __forceinline__ uint32_t brx_lane_class(int band_blocks) {
    const int c = band_blocks - 3;
    return (uint32_t)(c < 0 ? 0 : c > BRX_LANE_CLASSES - 1 ? BRX_LANE_CLASSES - 1 : c);
}

/* Park a window in ONE pass: qb[0, b-a) = F[a:b] (+16 bytes 0xFF), tbuf[0, tl) = join(new_fragment_bases[a:b]) clipped
 * to `tmax` bytes (+16 bytes 0xFE when it fits), *cost = edit bound of the pair, *odd = a symbol outside ACGT on either
 * side (wave-uniform).  Returns the joined length tl (may exceed tmax: the caller hands the read to k_mutate).
 *
 * PLANES: the pair also as 2-bit planes behind the bytes (pl: query lo[32] hi[32], target lo[48] hi[48]) for the lane
 * kernel, in the same pass: the query planes are ballots on the fragment bytes the lanes hold anyway; the target bytes --
 * whose positions depend on the insertions and deletions before them -- go to a per-wave LDS window as well, and the target
 * planes are ballots on that.  (Round 2 read both byte strings back from global memory: two more chains of dependent round
 * trips in a wave that has nothing else to do -- parking was 92 of the ~300 kcycles of a mutate cycle.) */
#define BRX_PARK_LDS BRX_LANE_TMAX
/* LDS of a parking wave: the joined window on its way to the slot (bytes) and to bit planes */
#define BRX_PARK_LDS_WORDS ((BRX_WIN_BYTES - BRX_WIN_Q) / 4 + 4)
__shared__ uint32_t brx_park_words[BRX_PARK_LDS_WORDS];
#define brx_park_lds (reinterpret_cast<uint8_t *>(brx_park_words))
#define brx_park_lds_rows ((uint32_t)(BRX_PARK_LDS_WORDS / 4))
struct __attribute__((packed, aligned(1))) BrxB16 { uint32_t x, y, z, w; };      /* sixteen bytes behind any address */
static_assert(BRX_ALIGN_SIZE + 16 <= 1024 && BRX_WIN_Q >= 1024, "a window is 64 lanes x 16 positions");

/* bit j of the result = bit `bit` of byte j of v (j = 0..3) */
__device__ __forceinline__ uint32_t brx_byte_bits(uint32_t v, int bit) { return (((v >> bit) & 0x01010101u) * 0x01020408u) >> 24; }

/* Round 4: ONE round of loads instead of sixteen.  A lane takes SIXTEEN CONSECUTIVE positions of the window (1000 <= 64 x 16):
 * their fragment bytes (one 16-byte load) and replacement words (four), then everything is registers and LDS -- the
 * lengths are summed per lane and scanned once over the wave, a lane writes its stretch of the joined string to the LDS
 * window (a changed position looks its characters up in the pool), and the window leaves for the slot in 16-byte rows;
 * both strings' bit planes come from the same registers / LDS rows by multiplication (brx_byte_bits) and a pair of
 * neighbouring lanes per plane word.  (The loop over 64-position steps it replaces made ~45 dependent global round trips
 * per parked window -- fragment byte and replacement, pool characters, scan, stores, sixteen times over: 77 k of the ~290 k
 * cycles of a mutate cycle, profiles/r04b.)  Memory image of the slot: as before. */
template <bool PLANES>
__device__ inline uint32_t wave_park(const brx_error_model &em, const uint8_t *F, const uint32_t *repl, uint32_t a, uint32_t b,
                                     uint8_t *qb, uint8_t *tbuf, uint32_t tmax, uint32_t *cost, bool *odd, uint32_t *pl = nullptr) {
    const int lane = lane_id();
    const uint32_t ql = b - a;
    const uint32_t p0 = a + 16u * (uint32_t)lane;
    const uint32_t nv = p0 >= b ? 0u : (b - p0 < 16u ? b - p0 : 16u);            /* positions of this lane inside the window */
    uint32_t fw[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    uint32_t rw[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) rw[i] = 0u;
    if (nv) {
        const BrxB16 f = *reinterpret_cast<const BrxB16 *>(F + p0);              /* F holds 16 bytes behind the read */
        fw[0] = f.x; fw[1] = f.y; fw[2] = f.z; fw[3] = f.w;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const BrxU4 r4 = *reinterpret_cast<const BrxU4 *>(repl + p0 + 4u * (uint32_t)q);
            rw[4 * q] = r4.x; rw[4 * q + 1] = r4.y; rw[4 * q + 2] = r4.z; rw[4 * q + 3] = r4.w;
        }
    }
    /* positions behind the window: byte 0xFF (the query's terminator), no replacement, no length */
    uint32_t L = 0u;
    bool o_ = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const bool in = (uint32_t)i < nv;
        if (!in) { rw[i] = 0u; fw[i >> 2] |= 0xFFu << (8 * (i & 3)); }
        else { L += rep_len(rw[i]); o_ |= ((fw[i >> 2] >> (8 * (i & 3))) & 0xFCu) != 0u; }
    }
    const uint32_t inc = wave_incl_scan(L);
    const uint32_t run = wave_bcast_u32(inc, 63);
    /* ---- query: the bytes in one 16-byte row per lane (rows up to 16 bytes behind the window carry the 0xFF terminator) ---- */
    if (16u * (uint32_t)lane < ql + 16u) {
        BrxU4 q4; q4.x = fw[0]; q4.y = fw[1]; q4.z = fw[2]; q4.w = fw[3];
        *reinterpret_cast<BrxU4 *>(qb + 16u * (uint32_t)lane) = q4;
    }
    if constexpr (PLANES) {
        uint32_t lo = 0u, hi = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t nvq = nv > 4u * (uint32_t)q ? (nv - 4u * (uint32_t)q >= 4u ? 0xFFFFFFFFu : (1u << (8u * (nv - 4u * (uint32_t)q))) - 1u) : 0u;
            const uint32_t v = fw[q] & nvq;
            lo |= brx_byte_bits(v, 0) << (4 * q); hi |= brx_byte_bits(v, 1) << (4 * q);
        }
        const uint32_t lo1 = (uint32_t)__shfl_down((int)lo, 1, 64), hi1 = (uint32_t)__shfl_down((int)hi, 1, 64);
        /* words 0 .. 2 ceil(ql / 64) - 1, as the 64-position steps wrote them */
        if (!(lane & 1) && 16u * (uint32_t)lane < ((ql + 63u) & ~63u)) { pl[lane >> 1] = lo | (lo1 << 16); pl[32 + (lane >> 1)] = hi | (hi1 << 16); }
    }
    /* ---- target: this lane's stretch of the joined string into the LDS window ---- */
    uint32_t o = inc - L, c = 0u;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if ((uint32_t)i < nv) {
            const uint32_t w = rw[i];
            const uint8_t fb = (uint8_t)(fw[i >> 2] >> (8 * (i & 3)));
            if (!w) { if (o < tmax) brx_park_lds[o] = fb; o += 1u; }
            else {
                const uint32_t len = (w >> 24) & 0x7Fu;
                bool has = false;
                for (uint32_t x = 0; x < len; ++x) {
                    const uint8_t ch = rep_char(em, w, x);
                    o_ |= ch > 3; has |= ch == fb;
                    if (o + x < tmax) brx_park_lds[o + x] = ch;
                }
                c += len < 2u ? 1u : len - (has ? 1u : 0u);                        /* rep_cost */
                o += len;
            }
        }
    }
    if (run <= tmax && lane < 16) brx_park_lds[run + (uint32_t)lane] = 0xFE;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);                           /* the LDS bytes of every lane are in place */
    const uint32_t nbytes = run <= tmax ? run + 16u : tmax;
    for (uint32_t g = (uint32_t)lane; 16u * g < nbytes; g += 64u)
        *reinterpret_cast<BrxU4 *>(tbuf + 16u * g) = *reinterpret_cast<const BrxU4 *>(brx_park_lds + 16u * g);
    if constexpr (PLANES) {
        const uint32_t tl = run < BRX_PARK_LDS ? run : BRX_PARK_LDS;
        for (uint32_t g0 = 0; 16u * g0 < ((tl + 63u) & ~63u); g0 += 64u) {
            const uint32_t g = g0 + (uint32_t)lane;
            const BrxU4 t4 = *reinterpret_cast<const BrxU4 *>(brx_park_lds + 16u * (g < brx_park_lds_rows ? g : 0u));
            const uint32_t tv[4] = {t4.x, t4.y, t4.z, t4.w};
            uint32_t lo = 0u, hi = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t at = 16u * g + 4u * (uint32_t)q;
                const uint32_t live = at >= tl ? 0u : (tl - at >= 4u ? 0xFFFFFFFFu : (1u << (8u * (tl - at))) - 1u);
                const uint32_t v = tv[q] & live;
                lo |= brx_byte_bits(v, 0) << (4 * q); hi |= brx_byte_bits(v, 1) << (4 * q);
            }
            const uint32_t lo1 = (uint32_t)__shfl_down((int)lo, 1, 64), hi1 = (uint32_t)__shfl_down((int)hi, 1, 64);
            if (!(lane & 1) && 16u * g < ((tl + 63u) & ~63u)) { pl[64 + (g >> 1)] = lo | (lo1 << 16); pl[64 + BRX_LANE_TMAX / 32 + (g >> 1)] = hi | (hi1 << 16); }
        }
    }
    *cost = wave_sum(c);
    *odd = __ballot(o_) != 0ull;
    return run;
}

/* -------------------------------------------------------------------------------------------------
 * k_mutate_seg
 * ----------------------------------------------------------------------------------------------- */
/* Every read is run to completion, its windows aligned in place with the wave-systolic aligner: the head set of a batch (its
 * longest chains, from the start) and the last reads of the bulk set, where a host round trip per alignment would cost more than
 * the alignment.  (Until round 5 the same template, INLINE = false, was the bulk passes' kernel -- a wave per read that parked
 * at every alignment, with the read staged in a 10 KB LDS slice; brx_passes.h replaced it.) */
/* PROFILE = true (BRX_PROFILE=1): shader-clock time of every phase of the loop is added to phase[8 r + i]:
 *   0 propose (draws, k-mer bytes, table lookups)   1 apply survivors   2 park (window join + copies, state)
*   3 in-place alignment   4 everything else   5 / 6 forward / traceback part of 3
 * The phase clock is wave-uniform scalar code; the default instantiations do not contain it. */
#define BRX_PHASE(next)                                                                                          \
    do { if constexpr (PROFILE) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); const uint64_t dt_ = now_ - ph_last;  \
             ph_last = now_; ph0 += ph_cur == 0 ? dt_ : 0; ph1 += ph_cur == 1 ? dt_ : 0; ph2 += ph_cur == 2 ? dt_ : 0;   \
             ph3 += ph_cur == 3 ? dt_ : 0; ph4 += ph_cur == 4 ? dt_ : 0; ph_cur = (next); } } while (0)

/* WPS = waves per SIMD the register budget is set for.  Measured (round 4, configs[3], six batches in flight): the run-to-
 * completion instantiation carries the wave aligner's registers beside the loop state and keeps 160 B per lane in scratch memory
 * at four waves per SIMD (128 VGPRs), nothing at two (214) -- and the un-spilled build is SLOWER (4.30-4.34 against 4.46 Gbases/s,
 * again 4.70-4.72 against 4.76 on the next tree): the head chain is one wave per SIMD, but its registers are taken from the
 * other five batches' kernels on the same SIMDs.  Four it is. */
/* What only parking, the in-place alignment and the epilogue use -- once per alignment cycle -- lives in device memory and is read
 * where it is used (a volatile load: not hoisted), not in kernel arguments: the kernel runs at its SGPR limit (106), a fifth of its
 * instructions were v_readlane / v_writelane spill traffic, and every argument is two SGPRs that stay live over the whole loop. */
struct MutAux {
    uint32_t *req_easy, *req_hard, *req_legacy, *legacy_ctr;
    uint8_t *winbuf;
    uint64_t *clk;
    uint8_t *scr_base;
    uint64_t scr_bytes;
    uint32_t *flags;
    uint64_t *phase;
};
template <typename T>
__device__ __forceinline__ T brx_cold(T const *field) {
    static_assert(sizeof(T) == 8, "pointers and 64-bit sizes");
    const uint64_t v = *reinterpret_cast<const volatile uint64_t *>(field);
    const uint64_t u = uni(v);
    T out; __builtin_memcpy(&out, &u, 8);
    return out;
}

template <bool PROFILE = false, int WPS = 4>
__global__ void __launch_bounds__(64, WPS) k_mutate_seg(BrxDev d, RS *rs, MS *msv, const uint32_t *active_in,
                                                    const uint32_t *n_in_ptr, uint32_t *ctr,
                                                    const MutAux *aux, const uint8_t *Fbuf, uint32_t *repl,
                                                    const uint32_t *F2buf, const uint32_t *Cbuf) {
    const int lane = lane_id();
    const brx_error_model &em = d.em;
    const int k = em.k;
    const uint32_t wave_index = blockIdx.x;                                               /* scratch slot of this wave */
    const uint32_t n_in = uni(*n_in_ptr);
    uint64_t ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, ph4 = 0, ph_last = 0, pclk[2] = {0, 0};
    int ph_cur = 4;
    for (;;) {
        const uint32_t qi = wave_pop(&ctr[MC_QUEUE]);
        if (qi >= n_in) break;
        const uint32_t r = active_in[qi];
        const RS s = rs[r];
        if (s.n == 0) continue;
        const uint64_t t_begin = __builtin_amdgcn_s_memtime();
        if constexpr (PROFILE) { ph0 = ph1 = ph2 = ph3 = ph4 = 0; pclk[0] = pclk[1] = 0; ph_last = t_begin; ph_cur = 4; }
        MS ms = msv[r];
        const uint64_t read = d.first_read + r;
        const uint32_t n = s.n;
        const uint8_t *F = Fbuf + s.F_off;
        uint32_t *rp = repl + s.F_off;
        /* ---- the read as 2-bit codes (k_build): k-mers of the proposal rounds come from these (3.75 KB per 15 kb read: L2
           resident); the changed test is on repl[].  A read that holds a symbol outside ACGT keeps to the bytes for those k-mers. ---- */
        const uint32_t *f2g = F2buf + (s.F_off >> 4);
        const uint32_t nw2 = (n + 15u) >> 4, nwc = (n + 31u) >> 5;
        const bool coded = uni(f2g[nw2] == 0u);
        const uint32_t *oddg = Cbuf + (s.F_off >> 4) + nwc;              /* a bit per base: a symbol outside ACGT (k_build) */
        const double target = s.target;
        const double dn = (double)n;
        const uint64_t max_i = (uint64_t)n - 1 - (uint64_t)k;
        const double need = dn * (1.0 - target);
        const uint64_t loop_cap = 100ull * (uint64_t)n;

        double errors = 0.0;
        uint64_t loops = 0;
        uint32_t change = 0, nalign = 0;
        uint32_t st_extra = ms.status;
        bool parked = false;
      for (;;) {                                   /* one trip per alignment of this read */
        errors = 0.0; loops = 0; change = 0; nalign = 0;
        /* phase 4 (brx_passes.h, MP_HUNGRY): a read taken over from the bulk passes between two survivors, no alignment pending --
           a resumed round at iteration round_loops (surv_lane = j_next = 0) whose errors are NOT blended */
        const bool hungry = ms.phase == 4u;
        bool resume = ms.phase == 1u || hungry;
        st_extra = ms.status;
        parked = false;
        if (resume) {
            errors = ms.errors; loops = ms.round_loops; change = ms.change; nalign = ms.nalign;
            if (!hungry) {
                const double id = ms.res_ncols ? (double)ms.res_nmatch / (double)ms.res_ncols : 0.0; /* misc.py:228-240 */
                if (n <= BRX_ALIGN_SIZE) errors = (1.0 - id) * dn;                                   /* simulate.py:333 */
                else {
                    const double est_err = (1.0 - id) * dn;
                    const double weight = (double)BRX_ALIGN_SIZE / dn;
                    errors = est_err * weight + errors * (1.0 - weight);                             /* simulate.py:344-346 */
                }
            }
        }
        bool done = !resume && need < 0.5;
        while (!done) {
            double est;
            if (resume) est = ms.est;
            else {
                if (loops + 1 > loop_cap) { loops += 1; break; }
                est = 1.0 - errors / dn;
                if ((double)change > 0.9 * dn || est <= target) { loops += 1; break; }
            }
            const uint64_t room = loop_cap - loops;
            const uint32_t B = room < 64 ? (uint32_t)room : 64u;
            /* ---- propose (identical draws on a resumed round) ---- */
            BRX_PHASE(0);
            BrxProp pr; pr.x = pr.y = pr.z = 0u;
            uint64_t ipos = 0;
            if ((uint32_t)lane < B) {
                uint32_t w[4];
                brx_draw4(d.seed, read, BRX_ST_MUT, loops + (uint64_t)lane, w);
                ipos = brx_mulhi64(((uint64_t)w[1] << 32) | w[0], max_i + 1);
                const uint32_t wi = (uint32_t)(ipos >> 4), sh = 2u * ((uint32_t)ipos & 15u);
                const uint32_t w0 = f2g[wi], w1 = f2g[wi + 1u];
                const uint32_t row = (uint32_t)(((((uint64_t)w0 << 32) | (uint64_t)w1) << sh) >> (64 - 2 * k));
                bool bad = false;                   /* a symbol outside ACGT in the k-mer: error_model.py:142-143 */
                if (!coded) {
                    const uint32_t p_lo = (uint32_t)ipos;             /* two words of the map: the one behind the last is zero */
                    const uint64_t both = (((uint64_t)oddg[(p_lo >> 5) + 1u] << 32) | (uint64_t)oddg[p_lo >> 5]) >> (p_lo & 31u);
                    bad = (both & ((1ull << k) - 1ull)) != 0ull;
                }
                if (bad) {
                    const BrxRc c = dev_random_change_split(w[3], k);
                    pr.x = dev_random_change_word(c, (uint32_t)F[ipos + c.pos]); pr.y = BRX_PROP_RANDOM | c.pos;
                } else pr = dev_propose_row(em, row, w[2], w[3]);
            }
            unsigned long long surv = __ballot(pr.y != 0u);
            BRX_PHASE(1);
            int j0 = 0;
            if (resume) { surv &= ~((1ull << ms.surv_lane) - 1ull); j0 = (int)ms.j_next; }
            bool first = resume;
            bool fresh = !resume;                  /* est is 1 - errors / dn of the CURRENT errors (a resumed round starts with the parked estimate and blended errors) */
            resume = false;
            /* ---- apply survivors in iteration order ---- */
            while (surv) {
                const int l = __ffsll((long long)surv) - 1;
                surv &= surv - 1;
                const uint64_t i0 = wave_bcast_u64(ipos, l);
                /* lane j < k takes position j of the k-mer: its replacement word, derived from the proposing lane's three
                   words, and the current state of that position with ONE load for all positions, then the untouched,
                   changed positions (simulate.py:309) are applied in order */
                const uint32_t wj = brx_prop_word(em, wave_bcast_u32(pr.x, l), wave_bcast_u32(pr.y, l), wave_bcast_u32(pr.z, l));
                uint32_t curj = 1u;
                if (lane < k) curj = rp[i0 + (uint64_t)lane];
                unsigned long long todo = __ballot(lane < k && wj != 0u && curj == 0u);
                if (first) todo &= ~((1ull << j0) - 1ull);
                const bool applies = todo != 0ull;
                /* a survivor whose changed positions are all taken already leaves errors, change and est as they are: the
                   square root, the division and the tests below would repeat the previous survivor's, bit for bit */
                const double scale = applies ? est * brx_sqrt(est) : 0.0;
                while (todo) {
                    const int j = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const uint32_t w = wave_bcast_u32(wj, j);
                    if (lane == j) rp[i0 + (uint64_t)j] = w;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      /* the next survivor tests repl[] */
                    change += 1;
                    const uint32_t len = (w >> 24) & 0x7Fu;
                    errors += (double)(len < 2 ? 1u : len - 1u) * scale;
                    if (change % BRX_ALIGN_INTERVAL == 0) {
                        /* ---- park the read: the window pair goes to its slot, the loop state to MS ---- */
                        BRX_PHASE(2);
                        uint32_t a = 0, b = n;
                        if (n > BRX_ALIGN_SIZE) {
                            uint32_t ww[4];
                            brx_draw4(d.seed, read, BRX_ST_WIN, (uint64_t)nalign, ww);
                            a = (uint32_t)brx_mulhi64(((uint64_t)ww[1] << 32) | ww[0], (uint64_t)n - BRX_ALIGN_SIZE + 1);
                            b = a + BRX_ALIGN_SIZE;
                        }
                        nalign += 1;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      /* repl[] as the other lanes left it */
                        __builtin_amdgcn_s_waitcnt(0);
                        /* one pass over the window: F[a:b] -> query slot, join(new[a:b]) -> target slot (clipped to the
                           slot; an overflowing window goes to the whole-read kernel), edit bound, non-ACGT flag */
                        uint32_t cost = 0;
                        uint8_t *qb = brx_cold(&aux->winbuf) + (uint64_t)r * BRX_WIN_STRIDE, *tbuf = qb + BRX_WIN_Q;
                        bool odd = false;
                        const uint32_t tl = wave_park<false>(em, F, rp, a, b, qb, tbuf, BRX_WIN_TMAX, &cost, &odd);
                        const uint32_t ql = b - a;
                        const bool fits = tl <= BRX_WIN_TMAX;       /* else: the whole-read kernel (k_mutate) starts the read over */
                        if (fits) {                                 /* the in-place aligner below reads the bytes back */
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            __builtin_amdgcn_s_waitcnt(0);
                        }
                        {
                            MS o = ms;
                            o.errors = errors; o.est = est; o.round_loops = loops; o.change = change; o.nalign = nalign;
                            o.phase = fits ? 1u : 3u;
                            o.surv_lane = (uint32_t)l; o.j_next = (uint32_t)(j + 1);
                            o.win_a = a; o.win_b = b; o.tl = tl; o.cost = cost; o.res_ncols = 0; o.res_nmatch = 0;
                            o.passes = ms.passes + 1; o.status = st_extra;
                            if (fits) ms = o;                       /* stays in registers: aligned below */
                            else {
                                uint32_t *c_legacy = brx_cold(&aux->req_legacy), *c_legacy_ctr = brx_cold(&aux->legacy_ctr);
                                if (lane == 0) { msv[r] = o; c_legacy[atomicAdd(c_legacy_ctr, 1u)] = r; }
                                ms.phase = 3u;
                            }
                        }
                        parked = true;
                        break;
                    }
                }
                if (parked) break;
                first = false;
                /* top-of-loop tests of the iteration that follows this survivor */
                if (applies || !fresh) {
                    const double est2 = 1.0 - errors / dn;
                    if ((double)change > 0.9 * dn || est2 <= target) { loops += (uint64_t)l + 2; done = true; break; }
                    est = est2;
                    fresh = true;
                }
            }
            if (parked || done) break;
            loops += B;
            if (B < 64) { loops += 1; break; }
        }
        BRX_PHASE(4);
        if (parked && ms.phase == 1u) {
            /* align the parked window here, at the top level where only MS is live, and resume the same read */
            const uint8_t *qb = brx_cold(&aux->winbuf) + (uint64_t)r * BRX_WIN_STRIDE, *tbuf = qb + BRX_WIN_Q;
            const uint64_t scr_bytes = brx_cold(&aux->scr_bytes);
            uint2 *tb = reinterpret_cast<uint2 *>(brx_cold(&aux->scr_base) + (uint64_t)wave_index * scr_bytes);
            int ncols = 0, nmatch = 0; bool nospace = false;
            BRX_PHASE(3);
            const bool ok = brx_wave_align<1, 1>(qb, (int)(ms.win_b - ms.win_a), tbuf, (int)ms.tl, (int)ms.cost, tb, scr_bytes / 8, nullptr,
                                              &ncols, &nmatch, &nospace, nullptr, PROFILE ? pclk : nullptr);
            BRX_PHASE(4);
            ms.res_ncols = (uint32_t)ncols; ms.res_nmatch = (uint32_t)nmatch;
            if (!ok && !nospace) ms.status |= BRX_RS_BAND;
            if (nospace) {
                uint32_t *flags = brx_cold(&aux->flags);
                if (lane == 0) { atomicOr(&flags[0], 1u); flags[8] = r; flags[9] = ms.win_b - ms.win_a; flags[10] = ms.tl; flags[11] = ms.cost; }
            }
            continue;
        }
        break;
      }
        uint64_t *ck = brx_cold(&aux->clk) + (uint64_t)r * 8;
        if constexpr (PROFILE) {
            BRX_PHASE(4);
            uint64_t *pp = brx_cold(&aux->phase) + (uint64_t)r * 8;
            if (lane == 0) {
                pp[0] += ph0; pp[1] += ph1; pp[2] += ph2; pp[3] += ph3; pp[4] += ph4; pp[5] += pclk[0]; pp[6] += pclk[1];
            }
        }
        if (parked) {
            if (lane == 0) ck[0] += __builtin_amdgcn_s_memtime() - t_begin;
            continue;
        }
        /* epilogue: lengths of the mutated read, trims (simulate.py:348-349), proven distance bound */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        uint32_t cost = 0;
        const uint32_t m = wave_join(em, F, rp, 0, n, nullptr, &cost);
        uint32_t st = 0, et = 0;
        if (lane < k) { st = rep_len(rp[lane]); et = rep_len(rp[n - k + lane]); }
        st = wave_sum(st); et = wave_sum(et);
        if (lane == 0) {
            RS *o = &rs[r];
            o->status = s.status | st_extra; o->m = m; o->ub = cost; o->start_trim = st; o->end_trim = et;
            o->loops = (uint32_t)loops; o->changes = change; o->naligns = nalign;
            o->units = 0;                                          /* sized by k_fin_join */
            msv[r].phase = 2u;
            ck[0] += __builtin_amdgcn_s_memtime() - t_begin; ck[1] = nalign;
        }
    }
}

/* -------------------------------------------------------------------------------------------------
 * k_win_wave: one parked window per wave (the windows k_win_lane does not take)
 * ----------------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(64, 5) k_win_wave(MS *msv, const uint32_t *req, const uint32_t *n_req_ptr, uint32_t *queue,
                                                  const uint8_t *winbuf, uint8_t *scr_base, uint64_t scr_bytes, uint32_t *flags,
                                                  uint32_t *next_ctr, uint32_t *next_cls) {
    const int lane = lane_id();
    const uint32_t n_req = uni(*n_req_ptr);
    /* the counter block of the NEXT pass: last read (as the input count) by this pass's k_mutate_seg, which is done */
    if (next_ctr && blockIdx.x == 0 && lane < (int)MC_WORDS) { next_ctr[lane] = 0u; next_cls[lane * BRX_CLS_STRIDE] = 0u; }
    uint2 *tb = reinterpret_cast<uint2 *>(scr_base + (uint64_t)blockIdx.x * scr_bytes);
    for (;;) {
        const uint32_t qi = wave_pop(queue);
        if (qi >= n_req) break;
        const uint32_t r = req[qi];
        const MS ms = msv[r];
        const uint8_t *qb = winbuf + (uint64_t)r * BRX_WIN_STRIDE, *tbuf = qb + BRX_WIN_Q;
        int ncols = 0, nmatch = 0; bool nospace = false;
        const bool ok = brx_wave_align<1>(qb, (int)(ms.win_b - ms.win_a), tbuf, (int)ms.tl, (int)ms.cost, tb, scr_bytes / 8,
                                          nullptr, &ncols, &nmatch, &nospace);
        if (lane == 0) {
            msv[r].res_ncols = (uint32_t)ncols; msv[r].res_nmatch = (uint32_t)nmatch;
            if (!ok && !nospace) msv[r].status = ms.status | BRX_RS_BAND;
            if (nospace) { atomicOr(&flags[0], 1u); flags[8] = r; flags[9] = ms.win_b - ms.win_a; flags[10] = ms.tl; flags[11] = ms.cost; }
        }
    }
}

/* -------------------------------------------------------------------------------------------------
 * k_win_lane: one parked window per LANE
 * ----------------------------------------------------------------------------------------------- */
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, dd, 64); v = o > v ? o : v; }
    return v;
}

/* -----------------------------------------------------------------------------------------------------------------
 * brx_lanes_align: up to 64 window alignments, one per LANE, band state in registers
 * -----------------------------------------------------------------------------------------------------------------
 * Same band (brx_make_geom), same cell recurrence and the same canonical traceback (up / 'I', left / 'D', diagonal) as
 * brx_wave_align; only the distance columns and matches of the path are produced (all the mutate loop uses).
 *
 * A lane holds BRX_LANE_W consecutive 32-row blocks of its window: slot x = block s_lo + x, s_lo = the first block of
 * the band.  The band moves down one block every 32 columns, at a column that depends on the lane's geometry; the
 * lanes are therefore skewed against each other: in loop trip jj a lane works on ITS column j = jj - off, with off
 * chosen so that every lane's band moves exactly in the trips jj = 0 (mod 32) -- the register shift is one uniform
 * block of code instead of a dynamically indexed register file (round 2 kept the band in LDS for that reason: 45 KB per
 * wave, three waves per CU, and ~60 of its ~200 instructions per column were LDS addressing and traffic).  Query planes enter a lane's
 * registers one block per shift, the target planes as a 32-column window per shift (two funnel shifts), straight from
 * the parked planes in global memory: no LDS at all.
 *
 * What the forward pass stores per cell is the MOVE of the canonical traceback in two bits -- up = 10, left = 01,
 * diagonal on equal symbols = 00, diagonal on different symbols = 11 -- instead of {Pv, Ph}: the walk then counts
 * matches without looking at the sequences again.  Layout [trip][slot][lane]: a store instruction writes 512
 * contiguous bytes at a wave-uniform base.
 */
__device__ __forceinline__ uint32_t brx_bfe_mask(uint32_t v, int b) { return (uint32_t)((int32_t)(v << (31 - b)) >> 31); }

/* The forward pass with WB band slots per column, WB a compile-time constant: the slots of a column are one straight-line block,
   so the independent parts of neighbouring slots (equality masks, stores, the move codes) overlap the carry chain that links them
   -- a lone wave issues a DEPENDENT instruction every ~8 cycles and an independent one every ~2.  (As one loop with `if (x >= Wb)
   break` -- rounds 2-5 -- every slot was a basic block of its own.) */
template <int TW, int WB>
__device__ __forceinline__ void brx_lanes_forward(const bool valid, const uint32_t *__restrict__ pl, const BrxGeom &g, const int NS, const int T,
                                                  const int off, const int qb, const int JJ, uint2 *__restrict__ tbw) {
    const int lane = lane_id();
    uint32_t P[WB], M[WB], QL[WB], QH[WB];
#pragma unroll
    for (int x = 0; x < WB; ++x) {
        P[x] = 0xFFFFFFFFu; M[x] = 0u;                      /* cells below the band grow by +1 per row */
        const bool in = valid && x < NS;
        QL[x] = in ? pl[x] : 0u; QH[x] = in ? pl[BRX_LANE_QW + x] : 0u;
    }
    int slo = 0;                                            /* block held in slot 0 */
    uint32_t TLw = 0u, THw = 0u;                            /* target planes of columns j0 .. j0 + 31, j0 = (jj & ~31) - off */
    const uint32_t *tlo = pl + 2 * BRX_LANE_QW, *thi = tlo + TW;
    for (int jj = 0; jj <= JJ; ++jj) {
        if ((jj & 31) == 0) {
            /* ---- the band moves down one block (lanes whose band still starts at block 0 stay) ---- */
            const int bq = (jj >> 5) + qb;
            if (valid && bq >= 1) {
#pragma unroll
                for (int x = 0; x + 1 < WB; ++x) { P[x] = P[x + 1]; M[x] = M[x + 1]; QL[x] = QL[x + 1]; QH[x] = QH[x + 1]; }
                const int nb = bq + WB - 1;
                P[WB - 1] = 0xFFFFFFFFu; M[WB - 1] = 0u;
                QL[WB - 1] = nb < NS ? pl[nb] : 0u; QH[WB - 1] = nb < NS ? pl[BRX_LANE_QW + nb] : 0u;
                slo = bq;
            }
            /* ---- target planes of the next 32 trips: bit t = target index (jj - off - 1) + t ---- */
            const int t0 = jj - off - 1;
            const int w0 = t0 >> 5, sh = t0 & 31;
            const bool in0 = valid && w0 >= 0 && w0 < TW, in1 = valid && w0 + 1 >= 0 && w0 + 1 < TW;
            const uint32_t l0 = in0 ? tlo[w0] : 0u, l1 = in1 ? tlo[w0 + 1] : 0u;
            const uint32_t h0 = in0 ? thi[w0] : 0u, h1 = in1 ? thi[w0 + 1] : 0u;
            TLw = __builtin_amdgcn_alignbit(l1, l0, (uint32_t)sh);
            THw = __builtin_amdgcn_alignbit(h1, h0, (uint32_t)sh);
        }
        const int j = jj - off;
        const bool act = valid && j >= 1 && j <= T;
        int hi = (j + g.dhi - 1) >> 5;                      /* last block of the band in column j ... */
        if (hi > NS - 1) hi = NS - 1;
        hi = act ? hi - slo : -1;                           /* ... as a slot; slots 0 .. hi are computed */
        const int b = jj & 31;
        const uint32_t m0 = brx_bfe_mask(TLw, b), m1 = brx_bfe_mask(THw, b);
        uint32_t hp = 1u, hm = 0u;                          /* above the band (and above row 1): +1 per column */
        uint2 *dst = tbw + ((uint64_t)jj * (uint64_t)WB) * 64u + (uint32_t)lane;
#pragma unroll
        for (int x = 0; x < WB; ++x) {
            const uint32_t pv0 = P[x], mv0 = M[x];
            const uint32_t Eq = ~((QL[x] ^ m0) | (QH[x] ^ m1));
            const uint32_t Xv = Eq | mv0;
            const uint32_t Eq2 = Eq | hm;
            const uint32_t Xh = (((Eq2 & pv0) + pv0) ^ pv0) | Eq2;
            const uint32_t Ph = mv0 | ~(Xh | pv0);
            const uint32_t Mh = pv0 & Xh;
            const uint32_t PhS = (Ph << 1) | hp;
            const uint32_t MhS = (Mh << 1) | hm;
            const uint32_t pv = MhS | ~(Xv | PhS);
            const uint32_t mv = PhS & Xv;
            const bool on = x <= hi;
            P[x] = on ? pv : pv0;
            M[x] = on ? mv : mv0;
            if (on) {
                const uint32_t dX = ~(pv | Ph | Eq);        /* diagonal move on different symbols */
                dst[(uint32_t)x * 64u] = make_uint2(pv | dX, (Ph & ~pv) | dX);
            }
            hp = Ph >> 31; hm = Mh >> 31;                   /* the computed slots are 0 .. hi: every carry that is used was computed */
        }
    }
}

template <int TW>
__device__ inline void brx_lanes_align(const bool valid, const uint32_t *__restrict__ pl, const int Q, const int T, const int kb,
                                       uint2 *__restrict__ tbw, uint32_t *out_ncols, uint32_t *out_nmatch, bool *out_ok) {
    constexpr int W = BRX_LANE_W;
    const int lane = lane_id();
    const BrxGeom g = brx_make_geom(Q > 0 ? Q : 1, T > 0 ? T : 1, kb);
    const int NS = (Q + 31) >> 5;
    int Wb = (int)wave_max_u32(valid ? (uint32_t)((g.dhi - g.dlo) / 32 + 2) : 0u);      /* slots in use: the widest band of the wave */
    Wb = Wb < 3 ? 3 : Wb;                                   /* (the narrowest instantiation; the store's row width follows) */
    const int off = (g.dlo - 1) & 31;                       /* jj = j + off; (j + dlo - 1) >> 5 = (jj >> 5) + qb */
    const int qb = (g.dlo - 1 - off) >> 5;                  /* exact: dlo - 1 - off is a multiple of 32; negative */
    const int JJ = (int)wave_max_u32(valid ? (uint32_t)(T + off) : 0u);
    static_assert(W == 8, "one instantiation of the forward pass per band width in use");
    switch (Wb) {
        case 3: brx_lanes_forward<TW, 3>(valid, pl, g, NS, T, off, qb, JJ, tbw); break;
        case 4: brx_lanes_forward<TW, 4>(valid, pl, g, NS, T, off, qb, JJ, tbw); break;
        case 5: brx_lanes_forward<TW, 5>(valid, pl, g, NS, T, off, qb, JJ, tbw); break;
        case 6: brx_lanes_forward<TW, 6>(valid, pl, g, NS, T, off, qb, JJ, tbw); break;
        case 7: brx_lanes_forward<TW, 7>(valid, pl, g, NS, T, off, qb, JJ, tbw); break;
        default: brx_lanes_forward<TW, 8>(valid, pl, g, NS, T, off, qb, JJ, tbw); break;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);                          /* this wave's stores are visible to its loads below */

    /* ---- traceback, canonical (up, left, diagonal), BRX_LANE_TBC columns fetched per round trip ----
       A lane owns its window, so the walk is bit arithmetic on the two code words it holds per column (block s0 of the
       row it starts the round in, and s0 - 1): the run of up moves in a column is the run of 'up' codes below the current
       row (one count-leading-zeros), the code of the row it stops in says left, match or mismatch. */
    int i = Q, j = T;
    uint32_t ncols = 0, nmatch = 0;
    bool ok = valid;
    bool go = valid && i > 0 && j > 0;
    while (__ballot(go) != 0ull) {
        const int s0 = go ? ((i - 1) >> 5) : 0;
        const int jst = j;
        uint2 A[BRX_LANE_TBC], Bv[BRX_LANE_TBC];
#pragma unroll
        for (int x = 0; x < BRX_LANE_TBC; ++x) {
            const int col = jst - x;
            A[x] = make_uint2(0u, 0u); Bv[x] = make_uint2(0u, 0u);
            if (go && col >= 1) {
                int sl = (col + g.dlo - 1) >> 5; if (sl < 0) sl = 0;
                const int xa = s0 - sl;
                const uint64_t rowb = (uint64_t)(col + off) * (uint64_t)Wb;
                if (xa >= 0 && xa < Wb) A[x] = tbw[(rowb + (uint32_t)xa) * 64u + (uint32_t)lane];
                if (xa >= 1 && xa - 1 < Wb) Bv[x] = tbw[(rowb + (uint32_t)(xa - 1)) * 64u + (uint32_t)lane];
            }
        }
        bool walk = go;
#pragma unroll
        for (int x = 0; x < BRX_LANE_TBC; ++x) {
            bool done = !(walk && i > 0 && j > 0);
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                if (!done) {
                    const int sb = (i - 1) >> 5;
                    if (sb != s0 && sb != s0 - 1) { walk = false; done = true; }
                    else {
                        const int jf = 32 * sb - g.dhi + 1 < 1 ? 1 : 32 * sb - g.dhi + 1;
                        long long jl = 32ll * (sb + 1) - g.dlo; if (jl > T) jl = T;
                        if (j < jf || j > jl) { ok = false; walk = false; go = false; done = true; }
                        else {
                            const bool top = sb == s0;
                            const uint32_t c1 = top ? A[x].x : Bv[x].x, c0 = top ? A[x].y : Bv[x].y;
                            const int bit = (i - 1) & 31;
                            const uint32_t stay = ~(c1 & ~c0) & (0xFFFFFFFFu >> (31 - bit));      /* rows at or above this one whose move is not 'up' */
                            if (stay == 0u) {
                                i -= bit + 1; ncols += (uint32_t)(bit + 1);
                                if (i == 0) done = true;
                            } else {
                                const int row = 31 - __clz((int)stay);
                                i -= bit - row; ncols += (uint32_t)(bit - row);
                                const uint32_t r1 = (c1 >> row) & 1u, r0 = (c0 >> row) & 1u;
                                if (r0 && !r1) { j -= 1; ncols += 1; }                              /* left */
                                else { nmatch += r1 ^ 1u; i -= 1; j -= 1; ncols += 1; }           /* diagonal: 00 match, 11 mismatch */
                                done = true;
                            }
                        }
                    }
                }
            }
            if (!done) walk = false;
        }
        go = go && ok && i > 0 && j > 0;
    }
    if (valid) {
        ncols += (uint32_t)(i + j);
        if (ok && (ncols - nmatch) > (uint32_t)kb) ok = false;
    }
    *out_ncols = ok ? ncols : 0u; *out_nmatch = ok ? nmatch : 0u; *out_ok = ok;
}


__global__ void __launch_bounds__(64, 4) k_win_lane(MS *msv, const uint32_t *req, const uint32_t *cls_cnt, uint32_t stride,
                                                     const uint8_t *winbuf, uint2 *tbw_base) {
    const int lane = lane_id();
    uint2 *tbw = tbw_base + (uint64_t)blockIdx.x * BRX_LANE_TB_UNITS;
    /* groups of 64 windows, the widest class first (it is also the longest-running: every column computes more slots) */
    uint32_t cnt[BRX_LANE_CLASSES], total = 0;
#pragma unroll
    for (int cidx = 0; cidx < BRX_LANE_CLASSES; ++cidx) { cnt[cidx] = uni(cls_cnt[cidx * BRX_CLS_STRIDE]); total += (cnt[cidx] + 63u) >> 6; }
    for (uint32_t grp = blockIdx.x; grp < total; grp += gridDim.x) {
        uint32_t g0 = grp, cls = 0, n_req = 0;
#pragma unroll
        for (int cidx = BRX_LANE_CLASSES - 1; cidx >= 0; --cidx) {
            const uint32_t ng = (cnt[cidx] + 63u) >> 6;
            if (n_req == 0u && g0 < ng) { cls = (uint32_t)cidx; n_req = cnt[cidx]; }
            else if (n_req == 0u) g0 -= ng;
        }
        const uint32_t idx = g0 * 64u + (uint32_t)lane;
        const bool valid = idx < n_req;
        const uint32_t r = valid ? req[(size_t)cls * stride + idx] : 0u;
        MS ms;
        if (valid) ms = msv[r];
        const int Q = valid ? (int)(ms.win_b - ms.win_a) : 0, T = valid ? (int)ms.tl : 0, kb = valid ? (int)ms.cost : 0;
        /* the window pair of every lane as bit planes: written by the parking wave behind the bytes of its slot */
        const uint32_t *pl = reinterpret_cast<const uint32_t *>(winbuf + (uint64_t)r * BRX_WIN_STRIDE + BRX_WIN_PLANES);
        uint32_t ncols = 0, nmatch = 0; bool ok = false;
        brx_lanes_align<BRX_LANE_TMAX / 32>(valid, pl, Q, T, kb, tbw, &ncols, &nmatch, &ok);
        if (valid) {
            msv[r].res_ncols = ncols; msv[r].res_nmatch = nmatch;
            if (!ok) msv[r].status = ms.status | BRX_RS_BAND;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
    }
}

#endif /* BRX_MUTATE_H */

/*
 * brx_finlanes.h -- the FINAL alignment of reads with a narrow band, one READ per LANE (round 4).
 *
 * get_qscores aligns the whole read against its fragment (/root/reference/badread/qscore_model.py:37,
 * edlib.align(..., task='path')).  With pacbio2021 / --identity 30,3 (BASELINE.json configs[4]) a 15 kb read differs
 * from its fragment in ~30 places: the proven edit bound puts the band at 30-60 diagonals -- TWO or three of the 64
 * lanes of the wave-systolic aligner (brx_align.h), which then issues ~24 instructions per column for one read
 * (k_fin_align<1,1,1>: 32.5 of that workload's 37.6 instructions per base, profiles/valu_per_base.json).
 *
 * Here a lane owns a read, as a lane of k_win_lane owns a window (brx_mutate.h, brx_lanes_align: same band from
 * brx_make_geom, same cell recurrence, same skew -- in loop trip jj a lane works on ITS column j = jj - off so that
 * every lane's band moves in the trips jj = 0 (mod 32) --, same 2-bit move codes in a [trip][slot][lane] layout, same
 * canonical walk: up / 'I', left / 'D', diagonal).  What differs:
 *   - the strings are as long as reads are: no plane arrays.  At every shift a lane turns the next 32 query bytes (the
 *     block entering its band) and the next 32 target bytes (its columns of the next 32 trips) into two plane words
 *     each, by multiplication (brx_byte_bits): ~4 instructions per trip;
 *   - W band blocks per lane in registers (BRX_FINL_W = 4: bands up to 88 diagonals);
 *   - the walk WRITES the ops ('=' 0, 'X' 1, 'I' 2, 'D' 3), last column first, downwards from ops_end -- what
 *     k_fin_qscore reads -- instead of counting them.
 * Symbols outside ACGT (reads over N runs) and wider bands keep to k_fin_align.
 */
#ifndef BRX_FINLANES_H
#define BRX_FINLANES_H

#define BRX_FINL_W 4                                   /* band blocks of 32 rows per lane */
#define BRX_FINL_TBC 16                                /* traceback columns fetched per round trip */
/* The traceback in STRIPS (round 5).  The move codes of every cell of the band were 8 bytes per block and
 * column and lane -- 23 MB per wave of 15 kb reads -- and the arena held them for 250-330 of a batch's 1024 groups at a time:
 * the kernel ran on a quarter of its waves (configs[4]: 10.4 instructions per base at 0.29 of the issue rate).  Now the
 * forward pass keeps only the band state {P, M} of every BRX_FINL_STRIP-th trip; the walk goes strip by strip from the last
 * one, recomputing a strip's codes from its checkpoint into a strip buffer of 256 trips (which stays in the cache) before it
 * walks it.  Twice the forward instructions (3 of the workload's 10.4 per base), 1/47 of the store for a 15 kb read. */
#define BRX_FINL_STRIP 256                             /* trips per strip: a multiple of 32 (the band moves and the target planes are fetched at trips = 0 mod 32) */
/* units (uint2) a wave needs for reads of up to t_max columns whose widest band has `blocks` blocks: the checkpoints [strip][slot][lane]
   and, behind them, the strip buffer [trip of the strip][slot][lane] */
__host__ __device__ inline uint64_t brx_finl_strips(uint32_t t_max) { return ((uint64_t)t_max + 66u) / BRX_FINL_STRIP + 2u; }
__host__ __device__ inline uint64_t brx_finl_units(uint32_t t_max, uint32_t blocks) {
    return (brx_finl_strips(t_max) + (uint64_t)BRX_FINL_STRIP) * blocks * 64u;
}

/* band blocks of a read's final alignment, or 0 when the band does not fit this aligner */
__host__ __device__ inline int brx_finl_blocks(uint32_t m, uint32_t n, uint32_t ub) {
    if (m == 0 || n == 0) return 0;
    const BrxGeom g = brx_make_geom((int)m, (int)n, (int)ub);
    if (g.G != 1) return 0;
    const int bb = (g.dhi - g.dlo) / 32 + 2;
    return bb <= BRX_FINL_W ? bb : 0;
}

/* plane words of 32 symbols at p[0..32) restricted to indices [0, len): bit t of *lo / *hi = bit 0 / 1 of p[t] */
__device__ __forceinline__ void brx_finl_planes32(const uint8_t *p, int len, uint32_t *lo, uint32_t *hi) {
    uint32_t l = 0u, h = 0u;
    if (len > 0) {
        const BrxB16 a = *reinterpret_cast<const BrxB16 *>(p), b = *reinterpret_cast<const BrxB16 *>(p + 16);
        const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) { l |= brx_byte_bits(v[q], 0) << (4 * q); h |= brx_byte_bits(v[q], 1) << (4 * q); }
        if (len < 32) { const uint32_t keep = (1u << len) - 1u; l &= keep; h &= keep; }
    }
    *lo = l; *hi = h;
}

/* Up to 64 final alignments, one per lane.  Qs / Ts: the lane's read (Q bytes, 32 readable bytes behind it) and fragment
 * (T bytes, likewise); kb: the proven bound; tbw: the wave's move-code store (brx_finl_units of the longest T of the
 * wave); ops_end: one past the last op of the lane's read.  Returns through out_*: columns, matches, ok. */
__device__ inline void brx_lanes_final(const bool valid, const uint8_t *__restrict__ Qs, const int Q, const uint8_t *__restrict__ Ts, const int T,
                                       const int kb, uint2 *__restrict__ tbw, uint8_t *__restrict__ ops_end,
                                       uint32_t *out_ncols, uint32_t *out_nmatch, bool *out_ok) {
    constexpr int W = BRX_FINL_W;
    const int lane = lane_id();
    const BrxGeom g = brx_make_geom(Q > 0 ? Q : 1, T > 0 ? T : 1, kb);
    const int NS = (Q + 31) >> 5;
    const int Wb = (int)wave_max_u32(valid ? (uint32_t)((g.dhi - g.dlo) / 32 + 2) : 0u);      /* slots in use: the widest band of the wave */
    const int off = (g.dlo - 1) & 31;                       /* jj = j + off; (j + dlo - 1) >> 5 = (jj >> 5) + qb */
    const int qb = (g.dlo - 1 - off) >> 5;                  /* exact: dlo - 1 - off is a multiple of 32; negative */
    const int JJ = (int)wave_max_u32(valid ? (uint32_t)(T + off) : 0u);

    uint32_t P[W], M[W], QL[W], QH[W];
    int slo = 0;                                            /* block held in slot 0 */
    uint32_t TLw = 0u, THw = 0u;                            /* target planes of the columns of the next 32 trips */
    uint2 *const ckpt = tbw;                                /* [strip][slot][lane] {P, M} before the strip's first trip */
    uint2 *const strip = tbw + ((uint64_t)(JJ / BRX_FINL_STRIP) + 1u) * (uint64_t)Wb * 64u;      /* [trip of the strip][slot][lane] move codes */
    /* the lane's state before trip jj0 (a multiple of BRX_FINL_STRIP): band state from the checkpoint (jj0 = 0: the matrix's left
       edge), the band's position and the query planes of its blocks from the strings */
    auto restore = [&](const int jj0) {
        const int b = (jj0 >> 5) - 1 + qb;
        slo = (valid && jj0 > 0 && b >= 1) ? b : 0;
#pragma unroll
        for (int x = 0; x < W; ++x) {
            P[x] = 0xFFFFFFFFu; M[x] = 0u;                  /* cells below the band grow by +1 per row */
            if (jj0 > 0 && x < Wb) { const uint2 v = ckpt[((uint64_t)(jj0 / BRX_FINL_STRIP) * (uint64_t)Wb + (uint32_t)x) * 64u + (uint32_t)lane]; P[x] = v.x; M[x] = v.y; }
            QL[x] = 0u; QH[x] = 0u;
            if (valid && slo + x < NS) brx_finl_planes32(Qs + 32 * (slo + x), Q - 32 * (slo + x), &QL[x], &QH[x]);
        }
        TLw = 0u; THw = 0u;
    };
    /* trips jj0 .. jj1 - 1.  codes == nullptr: the first pass, which only leaves the checkpoints; else the move codes of the
       trips go to codes[jj - jj0][slot][lane] */
    auto forward = [&](const int jj0, const int jj1, uint2 *const codes) {
        for (int jj = jj0; jj < jj1; ++jj) {
            if ((jj & 31) == 0) {
                if (codes == nullptr && (jj & (BRX_FINL_STRIP - 1)) == 0 && jj > 0) {
#pragma unroll
                    for (int x = 0; x < W; ++x) { if (x < Wb) ckpt[((uint64_t)(jj / BRX_FINL_STRIP) * (uint64_t)Wb + (uint32_t)x) * 64u + (uint32_t)lane] = make_uint2(P[x], M[x]); }
                }
                /* ---- the band moves down one block (lanes whose band still starts at block 0 stay) ---- */
                const int bq = (jj >> 5) + qb;
                if (valid && bq >= 1) {
#pragma unroll
                    for (int x = 0; x + 1 < W; ++x) { P[x] = P[x + 1]; M[x] = M[x + 1]; QL[x] = QL[x + 1]; QH[x] = QH[x + 1]; }
                    const int nb = bq + W - 1;
                    P[W - 1] = 0xFFFFFFFFu; M[W - 1] = 0u;
                    QL[W - 1] = 0u; QH[W - 1] = 0u;
                    if (nb < NS) brx_finl_planes32(Qs + 32 * nb, Q - 32 * nb, &QL[W - 1], &QH[W - 1]);
                    slo = bq;
                }
                /* ---- target planes of the next 32 trips: bit t = target index (jj - off - 1) + t ---- */
                const int t0 = jj - off - 1;
                TLw = 0u; THw = 0u;
                if (valid && t0 < T && t0 + 32 > 0) {
                    if (t0 >= 0) brx_finl_planes32(Ts + t0, T - t0, &TLw, &THw);
                    else {                                  /* the first window of a lane with off < 31 starts left of the string */
                        uint32_t l0, h0;
                        brx_finl_planes32(Ts, T, &l0, &h0);
                        TLw = l0 << (uint32_t)(-t0); THw = h0 << (uint32_t)(-t0);
                    }
                }
            }
            const int j = jj - off;
            const bool act = valid && j >= 1 && j <= T;
            int hi = (j + g.dhi - 1) >> 5;                  /* last block of the band in column j ... */
            if (hi > NS - 1) hi = NS - 1;
            hi = act ? hi - slo : -1;                       /* ... as a slot; slots 0 .. hi are computed */
            const int b = jj & 31;
            const uint32_t m0 = brx_bfe_mask(TLw, b), m1 = brx_bfe_mask(THw, b);
            uint32_t hp = 1u, hm = 0u;                      /* above the band (and above row 1): +1 per column */
            uint2 *dst = codes + ((uint64_t)(jj - jj0) * (uint64_t)Wb) * 64u + (uint32_t)lane;
#pragma unroll
            for (int x = 0; x < W; ++x) {
                if (x >= Wb) break;
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
                if (on && codes != nullptr) {
                    const uint32_t dX = ~(pv | Ph | Eq);    /* diagonal move on different symbols */
                    dst[(uint32_t)x * 64u] = make_uint2(pv | dX, (Ph & ~pv) | dX);
                }
                hp = Ph >> 31; hm = Mh >> 31;               /* the computed slots are 0 .. hi: every carry that is used was computed */
            }
        }
    };
    restore(0);
    forward(0, JJ + 1, nullptr);                            /* first pass: the checkpoints */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);                          /* this wave's stores are visible to its loads below */

    /* ---- traceback, canonical (up, left, diagonal), strip by strip from the last one: the strip's codes again from its
       checkpoint, then the walk of brx_lanes_align over them -- BRX_FINL_TBC columns fetched per round trip -- writing the ops.
       code 10 = up, 01 = left, 00 = diagonal on equal symbols, 11 = on different ones.  Checkpoints and strip buffer are
       columns of the LANE: no lane reads what another wrote. ---- */
    int i = Q, j = T;
    uint32_t ncols = 0, nmatch = 0;
    bool ok = valid;
    bool go = valid && i > 0 && j > 0;
    for (int k0 = (JJ / BRX_FINL_STRIP) * BRX_FINL_STRIP; k0 >= 0; k0 -= BRX_FINL_STRIP) {
        if (__ballot(go && j + off >= k0) == 0ull) continue;    /* no lane's path is inside this strip (reads of different lengths) */
        restore(k0);
        forward(k0, (k0 + BRX_FINL_STRIP < JJ + 1) ? k0 + BRX_FINL_STRIP : JJ + 1, strip);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        while (__ballot(go && j + off >= k0) != 0ull) {
            const bool here = go && j + off >= k0;          /* the lane's column lies in this strip */
            const int s0 = here ? ((i - 1) >> 5) : 0;
            const int jst = j;
            uint2 A[BRX_FINL_TBC], Bv[BRX_FINL_TBC];
#pragma unroll
            for (int x = 0; x < BRX_FINL_TBC; ++x) {
                const int col = jst - x;
                A[x] = make_uint2(0u, 0u); Bv[x] = make_uint2(0u, 0u);
                if (here && col >= 1 && col + off >= k0) {
                    int sl = (col + g.dlo - 1) >> 5; if (sl < 0) sl = 0;
                    const int xa = s0 - sl;
                    const uint64_t rowb = (uint64_t)(col + off - k0) * (uint64_t)Wb;
                    if (xa >= 0 && xa < Wb) A[x] = strip[(rowb + (uint32_t)xa) * 64u + (uint32_t)lane];
                    if (xa >= 1 && xa - 1 < Wb) Bv[x] = strip[(rowb + (uint32_t)(xa - 1)) * 64u + (uint32_t)lane];
                }
            }
            bool walk = here;
#pragma unroll
            for (int x = 0; x < BRX_FINL_TBC; ++x) {
                if (jst - x + off < k0) walk = false;       /* the next column belongs to the strip before this one: not computed yet */
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
                                const int row = stay == 0u ? -1 : 31 - __clz((int)stay);
                                const int ups = bit - row;                                            /* the run of 'up' moves below the row it stops in */
                                for (int t = 0; t < ups; ++t) ops_end[-1 - (int)(ncols + (uint32_t)t)] = BRX_OP_I;
                                i -= ups; ncols += (uint32_t)ups;
                                if (row < 0) { if (i == 0) done = true; }                             /* the whole block was 'up': the block above is next */
                                else {
                                    const uint32_t r1 = (c1 >> row) & 1u, r0 = (c0 >> row) & 1u;
                                    if (r0 && !r1) { ops_end[-1 - (int)ncols] = BRX_OP_D; j -= 1; ncols += 1; }               /* left */
                                    else { ops_end[-1 - (int)ncols] = (uint8_t)(r1 ? BRX_OP_X : BRX_OP_EQ); nmatch += r1 ^ 1u; i -= 1; j -= 1; ncols += 1; }
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
    }
    if (valid && ok) {
        /* the corner: what is left of the read above row i ('I') or of the fragment left of column j ('D') */
        for (int t = 0; t < i; ++t) ops_end[-1 - (int)(ncols + (uint32_t)t)] = BRX_OP_I;
        ncols += (uint32_t)i;
        for (int t = 0; t < j; ++t) ops_end[-1 - (int)(ncols + (uint32_t)t)] = BRX_OP_D;
        ncols += (uint32_t)j;
        if ((ncols - nmatch) > (uint32_t)kb) ok = false;
    }
    *out_ncols = ok ? ncols : 0u; *out_nmatch = ok ? nmatch : 0u; *out_ok = ok;
}

#endif /* BRX_FINLANES_H */

/*
 * brx_quad.h -- FOUR final alignments per wavefront, one per 16-lane DPP row (round 5).
 *
 * get_qscores aligns the whole read against its fragment (/root/reference/badread/qscore_model.py:37,
 * edlib.align(..., task='path')).  The wave-systolic aligner of brx_align.h gives an alignment all 64 lanes, one 32-row
 * superblock per lane, and a band of b superblocks keeps b + 2 of them busy: on BASELINE.json configs[3] 14 % of the
 * simulated bases belong to reads whose band is at most 13 superblocks and another 22 % to reads with 14-26 -- for those
 * the 24 instructions of a column update were issued for 64 lanes to advance 6-28 of them (k_fin_align<1,1,1>: 25.8 of the
 * path's 90.5 instructions per base at 0.48 useful lanes, profiles/valu_per_base.json of round 4).
 *
 * Here a ROW of 16 lanes owns an alignment: the same recurrence, the same one-column lag between neighbouring
 * superblocks, the same U columns per loop trip and the same traceback store rows as brx_align_forward_u -- but the
 * carry travels by DPP row_ror:1 (a rotation inside the row: native on CDNA, no patching of the row ends), a lane hops 16
 * superblocks ahead when its own leaves the band, and the bookkeeping that was scalar because it was the same for the
 * whole wave (entries, exits, the refills of the target window) is either per lane or made the same for the four rows:
 *   - bands of up to BRX_QUAD_SPAN = 13 superblocks: one word per lane, eight columns per trip, for up to 416 diagonals;
 *     two words per lane, four columns per trip, for up to 832 (the 15-26-superblock reads, which two 32-lane halves
 *     would take at the same rate -- but rows need no lane patching);
 *   - the four alignments of a wave start together (reads are listed by store size: similar lengths) and share the loop
 *     counter, so the corner trips (columns left of column 1: rolled path) coincide;
 *   - every global LOAD of the loop happens in one wave-uniform refill event per 32 trips: each row keeps a 1 KB LDS ring
 *     of target bytes AND a 1 KB LDS ring of query planes (16 bytes per word); a lane entering a new superblock reads its
 *     planes from LDS.  (brx_align_forward_u loads them from global memory at every entry -- a load in a loop of stores
 *     waits for every store before it, brx_align.h -- once per four trips; with four rows that would be every trip.);
 *   - the four traceback stores are ONE slab whose rows hold the rows of the four alignments side by side (BrxGeom.WSrow,
 *     .slot0), so a store row's address is still wave-uniform: scalar base + lane offset.
 * The planes tables are built and the tracebacks walked one alignment after the other by the whole wave (brx_build_planes,
 * brx_align_traceback: unchanged, 64 lanes speculating down the diagonal).  Reads with symbols outside ACGT keep to
 * k_fin_align (no rare-symbol path here); a read whose traceback leaves the windowed store is repeated by k_fin_align in
 * the second phase like any other.  Results are those of brx_wave_align: same band, same cells, same canonical path.
 */
#ifndef BRX_QUAD_H
#define BRX_QUAD_H

#define BRX_QUAD_LW 16            /* lanes of an alignment */
#define BRX_QUAD_SPAN 13          /* widest band in superblocks: R * 15 + 18 - U >= band width + R is what the 16-lane hop needs */
#define BRX_QUAD_MAXG 2
#define BRX_QRING_TW 256          /* 32-bit words of a row's target ring: four 256-byte chunks */
#define BRX_QRING_PW 64           /* entries (16 bytes: one query word's planes) of a row's planes ring */
#define BRX_QUAD_PERIOD 32        /* trips between refill events */

__host__ __device__ inline BrxGeom brx_make_geom_quad(int Q, int T, int k, int hmul) {
    return brx_make_geom_span(Q, T, k, hmul, BRX_QUAD_SPAN, BRX_QUAD_MAXG);
}
/* words per lane (1 or 2) of a read's final alignment as one of four per wave, or 0 when its band is too wide */
__host__ __device__ inline int brx_quad_words(uint32_t m, uint32_t n, uint32_t ub) {
    if (m == 0 || n == 0) return 0;
    return brx_make_geom_quad((int)m, (int)n, (int)ub, 0).G;
}
/* 8-byte units of the slab of a group of up to four alignments: rows of 4 x WSq slots x G words for the longest of them,
   then each one's planes table */
__host__ __device__ inline uint64_t brx_quad_units(const BrxGeom *g4, int n) {
    int wsq = 1, rows = 0, G = 1;
    uint64_t peq = 0;
    for (int i = 0; i < n; ++i) {
        if (g4[i].G == 0) continue;
        G = g4[i].G;
        if (g4[i].WSp > wsq) wsq = g4[i].WSp;
        if (g4[i].t_end + 1 > rows) rows = g4[i].t_end + 1;
        peq += brx_peq_units(g4[i]);
    }
    return (uint64_t)rows * (uint64_t)(4 * wsq) * (uint64_t)G + peq;
}

__shared__ uint32_t brx_qring_t[4 * BRX_QRING_TW];
__shared__ uint4 brx_qring_p[4 * BRX_QRING_PW];

/* lane i receives the value of lane (i - 1) mod 16 of its row: DPP row_ror:1 */
__device__ __forceinline__ int brx_from_row_lane_above(int v) {
    return __builtin_amdgcn_mov_dpp(v, 0x121 /* row_ror:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t brx_wave_max_rows(uint32_t v) {        /* maximum over the four rows of a value that is the same inside a row */
    const uint32_t a = (uint32_t)__shfl_xor((int)v, 16, 64);
    v = a > v ? a : v;
    const uint32_t b = (uint32_t)__shfl_xor((int)v, 32, 64);
    v = b > v ? b : v;
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

/* Forward pass of up to four alignments.  Ts, g, planes: the ROW's target bytes (16 readable bytes behind the end),
   geometry (g.WSp already the wave's common slot count, g.slot0 the row's first slot; live = the row holds an alignment)
   and planes table; tb: the wave's slab (uniform). */
template <int U, int G>
__device__ inline void brx_quad_forward(const bool live, const uint8_t *__restrict__ Ts, const BrxGeom g, uint2 *__restrict__ tb,
                                        const uint2 *__restrict__ planes) {
    static_assert((U == 8 && G == 1) || (U == 4 && G == 2), "columns per trip x words per lane");
    constexpr int LU = U == 8 ? 3 : 2;
    constexpr int R = 32 * G;
    constexpr int LW = BRX_QUAD_LW;
    constexpr int NEVER = 0x7FFFFFFF;
    const int lane = threadIdx.x & 63;
    const int l16 = lane & 15, row = lane >> 4;
    uint32_t *const ringT = brx_qring_t + row * BRX_QRING_TW;
    uint4 *const ringP = brx_qring_p + row * BRX_QRING_PW;
    const uint64_t tb_addr = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)tb >> 32)) << 32) |
                             (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint64_t)tb);
    const int WSq = __builtin_amdgcn_readfirstlane(g.WSp);            /* the same in every row (brx_quad_align) */
    const int NS = live ? g.NS : 0, NW = live ? g.NW : 0, T = live ? g.T : 0;
    const int dlo = g.dlo, dhi = g.dhi;
    auto jfirst = [&](int x) { const int j = R * x - dhi + 1; return j < 1 ? 1 : j; };
    auto jlast = [&](int x) { const long long j = (long long)R * (x + 1) - dlo; return j > T ? T : (int)j; };
    auto tf_of = [&](int x) { return (jfirst(x) + x - 1) >> LU; };

    int s = l16;
    int slot = l16 % WSq;
    const int slot_step = LW % WSq;
    uint32_t slot8 = 0;
    int tf = NEVER, tl = NEVER;
    uint32_t tspan = 0;
    const uint32_t keep_lim = (uint32_t)(2 * g.H + R - 1);
    int keep_base = 0;
    auto window = [&]() {                            /* everything that depends on s */
        tf = NEVER; tl = NEVER; tspan = 0;
        if (s < NS) {
            const int jf = jfirst(s), jl = jlast(s);
            slot8 = 8u * (uint32_t)G * (uint32_t)(g.slot0 + slot);
            tl = (jl + s - 1) >> LU;
            if (jl >= jf) { tf = (jf + s - 1) >> LU; tspan = (uint32_t)(tl - tf); }   /* empty window: never active, but it still hops at tl */
        }
        keep_base = R * s + g.H + R - 1;
    };
    uint32_t Pv[G], Mv[G];
    BrxQPlanes qp[G];
    auto enter = [&]() {                             /* the lane takes superblock s: its window, its planes from the LDS ring */
        window();
#pragma unroll
        for (int x = 0; x < G; ++x) {
            Pv[x] = 0xFFFFFFFFu; Mv[x] = 0;          /* cells below the band grow by +1 per row */
            const int w = s * G + x;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (w < NW) v = ringP[w & (BRX_QRING_PW - 1)];
            qp[x] = BrxQPlanes{v.x, v.y, v.z, v.w};
        }
    };
    constexpr uint32_t IDLE = 0x80000000u;
    uint32_t carry = IDLE;

    /* ---- the rows' rings: target chunks 0..2, planes of the words 0..47 ---- */
    auto fetch_target = [&](int c) -> uint4 {
        const int idx = 256 * c + 16 * l16;
        uint4 v = make_uint4(0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu);
        if (idx < T) {                               /* T + 16 bytes are readable */
            const uint32_t *p = reinterpret_cast<const uint32_t *>(Ts + idx);
            v = make_uint4(p[0], p[1], p[2], p[3]);
        }
        return v;
    };
    auto put_target = [&](int c, uint4 v) {
        uint32_t *d = ringT + (c & 3) * 64 + 4 * l16;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    };
    auto fetch_planes = [&](int c) -> uint4 {
        const int w = 16 * c + l16;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (w < NW) { const uint2 a = planes[2 * w], b = planes[2 * w + 1]; v = make_uint4(a.x, a.y, b.x, b.y); }
        return v;
    };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        put_target(c, fetch_target(c));
        ringP[(16 * c + l16) & (BRX_QRING_PW - 1)] = fetch_planes(c);
    }
    int nc = 3, pc = 3;                              /* next chunks to fetch (per row) */
    uint4 pendT = make_uint4(0u, 0u, 0u, 0u), pendP = make_uint4(0u, 0u, 0u, 0u);
    int pendT_c = -1, pendP_c = -1;                  /* chunk a pending fetch belongs to, -1 = none */
    __builtin_amdgcn_wave_barrier();                 /* other lanes' ring entries are read below (LDS operations of a wave run in order) */
    enter();

    /* last trip of any row; trips in which a lane of the first superblocks would compute columns left of column 1 */
    const int tau_end = (int)brx_wave_max_rows(live && NS > 0 ? (uint32_t)(((T + NS - 2) >> LU) + 1) : 0u) - 1;
    int sv = (dhi + U - 2) / R;
    if (sv > NS - 1) sv = NS - 1;
    if (sv > LW - 1) sv = LW - 1;
    const int tau_pro = (int)brx_wave_max_rows(live && sv >= 1 ? (uint32_t)(tf_of(sv) + 1) : 0u) - 1;

    BRX_GLOBAL char *row0 = (BRX_GLOBAL char *)((BRX_GLOBAL uint64_t *)tb_addr + (size_t)g.WSrow * (size_t)G);
    const size_t row_bytes = 8 * (size_t)__builtin_amdgcn_readfirstlane(g.WSrow) * (size_t)G;
    const size_t trip_bytes = (size_t)U * row_bytes;
    auto ring_bytes = [&](int tau_, uint32_t *x0, uint32_t *x1) {
        const uint32_t b0 = (uint32_t)(U * tau_ - s);
        const uint32_t d = b0 >> 2;
        const uint32_t w0 = ringT[d & (BRX_QRING_TW - 1)], w1 = ringT[(d + 1) & (BRX_QRING_TW - 1)];
        *x0 = brx_funnel_bytes(w1, w0, b0);
        if constexpr (U == 8) { const uint32_t w2 = ringT[(d + 2) & (BRX_QRING_TW - 1)]; *x1 = brx_funnel_bytes(w2, w1, b0); }
        else *x1 = 0u;
    };
    /* superblocks whose (unclamped) last trip is at most t: the largest such x, -1 if none */
    auto left_by = [&](int t) { const int num = U * (t + 1) + dlo - R; return num >= 0 ? num / (R + 1) : -1; };
    uint32_t xn0, xn1;
    ring_bytes(0, &xn0, &xn1);
    for (int tau = 0; tau <= tau_end; ++tau, row0 += trip_bytes) {
        /* ---- refill event, the same trips for the four rows: what the last event fetched goes into the rings, and each row
                fetches the target chunk / the sixteen plane entries it will need from the next event on ---- */
        if (__builtin_expect((tau & (BRX_QUAD_PERIOD - 1)) == 0 && tau > 0, 0)) {
            if (pendT_c >= 0) put_target(pendT_c, pendT);
            if (pendP_c >= 0) ringP[(16 * pendP_c + l16) & (BRX_QRING_PW - 1)] = pendP;
            pendT_c = -1; pendP_c = -1;
            const int t2 = tau + 2 * BRX_QUAD_PERIOD;                    /* what is fetched now is in LDS from tau + PERIOD and must serve until t2 */
            const int gone = left_by(t2);
            const int s_top_lo = gone > 1 ? gone - 1 : 0;                /* never above the first superblock still in the band at t2 */
            if (live && 256 * nc < U * t2 + U - s_top_lo + 32 && 256 * nc < T) { pendT = fetch_target(nc); pendT_c = nc; nc += 1; }
            if (live && 16 * pc < G * (gone + LW + 3) && 16 * pc < NW) { pendP = fetch_planes(pc); pendP_c = pc; pc += 1; }
            __builtin_amdgcn_wave_barrier();                              /* (the bytes of this trip were read before the ring changed: a chunk far behind them was replaced) */
        }

        /* ---- U column updates ---- */
        const uint32_t x0 = xn0, x1 = xn1;
        const bool act = (uint32_t)(tau - tf) <= tspan;
        const bool keep = brx_keep_trip(g.slope, keep_base, keep_lim, brx_jrep_trip(U, tau, s));
        uint32_t P[G], M[G];
#pragma unroll
        for (int x = 0; x < G; ++x) { P[x] = Pv[x]; M[x] = Mv[x]; }
        if (__builtin_expect(tau <= tau_pro, 0)) {
            /* rolled: the corner trips, whose first columns may lie left of column 1 */
#pragma unroll 1
            for (int c = 0; c < U; ++c) {
                const uint32_t nb = (uint32_t)brx_from_row_lane_above((int)carry);
                const int j = U * tau + c + 1 - s;
                const bool real = act && j >= 1;
                uint32_t hm = nb & 1u, hp = nb >> 31;
                const uint32_t ch = ((c < 4 ? x0 : x1) >> (8 * (c & 3))) & 0xFFu;
                BRX_GLOBAL uint64_t *dst = (BRX_GLOBAL uint64_t *)(row0 + (size_t)c * row_bytes + slot8);
#pragma unroll
                for (int x = 0; x < G; ++x) {
                    const uint32_t Eq = brx_eq_acgt(qp[x], 0u - (ch & 1u), 0u - ((ch >> 1) & 1u));
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
                const uint32_t nb = (uint32_t)brx_from_row_lane_above((int)carry);
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
            if (act && keep) {
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

        /* ---- the lane's superblock has left the band: it takes superblock s + 16 (per lane: the rows differ) ---- */
        if (tau >= tl) {
            s += LW; slot += slot_step; if (slot >= WSq) slot -= WSq;
            enter();
        }
        ring_bytes(tau + 1, &xn0, &xn1);
    }
}

/* geometry of row r as wave-uniform values (every lane of a row holds its row's) */
__device__ inline BrxGeom brx_quad_row_geom(const BrxGeom &g, int r) {
    BrxGeom o;
    const int src = 16 * r;
#define BRX_QROW(f) o.f = __builtin_amdgcn_readfirstlane(__shfl((int)g.f, src, 64))
    BRX_QROW(Q); BRX_QROW(T); BRX_QROW(dlo); BRX_QROW(dhi); BRX_QROW(G); BRX_QROW(R); BRX_QROW(NS); BRX_QROW(NW);
    BRX_QROW(WSp); BRX_QROW(WSrow); BRX_QROW(slot0); BRX_QROW(K); BRX_QROW(U); BRX_QROW(t_end); BRX_QROW(H);
#undef BRX_QROW
    o.slope = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)g.slope, src, 64));
    return o;
}
__device__ inline uint64_t brx_quad_row_u64(uint64_t v, int r) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)(uint32_t)v, 16 * r, 64));
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)(uint32_t)(v >> 32), 16 * r, 64));
    return ((uint64_t)hi << 32) | (uint64_t)lo;
}

/* Up to four alignments, one per row; every lane of a row passes its row's arguments (valid = the row holds a pair with
 * Q > 0 and T > 0 whose quad geometry has G words per lane).  tb / tb_cap: the wave's slab.  Per row on return: *n_cols,
 * *n_match and *st = 0 done, 1 the traceback left the band or the stored window (or cost above k), 2 the slab is too small. */
template <int G>
__device__ inline void brx_quad_align(const bool valid, const uint8_t *Qs, const int Q, const uint8_t *Ts, const int T, const int k,
                                      const int hmul, uint2 *tb, const uint64_t tb_cap, uint8_t *ops_end,
                                      int *n_cols, int *n_match, int *st) {
    const int lane = threadIdx.x & 63;
    const int row = lane >> 4;
    BrxGeom g = brx_make_geom_quad(valid ? Q : 1, valid ? T : 1, valid ? k : 0, hmul);
    const bool live = valid && g.G == G;
    const int WSq = (int)brx_wave_max_rows(live ? (uint32_t)g.WSp : 1u);
    const int rows = (int)brx_wave_max_rows(live ? (uint32_t)(g.t_end + 1) : 0u);
    g.WSp = WSq; g.WSrow = 4 * WSq; g.slot0 = row * WSq;
    /* the slab: rows x (4 x WSq) slots x G words, then the planes tables of the rows */
    const uint64_t tb_units = (uint64_t)rows * (uint64_t)(4 * WSq) * (uint64_t)G;
    const uint64_t peq = live ? brx_peq_units(g) : 0ull;
    uint64_t planes_at = tb_units, total = tb_units;
    for (int r = 0; r < 4; ++r) {
        const uint64_t p = brx_quad_row_u64(peq, r);
        if (r < row) planes_at += p;
        total += p;
    }
    *n_cols = 0; *n_match = 0; *st = live ? 0 : 1;
    if (total > tb_cap) { *st = 2; return; }
    uint2 *planes = tb + planes_at;
    for (int r = 0; r < 4; ++r) {                    /* the planes of every query word, one query after the other by the whole wave */
        if (!__builtin_amdgcn_readfirstlane((int)__shfl((int)live, 16 * r, 64))) continue;
        const BrxGeom gr = brx_quad_row_geom(g, r);
        brx_build_planes(reinterpret_cast<const uint8_t *>(brx_quad_row_u64((uint64_t)Qs, r)), gr,
                         reinterpret_cast<uint2 *>(brx_quad_row_u64((uint64_t)planes, r)));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);                   /* the tables are read back by other lanes of this wave */
    brx_quad_forward<(G == 1 ? 8 : 4), G>(live, Ts, g, tb, planes);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);                   /* stores of this wave visible to its own later loads */
    for (int r = 0; r < 4; ++r) {
        if (!__builtin_amdgcn_readfirstlane((int)__shfl((int)live, 16 * r, 64))) continue;
        const BrxGeom gr = brx_quad_row_geom(g, r);
        const int kr = __builtin_amdgcn_readfirstlane(__shfl(k, 16 * r, 64));
        int nc = 0, nm = 0;
        bool ok = brx_align_traceback(reinterpret_cast<const uint8_t *>(brx_quad_row_u64((uint64_t)Qs, r)),
                                      reinterpret_cast<const uint8_t *>(brx_quad_row_u64((uint64_t)Ts, r)), gr, tb,
                                      reinterpret_cast<uint8_t *>(brx_quad_row_u64((uint64_t)ops_end, r)), &nc, &nm);
        if (ok && nc - nm > kr) ok = false;
        if (row == r) { *n_cols = nc; *n_match = nm; *st = ok ? 0 : 1; }
    }
}

#endif /* BRX_QUAD_H */

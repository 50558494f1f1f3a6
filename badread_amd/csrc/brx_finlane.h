/*
 * brx_finlane.h -- final alignment of get_qscores (/root/reference/badread/qscore_model.py:37) with ONE
 * READ PER LANE, included by brx_kernels.h.
 *
 * The wave-systolic aligner (brx_align.h) spends ~1000 cycles of latency on every column of a read and
 * keeps 20-30 of its 64 lanes busy; 64 reads per wave, each lane walking its own band of 32-row blocks,
 * need ~13x fewer wave-cycles per read for the common band widths.  Per lane:
 *   query    2-bit planes {lo, hi} per 32-row block, packed by k_fin_join, a ring of WR blocks in LDS
 *   target   a 64-byte ring in LDS, 32 bytes refilled every 32 columns
 *   state    {Pv, Mv} of the blocks inside the band, a ring of WR blocks in LDS
 *   store    {Pv, Ph} per block and column into the read's own traceback region, [column][block - first
 *            block of that column] -- each lane streams contiguous bytes
 * All global loads happen in one refill phase every 32 columns, so the in-order vmcnt wait behind the
 * traceback stores is paid once per ~1000 block updates.  Traceback: canonical (up, left, diagonal),
 * per lane, 8 columns of band words prefetched per memory round trip.
 *
 * Classes (RS.klass, set by k_fin_join): BRX_KL_LANE32 / BRX_KL_LANE64 = pure ACGT pairs whose band spans
 * at most 30 / 62 blocks; everything else keeps the wave-systolic kernels (klass = words per lane).
 *
 * Measured (profiles/README.md, r01c): 8x fewer wave-cycles per read than the wave-systolic kernel, but
 * 36 KB of LDS per wave admit only 4 waves per CU against 20 for k_fin_align<1,1,1>, and a batch of 16 k
 * reads is only 256 such waves -- the launch becomes a handful of very long waves.  It is therefore OFF by
 * default (BRX_FIN_LANE=1 enables it); it pays once >= ~100 k reads are in flight per GPU.
 */
#ifndef BRX_FINLANE_H
#define BRX_FINLANE_H

#define BRX_KL_LANE32 100u
#define BRX_KL_LANE64 101u

__device__ __forceinline__ int brx_band_blocks(const BrxGeom &g) { return (g.dhi - g.dlo) / 32 + 2; }
/* traceback units (8 bytes) of a lane-aligned read: (columns + 2) rows of band_blocks words */
__host__ __device__ inline uint64_t brx_lane_units(int T, int band_blocks) { return (uint64_t)(T + 2) * (uint64_t)band_blocks; }

/* -------------------------------------------------------------------------------------------------
 * k_fin_join: one wave per read.  join(new_fragment_bases) -> seq (+ pad), class of the read, planes.
 * ----------------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(64) k_fin_join(BrxDev d, RS *rs, uint32_t *queue, const uint8_t *Fbuf, const uint32_t *repl,
                                                  uint8_t *seqbuf, uint2 *qplanes, int lane_classes) {
    const int lane = lane_id();
    const brx_error_model &em = d.em;
    for (;;) {
        const uint32_t r = wave_pop(queue);
        if (r >= d.n_reads) break;
        const RS s = rs[r];
        if (s.n == 0) continue;
        const uint32_t n = s.n, m = s.m;
        const uint8_t *F = Fbuf + s.F_off;
        const uint32_t *rp = repl + s.F_off;
        uint8_t *seq = seqbuf + s.seq_off;
        wave_join(em, F, rp, 0, n, seq, nullptr);
        for (uint32_t x = lane; x < 16; x += 64) seq[m + x] = 0xFE;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        bool odd = false;
        for (uint32_t x = lane; x < n; x += 64) odd |= F[x] > 3;
        for (uint32_t x = lane; x < m; x += 64) odd |= seq[x] > 3;
        const BrxGeom g = brx_make_geom((int)m, (int)n, (int)s.ub);
        uint32_t klass = g.G ? (uint32_t)g.G : 64u;
        if (lane_classes && __ballot(odd) == 0ull && g.G != 0 && m > 0 && n > 0) {
            const int bb = brx_band_blocks(g);
            if (bb <= 30) klass = BRX_KL_LANE32;
            else if (bb <= 62) klass = BRX_KL_LANE64;
        }
        if (klass >= BRX_KL_LANE32) {
            uint2 *qp = qplanes + (s.seq_off >> 4);
            for (uint32_t p = 0; 64 * p < m; ++p) {
                const uint32_t x = 64 * p + lane;
                const uint32_t c = x < m ? seq[x] : 0u;
                const unsigned long long lo = __ballot(c & 1u), hi = __ballot(c & 2u);
                if (lane < 2 && 64 * p + 32 * lane < m) qp[2 * p + lane] = make_uint2((uint32_t)(lo >> (32 * lane)), (uint32_t)(hi >> (32 * lane)));
            }
        }
        if (lane == 0) rs[r].klass = klass;
    }
}

/* reads of [b, e) of the processing order whose class is `want`, in order; one wave */
__global__ void __launch_bounds__(64) k_class_list(const RS *rs, const uint32_t *order, uint32_t b, uint32_t e, uint32_t want,
                                                    uint32_t *list, uint32_t *count) {
    const int lane = lane_id();
    uint32_t run = 0;
    for (uint32_t base = b; base < e; base += 64) {
        const uint32_t i = base + lane;
        const uint32_t r = i < e ? order[i] : 0u;
        const uint32_t hit = (i < e && rs[r].n != 0 && rs[r].klass == want) ? 1u : 0u;
        const uint32_t inc = wave_incl_scan(hit);
        if (hit) list[run + inc - 1] = r;
        run += wave_bcast_u32(inc, 63);
    }
    if (lane == 0) *count = run;
}

/* -------------------------------------------------------------------------------------------------
 * k_fin_lane<WR>: one read per lane, 64 consecutive reads of a class list per wave
 * ----------------------------------------------------------------------------------------------- */
template <int WR>
__global__ void __launch_bounds__(64) k_fin_lane(RS *rs, const uint32_t *list, const uint32_t *count_ptr, uint32_t *queue,
                                                  const uint8_t *Fbuf, const uint8_t *seqbuf, const uint2 *qplanes,
                                                  uint8_t *opsbuf, uint8_t *tb_base, uint64_t *clk) {
    __shared__ uint2 pl[WR][64];            /* query planes {lo, hi} of block b at [b & (WR-1)]      */
    __shared__ uint2 stt[WR][64];           /* {Pv, Mv} of block b at [b & (WR-1)]                   */
    __shared__ uint32_t tring[16][64];      /* target bytes [4w, 4w+4) of the lane at [w & 15]       */
    const int lane = lane_id();
    const uint32_t count = uni(*count_ptr);
    for (;;) {
        const uint32_t g0 = wave_pop(queue) * 64u;
        if (g0 >= count) break;
        const uint64_t t_begin = __builtin_amdgcn_s_memtime();
        const uint32_t idx = g0 + (uint32_t)lane;
        const bool valid = idx < count;
        const uint32_t r = valid ? list[idx] : 0u;
        RS s;
        if (valid) s = rs[r];
        const int Q = valid ? (int)s.m : 0, T = valid ? (int)s.n : 0, kb = valid ? (int)s.ub : 0;
        const uint8_t *Fp = Fbuf + (valid ? s.F_off : 0);
        const uint8_t *seq = seqbuf + (valid ? s.seq_off : 0);
        const uint2 *qp = qplanes + (valid ? (s.seq_off >> 4) : 0);
        uint2 *tb = reinterpret_cast<uint2 *>(tb_base + (valid ? s.tb_off : 0));
        uint8_t *ops_end = opsbuf + (valid ? s.ops_off + (uint64_t)T + (uint64_t)Q : 0);
        const BrxGeom g = brx_make_geom(Q > 0 ? Q : 1, T > 0 ? T : 1, kb);
        const int NS = (Q + 31) >> 5;
        const uint32_t lastmask = (Q & 31) ? ((1u << (Q & 31)) - 1u) : 0xFFFFFFFFu;
        const int Wl = valid ? brx_band_blocks(g) : 0;                /* this read's band height = its traceback row stride */
        const int Wb = (int)wave_max_u32((uint32_t)Wl);
        const int Tmax = (int)wave_max_u32((uint32_t)T);

        /* ---- forward ---- */
        int s_hi = -1, loaded_hi = -1;
        for (int j = 1; j <= Tmax; ++j) {
            const bool act = j <= T;
            if ((j & 31) == 1) {
                /* refill phase: the only global loads of the forward pass.  Target bytes of columns j..j+31,
                   query planes of every block the band can reach before the next phase. */
                if (act) {
                    const uint4 *src = reinterpret_cast<const uint4 *>(Fp + (j - 1));
                    const uint4 a = src[0], b4 = (j - 1 + 16 < T + 16) ? src[1] : make_uint4(0u, 0u, 0u, 0u);
                    const int w0 = ((j - 1) >> 2) & 15;
                    tring[w0 + 0][lane] = a.x; tring[w0 + 1][lane] = a.y; tring[w0 + 2][lane] = a.z; tring[w0 + 3][lane] = a.w;
                    tring[w0 + 4][lane] = b4.x; tring[w0 + 5][lane] = b4.y; tring[w0 + 6][lane] = b4.z; tring[w0 + 7][lane] = b4.w;
                }
                int want_hi = (j + 31 + g.dhi - 1) >> 5;
                if (want_hi > NS - 1) want_hi = NS - 1;
                if (!act) want_hi = loaded_hi;
                while (__ballot(loaded_hi < want_hi) != 0ull) {
                    if (loaded_hi < want_hi) { loaded_hi += 1; pl[loaded_hi & (WR - 1)][lane] = qp[loaded_hi]; }
                }
            }
            /* blocks entering the band at this column: cells below the band grow by +1 per row */
            int new_hi = (j + g.dhi - 1) >> 5;
            if (new_hi > NS - 1) new_hi = NS - 1;
            if (!act) new_hi = s_hi;
            while (__ballot(s_hi < new_hi) != 0ull) {
                if (s_hi < new_hi) { s_hi += 1; stt[s_hi & (WR - 1)][lane] = make_uint2(0xFFFFFFFFu, 0u); }
            }
            int s_lo = (j + g.dlo - 1) >> 5;
            if (s_lo < 0) s_lo = 0;
            const uint32_t c = (tring[((j - 1) >> 2) & 15][lane] >> (8 * ((j - 1) & 3))) & 3u;
            const uint32_t m0 = 0u - (c & 1u), m1 = 0u - (c >> 1);
            uint32_t hp = 1u, hm = 0u;
            uint2 *row = tb + (size_t)j * (size_t)Wl;
            for (int x = 0; x < Wb; ++x) {
                const int sb = s_lo + x;
                const bool on = act && sb <= s_hi;
                const int slot = (on ? sb : 0) & (WR - 1);
                const uint2 st = stt[slot][lane], p = pl[slot][lane];
                uint32_t pv = st.x, mv = st.y;
                uint32_t Eq = ~((p.x ^ m0) | (p.y ^ m1));
                if (sb == NS - 1) Eq &= lastmask;
                const uint32_t Xv = Eq | mv;
                const uint32_t Eq2 = Eq | hm;
                const uint32_t Xh = (((Eq2 & pv) + pv) ^ pv) | Eq2;
                const uint32_t Ph = mv | ~(Xh | pv);
                const uint32_t Mh = pv & Xh;
                const uint32_t PhS = (Ph << 1) | hp;
                const uint32_t MhS = (Mh << 1) | hm;
                pv = MhS | ~(Xv | PhS);
                mv = PhS & Xv;
                if (on) {
                    stt[slot][lane] = make_uint2(pv, mv);
                    row[x] = make_uint2(pv, Ph);
                    hp = Ph >> 31; hm = Mh >> 31;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        const uint64_t t_fwd = __builtin_amdgcn_s_memtime();

        /* ---- traceback, canonical (up, left, diagonal), 8 columns fetched per round trip ---- */
        int i = Q, j = T;
        uint32_t pos = 0, nmatch = 0;
        bool ok = valid;
        bool go = valid && i > 0 && j > 0;
        while (__ballot(go) != 0ull) {
            const int s0 = go ? ((i - 1) >> 5) : 0;
            const int jst = j;
            uint2 A[8], Bv[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int col = jst - x;
                A[x] = make_uint2(0u, 0u); Bv[x] = make_uint2(0u, 0u);
                if (go && col >= 1) {
                    int lo = (col + g.dlo - 1) >> 5; if (lo < 0) lo = 0;
                    int hi = (col + g.dhi - 1) >> 5; if (hi > NS - 1) hi = NS - 1;
                    const uint2 *rw = tb + (size_t)col * (size_t)Wl;
                    if (s0 >= lo && s0 <= hi) A[x] = rw[s0 - lo];
                    if (s0 - 1 >= lo && s0 - 1 <= hi) Bv[x] = rw[s0 - 1 - lo];
                }
            }
            bool walk = go;
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                for (int guard = 0; guard < 72; ++guard) {
                    const bool here = walk && i > 0 && j == jst - x && j > 0;
                    if (__ballot(here) == 0ull) break;
                    if (here) {
                        const int sb = (i - 1) >> 5;
                        if (sb != s0 && sb != s0 - 1) walk = false;                 /* left the two fetched blocks: refetch */
                        else {
                            int lo = (j + g.dlo - 1) >> 5; if (lo < 0) lo = 0;
                            int hi = (j + g.dhi - 1) >> 5; if (hi > NS - 1) hi = NS - 1;
                            if (sb < lo || sb > hi) { ok = false; walk = false; go = false; }
                            else {
                                const uint32_t sel = 0u - (uint32_t)(sb == s0);
                                const uint32_t vx = (A[x].x & sel) | (Bv[x].x & ~sel), vy = (A[x].y & sel) | (Bv[x].y & ~sel);
                                const int bit = (i - 1) & 31;
                                uint8_t op;
                                if ((vx >> bit) & 1u) { op = BRX_OP_I; i -= 1; }
                                else if ((vy >> bit) & 1u) { op = BRX_OP_D; j -= 1; }
                                else {
                                    const bool eq = seq[i - 1] == Fp[j - 1];
                                    nmatch += (uint32_t)eq;
                                    op = eq ? BRX_OP_EQ : BRX_OP_X;
                                    i -= 1; j -= 1;
                                }
                                ops_end[-(long)pos - 1] = op;
                                pos += 1;
                            }
                        }
                    }
                }
            }
            go = go && ok && i > 0 && j > 0;
        }
        if (valid) {
            if (ok) {
                for (int x = 0; x < i; ++x) ops_end[-(long)(pos + (uint32_t)x) - 1] = BRX_OP_I;
                for (int x = 0; x < j; ++x) ops_end[-(long)(pos + (uint32_t)x) - 1] = BRX_OP_D;
                pos += (uint32_t)(i + j);
                if ((pos - nmatch) > (uint32_t)kb) ok = false;
            }
            RS *o = &rs[r];
            o->status = s.status | (ok ? 0u : BRX_RS_BAND);
            o->n_cols = pos; o->n_match = nmatch;
            uint64_t *ck = clk + (uint64_t)r * 8;
            const uint64_t t_end = __builtin_amdgcn_s_memtime();
            ck[3] = (t_end - t_begin) / 64; ck[4] = (t_fwd - t_begin) / 64; ck[5] = (t_end - t_fwd) / 64; ck[7] = WR == 32 ? BRX_KL_LANE32 : BRX_KL_LANE64;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
    }
}

#endif /* BRX_FINLANE_H */

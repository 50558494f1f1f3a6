/*
 * brx_wg_align.h -- the forward pass of one WIDE-band alignment on a workgroup of W waves (BRX_FIN_WG=1; the default final
 * stage runs these reads on one wave with 4-16 words per lane, brx_align.h).
 *
 * Why: a read that is long and inaccurate has a band of thousands of rows.  On one wave a lane then owns 4-16 words and a
 * column costs ~100-450 dependent instructions, for 50 000 columns in a row: k_fin_align<4,4,4> and <16,8,...> run 31 and
 * 75 ms alone on the chip at 1-5 % of its issue rate and set the length of the final stage.  Here the 64 W lanes of a
 * workgroup form ONE systolic array with one word per lane and the one-word schedule of brx_align_forward_k4 (superblock s
 * handles columns 4 (tau - s) + 1 .. + 4 in trip tau; ~108 instructions per trip per wave): the carry crosses a wave
 * boundary through one LDS word per wave and trip, the target ring is one LDS window of 4 x 64 W words refilled by all lanes,
 * and a workgroup barrier per trip keeps the waves in step.  Same cell recurrence, same store layout as a one-word band on one
 * wave (geometry: brx_make_geom with lanes = 64 W), so brx_align_traceback reads the store unchanged (wave 0 runs it).
 */
#ifndef BRX_WG_ALIGN_H
#define BRX_WG_ALIGN_H

#define BRX_WGA_MAXW 16

__shared__ uint32_t brx_wga_ring[4 * 64 * BRX_WGA_MAXW];       /* 4 chunks of 64 W target words */
__shared__ uint32_t brx_wga_carry[2][BRX_WGA_MAXW];             /* carry of every wave's last lane, by trip parity */
__shared__ uint32_t brx_wga_odd[4][BRX_WGA_MAXW];               /* ring quarter x wave: a symbol other than A/C/G/T in its words */

/* brx_make_geom for a systolic array of `lanes` lanes (64 W for a workgroup of W waves): a band of up to lanes - 8
   superblocks keeps one word per lane.  (A copy, so that the one-wave geometry function compiles as it always did.) */
__host__ __device__ inline BrxGeom brx_make_geom_lanes(int Q, int T, int k, int hmul, int lanes) {
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
    while (G <= BRX_GEOM_MAXG && (long long)bw > (long long)(lanes - 8) * 32ll * G) G *= 2;
    if (G > BRX_GEOM_MAXG) { g.G = 0; g.R = 0; g.NS = 0; g.NW = 0; g.WSp = 0; g.K = 1; g.t_end = 0; return g; }
    g.G = G; g.R = 32 * G;
    g.NW = (Q + 31) / 32;
    g.NS = (Q + g.R - 1) / g.R;
    g.WSp = (bw + g.R - 2) / (g.R + 1) + 2;
    if (g.WSp > g.NS) g.WSp = g.NS;
    if (g.WSp < 1) g.WSp = 1;
    g.K = G == 1 ? 4 : 1;
    g.t_end = (G == 1 ? (T + 3) / 4 * 4 : T) + g.K * (g.NS - 1);   /* K = 4: whole trips, the last one may run past column T */
    g.H = BRX_H_ALL; g.slope = 0;
    if (hmul != 0 && G <= 16 && T > 0 && (uint64_t)Q < ((uint64_t)T << 11)) {
        const int H = hmul > 0 ? hmul * (int)brx_isqrt((uint32_t)k) + 24 : 8;
        const int slots = (2 * H + g.R - 1) / g.R + 1;      /* superblocks that can meet the window in one store row */
        if (slots < g.WSp) { g.WSp = slots; g.H = H; g.slope = (uint32_t)(((uint64_t)Q << 20) / (uint64_t)T); }
    }
    return g;
}

/* does the workgroup aligner take this pair?  (uniform inputs; the store must hold the one-word geometry) */
__host__ __device__ inline bool brx_wg_eligible(uint32_t m, uint32_t n, uint32_t ub, int hmul, int lanes, uint64_t cap_units, BrxGeom *out) {
    if (m == 0 || n == 0) return false;
    const BrxGeom g = brx_make_geom_lanes((int)m, (int)n, (int)ub, hmul, lanes);
    if (out) *out = g;
    return g.G == 1 && brx_align_units(g) <= cap_units;
}

/* All 64 W threads of the workgroup call (uniform arguments). */
template <int W>
__device__ inline void brx_align_forward_wg(const uint8_t *__restrict__ Qs, const uint8_t *__restrict__ Ts, const BrxGeom g, uint2 *__restrict__ tb) {
    constexpr int LANES = 64 * W;
    constexpr int RWORDS = 4 * LANES;
    constexpr int NEVER = 0x7FFFFFFF;
    constexpr int K = 4;
    const int L = (int)threadIdx.x;
    const int lane = L & 63, wave = L >> 6;
    const uint64_t tb_addr = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)tb >> 32)) << 32) |
                             (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint64_t)tb);
    int s = L;
    uint32_t slot8 = 0;
    int tf = NEVER, tl = NEVER;
    uint32_t tspan = 0;
    const uint32_t keep_lim = (uint32_t)(2 * g.H + g.R - 1);
    int keep_base = 0;
    int64_t acc = 0;
    const int64_t acc_step = (int64_t)K * (int64_t)g.slope;
    auto window = [&](int tau_now) {
        tf = NEVER; tl = NEVER; tspan = 0;
        if (s < g.NS) {
            const int jf = brx_jfirst(g, s), jl = brx_jlast(g, s);
            slot8 = 8u * (uint32_t)(s % g.WSp);
            tl = s + (jl - 1) / K;
            if (jl >= jf) { tf = s + (jf - 1) / K; tspan = (uint32_t)(tl - tf); }
        }
        keep_base = g.R * s + g.H + g.R - 1;
        acc = (int64_t)(K * (tau_now - s) + 2) * (int64_t)g.slope;
    };
    window(0);
    uint32_t Pv = 0xFFFFFFFFu, Mv = 0;
    BrxQPlanes qp = {0u, 0u, 0u, 0u};
    uint32_t carry = 0xF0u;

    /* chunk c = target words [LANES c, LANES (c + 1)), in ring quarter c & 3 */
    auto fetch_chunk = [&](int c) -> uint32_t {
        const long long idx = 4ll * LANES * c + 4ll * L;
        return (idx + 4 <= (long long)g.T + 16) ? *reinterpret_cast<const uint32_t *>(Ts + idx) : 0xFEFEFEFEu;
    };
    auto chunk_odd = [&](int c, uint32_t v) -> uint32_t {
        const long long idx = 4ll * LANES * c + 4ll * L;
        bool o = false;
#pragma unroll
        for (int b = 0; b < 4; ++b) o |= idx + b < (long long)g.T && ((v >> (8 * b)) & 0xFFu) > 3u;
        return __ballot(o) != 0ull ? 1u : 0u;
    };
    auto gather_odd = [&]() -> uint32_t {                    /* any flagged quarter, over all waves */
        uint32_t v = 0;
        if (lane < 4 * W) v = brx_wga_odd[lane / W][lane % W];
        return __ballot(v != 0u) != 0ull ? 1u : 0u;
    };
    uint32_t pending = 0;
    if (lane == 0) { brx_wga_odd[3][wave] = 0; brx_wga_carry[0][wave] = 0xF0u; brx_wga_carry[1][wave] = 0xF0u; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pending = fetch_chunk(c);
        brx_wga_ring[(c & 3) * LANES + L] = pending;
        const uint32_t o = chunk_odd(c, pending);
        if (lane == 0) brx_wga_odd[c][wave] = o;
    }
    __syncthreads();
    uint32_t odd = gather_odd();
    bool refilled = false;
    int s_top = 0;
    int tl_top = (brx_jlast(g, 0) - 1) / K;
    int next_entry = brx_wave_min(tf), next_hop = brx_wave_min(tl);
    const int tau_end = (g.NS - 1) + (g.T - 1) / K;
    const size_t wsp = (size_t)g.WSp;
    BRX_GLOBAL char *row0 = (BRX_GLOBAL char *)((BRX_GLOBAL uint64_t *)tb_addr + wsp);
    BRX_GLOBAL char *row1 = row0 + 8 * wsp, *row2 = row0 + 16 * wsp, *row3 = row0 + 24 * wsp;
    const size_t trip_bytes = 8 * (size_t)K * wsp;
    uint32_t wnext = brx_wga_ring[(uint32_t)(0 - s) & (RWORDS - 1)];
    for (int tau = 0; tau <= tau_end; ++tau, row0 += trip_bytes, row1 += trip_bytes, row2 += trip_bytes, row3 += trip_bytes, acc += acc_step) {
        /* ---- refill of the target window (the same decisions in every wave: they depend on uniform values only) ---- */
        while (__builtin_expect(s_top < g.NS - 1 && tau > tl_top, 0)) { s_top += 1; tl_top = s_top + (brx_jlast(g, s_top) - 1) / K; }
        const int fq = tau - s_top;                  /* newest column group in use */
        if (__builtin_expect((fq & (LANES / 4 - 1)) == 0 && fq > 0, 0)) {
            /* the band spans fewer than LANES - 2 superblocks: when the front enters chunk m (fq = LANES m) chunk m - 2 is dead
               and chunk m + 2 takes its quarter; its load was issued a quarter chunk earlier */
            const int ph = (fq / (LANES / 4)) & 3;
            if (ph == 3) pending = fetch_chunk(fq / LANES + 3);
            else if (ph == 0) {
                const int c = fq / LANES + 2;
                brx_wga_ring[(c & 3) * LANES + L] = pending;
                const uint32_t o = chunk_odd(c, pending);
                if (lane == 0) brx_wga_odd[c & 3][wave] = o;
                refilled = true;
            }
        }
        if (__builtin_expect(tau == next_entry, 0)) {
            if (tau == tf) {
                Pv = 0xFFFFFFFFu; Mv = 0;
                qp = brx_query_planes(Qs, s, g.Q);
            }
            next_entry = brx_wave_min(tf > tau ? tf : NEVER);
        }
        /* ---- the carry of the lane above: by DPP inside the wave, through LDS from the wave before ---- */
        uint32_t nb = (uint32_t)brx_from_lane_above((int)carry);
        if (lane == 0) nb = brx_wga_carry[(tau + 1) & 1][(wave + W - 1) % W];
        const uint32_t w = wnext;
        const bool act = (uint32_t)(tau - tf) <= tspan;
        const bool keep = (uint32_t)(keep_base - (int)(uint32_t)((uint64_t)acc >> 20)) <= keep_lim;
        bool rare = false;
        if (__builtin_expect(odd != 0u, 0)) {
            bool lr = false;
#pragma unroll
            for (int c = 0; c < K; ++c) lr |= ((w >> (8 * c)) & 0xFFu) > 3u;
            rare = __ballot(lr && act) != 0ull;                  /* per wave: each takes the path its own lanes need */
        }
        uint32_t P = Pv, M = Mv, accP = 0, accM = 0;
        if (__builtin_expect(rare, 0)) {
#pragma unroll 1
            for (int c = 0; c < K; ++c) {
                const uint32_t hm = (nb >> (3 - c)) & 1u, hp = (nb >> (7 - c)) & 1u;
                const uint32_t ch = (w >> (8 * c)) & 0xFFu;
                uint32_t Eq = brx_eq_acgt(qp, 0u - (ch & 1u), 0u - ((ch >> 1) & 1u));
                if (ch == 4u) Eq = qp.n;
                if (act && ch > 4u) {
                    uint32_t mq = 0;
#pragma unroll 1
                    for (int rr = 0; rr < 32; ++rr) { const int qi = 32 * s + rr; if (qi < g.Q && Qs[qi] == ch) mq |= 1u << rr; }
                    Eq = mq;
                }
                const uint32_t Xv = Eq | M;
                const uint32_t Eq2 = Eq | hm;
                const uint32_t Xh = (((Eq2 & P) + P) ^ P) | Eq2;
                const uint32_t Ph = M | ~(Xh | P);
                const uint32_t Mh = P & Xh;
                const uint32_t PhS = (Ph << 1) | hp;
                const uint32_t MhS = (Mh << 1) | hm;
                P = MhS | ~(Xv | PhS);
                M = PhS & Xv;
                if (act && keep) *(BRX_GLOBAL uint64_t *)(row0 + 8 * (size_t)c * wsp + slot8) = ((uint64_t)Ph << 32) | (uint64_t)P;
                accP = (accP << 1) | (Ph >> 31);
                accM = (accM << 1) | (Mh >> 31);
            }
        } else {
            uint32_t pvs[K], phs[K];
#pragma unroll
            for (int c = 0; c < K; ++c) {
                const uint32_t hm = (nb >> (3 - c)) & 1u, hp = (nb >> (7 - c)) & 1u;
                const uint32_t Eq = brx_eq_acgt(qp, brx_bit_mask(w, 8 * c), brx_bit_mask(w, 8 * c + 1));
                const uint32_t Xv = Eq | M;
                const uint32_t Eq2 = Eq | hm;
                const uint32_t Xh = (((Eq2 & P) + P) ^ P) | Eq2;
                const uint32_t Ph = M | ~(Xh | P);
                const uint32_t Mh = P & Xh;
                const uint32_t PhS = (Ph << 1) | hp;
                const uint32_t MhS = (Mh << 1) | hm;
                P = MhS | ~(Xv | PhS);
                M = PhS & Xv;
                pvs[c] = P; phs[c] = Ph;
                accP = __builtin_amdgcn_alignbit(accP, Ph, 31);
                accM = __builtin_amdgcn_alignbit(accM, Mh, 31);
            }
            if (act && keep) {
                *(BRX_GLOBAL uint64_t *)(row0 + slot8) = ((uint64_t)phs[0] << 32) | (uint64_t)pvs[0];
                *(BRX_GLOBAL uint64_t *)(row1 + slot8) = ((uint64_t)phs[1] << 32) | (uint64_t)pvs[1];
                *(BRX_GLOBAL uint64_t *)(row2 + slot8) = ((uint64_t)phs[2] << 32) | (uint64_t)pvs[2];
                *(BRX_GLOBAL uint64_t *)(row3 + slot8) = ((uint64_t)phs[3] << 32) | (uint64_t)pvs[3];
            }
        }
        Pv = act ? P : Pv;
        Mv = act ? M : Mv;
        carry = act ? ((accP << 4) | accM) : 0xF0u;
        if (lane == 63) brx_wga_carry[tau & 1][wave] = carry;
        if (__builtin_expect(tau == next_hop, 0)) {
            if (tau >= tl) { s += LANES; window(tau); }
            next_hop = brx_wave_min(tl);
            next_entry = brx_wave_min(tf > tau ? tf : NEVER);
        }
        __syncthreads();                                 /* this trip's carries and ring words are visible to the next trip */
        if (refilled) { odd = gather_odd(); refilled = false; }
        wnext = brx_wga_ring[(uint32_t)(tau + 1 - s) & (RWORDS - 1)];
    }
}

#endif /* BRX_WG_ALIGN_H */

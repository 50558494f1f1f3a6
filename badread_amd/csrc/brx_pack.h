/*
 * brx_pack.h -- eight window alignments per wavefront ("packed" banded Myers), for the identity checks of the
 * mutate loop (/root/reference/badread/simulate.py:325-346: edlib.align of a 1000-base window of the fragment
 * against its mutated version every 25 changes).
 *
 * Why: a window pair with ~5 % errors has a band of 2-5 superblocks of 32 rows, so the wave-systolic aligner of
 * brx_align.h keeps 3-6 of its 64 lanes busy -- and the path is bound by wave-level VALU issue (tools/native/
 * valu_bench.hip: ~6.5e11 integer wave-instructions/s on the chip, of which the round-1 pipeline used 43 %).  Here a
 * wave is cut into 8 groups of 8 lanes; group g aligns window g with the same schedule as brx_align_forward_k4
 * (lane p of the group owns superblocks p, p+8, ...; four columns per trip; superblock s handles columns 4(tau-s)+1..+4
 * in trip tau; carries travel to the next lane of the GROUP by DPP row rotation), so one instruction stream does the
 * work of eight.  Inputs are 2-bit planes in LDS (query rows / target columns: 1 bit-plane word per 32 symbols), so the
 * loop has no global loads at all and the equality mask of a column is three bit operations.
 *
 * Same results as brx_wave_align by construction: same band (brx_make_geom), same cell recurrence, same traceback
 * store layout and the same canonical traceback (up/'I', left/'D', diagonal) -- only the lane mapping differs.
 * Windows it cannot take (symbols outside ACGT, bands wider than BRX_PACK_BW_MAX diagonals, more than
 * BRX_PACK_TMAX columns) stay with the wave aligner.
 */
#ifndef BRX_PACK_H
#define BRX_PACK_H

#define BRX_PACK_GL 8                       /* lanes per window                                                  */
#define BRX_PACK_NG 8                       /* windows per wave                                                  */
#define BRX_PACK_QW 32                      /* query plane words: windows of up to 1024 rows                    */
#define BRX_PACK_TMAX 1536                  /* target columns                                                    */
#define BRX_PACK_TW (BRX_PACK_TMAX / 32 + 1)
#define BRX_PACK_BW_MAX 216                 /* band diagonals: a lane must be done with superblock s before s + 8 enters
                                               (36 GL - 32 diagonals of slack from the four-column skew), with margin */
#define BRX_PACK_TB_UNITS 12288u            /* uint2 units of traceback store per window slot (96 KB)            */

struct BrxPackWin {                         /* one window slot in LDS */
    uint32_t qlo[BRX_PACK_QW], qhi[BRX_PACK_QW];
    uint32_t tlo[BRX_PACK_TW], thi[BRX_PACK_TW];
    uint32_t Q, T, k;                       /* rows, columns, proven edit bound; Q == 0: slot empty              */
    uint32_t ncols, nmatch, ok;             /* results                                                           */
};

/* can the packed aligner take this pair?  (wave-uniform inputs) */
__device__ __forceinline__ bool brx_pack_eligible(uint32_t Q, uint32_t T, uint32_t k, bool odd) {
    if (odd || Q == 0 || T == 0 || Q > 32u * BRX_PACK_QW || T > BRX_PACK_TMAX) return false;
    const BrxGeom g = brx_make_geom((int)Q, (int)T, (int)k);
    return g.G == 1 && (g.dhi - g.dlo + 1) <= BRX_PACK_BW_MAX && brx_tb_units(g) <= BRX_PACK_TB_UNITS;
}

/* lane i receives the value of lane i - 1 of its 8-lane group (lane 0 of the group: lane 7 of the group).
   DPP rotates rows of 16 lanes: row_ror:1 is right for 14 of the 16 lanes, row_ror:9 for the two group leaders. */
__device__ __forceinline__ uint32_t brx_group_ror1(uint32_t v) {
    const uint32_t a = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121 /* row_ror:1 */, 0xF, 0xF, false);
    const uint32_t b = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x129 /* row_ror:9 */, 0xF, 0xF, false);
    return (threadIdx.x & 7u) ? a : b;
}

/* All 64 lanes call.  slots[g] holds window g (Q == 0: empty).  tb: BRX_PACK_NG x BRX_PACK_TB_UNITS uint2 of global
   scratch owned by this wave.  Results go to slots[g].ncols / nmatch / ok. */
__device__ inline void brx_pack_align(BrxPackWin *slots, uint2 *__restrict__ tb_wave) {
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, p = lane & 7;
    BrxPackWin &W = slots[grp];
    const int Q = (int)W.Q, T = (int)W.T, kb = (int)W.k;
    const bool have = Q > 0 && T > 0;
    const BrxGeom g = brx_make_geom(have ? Q : 1, have ? T : 1, kb);
    uint2 *tb = tb_wave + (size_t)grp * BRX_PACK_TB_UNITS;
    constexpr int NEVER = 0x7FFFFFFF, JNEVER = 0x3FFFFFFF, K = 4;

    /* ---- forward ---- */
    int s = p;
    int jf = JNEVER, jl = -1, slot = 0, tf = NEVER, tl = NEVER;
    uint32_t qlo = 0, qhi = 0, qmask = 0;
    auto enter = [&]() {                               /* geometry + planes of superblock s for this lane */
        if (have && s < g.NS) {
            jf = brx_jfirst(g, s); jl = brx_jlast(g, s); slot = s % g.WSp;
            tf = s + (jf - 1) / K; tl = s + (jl - 1) / K;
            if (jl < jf) tf = NEVER;
            qlo = W.qlo[s]; qhi = W.qhi[s];
            qmask = (s == g.NS - 1 && (Q & 31)) ? ((1u << (Q & 31)) - 1u) : 0xFFFFFFFFu;
        } else { jf = JNEVER; jl = -1; tf = NEVER; tl = NEVER; }
    };
    enter();
    uint32_t Pv = 0xFFFFFFFFu, Mv = 0, carry = 0;
    int tau_end = have ? (g.NS - 1) + (g.T - 1) / K : -1;
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) { const int o = __shfl_xor(tau_end, dd, 64); tau_end = o > tau_end ? o : tau_end; }
    const int wsp = g.WSp;
    uint2 *dst = tb + (size_t)wsp + (size_t)slot;      /* row 4 tau + 1, this lane's slot */
    for (int tau = 0; tau <= tau_end; ++tau, dst += (size_t)K * (size_t)wsp) {
        const uint32_t nb = brx_group_ror1(carry);
        const int jb = K * (tau - s);                  /* columns jb + 1 .. jb + 4 (0-based target index jb .. jb + 3) */
        uint32_t tlo4 = 0, thi4 = 0;
        {
            int wi = jb >> 5;
            wi = wi < 0 ? 0 : (wi >= BRX_PACK_TW ? BRX_PACK_TW - 1 : wi);
            tlo4 = W.tlo[wi] >> (jb & 31); thi4 = W.thi[wi] >> (jb & 31);
        }
        if (tau == tf) { Pv = 0xFFFFFFFFu; Mv = 0; }  /* cells below the band grow by +1 per row */
        uint32_t out = 0;
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const int j = jb + 1 + c;
            const uint32_t actm = ~(uint32_t)(((j - jf) | (jl - j)) >> 31);     /* all ones iff jf <= j <= jl */
            const uint32_t m0 = 0u - ((tlo4 >> c) & 1u), m1 = 0u - ((thi4 >> c) & 1u);
            const uint32_t Eq = ~((qlo ^ m0) | (qhi ^ m1)) & qmask;
            const uint32_t hin = (nb >> (2 * c)) & 3u;
            const uint32_t hp = (0x9u >> hin) & 1u, hm = (0x2u >> hin) & 1u;
            const uint32_t Xv = Eq | Mv;
            const uint32_t Eq2 = Eq | hm;
            const uint32_t Xh = (((Eq2 & Pv) + Pv) ^ Pv) | Eq2;
            const uint32_t Ph = Mv | ~(Xh | Pv);
            const uint32_t Mh = Pv & Xh;
            const uint32_t PhS = (Ph << 1) | hp;
            const uint32_t MhS = (Mh << 1) | hm;
            const uint32_t pv = MhS | ~(Xv | PhS);
            const uint32_t mv = PhS & Xv;
            if (actm) dst[(size_t)c * (size_t)wsp] = make_uint2(pv, Ph);
            Pv = brx_bfi(actm, pv, Pv);
            Mv = brx_bfi(actm, mv, Mv);
            out |= (((Ph >> 31) + 2u - (Mh >> 31)) & actm) << (2 * c);
        }
        carry = out;
        if (tau >= tl) {                              /* the superblock has left the band: take superblock s + 8 */
            const int oslot = slot;
            s += BRX_PACK_GL;
            enter();
            dst += (ptrdiff_t)slot - (ptrdiff_t)oslot;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);                    /* the stores of this wave are visible to its loads below */

    /* ---- traceback: the group's 8 lanes speculate down the diagonal ---- */
    int i = Q, j = T;
    uint32_t ncols = 0, nmatch = 0;
    bool ok = have, go = have;
    while (__ballot(go) != 0ull) {
        const int ci = i - p, cj = j - p;
        const bool valid = go && ci >= 1 && cj >= 1;
        bool inband = false, up = false, left = false, eq = false;
        if (valid) {
            const int sb = (ci - 1) >> 5;
            inband = cj >= brx_jfirst(g, sb) && cj <= brx_jlast(g, sb);
            if (inband) {
                const uint2 v = tb[(size_t)(cj + K * sb) * (size_t)wsp + (size_t)(sb % wsp)];
                const int bit = (ci - 1) & 31;
                up = (v.x >> bit) & 1u; left = (v.y >> bit) & 1u;
            }
            const int qi = ci - 1, tj = cj - 1;
            const uint32_t qc = ((W.qlo[qi >> 5] >> (qi & 31)) & 1u) | (((W.qhi[qi >> 5] >> (qi & 31)) & 1u) << 1);
            const uint32_t tc = ((W.tlo[tj >> 5] >> (tj & 31)) & 1u) | (((W.thi[tj >> 5] >> (tj & 31)) & 1u) << 1);
            eq = qc == tc;
        }
        const bool diag = valid && inband && !up && !left;
        const uint32_t sh = 8u * (uint32_t)grp;
        const uint32_t dm = (uint32_t)(__ballot(diag) >> sh) & 0xFFu;
        const uint32_t um = (uint32_t)(__ballot(up) >> sh) & 0xFFu;
        const uint32_t vm = (uint32_t)(__ballot(valid && inband) >> sh) & 0xFFu;
        const uint32_t em = (uint32_t)(__ballot(eq) >> sh) & 0xFFu;
        if (go) {
            const int run = __ffs((int)(~dm | 0x100u)) - 1;                    /* 0..8 diagonal moves */
            nmatch += (uint32_t)__popc(em & ((1u << run) - 1u));
            ncols += (uint32_t)run; i -= run; j -= run;
            if (run < BRX_PACK_GL && i > 0 && j > 0) {
                if (!((vm >> run) & 1u)) { ok = false; go = false; }
                else { if ((um >> run) & 1u) i -= 1; else j -= 1; ncols += 1; }
            }
            go = go && i > 0 && j > 0;
        }
    }
    if (have && p == 0) {
        if (ok) ncols += (uint32_t)(i + j);
        if (ok && (ncols - nmatch) > (uint32_t)kb) ok = false;
        W.ncols = ok ? ncols : 0u; W.nmatch = ok ? nmatch : 0u; W.ok = ok ? 1u : 0u;
    }
}

#endif /* BRX_PACK_H */

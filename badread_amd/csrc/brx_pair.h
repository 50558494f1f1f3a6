/*
 * brx_pair.h -- TWO final alignments per wavefront, for reads whose one-word band needs at most 30 of a wave's lanes
 * (BRX_FIN_PAIR=1; the default final stage runs one read per wave, brx_align.h).
 *
 * Why: the forward pass of brx_align_forward_k4 issues ~108 wave-instructions per four-column trip whatever the number
 * of busy lanes, and a read of the one-word class keeps ~bw/36 + 2 of the 64 busy (a superblock of 32 rows lives
 * (bw + 32)/4 trips and a new one enters every 9 trips): 15-27 lanes for the bands of up to 896 diagonals taken here --
 * about 60 % of that class's bases.  Lane l works for read (l & 1) as position l >> 1 of a 32-lane systolic array: the same
 * trip schedule (superblock s handles columns 4 (tau - s) + 1 .. + 4 in trip tau, lane position s mod 32), the carry two
 * lanes up (two DPP wave_ror:1), a target ring per read in LDS.  What was wave-uniform per read -- geometry, store base,
 * ring refill state -- is selected by l & 1; the traceback store of a lane is addressed per lane.  Same cell recurrence,
 * same store layout: brx_align_traceback reads it unchanged, one read after the other on all 64 lanes, so the results are
 * those of brx_wave_align by construction (checked against the oracle through the interpreted kernels,
 * tests/test_emulated_device.py).
 */
#ifndef BRX_PAIR_H
#define BRX_PAIR_H

#define BRX_PAIR_MAX_BW 896                  /* band diagonals a paired read may have: bw / 36 + 2 <= 27 of the 32 positions */

__shared__ uint32_t brx_ring32_pair[2 * (BRX_RING_BYTES / 4)];

__device__ __forceinline__ bool brx_pair_eligible(const BrxGeom &g) {
    return g.G == 1 && g.Q > 0 && g.T > 0 && (g.dhi - g.dlo + 1) <= BRX_PAIR_MAX_BW;
}

/* All 64 lanes call.  Read A on the even lanes, read B on the odd ones (gB.NS == 0: no second read).  Both stores must
   hold brx_tb_units of their geometry. */
__device__ inline void brx_align_forward_k4_pair(const uint8_t *__restrict__ QsA, const uint8_t *__restrict__ TsA, const BrxGeom gA, uint2 *tbA,
                                                 const uint8_t *__restrict__ QsB, const uint8_t *__restrict__ TsB, const BrxGeom gB, uint2 *tbB) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 1, p = lane >> 1;
    constexpr int NEVER = 0x7FFFFFFF;
    constexpr int K = 4;
    constexpr int RW = BRX_RING_BYTES / 4;                          /* ring words per read */
    /* ---- what belongs to the lane's read ---- */
    const BrxGeom g = r ? gB : gA;
    const uint8_t *Qs = r ? QsB : QsA;
    const uint64_t tb_lane = (uint64_t)(r ? tbB : tbA);
    const uint32_t wsp8 = 8u * (uint32_t)g.WSp;                     /* bytes of a store row */
    const int ring_at = r * RW;
    int s = p;
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

    /* ---- the two target rings: refilled by all 64 lanes, one read after the other (wave-uniform code per read) ---- */
    int s_top[2] = {0, 0};
    int tl_top[2] = {gA.NS > 0 ? (brx_jlast(gA, 0) - 1) / K : NEVER, gB.NS > 0 ? (brx_jlast(gB, 0) - 1) / K : NEVER};
    uint32_t odd[2] = {0u, 0u}, pending[2] = {0u, 0u};
    auto fetch_chunk = [&](int x, int c) -> uint32_t {
        const uint8_t *Ts = x ? TsB : TsA;
        const int T = x ? gB.T : gA.T;
        const int idx = 256 * c + 4 * lane;
        return (idx + 4 <= T + 16) ? *reinterpret_cast<const uint32_t *>(Ts + idx) : 0xFEFEFEFEu;
    };
    auto chunk_odd = [&](int x, int c, uint32_t v) -> uint32_t {
        const int T = x ? gB.T : gA.T;
        const int idx = 256 * c + 4 * lane;
        bool o = false;
#pragma unroll
        for (int b = 0; b < 4; ++b) o |= idx + b < T && ((v >> (8 * b)) & 0xFFu) > 3u;
        return __ballot(o) != 0ull ? 1u : 0u;
    };
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        if ((x ? gB.NS : gA.NS) <= 0) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pending[x] = fetch_chunk(x, c);
            brx_ring32_pair[x * RW + (c & 3) * 64 + lane] = pending[x];
            odd[x] |= chunk_odd(x, c, pending[x]) << c;
        }
    }
    int next_entry = brx_wave_min(tf), next_hop = brx_wave_min(tl);
    const int tau_end_a = gA.NS > 0 ? (gA.NS - 1) + (gA.T - 1) / K : -1, tau_end_b = gB.NS > 0 ? (gB.NS - 1) + (gB.T - 1) / K : -1;
    const int tau_end = tau_end_a > tau_end_b ? tau_end_a : tau_end_b;
    uint64_t row_off = wsp8;                                        /* byte offset of store row 4 tau + 1 in the lane's store */
    const uint64_t trip_bytes = (uint64_t)K * wsp8;
    uint32_t wnext = brx_ring32_pair[ring_at + ((uint32_t)(0 - s) & (RW - 1))];
    for (int tau = 0; tau <= tau_end; ++tau, row_off += trip_bytes, acc += acc_step) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const BrxGeom &gx = x ? gB : gA;
            if (gx.NS <= 0) continue;
            while (__builtin_expect(s_top[x] < gx.NS - 1 && tau > tl_top[x], 0)) { s_top[x] += 1; tl_top[x] = s_top[x] + (brx_jlast(gx, s_top[x]) - 1) / K; }
            const int fq = tau - s_top[x];
            if (__builtin_expect((fq & 15) == 0 && fq > 0, 0)) {
                const int ph = (fq >> 4) & 3;
                if (ph == 3) pending[x] = fetch_chunk(x, (fq >> 6) + 3);
                else if (ph == 0) {
                    const int c = (fq >> 6) + 2;
                    brx_ring32_pair[x * RW + (c & 3) * 64 + lane] = pending[x];
                    odd[x] = (odd[x] & ~(1u << (c & 3))) | (chunk_odd(x, c, pending[x]) << (c & 3));
                }
            }
        }
        if (__builtin_expect(tau == next_entry, 0)) {
            if (tau == tf) {
                Pv = 0xFFFFFFFFu; Mv = 0;
                qp = brx_query_planes(Qs, s, g.Q);
            }
            next_entry = brx_wave_min(tf > tau ? tf : NEVER);
        }
        /* the carry of the superblock above: two lanes up (lanes 0 and 1 take those of lanes 62 and 63) */
        const uint32_t nb = (uint32_t)brx_from_lane_above(brx_from_lane_above((int)carry));
        const uint32_t w = wnext;
        const bool act = (uint32_t)(tau - tf) <= tspan;
        const bool keep = (uint32_t)(keep_base - (int)(uint32_t)((uint64_t)acc >> 20)) <= keep_lim;
        bool rare = false;
        if (__builtin_expect((odd[0] | odd[1]) != 0u, 0)) {
            bool lr = false;
#pragma unroll
            for (int c = 0; c < K; ++c) lr |= ((w >> (8 * c)) & 0xFFu) > 3u;
            rare = __ballot(lr && act) != 0ull;
        }
        uint32_t P = Pv, M = Mv, accP = 0, accM = 0;
        BRX_GLOBAL char *dst = (BRX_GLOBAL char *)(tb_lane + row_off + slot8);
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
                if (act && keep) *(BRX_GLOBAL uint64_t *)(dst + (uint64_t)c * wsp8) = ((uint64_t)Ph << 32) | (uint64_t)P;
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
#pragma unroll
                for (int c = 0; c < K; ++c) *(BRX_GLOBAL uint64_t *)(dst + (uint64_t)c * wsp8) = ((uint64_t)phs[c] << 32) | (uint64_t)pvs[c];
            }
        }
        Pv = act ? P : Pv;
        Mv = act ? M : Mv;
        carry = act ? ((accP << 4) | accM) : 0xF0u;
        if (__builtin_expect(tau == next_hop, 0)) {
            if (tau >= tl) { s += 32; window(tau); }
            next_hop = brx_wave_min(tl);
            next_entry = brx_wave_min(tf > tau ? tf : NEVER);
        }
        wnext = brx_ring32_pair[ring_at + ((uint32_t)(tau + 1 - s) & (RW - 1))];
    }
}

#endif /* BRX_PAIR_H */

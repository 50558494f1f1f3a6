/*
 * brx_mutate_wg.h -- the mutate loop of sequence_fragment (/root/reference/badread/simulate.py:272-346) as ONE launch:
 * workgroups of eight waves, one read per wave, window alignments of the workgroup packed into one wave.
 *
 * Round 1 ran the loop as ~66 passes of {k_mutate_seg, k_win_lane, k_win_wave} with host round trips, plus an in-place
 * tail (brx_mutate.h; still there as the BRX_MUTATE_WG=0 route).  Its costs: the lane-per-window kernel needs 2.4-7 ms
 * per pass (one wave per CU, a chain of LDS round trips per column), the in-place alignments of the tail spend a whole
 * wave's instruction stream on 4-6 busy lanes (17 % of all VALU instructions of a batch), the loop state of every read
 * travels through global memory once per pass, and a batch's mutate stage takes 375 ms alone on the GPU.
 *
 * Here a workgroup owns eight reads at a time.  Every wave runs its read's loop (64 proposals per round, survivors
 * applied in iteration order: same code and same draws as k_mutate_seg) until the 25th change asks for an identity
 * check; it then writes the window pair as 2-bit planes into the workgroup's LDS (the pair never touches global
 * memory) and waits at a workgroup barrier.  Wave 0 aligns the (up to) eight parked windows AT ONCE with the packed
 * aligner (brx_pack.h: 8 lanes per window), a second barrier releases the waves, and each resumes its loop with the
 * alignment's result -- the loop state never leaves its registers.  A wave whose read is finished pulls the next
 * (longest-first) read from the queue, so a workgroup always carries eight reads in different stages.  Windows the
 * packed aligner cannot take (symbols outside ACGT in N runs, very wide bands) go through global memory to the wave
 * aligner, run by wave 0 after the packed ones.
 *
 * LDS per workgroup: thr16[4^7] (32 KB: the high halves of the error model's self thresholds -- 93 % of the k-mer draws
 * are rejected by one 16-bit LDS compare), 8 window slots of 2-bit planes (5.3 KB),
 * the wave aligner's target ring (1 KB).  Two workgroups per CU.
 *
 * Results are identical to the sequential loop and to k_mutate_seg: proposals are pure functions of (seed, read,
 * iteration); the alignment result is applied exactly where the in-place alignment was; the packed and the wave aligner
 * produce the same canonical path.
 */
#ifndef BRX_MUTATE_WG_H
#define BRX_MUTATE_WG_H

#define BRX_WG_WAVES 8
#define BRX_WG_THR_ROWS 16384               /* 4^7: k = 7 error models; other models read self_thr from global memory */

/* One pass over the window [a, b) of a read: query planes from F, target planes from join(new_fragment_bases[a:b]),
 * edit bound, non-ACGT flag.  Returns the joined length (may exceed the planes: the caller re-parks through global
 * memory).  The slot's planes must not be in use (the previous alignment of this wave is complete). */
__device__ __forceinline__ uint32_t wave_park_planes(const brx_error_model &em, const uint8_t *F, const uint32_t *repl, uint32_t a, uint32_t b,
                                                     BrxPackWin &W, uint32_t *cost, bool *odd) {
    const int lane = lane_id();
    for (int x = lane; x < BRX_PACK_QW; x += 64) { W.qlo[x] = 0; W.qhi[x] = 0; }
    for (int x = lane; x < BRX_PACK_TW; x += 64) { W.tlo[x] = 0; W.thi[x] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    uint32_t run = 0, c = 0;
    bool o_ = false;
    const bool q_fits = (b - a) <= 32u * BRX_PACK_QW;
    uint32_t it = 0;
    for (uint32_t base = a; base < b; base += 64, ++it) {
        const uint32_t p = base + lane;
        const bool valid = p < b;
        uint32_t w = 0, len = 0;
        uint32_t fb = 0;
        if (valid) { fb = F[p]; w = repl[p]; len = rep_len(w); c += rep_cost(w); o_ |= fb > 3; }
        const unsigned long long lo = __ballot(valid && (fb & 1u)), hi = __ballot(valid && (fb & 2u));
        if (q_fits && lane < 2) {
            W.qlo[2 * it + lane] = (uint32_t)(lo >> (32 * lane));
            W.qhi[2 * it + lane] = (uint32_t)(hi >> (32 * lane));
        }
        const uint32_t inc = wave_incl_scan(len);
        if (valid) {
            const uint32_t o = run + inc - len;
            for (uint32_t x = 0; x < len; ++x) {
                const uint32_t ch = w ? (uint32_t)rep_char(em, w, x) : fb;
                o_ |= ch > 3;
                const uint32_t at = o + x;
                if (at < BRX_PACK_TMAX) {
                    if (ch & 1u) atomicOr(&W.tlo[at >> 5], 1u << (at & 31));
                    if (ch & 2u) atomicOr(&W.thi[at >> 5], 1u << (at & 31));
                }
            }
        }
        run += wave_bcast_u32(inc, 63);
    }
    *cost = wave_sum(c);
    *odd = __ballot(o_) != 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    return run;
}

template <bool PROFILE = false>
__global__ void __launch_bounds__(64 * BRX_WG_WAVES, 4) k_mutate_wg(BrxDev d, RS *rs, MS *msv, const uint32_t *list, uint32_t n_list, uint32_t *queue,
                                                                   uint32_t *legacy_list, uint32_t *legacy_ctr,
                                                                   const uint8_t *Fbuf, uint32_t *repl, uint8_t *winbuf,
                                                                   uint2 *pack_tb, uint8_t *scr_base, uint64_t scr_bytes,
                                                                   uint32_t *flags, uint64_t *clk, uint64_t *phase) {
    __shared__ uint16_t s_thr16[BRX_WG_THR_ROWS];
    __shared__ BrxPackWin s_win[BRX_WG_WAVES];
    __shared__ uint32_t s_fb[BRX_WG_WAVES][4];        /* window for the wave aligner: read, rows, columns, bound (rows == 0: none) */
    __shared__ uint32_t s_any[2];
    const int lane = lane_id();
    const int wid = (int)(threadIdx.x >> 6);
    const brx_error_model &em = d.em;
    const int k = em.k;
    const bool use_thr = em.type == 1 && em.n_rows <= BRX_WG_THR_ROWS;
    if (use_thr) for (uint32_t x = threadIdx.x; x < em.n_rows; x += blockDim.x) s_thr16[x] = (uint16_t)(em.d_self_thr[x] >> 16);
    if (threadIdx.x < 2) s_any[threadIdx.x] = 0;
    if (lane == 0) { s_win[wid].Q = 0; s_fb[wid][1] = 0; }
    __syncthreads();
    uint2 *tb_pack = pack_tb + (size_t)blockIdx.x * (size_t)BRX_PACK_NG * (size_t)BRX_PACK_TB_UNITS;
    uint2 *tb_wave = reinterpret_cast<uint2 *>(scr_base + (uint64_t)blockIdx.x * scr_bytes);

    /* the wave's current read */
    bool have = false;
    uint32_t r = 0, n = 0;
    RS s;
    MS ms;
    uint64_t read = 0, max_i = 0, loop_cap = 0, t_begin = 0;
    const uint8_t *F = nullptr;
    uint32_t *rp = nullptr;
    double target = 0.0, dn = 0.0, need = 0.0;
    uint64_t ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, ph4 = 0, ph_last = 0;
    int ph_cur = 4;

    for (uint32_t round = 0;; ++round) {
        bool parked = false;
        /* ---- phase A: every wave runs its read (and, when that one finishes, the next) up to an identity check ---- */
        for (;;) {
            if (!have) {
                const uint32_t qi = wave_pop(queue);
                if (qi >= n_list) break;
                r = list[qi];
                s = rs[r];
                if (s.n == 0) continue;
                ms = msv[r];                  /* zero for a fresh read; a read parked by the pass pipeline resumes (phase 1) */
                have = true;
                read = d.first_read + r; n = s.n;
                F = Fbuf + s.F_off; rp = repl + s.F_off;
                target = s.target; dn = (double)n;
                max_i = (uint64_t)n - 1 - (uint64_t)k;
                need = dn * (1.0 - target);
                loop_cap = 100ull * (uint64_t)n;
                t_begin = __builtin_amdgcn_s_memtime();
                if constexpr (PROFILE) { ph0 = ph1 = ph2 = ph3 = ph4 = 0; ph_last = t_begin; ph_cur = 4; }
            }
            double errors = 0.0;
            uint64_t loops = 0;
            uint32_t change = 0, nalign = 0;
            bool resume = ms.phase == 1u;
            bool legacy = false;
            if (resume) {
                errors = ms.errors; loops = ms.round_loops; change = ms.change; nalign = ms.nalign;
                const double id = ms.res_ncols ? (double)ms.res_nmatch / (double)ms.res_ncols : 0.0;     /* misc.py:228-240 */
                if (n <= BRX_ALIGN_SIZE) errors = (1.0 - id) * dn;                                       /* simulate.py:333 */
                else {
                    const double est_err = (1.0 - id) * dn;
                    const double weight = (double)BRX_ALIGN_SIZE / dn;
                    errors = est_err * weight + errors * (1.0 - weight);                                 /* simulate.py:344-346 */
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
                uint32_t rep[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) rep[j] = 0;
                bool live = false;
                uint64_t ipos = 0;
                if ((uint32_t)lane < B) {
                    uint32_t w[4];
                    brx_draw4(d.seed, read, BRX_ST_MUT, loops + (uint64_t)lane, w);
                    ipos = brx_mulhi64(((uint64_t)w[1] << 32) | w[0], max_i + 1);
                    uint8_t kmer[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) kmer[j] = j < k ? F[ipos + j] : 0;
                    live = dev_choose_alt(em, kmer, w[2], w[3], rep, use_thr ? s_thr16 : (const uint16_t *)nullptr);
                }
                unsigned long long surv = __ballot(live);
                BRX_PHASE(1);
                int j0 = 0;
                if (resume) { surv &= ~((1ull << ms.surv_lane) - 1ull); j0 = (int)ms.j_next; }
                bool first = resume;
                resume = false;
                /* ---- apply survivors in iteration order ---- */
                while (surv) {
                    const int l = __ffsll((long long)surv) - 1;
                    surv &= surv - 1;
                    const uint64_t i0 = wave_bcast_u64(ipos, l);
                    const double scale = est * brx_sqrt(est);
                    uint32_t wj = 0;
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj) {
                        if (jj < k) { const uint32_t v = wave_bcast_u32(rep[jj], l); wj = (lane == jj) ? v : wj; }
                    }
                    const uint32_t curj = lane < k ? rp[i0 + (uint64_t)lane] : 1u;
                    unsigned long long todo = __ballot(lane < k && wj != 0u && curj == 0u);
                    if (first) todo &= ~((1ull << j0) - 1ull);
                    while (todo) {
                        const int j = __ffsll((long long)todo) - 1;
                        todo &= todo - 1;
                        const uint32_t w = wave_bcast_u32(wj, j);
                        if (lane == j) rp[i0 + (uint64_t)j] = w;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        change += 1;
                        const uint32_t len = (w >> 24) & 0x7Fu;
                        errors += (double)(len < 2 ? 1u : len - 1u) * scale;
                        if (change % BRX_ALIGN_INTERVAL == 0) {
                            /* ---- identity check: park the window in the workgroup's LDS, the loop state in `ms` ---- */
                            BRX_PHASE(2);
                            uint32_t a = 0, b = n;
                            if (n > BRX_ALIGN_SIZE) {
                                uint32_t ww[4];
                                brx_draw4(d.seed, read, BRX_ST_WIN, (uint64_t)nalign, ww);
                                a = (uint32_t)brx_mulhi64(((uint64_t)ww[1] << 32) | ww[0], (uint64_t)n - BRX_ALIGN_SIZE + 1);
                                b = a + BRX_ALIGN_SIZE;
                            }
                            nalign += 1;
                            __builtin_amdgcn_s_waitcnt(0);
                            uint32_t cost = 0;
                            bool odd = false;
                            const uint32_t ql = b - a;
                            uint32_t tl = wave_park_planes(em, F, rp, a, b, s_win[wid], &cost, &odd);
                            const bool packed = brx_pack_eligible(ql, tl, cost, odd);
                            if (!packed) {
                                /* through global memory to the wave aligner (or, if it outgrows its slot, to k_mutate) */
                                uint8_t *qb = winbuf + (uint64_t)r * BRX_WIN_STRIDE, *tbuf = qb + BRX_WIN_Q;
                                tl = wave_park(em, F, rp, a, b, qb, tbuf, BRX_WIN_TMAX, &cost, &odd);
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                                __builtin_amdgcn_s_waitcnt(0);
                                legacy = tl > BRX_WIN_TMAX;
                            }
                            ms.errors = errors; ms.est = est; ms.round_loops = loops; ms.change = change; ms.nalign = nalign;
                            ms.phase = 1u; ms.surv_lane = (uint32_t)l; ms.j_next = (uint32_t)(j + 1);
                            ms.win_a = a; ms.win_b = b; ms.tl = tl; ms.cost = cost; ms.res_ncols = 0; ms.res_nmatch = 0;
                            ms.passes += 1;
                            if (lane == 0) {
                                s_win[wid].Q = packed ? ql : 0u; s_win[wid].T = tl; s_win[wid].k = cost;
                                s_win[wid].ncols = 0; s_win[wid].nmatch = 0; s_win[wid].ok = 0;
                                s_fb[wid][0] = r; s_fb[wid][1] = (packed || legacy) ? 0u : ql; s_fb[wid][2] = tl; s_fb[wid][3] = cost;
                            }
                            parked = true;
                            break;
                        }
                    }
                    if (parked) break;
                    first = false;
                    /* top-of-loop tests of the iteration that follows this survivor */
                    const double est2 = 1.0 - errors / dn;
                    if ((double)change > 0.9 * dn || est2 <= target) { loops += (uint64_t)l + 2; done = true; break; }
                    est = est2;
                }
                if (parked || done) break;
                loops += B;
                if (B < 64) { loops += 1; break; }
            }
            BRX_PHASE(4);
            if (parked && legacy) {
                /* the joined window does not fit a slot: the whole-read kernel starts this read over */
                if (lane == 0) {
                    legacy_list[atomicAdd(legacy_ctr, 1u)] = r;
                    s_win[wid].Q = 0; s_fb[wid][1] = 0;
                }
                parked = false; have = false;
                continue;
            }
            if (parked) break;
            /* epilogue: lengths of the mutated read, trims (simulate.py:348-349), proven distance bound */
            __builtin_amdgcn_s_waitcnt(0);
            uint32_t cost = 0;
            const uint32_t m = wave_join(em, F, rp, 0, n, nullptr, &cost);
            uint32_t st = 0, et = 0;
            if (lane < k) { st = rep_len(rp[lane]); et = rep_len(rp[n - k + lane]); }
            st = wave_sum(st); et = wave_sum(et);
            if constexpr (PROFILE) {
                BRX_PHASE(4);
                if (lane == 0) { uint64_t *pp = phase + (uint64_t)r * 8; pp[0] += ph0; pp[1] += ph1; pp[2] += ph2; pp[3] += ph3; pp[4] += ph4; }
            }
            if (lane == 0) {
                RS *o = &rs[r];
                o->status = s.status | ms.status; o->m = m; o->ub = cost; o->start_trim = st; o->end_trim = et;
                o->loops = (uint32_t)loops; o->changes = change; o->naligns = nalign;
                o->units = 0;                                          /* sized by k_fin_join */
                msv[r].phase = 2u;
                uint64_t *ck = clk + (uint64_t)r * 8;
                ck[0] += __builtin_amdgcn_s_memtime() - t_begin; ck[1] = nalign;
            }
            have = false;
        }
        /* ---- the workgroup's identity checks ---- */
        if (lane == 0) {
            if (parked) s_any[round & 1u] = 1u;
            else { s_win[wid].Q = 0; s_fb[wid][1] = 0; }
        }
        BRX_PHASE(3);
        __syncthreads();
        const bool any = uni(s_any[round & 1u]) != 0u;
        if (!any) break;                                  /* no wave of the workgroup has a read left */
        if (wid == 0) {
            if (lane == 0) s_any[(round + 1u) & 1u] = 0u;
            brx_pack_align(s_win, tb_pack);
            for (int w = 0; w < BRX_WG_WAVES; ++w) {
                const uint32_t ql_ = uni(s_fb[w][1]);
                if (!ql_) continue;
                const uint32_t rr = uni(s_fb[w][0]), tl_ = uni(s_fb[w][2]), cost_ = uni(s_fb[w][3]);
                const uint8_t *qb = winbuf + (uint64_t)rr * BRX_WIN_STRIDE, *tbuf = qb + BRX_WIN_Q;
                int ncols = 0, nmatch = 0; bool nospace = false;
                const bool ok = brx_wave_align<1>(qb, (int)ql_, tbuf, (int)tl_, (int)cost_, tb_wave, scr_bytes / 8, nullptr,
                                                  &ncols, &nmatch, &nospace);
                if (lane == 0) {
                    s_win[w].ncols = (uint32_t)ncols; s_win[w].nmatch = (uint32_t)nmatch; s_win[w].ok = (ok || nospace) ? 1u : 0u;
                    if (nospace) { atomicOr(&flags[0], 1u); flags[8] = rr; flags[9] = ql_; flags[10] = tl_; flags[11] = cost_; }
                }
            }
        }
        __syncthreads();
        BRX_PHASE(4);
        if (parked) {
            ms.res_ncols = uni(s_win[wid].ncols); ms.res_nmatch = uni(s_win[wid].nmatch);
            if (!uni(s_win[wid].ok)) ms.status |= BRX_RS_BAND;
        }
    }
}

#endif /* BRX_MUTATE_WG_H */

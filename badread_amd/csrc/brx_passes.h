/*
 * brx_passes.h -- the mutate loop of the BULK set (/root/reference/badread/simulate.py:272-346), round 6, included by
 * brx_kernels.h behind brx_mutate.h.
 *
 * What the loop looks like from the hardware's side.  An iteration draws a position i, reads the k-mer of the ORIGINAL
 * fragment there (simulate.py:296: `fragment[i:i+k_size]`, never `new_fragment_bases`) and asks the error model for an
 * alternative (error_model.py:135-160).  Nothing of that depends on what earlier iterations did: the PROPOSAL of iteration t
 * is a pure function of (seed, read, t) and the fragment.  Only what happens to the ~7 % of proposals that change the k-mer
 * (the "survivors") is sequential: the changed-position test (simulate.py:309), the running error estimate with its
 * `est ** 1.5` (:321) and the alignment every 25th change (:325-346).
 *
 * Rounds 2-5 ran both halves in one wave per read (k_mutate_seg<false>): 64 proposals, then the survivors one after the other
 * by the wave -- seven of its 64 lanes take part in an application, one lane's worth of double arithmetic decides what follows --
 * with three dependent table loads per round in front of every application, and three to four device-scope atomics per read and
 * pass on two cache lines for its lists (11.4 ns each, chip-wide: what the kernel really waited for).  Here the loop is cut into
 * device functions, each shaped for what it does:
 *
 *   brx_propose_ahead   the WAVE, for one read: 64 iterations per lane-round, BRX_POST_U rounds per trip with their table loads in
 *                       flight together; the survivors are appended, in iteration order, to the read's RING (20 bytes an entry).
 *                       No loop state.  A survivor that is not consumed stays in the ring: nothing is proposed twice.
 *   brx_apply_read      one LANE, for its read: plain sequential code -- take the survivors from the ring in order, test the changed
 *                       map, write replacement words, update the estimate -- until the 25th change (MP_PARK), the end of the loop
 *                       (MP_FINISH) or an empty ring (MP_HUNGRY).  A read's map, replacement words and ring are touched by ONE
 *                       lane, so program order is all the ordering there is: no fences, no LDS.
 *   brx_lane_park       one LANE, for its read: the window's two strings as bit planes, the replacements spliced into the
 *                       fragment's plane words as bit ranges, guided by the changed map.
 *   brx_lanes_align     (brx_mutate.h) one LANE per window: band in registers, 2-bit move codes, canonical walk.
 *
 * and two ways of driving them:
 *
 *   k_mut_fill + k_mut_lanes (the default)   every survivor a read is expected to need is proposed before the loop starts; then ONE
 *                       launch in which a wave keeps 64 reads (neighbours in the order by expected changes / error rate) through
 *                       up to BRX_LANES_CYCLES alignment cycles: apply -> park -> align -> apply ..., no launch, no list, no atomic,
 *                       no host.  What is left of the reads with the most cycles is run to completion by k_mutate_seg.
 *   host-driven passes (BRX_MUTATE_PASSES=1)  {k_mut_apply, k_mut_post, k_pass_lists, k_win_lane, k_win_wave} per pass, the windows
 *                       regrouped by band class between the kernels, an in-place tail of BRX_TAIL_READS reads: the route of the
 *                       round's first half, kept for the A/B of DESIGN.md section 7 and as a second witness of the takeover
 *                       (k_mutate_seg picks a read up parked or hungry).
 *
 * Results are identical to the sequential loop either way, and to k_mutate_seg (MS keeps its meaning: round_loops / surv_lane /
 * j_next name the survivor to resume in).
 */
#ifndef BRX_PASSES_H
#define BRX_PASSES_H

#ifndef BRX_SV_CAP
#define BRX_SV_CAP 256u                                  /* ring entries per read (power of two) */
#endif
#ifndef BRX_SV_STOCK
#define BRX_SV_STOCK 56u                                 /* k_mut_post proposes until the ring holds this many survivors */
#endif
#ifndef BRX_POST_U
#define BRX_POST_U 2                                     /* proposal rounds (of 64 iterations) in flight per trip */
#endif
static_assert((BRX_SV_CAP & (BRX_SV_CAP - 1u)) == 0u && BRX_SV_CAP >= 64u * BRX_POST_U + BRX_SV_STOCK, "ring: a power of two with room for a trip");

/* ring bookkeeping of a read: sequence numbers of the first unconsumed and the first free entry, the first iteration that has
   not been proposed yet (a multiple of 64 until the loop cap is reached) */
struct PQ { uint32_t head, tail, next_t, pad; };

/* MS.phase between the kernels of a pass (0-3 as in brx_mutate.h) */
enum { MP_HUNGRY = 4,      /* ring empty, no alignment pending: resume at iteration round_loops (k_mutate_seg: a resumed round without blending) */
       MP_PARK = 5,        /* k_mut_apply -> k_mut_post: park the window; surv_lane / j_next say where the loop stopped */
       MP_FINISH = 6 };    /* k_mut_apply -> k_mut_post: the loop is over (round_loops = loop_count), write the epilogue */

/* iterations are counted in 32 bits here (RS.loops is 32 bits wide anyway): the reference's cap of 100 x frag_len iterations
   (simulate.py:280) is applied as written for reads of up to 42.9 Mb and at 2^32 - 256 iterations beyond */
__device__ __forceinline__ uint32_t brx_loop_cap32(uint32_t n) {
    const uint64_t cap = 100ull * (uint64_t)n;
    return cap > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)cap;
}

/* -------------------------------------------------------------------------------------------------
 * k_mut_apply: one read per lane
 * ----------------------------------------------------------------------------------------------- */
/* The sequential half of the loop for read r, by the calling LANE: its survivors from the ring until the 25th change, the end of
   the loop or an empty ring.  Returns the outcome (MP_PARK / MP_FINISH / MP_HUNGRY; 0 for an empty read) and leaves it in MS.phase. */
__device__ inline uint32_t brx_apply_read(const BrxDev &d, const RS *rs, MS *msv, PQ *pq, const uint32_t r,
                                          const uint4 *ra, const uint32_t *rz, const uint32_t ring_cap, uint32_t *repl, uint32_t *Cbuf) {
    const uint32_t n = rs[r].n;
    if (n == 0u) return 0u;
    const brx_error_model &em = d.em;
    const int k = em.k;
    const uint64_t F_off = rs[r].F_off;
    const double target = rs[r].target;
    uint32_t *rp = repl + F_off;
    uint32_t *cm = Cbuf + (F_off >> 4);
    MS *mp = &msv[r];
    const uint32_t phase = mp->phase;
    const double dn = (double)n;
    const uint32_t cap = brx_loop_cap32(n);
    double errors = 0.0, est = 1.0;
    uint32_t change = 0u, j0 = 0u;
    bool fresh = true;                              /* est is 1 - errors / dn of the CURRENT errors */
    uint32_t outcome = MP_HUNGRY, loops = 0u, t_at = 0u, jn = 0u;
    bool over = false;
    if (phase == 0u) {
        if (dn * (1.0 - target) < 0.5) { over = true; outcome = MP_FINISH; loops = 0u; }                 /* simulate.py:274 */
        else {
            est = 1.0 - errors / dn;
            if (est <= target) { over = true; outcome = MP_FINISH; loops = 1u; }                          /* :290-291 in iteration 1 */
        }
    } else {
        errors = mp->errors; est = mp->est; change = mp->change;
        if (phase == 1u) {                                                                                /* the window's alignment came back */
            const uint32_t nc = mp->res_ncols, nm = mp->res_nmatch;
            const double id = nc ? (double)nm / (double)nc : 0.0;                                         /* misc.py:228-240 */
            if (n <= BRX_ALIGN_SIZE) errors = (1.0 - id) * dn;                                            /* simulate.py:333 */
            else {
                const double est_err = (1.0 - id) * dn;
                const double weight = (double)BRX_ALIGN_SIZE / dn;
                errors = est_err * weight + errors * (1.0 - weight);                                      /* simulate.py:344-346 */
            }
            j0 = mp->j_next;
            fresh = false;                          /* the rest of this k-mer is applied with the estimate of its iteration (:321) */
        }
    }
    PQ q = pq[r];
    uint32_t slot = q.head % ring_cap;
    /* the entry behind the one being applied is in flight while it is: a survivor costs one dependent round trip (its map words),
       not two */
    uint4 e_next = make_uint4(0u, 0u, 0u, 0u);
    uint32_t pz_next = 0u;
    if (!over && q.head != q.tail) { e_next = ra[slot]; pz_next = rz[slot]; }
    while (!over) {
        if (q.head == q.tail) {
            if (q.next_t >= cap) { outcome = MP_FINISH; loops = cap + 1u; }                               /* :280-281 */
            else { outcome = MP_HUNGRY; t_at = q.next_t; }
            break;
        }
        const uint4 e = e_next;
        const uint32_t pz = pz_next;
        {
            const uint32_t ns = slot + 1u == ring_cap ? 0u : slot + 1u;
            if (q.head + 1u != q.tail) { e_next = ra[ns]; pz_next = rz[ns]; }
        }
        const uint32_t t = e.x, i0 = e.y, px = e.z, py = e.w;
        /* the changed map of positions i0 .. i0 + k - 1 */
        const uint32_t wi = i0 >> 5, sh = i0 & 31u;
        const uint32_t m0 = cm[wi], m1 = cm[wi + 1u];                                                     /* the word behind the map is the map of odd symbols: readable */
        const uint64_t cur = ((((uint64_t)m1) << 32) | (uint64_t)m0) >> sh;
        uint64_t set = 0ull;
        bool applies = false, parked = false;
        double scale = 0.0;
        uint32_t before = 0u;
        for (int j = 0; j < k; ++j) {
            /* the replacement word of position j (brx_prop_word, one position at a time) */
            uint32_t wj = 0u;
            if (py & BRX_PROP_RANDOM) wj = (uint32_t)j == (py & 0xFFu) ? px : 0u;
            else {
                const uint32_t len = (py & 0x10000u) ? (uint32_t)em.d_pool[px + 2u + (uint32_t)j] : (j < 8 ? (pz >> (4u * (uint32_t)j)) & 15u : 0u);
                if ((py >> j) & 1u) wj = 0x80000000u | (len << 24) | (px + 2u + (uint32_t)k + before);
                before += len;
            }
            if ((uint32_t)j < j0 || wj == 0u || ((cur >> j) & 1ull)) continue;                            /* simulate.py:309 */
            if (!applies) { applies = true; scale = est * brx_sqrt(est); }                                /* :321 */
            rp[i0 + (uint32_t)j] = wj;
            set |= 1ull << j;
            change += 1u;
            const uint32_t len = (wj >> 24) & 0x7Fu;
            errors += (double)(len < 2u ? 1u : len - 1u) * scale;
            if (change % BRX_ALIGN_INTERVAL == 0u) { parked = true; jn = (uint32_t)j + 1u; break; }      /* :325 */
        }
        if (set) {
            const uint64_t add = set << sh;
            if ((uint32_t)add) cm[wi] = m0 | (uint32_t)add;
            if ((uint32_t)(add >> 32)) cm[wi + 1u] = m1 | (uint32_t)(add >> 32);
        }
        if (parked) { outcome = MP_PARK; t_at = t; break; }                                               /* the entry stays: the next pass resumes in it */
        q.head += 1u;
        slot = slot + 1u == ring_cap ? 0u : slot + 1u;
        j0 = 0u;
        if (applies || !fresh) {                    /* top-of-loop tests of the iteration behind this survivor (:285-291) */
            const double est2 = 1.0 - errors / dn;
            if ((double)change > 0.9 * dn || est2 <= target) { outcome = MP_FINISH; loops = t + 2u; break; }
            est = est2;
            fresh = true;
        }
    }
    pq[r].head = q.head;
    mp->errors = errors; mp->est = est; mp->change = change;
    if (outcome == MP_FINISH) mp->round_loops = (uint64_t)loops;
    else { mp->round_loops = (uint64_t)(t_at & ~63u); mp->surv_lane = t_at & 63u; mp->j_next = outcome == MP_PARK ? jn : 0u; }
    mp->phase = outcome;
    return outcome;
}

__global__ void __launch_bounds__(64, 8) k_mut_apply(BrxDev d, const RS *rs, MS *msv, PQ *pq, const uint32_t *active_in, const uint32_t *n_in_ptr,
                                                      const uint4 *sv_a, const uint32_t *sv_z, uint32_t *repl, uint32_t *Cbuf) {
    const uint32_t idx = blockIdx.x * 64u + (uint32_t)lane_id();
    if (idx >= *n_in_ptr) return;
    const uint32_t r = active_in[idx];
    (void)brx_apply_read(d, rs, msv, pq, r, sv_a + (size_t)r * BRX_SV_CAP, sv_z + (size_t)r * BRX_SV_CAP, BRX_SV_CAP, repl, Cbuf);
}

/* -------------------------------------------------------------------------------------------------
 * k_mut_post: one read per wave -- park / epilogue, then propose ahead
 * ----------------------------------------------------------------------------------------------- */
/* The proposal of iteration t of `read` (identical to the proposal round of k_mutate_seg): false = the k-mer stays. */
__device__ __forceinline__ bool brx_propose_iter(const BrxDev &d, uint64_t read, uint32_t t, uint32_t max_i1, const uint32_t *f2g, const uint32_t *oddg,
                                                 bool coded, const uint8_t *F, uint32_t *ipos_out, BrxProp *out) {
    const brx_error_model &em = d.em;
    const int k = em.k;
    uint32_t w[4];
    brx_draw4(d.seed, read, BRX_ST_MUT, (uint64_t)t, w);
    const uint32_t ipos = (uint32_t)brx_mulhi64(((uint64_t)w[1] << 32) | w[0], (uint64_t)max_i1);
    const uint32_t wi = ipos >> 4, sh = 2u * (ipos & 15u);
    const uint32_t w0 = f2g[wi], w1 = f2g[wi + 1u];
    const uint32_t row = (uint32_t)(((((uint64_t)w0 << 32) | (uint64_t)w1) << sh) >> (64 - 2 * k));
    bool bad = false;                               /* a symbol outside ACGT in the k-mer: error_model.py:142-143 */
    if (!coded) {
        const uint64_t both = (((uint64_t)oddg[(ipos >> 5) + 1u] << 32) | (uint64_t)oddg[ipos >> 5]) >> (ipos & 31u);
        bad = (both & ((1ull << k) - 1ull)) != 0ull;
    }
    BrxProp pr;
    if (bad) {
        const BrxRc c = dev_random_change_split(w[3], k);
        pr.x = dev_random_change_word(c, (uint32_t)F[ipos + c.pos]); pr.y = BRX_PROP_RANDOM | c.pos; pr.z = 0u;
    } else pr = dev_propose_row(em, row, w[2], w[3]);
    *ipos_out = ipos; *out = pr;
    return pr.y != 0u;
}

/* Propose ahead for read r, by the whole WAVE: survivors of the next iterations into the ring, in iteration order, until it holds
   `stock` of them (or cannot take another trip's worth, or the loop cap is reached). */
template <int U>
__device__ inline void brx_propose_ahead(const BrxDev &d, const uint32_t r, const RS *rs, PQ *pq, const uint8_t *Fbuf, const uint32_t *F2buf,
                                         const uint32_t *Cbuf, uint4 *ra, uint32_t *rz, const uint32_t ring_cap, const uint32_t stock) {
    const int lane = lane_id();
    const uint32_t n = uni(rs[r].n);
    if (n == 0u) return;
    const uint64_t F_off = uni(rs[r].F_off);
    const int k = d.em.k;
    const uint8_t *F = Fbuf + F_off;
    PQ q = pq[r];
    q.head = uni(q.head); q.tail = uni(q.tail); q.next_t = uni(q.next_t);
    const uint32_t tail0 = q.tail, next0 = q.next_t;
    const uint32_t cap = brx_loop_cap32(n);
    const uint32_t max_i1 = n - (uint32_t)k;                   /* max_kmer_index + 1 (simulate.py:270) */
    const uint32_t *f2g = F2buf + (F_off >> 4);
    const uint32_t nw2 = (n + 15u) >> 4, nwc = (n + 31u) >> 5;
    const bool coded = uni(f2g[nw2] == 0u);
    const uint32_t *oddg = Cbuf + (F_off >> 4) + nwc;
    uint32_t have = q.tail - q.head;
    while (have < stock && have + 64u * (uint32_t)U <= ring_cap && q.next_t < cap) {
        uint32_t ip[U]; BrxProp pr[U]; bool sv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t t = q.next_t + 64u * (uint32_t)u + (uint32_t)lane;
            sv[u] = false; ip[u] = 0u; pr[u].x = pr[u].y = pr[u].z = 0u;
            if (t < cap) sv[u] = brx_propose_iter(d, d.first_read + r, t, max_i1, f2g, oddg, coded, F, &ip[u], &pr[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned long long m = __ballot(sv[u]);
            if (sv[u]) {
                const uint32_t slot = (q.tail + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) % ring_cap;
                ra[slot] = make_uint4(q.next_t + 64u * (uint32_t)u + (uint32_t)lane, ip[u], pr[u].x, pr[u].y);
                rz[slot] = pr[u].z;
            }
            const uint32_t cnt = (uint32_t)__popcll(m);
            q.tail += cnt; have += cnt;
        }
        const uint32_t room = cap - q.next_t;
        q.next_t += room < 64u * (uint32_t)U ? room : 64u * (uint32_t)U;
    }
    if (lane == 0 && (q.tail != tail0 || q.next_t != next0)) { pq[r].tail = q.tail; pq[r].next_t = q.next_t; }
}

template <int U>
__global__ void __launch_bounds__(64, 8) k_mut_post(BrxDev d, RS *rs, MS *msv, PQ *pq, const uint32_t *active_in, const uint32_t *n_in_ptr,
                                                     const MutAux aux, const uint8_t *Fbuf, uint32_t *repl,
                                                     const uint32_t *F2buf, const uint32_t *Cbuf, uint4 *sv_a, uint32_t *sv_z) {
    const int lane = lane_id();
    const brx_error_model &em = d.em;
    const int k = em.k;
    const uint32_t n_in = uni(*n_in_ptr);
    for (uint32_t qi = blockIdx.x; qi < n_in; qi += gridDim.x) {
        const uint32_t r = uni(active_in[qi]);
        const RS s = rs[r];
        if (s.n == 0) continue;
        const uint64_t t_begin = __builtin_amdgcn_s_memtime();
        MS ms = msv[r];
        const uint32_t n = s.n;
        const uint8_t *F = Fbuf + s.F_off;
        uint32_t *rp = repl + s.F_off;
        uint64_t *ck = aux.clk + (uint64_t)r * 8;
        bool goes_on = false;                       /* the read takes part in the next pass */
        if (ms.phase == (uint32_t)MP_PARK) {
            /* ---- park the read: the window pair goes to its slot (simulate.py:325-343) ---- */
            uint32_t a = 0, b = n;
            if (n > BRX_ALIGN_SIZE) {
                uint32_t ww[4];
                brx_draw4(d.seed, d.first_read + r, BRX_ST_WIN, (uint64_t)ms.nalign, ww);
                a = (uint32_t)brx_mulhi64(((uint64_t)ww[1] << 32) | ww[0], (uint64_t)n - BRX_ALIGN_SIZE + 1);
                b = a + BRX_ALIGN_SIZE;
            }
            uint32_t cost = 0;
            uint8_t *qb = aux.winbuf + (uint64_t)r * BRX_WIN_STRIDE, *tbuf = qb + BRX_WIN_Q;
            bool odd = false;
            const uint32_t tl = wave_park<true>(em, F, rp, a, b, qb, tbuf, BRX_WIN_TMAX, &cost, &odd, reinterpret_cast<uint32_t *>(qb + BRX_WIN_PLANES));
            const uint32_t ql = b - a;
            uint32_t klass = MC_LEGACY;
            int band_blocks_of = 0;
            if (tl <= BRX_WIN_TMAX) {
                const BrxGeom g = brx_make_geom((int)ql, (int)tl, (int)cost);
                const int band_blocks = (g.dhi - g.dlo) / 32 + 2;
                band_blocks_of = band_blocks;
                /* "easy" windows go to the throughput kernel: one window per LANE (k_win_lane) */
                const bool easy = !odd && g.G == 1 && tl <= BRX_LANE_TMAX && ql > 0 && tl > 0 && band_blocks <= BRX_LANE_W;
                klass = easy ? MC_EASY : MC_HARD;
            }
            if (lane == 0) {
                MS *o = &msv[r];
                o->nalign = ms.nalign + 1u;
                o->phase = klass == MC_LEGACY ? 3u : 1u;
                o->win_a = a; o->win_b = b; o->tl = tl; o->cost = cost; o->res_ncols = 0; o->res_nmatch = 0;
                o->passes = ms.passes + 1u;
                o->win_kind = klass | (brx_lane_class(band_blocks_of) << 8);     /* k_pass_lists enters the read in the list of its aligner */
            }
            goes_on = klass != MC_LEGACY;
        } else if (ms.phase == (uint32_t)MP_FINISH) {
            /* the loop is over: the epilogue waits for k_mut_epilogue, once, behind the last pass */
        } else goes_on = true;                      /* not started (the fill before the first pass) or hungry */
        if (goes_on) brx_propose_ahead<U>(d, r, rs, pq, Fbuf, F2buf, Cbuf, sv_a + (size_t)r * BRX_SV_CAP, sv_z + (size_t)r * BRX_SV_CAP, BRX_SV_CAP, BRX_SV_STOCK);
        if (lane == 0) ck[0] += __builtin_amdgcn_s_memtime() - t_begin;
    }
}

/* -------------------------------------------------------------------------------------------------
 * k_mut_epilogue: one finished read per wave, ONCE behind the last pass
 * -----------------------------------------------------------------------------------------------
 * Lengths of the mutated read, trims (simulate.py:348-349), proven distance bound: a walk over the whole read (n / 64 steps of
 * dependent loads).  Inside the passes -- where rounds 2-5 and the first k_mut_post had it -- the ~1000 reads that finish in a
 * pass made that pass's kernel as long as ONE such walk, 64 times per batch; nothing needs these numbers before the final stage. */
__global__ void __launch_bounds__(64, 8) k_mut_epilogue(BrxDev d, RS *rs, MS *msv, const uint32_t *list, uint32_t n_list, const MutAux aux,
                                                         const uint8_t *Fbuf, const uint32_t *repl) {
    const int lane = lane_id();
    const brx_error_model &em = d.em;
    const int k = em.k;
    for (uint32_t qi = blockIdx.x; qi < n_list; qi += gridDim.x) {
        const uint32_t r = uni(list[qi]);
        if (uni(msv[r].phase) != (uint32_t)MP_FINISH) continue;
        const RS s = rs[r];
        const MS ms = msv[r];
        const uint32_t n = s.n;
        const uint8_t *F = Fbuf + s.F_off;
        const uint32_t *rp = repl + s.F_off;
        uint32_t cost = 0;
        const uint32_t m = wave_join(em, F, rp, 0, n, nullptr, &cost);
        uint32_t st = 0, et = 0;
        if (lane < k) { st = rep_len(rp[lane]); et = rep_len(rp[n - k + lane]); }
        st = wave_sum(st); et = wave_sum(et);
        if (lane == 0) {
            RS *o = &rs[r];
            o->status = s.status | ms.status; o->m = m; o->ub = cost; o->start_trim = st; o->end_trim = et;
            o->loops = (uint32_t)ms.round_loops; o->changes = ms.change; o->naligns = ms.nalign;
            o->units = 0;                                          /* sized by k_fin_join */
            msv[r].phase = 2u;
            aux.clk[(uint64_t)r * 8 + 1] = ms.passes;
        }
    }
}

/* -------------------------------------------------------------------------------------------------
 * k_pass_lists: one read per lane -- the lists of the pass's window kernels and of the next pass
 * -----------------------------------------------------------------------------------------------
 * Every read of the pass enters the list its state names: the next pass's input (parked or hungry), the lane kernel's list of its
 * band class, the wave kernel's list, the whole-read kernel's list.  An append is
 * `atomicAdd(counter, pred)` by EVERY lane on a wave-uniform address: the compiler turns that into one atomic per wave and a prefix
 * over the lanes.  (Rounds 2-5 appended from one lane of a wave per read: three to four device-scope atomics per read and pass on
 * two cache lines -- at 11.4 ns per atomic on a line, profiles/r06_atomic_bench.jsonl, the first pass of a 65536-read batch could
 * not take less than 2.2 ms whatever its waves did: the pass kernel of those rounds was waiting for THIS 60 % of its time.) */
__global__ void __launch_bounds__(64) k_pass_lists(const MS *msv, const uint32_t *active_in, const uint32_t *n_in_ptr, uint32_t *active_out,
                                                    uint32_t *ctr, uint32_t *lane_cls, const MutAux aux, uint32_t n_reads) {
    const uint32_t idx = blockIdx.x * 64u + (uint32_t)lane_id();
    const bool in = idx < *n_in_ptr;
    const uint32_t r = in ? active_in[idx] : 0u;
    const uint32_t phase = in ? msv[r].phase : 0xFFu, kind = in ? msv[r].win_kind : 0u;
    const bool parked = phase == 1u;
    const uint32_t klass = kind & 0xFFu, cls = kind >> 8;
    {
        const bool p = parked || phase == (uint32_t)MP_HUNGRY;
        const uint32_t at = atomicAdd(&ctr[MC_OUT], p ? 1u : 0u);
        if (p) active_out[at] = r;
    }
#pragma unroll
    for (uint32_t cc = 0; cc < (uint32_t)BRX_LANE_CLASSES; ++cc) {
        const bool p = parked && klass == (uint32_t)MC_EASY && cls == cc;
        if (__ballot(p) == 0ull) continue;
        const uint32_t at = atomicAdd(&lane_cls[cc * BRX_CLS_STRIDE], p ? 1u : 0u);
        if (p) aux.req_easy[(size_t)cc * n_reads + at] = r;
    }
    {
        const bool p = parked && klass == (uint32_t)MC_HARD;
        if (__ballot(p) != 0ull) {
            const uint32_t at = atomicAdd(&ctr[MC_HARD], p ? 1u : 0u);
            if (p) aux.req_hard[at] = r;
        }
    }
    {
        const bool p = phase == 3u;
        if (__ballot(p) != 0ull) {
            const uint32_t at = atomicAdd(aux.legacy_ctr, p ? 1u : 0u);
            if (p) aux.req_legacy[at] = r;
        }
    }
}

/* =================================================================================================
 * k_mut_lanes: the whole mutate loop of 64 reads in ONE wave, from the first iteration to the last
 * =================================================================================================
 * The passes above regroup the reads between kernels (lists by band class, a launch per step, the host in between).  Nothing in
 * the loop needs that: a read's survivors are in its ring (proposed ahead, all of them, by k_mut_fill before this kernel starts), the
 * sequential half is one lane's work (brx_apply_read), a window's alignment is one lane's work (brx_lanes_align: the band in
 * registers), and so -- this round -- is its parking (brx_lane_park: the lane joins its own 1000 positions and packs both strings
 * into bit planes as it goes).  A wave therefore keeps its 64 reads (neighbours in the order by expected changes: equal work) and
 * turns the crank until the last of them is done: apply -> park -> align -> apply ..., no launch, no list, no atomic, no host.
 * What a lane cannot do alone is taken by the whole wave for that one read: a window the lane aligner does not take (symbols outside
 * ACGT, more than eight band blocks) goes through wave_park + brx_wave_align as in k_win_wave; a ring that runs empty is refilled by
 * brx_propose_ahead; a window that overflows its slot sends the read to the whole-read kernel.
 * Instructions per read and alignment cycle: ~3 k (apply 60, park 250, align 1.9-2.6 k as 1/64 of the wave's) against ~69 k of the
 * in-place kernel (k_mutate_seg: 61 k of them one window on a whole wave) and ~8 k + 1.9 k of the passes.
 *
 * Ring of read r: entries [ring_base(r), + ring_cap(r)) of sv_a / sv_z with ring_base = F_off / 8 + 128 r, ring_cap = n / 8 + 128:
 * the fragment offsets are a prefix sum over the reads already, so no layout pass is needed; n / 8 entries hold every survivor of a
 * read down to ~88 % identity, a rougher read wraps around and is refilled. */
#ifndef BRX_LANES_MAX_WAVES
#define BRX_LANES_MAX_WAVES 1024u                        /* waves of one k_mut_lanes launch (a 6.6 MB store of move codes each); a bigger batch's groups of 64 reads follow each other in a wave */
#endif
#ifndef BRX_RING_SHIFT
#define BRX_RING_SHIFT 3                                 /* ring entries per read: n >> BRX_RING_SHIFT ... */
#define BRX_RING_MIN 128u                                /* ... + BRX_RING_MIN (at least a trip's 64 U; the tests build tiny rings) */
#endif
__device__ __forceinline__ uint64_t brx_ring_base(const RS *rs, uint32_t r) { return (rs[r].F_off >> BRX_RING_SHIFT) + (uint64_t)BRX_RING_MIN * (uint64_t)r; }
__device__ __forceinline__ uint32_t brx_ring_cap(const RS *rs, uint32_t r) { return (rs[r].n >> BRX_RING_SHIFT) + BRX_RING_MIN; }

/* Survivors a read is expected to need over its whole loop: every change adds at least target^1.5 errors while the loop runs
   (simulate.py:321) and a survivor applies one change or more. */
__device__ __forceinline__ uint32_t brx_ring_want(uint32_t n, double target) {
    const double need = (double)n * (1.0 - target);
    if (need < 0.5) return 0u;
    const double t = target > 0.05 ? target : 0.05;
    const double s = 0.9 * need / (t * brx_sqrt(t)) + 32.0;              /* (measured: a read takes ~0.8 survivors per error it needs; what is short is refilled in the kernel) */
    return s > 4.0e9 ? 0xF0000000u : (uint32_t)s;
}

/* all the survivors a read is expected to need, before the loop starts (a wave per read; many rounds each, U in flight) */
template <int U>
__global__ void __launch_bounds__(64, 8) k_mut_fill(BrxDev d, const RS *rs, PQ *pq, const uint32_t *list, uint32_t n_list, const uint8_t *Fbuf,
                                                     const uint32_t *F2buf, const uint32_t *Cbuf, uint4 *sv_a, uint32_t *sv_z) {
    for (uint32_t qi = blockIdx.x; qi < n_list; qi += gridDim.x) {
        const uint32_t r = uni(list[qi]);
        const uint32_t n = uni(rs[r].n);
        if (n == 0u) continue;
        const uint64_t base = uni(brx_ring_base(rs, r));
        const uint32_t cap = uni(brx_ring_cap(rs, r));
        uint32_t want = brx_ring_want(n, rs[r].target);
        want = uni(want);
        const uint32_t room = cap - 64u * (uint32_t)U;
        brx_propose_ahead<U>(d, r, rs, pq, Fbuf, F2buf, Cbuf, sv_a + base, sv_z + base, cap, want < room ? want : room);
    }
}

/* The window [a, b) of read r, parked by ONE lane: query planes pl[0, 32) / pl[32, 64) and target planes pl[64, ...) / pl[64 + TW, ...)
   as wave_park<true> writes them (bit x of word x / 32: symbol x; lo = bit 0 of the code, hi = bit 1), joined length, edit bound
   and "a symbol outside ACGT".  Target planes are written for the first BRX_LANE_TMAX symbols only (a longer window is not the
   lane aligner's).
   32 positions per step: their fragment bytes become a query plane word by multiplication (brx_byte_bits), and the TARGET is that
   word with the step's replacements spliced in -- the run of untouched positions in front of a replaced one is appended as a bit
   range, then the replacement's characters.  Which positions are replaced comes from the read's changed map (one word per 32
   positions, kept by brx_apply_read), so the replacement words and the pool are read for the ~5 % of positions that have one,
   not for all 1000.  (A first version walked position by position over the bytes and the replacement words: 4 KB of loads and
   ~20 k instructions per window, a third of the wave's cycle: profiles/r06_lanes_phase_probe.json.) */
__device__ inline void brx_lane_park(const brx_error_model &em, const uint8_t *__restrict__ F, const uint32_t *__restrict__ rp,
                                     const uint32_t *__restrict__ cm, const uint32_t a, const uint32_t b,
                                     uint32_t *__restrict__ pl, uint32_t *tl_out, uint32_t *cost_out, bool *odd_out) {
    constexpr uint32_t TW = BRX_LANE_TMAX / 32;
    uint32_t tl = 0, cost = 0;
    bool odd = false;
    uint32_t tlo = 0, thi = 0;                     /* the target word being filled: bits [0, tl & 31) */
    const uint32_t ql = b - a;
    /* append `nb` (1..32) symbols given as plane bits (zero above nb) to the target */
    auto append = [&](uint32_t lo, uint32_t hi, uint32_t nb) {
        const uint32_t sh = tl & 31u;
        if (tl < BRX_LANE_TMAX) {
            tlo |= lo << sh; thi |= hi << sh;
            if (sh + nb >= 32u) {
                pl[64u + (tl >> 5)] = tlo; pl[64u + TW + (tl >> 5)] = thi;
                tlo = sh ? lo >> (32u - sh) : 0u; thi = sh ? hi >> (32u - sh) : 0u;
            }
        }
        tl += nb;
    };
    for (uint32_t x0 = 0; x0 < ql; x0 += 32u) {
        const uint32_t p0 = a + x0;
        const uint32_t nv = ql - x0 < 32u ? ql - x0 : 32u;
        const BrxB16 f0 = *reinterpret_cast<const BrxB16 *>(F + p0);             /* F holds 16 bytes behind the read; a window's last step may read 16 more: inside the buffers' slack */
        const BrxB16 f1 = *reinterpret_cast<const BrxB16 *>(F + p0 + 16u);
        const uint32_t m0 = cm[p0 >> 5], m1 = cm[(p0 >> 5) + 1u];                 /* the word behind the map is the map of odd symbols: readable */
        const uint32_t live = nv >= 32u ? 0xFFFFFFFFu : (1u << nv) - 1u;
        uint32_t changed = (uint32_t)(((((uint64_t)m1) << 32) | (uint64_t)m0) >> (p0 & 31u)) & live;
        const uint32_t fw[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
        uint32_t qlo = 0, qhi = 0, oddbits = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            qlo |= brx_byte_bits(fw[q], 0) << (4 * q); qhi |= brx_byte_bits(fw[q], 1) << (4 * q);
            oddbits |= (fw[q] & 0xFCFCFCFCu) ? (0xFu << (4 * q)) : 0u;           /* some byte of these four is outside ACGT (which one: below) */
        }
        qlo &= live; qhi &= live;
        if (oddbits & live) {                                                      /* rare: look at the bytes of the live positions */
            for (uint32_t i = 0; i < nv; ++i) odd |= (((fw[i >> 2] >> (8u * (i & 3u))) & 0xFCu) != 0u);
        }
        pl[x0 >> 5] = qlo; pl[32u + (x0 >> 5)] = qhi;
        uint32_t cur = 0;                                                          /* positions [0, cur) of this step are in the target */
        while (changed) {
            const uint32_t c = (uint32_t)__ffs((int)changed) - 1u;
            changed &= changed - 1u;
            if (c > cur) { const uint32_t g = c - cur, mk = (1u << g) - 1u; append((qlo >> cur) & mk, (qhi >> cur) & mk, g); }
            const uint32_t w = rp[p0 + c];
            const uint32_t sym = (fw[c >> 2] >> (8u * (c & 3u))) & 0xFFu;
            const uint32_t len = (w >> 24) & 0x7Fu;
            bool has = false;
            for (uint32_t y = 0; y < len; ++y) {
                const uint32_t ch = (uint32_t)rep_char(em, w, y);
                odd |= ch > 3u; has |= ch == sym;
                append(ch & 1u, (ch >> 1) & 1u, 1u);
            }
            cost += len < 2u ? 1u : len - (has ? 1u : 0u);                          /* rep_cost */
            cur = c + 1u;
        }
        if (nv > cur) { const uint32_t g = nv - cur, mk = g >= 32u ? 0xFFFFFFFFu : (1u << g) - 1u; append((qlo >> cur) & mk, (qhi >> cur) & mk, g); }
    }
    if (tl <= BRX_LANE_TMAX && (tl & 31u) != 0u) { pl[64u + (tl >> 5)] = tlo; pl[64u + TW + (tl >> 5)] = thi; }
    *tl_out = tl; *cost_out = cost; *odd_out = odd;
}

/* PROFILE (BRX_PROFILE=1): shader-clock time of the wave per step of the cycle, added to phase[8 r0 + i] of the wave's first read:
   0 apply   1 refill (rings that ran empty)   2 park   3 whole-wave windows   4 lane aligner   7 cycles run */
#define BRX_LPH(i) do { if constexpr (PROFILE) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); lph[i] += now_ - lph_t; lph_t = now_; } } while (0)
template <int U, bool PROFILE = false>
__global__ void __launch_bounds__(64, 4) k_mut_lanes(BrxDev d, RS *rs, MS *msv, PQ *pq, const uint32_t *list, uint32_t n_list, const MutAux aux,
                                                      const uint8_t *Fbuf, uint32_t *repl, const uint32_t *F2buf, uint32_t *Cbuf,
                                                      uint4 *sv_a, uint32_t *sv_z, uint2 *tbw_base, uint32_t max_cycles,
                                                      uint32_t *left_list, uint32_t *left_ctr) {
    const int lane = lane_id();
    const brx_error_model &em = d.em;
    uint2 *tbw = tbw_base + (uint64_t)blockIdx.x * BRX_LANE_TB_UNITS;
    uint2 *tb_wave = reinterpret_cast<uint2 *>(aux.scr_base + (uint64_t)blockIdx.x * aux.scr_bytes);      /* the wave aligner's store (hard windows) */
    for (uint32_t grp = blockIdx.x; 64u * grp < n_list; grp += gridDim.x) {
        const uint32_t idx = 64u * grp + (uint32_t)lane;
        const uint32_t r = idx < n_list ? list[idx] : 0u;
        bool active = idx < n_list && rs[r].n != 0u;
        const uint64_t base = active ? brx_ring_base(rs, r) : 0ull;
        const uint32_t cap = active ? brx_ring_cap(rs, r) : 1u;
        const uint32_t n = active ? rs[r].n : 0u;
        const uint8_t *F = Fbuf + (active ? rs[r].F_off : 0ull);
        uint32_t *rp = repl + (active ? rs[r].F_off : 0ull);
        uint32_t *pl = reinterpret_cast<uint32_t *>(aux.winbuf + (uint64_t)r * BRX_WIN_STRIDE + BRX_WIN_PLANES);
        uint64_t t_last = __builtin_amdgcn_s_memtime();
        uint64_t lph[6] = {0, 0, 0, 0, 0, 0}, lph_t = t_last;
        /* at most max_cycles alignment cycles here: a lane walks its window's 1000 columns alone (~1.4 ms a cycle whatever the chip
           does beside it), and the reads with the most cycles are the batch's critical path -- what is left of them is run to
           completion by k_mutate_seg, one wave per read (0.2-0.4 ms a cycle), which takes a read over in any state */
        for (uint32_t cyc = 0; cyc < max_cycles && __ballot(active) != 0ull; ++cyc) {
            /* ---- the sequential half, every lane its own read ---- */
            uint32_t outcome = 0u;
            if (active) outcome = brx_apply_read(d, rs, msv, pq, r, sv_a + base, sv_z + base, cap, repl, Cbuf);
            if (outcome == (uint32_t)MP_FINISH) active = false;                      /* epilogue: k_mut_epilogue, behind this kernel */
            BRX_LPH(0);
            /* ---- a ring that ran empty: the wave proposes ahead for that read (rare: the rings are filled for the whole loop) ---- */
            {
                unsigned long long hungry = __ballot(active && outcome == (uint32_t)MP_HUNGRY);
                while (hungry) {
                    const int l = __ffsll((long long)hungry) - 1;
                    hungry &= hungry - 1;
                    const uint32_t rr = wave_bcast_u32(r, l);
                    const uint64_t bb = wave_bcast_u64(base, l);
                    const uint32_t cc = wave_bcast_u32(cap, l);
                    const uint32_t room = cc - 64u * (uint32_t)U;
                    brx_propose_ahead<U>(d, rr, rs, pq, Fbuf, F2buf, Cbuf, sv_a + bb, sv_z + bb, cc, room < 256u ? room : 256u);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_waitcnt(0);                                        /* the new entries are read by the reads' own lanes */
            }
            BRX_LPH(1);
            /* ---- park: every lane the window of its own read (simulate.py:325-343) ---- */
            const bool parking = active && outcome == (uint32_t)MP_PARK;
            uint32_t a = 0, b = 0, tl = 0, cost = 0;
            bool odd = false;
            if (parking) {
                b = n;
                const uint32_t nal = msv[r].nalign;
                if (n > BRX_ALIGN_SIZE) {
                    uint32_t ww[4];
                    brx_draw4(d.seed, d.first_read + r, BRX_ST_WIN, (uint64_t)nal, ww);
                    a = (uint32_t)brx_mulhi64(((uint64_t)ww[1] << 32) | ww[0], (uint64_t)n - BRX_ALIGN_SIZE + 1);
                    b = a + BRX_ALIGN_SIZE;
                }
                brx_lane_park(em, F, rp, Cbuf + (rs[r].F_off >> 4), a, b, pl, &tl, &cost, &odd);
                MS *o = &msv[r];
                o->nalign = nal + 1u; o->passes += 1u;
                o->win_a = a; o->win_b = b; o->tl = tl; o->cost = cost; o->res_ncols = 0; o->res_nmatch = 0;
                o->phase = tl <= BRX_WIN_TMAX ? 1u : 3u;
            }
            const uint32_t ql = b - a;
            bool easy = false;
            if (parking && tl <= BRX_WIN_TMAX) {
                const BrxGeom g = brx_make_geom((int)ql, (int)tl, (int)cost);
                easy = !odd && g.G == 1 && tl <= BRX_LANE_TMAX && ql > 0 && tl > 0 && (g.dhi - g.dlo) / 32 + 2 <= BRX_LANE_W;
            }
            /* a window that does not fit its slot: the whole-read kernel starts the read over (k_mutate) */
            if (parking && tl > BRX_WIN_TMAX) { aux.req_legacy[atomicAdd(aux.legacy_ctr, 1u)] = r; active = false; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);                                            /* planes and MS of every lane are in place */
            BRX_LPH(2);
            /* ---- windows the lane aligner does not take: one at a time on the whole wave (k_win_wave's way) ---- */
            {
                unsigned long long hard = __ballot(parking && tl <= BRX_WIN_TMAX && !easy);
                while (hard) {
                    const int l = __ffsll((long long)hard) - 1;
                    hard &= hard - 1;
                    const uint32_t rr = wave_bcast_u32(r, l);
                    const RS s2 = rs[rr];
                    const uint32_t a2 = wave_bcast_u32(a, l), b2 = wave_bcast_u32(b, l);
                    uint8_t *qb = aux.winbuf + (uint64_t)rr * BRX_WIN_STRIDE, *tbuf = qb + BRX_WIN_Q;
                    uint32_t cost2 = 0; bool odd2 = false;
                    const uint32_t tl2 = wave_park<false>(em, Fbuf + s2.F_off, repl + s2.F_off, a2, b2, qb, tbuf, BRX_WIN_TMAX, &cost2, &odd2);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_s_waitcnt(0);                                    /* the aligner reads the bytes back */
                    int ncols = 0, nmatch = 0; bool nospace = false;
                    const bool ok = brx_wave_align<1>(qb, (int)(b2 - a2), tbuf, (int)tl2, (int)cost2, tb_wave, aux.scr_bytes / 8, nullptr,
                                                      &ncols, &nmatch, &nospace);
                    if (lane == 0) {
                        msv[rr].res_ncols = (uint32_t)ncols; msv[rr].res_nmatch = (uint32_t)nmatch;
                        if (!ok && !nospace) msv[rr].status |= BRX_RS_BAND;
                        if (nospace) { atomicOr(&aux.flags[0], 1u); aux.flags[8] = rr; aux.flags[9] = b2 - a2; aux.flags[10] = tl2; aux.flags[11] = cost2; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_s_waitcnt(0);
                }
            }
            BRX_LPH(3);
            /* ---- every other window: one per lane, the band in registers (brx_lanes_align) ---- */
            if (__ballot(easy) != 0ull) {
                uint32_t ncols = 0, nmatch = 0; bool ok = false;
                brx_lanes_align<BRX_LANE_TMAX / 32>(easy, pl, (int)ql, (int)tl, (int)cost, tbw, &ncols, &nmatch, &ok);
                if (easy) {
                    msv[r].res_ncols = ncols; msv[r].res_nmatch = nmatch;
                    if (!ok) msv[r].status |= BRX_RS_BAND;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_waitcnt(0);                                        /* the store of move codes is written again next cycle */
            }
            BRX_LPH(4);
            if constexpr (PROFILE) lph[5] += 1;
            if (idx < n_list && n != 0u) {                                            /* brx_last_read_cycles: the wave's time, charged to its reads while they run */
                const uint64_t now = __builtin_amdgcn_s_memtime();
                if (active || outcome != 0u) aux.clk[(uint64_t)r * 8] += now - t_last;
                t_last = now;
            }
        }
        if constexpr (PROFILE) {
            const uint32_t r0 = wave_bcast_u32(r, 0);
            if (lane < 6) aux.phase[(uint64_t)r0 * 8 + (lane == 5 ? 7u : (uint32_t)lane)] += lane == 0 ? lph[0] : lane == 1 ? lph[1] : lane == 2 ? lph[2] : lane == 3 ? lph[3] : lane == 4 ? lph[4] : lph[5];
        }
        /* the reads that are not done: parked with their alignment's result (phase 1) or between two survivors (MP_HUNGRY) */
        {
            const uint32_t at = atomicAdd(left_ctr, active ? 1u : 0u);                /* one atomic per wave (the compiler's wave reduction) */
            if (active) left_list[at] = r;
        }
    }
}

#endif /* BRX_PASSES_H */

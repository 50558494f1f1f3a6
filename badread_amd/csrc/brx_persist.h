/*
 * brx_persist.h -- the mutate loop of sequence_fragment (/root/reference/badread/simulate.py:272-346) as ONE persistent
 * launch per device batch, included by brx_kernels.h.
 *
 * Round 2 ran the loop as ~75 passes of {k_mutate_seg, k_win_lane, k_win_wave} with a host round trip every few passes,
 * beside a run-to-completion chain for the longest reads and an in-place tail for the last twelfth of the batch.  The
 * passes were cheap in instructions (one window per LANE) but every pass was a barrier over the whole batch; head and tail
 * escaped the barriers by aligning every window with a whole wave (61 k wave-instructions per window against 1.9 k): 25 of
 * the pipeline's 137 VALU instructions per simulated base for a third of the bases.
 *
 * Here the waves of one launch pull work from device queues until the batch is done; nothing waits for anything but
 * its own data:
 *
 *   segment   a wave takes a read (fresh, longest first, or one whose identity check has come back), runs the loop --
 *             64 k-mer proposals per round, survivors applied in iteration order, exactly k_mutate_seg's code and draws --
 *             until the read is finished or the 25th change asks for an identity check.  There it either aligns the
 *             window IN PLACE with the wave aligner (windows the lane aligner cannot take; reads with many checks
 *             still ahead, whose chain of checks is the batch's critical path; the drain of the batch), or PARKS the
 *             read: window pair as 2-bit planes + loop state to global memory, read index to the lane queue.
 *   lanes     a wave that finds 64 parked windows aligns them at once, one window per LANE, band state in REGISTERS
 *             (brx_lanes_align below), and hands the reads back through the return queue.
 *
 * The error model's self thresholds (high halves, 32 KB for k = 7) are staged in LDS once per workgroup for the whole
 * stage: the ~93 % of draws that leave the k-mer unchanged are settled by one 16-bit LDS compare (SURVEY.md section
 * 0.6 / Appendix C: the g1 row of the scope table).
 *
 * Queues.  BRX_PQ_NX sets (one per XCD: a read stays with the set it was dealt to, so its state, its replacement words
 * and its planes are written and read through ONE L2 as long as nobody steals -- a workgroup serves the set of the XCD
 * it runs on (HW_REG_XCC_ID) first and the others when its own has nothing to do).  Each set: a slice of the
 * longest-first processing order, a ring of parked windows, a ring of returned reads.  Rings are ticket rings with a
 * semaphore of completed pushes: a consumer takes from the semaphore first, then a ticket, then waits (briefly) for the
 * slot of that ticket to be written.  No wave ever waits for work that only another wave can create: whoever creates
 * work goes on looking for work, so a single wave can finish a batch (the CPU interpretation of these kernels runs the
 * workgroups one after another).
 *
 * Hand-offs follow the rules of the chip (8 XCDs with private L2s, per-CU L1 never refreshed by other CUs' stores):
 * every word another wave will read is written with an agent-scope (write-through) store, the producer drains its
 * stores (s_waitcnt vmcnt(0)) before it publishes the ticket, the consumer issues ONE agent-scope acquire after it has
 * its ticket and reads the loop state with agent-scope loads.  Results do not depend on which wave or XCD does what:
 * every draw is a pure function of (seed, read, iteration), and an alignment result is applied exactly where the
 * in-place alignment was.
 */
#ifndef BRX_PERSIST_H
#define BRX_PERSIST_H

#define BRX_PQ_NX 8                                   /* queue sets */
#define BRX_PS_WAVES 4                                /* waves per workgroup of k_mutate_persist */
#define BRX_PS_THR_ROWS 16384                         /* 4^7: k = 7 error models keep their self thresholds in LDS */
#define BRX_PL_W 8                                    /* lane aligner: band blocks (32 rows) a lane holds in registers */
#define BRX_PL_QW 32                                  /* query plane words: windows of up to 1024 rows */
#define BRX_PL_TMAX 1280                              /* target columns of a parked window */
#define BRX_PL_TW (BRX_PL_TMAX / 32)
#define BRX_PL_WORDS (2 * BRX_PL_QW + 2 * BRX_PL_TW)  /* plane words per read: q_lo[32] q_hi[32] t_lo[40] t_hi[40] */
#ifndef BRX_PL_TBC
#define BRX_PL_TBC 16                                 /* lane aligner: traceback columns fetched per round */
#endif
#define BRX_PL_TB_UNITS ((uint64_t)(BRX_PL_TMAX + 34) * BRX_PL_W * 64)   /* uint2 units of move codes per lane-aligning wave */

struct PQ {                                           /* one queue set; zeroed by the host before the launch */
    uint32_t fresh_next, pad0;
    int32_t lane_avail; uint32_t lane_head, lane_tail, pad1;
    int32_t ret_avail; uint32_t ret_head, ret_tail, pad2;
    uint32_t pad3[6];
};
struct PQGlobal {                                     /* behind the sets */
    uint32_t finished;                                /* reads that have left the stage */
    uint32_t lane_batches, lane_windows, inplace_windows, steals, pad[11];
};

#ifdef __HIP_EMU__
#define BRX_SPIN() emu::spin_yield()
#define BRX_SPIN_LONG() emu::spin_yield()
#define BRX_DRAIN() ((void)0)
#define BRX_XCC_ID() (blockIdx.x % BRX_PQ_NX)
#define BRX_PRIO(n) ((void)0)
#else
#define BRX_PRIO(n) __builtin_amdgcn_s_setprio(n)
#define BRX_SPIN() __builtin_amdgcn_s_sleep(16)
#define BRX_SPIN_LONG() __builtin_amdgcn_s_sleep(127)
/* inline asm: the compiler may drop a builtin wait it believes redundant (MI355X_MICROARCH.md, compiler hazard) */
#define BRX_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
__device__ __forceinline__ uint32_t brx_xcc_id_() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v)); return v; }
#define BRX_XCC_ID() (brx_xcc_id_() % BRX_PQ_NX)
#endif

#define BRX_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define BRX_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

__device__ __forceinline__ uint32_t ps_slice_n(uint32_t n_reads, uint32_t x) { return n_reads > x ? (n_reads - x + BRX_PQ_NX - 1) / BRX_PQ_NX : 0u; }

/* loop state through agent-scope accesses (a struct load of a wave-uniform address may go through the scalar cache) */
static_assert(sizeof(MS) == 80, "MS is moved as ten 8-byte words");
__device__ __forceinline__ MS ms_load(const MS *p) {
    union { MS m; uint64_t w[10]; } u;
    const uint64_t *q = reinterpret_cast<const uint64_t *>(p);
#pragma unroll
    for (int i = 0; i < 10; ++i) u.w[i] = BRX_LD(q + i);
    return u.m;
}
__device__ __forceinline__ void ms_store(MS *p, const MS &m) {          /* one lane calls */
    union { MS m; uint64_t w[10]; } u;
    u.m = m;
    uint64_t *q = reinterpret_cast<uint64_t *>(p);
#pragma unroll
    for (int i = 0; i < 10; ++i) BRX_ST(q + i, u.w[i]);
}

/* ---- ticket rings ------------------------------------------------------------------------------------------- */
/* take `want` completed pushes from the semaphore (all lanes call; wave-uniform result) */
__device__ __forceinline__ bool ps_take(int32_t *avail, int want) {
    const int lane = lane_id();
    const int old = uni(atomicAdd(avail, lane == 0 ? -want : 0));
    if (old >= want) return true;
    atomicAdd(avail, lane == 0 ? want : 0);                             /* raced with another consumer: give it back */
    return false;
}
/* read and clear the slot of ticket t; the producer may still be between its ticket and its store.
   ps_slot_take: every calling lane has its OWN ticket.  ps_slot_take_wave: the wave holds one ticket -- every lane reads
   the slot, and only when all of them have seen the entry does lane 0 clear it. */
__device__ __forceinline__ uint32_t ps_slot_take(uint32_t *ring, uint32_t mask, uint32_t t) {
    uint32_t *slot = ring + (t & mask);
    uint32_t v = BRX_LD(slot);
    while (v == 0u) { BRX_SPIN(); v = BRX_LD(slot); }
    BRX_ST(slot, 0u);
    return v - 1u;
}
__device__ __forceinline__ uint32_t ps_slot_take_wave(uint32_t *ring, uint32_t mask, uint32_t t) {
    uint32_t *slot = ring + (t & mask);
    uint32_t v = BRX_LD(slot);
    while (v == 0u) { BRX_SPIN(); v = BRX_LD(slot); }
    v = uni(v);
    if (lane_id() == 0) BRX_ST(slot, 0u);
    return v - 1u;
}

/* =================================================================================================================
 * brx_lanes_align: up to 64 window alignments, one per LANE, band state in registers
 * =================================================================================================================
 * Same band (brx_make_geom), same cell recurrence and the same canonical traceback (up / 'I', left / 'D', diagonal) as
 * brx_wave_align; only the distance columns and matches of the path are produced (all the mutate loop uses).
 *
 * A lane holds BRX_PL_W consecutive 32-row blocks of its window: slot x = block s_lo + x, s_lo = the first block of
 * the band.  The band moves down one block every 32 columns, at a column that depends on the lane's geometry; the
 * lanes are therefore skewed against each other: in loop trip jj a lane works on ITS column j = jj - off, with off
 * chosen so that every lane's band moves exactly in the trips jj = 0 (mod 32) -- the register shift is one uniform
 * block of code instead of a dynamically indexed register file (round 2 kept the band in LDS for that reason: 45 KB
 * per wave and ~60 of its ~200 instructions per column were LDS addressing and traffic).  Query planes enter a lane's
 * registers one block per shift, the target planes as a 32-column window per shift (two funnel shifts), straight from
 * the parked planes in global memory: no LDS at all.
 *
 * What the forward pass stores per cell is the MOVE of the canonical traceback in two bits -- up = 10, left = 01,
 * diagonal on equal symbols = 00, diagonal on different symbols = 11 -- instead of {Pv, Ph}: the walk then counts
 * matches without looking at the sequences again.  Layout [trip][slot][lane]: a store instruction writes 512
 * contiguous bytes at a wave-uniform base.
 */
__device__ __forceinline__ uint32_t brx_bfe_mask(uint32_t v, int b) { return (uint32_t)((int32_t)(v << (31 - b)) >> 31); }

#ifndef BRX_LANES_ATTR
#define BRX_LANES_ATTR inline
#endif
__device__ BRX_LANES_ATTR void brx_lanes_align(const bool valid, const uint32_t *__restrict__ pl, const int Q, const int T, const int kb,
                                       uint2 *__restrict__ tbw, uint32_t *out_ncols, uint32_t *out_nmatch, bool *out_ok) {
    constexpr int W = BRX_PL_W;
    const int lane = lane_id();
    const BrxGeom g = brx_make_geom(Q > 0 ? Q : 1, T > 0 ? T : 1, kb);
    const int NS = (Q + 31) >> 5;
    const int Wb = (int)wave_max_u32(valid ? (uint32_t)((g.dhi - g.dlo) / 32 + 2) : 0u);      /* slots in use: the widest band of the wave */
    const int off = (g.dlo - 1) & 31;                       /* jj = j + off; (j + dlo - 1) >> 5 = (jj >> 5) + qb */
    const int qb = (g.dlo - 1 - off) >> 5;                  /* exact: dlo - 1 - off is a multiple of 32; negative */
    const int JJ = (int)wave_max_u32(valid ? (uint32_t)(T + off) : 0u);

    uint32_t P[W], M[W], QL[W], QH[W];
#pragma unroll
    for (int x = 0; x < W; ++x) {
        P[x] = 0xFFFFFFFFu; M[x] = 0u;                      /* cells below the band grow by +1 per row */
        const bool in = valid && x < NS;
        QL[x] = in ? pl[x] : 0u; QH[x] = in ? pl[BRX_PL_QW + x] : 0u;
    }
    int slo = 0;                                            /* block held in slot 0 */
    uint32_t TLw = 0u, THw = 0u;                            /* target planes of columns j0 .. j0 + 31, j0 = (jj & ~31) - off */
    const uint32_t *tlo = pl + 2 * BRX_PL_QW, *thi = tlo + BRX_PL_TW;
    for (int jj = 0; jj <= JJ; ++jj) {
        if ((jj & 31) == 0) {
            /* ---- the band moves down one block (lanes whose band still starts at block 0 stay) ---- */
            const int bq = (jj >> 5) + qb;
            if (valid && bq >= 1) {
#pragma unroll
                for (int x = 0; x + 1 < W; ++x) { P[x] = P[x + 1]; M[x] = M[x + 1]; QL[x] = QL[x + 1]; QH[x] = QH[x + 1]; }
                const int nb = bq + W - 1;
                P[W - 1] = 0xFFFFFFFFu; M[W - 1] = 0u;
                QL[W - 1] = nb < NS ? pl[nb] : 0u; QH[W - 1] = nb < NS ? pl[BRX_PL_QW + nb] : 0u;
                slo = bq;
            }
            /* ---- target planes of the next 32 trips: bit t = target index (jj - off - 1) + t ---- */
            const int t0 = jj - off - 1;
            const int w0 = t0 >> 5, sh = t0 & 31;
            const bool in0 = valid && w0 >= 0 && w0 < BRX_PL_TW, in1 = valid && w0 + 1 >= 0 && w0 + 1 < BRX_PL_TW;
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
        uint2 *dst = tbw + ((uint64_t)jj * (uint64_t)Wb) * 64u + (uint32_t)lane;
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
            if (on) {
                const uint32_t dX = ~(pv | Ph | Eq);        /* diagonal move on different symbols */
                dst[(uint32_t)x * 64u] = make_uint2(pv | dX, (Ph & ~pv) | dX);
            }
            hp = Ph >> 31; hm = Mh >> 31;                   /* the computed slots are 0 .. hi: every carry that is used was computed */
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);                          /* this wave's stores are visible to its loads below */

    /* ---- traceback, canonical (up, left, diagonal), BRX_PL_TBC columns fetched per round trip ----
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
        uint2 A[BRX_PL_TBC], Bv[BRX_PL_TBC];
#pragma unroll
        for (int x = 0; x < BRX_PL_TBC; ++x) {
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
        for (int x = 0; x < BRX_PL_TBC; ++x) {
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

/* =================================================================================================================
 * k_mutate_persist
 * ================================================================================================================= */
struct PsArgs {
    RS *rs; MS *msv;
    const uint32_t *order;             /* the reads of this launch, longest first; set x owns order[x + BRX_PQ_NX t] */
    uint32_t n_items;
    PQ *pq; PQGlobal *pg;
    uint32_t *lane_ring, *ret_ring;    /* [BRX_PQ_NX][ring_mask + 1] */
    uint32_t ring_mask;
    uint32_t *req_legacy, *legacy_ctr; /* reads whose window does not fit a slot: the whole-read kernel, after this one */
    const uint8_t *Fbuf; uint32_t *repl;
    uint32_t *planes;                  /* BRX_PL_WORDS per read */
    uint8_t *scr_base; uint64_t scr_bytes;      /* per wave: window bytes + traceback store of the in-place aligner */
    uint2 *lane_tb;                    /* BRX_PL_TB_UNITS per workgroup */
    uint32_t *flags; uint64_t *clk;
    uint32_t long_cycles;              /* more identity checks than this still ahead: align in place (the read is on the batch's critical path) */
    uint32_t low_water;                /* fewer reads than this left in the stage: align in place (the drain) */
    uint32_t patience;                 /* idle scans before a wave takes fewer than 64 parked windows */
    uint32_t exit_idle;                /* idle scans (with no fresh read left anywhere) before a wave leaves the launch: whatever is
                                          still in flight is finished by the waves that hold it, and the wave slot goes to the other
                                          batches' kernels instead of to a spin loop (the batch's last reads are chains of in-place checks) */
};

__global__ void __launch_bounds__(64 * BRX_PS_WAVES, 4) k_mutate_persist(BrxDev d, PsArgs A) {
    __shared__ uint16_t s_thr16[BRX_PS_THR_ROWS];
    __shared__ uint32_t s_lane_lock;
    const int lane = lane_id();
    const brx_error_model &em = d.em;
    const int k = em.k;
    const bool use_thr = em.type == 1 && em.n_rows <= BRX_PS_THR_ROWS;
    if (use_thr) for (uint32_t x = threadIdx.x; x < em.n_rows; x += blockDim.x) s_thr16[x] = (uint16_t)(em.d_self_thr[x] >> 16);
    if (threadIdx.x == 0) s_lane_lock = 0u;
    __syncthreads();
    const uint32_t wave_index = blockIdx.x * BRX_PS_WAVES + (threadIdx.x >> 6);
    const uint32_t n_reads = A.n_items;
    const uint32_t my_set = uni((uint32_t)BRX_XCC_ID());
    uint8_t *const scr = A.scr_base + (uint64_t)wave_index * A.scr_bytes;
    uint8_t *const qb = scr, *const tbuf = scr + BRX_WIN_Q;
    uint2 *const tb_inplace = reinterpret_cast<uint2 *>(scr + BRX_WIN_BYTES);
    const uint64_t tb_inplace_cap = (A.scr_bytes - BRX_WIN_BYTES) / 8;
    uint2 *const tb_lanes = A.lane_tb + (uint64_t)blockIdx.x * BRX_PL_TB_UNITS;
    uint32_t idle = 0;

    for (;;) {
        /* ---------------- find work: own set first, then the others ---------------- */
        int what = 0;                  /* 1 lane batch, 2 returned read, 3 fresh read */
        uint32_t set = 0, r = 0, t0 = 0;
        int n_lane = 0;
        bool fresh_left = false;
        for (uint32_t dx = 0; dx < BRX_PQ_NX && !what; ++dx) {
            const uint32_t x = (my_set + dx) % BRX_PQ_NX;
            PQ *q = A.pq + x;
            /* parked windows: a full wave of them, or -- after `patience` idle scans -- whatever is there */
            const int av = uni(BRX_LD(&q->lane_avail));
            const int want = av >= 64 ? 64 : (idle >= A.patience ? av : 0);
            if (want > 0) {
                const uint32_t held = uni(atomicOr(&s_lane_lock, lane == 0 ? 1u : 0u));      /* one move-code store per workgroup */
                if (!(held & 1u)) {
                    if (ps_take(&q->lane_avail, want)) {
                        t0 = uni(atomicAdd(&q->lane_head, lane == 0 ? (uint32_t)want : 0u));
                        what = 1; set = x; n_lane = want;
                        break;
                    }
                    if (lane == 0) s_lane_lock = 0u;
                }
            }
            if (uni(BRX_LD(&q->ret_avail)) > 0 && ps_take(&q->ret_avail, 1)) {
                const uint32_t t = uni(atomicAdd(&q->ret_head, lane == 0 ? 1u : 0u));
                r = ps_slot_take_wave(A.ret_ring + (uint64_t)x * (A.ring_mask + 1u), A.ring_mask, t);
                what = 2; set = x;
                break;
            }
            const uint32_t fn = ps_slice_n(n_reads, x);
            if (uni(BRX_LD(&q->fresh_next)) < fn) {
                fresh_left = true;
                const uint32_t t = uni(atomicAdd(&q->fresh_next, lane == 0 ? 1u : 0u));
                if (t < fn) { r = A.order[x + BRX_PQ_NX * t]; what = 3; set = x; break; }
            }
        }
        if (!what) {
            if (uni(BRX_LD(&A.pg->finished)) >= n_reads) break;                 /* the batch has left the stage */
            idle += 1;
            if (idle >= A.exit_idle && !fresh_left) break;                      /* nothing to take for a while: the waves that hold work finish it */
            /* back off: a scan is ~25 L2 loads, and hundreds of idle waves polling at full rate take the memory system away from
               the waves that work (3.4 us per s_sleep 127; up to ~27 us between scans) */
            for (uint32_t z = 0; z < (idle < 8u ? idle : 8u); ++z) BRX_SPIN_LONG();
            continue;
        }
        idle = 0;
        if (set != my_set && lane == 0) atomicAdd(&A.pg->steals, 1u);

        /* ---------------- 64 parked windows, one per lane ---------------- */
        if (what == 1) {
            uint32_t *lring = A.lane_ring + (uint64_t)set * (A.ring_mask + 1u);
            const bool valid = lane < n_lane;
            uint32_t rr = 0;
            if (valid) rr = ps_slot_take(lring, A.ring_mask, t0 + (uint32_t)lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                  /* ONE acquire behind the tickets: the planes are read with plain loads */
            int Q = 0, T = 0, kb = 0; uint32_t st_in = 0;
            if (valid) {
                const MS *m = A.msv + rr;
                const uint32_t a = BRX_LD(&m->win_a), b = BRX_LD(&m->win_b);
                Q = (int)(b - a); T = (int)BRX_LD(&m->tl); kb = (int)BRX_LD(&m->cost); st_in = BRX_LD(&m->status);
            }
            uint32_t ncols = 0, nmatch = 0; bool ok = false;
            BRX_PRIO(2);                                                        /* 64 reads wait for this wave */
            brx_lanes_align(valid, A.planes + (uint64_t)rr * BRX_PL_WORDS, Q, T, kb, tb_lanes, &ncols, &nmatch, &ok);
            BRX_PRIO(0);
            if (lane == 0) s_lane_lock = 0u;
            if (valid) {
                MS *m = A.msv + rr;
                BRX_ST(&m->res_ncols, ncols); BRX_ST(&m->res_nmatch, nmatch);
                if (!ok) BRX_ST(&m->status, st_in | BRX_RS_BAND);
            }
            BRX_DRAIN();
            uint32_t *rring = A.ret_ring + (uint64_t)set * (A.ring_mask + 1u);
            PQ *q = A.pq + set;
            const uint32_t tk = atomicAdd(&q->ret_tail, valid ? 1u : 0u);
            if (valid) BRX_ST(rring + (tk & A.ring_mask), rr + 1u);
            BRX_DRAIN();
            atomicAdd(&q->ret_avail, lane == 0 ? n_lane : 0);
            if (lane == 0) { atomicAdd(&A.pg->lane_batches, 1u); atomicAdd(&A.pg->lane_windows, (uint32_t)n_lane); }
            continue;
        }

        /* ---------------- one read: the loop until it is finished or parked ---------------- */
        const RS s = A.rs[r];
        if (s.n == 0) { atomicAdd(&A.pg->finished, lane == 0 ? 1u : 0u); continue; }
        const uint64_t t_begin = __builtin_amdgcn_s_memtime();
        MS ms;
        if (what == 2) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                  /* behind the ticket: replacement words written by other CUs */
            ms = ms_load(A.msv + r);
        }
        else { memset(&ms, 0, sizeof(ms)); }
        const uint64_t read = d.first_read + r;
        const uint32_t n = s.n;
        const uint8_t *F = A.Fbuf + s.F_off;
        uint32_t *rp = A.repl + s.F_off;
        const double target = s.target;
        const double dn = (double)n;
        const uint64_t max_i = (uint64_t)n - 1 - (uint64_t)k;
        const double need = dn * (1.0 - target);
        const uint64_t loop_cap = 100ull * (uint64_t)n;

        double errors = 0.0;
        uint64_t loops = 0;
        uint32_t change = 0, nalign = 0;
        uint32_t st_extra = ms.status;
        bool parked = false, gone = false;
        for (;;) {                                   /* one trip per in-place alignment of this read */
            errors = 0.0; loops = 0; change = 0; nalign = 0;
            bool resume = ms.phase == 1u;
            st_extra = ms.status;
            parked = false;
            bool inplace = false;
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
                uint32_t rep[16];
#pragma unroll
                for (int jx = 0; jx < 16; ++jx) rep[jx] = 0;
                bool live = false;
                uint64_t ipos = 0;
                if ((uint32_t)lane < B) {
                    uint32_t w[4];
                    brx_draw4(d.seed, read, BRX_ST_MUT, loops + (uint64_t)lane, w);
                    ipos = brx_mulhi64(((uint64_t)w[1] << 32) | w[0], max_i + 1);
                    uint8_t kmer[16];
#pragma unroll
                    for (int jx = 0; jx < 16; ++jx) kmer[jx] = jx < k ? F[ipos + jx] : 0;
                    live = dev_choose_alt(em, kmer, w[2], w[3], rep, use_thr ? s_thr16 : (const uint16_t *)nullptr);
                }
                unsigned long long surv = __ballot(live);
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
                        if (lane == j) BRX_ST(&rp[i0 + (uint64_t)j], w);          /* write-through: the next segment of this read may run on another CU */
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        change += 1;
                        const uint32_t len = (w >> 24) & 0x7Fu;
                        errors += (double)(len < 2 ? 1u : len - 1u) * scale;
                        if (change % BRX_ALIGN_INTERVAL == 0) {
                            /* ---- an identity check (simulate.py:325-346) ---- */
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
                            const uint32_t tl = wave_park(em, F, rp, a, b, qb, tbuf, BRX_WIN_TMAX, &cost, &odd);
                            const uint32_t ql = b - a;
                            int route = 2;                                 /* 0 lane queue, 1 in place, 2 whole-read kernel */
                            if (tl <= BRX_WIN_TMAX) {
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                                __builtin_amdgcn_s_waitcnt(0);
                                const BrxGeom g = brx_make_geom((int)ql, (int)tl, (int)cost);
                                const int band_blocks = (g.dhi - g.dlo) / 32 + 2;
                                const bool easy = !odd && g.G == 1 && tl <= BRX_PL_TMAX && ql <= 32u * BRX_PL_QW && ql > 0 && tl > 0 && band_blocks <= BRX_PL_W;
                                /* identity checks still ahead of this read (scheduling only: results do not depend on it) */
                                const double ahead = (need - errors) / ((double)BRX_ALIGN_INTERVAL * scale);
                                const uint32_t left = n_reads - uni(BRX_LD(&A.pg->finished));
                                route = (easy && ahead <= (double)A.long_cycles && left >= A.low_water) ? 0 : 1;
                            }
                            MS o = ms;
                            o.errors = errors; o.est = est; o.round_loops = loops; o.change = change; o.nalign = nalign;
                            o.phase = route == 2 ? 3u : 1u;
                            o.surv_lane = (uint32_t)l; o.j_next = (uint32_t)(j + 1);
                            o.win_a = a; o.win_b = b; o.tl = tl; o.cost = cost; o.res_ncols = 0; o.res_nmatch = 0;
                            o.passes = ms.passes + 1; o.status = st_extra;
                            if (route == 1) { ms = o; inplace = true; }              /* stays in registers: aligned below */
                            else if (route == 0) {
                                /* the pair as 2-bit planes: 64 symbols per ballot; lane `it` keeps the words of step `it` and
                                   writes them with one 8-byte write-through store per plane */
                                uint64_t pq_lo = 0, pq_hi = 0, pt_lo = 0, pt_hi = 0;
                                for (uint32_t it = 0; 64u * it < ql; ++it) {
                                    const uint32_t x = 64u * it + (uint32_t)lane;
                                    const uint32_t c = x < ql ? qb[x] : 0u;
                                    const unsigned long long lo = __ballot(c & 1u), hi = __ballot(c & 2u);
                                    if ((uint32_t)lane == it) { pq_lo = lo; pq_hi = hi; }
                                }
                                for (uint32_t it = 0; 64u * it < tl; ++it) {
                                    const uint32_t x = 64u * it + (uint32_t)lane;
                                    const uint32_t c = x < tl ? tbuf[x] : 0u;
                                    const unsigned long long lo = __ballot(c & 1u), hi = __ballot(c & 2u);
                                    if ((uint32_t)lane == it) { pt_lo = lo; pt_hi = hi; }
                                }
                                uint64_t *pl64 = reinterpret_cast<uint64_t *>(A.planes + (uint64_t)r * BRX_PL_WORDS);
                                if (lane < BRX_PL_QW / 2) { BRX_ST(pl64 + lane, pq_lo); BRX_ST(pl64 + BRX_PL_QW / 2 + lane, pq_hi); }
                                if (lane < BRX_PL_TW / 2) { BRX_ST(pl64 + BRX_PL_QW + lane, pt_lo); BRX_ST(pl64 + BRX_PL_QW + BRX_PL_TW / 2 + lane, pt_hi); }
                                if (lane == 0) ms_store(A.msv + r, o);
                                BRX_DRAIN();
                                PQ *q = A.pq + set;
                                const uint32_t tk = uni(atomicAdd(&q->lane_tail, lane == 0 ? 1u : 0u));
                                if (lane == 0) BRX_ST(A.lane_ring + (uint64_t)set * (A.ring_mask + 1u) + (tk & A.ring_mask), r + 1u);
                                BRX_DRAIN();
                                atomicAdd(&q->lane_avail, lane == 0 ? 1 : 0);
                            } else {
                                if (lane == 0) {
                                    ms_store(A.msv + r, o);
                                    A.req_legacy[atomicAdd(A.legacy_ctr, 1u)] = r;
                                }
                                gone = true;
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
            if (parked && inplace) {
                /* align the window here, at the top level where only MS is live, and resume the same read */
                int ncols = 0, nmatch = 0; bool nospace = false;
                const bool ok = brx_wave_align<1, 1, true>(qb, (int)(ms.win_b - ms.win_a), tbuf, (int)ms.tl, (int)ms.cost, tb_inplace, tb_inplace_cap, nullptr,
                                                           &ncols, &nmatch, &nospace);
                ms.res_ncols = (uint32_t)ncols; ms.res_nmatch = (uint32_t)nmatch;
                if (!ok && !nospace) ms.status |= BRX_RS_BAND;
                if (nospace && lane == 0) { atomicOr(&A.flags[0], 1u); A.flags[8] = r; A.flags[9] = ms.win_b - ms.win_a; A.flags[10] = ms.tl; A.flags[11] = ms.cost; }
                if (lane == 0) atomicAdd(&A.pg->inplace_windows, 1u);
                continue;
            }
            break;
        }
        uint64_t *ck = A.clk + (uint64_t)r * 8;
        if (parked) {
            if (lane == 0) atomicAdd((unsigned long long *)&ck[0], (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin));
            if (gone) atomicAdd(&A.pg->finished, lane == 0 ? 1u : 0u);         /* the whole-read kernel takes it from here */
            continue;
        }
        /* epilogue: lengths of the mutated read, trims (simulate.py:348-349), proven distance bound */
        __builtin_amdgcn_s_waitcnt(0);
        uint32_t cost = 0;
        const uint32_t m = wave_join(em, F, rp, 0, n, nullptr, &cost);
        uint32_t st = 0, et = 0;
        if (lane < k) { st = rep_len(rp[lane]); et = rep_len(rp[n - k + lane]); }
        st = wave_sum(st); et = wave_sum(et);
        if (lane == 0) {
            RS *o = &A.rs[r];
            o->status = s.status | st_extra; o->m = m; o->ub = cost; o->start_trim = st; o->end_trim = et;
            o->loops = (uint32_t)loops; o->changes = change; o->naligns = nalign;
            o->units = 0;                                          /* sized by k_fin_join */
            A.msv[r].phase = 2u;
            atomicAdd((unsigned long long *)&ck[0], (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin));
            ck[1] = ms.passes;
        }
        atomicAdd(&A.pg->finished, lane == 0 ? 1u : 0u);
    }
}

#endif /* BRX_PERSIST_H */

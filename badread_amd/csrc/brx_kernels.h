/*
 * brx_kernels.h -- HIP kernels of the simulate hot path (gfx950, wave64).  Included once by
 * brx_hip.hip.  Reference functions each kernel replaces are cited at the kernel.
 *
 * Work decomposition ("one read per wavefront"):
 *   k_plan      1 lane  = 1 read   sequential draws of build_fragment (tiny, divergent)
 *   k_build     1 wave  = 1 read   fragment bytes from the 2-bit reference, coalesced
 *   k_mutate_seg / k_win_lane / k_win_wave   (brx_mutate.h) the mutate loop as passes: 64 k-mer proposals per
 *                                  step, survivors applied in lane order, every 25th change a window alignment
 *   k_mutate    1 wave  = 1 read   the same loop run whole by one wave with inline alignments: only for reads
 *                                  whose window does not fit a pass slot
 *   k_fin_join  1 wave  = 1 read   join of the mutated read, band class, traceback size
 *   k_fin_align 1 wave  = 1 read   banded Myers + traceback, windowed store (one instantiation per band class)
 *   k_fin_qscore 1 wave = 1 read   qscore windows -> quals
 *   k_emit      1 wave  = 1 read   FASTQ bytes
 * Waves of the heavy kernels are persistent and pull reads (longest first) from a device queue.
 */
#ifndef BRX_KERNELS_H
#define BRX_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/brx.h"
#include "../../include/brx_spec.h"
#include "brx_align.h"

#define BRX_ALIGN_INTERVAL 25     /* settings.ALIGNMENT_INTERVAL */
#define BRX_ALIGN_SIZE 1000       /* settings.ALIGNMENT_SIZE     */
#ifndef BRX_MAX_BASE_SEGS
#define BRX_MAX_BASE_SEGS 64      /* base segments a lane keeps in private memory; longer lists continue in a global overflow list */
#endif
#define BRX_OVF_SEGS 4096         /* segments per overflow list */
#define BRX_OVF_LISTS 64          /* overflow lists per batch (two per overflowing read: the planner runs twice) */

enum { SEG_REF = 0, SEG_ADAPTER = 1, SEG_RANDOM = 2, SEG_JUNK = 3 };
enum { PC_JUNK = 0, PC_RANDOM = 1, PC_REAL = 2, PC_HAIRPIN = 3 };

struct PSeg { uint32_t w0, start, len, dst; };                 /* w0 = type | b<<2 (3 bits) | a<<5 */
struct PPiece { uint32_t w0, left_over; uint64_t start, end; }; /* w0 = type | strand<<2 | contig<<3 */

/* per-read working state, one per read of the batch */
struct RS {
    uint32_t status, n_segs, n_pieces, frag_len;
    uint32_t n, m, ub, start_trim;
    uint32_t end_trim, seg_off, piece_off, n_cols;
    uint32_t n_match, loops, changes, naligns;
    uint32_t seq_len, rec_len, hdr_len, klass;     /* klass: band class of the final alignment (words per lane) | BRX_KL_RETRY, set by k_fin_join */
    uint64_t F_off, seq_off, ops_off, units, tb_off, rec_off;
    double target, qerr;
};

struct BrxDev {
    brx_reference ref;
    brx_error_model em;
    brx_qscore_model qm;
    brx_sim_params p;
    uint64_t seed, first_read;
    uint32_t n_reads, raw_mode;      /* raw_mode: sequence_fragments (no plan, raw output) */
    int tb_hmul;                     /* window of the final traceback store: H = tb_hmul sqrt(ub) + 24 rows (brx_make_geom); 0 = full */
    PSeg *plan_ovf;                  /* BRX_OVF_LISTS x BRX_OVF_SEGS: continuation of base-segment lists longer than BRX_MAX_BASE_SEGS */
    uint32_t *plan_ovf_ctr;
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ uint32_t wave_bcast_u32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
__device__ __forceinline__ uint64_t wave_bcast_u64(uint64_t v, int src) {
    uint32_t lo = wave_bcast_u32((uint32_t)v, src), hi = wave_bcast_u32((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
/* A value every lane of the wave holds identically (queue index, lengths, loop flags): moving it
 * to an SGPR lets the compiler keep the control flow that depends on it scalar (s_cbranch) instead
 * of exec-masked, and frees the VGPRs.  Lane 0 must be active. */
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool uni(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }
__device__ __forceinline__ uint64_t uni(uint64_t v) {
    uint32_t lo = uni((uint32_t)v), hi = uni((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

/* Pop one index from a device work queue for the whole wave.  Deliberately NOT written as
 * `if (lane == 0) v = atomicAdd(q, 1); v = broadcast(v)`: hipcc threads the lane-0 branch through
 * the enclosing persistent loop, after which lanes 1..63 run the loop body on their own with a
 * stale index and every wave-level operation (shuffles, ballots) sees a partial exec mask --
 * observed as a hang of k_align_batch on gfx950.  Every lane takes part in the atomic instead
 * (lanes 1..63 add zero; the backend's atomic optimizer folds it into one atomic per wave). */
__device__ __forceinline__ uint32_t wave_pop(uint32_t *queue) {
    uint32_t old = atomicAdd(queue, lane_id() == 0 ? 1u : 0u);
    return uni(old);
}

/* inclusive scan over the 64 lanes */
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
    return v;
}

/* =============================================================================================
 * k_plan: build_fragment (simulate.py:91-115) and everything it draws, one lane per read.
 * Two passes with identical draws: COUNT sizes the per-read segment/piece lists, FILL writes them.
 * ========================================================================================== */
struct CountEmit {
    uint32_t nseg, npiece; uint64_t len;
    __device__ void seg(uint32_t type, uint32_t a, uint32_t b, uint32_t start, uint32_t l) { if (l) { nseg++; len += l; } }
    __device__ void piece(uint32_t type, uint32_t contig, uint32_t strand, uint64_t s, uint64_t e, uint32_t lo) { npiece++; }
};
struct WriteEmit {
    PSeg *segs; PPiece *pieces; uint32_t nseg, npiece; uint64_t len;
    __device__ void seg(uint32_t type, uint32_t a, uint32_t b, uint32_t start, uint32_t l) {
        if (!l) return;
        PSeg s; s.w0 = type | (b << 2) | (a << 5); s.start = start; s.len = l; s.dst = (uint32_t)len;
        segs[nseg++] = s; len += l;
    }
    __device__ void piece(uint32_t type, uint32_t contig, uint32_t strand, uint64_t s, uint64_t e, uint32_t lo) {
        PPiece p; p.w0 = type | (strand << 2) | (contig << 3); p.left_over = lo; p.start = s; p.end = e;
        pieces[npiece++] = p;
    }
};

/* The pieces of a read before glitches (simulate.py:94-113): adapters, fragments (two segments when a circular contig
 * wraps or a hairpin turns), chimera joins -- the reference's chimera loop is unbounded (simulate.py:101-110).  The first
 * BRX_MAX_BASE_SEGS segments live in the lane's private memory; a longer list (a read with ~20+ chimera joins: one in
 * 10^6 at --chimeras 50) continues in one of the batch's global overflow lists, taken with an atomic on first use. */
struct BaseList {
    PSeg s[BRX_MAX_BASE_SEGS]; int n; uint64_t len; bool overflow;
    PSeg *ext; PSeg *ovf_pool; uint32_t *ovf_ctr;
    __device__ void push(uint32_t type, uint32_t a, uint32_t b, uint32_t start, uint32_t l) {
        if (!l) return;
        PSeg v; v.w0 = type | (b << 2) | (a << 5); v.start = start; v.len = l; v.dst = 0;
        if (n < BRX_MAX_BASE_SEGS) s[n] = v;
        else {
            if (!ext && !overflow && ovf_pool) {
                const uint32_t slot = atomicAdd(ovf_ctr, 1u);
                if (slot < BRX_OVF_LISTS) ext = ovf_pool + (size_t)slot * BRX_OVF_SEGS;
            }
            if (!ext || n - BRX_MAX_BASE_SEGS >= BRX_OVF_SEGS) { overflow = true; return; }
            ext[n - BRX_MAX_BASE_SEGS] = v;
        }
        ++n; len += l;
    }
    __device__ PSeg at(int i) const { return i < BRX_MAX_BASE_SEGS ? s[i] : ext[i - BRX_MAX_BASE_SEGS]; }
};

/* fragment_lengths.py:47-52 */
__device__ inline uint64_t plan_fragment_length(const brx_sim_params &p, brx_rng *g) {
    if (p.frag_stdev == 0.0) return (uint64_t)brx_round_half_even(p.frag_mean);
    double v = brx_std_gamma(g, p.gamma_k) * p.gamma_t;
    int64_t L = brx_round_half_even(v);
    return (uint64_t)(L < 1 ? 1 : L);
}

/* simulate.py:183-246 */
template <class E>
__device__ bool plan_real_fragment(const BrxDev &d, brx_rng *g, uint64_t length, BaseList &base, E &em) {
    const brx_reference &r = d.ref;
    uint32_t contig = 0;
    if (r.n_contigs > 1) {
        double x = brx_next_double(g) * r.total_weight;
        while (contig < r.n_contigs - 1 && !(r.d_cum_weight[contig] > x)) ++contig;
    }
    uint32_t strand = (brx_next_double(g) < 0.5) ? 0u : 1u;
    brx_contig ct = r.d_contigs[contig];
    bool circular = ct.flags & 1u;
    bool hairpin = strand == 0 ? ((ct.flags >> 2) & 1u) : ((ct.flags >> 1) & 1u);
    uint64_t len_c = ct.length;
    if (length >= len_c && !circular && !hairpin) {
        em.piece(PC_REAL, contig, strand, 0, len_c, 0);
        base.push(SEG_REF, contig, strand, 0, (uint32_t)len_c);
        return true;
    }
    if (length > len_c && circular) return false;
    uint64_t start = brx_next_below(g, len_c);
    uint64_t end = start + length;
    if (circular) {
        em.piece(PC_REAL, contig, strand, start, end, 0);
        if (end <= len_c) base.push(SEG_REF, contig, strand, (uint32_t)start, (uint32_t)length);
        else {
            base.push(SEG_REF, contig, strand, (uint32_t)start, (uint32_t)(len_c - start));
            base.push(SEG_REF, contig, strand, 0, (uint32_t)(end - len_c));
        }
        return true;
    }
    if (end > len_c) {
        if (hairpin) {
            uint64_t fwd = len_c - start;
            uint64_t left_over = length - fwd < fwd ? length - fwd : fwd;
            em.piece(PC_HAIRPIN, contig, strand, start, len_c, (uint32_t)left_over);
            base.push(SEG_REF, contig, strand, (uint32_t)start, (uint32_t)fwd);
            base.push(SEG_REF, contig, strand ^ 1u, 0, (uint32_t)left_over);
            return true;
        }
        end = len_c;
    }
    em.piece(PC_REAL, contig, strand, start, end, 0);
    base.push(SEG_REF, contig, strand, (uint32_t)start, (uint32_t)(end - start));
    return true;
}

/* simulate.py:148-165 */
template <class E>
__device__ bool plan_get_fragment(const BrxDev &d, brx_rng *g, BaseList &base, E &em, uint32_t *next_serial) {
    const brx_sim_params &p = d.p;
    uint64_t length = plan_fragment_length(p, g);
    double u = brx_next_double(g);
    if (u < p.junk_rate) {
        uint32_t unit_len = 1u + (uint32_t)brx_next_below(g, 5);
        uint32_t unit = 0;
        for (uint32_t i = 0; i < unit_len; ++i) unit |= (uint32_t)brx_next_below(g, 4) << (2 * i);
        em.piece(PC_JUNK, 0, 0, 0, 0, 0);
        base.push(SEG_JUNK, unit, unit_len, 0, (uint32_t)length);
        return true;
    }
    if (u < p.junk_rate + p.random_rate) {
        em.piece(PC_RANDOM, 0, 0, 0, 0, 0);
        base.push(SEG_RANDOM, (*next_serial)++, 0, 0, (uint32_t)length);
        return true;
    }
    for (int attempt = 0; attempt < 1000; ++attempt)
        if (plan_real_fragment(d, g, length, base, em)) return true;
    return false;
}

template <class E>
__device__ void plan_copy_range(const BaseList &base, uint64_t a, uint64_t b, E &em) {
    uint64_t pos = 0;
    for (int s = 0; s < base.n && pos < b; ++s) {
        const PSeg sg = base.at(s);
        uint64_t lo = pos, hi = pos + sg.len;
        pos = hi;
        if (hi <= a) continue;
        uint64_t x0 = a > lo ? a : lo, x1 = b < hi ? b : hi;
        em.seg(sg.w0 & 3u, sg.w0 >> 5, (sg.w0 >> 2) & 7u, sg.start + (uint32_t)(x0 - lo), (uint32_t)(x1 - x0));
    }
}

template <class E>
__device__ void plan_read(const BrxDev &d, uint64_t read, E &em, uint32_t *status, double *target) {
    const brx_sim_params &p = d.p;
    brx_rng g;
    brx_rng_init(&g, d.seed, read, BRX_ST_PLAN);
    uint32_t next_serial = 2;
    BaseList base; base.n = 0; base.len = 0; base.overflow = false;
    base.ext = nullptr; base.ovf_pool = d.plan_ovf; base.ovf_ctr = d.plan_ovf_ctr;
    *status = 0; *target = 0.0;

    if (p.start_adapter_len > 0 && p.start_rate != 0.0 && p.start_amount != 0.0) {      /* simulate.py:361-370 */
        if (brx_next_double(&g) < p.start_rate) {
            if (p.start_amount == 1.0) base.push(SEG_ADAPTER, 0, 0, 0, p.start_adapter_len);
            else {
                double f = brx_beta(&g, 2.0 * p.start_amount, 2.0 - 2.0 * p.start_amount);
                uint32_t L = (uint32_t)((double)p.start_adapter_len * f);
                base.push(SEG_ADAPTER, 0, 0, p.start_adapter_len - L, L);
            }
        }
    }
    bool ok = plan_get_fragment(d, &g, base, em, &next_serial);
    while (ok && brx_next_double(&g) < p.chimera_rate) {                                 /* simulate.py:101-110 */
        if (brx_next_double(&g) < 0.25) base.push(SEG_ADAPTER, 1, 0, 0, p.end_adapter_len);
        if (brx_next_double(&g) < 0.25) base.push(SEG_ADAPTER, 0, 0, 0, p.start_adapter_len);
        ok = plan_get_fragment(d, &g, base, em, &next_serial);
    }
    if (!ok) { *status |= BRX_RS_NOFRAG; return; }
    if (p.end_adapter_len > 0 && p.end_rate != 0.0 && p.end_amount != 0.0) {            /* simulate.py:373-381 */
        if (brx_next_double(&g) < p.end_rate) {
            if (p.end_amount == 1.0) base.push(SEG_ADAPTER, 1, 0, 0, p.end_adapter_len);
            else {
                double f = brx_beta(&g, 2.0 * p.end_amount, 2.0 - 2.0 * p.end_amount);
                uint32_t L = (uint32_t)((double)p.end_adapter_len * f);
                base.push(SEG_ADAPTER, 1, 0, 0, L);
            }
        }
    }
    if (base.overflow) *status |= BRX_RS_TOO_MANY_SEGS;
    uint64_t base_len = base.len;
    if (p.glitch_rate == 0.0) plan_copy_range(base, 0, base_len, em);                   /* simulate.py:459-482 */
    else {
        double p_rate = p.glitch_rate > 1.0 ? 1.0 / p.glitch_rate : 1.0;
        double p_size = p.glitch_size > 1.0 ? 1.0 / p.glitch_size : 1.0;
        double p_skip = p.glitch_skip > 1.0 ? 1.0 / p.glitch_skip : 1.0;
        uint64_t i = 0;
        for (;;) {
            uint64_t dist = (uint64_t)brx_geometric(&g, p_rate);
            uint64_t e = i + dist < base_len ? i + dist : base_len;
            plan_copy_range(base, i, e, em);
            i += dist;
            if (i >= base_len) break;
            if (p.glitch_size > 0.0) {
                uint64_t sz = (uint64_t)brx_geometric(&g, p_size);
                em.seg(SEG_RANDOM, next_serial++, 0, 0, (uint32_t)sz);
            }
            if (p.glitch_skip > 0.0) i += (uint64_t)brx_geometric(&g, p_skip);
            if (i >= base_len) break;
        }
    }
    if (p.identity_mode == 0) *target = p.id_max;                                       /* identities.py:76-93 */
    else if (p.identity_mode == 1) *target = p.id_max * brx_beta(&g, p.id_a, p.id_b);
    else {
        for (;;) {
            double q = p.id_a + p.id_b * brx_normal(&g);
            double id = 1.0 - brx_exp((-q / 10.0) * 2.302585092994046);
            if (id >= 0.0 && id <= 100.0) { *target = id; break; }
        }
    }
}

__global__ void __launch_bounds__(64) k_plan_count(BrxDev d, RS *rs) {
    uint32_t r = blockIdx.x * 64 + threadIdx.x;
    if (r >= d.n_reads) return;
    CountEmit em; em.nseg = 0; em.npiece = 0; em.len = 0;
    uint32_t status; double target;
    plan_read(d, d.first_read + r, em, &status, &target);
    RS s; memset(&s, 0, sizeof(s));
    s.status = status; s.n_segs = em.nseg; s.n_pieces = em.npiece; s.frag_len = (uint32_t)em.len;
    s.n = (status & BRX_RS_NOFRAG) ? 0u : (uint32_t)em.len + 2u * (uint32_t)d.em.k;
    s.target = target;
    rs[r] = s;
}

__global__ void __launch_bounds__(64) k_plan_fill(BrxDev d, RS *rs, PSeg *segs, PPiece *pieces) {
    uint32_t r = blockIdx.x * 64 + threadIdx.x;
    if (r >= d.n_reads) return;
    WriteEmit em; em.segs = segs + rs[r].seg_off; em.pieces = pieces + rs[r].piece_off;
    em.nseg = 0; em.npiece = 0; em.len = 0;
    uint32_t status; double target;
    plan_read(d, d.first_read + r, em, &status, &target);
}

/* exclusive scans after the COUNT pass, one wave.  totals: [0]=segs [1]=pieces [2]=F bytes */
__global__ void __launch_bounds__(64) k_scan_plan(uint32_t n_reads, RS *rs, uint64_t *totals) {
    const int lane = lane_id();
    uint64_t seg_run = 0, piece_run = 0, f_run = 0;
    for (uint32_t base = 0; base < n_reads; base += 64) {
        uint32_t r = base + lane;
        uint32_t ns = 0, np = 0, fb = 0;
        if (r < n_reads) { ns = rs[r].n_segs; np = rs[r].n_pieces; fb = rs[r].n ? ((rs[r].n + 16u + 15u) & ~15u) : 0u; }
        uint32_t is = wave_incl_scan(ns), ip = wave_incl_scan(np);
        /* F bytes can exceed 32 bits over a batch: scan in 16-byte units */
        uint32_t iff = wave_incl_scan(fb >> 4);
        if (r < n_reads) {
            rs[r].seg_off = (uint32_t)(seg_run + is - ns);
            rs[r].piece_off = (uint32_t)(piece_run + ip - np);
            rs[r].F_off = f_run + ((uint64_t)(iff - (fb >> 4)) << 4);
        }
        seg_run += wave_bcast_u32(is, 63); piece_run += wave_bcast_u32(ip, 63);
        f_run += (uint64_t)wave_bcast_u32(iff, 63) << 4;
    }
    if (lane == 0) { totals[0] = seg_run; totals[1] = piece_run; totals[2] = f_run; }
}

/* Processing order, most work first: counting sort on the EXPECTED NUMBER OF CHANGES n (1 - target identity) in steps of 32
 * (1024 buckets), one wave.  A read's mutate loop runs one alignment cycle per 25 changes, and its final alignment's band is as
 * wide as its changes: both the passes a read needs and the class of its final alignment follow the changes, not the length
 * (rounds 1-3 sorted by length: a 22 kb read at 85 % identity -- 130 cycles -- sat among reads that are done after 35). */
#define BRX_ORDER_BUCKETS 1024
#define BRX_ORDER_IDBINS 16
/* Secondary key (round 6): the error rate 1 - target in bins of 2 %.  k_mut_lanes keeps 64 neighbours of this order in one wave and its
   lane aligner computes, for every lane, as many band blocks per column as the wave's WIDEST window needs -- a window's band is its
   edit bound, ~1000 (1 - identity): among 64 reads taken as the beta law deals them there is nearly always one below 90 %.  Inside a
   bucket of equal expected changes (equal work: what the primary key is for) the reads are therefore listed by error rate. */
__device__ __forceinline__ uint32_t brx_order_key(const RS &s) {
    const double e = (double)s.n * (1.0 - s.target);
    uint32_t key = e > 0.0 ? (uint32_t)(e * (1.0 / 32.0)) : 0u;
    if (s.n == 0) key = 0;
    key = key > BRX_ORDER_BUCKETS - 1u ? BRX_ORDER_BUCKETS - 1u : key;
    const double er = (1.0 - s.target) * 50.0;
    uint32_t idb = er > 0.0 ? (uint32_t)er : 0u;
    idb = idb > BRX_ORDER_IDBINS - 1u ? BRX_ORDER_IDBINS - 1u : idb;
    return key * BRX_ORDER_IDBINS + (s.n == 0 ? 0u : idb);
}
__global__ void __launch_bounds__(64) k_order(uint32_t n_reads, const RS *rs, uint32_t *order) {
    constexpr uint32_t NB = BRX_ORDER_BUCKETS * BRX_ORDER_IDBINS;
    __shared__ uint32_t hist[NB];
    __shared__ uint32_t part[64];
    const int lane = lane_id();
    for (uint32_t b = lane; b < NB; b += 64) hist[b] = 0;
    __syncthreads();
    for (uint32_t r = lane; r < n_reads; r += 64) atomicAdd(&hist[NB - 1u - brx_order_key(rs[r])], 1u);
    __syncthreads();
    {   /* exclusive prefix over the buckets: every lane its own stretch, then the stretches' starts */
        constexpr uint32_t PER = NB / 64;
        uint32_t sum = 0;
        for (uint32_t b = (uint32_t)lane * PER; b < ((uint32_t)lane + 1u) * PER; ++b) sum += hist[b];
        part[lane] = sum;
        __syncthreads();
        uint32_t run = 0;
        for (int l = 0; l < lane; ++l) run += part[l];
        for (uint32_t b = (uint32_t)lane * PER; b < ((uint32_t)lane + 1u) * PER; ++b) { const uint32_t c = hist[b]; hist[b] = run; run += c; }
    }
    __syncthreads();
    for (uint32_t r = lane; r < n_reads; r += 64) {
        uint32_t slot = atomicAdd(&hist[NB - 1u - brx_order_key(rs[r])], 1u);
        order[slot] = r;
    }
}

/* =============================================================================================
 * k_build: the string slicing of get_real_fragment (simulate.py:206-246), reverse_complement
 * (misc.py:56-71), junk/random/adapter pieces and the two random pads of sequence_fragment
 * (simulate.py:260).  One wave per read; consecutive lanes take consecutive bases.
 * ========================================================================================== */
__device__ inline uint32_t ref_code_acgt(const brx_reference &r, const brx_contig &ct, uint32_t strand, uint32_t pos) {
    uint64_t f = strand == 0 ? (uint64_t)pos : (uint64_t)ct.length - 1 - pos;
    uint64_t gidx = ct.base_off + f;
    uint32_t code = (r.d_packed[gidx >> 4] >> (2 * (gidx & 15))) & 3u;
    return strand == 0 ? code : (3u - code);       /* complement of ACGT codes 0..3 */
}

/* F2 / changed map (round 4): besides the bytes, the fragment is kept as 2-bit codes, sixteen bases per word, base p of
 * a read in bits 30 - 2 (p % 16) of word p / 16 (most significant first, so that a k-mer read across two words is a
 * number whose digits are in the order error_model.py:135-160 indexes its table by), at word F_off / 16 of F2buf; the word
 * behind the last one says whether the read holds a symbol outside ACGT (then the mutate loop keeps to the bytes).
 * Cbuf (same index) is a bit per base: set when the position has been replaced (repl[p] != 0).  The proposal rounds of
 * the proposal rounds read k-mers from THESE, and brx_apply_read / brx_lane_park the changed bits (brx_passes.h), instead of
 * seven byte loads and a 4-byte word at a random position of a 15-60 KB read. */
__global__ void __launch_bounds__(64) k_build(BrxDev d, const RS *rs, const PSeg *segs, uint8_t *Fbuf, uint32_t *repl,
                                              uint32_t *F2buf, uint32_t *Cbuf) {
    const uint32_t r = blockIdx.x;
    const int lane = lane_id();
    const RS s = rs[r];
    if (s.n == 0) return;
    const uint64_t read = d.first_read + r;
    const int k = d.em.k;
    uint8_t *F = Fbuf + s.F_off;
    uint32_t *rp = repl + s.F_off;
    const uint32_t n = s.n;
    for (uint32_t x = lane; x < n + 16; x += 64) { if (x < n) rp[x] = 0; else F[x] = 0xFF; }
    if (lane < k) {
        F[lane] = (uint8_t)brx_random_base(d.seed, read, 0, (uint64_t)lane);
        F[n - k + lane] = (uint8_t)brx_random_base(d.seed, read, 1, (uint64_t)lane);
    }
    uint8_t *dst0 = F + k;
    for (uint32_t si = 0; si < (d.raw_mode ? 0u : s.n_segs); ++si) {     /* raw mode: the fragment bytes were copied by k_copy_frags */
        const PSeg sg = segs[s.seg_off + si];
        const uint32_t type = sg.w0 & 3u, b = (sg.w0 >> 2) & 7u, a = sg.w0 >> 5;
        uint8_t *dst = dst0 + sg.dst;
        if (type == SEG_REF) {
            const brx_contig ct = d.ref.d_contigs[a];
            for (uint32_t x = lane; x < sg.len; x += 64) dst[x] = (uint8_t)ref_code_acgt(d.ref, ct, b, sg.start + x);
            /* non-ACGT runs overlapping the forward range of this segment: OTHER lanes overwrite the 2-bit codes stored
               above, so those stores must be complete first (do not lean on same-address store order across lanes) */
            if (d.ref.n_exceptions) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_waitcnt(0);
                uint64_t g0, g1;
                if (b == 0) { g0 = ct.base_off + sg.start; g1 = g0 + sg.len; }
                else { g1 = ct.base_off + ((uint64_t)ct.length - sg.start); g0 = g1 - sg.len; }
                uint32_t lo = 0, hi = d.ref.n_exceptions;
                while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (d.ref.d_exceptions[mid].end > g0) hi = mid; else lo = mid + 1; }
                for (uint32_t e = lo; e < d.ref.n_exceptions; ++e) {
                    const brx_exception ex = d.ref.d_exceptions[e];
                    if (ex.start >= g1) break;
                    uint64_t o0 = ex.start > g0 ? ex.start : g0, o1 = ex.end < g1 ? ex.end : g1;
                    uint8_t code = (uint8_t)(b == 0 ? ex.code : d.ref.comp[ex.code & 15u]);
                    for (uint64_t gg = o0 + lane; gg < o1; gg += 64) {
                        uint64_t x = b == 0 ? gg - g0 : (g1 - 1 - gg);
                        dst[x] = code;
                    }
                }
            }
        } else if (type == SEG_ADAPTER) {
            const uint8_t *ad = a == 0 ? d.p.d_start_adapter : d.p.d_end_adapter;
            for (uint32_t x = lane; x < sg.len; x += 64) dst[x] = ad[sg.start + x];
        } else if (type == SEG_RANDOM) {
            for (uint32_t x = lane; x < sg.len; x += 64) dst[x] = (uint8_t)brx_random_base(d.seed, read, a, (uint64_t)sg.start + x);
        } else {                                  /* junk: `a` = repeat unit (2 bits/base), `b` = its length */
            for (uint32_t x = lane; x < sg.len; x += 64) dst[x] = (uint8_t)((a >> (2 * ((sg.start + x) % b))) & 3u);
        }
    }
    /* ---- the same fragment as 2-bit codes, and an empty changed map ---- */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);                 /* the bytes of every lane are in place */
    uint32_t *f2 = F2buf + (s.F_off >> 4), *cm = Cbuf + (s.F_off >> 4);
    const uint32_t nw = (n + 15u) >> 4, nwc = (n + 31u) >> 5;
    for (uint32_t w = lane; w <= nw; w += 64) cm[w] = 0u;       /* changed map [0, nwc), map of the symbols outside ACGT [nwc, 2 nwc) */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    bool odd = false;
    for (uint32_t w = lane; w < nw; w += 64) {
        const uint4 q = *reinterpret_cast<const uint4 *>(F + 16u * w);       /* F + F_off is 16-byte aligned (k_scan_plan) */
        const uint32_t v[4] = {q.x, q.y, q.z, q.w};
        uint32_t word = 0, oddbits = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            /* bytes beyond n are 0xFF (set above): they count as codes, not as odd symbols */
            const uint32_t left = n - 16u * w - 4u * (uint32_t)i;      /* bases of the read from this dword on (may wrap: then none) */
            const uint32_t live = 16u * w + 4u * (uint32_t)i >= n ? 0u : left >= 4u ? 0xFFFFFFFFu : (1u << (8u * left)) - 1u;
            const uint32_t hi6 = ((v[i] & live) >> 2) & 0x3F3F3F3Fu;          /* per byte: non-zero iff the symbol is outside 0..3 */
            oddbits |= ((((hi6 + 0x3F3F3F3Fu) >> 6) & 0x01010101u) * 0x01020408u >> 24) << (4 * i);
            uint32_t x = v[i] & 0x03030303u;                           /* byte j (bits 8j) -> digit j, first base most significant */
            x = ((x & 0x3u) << 6) | ((x >> 4) & 0x30u) | ((x >> 14) & 0xCu) | (x >> 24);
            word |= x << (24 - 8 * i);
        }
        f2[w] = word;
        if (oddbits) { odd = true; atomicOr(&cm[nwc + (w >> 1)], oddbits << (16u * (w & 1u))); }
    }
    const bool any_odd = __ballot(odd) != 0ull;
    if (lane == 0) f2[nw] = any_odd ? 1u : 0u;
}

/* sequence_fragments entry: copy caller fragments behind the start pad */
__global__ void __launch_bounds__(64) k_copy_frags(BrxDev d, const RS *rs, const uint8_t *frags, const uint64_t *frag_off, uint8_t *Fbuf) {
    const uint32_t r = blockIdx.x;
    const RS s = rs[r];
    uint8_t *F = Fbuf + s.F_off + d.em.k;
    const uint8_t *src = frags + frag_off[r];
    for (uint32_t x = lane_id(); x < s.frag_len; x += 64) F[x] = src[x];
}

__global__ void __launch_bounds__(64) k_init_raw(BrxDev d, RS *rs, const uint64_t *frag_off, const double *target) {
    uint32_t r = blockIdx.x * 64 + threadIdx.x;
    if (r >= d.n_reads) return;
    RS s; memset(&s, 0, sizeof(s));
    s.frag_len = (uint32_t)(frag_off[r + 1] - frag_off[r]);
    s.n = s.frag_len + 2u * (uint32_t)d.em.k;
    s.target = target[r];
    rs[r] = s;
}

/* =============================================================================================
 * error model lookup: ErrorModel.add_errors_to_kmer / add_one_random_change
 * (error_model.py:135-176).  rep[j] = 0x80000000 | len<<24 | pool offset, or 0 if unchanged.
 * ========================================================================================== */
/* The draw of add_one_random_change split into its choices -- two divisions by the run-time k.  A call, not inline code:
 * table models reach it for a few draws in a million, and a dozen inlined copies of the divisions were a fifth of the mutate
 * kernels' instructions (which have to stay inside the instruction cache next to five other batches' kernels). */
struct BrxRc { uint32_t type, pos, rest; };
__device__ __attribute__((noinline)) BrxRc dev_random_change_split(uint32_t w3, int k) {
    BrxRc c;
    c.type = w3 % 3u;
    c.pos = (w3 / 3u) % (uint32_t)k;
    c.rest = w3 / (3u * (uint32_t)k);
    return c;
}
__device__ __forceinline__ uint32_t dev_random_change_word(const BrxRc c, uint32_t o) {
    if (c.type == 0) return 0x80000000u | (1u << 24) | (o < 4 ? ((o + 1u + c.rest % 3u) & 3u) : (c.rest & 3u));
    if (c.type == 1) {
        const uint32_t after = c.rest & 1u, nb = (c.rest >> 1) & 3u;
        const uint32_t x = after ? o : nb, y = after ? nb : o;
        return 0x80000000u | (2u << 24) | (16u + 2u * (16u * x + y));
    }
    return 0x80000000u;
}
__device__ inline void dev_random_change(const uint8_t *kmer, int k, uint32_t w3, uint32_t *rep) {
    const BrxRc c = dev_random_change_split(w3, k);
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) if (j < k) o = ((uint32_t)j == c.pos) ? kmer[j] : o;
    const uint32_t word = dev_random_change_word(c, o);
#pragma unroll
    for (int j = 0; j < 16; ++j) if (j < k) rep[j] = ((uint32_t)j == c.pos) ? word : 0u;
}

/* returns false if the k-mer is unchanged */
__device__ inline bool dev_choose_alt(const brx_error_model &em, const uint8_t *kmer, uint32_t w2, uint32_t w3, uint32_t *rep) {
    const int k = em.k;
    if (em.type == 0) { dev_random_change(kmer, k, w3, rep); return true; }
    uint32_t row = 0; bool bad = false;
#pragma unroll
    for (int j = 0; j < 16; ++j) if (j < k) { bad |= kmer[j] > 3; row = (row << 2) | (kmer[j] & 3u); }
    if (bad) { dev_random_change(kmer, k, w3, rep); return true; }
    if (w2 < em.d_self_thr[row]) return false;                   /* the common case (~93 % of draws, simulate.py:300): unchanged */
    uint32_t a0 = em.d_row_off[row], a1 = em.d_row_off[row + 1];
    if (a0 == a1) { dev_random_change(kmer, k, w3, rep); return true; }
    /* first alternative whose cumulative threshold exceeds the draw (thresholds are non-decreasing):
       binary search -- the linear walk of up to 26 dependent loads dominated the proposal rounds */
    uint32_t a = a0, hi_ = a1;
    while (a < hi_) { const uint32_t mid = (a + hi_) >> 1; if (w2 < em.d_thr[mid]) hi_ = mid; else a = mid + 1; }
    if (a == a1) {
        if (em.d_thr[a1 - 1] == 0xFFFFFFFFu) a = a1 - 1;
        else { dev_random_change(kmer, k, w3, rep); return true; }
    }
    uint32_t o = em.d_desc[a];
    uint32_t diff = (uint32_t)em.d_pool[o] | ((uint32_t)em.d_pool[o + 1] << 8);
    if (diff == 0) return false;
    uint32_t coff = o + 2u + (uint32_t)k;
#pragma unroll
    for (int j = 0; j < 16; ++j) if (j < k) {
        uint32_t len = em.d_pool[o + 2 + (uint32_t)j];
        rep[j] = ((diff >> j) & 1u) ? (0x80000000u | (len << 24) | coff) : 0u;
        coff += len;
    }
    return true;
}

/* add_one_random_change for a k-mer of ACGT codes given as its table row (digits most significant first). */
__device__ __forceinline__ void dev_random_change_row(uint32_t row, int k, uint32_t w3, uint32_t *rep) {
    const BrxRc c = dev_random_change_split(w3, k);
    const uint32_t word = dev_random_change_word(c, (row >> (2u * ((uint32_t)k - 1u - c.pos))) & 3u);
#pragma unroll
    for (int j = 0; j < 16; ++j) if (j < k) rep[j] = ((uint32_t)j == c.pos) ? word : 0u;
}

struct __attribute__((packed, aligned(4))) BrxU4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) BrxB16x3 { uint32_t x, y, z; };        /* twelve bytes behind any address */      /* four words behind a 4-byte aligned address */

/* A PROPOSAL of the mutate rounds is three words per lane (round 4; it was an array of sixteen replacement words, filled
 * by every proposing lane and read back with sixteen broadcasts per survivor):
 *   y = 0                       the k-mer stays as it is (not a survivor)
 *   y = BRX_PROP_RANDOM | pos   add_one_random_change: x = the replacement word of position pos
 *   y = BRX_PROP_TABLE | diff | long << 16   an alternative of the table: x = its descriptor's pool offset, z = the lengths of
 *                               positions 0-7 in 4 bits each (long: the lengths are read from the pool)
 * The lanes j < k of the wave expand a survivor's proposal TOGETHER (brx_prop_word): lane j derives the word of position j. */
#define BRX_PROP_TABLE 0x80000000u
#define BRX_PROP_RANDOM 0x40000000u
struct BrxProp { uint32_t x, y, z; };

__device__ __forceinline__ BrxProp brx_prop_random(const brx_error_model &em, uint32_t row, uint32_t w3) {
    const BrxRc c = dev_random_change_split(w3, em.k);
    BrxProp p; p.x = dev_random_change_word(c, (row >> (2u * ((uint32_t)em.k - 1u - c.pos))) & 3u); p.y = BRX_PROP_RANDOM | c.pos; p.z = 0u;
    return p;
}

/* ErrorModel.add_errors_to_kmer (error_model.py:135-160) for a k-mer of ACGT codes given as its table row (digits most
 * significant first, as read from F2), on the lookup-order tables (include/brx.h: d_rowx, d_altx).  Every step is ONE load:
 * the row entry (self threshold and both ends of the row's alternatives), the thresholds in blocks of eight (first
 * alternative whose cumulative threshold exceeds the draw: the alternatives of a row are few and the likely ones come
 * first), the alternative's descriptor with its lengths. */
__device__ inline BrxProp dev_propose_row(const brx_error_model &em, uint32_t row, uint32_t w2, uint32_t w3) {
    BrxProp none; none.x = none.y = none.z = 0u;
    if (em.type == 0) return brx_prop_random(em, row, w3);
    const BrxU4 e = *reinterpret_cast<const BrxU4 *>(em.d_rowx + 2u * row);
    if (w2 < e.x) return none;                                   /* the common case (~93 % of draws, simulate.py:300): unchanged */
    const uint32_t a0 = e.y, a1 = e.w;
    if (a0 == a1) return brx_prop_random(em, row, w3);
    uint32_t a = a1, last = 0u;
    for (uint32_t c = a0; c < a1 && a == a1; c += 8u) {
        const BrxU4 t0 = *reinterpret_cast<const BrxU4 *>(em.d_thr + c), t1 = *reinterpret_cast<const BrxU4 *>(em.d_thr + c + 4u);
        const uint32_t t[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        uint32_t hit = 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool in = c + (uint32_t)i < a1;
            hit |= (in && w2 < t[i]) ? (1u << i) : 0u;
            last = (c + (uint32_t)i == a1 - 1u) ? t[i] : last;
        }
        if (hit) a = c + (uint32_t)(__ffs((int)hit) - 1);
    }
    if (a == a1) {
        if (last == 0xFFFFFFFFu) a = a1 - 1u;
        else return brx_prop_random(em, row, w3);
    }
    const BrxU4 inf = *reinterpret_cast<const BrxU4 *>(em.d_altx + 4u * a);
    if ((inf.y & 0xFFFFu) == 0u) return none;
    BrxProp p; p.x = inf.x; p.y = BRX_PROP_TABLE | (inf.y & 0x1FFFFu); p.z = inf.z;
    return p;
}

/* The replacement word of position `lane` (0 for an unchanged position and for lanes >= k) of a proposal whose three words
 * are wave-uniform here (broadcast from the proposing lane). */
__device__ __forceinline__ uint32_t brx_prop_word(const brx_error_model &em, uint32_t px, uint32_t py, uint32_t pz) {
    const int lane = lane_id();
    const int k = em.k;
    if (py & BRX_PROP_RANDOM) return (uint32_t)lane == (py & 0xFFu) ? px : 0u;
    uint32_t len, before;
    if (py & 0x10000u) {                                         /* a length above 15 (or k > 8): lengths from the pool */
        len = lane < k ? (uint32_t)em.d_pool[px + 2u + (uint32_t)lane] : 0u;
        before = wave_incl_scan(len) - len;
    } else {
        const uint32_t l7 = (uint32_t)lane & 7u;
        len = lane < 8 ? (pz >> (4u * l7)) & 15u : 0u;
        const uint32_t m = pz & ((1u << (4u * l7)) - 1u);        /* the lengths of the positions before this one ... */
        const uint32_t t = (m & 0x0F0F0F0Fu) + ((m >> 4) & 0x0F0F0F0Fu);
        before = (t * 0x01010101u) >> 24;                        /* ... summed */
    }
    const bool on = lane < k && ((py >> lane) & 1u);
    return on ? (0x80000000u | (len << 24) | (px + 2u + (uint32_t)k + before)) : 0u;
}

__device__ __forceinline__ uint32_t rep_len(uint32_t w) { return w ? ((w >> 24) & 0x7Fu) : 1u; }

__device__ inline uint8_t rep_char(const brx_error_model &em, uint32_t w, uint32_t x) {
    uint32_t off = w & 0x00FFFFFFu;
    if (off < 16) return (uint8_t)off;
    if (off < BRX_POOL_PREAMBLE) { uint32_t v = (off - 16) >> 1; return (uint8_t)(x == 0 ? (v >> 4) : (v & 15u)); }
    return em.d_pool[off + x];
}

/* Proven bound on the edits one position contributes: the distance between the one-base string `orig` and its replacement of
   `len` characters -- 1 for a deletion or a substitution, len - 1 for a string that still holds the base (an insertion beside
   it), len for one that does not.  (Round 2 charged `len` for every string: a third of nanopore2023's alternatives are the base
   plus one inserted character, so the bound -- the band width of every alignment of the read -- ran ~30 % above the distance.) */
__device__ inline uint32_t rep_cost(const brx_error_model &em, uint32_t w, uint32_t orig) {
    if (!w) return 0u;
    const uint32_t l = (w >> 24) & 0x7Fu;
    if (l < 2u) return 1u;
    bool has = false;
    for (uint32_t x = 0; x < l; ++x) has |= (uint32_t)rep_char(em, w, x) == orig;
    return l - (has ? 1u : 0u);
}

/* join(new_fragment_bases[a:b]) by the whole wave: writes the characters to out (may be null for
 * sizing), returns the total length; *cost receives an upper bound on the edit distance between
 * F[a:b] and the joined string (rep_cost per changed position). */
__device__ inline uint32_t wave_join(const brx_error_model &em, const uint8_t *F, const uint32_t *repl,
                                     uint32_t a, uint32_t b, uint8_t *out, uint32_t *cost) {
    const int lane = lane_id();
    uint32_t run = 0, c = 0;
    for (uint32_t base = a; base < b; base += 64) {
        uint32_t p = base + lane;
        uint32_t w = 0, len = 0;
        if (p < b) { w = repl[p]; len = rep_len(w); if (w) c += rep_cost(em, w, F[p]); }
        uint32_t inc = wave_incl_scan(len);
        if (out && p < b) {
            uint32_t o = run + inc - len;
            if (!w) out[o] = F[p];
            else for (uint32_t x = 0; x < len; ++x) out[o + x] = rep_char(em, w, x);
        }
        run += wave_bcast_u32(inc, 63);
    }
    if (cost) *cost = wave_sum(c);
    return run;
}

/* =============================================================================================
 * k_mutate: the while-loop of sequence_fragment (simulate.py:272-346).
 * 64 iterations of the loop are PROPOSED at once (iteration t = loops + lane draws its k-mer
 * position and alternative from Philox block t); proposals that leave the k-mer unchanged only
 * advance loop_count, the others are APPLIED one by one in iteration order with the running
 * error estimate, so the result equals the sequential loop exactly.
 * Per-wave scratch (win): [0,4096) query copy, [4096, 4096+tgt_cap) target, then traceback store.
 * ========================================================================================== */
struct MutScratch { uint8_t *base; uint64_t bytes; };

__global__ void __launch_bounds__(64) k_mutate(BrxDev d, RS *rs, const uint32_t *list, const uint32_t *n_list_ptr, uint32_t *queue,
                                                uint8_t *Fbuf, uint32_t *repl, uint8_t *win_base, uint64_t win_bytes,
                                                uint32_t *flags, uint64_t *clk) {
    const int lane = lane_id();
    const brx_error_model &em = d.em;
    const int k = em.k;
    uint8_t *win = win_base + (uint64_t)blockIdx.x * win_bytes;
    const uint32_t n_list = uni(*n_list_ptr);
    for (;;) {
        const uint32_t qi = wave_pop(queue);
        if (qi >= n_list) break;
        const uint32_t r = list[qi];
        RS s = rs[r];
        if (s.n == 0) continue;
        /* whole-read fallback of the multi-pass pipeline (brx_mutate.h): start the read over */
        for (uint32_t x = lane; x < s.n; x += 64) repl[s.F_off + x] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        const uint64_t t_begin = __builtin_amdgcn_s_memtime();
        uint64_t aclk[2] = {0, 0};
        const uint64_t read = d.first_read + r;
        const uint32_t n = s.n;
        const uint8_t *F = Fbuf + s.F_off;
        uint32_t *rp = repl + s.F_off;
        const double target = s.target;
        const double dn = (double)n;
        double errors = 0.0;
        uint64_t loops = 0;
        uint32_t change = 0, nalign = 0;
        const uint64_t max_i = (uint64_t)n - 1 - (uint64_t)k;
        const double need = dn * (1.0 - target);
        const uint64_t loop_cap = 100ull * (uint64_t)n;
        bool done = need < 0.5;
        while (!done) {
            if (loops + 1 > loop_cap) { loops += 1; break; }
            double est = 1.0 - errors / dn;
            if ((double)change > 0.9 * dn || est <= target) { loops += 1; break; }
            uint64_t room = loop_cap - loops;
            uint32_t B = room < 64 ? (uint32_t)room : 64u;
            /* ---- propose ---- */
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
                live = dev_choose_alt(em, kmer, w[2], w[3], rep);
            }
            unsigned long long surv = __ballot(live);
            /* ---- apply survivors in iteration order ---- */
            while (surv) {
                int l = __ffsll((long long)surv) - 1;
                surv &= surv - 1;
                const uint64_t i0 = wave_bcast_u64(ipos, l);
                const double scale = est * brx_sqrt(est);
                for (int j = 0; j < k; ++j) {
                    uint32_t mine = 0;
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj) mine = (jj == j) ? rep[jj] : mine;
                    uint32_t w = wave_bcast_u32(mine, l);
                    if (!w) continue;
                    uint64_t pos = i0 + (uint64_t)j;
                    const uint32_t cur = uni(rp[pos]);      /* every lane has loaded before lane 0 stores below */
                    if (cur) continue;
                    if (lane == 0) rp[pos] = w;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    change += 1;
                    uint32_t len = (w >> 24) & 0x7Fu;
                    errors += (double)(len < 2 ? 1u : len - 1u) * scale;
                    if (change % BRX_ALIGN_INTERVAL == 0) {
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
                        uint32_t tl = wave_join(em, F, rp, a, b, nullptr, &cost);
                        uint32_t ql = b - a;
                        uint64_t qpad = ((uint64_t)ql + 16 + 15) & ~15ull, tpad = ((uint64_t)tl + 16 + 15) & ~15ull;
                        int ncols = 0, nmatch = 0; bool nospace = false;
                        if (qpad + tpad + 64 > win_bytes) nospace = true;
                        else {
                            uint8_t *qb = win, *tbuf = win + qpad;
                            for (uint32_t x = lane; x < ql + 16; x += 64) qb[x] = x < ql ? F[a + x] : 0xFF;
                            for (uint32_t x = lane; x < 16; x += 64) tbuf[tl + x] = 0xFE;
                            wave_join(em, F, rp, a, b, tbuf, nullptr);
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            __builtin_amdgcn_s_waitcnt(0);
                            uint2 *tb = reinterpret_cast<uint2 *>(win + qpad + tpad);
                            uint64_t cap = (win_bytes - qpad - tpad) / 8;
                            bool ok = brx_wave_align<1>(qb, (int)ql, tbuf, (int)tl, (int)cost, tb, cap, nullptr, &ncols, &nmatch, &nospace,
                                                     nullptr, aclk);
                            if (!ok && !nospace) s.status |= BRX_RS_BAND;
                        }
                        if (nospace) { if (lane == 0) atomicOr(&flags[0], 1u); }
                        double id = ncols ? (double)nmatch / (double)ncols : 0.0;
                        if (n <= BRX_ALIGN_SIZE) errors = (1.0 - id) * dn;
                        else {
                            double est_err = (1.0 - id) * dn;
                            double weight = (double)BRX_ALIGN_SIZE / dn;
                            errors = est_err * weight + errors * (1.0 - weight);
                        }
                    }
                }
                /* top-of-loop tests of the iteration that follows this survivor */
                double est2 = 1.0 - errors / dn;
                if ((double)change > 0.9 * dn || est2 <= target) { loops += (uint64_t)l + 2; done = true; break; }
                est = est2;
            }
            if (done) break;
            loops += B;
            if (B < 64) { loops += 1; break; }
        }
        /* epilogue: lengths of the mutated read, trims (simulate.py:348-349), proven distance bound */
        __builtin_amdgcn_s_waitcnt(0);
        uint32_t cost = 0;
        uint32_t m = wave_join(em, F, rp, 0, n, nullptr, &cost);
        uint32_t st = 0, et = 0;
        if (lane < k) { st = rep_len(rp[lane]); et = rep_len(rp[n - k + lane]); }
        st = wave_sum(st); et = wave_sum(et);
        if (lane == 0) {
            RS *o = &rs[r];
            o->status = s.status; o->m = m; o->ub = cost; o->start_trim = st; o->end_trim = et;
            o->loops = (uint32_t)loops; o->changes = change; o->naligns = nalign;
            o->units = 0;                                          /* sized by k_fin_join */
            uint64_t *ck = clk + (uint64_t)r * 8;
            ck[0] = __builtin_amdgcn_s_memtime() - t_begin; ck[1] = aclk[0]; ck[2] = aclk[1];
        }
    }
}

#include "brx_mutate.h"
#include "brx_passes.h"
#include "brx_model.h"
#include "brx_gzip_dev.h"

/* offsets for the final stage of one SET of reads (list[0..n): a range of the processing order), relative to the
   set's seq / ops buffers.  totals: [0]=seq bytes [1]=ops bytes */
__global__ void __launch_bounds__(64) k_scan_mut(uint32_t n, RS *rs, const uint32_t *list, uint64_t *totals) {
    const int lane = lane_id();
    uint64_t seq_run = 0, ops_run = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        const uint32_t r = i < n ? list[i] : 0u;
        uint32_t sb = 0, ob = 0;
        if (i < n && rs[r].n) {
            sb = 2u * ((rs[r].m + 16u + 15u) >> 4);                 /* seq + qual, 16-byte units */
            ob = (rs[r].m + rs[r].n + 16u + 15u) >> 4;
        }
        uint32_t is = wave_incl_scan(sb), io = wave_incl_scan(ob);
        if (i < n) { rs[r].seq_off = (seq_run + is - sb) << 4; rs[r].ops_off = (ops_run + io - ob) << 4; }
        seq_run += wave_bcast_u32(is, 63); ops_run += wave_bcast_u32(io, 63);
    }
    if (lane == 0) { totals[0] = seq_run << 4; totals[1] = ops_run << 4; }
}

/* rs[order[i]].tb_off = off[i] (and .units = units[i] when given): the host lays traceback stores out in
   processing order; the retry phase also replaces the windowed sizes by full-band sizes */
__global__ void __launch_bounds__(64) k_set_tboff(uint32_t n, RS *rs, const uint32_t *order, const uint64_t *off, const uint64_t *units) {
    uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) { rs[order[i]].tb_off = off[i]; if (units) rs[order[i]].units = units[i]; }
}

/* =============================================================================================
 * final stage: the tail of sequence_fragment (simulate.py:348-356) and get_qscores
 * (qscore_model.py:32-75): join the mutated read, align it against the perfect fragment, walk the
 * per-column ops to give every read base its <=k-op cigar window, look the window up with the
 * centre-preserving fallback of get_qscore (:273-287) and sample a score.
 * ========================================================================================== */
__device__ inline int64_t qs_lookup(const brx_qscore_model &qm, uint64_t key) {
    uint64_t h = key * 0x9E3779B97F4A7C15ull;
    uint32_t slot = (uint32_t)(h >> 32) & (qm.hash_size - 1);
    for (;;) {
        uint64_t kk = qm.d_hash_key[slot];
        if (kk == key) return (int64_t)qm.d_hash_row[slot];
        if (kk == ~0ull) return -1;
        slot = (slot + 1) & (qm.hash_size - 1);
    }
}

/* The final stage is three kernels so that each gets its own register budget (occupancy is what hides
 * the latency of the serial column steps):
 *   k_fin_join                    join(new_fragment_bases) (simulate.py:351) + pad; band class of the read
 *   k_fin_align<MAXG, GLO, GHI>   banded Myers + traceback for the reads whose band geometry has
 *                                 GLO <= words-per-lane <= GHI; every instantiation walks the same queue
 *                                 range and skips the other classes' reads, so narrow and wide reads run
 *                                 side by side on two streams
 *   k_fin_qscore                  cigar windows -> qscore rows -> quality bytes, per-read statistics
 * Two phases.  Phase 0 aligns every read with the WINDOWED traceback store (brx_make_geom): the forward pass
 * computes the whole band but writes only the superblocks near the straight line from corner to corner, which
 * is where the path lives -- a third to an eighth of the bytes.  A read whose traceback asks for a cell that
 * was not written gets BRX_KL_RETRY and is aligned again in phase 1 with the full store (the host sizes that
 * pass).  Results never depend on which phase produced them. */
#define BRX_KL_RETRY 0x10000u     /* phase 0 missed: repeat in phase 1                                   */
#define BRX_KL_FULL 0x20000u      /* never windowed: the read holds a junk piece (low-complexity repeats, where
                                     the canonical traceback collects every indel at one end of the repeat) */
#define BRX_KL_LANES 0x40000u     /* narrow band, ACGT only: aligned one read per lane (brx_finlanes.h, k_fin_lanes) */
#define BRX_KL_QUAD 0x80000u      /* band of up to 13 superblocks of one or two words, ACGT only: four reads per wave (brx_quad.h, k_fin_quad) */
#include "brx_finlanes.h"
#include "brx_quad.h"

/* One set of reads (order[q_begin..q_end)).  seq_base / ops_base: where the set's seq and ops buffers start in the
   arena; from here on RS.seq_off / RS.ops_off are offsets from the arena base (`arena`), whichever set a read is in. */
__global__ void __launch_bounds__(64) k_fin_join(BrxDev d, RS *rs, const uint32_t *order, uint32_t q_begin, uint32_t q_end, uint32_t *queue,
                                                  uint64_t seq_base, uint64_t ops_base, const uint8_t *Fbuf, const uint32_t *repl,
                                                  const PPiece *pieces, uint8_t *arena, const uint32_t *F2buf, int fin_lanes, int fin_quad) {
    const int lane = lane_id();
    const brx_error_model &em = d.em;
    for (;;) {
        const uint32_t qi = q_begin + wave_pop(queue);
        if (qi >= q_end) break;
        const uint32_t r = order[qi];
        const RS s = rs[r];
        if (s.n == 0) {
            if (lane == 0) { rs[r].seq_off = seq_base; rs[r].ops_off = ops_base; }
            continue;
        }
        uint8_t *seq = arena + seq_base + s.seq_off;
        wave_join(em, Fbuf + s.F_off, repl + s.F_off, 0, s.n, seq, nullptr);
        /* What the aligners read past the end of a string (values masked or never used: only readability matters): the wave and
           row aligners up to 16 bytes -- the pad written here and behind F by k_build --, the lane aligner up to 31 (brx_finl_planes32
           loads 32 bytes at a time).  For the read that stays inside its own area (seq is followed by its qual half of at least 16
           bytes, k_scan_mut); for the fragment it is the next read's F or the 64 bytes of slack behind Fbuf / the seq buffer
           (brx_hip.hip: A.take(f_bytes + 64), A.take(seq_bytes + 64)). */
        for (uint32_t x = lane; x < 16; x += 64) seq[s.m + x] = 0xFE;
        if (lane == 0) {
            const BrxGeom g = brx_make_geom((int)s.m, (int)s.n, (int)s.ub);
            bool junk = false, too_wide;
            if (!d.raw_mode) for (uint32_t i = 0; i < s.n_pieces; ++i) junk |= (pieces[s.piece_off + i].w0 & 3u) == PC_JUNK;
            RS *o = &rs[r];
            o->seq_off = seq_base + s.seq_off; o->ops_off = ops_base + s.ops_off;
            /* a band of up to four blocks between strings of ACGT only (the read inherits every symbol outside ACGT from its
               fragment: F2's flag word says whether there is one) is aligned one read per lane */
            const bool acgt = F2buf[(s.F_off >> 4) + ((s.n + 15u) >> 4)] == 0u;
            const bool lanes = fin_lanes && !junk && acgt && brx_finl_blocks(s.m, s.n, s.ub) > 0;
            /* ... and a band of up to 13 superblocks of one or two words four reads per wave, a row of 16 lanes each */
            const bool quad = fin_quad && !lanes && acgt && g.G == 1 && (brx_quad_words(s.m, s.n, s.ub) & fin_quad) != 0;
            o->klass = (g.G ? (uint32_t)g.G : 0xFFFFu) | (junk ? BRX_KL_FULL : 0u) | (lanes ? BRX_KL_LANES : 0u) | (quad ? BRX_KL_QUAD : 0u);
            o->units = brx_final_units(s.m, s.n, s.ub, junk ? 0 : d.tb_hmul, &too_wide);     /* traceback store + col_of[] */
            if (too_wide) o->status = s.status | BRX_RS_BAND;
        }
    }
}

/* The narrow-band class of one set, 64 reads per wave: `list` holds its reads by fragment length, longest first, a GROUP is 64
   consecutive entries; `ctr` / `slabs` as for k_fin_align below, counted in groups: slab t holds the move codes of the t-th longest
   group and of every group the wave pops after it. */
__global__ void __launch_bounds__(64, 4) k_fin_lanes(BrxDev d, RS *rs, const uint32_t *list, uint32_t n_list, unsigned long long *ctr,
                                                      const uint64_t *slabs, uint32_t *retries, const uint8_t *Fbuf, uint8_t *seqbuf, uint8_t *opsbuf,
                                                      uint8_t *slab_base, uint64_t *clk) {
    const int lane = lane_id();
    const uint32_t n_groups = (n_list + 63u) >> 6;
    const uint64_t first = uni((uint64_t)atomicAdd(ctr, lane == 0 ? ((1ull << 32) | 1ull) : 0ull));
    uint32_t gi = (uint32_t)first;
    if (gi >= n_groups) return;
    const uint32_t ticket = (uint32_t)(first >> 32);
    const uint64_t slab_at = slabs[ticket], tb_cap = slabs[ticket + 1] - slab_at;
    uint2 *tb = reinterpret_cast<uint2 *>(slab_base) + slab_at;
    for (; gi < n_groups; gi = (uint32_t)uni((uint64_t)atomicAdd(ctr, lane == 0 ? 1ull : 0ull))) {
        const uint32_t idx = gi * 64u + (uint32_t)lane;
        bool valid = idx < n_list;
        const uint32_t r = valid ? list[idx] : 0u;
        RS s;
        if (valid) s = rs[r];
        const uint64_t t_begin = __builtin_amdgcn_s_memtime();
        const uint32_t n = valid ? s.n : 0u, m = valid ? s.m : 0u;
        /* the slab holds the longest read of the group (the host sized it from the same list): a group that does not fit is a bug */
        const uint32_t blocks = wave_max_u32(valid ? (uint32_t)brx_finl_blocks(m, n, s.ub) : 0u);      /* = the Wb of brx_lanes_final */
        const bool fits = brx_finl_units(wave_max_u32(n), blocks) <= tb_cap;
        const uint8_t *F = Fbuf + (valid ? s.F_off : 0ull);
        const uint8_t *seq = seqbuf + (valid ? s.seq_off : 0ull);
        uint8_t *ops_end = opsbuf + (valid ? s.ops_off + (uint64_t)n + (uint64_t)m : 0ull);
        uint32_t ncols = 0, nmatch = 0; bool ok = false;
        brx_lanes_final(valid && fits, seq, (int)m, F, (int)n, valid ? (int)s.ub : 0, tb, ops_end, &ncols, &nmatch, &ok);
        if (valid) {
            RS *o = &rs[r];
            if (!ok) {          /* not fatal here -- the second phase repeats the read with k_fin_align and the full store; only that failing is BRX_RS_BAND */
                o->klass = (s.klass & ~BRX_KL_LANES) | BRX_KL_RETRY;
                atomicAdd(retries, 1u);
                clk[(uint64_t)r * 8 + 2] = 1;
            } else {
                o->status = s.status;
                o->n_cols = ncols; o->n_match = nmatch;
            }
            uint64_t *ck = clk + (uint64_t)r * 8;
            ck[3] = __builtin_amdgcn_s_memtime() - t_begin; ck[7] = (uint64_t)(s.klass & 0xFFFFu) | 0x20000u;     /* bit 17: aligned one read per lane */
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);                      /* the slab is written again by the next group */
    }
}

/* The reads of one set whose band takes at most 13 superblocks of G words, FOUR per wave (brx_quad.h): `list` holds them by
   store size, largest first, a GROUP is four consecutive entries -- one per row of 16 lanes --; `ctr` / `slabs` as for k_fin_align
   below, counted in groups.  A read whose traceback leaves the stored window (or whose group does not fit its slab: a bug) is
   repeated by k_fin_align in the second phase, with the full store. */
template <int G>
__global__ void __launch_bounds__(64, 4) k_fin_quad(BrxDev d, RS *rs, const uint32_t *list, uint32_t n_list, unsigned long long *ctr,
                                                     const uint64_t *slabs, uint32_t *retries, const uint8_t *Fbuf, uint8_t *seqbuf,
                                                     uint8_t *opsbuf, uint8_t *slab_base, uint64_t *clk) {
    const int lane = lane_id();
    const int row = lane >> 4;
    const uint32_t n_groups = (n_list + 3u) >> 2;
    const uint64_t first = uni((uint64_t)atomicAdd(ctr, lane == 0 ? ((1ull << 32) | 1ull) : 0ull));
    uint32_t gi = (uint32_t)first;
    if (gi >= n_groups) return;
    const uint32_t ticket = (uint32_t)(first >> 32);
    const uint64_t slab_at = slabs[ticket], tb_cap = slabs[ticket + 1] - slab_at;
    uint2 *tb = reinterpret_cast<uint2 *>(slab_base) + slab_at;
    for (; gi < n_groups; gi = (uint32_t)uni((uint64_t)atomicAdd(ctr, lane == 0 ? 1ull : 0ull))) {
        const uint32_t idx = gi * 4u + (uint32_t)row;
        const bool valid = idx < n_list;
        const uint32_t r = valid ? list[idx] : 0u;
        RS s;
        if (valid) s = rs[r];
        const uint64_t t_begin = __builtin_amdgcn_s_memtime();
        const uint32_t n = valid ? s.n : 0u, m = valid ? s.m : 0u;
        const uint8_t *F = Fbuf + (valid ? s.F_off : 0ull);
        const uint8_t *seq = seqbuf + (valid ? s.seq_off : 0ull);
        uint8_t *ops_end = opsbuf + (valid ? s.ops_off + (uint64_t)n + (uint64_t)m : 0ull);
        int ncols = 0, nmatch = 0, st = 1;
        brx_quad_align<G>(valid && m > 0 && n > 0, seq, (int)m, F, (int)n, valid ? (int)s.ub : 0,
                          (valid && !(s.klass & BRX_KL_FULL)) ? d.tb_hmul : 0, tb, tb_cap, ops_end, &ncols, &nmatch, &st);
        if (valid && (lane & 15) == 0) {
            RS *o = &rs[r];
            if (st != 0) {                                      /* window miss: k_fin_align repeats the read with the full store */
                o->klass = (s.klass & ~BRX_KL_QUAD) | BRX_KL_RETRY;
                atomicAdd(retries, 1u);
                clk[(uint64_t)r * 8 + 2] = 1;
            } else {
                o->status = s.status;
                o->n_cols = (uint32_t)ncols; o->n_match = (uint32_t)nmatch;
            }
            uint64_t *ck = clk + (uint64_t)r * 8;
            ck[3] = __builtin_amdgcn_s_memtime() - t_begin; ck[7] = (uint64_t)(s.klass & 0xFFFFu) | 0x10000u;     /* bit 16: aligned as one of four */
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);                      /* the slab is written again by the next group */
    }
}

/* One band class of one set: `list` holds the class's reads (longest first), `ctr` is the 64-bit counter of the slab scheme
   (brx_hip.hip, launch_final_phase): low half = list position, high half = waves that have started.  A wave's first pop adds
   to both halves at once -- its ticket t is then at most the position it popped -- and slab t (slabs[t] .. slabs[t + 1], in
   8-byte units from slab_base) holds the traceback store of every read the wave will ever pop. */
#ifndef BRX_FIN4_WAVES
#define BRX_FIN4_WAVES 4         /* waves per SIMD the four-word class is compiled for (4: 128 VGPRs and 8 spilled words; 3: 147 VGPRs) */
#endif
template <int MAXG, int GLO, int GHI>
#ifndef BRX_FIN1_WAVES
#define BRX_FIN1_WAVES 5
#endif
#ifndef BRX_FIN2_WAVES
#define BRX_FIN2_WAVES 4
#endif
__global__ void __launch_bounds__(64, (MAXG == 1 ? BRX_FIN1_WAVES : MAXG == 2 ? BRX_FIN2_WAVES : MAXG == 4 ? BRX_FIN4_WAVES : 1)) k_fin_align(BrxDev d, RS *rs, const uint32_t *list, uint32_t n_list,
                                                   unsigned long long *ctr, const uint64_t *slabs, uint32_t *retries, int phase, const uint8_t *Fbuf,
                                                   uint8_t *seqbuf, uint8_t *opsbuf, uint8_t *slab_base, uint64_t *clk) {
    const int lane = lane_id();
    const uint64_t first = uni((uint64_t)atomicAdd(ctr, lane == 0 ? ((1ull << 32) | 1ull) : 0ull));
    uint32_t qi = (uint32_t)first;
    if (qi >= n_list) return;
    const uint32_t ticket = (uint32_t)(first >> 32);
    const uint64_t slab_at = slabs[ticket], tb_cap = slabs[ticket + 1] - slab_at;
    uint2 *tb = reinterpret_cast<uint2 *>(slab_base) + slab_at;
    for (; qi < n_list; qi = (uint32_t)uni((uint64_t)atomicAdd(ctr, lane == 0 ? 1ull : 0ull))) {
        const uint32_t r = list[qi];
        const RS s = rs[r];
        const int klass = (int)(s.klass & 0xFFFFu);
        const uint64_t t_begin = __builtin_amdgcn_s_memtime();
        uint64_t aclk[2] = {0, 0};
        const uint32_t n = s.n, m = s.m;
        const uint8_t *F = Fbuf + s.F_off;
        uint8_t *seq = seqbuf + s.seq_off;                                 /* joined and padded by k_fin_join */
        uint8_t *ops_end = opsbuf + s.ops_off + (uint64_t)n + (uint64_t)m;
        int ncols = 0, nmatch = 0; bool nospace = false;
        const bool ok = brx_wave_align<MAXG, (GLO < MAXG ? GLO : MAXG)>(seq, (int)m, F, (int)n, (int)s.ub, tb, tb_cap, ops_end, &ncols, &nmatch,
                                             &nospace, nullptr, aclk, (phase == 0 && !(s.klass & BRX_KL_FULL)) ? d.tb_hmul : 0);
        if (lane == 0) {
            RS *o = &rs[r];
            if (!ok && phase == 0) {
                o->klass = s.klass | BRX_KL_RETRY;
                atomicAdd(retries, 1u);
                clk[(uint64_t)r * 8 + 2] = 1;                               /* brx_last_read_cycles: window miss */
            } else {
                o->status = s.status | (ok ? 0u : BRX_RS_BAND);
                o->n_cols = (uint32_t)ncols; o->n_match = (uint32_t)nmatch;
            }
            uint64_t *ck = clk + (uint64_t)r * 8;
            ck[3] = __builtin_amdgcn_s_memtime() - t_begin; ck[4] = aclk[0]; ck[5] = aclk[1];
            ck[7] = (uint64_t)klass;
        }
    }
}

#define BRX_QS_HOT_MAX 128
#define BRX_QS_PEND 128          /* power of two, at least 127: up to 63 windows wait while 64 more arrive */
__global__ void __launch_bounds__(64) k_fin_qscore(BrxDev d, RS *rs, const uint32_t *order, uint32_t q_begin, uint32_t q_end,
                                                    uint32_t *queue, int phase, int klo, int khi, uint8_t *seqbuf, const uint8_t *opsbuf,
                                                    uint8_t *tb_base, uint64_t *clk) {
    __shared__ uint32_t qhist[256];
    __shared__ uint32_t hot_thr[BRX_QS_HOT_MAX], hot_score[BRX_QS_HOT_MAX];
    __shared__ uint8_t hot_idx[256];
    __shared__ uint32_t pend_sp[BRX_QS_PEND], pend_h[BRX_QS_PEND], pend_c[BRX_QS_PEND];          /* windows waiting for the slow path (a ring): base, half width, column */
    __shared__ uint32_t draw_lds[512];                                      /* the draws of two chunks of 256 bases */
    const int lane = lane_id();
    const brx_qscore_model &qm = d.qm;
    /* The full-width window of matches ('=' x k: 89 % of all lookups with nanopore2023) keeps its row in LDS: no hash
       probe and no chain of dependent global loads for the threshold search on the common path. */
    uint32_t hot_n = 0;
    {
        const int64_t row = qs_lookup(qm, (uint64_t)qm.k << 56);
        if (row >= 0) {
            const uint32_t e0 = qm.d_row_off[row], e1 = qm.d_row_off[row + 1];
            if (e1 - e0 <= BRX_QS_HOT_MAX) {
                hot_n = e1 - e0;
                for (uint32_t x = lane; x < hot_n; x += 64) { hot_thr[x] = qm.d_thr[e0 + x]; hot_score[x] = qm.d_score[e0 + x]; }
            }
        }
        __syncthreads();
        /* hot_idx[b] = first entry whose threshold exceeds b << 24: a draw starts its search at the entry of its top byte and
           walks one or two entries, instead of seven dependent probes of a binary search over the whole row */
        for (uint32_t b = lane; b < 256u && hot_n; b += 64) {
            const uint32_t u = b << 24;
            uint32_t e = 0, hi_ = hot_n - 1;
            while (e < hi_) { const uint32_t mid = (e + hi_) >> 1; if (u < hot_thr[mid]) hi_ = mid; else e = mid + 1; }
            hot_idx[b] = (uint8_t)e;
        }
        __syncthreads();
    }
    for (;;) {
        const uint32_t qi = q_begin + wave_pop(queue);
        if (qi >= q_end) break;
        const uint32_t r = order[qi];
        RS s = rs[r];
        if (s.n == 0) continue;
        if (((s.klass & BRX_KL_RETRY) != 0u) != (phase != 0)) continue;     /* window misses are scored in phase 1 */
        if ((int)(s.klass & 0xFFFFu) < klo || (int)(s.klass & 0xFFFFu) > khi) continue;   /* band classes of this launch */
        const uint64_t t_begin = __builtin_amdgcn_s_memtime();
        const uint64_t read = d.first_read + r;
        const uint32_t n = s.n, m = s.m;
        uint8_t *seq = seqbuf + s.seq_off;
        uint8_t *qual = seq + (((uint64_t)m + 16 + 15) & ~15ull);
        const uint8_t *ops_end = opsbuf + s.ops_off + (uint64_t)n + (uint64_t)m;
        const bool ok = !(s.status & BRX_RS_BAND);
        const uint32_t ncols = s.n_cols;
        const uint8_t *ops = ops_end - ncols;

        for (int b = lane; b < 256; b += 64) qhist[b] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();

        const uint32_t margin = (uint32_t)(qm.k - 1) / 2;
        const uint32_t maxrun = (1u << qm.gap_bits) - 1u;
        bool qmiss = false;
        /* Round 6: ONE pass over the alignment's COLUMNS, 64 per step, with the next steps' op bytes in flight.  (Rounds 1-5 walked
           the read's BASES: a first pass wrote the column of every base to col_of[] -- 4 bytes per base -- and every step of the
           second loaded two of them and THEN the twelve op bytes behind the first: two dependent round trips per 64 bases, 240 steps
           per 15 kb read, in a wave that has nothing else to do -- the kernel sat in s_waitcnt 75 % of its time,
           profiles/r05_pmc_per_kernel.csv.)  A lane owns a column; the base it holds, if any, has index = bases before the step +
           rank among the step's non-'D' columns.  Two speeds, as before: the full-width all-match window (89 % of the windows;
           qscore_model.py:273 finds its row at the first try) is recognised from the '=' masks of three neighbouring steps --
           the 2 margin + 1 columns around the lane are all '=' -- by wave-uniform shifts, and takes the row in LDS; every other
           window waits in an LDS ring with its centre column and 64 of them walk their ops, probe the hash and search their
           rows together. */
        uint32_t q_head = 0, q_tail = 0;                                   /* ring of BRX_QS_PEND entries, wave-uniform */
        auto slow_lane = [&](uint32_t sp, uint32_t h, uint32_t c) {
            /* ops and D-runs of the widest window (2 h + 1 read bases around sp, whose column is c): 2 bits per op, op i at bits
               2 i; 4 bits per gap (saturated), the gap after op i at bits 4 i */
            uint64_t opsbits = 0, gapbits = 0;
            {
                uint32_t c0 = c;                                           /* column of base sp - h: h bases to the left */
                for (uint32_t left = 0; left < h; ) { c0 -= 1u; if (ops[c0] != BRX_OP_D) left += 1u; }
                uint32_t idx = 0, rn = 0;
                for (uint32_t cc = c0; idx < 2u * h + 1u; ++cc) {
                    const uint32_t op = ops[cc];
                    if (op == BRX_OP_D) { rn += 1; continue; }
                    if (idx > 0) { const uint32_t code = rn >= maxrun ? maxrun : rn; gapbits |= (uint64_t)code << (4 * (idx - 1)); }
                    opsbits |= (uint64_t)op << (2 * idx);
                    rn = 0; idx += 1;
                }
            }
            uint32_t score = 0; bool found = false;
            uint32_t hh = h;
            for (;;) {
                /* sub-window of 2hh+1 ops centred on op index h of the widest window */
                uint32_t first = h - hh, cnt = 2 * hh + 1;
                uint64_t key = (uint64_t)cnt << 56;
                int shift = 0;
                for (uint32_t x = 0; x < cnt; ++x) {
                    if (x > 0) { key |= ((gapbits >> (4 * (first + x - 1))) & 15ull) << shift; shift += qm.gap_bits; }
                    key |= ((opsbits >> (2 * (first + x))) & 3ull) << shift; shift += 2;
                }
                int64_t row = qs_lookup(qm, key);
                if (row >= 0) {
                    uint32_t e0 = qm.d_row_off[row], e1 = qm.d_row_off[row + 1];
                    uint32_t w4[4];
                    brx_draw4(d.seed, read, BRX_ST_QS, (uint64_t)(sp >> 2), w4);
                    uint32_t u = w4[sp & 3];
                    /* first entry whose cumulative threshold exceeds the draw, else the last one: binary
                       search over the row (rows hold up to ~90 scores; thresholds are non-decreasing) */
                    uint32_t e = e0, hi_ = e1 - 1;
                    while (e < hi_) { const uint32_t mid = (e + hi_) >> 1; if (u < qm.d_thr[mid]) hi_ = mid; else e = mid + 1; }
                    score = qm.d_score[e]; found = true;
                    break;
                }
                if (hh == 0) break;
                hh -= 1;
            }
            if (!found) qmiss = true;
            qual[sp] = (uint8_t)(score + 33);
            atomicAdd(&qhist[score & 255u], 1u);
        };
        auto drain = [&](uint32_t count) {                                 /* the `count` oldest queued windows, one per lane */
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if ((uint32_t)lane < count) {
                const uint32_t e = (q_head + (uint32_t)lane) & (BRX_QS_PEND - 1u);
                slow_lane(pend_sp[e], pend_h[e], pend_c[e]);
            }
            q_head += count;
            __syncthreads();                                               /* the slots may be written again */
        };
        /* The draw of base sp is word sp & 3 of block sp >> 2 (qscore_model.py:54-71 draws once per base).  The wave computes the
           blocks of 256 bases at a time (one Philox block per lane) into one of two LDS slots; a step's bases lie in at most two
           consecutive chunks of 256. */
        uint32_t chunks_drawn = 0;                                         /* chunks [0, chunks_drawn) are or were in LDS: chunk q in slot q & 1 */
        if (ok) {
            const uint32_t nsteps = (ncols + 63u) >> 6;
            auto load_step = [&](uint32_t step) -> uint32_t {              /* the op byte of this lane's column in `step` (0xFF: no column) */
                const uint32_t c = 64u * step + (uint32_t)lane;
                return (step < nsteps && c < ncols) ? (uint32_t)ops[c] : 0xFFu;
            };
            uint32_t op_cur = load_step(0), op_next = load_step(1);
            unsigned long long e_prev = 0ull, e_cur = __ballot(op_cur == 0u), e_next = 0ull;
            uint32_t sp_base = 0;                                          /* read bases in the columns before this step */
            for (uint32_t step = 0; step < nsteps; ++step) {
                const uint32_t op_nn = load_step(step + 2u);               /* in flight over two steps */
                e_next = __ballot(op_next == 0u);
                const unsigned long long base_m = __ballot(op_cur != BRX_OP_D && op_cur != 0xFFu);
                /* columns whose 2 margin + 1 neighbourhood is all '=' (wave-uniform) */
                unsigned long long hot_m = e_cur;
                for (uint32_t dd = 1; dd <= margin; ++dd)
                    hot_m &= ((e_cur >> dd) | (e_next << (64u - dd))) & ((e_cur << dd) | (e_prev >> (64u - dd)));
                /* the draws of this step's bases: chunks (sp_base >> 8) and possibly the next */
                while (chunks_drawn <= ((sp_base + 63u) >> 8)) {
                    uint32_t w4[4];
                    brx_draw4(d.seed, read, BRX_ST_QS, (uint64_t)(chunks_drawn * 64u) + (uint64_t)lane, w4);
                    uint32_t *slot = draw_lds + (chunks_drawn & 1u) * 256u + 4u * (uint32_t)lane;
                    slot[0] = w4[0]; slot[1] = w4[1]; slot[2] = w4[2]; slot[3] = w4[3];
                    chunks_drawn += 1u;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_s_waitcnt(0);
                    __syncthreads();
                }
                const bool valid = ((base_m >> lane) & 1ull) != 0ull;
                const uint32_t sp = sp_base + (uint32_t)__popcll(base_m & ((1ull << lane) - 1ull));
                uint32_t h = margin;
                bool hot = false;
                if (valid) {
                    if (sp < h) h = sp;
                    if (m - 1 - sp < h) h = m - 1 - sp;
                    hot = hot_n && h == margin && ((hot_m >> lane) & 1ull) != 0ull;
                }
                if (hot) {
                    const uint32_t u = draw_lds[((sp >> 8) & 1u) * 256u + (sp & 255u)];
                    uint32_t e = hot_idx[u >> 24];             /* first entry whose cumulative threshold exceeds the draw, else the last one */
                    while (e < hot_n - 1u && u >= hot_thr[e]) e += 1u;
                    const uint32_t score = hot_score[e];
                    qual[sp] = (uint8_t)(score + 33);
                    atomicAdd(&qhist[score & 255u], 1u);
                }
                const bool slow = valid && !hot;
                const unsigned long long sm = __ballot(slow);
                if (sm) {
                    if (slow) {
                        const uint32_t e = (q_tail + (uint32_t)__popcll(sm & ((1ull << lane) - 1ull))) & (BRX_QS_PEND - 1u);
                        pend_sp[e] = sp; pend_h[e] = h; pend_c[e] = 64u * step + (uint32_t)lane;
                    }
                    q_tail += (uint32_t)__popcll(sm);
                    if (q_tail - q_head >= 64u) drain(64u);
                }
                sp_base += (uint32_t)__popcll(base_m);
                e_prev = e_cur; e_cur = e_next;
                op_cur = op_next; op_next = op_nn;
            }
        }
        if (q_tail != q_head) drain(q_tail - q_head);
        if (__ballot(qmiss)) s.status |= BRX_RS_QMISS;
        __syncthreads();
        if (lane == 0) {
            double qerr = 0.0;
            for (int q = 0; q < 256; ++q) {
                uint32_t c = qhist[q];
                if (c) qerr += (double)c * brx_exp(-(double)q / 10.0 * 2.302585092994046);
            }
            uint32_t lo = s.start_trim, hi = m >= s.end_trim ? m - s.end_trim : 0;
            if (s.end_trim == 0) hi = 0;
            if (hi < lo) hi = lo;
            RS *o = &rs[r];
            o->status = s.status | ((hi - lo) == 0 ? BRX_RS_EMPTY : 0u);
            o->qerr = qerr;
            o->seq_len = hi - lo;
            clk[(uint64_t)r * 8 + 6] = __builtin_amdgcn_s_memtime() - t_begin;
        }
        __syncthreads();
    }
}

/* =============================================================================================
 * FASTQ record (simulate.py:73-82).  One templated writer is used both to size and to write.
 * ========================================================================================== */
struct CountSink { uint32_t n; __device__ void put(uint8_t) { n++; } };
struct ByteSink { uint8_t *p; uint32_t n; __device__ void put(uint8_t c) { p[n++] = c; } };

template <class S> __device__ void put_str(S &s, const char *t) { while (*t) s.put((uint8_t)*t++); }
template <class S> __device__ void put_dec(S &s, uint64_t v) {
    char buf[24]; int n = 0;
    do { buf[n++] = (char)('0' + (v % 10)); v /= 10; } while (v);
    while (n) s.put((uint8_t)buf[--n]);
}
template <class S> __device__ void put_hex(S &s, uint32_t v, int digits) {
    for (int i = digits - 1; i >= 0; --i) { uint32_t x = (v >> (4 * i)) & 15u; s.put((uint8_t)(x < 10 ? '0' + x : 'a' + x - 10)); }
}
/* round(v*1000) to nearest, ties to even, on the exact binary value of v (what '%.3f' prints) */
__device__ inline uint64_t milli_round(double v) {
    uint64_t u = brx_d2u(v);
    int e = (int)((u >> 52) & 0x7FF);
    uint64_t mant = u & 0xFFFFFFFFFFFFFull;
    if (e == 0) return 0;                      /* zero / subnormal */
    mant |= 1ull << 52;
    int sh = 1075 - e;                         /* v = mant * 2^-sh */
    uint64_t scaled = mant * 1000ull;
    if (sh <= 0) return scaled << (-sh);
    if (sh >= 64) return 0;
    uint64_t q = scaled >> sh, rem = scaled & ((1ull << sh) - 1ull), half = 1ull << (sh - 1);
    if (rem > half || (rem == half && (q & 1ull))) q += 1;
    return q;
}
template <class S>
__device__ void put_header(S &s, const BrxDev &d, uint64_t read, const RS &r, const PPiece *pieces) {
    uint32_t w[4];
    brx_draw4(d.seed, read, BRX_ST_NAME, 0, w);            /* uuid.UUID(int=getrandbits(128)) */
    s.put('@');
    put_hex(s, w[0], 8); s.put('-'); put_hex(s, w[1] >> 16, 4); s.put('-'); put_hex(s, w[1] & 0xFFFFu, 4); s.put('-');
    put_hex(s, w[2] >> 16, 4); s.put('-'); put_hex(s, w[2] & 0xFFFFu, 4); put_hex(s, w[3], 8); s.put(' ');
    for (uint32_t i = 0; i < r.n_pieces; ++i) {
        const PPiece pc = pieces[r.piece_off + i];
        uint32_t type = pc.w0 & 3u, strand = (pc.w0 >> 2) & 1u, contig = pc.w0 >> 3;
        if (i > 0) put_str(s, "chimera ");
        if (type == PC_JUNK) put_str(s, "junk_seq ");
        else if (type == PC_RANDOM) put_str(s, "random_seq ");
        else {
            const brx_contig ct = d.ref.d_contigs[contig];
            for (uint32_t x = 0; x < ct.name_len; ++x) s.put(d.ref.d_names[ct.name_off + x]);
            s.put(','); s.put(strand ? '-' : '+'); put_str(s, "strand,");
            put_dec(s, pc.start); s.put('-'); put_dec(s, pc.end);
            if (type == PC_HAIRPIN) { put_str(s, " (hairpin) 0-"); put_dec(s, pc.left_over); }
            s.put(' ');
        }
    }
    put_str(s, "length="); put_dec(s, r.seq_len);
    put_str(s, " error-free_length="); put_dec(s, r.frag_len);
    put_str(s, " read_identity=");
    double ident = r.n_cols ? (double)r.n_match / (double)r.n_cols : 0.0;
    uint64_t mr = milli_round(ident * 100.0);
    put_dec(s, mr / 1000); s.put('.');
    uint32_t fr = (uint32_t)(mr % 1000);
    s.put((uint8_t)('0' + fr / 100)); s.put((uint8_t)('0' + (fr / 10) % 10)); s.put((uint8_t)('0' + fr % 10));
    s.put('%'); s.put('\n');
}

__global__ void __launch_bounds__(64) k_recsize(BrxDev d, RS *rs, const PPiece *pieces) {
    uint32_t r = blockIdx.x * 64 + threadIdx.x;
    if (r >= d.n_reads) return;
    RS s = rs[r];
    uint32_t len = 0, hl = 0;
    if (s.n && s.seq_len && !(s.status & BRX_RS_NOFRAG)) {
        if (d.raw_mode) len = 2 * s.seq_len;
        else {
            CountSink c; c.n = 0;
            put_header(c, d, d.first_read + r, s, pieces);
            hl = c.n; len = hl + 2 * s.seq_len + 4;
        }
    }
    rs[r].rec_len = len; rs[r].hdr_len = hl;
}

/* record offsets in read order.  totals[5] = output bytes */
__global__ void __launch_bounds__(64) k_scan_rec(uint32_t n_reads, RS *rs, uint64_t *totals) {
    const int lane = lane_id();
    uint64_t run = 0;
    for (uint32_t base = 0; base < n_reads; base += 64) {
        uint32_t r = base + lane;
        uint32_t len = r < n_reads ? rs[r].rec_len : 0u;
        /* a 64-read group stays far below 2^32 bytes */
        uint32_t inc = wave_incl_scan(len);
        if (r < n_reads) rs[r].rec_off = run + inc - len;
        run += wave_bcast_u32(inc, 63);
    }
    if (lane == 0) totals[5] = run;
}

__global__ void __launch_bounds__(64) k_emit(BrxDev d, const RS *rs, const PPiece *pieces, const uint8_t *seqbuf, uint8_t *out) {
    const uint32_t r = blockIdx.x;
    const int lane = lane_id();
    const RS s = rs[r];
    if (s.rec_len == 0) return;
    uint8_t *o = out + s.rec_off;
    const uint8_t *seq = seqbuf + s.seq_off + s.start_trim;
    const uint8_t *qual = seqbuf + s.seq_off + (((uint64_t)s.m + 16 + 15) & ~15ull) + s.start_trim;
    const uint32_t L = s.seq_len;
    if (d.raw_mode) {
        for (uint32_t x = lane; x < L; x += 64) { o[x] = seq[x]; o[L + x] = qual[x]; }
        return;
    }
    if (lane == 0) { ByteSink b; b.p = o; b.n = 0; put_header(b, d, d.first_read + r, s, pieces); }
    uint8_t *p = o + s.hdr_len;
    for (uint32_t x = lane; x < L; x += 64) { p[x] = d.ref.sym[seq[x] & 15u]; p[L + 3 + x] = qual[x]; }
    if (lane == 0) { p[L] = '\n'; p[L + 1] = '+'; p[L + 2] = '\n'; p[2 * L + 3] = '\n'; }
}

__global__ void __launch_bounds__(64) k_stats(BrxDev d, const RS *rs, brx_read_stats *out) {
    uint32_t r = blockIdx.x * 64 + threadIdx.x;
    if (r >= d.n_reads) return;
    const RS s = rs[r];
    brx_read_stats o;
    o.status = s.status; o.frag_len = s.frag_len; o.seq_len = s.seq_len; o.n_cols = s.n_cols; o.n_match = s.n_match;
    o.padded_len = s.m; o.loop_count = s.loops; o.change_count = s.changes; o.n_alignments = s.naligns;
    o.rec_len = s.rec_len; o.rec_off = s.rec_off; o.target_identity = s.target; o.qerr_sum = s.qerr;
    out[r] = o;
}

/* =============================================================================================
 * brx_align_batch: one pair per wave, band doubling when no bound is given.
 * Per-pair scratch at scr + off[i]: [query copy][target copy][traceback store].
 * Bytes are permuted so that A,C,G,T,N become 0..4 (equality is preserved).
 * ========================================================================================== */
__device__ __forceinline__ uint8_t perm_byte(uint8_t b) {
    switch (b) {
    case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case 'N': return 4;
    case 0: return 'A'; case 1: return 'C'; case 2: return 'G'; case 3: return 'T'; case 4: return 'N';
    default: return b;
    }
}

__global__ void __launch_bounds__(64) k_align_batch(uint32_t n_pairs, uint32_t p_begin, uint32_t p_end, uint32_t *queue,
                                                     const uint8_t *qs, const uint64_t *q_off, const uint8_t *ts, const uint64_t *t_off,
                                                     const int32_t *k_hint, int32_t *dist, uint32_t *ncols_out, uint32_t *nmatch_out,
                                                     uint8_t *ops_out, const uint64_t *ops_off,
                                                     uint8_t *scr, const uint64_t *scr_off, const uint64_t *scr_bytes,
                                                     uint32_t *prog) {
    const int lane = lane_id();
    for (;;) {
        const uint32_t i = p_begin + wave_pop(queue);
        BRX_PROG(prog, 0, i + 1);
        if (i >= p_end) break;
        const uint32_t Q = (uint32_t)(q_off[i + 1] - q_off[i]), T = (uint32_t)(t_off[i + 1] - t_off[i]);
        const uint8_t *q = qs + q_off[i], *t = ts + t_off[i];
        uint8_t *base = scr + scr_off[i];
        const uint64_t bytes = scr_bytes[i];
        uint64_t qpad = ((uint64_t)Q + 16 + 15) & ~15ull, tpad = ((uint64_t)T + 16 + 15) & ~15ull;
        uint8_t *qb = base, *tbuf = base + qpad;
        for (uint32_t x = lane; x < Q + 16; x += 64) qb[x] = x < Q ? perm_byte(q[x]) : 0xFF;
        for (uint32_t x = lane; x < T + 16; x += 64) tbuf[x] = x < T ? perm_byte(t[x]) : 0xFE;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        uint2 *tb = reinterpret_cast<uint2 *>(base + qpad + tpad);
        uint64_t cap = bytes > qpad + tpad ? (bytes - qpad - tpad) / 8 : 0;
        uint8_t *ops_end = ops_out ? ops_out + ops_off[i] + Q + T : nullptr;
        int kh = k_hint[i];
        int maxk = (int)(Q > T ? Q : T);
        int kk = kh >= 0 ? kh : (maxk < 64 ? maxk : 64);
        int nc = 0, nm = 0; bool ok = false, nospace = false;
        BRX_PROG(prog, 1, 1);
        for (int round = 0; round < 40; ++round) {
            BRX_PROG(prog, 2, round + 1);
            ok = uni(brx_wave_align<16>(qb, (int)Q, tbuf, (int)T, kk, tb, cap, ops_end, &nc, &nm, &nospace, prog));
            nc = uni(nc); nm = uni(nm); nospace = uni(nospace);
            if (ok || nospace || kh >= 0 || kk >= maxk) break;
            kk = kk * 2 > maxk ? maxk : kk * 2;
        }
        BRX_PROG(prog, 1, 2);
        if (ok && ops_end) {
            /* move the ops (written backwards from the end of the area) to its start */
            uint8_t *dst = ops_out + ops_off[i];
            const uint8_t *src = ops_end - nc;
            if (dst != src) for (uint32_t base2 = 0; base2 < (uint32_t)nc; base2 += 64) {
                uint32_t x = base2 + lane;
                uint8_t v = x < (uint32_t)nc ? src[x] : 0;
                __builtin_amdgcn_s_waitcnt(0);
                if (x < (uint32_t)nc) dst[x] = v;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_waitcnt(0);
            }
        }
        BRX_PROG(prog, 1, 3);
        if (lane == 0) {
            dist[i] = ok ? (nc - nm) : (nospace ? -2 : -1);
            ncols_out[i] = ok ? (uint32_t)nc : 0u;
            nmatch_out[i] = ok ? (uint32_t)nm : 0u;
        }
    }
}

#endif /* BRX_KERNELS_H */

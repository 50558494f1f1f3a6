/*
 * brx_hip.hip -- host side of libbrx_hip.so: the C-ABI declared in include/brx.h.
 *
 * Build (see badread_amd/build.py):
 *   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared brx_hip.hip -o libbrx_hip.so
 * -ffp-contract=off is part of the numerical spec (include/brx_spec.h): the device must round
 * every double operation exactly like the CPU oracle.
 *
 * All device memory comes from the caller: descriptors point at caller tensors, and every
 * temporary lives in the caller's scratch arena, carved by a bump allocator per call.
 */
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <vector>

#include "brx_kernels.h"

#define BRX_MAX_CHUNKS 60

struct brx_ctx {
    int device;
    int n_cu;
    int waves_per_cu;
    BrxDev dev;
    bool has_ref, has_em, has_qm, has_params;
    uint8_t *scratch;
    size_t scratch_bytes;
    size_t scratch_needed, output_needed;
    uint64_t win_bytes;
    uint64_t *h_totals;          /* pinned, 16 x u64 followed by 64 x u32 debug progress words */
    uint32_t *h_prog, *d_prog;   /* host / device views of the progress words */
    hipEvent_t ev_b[BRX_STAGE_COUNT], ev_e[BRX_STAGE_COUNT];   /* begin / end of each stage on the launch stream */
    float stage_ms[BRX_STAGE_COUNT];
    uint32_t final_launches, mutate_passes;
    int mutate_inline;
    int fin_balance;             /* BRX_FIN_BALANCE: 1 = the two-word band class runs on the main stream behind the one-word class */
    uint32_t seg_waves_per_cu;   /* BRX_SEG_WAVES_PER_CU: persistent waves of k_mutate_seg per CU */
    uint32_t tail_reads;         /* BRX_TAIL_READS: this few reads left in the mutate stage -> one in-place launch */
    int tb_hmul;                 /* BRX_TB_WINDOW: window of the final traceback store in sqrt(ub) units (2; 0 = full store; -1 = 8 rows, test) */
    uint32_t window_misses;      /* reads of the last batch whose final traceback left the stored window (phase 1) */
    uint32_t lane_threshold;
    hipStream_t side;            /* second stream: the wide-band align kernels run beside the narrow one (one stream for all
                                    three wide classes: a stream per class measured 30 % slower, r01d) */
    hipEvent_t ev_a1b[BRX_MAX_CHUNKS], ev_a1e[BRX_MAX_CHUNKS];   /* k_fin_align<1,1,1> of every scratch chunk */
    hipEvent_t ev_qsb[BRX_MAX_CHUNKS], ev_qse[BRX_MAX_CHUNKS];   /* k_fin_qscore of every scratch chunk      */
    hipEvent_t ev_fork, ev_join;
    uint64_t *d_clk, *d_phase; uint32_t clk_reads;
    int profile;                 /* BRX_PROFILE=1: the mutate kernels time their phases (brx_last_phase_cycles) */
    char err[512];
};

static int fail(brx_ctx *c, int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof(c->err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return fail((c), BRX_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

static bool brx_debug() { static int v = -1; if (v < 0) { const char *e = getenv("BRX_DEBUG"); v = (e && *e && *e != '0') ? 1 : 0; } return v == 1; }
#define DBG(...) do { if (brx_debug()) { fprintf(stderr, "[brx] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)

struct Arena {
    uint8_t *base; size_t cap; size_t used;
    void *take(size_t bytes) {
        size_t at = (used + 255) & ~(size_t)255;
        used = at + bytes;
        return used <= cap ? base + at : nullptr;
    }
    bool ok() const { return used <= cap; }
};

extern "C" const char *brx_version(void) { return "brx-hip 0.1 (gfx950)"; }

static char g_create_err[512] = "";

/* every stream, event and pinned buffer the context owns (the context is calloc'ed: unset handles are null) */
static void release(brx_ctx *c) {
    if (!c) return;
    for (int i = 0; i < BRX_STAGE_COUNT; ++i) {
        if (c->ev_b[i]) (void)hipEventDestroy(c->ev_b[i]);
        if (c->ev_e[i]) (void)hipEventDestroy(c->ev_e[i]);
    }
    for (int i = 0; i < BRX_MAX_CHUNKS; ++i) {
        if (c->ev_a1b[i]) (void)hipEventDestroy(c->ev_a1b[i]);
        if (c->ev_a1e[i]) (void)hipEventDestroy(c->ev_a1e[i]);
        if (c->ev_qsb[i]) (void)hipEventDestroy(c->ev_qsb[i]);
        if (c->ev_qse[i]) (void)hipEventDestroy(c->ev_qse[i]);
    }
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->h_totals) (void)hipHostFree(c->h_totals);
    free(c);
}

static int create_fail(brx_ctx *c, const char *what, hipError_t e) {
    snprintf(g_create_err, sizeof(g_create_err), "brx_create: %s failed: %s (%d)", what, hipGetErrorString(e), (int)e);
    release(c);
    return BRX_E_HIP;
}

extern "C" int brx_create(int device_id, brx_ctx **out) {
    if (!out) return BRX_E_ARG;
    g_create_err[0] = 0;
    brx_ctx *c = (brx_ctx *)calloc(1, sizeof(brx_ctx));
    if (!c) return BRX_E_ARG;
    c->device = device_id;
    hipError_t e;
    if ((e = hipSetDevice(device_id)) != hipSuccess) return create_fail(c, "hipSetDevice", e);
    int n_cu = 0;
    if ((e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device_id)) != hipSuccess)
        return create_fail(c, "hipDeviceGetAttribute(MultiprocessorCount)", e);
    c->n_cu = n_cu > 0 ? n_cu : 256;
    const char *w = getenv("BRX_WAVES_PER_CU");
    c->waves_per_cu = w ? atoi(w) : 16;
    if (c->waves_per_cu < 1) c->waves_per_cu = 1;
    const char *wb = getenv("BRX_WIN_KB");
    c->win_bytes = (uint64_t)(wb ? atoi(wb) : 256) << 10;
    if ((e = hipHostMalloc((void **)&c->h_totals, 16 * sizeof(uint64_t) + 64 * sizeof(uint32_t), hipHostMallocMapped)) != hipSuccess)
        return create_fail(c, "hipHostMalloc", e);
    c->h_prog = (uint32_t *)(c->h_totals + 16);
    memset(c->h_prog, 0, 64 * sizeof(uint32_t));
    void *dp = nullptr;
    if ((e = hipHostGetDevicePointer(&dp, c->h_prog, 0)) != hipSuccess) return create_fail(c, "hipHostGetDevicePointer", e);
    c->d_prog = (uint32_t *)dp;
    for (int i = 0; i < BRX_STAGE_COUNT; ++i) {
        if ((e = hipEventCreate(&c->ev_b[i])) != hipSuccess) return create_fail(c, "hipEventCreate", e);
        if ((e = hipEventCreate(&c->ev_e[i])) != hipSuccess) return create_fail(c, "hipEventCreate", e);
    }
    if ((e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking)) != hipSuccess) return create_fail(c, "hipStreamCreate", e);
    for (int i = 0; i < BRX_MAX_CHUNKS; ++i) {
        if ((e = hipEventCreate(&c->ev_a1b[i])) != hipSuccess || (e = hipEventCreate(&c->ev_a1e[i])) != hipSuccess ||
            (e = hipEventCreate(&c->ev_qsb[i])) != hipSuccess || (e = hipEventCreate(&c->ev_qse[i])) != hipSuccess)
            return create_fail(c, "hipEventCreate", e);
    }
    if ((e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)) != hipSuccess) return create_fail(c, "hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming)) != hipSuccess) return create_fail(c, "hipEventCreate", e);
    { const char *mi = getenv("BRX_MUTATE_INLINE"); c->mutate_inline = (mi && atoi(mi)) ? 1 : 0; }
    { const char *pf = getenv("BRX_PROFILE"); c->profile = (pf && atoi(pf)) ? 1 : 0; }
    { const char *fb = getenv("BRX_FIN_BALANCE"); c->fin_balance = fb ? atoi(fb) : 1; }
    { const char *tw = getenv("BRX_TB_WINDOW"); c->tb_hmul = tw ? atoi(tw) : 2; }
    { const char *tr = getenv("BRX_TAIL_READS"); c->tail_reads = tr ? (uint32_t)atoi(tr) : 2048u; }
    { const char *sw = getenv("BRX_SEG_WAVES_PER_CU"); c->seg_waves_per_cu = sw && atoi(sw) > 0 ? (uint32_t)atoi(sw) : 8u; }
    { const char *lt = getenv("BRX_LANE_THRESHOLD"); c->lane_threshold = lt ? (uint32_t)atoi(lt) : 3000u; }
    c->err[0] = 0;
    *out = c;
    return BRX_OK;
}

extern "C" void brx_destroy(brx_ctx *c) { release(c); }

extern "C" const char *brx_last_error(const brx_ctx *c) { return c ? c->err : g_create_err; }

extern "C" int brx_set_reference(brx_ctx *c, const brx_reference *r) {
    if (!c || !r) return BRX_E_ARG;
    if (r->n_contigs == 0 || !r->d_packed || !r->d_contigs) return fail(c, BRX_E_ARG, "reference has no contigs");
    c->dev.ref = *r; c->has_ref = true; return BRX_OK;
}
extern "C" int brx_set_error_model(brx_ctx *c, const brx_error_model *m) {
    if (!c || !m) return BRX_E_ARG;
    if (m->k < 1 || m->k > 16) return fail(c, BRX_E_ARG, "error model k-mer size %d out of range", m->k);
    c->dev.em = *m; c->has_em = true; return BRX_OK;
}
extern "C" int brx_set_qscore_model(brx_ctx *c, const brx_qscore_model *m) {
    if (!c || !m) return BRX_E_ARG;
    if (m->k < 1 || (m->k & 1) == 0 || 2 * m->k + m->gap_bits * (m->k - 1) > 56 || (m->hash_size & (m->hash_size - 1)))
        return fail(c, BRX_E_ARG, "unsupported qscore model geometry");
    c->dev.qm = *m; c->has_qm = true; return BRX_OK;
}
extern "C" int brx_set_params(brx_ctx *c, const brx_sim_params *p) {
    if (!c || !p) return BRX_E_ARG;
    c->dev.p = *p; c->has_params = true; return BRX_OK;
}
extern "C" int brx_set_scratch(brx_ctx *c, void *d_scratch, size_t bytes) {
    if (!c) return BRX_E_ARG;
    c->scratch = (uint8_t *)d_scratch; c->scratch_bytes = bytes; return BRX_OK;
}
extern "C" size_t brx_scratch_needed(const brx_ctx *c) { return c ? c->scratch_needed : 0; }
extern "C" size_t brx_output_needed(const brx_ctx *c) { return c ? c->output_needed : 0; }
extern "C" int brx_last_stage_ms(const brx_ctx *c, float ms[BRX_STAGE_COUNT]) {
    if (!c || !ms) return BRX_E_ARG;
    for (int i = 0; i < BRX_STAGE_COUNT; ++i) ms[i] = c->stage_ms[i];
    return BRX_OK;
}

/* wait for the stream; with BRX_DEBUG set, poll instead and on a stall print the kernel's progress
 * words and leave the process (a hung kernel must not take the GPU box down with it) */
static int wait_stream(brx_ctx *c, hipStream_t st, const char *what) {
    if (!brx_debug()) { HIPCHK(c, hipStreamSynchronize(st)); return BRX_OK; }
    const char *w = getenv("BRX_WATCHDOG_S");
    int limit_ms = (w ? atoi(w) : 15) * 1000;
    for (int ms = 0;; ms += 20) {
        hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return BRX_OK;
        if (e != hipErrorNotReady) return fail(c, BRX_E_HIP, "%s: %s", what, hipGetErrorString(e));
        if (ms >= limit_ms) {
            fprintf(stderr, "[brx] WATCHDOG: %s stalled; progress words:", what);
            for (int i = 0; i < 64; ++i) fprintf(stderr, "%s%u", (i % 8) ? " " : " | ", c->h_prog[i]);
            fprintf(stderr, "\n"); fflush(stderr);
            _exit(99);
        }
        usleep(20000);
    }
}

extern "C" int brx_last_read_cycles(brx_ctx *c, uint64_t *h_out, uint32_t n_reads) {
    if (!c || !h_out) return BRX_E_ARG;
    if (!c->d_clk || n_reads > c->clk_reads) return fail(c, BRX_E_STATE, "no per-read cycle counters for %u reads", n_reads);
    HIPCHK(c, hipMemcpy(h_out, c->d_clk, (size_t)n_reads * 64, hipMemcpyDeviceToHost));
    return BRX_OK;
}

extern "C" int brx_last_phase_cycles(brx_ctx *c, uint64_t *h_out, uint32_t n_reads) {
    if (!c || !h_out) return BRX_E_ARG;
    if (!c->d_phase || n_reads > c->clk_reads) return fail(c, BRX_E_STATE, "no phase counters for %u reads", n_reads);
    HIPCHK(c, hipMemcpy(h_out, c->d_phase, (size_t)n_reads * 64, hipMemcpyDeviceToHost));
    return BRX_OK;
}

extern "C" uint32_t brx_last_mutate_passes(const brx_ctx *c) { return c ? c->mutate_passes : 0; }
extern "C" uint32_t brx_last_final_launches(const brx_ctx *c) { return c ? c->final_launches : 0; }
extern "C" uint32_t brx_last_window_misses(const brx_ctx *c) { return c ? c->window_misses : 0; }

static int read_totals(brx_ctx *c, hipStream_t st, const uint64_t *d_totals, int n) {
    HIPCHK(c, hipMemcpyAsync(c->h_totals, d_totals, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    return wait_stream(c, st, "pipeline stage");
}

static int scratch_short(brx_ctx *c, size_t needed) {
    c->scratch_needed = needed;
    return fail(c, BRX_E_SCRATCH, "scratch arena too small: need about %zu bytes, have %zu", needed, c->scratch_bytes);
}

/* ---------------------------------------------------------------------------------------------
 * the shared pipeline: plan (or raw fragments) -> build -> mutate -> final -> records
 * ------------------------------------------------------------------------------------------- */
static int run_pipeline(brx_ctx *c, uint64_t seed, uint64_t first_read, uint32_t n_reads, bool raw,
                        const uint8_t *d_frags, const uint64_t *d_frag_off, const double *d_target,
                        uint8_t *d_out, size_t out_cap, brx_read_stats *d_stats, size_t *out_bytes, hipStream_t st) {
    if (!c->has_em || !c->has_qm) return fail(c, BRX_E_STATE, "error/qscore model not set");
    if (!raw && (!c->has_ref || !c->has_params)) return fail(c, BRX_E_STATE, "reference or parameters not set");
    if (!c->scratch) return fail(c, BRX_E_STATE, "scratch arena not set");
    if (out_bytes) *out_bytes = 0;
    if (n_reads == 0) return BRX_OK;
    HIPCHK(c, hipSetDevice(c->device));
    BrxDev dev = c->dev;
    dev.seed = seed; dev.first_read = first_read; dev.n_reads = n_reads; dev.raw_mode = raw ? 1u : 0u;
    dev.tb_hmul = c->tb_hmul;
    Arena A; A.base = c->scratch; A.cap = c->scratch_bytes; A.used = 0;
    const uint32_t nb64 = (n_reads + 63) / 64;
    const uint32_t n_waves = std::min<uint64_t>(n_reads, (uint64_t)c->n_cu * (uint64_t)c->waves_per_cu);

    RS *rs = (RS *)A.take((size_t)n_reads * sizeof(RS));
    uint64_t *totals = (uint64_t *)A.take(16 * sizeof(uint64_t));
    uint32_t *order = (uint32_t *)A.take((size_t)n_reads * 4);
    uint32_t *counters = (uint32_t *)A.take(2048 * 4);      /* [0] join queue, [1] flags, [2] window misses of the final stage, [16 + 16 x (phase, chunk)] final-stage queue heads */
    uint64_t *units_sorted = (uint64_t *)A.take((size_t)n_reads * 8);
    uint64_t *tboff_sorted = (uint64_t *)A.take((size_t)n_reads * 8);
    uint64_t *clk = (uint64_t *)A.take((size_t)n_reads * 64);     /* per-read cycle counters, brx_last_read_cycles() */
    uint64_t *phase = (uint64_t *)A.take((size_t)n_reads * 64);   /* mutate phase cycles (BRX_PROFILE=1), brx_last_phase_cycles() */
    if (!A.ok()) return scratch_short(c, A.used + (size_t)n_reads * 200000);
    HIPCHK(c, hipMemsetAsync(counters, 0, 2048 * 4, st));
    HIPCHK(c, hipMemsetAsync(totals, 0, 16 * 8, st));
    HIPCHK(c, hipMemsetAsync(clk, 0, (size_t)n_reads * 64, st));
    HIPCHK(c, hipMemsetAsync(phase, 0, (size_t)n_reads * 64, st));
    c->d_clk = clk; c->d_phase = phase; c->clk_reads = n_reads;

    /* ---- stage: plan ---- */
    HIPCHK(c, hipEventRecord(c->ev_b[BRX_STAGE_PLAN], st));
    if (raw) hipLaunchKernelGGL(k_init_raw, dim3(nb64), dim3(64), 0, st, dev, rs, d_frag_off, d_target);
    else hipLaunchKernelGGL(k_plan_count, dim3(nb64), dim3(64), 0, st, dev, rs);
    hipLaunchKernelGGL(k_scan_plan, dim3(1), dim3(64), 0, st, n_reads, rs, totals);
    int rc = read_totals(c, st, totals, 3);
    if (rc) return rc;
    const uint64_t tot_segs = c->h_totals[0], tot_pieces = c->h_totals[1], f_bytes = c->h_totals[2];
    PSeg *segs = (PSeg *)A.take((size_t)(tot_segs + 1) * sizeof(PSeg));
    PPiece *pieces = (PPiece *)A.take((size_t)(tot_pieces + 1) * sizeof(PPiece));
    uint8_t *Fbuf = (uint8_t *)A.take((size_t)f_bytes + 64);
    uint32_t *repl = (uint32_t *)A.take(((size_t)f_bytes + 64) * 4);
    const uint32_t side_waves = std::min<uint32_t>(n_reads, 4096u);                 /* wave-level window aligner / legacy */
    const uint32_t lane_waves = std::min<uint32_t>((n_reads + 63) / 64, 512u);      /* lane-level window aligner          */
    uint8_t *win = (uint8_t *)A.take((size_t)side_waves * c->win_bytes);
    MS *msv = (MS *)A.take((size_t)n_reads * sizeof(MS));
    uint32_t *mctr = (uint32_t *)A.take(4 * MC_WORDS * sizeof(uint32_t));
    uint32_t *active_a = (uint32_t *)A.take((size_t)n_reads * 4);
    uint32_t *active_b = (uint32_t *)A.take((size_t)n_reads * 4);
    uint32_t *req_easy = (uint32_t *)A.take((size_t)n_reads * 4);
    uint32_t *req_hard = (uint32_t *)A.take((size_t)n_reads * 4);
    uint32_t *req_legacy = (uint32_t *)A.take((size_t)n_reads * 4);
    uint8_t *winbuf = (uint8_t *)A.take((size_t)n_reads * BRX_WIN_STRIDE + 64);
    uint2 *lane_tb = (uint2 *)A.take((size_t)lane_waves * BRX_LANE_TB_UNITS * sizeof(uint2));
    if (!A.ok()) return scratch_short(c, A.used + (size_t)f_bytes * 6 + ((size_t)1 << 28));
    if (!raw) hipLaunchKernelGGL(k_plan_fill, dim3(nb64), dim3(64), 0, st, dev, rs, segs, pieces);
    HIPCHK(c, hipEventRecord(c->ev_e[BRX_STAGE_PLAN], st));
    HIPCHK(c, hipEventRecord(c->ev_b[BRX_STAGE_BUILD], st));

    /* ---- stage: build ---- */
    if (raw) hipLaunchKernelGGL(k_copy_frags, dim3(n_reads), dim3(64), 0, st, dev, rs, d_frags, d_frag_off, Fbuf);
    hipLaunchKernelGGL(k_build, dim3(n_reads), dim3(64), 0, st, dev, rs, segs, Fbuf, repl);
    hipLaunchKernelGGL(k_order, dim3(1), dim3(64), 0, st, n_reads, rs, order);
    HIPCHK(c, hipEventRecord(c->ev_e[BRX_STAGE_BUILD], st));
    HIPCHK(c, hipEventRecord(c->ev_b[BRX_STAGE_MUTATE], st));

    /* ---- stage: mutate (multi-pass: segments of the loop, parked window alignments; brx_mutate.h) ---- */
    {
        const uint32_t seg_waves = std::min<uint64_t>(n_reads, (uint64_t)c->n_cu * (uint64_t)c->seg_waves_per_cu);
        HIPCHK(c, hipMemsetAsync(msv, 0, (size_t)n_reads * sizeof(MS), st));
        HIPCHK(c, hipMemsetAsync(mctr, 0, 4 * MC_WORDS * sizeof(uint32_t), st));
        uint32_t *h_ctr = reinterpret_cast<uint32_t *>(c->h_totals + 8);          /* pinned */
        memset(h_ctr, 0, MC_WORDS * sizeof(uint32_t));
        h_ctr[MC_OUT] = n_reads;
        HIPCHK(c, hipMemcpyAsync(mctr + 2 * MC_WORDS, h_ctr, MC_WORDS * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipStreamSynchronize(st));
        uint32_t *legacy_ctr = mctr + 3 * MC_WORDS;                                 /* [0] count, [1] queue; not reset per pass */
        const uint32_t *n_in = mctr + 2 * MC_WORDS + MC_OUT;
        const uint32_t *act_in = order;
        uint32_t n_up = n_reads, pass = 0;
        const uint32_t lane_threshold = c->lane_threshold;   /* fewer active reads than this: one wave per window (lower latency) */
        /* this few reads left: run them to completion on the GPU, aligning in place (no host round trips).
           BRX_MUTATE_INLINE=1 does that for the whole batch: one launch, no passes. */
        const uint32_t tail_reads = c->mutate_inline ? 0xFFFFFFFFu : c->tail_reads;
        auto read_counts = [&](uint32_t *ctr) -> int {
            HIPCHK(c, hipMemcpyAsync(h_ctr, ctr, MC_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            return wait_stream(c, st, "mutate pass");
        };
        for (; n_up > 0 && pass < (1u << 20); ++pass) {
            uint32_t *ctr = mctr + (pass & 1u) * MC_WORDS;
            uint32_t *act_out = (pass & 1u) ? active_b : active_a;
            HIPCHK(c, hipMemsetAsync(ctr, 0, MC_WORDS * sizeof(uint32_t), st));
            if (n_up <= tail_reads) {
                if (c->profile)
                    hipLaunchKernelGGL((k_mutate_seg<true, true>), dim3(std::min(n_up, side_waves)), dim3(64), 0, st, dev, rs, msv, act_in, n_in, act_out, ctr,
                                       req_easy, req_hard, req_legacy, legacy_ctr, Fbuf, repl, winbuf, clk, lane_threshold,
                                       win, (uint64_t)c->win_bytes, counters + 1, phase);
                else
                    hipLaunchKernelGGL((k_mutate_seg<true, false>), dim3(std::min(n_up, side_waves)), dim3(64), 0, st, dev, rs, msv, act_in, n_in, act_out, ctr,
                                       req_easy, req_hard, req_legacy, legacy_ctr, Fbuf, repl, winbuf, clk, lane_threshold,
                                       win, (uint64_t)c->win_bytes, counters + 1, phase);
                rc = read_counts(ctr);
                if (rc) return rc;
                n_up = h_ctr[MC_OUT];                   /* 0 unless a window overflowed its slot (then: legacy list) */
                ++pass;
                break;
            }
            if (c->profile)
                hipLaunchKernelGGL((k_mutate_seg<false, true>), dim3(std::min(seg_waves, n_up)), dim3(64), 0, st, dev, rs, msv, act_in, n_in, act_out,
                                   ctr, req_easy, req_hard, req_legacy, legacy_ctr, Fbuf, repl, winbuf, clk, lane_threshold,
                                   win, (uint64_t)c->win_bytes, counters + 1, phase);
            else
                hipLaunchKernelGGL((k_mutate_seg<false, false>), dim3(std::min(seg_waves, n_up)), dim3(64), 0, st, dev, rs, msv, act_in, n_in, act_out,
                                   ctr, req_easy, req_hard, req_legacy, legacy_ctr, Fbuf, repl, winbuf, clk, lane_threshold,
                                   win, (uint64_t)c->win_bytes, counters + 1, phase);
            if (n_up > lane_threshold)
                hipLaunchKernelGGL(k_win_lane, dim3(std::min(lane_waves, (n_up + 63) / 64)), dim3(64), 0, st, msv, req_easy,
                                   ctr + MC_EASY, winbuf, lane_tb);
            hipLaunchKernelGGL(k_win_wave, dim3(std::min(side_waves, n_up)), dim3(64), 0, st, msv, req_hard, ctr + MC_HARD,
                               ctr + 5, winbuf, win, (uint64_t)c->win_bytes, counters + 1);
            /* the active count only shrinks: look at it every 4th pass while it is large, every pass near the end */
            if (n_up <= 4 * std::min<uint32_t>(tail_reads, 48u) || n_up <= tail_reads + 64 || (pass & 3u) == 3u) {
                rc = read_counts(ctr);
                if (rc) return rc;
                n_up = h_ctr[MC_OUT];
            }
            n_in = ctr + MC_OUT;
            act_in = act_out;
        }
        c->mutate_passes = pass;
        if (n_up > 0) return fail(c, BRX_E_INTERNAL, "mutate pipeline did not converge after %u passes", pass);
        /* reads whose window did not fit a slot: the whole-read kernel with the inline wave aligner */
        HIPCHK(c, hipMemcpyAsync(h_ctr, legacy_ctr, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (h_ctr[0] > 0)
            hipLaunchKernelGGL(k_mutate, dim3(std::min(side_waves, h_ctr[0])), dim3(64), 0, st, dev, rs, req_legacy, legacy_ctr,
                               legacy_ctr + 1, Fbuf, repl, win, (uint64_t)c->win_bytes, counters + 1, clk);
    }
    HIPCHK(c, hipEventRecord(c->ev_e[BRX_STAGE_MUTATE], st));
    HIPCHK(c, hipEventRecord(c->ev_b[BRX_STAGE_SCAN], st));
    hipLaunchKernelGGL(k_scan_mut, dim3(1), dim3(64), 0, st, n_reads, rs, totals);
    HIPCHK(c, hipEventRecord(c->ev_e[BRX_STAGE_SCAN], st));
    rc = read_totals(c, st, totals, 5);
    if (rc) return rc;
    {
        uint32_t flags = 0;
        HIPCHK(c, hipMemcpyAsync(&flags, counters + 1, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (flags & 1u) {                              /* an in-loop alignment did not fit its window scratch */
            c->win_bytes *= 4;
            return scratch_short(c, c->scratch_bytes + (size_t)side_waves * c->win_bytes);
        }
    }
    const uint64_t seq_bytes = c->h_totals[3], ops_bytes = c->h_totals[4];
    uint8_t *seqbuf = (uint8_t *)A.take((size_t)seq_bytes + 64);
    uint8_t *opsbuf = (uint8_t *)A.take((size_t)ops_bytes + 64);
    if (!A.ok()) return scratch_short(c, A.used + ((size_t)1 << 28));

    /* ---- stage: final alignment + qscores, in chunks that fit the remaining arena ----
     * phase 0: every read, windowed traceback store (sizes computed on the device at the end of mutate);
     * phase 1: only the reads whose traceback left the stored window (normally none), full store. */
    HIPCHK(c, hipEventRecord(c->ev_b[BRX_STAGE_FINAL], st));
    hipLaunchKernelGGL(k_fin_join, dim3(std::min<uint64_t>(n_reads, (uint64_t)c->n_cu * 16u)), dim3(64), 0, st, dev, rs, counters + 0,
                       Fbuf, repl, pieces, seqbuf);
    std::vector<uint32_t> h_order(n_reads);
    std::vector<RS> h_rs(n_reads);
    HIPCHK(c, hipMemcpyAsync(h_order.data(), order, (size_t)n_reads * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(h_rs.data(), rs, (size_t)n_reads * sizeof(RS), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const size_t tb_at = (A.used + 255) & ~(size_t)255;
    const size_t tb_cap = c->scratch_bytes > tb_at ? c->scratch_bytes - tb_at : 0;
    uint8_t *tb_base = c->scratch + tb_at;
    std::vector<uint64_t> h_tboff(n_reads), h_units(n_reads);
    size_t timed_chunks = 0;
    c->final_launches = 0;
    c->window_misses = 0;
    for (int phase = 0; phase < 2; ++phase) {
        if (phase == 1) {
            uint32_t misses = 0;
            HIPCHK(c, hipMemcpyAsync(&misses, counters + 2, 4, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            c->window_misses = misses;
            if (!misses) break;
            HIPCHK(c, hipMemcpyAsync(h_rs.data(), rs, (size_t)n_reads * sizeof(RS), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
        }
        uint64_t max_units = 0, sum_units = 0;
        for (uint32_t i = 0; i < n_reads; ++i) {
            const RS &r = h_rs[h_order[i]];
            uint64_t u = r.units;
            if (phase == 1) {
                bool too_wide;
                u = (r.n && (r.klass & BRX_KL_RETRY)) ? brx_final_units(r.m, r.n, r.ub, 0, &too_wide) : 0;
            }
            h_units[i] = u;
            max_units = std::max(max_units, u); sum_units += u;
        }
        if ((max_units + 64) * 8 > tb_cap) {
            size_t want = std::min<uint64_t>((sum_units + 64ull * n_reads) * 8, (uint64_t)8 << 30);
            return scratch_short(c, tb_at + std::max<size_t>((size_t)(max_units + 64) * 8, want));
        }
        std::vector<std::pair<uint32_t, uint32_t>> chunks;
        {
            uint32_t begin = 0; uint64_t used = 0;
            for (uint32_t i = 0; i < n_reads; ++i) {
                uint64_t need = ((h_units[i] + 31) & ~31ull) * 8;      /* 256-byte granules */
                if (used + need > tb_cap) { chunks.push_back({begin, i}); begin = i; used = 0; }
                h_tboff[i] = used; used += need;
            }
            chunks.push_back({begin, n_reads});
        }
        if (chunks.size() > BRX_MAX_CHUNKS) return scratch_short(c, tb_at + (size_t)std::min<uint64_t>((sum_units + 64ull * n_reads) * 8 / 8 + 1, (uint64_t)64 << 30));
        /* tb_off (and, in phase 1, the full-band units) go back through staging arrays in processing order */
        HIPCHK(c, hipMemcpyAsync(tboff_sorted, h_tboff.data(), (size_t)n_reads * 8, hipMemcpyHostToDevice, st));
        if (phase == 1) HIPCHK(c, hipMemcpyAsync(units_sorted, h_units.data(), (size_t)n_reads * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_set_tboff, dim3(nb64), dim3(64), 0, st, n_reads, rs, order, tboff_sorted, phase == 1 ? units_sorted : (uint64_t *)nullptr);
        for (size_t ci = 0; ci < chunks.size(); ++ci) {
            uint32_t b = chunks[ci].first, e = chunks[ci].second;
            if (e == b) continue;
            uint32_t waves = std::min<uint64_t>(e - b, (uint64_t)c->n_cu * (uint64_t)c->waves_per_cu);
            uint32_t *cq = counters + 16 + 16 * ((size_t)phase * BRX_MAX_CHUNKS + ci);     /* this chunk's queue heads */
            /* The widest bands (8+ words per lane: a few dozen reads, but each a chain of ~100 k column steps of
               ~2 us) and the 4-word class start first, on the side stream; the main stream aligns the one- and
               two-word classes (most of the reads) beside them and scores those reads without waiting; the wide
               reads are scored after the join. */
            HIPCHK(c, hipEventRecord(c->ev_fork, st));
            HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
            hipLaunchKernelGGL((k_fin_align<16, 8, 64>), dim3(std::min<uint32_t>(waves, (uint32_t)c->n_cu * 4u)), dim3(64), 0, c->side,
                               dev, rs, order, b, e, cq + 0, counters + 2, phase, Fbuf, seqbuf, opsbuf, tb_base, clk);
            hipLaunchKernelGGL((k_fin_align<4, 4, 4>), dim3(std::min<uint32_t>(waves, (uint32_t)c->n_cu * 8u)), dim3(64), 0, c->side,
                               dev, rs, order, b, e, cq + 4, counters + 2, phase, Fbuf, seqbuf, opsbuf, tb_base, clk);
            if (!c->fin_balance)
                hipLaunchKernelGGL((k_fin_align<2, 2, 2>), dim3(waves), dim3(64), 0, c->side,
                                   dev, rs, order, b, e, cq + 1, counters + 2, phase, Fbuf, seqbuf, opsbuf, tb_base, clk);
            HIPCHK(c, hipEventRecord(c->ev_join, c->side));
            const bool timed = phase == 0;
            if (timed) HIPCHK(c, hipEventRecord(c->ev_a1b[ci], st));
            hipLaunchKernelGGL((k_fin_align<1, 1, 1>), dim3(waves), dim3(64), 0, st, dev, rs, order, b, e, cq + 2, counters + 2, phase,
                               Fbuf, seqbuf, opsbuf, tb_base, clk);
            if (timed) HIPCHK(c, hipEventRecord(c->ev_a1e[ci], st));
            if (c->fin_balance)
                hipLaunchKernelGGL((k_fin_align<2, 2, 2>), dim3(waves), dim3(64), 0, st,
                                   dev, rs, order, b, e, cq + 1, counters + 2, phase, Fbuf, seqbuf, opsbuf, tb_base, clk);
            if (timed) HIPCHK(c, hipEventRecord(c->ev_qsb[ci], st));
            hipLaunchKernelGGL(k_fin_qscore, dim3(waves), dim3(64), 0, st, dev, rs, order, b, e, cq + 3, phase, 1, c->fin_balance ? 2 : 1,
                               seqbuf, opsbuf, tb_base, clk);
            if (timed) HIPCHK(c, hipEventRecord(c->ev_qse[ci], st));
            HIPCHK(c, hipStreamWaitEvent(st, c->ev_join, 0));
            hipLaunchKernelGGL(k_fin_qscore, dim3(std::min<uint32_t>(waves, (uint32_t)c->n_cu * 8u)), dim3(64), 0, st, dev, rs, order, b, e,
                               cq + 5, phase, c->fin_balance ? 3 : 2, 0xFFFF, seqbuf, opsbuf, tb_base, clk);
        }
        if (phase == 0) { timed_chunks = chunks.size(); c->final_launches = (uint32_t)chunks.size(); }
    }
    HIPCHK(c, hipEventRecord(c->ev_e[BRX_STAGE_FINAL], st));
    HIPCHK(c, hipEventRecord(c->ev_b[BRX_STAGE_EMIT], st));

    /* ---- stage: records ---- */
    hipLaunchKernelGGL(k_recsize, dim3(nb64), dim3(64), 0, st, dev, rs, pieces);
    hipLaunchKernelGGL(k_scan_rec, dim3(1), dim3(64), 0, st, n_reads, rs, totals);
    rc = read_totals(c, st, totals, 6);
    if (rc) return rc;
    const uint64_t rec_bytes = c->h_totals[5];
    if (rec_bytes > out_cap) {
        c->output_needed = rec_bytes;
        return fail(c, BRX_E_OUTPUT, "output buffer too small: need %llu bytes", (unsigned long long)rec_bytes);
    }
    hipLaunchKernelGGL(k_emit, dim3(n_reads), dim3(64), 0, st, dev, rs, pieces, seqbuf, d_out);
    hipLaunchKernelGGL(k_stats, dim3(nb64), dim3(64), 0, st, dev, rs, d_stats);
    HIPCHK(c, hipEventRecord(c->ev_e[BRX_STAGE_EMIT], st));
    { int rcw = wait_stream(c, st, "final stage / k_emit"); if (rcw) return rcw; }
    HIPCHK(c, hipGetLastError());
    for (int i = 0; i < BRX_STAGE_COUNT; ++i) {
        float ms = 0.f;
        if (i != BRX_STAGE_ALIGN1 && i != BRX_STAGE_QSCORE) (void)hipEventElapsedTime(&ms, c->ev_b[i], c->ev_e[i]);
        c->stage_ms[i] = ms;
    }
    /* per-launch AVERAGE over the scratch chunks, which is what a kernel trace reports */
    for (size_t ci = 0; ci < timed_chunks; ++ci) {
        float a = 0.f, q = 0.f;
        (void)hipEventElapsedTime(&a, c->ev_a1b[ci], c->ev_a1e[ci]);
        (void)hipEventElapsedTime(&q, c->ev_qsb[ci], c->ev_qse[ci]);
        c->stage_ms[BRX_STAGE_ALIGN1] += a / (float)timed_chunks;
        c->stage_ms[BRX_STAGE_QSCORE] += q / (float)timed_chunks;
    }
    if (out_bytes) *out_bytes = (size_t)rec_bytes;
    /* a read that exhausted its 1000 tries is fatal in the reference (simulate.py:164) */
    if (!raw) {
        HIPCHK(c, hipMemcpyAsync(h_rs.data(), rs, (size_t)n_reads * sizeof(RS), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        for (uint32_t i = 0; i < n_reads; ++i) if (h_rs[i].status & BRX_RS_NOFRAG) {
            snprintf(c->err, sizeof(c->err), "read %llu failed to generate a sequence fragment", (unsigned long long)(first_read + i));
            return BRX_E_NOFRAG;
        }
    }
    return BRX_OK;
}

extern "C" int brx_simulate_batch(brx_ctx *c, uint64_t seed, uint64_t first_read, uint32_t n_reads,
                                  uint8_t *d_out, size_t out_cap, brx_read_stats *d_stats,
                                  size_t *out_bytes, void *hip_stream) {
    if (!c || !d_out || !d_stats) return BRX_E_ARG;
    return run_pipeline(c, seed, first_read, n_reads, false, nullptr, nullptr, nullptr, d_out, out_cap, d_stats, out_bytes,
                        (hipStream_t)hip_stream);
}

extern "C" int brx_sequence_fragments(brx_ctx *c, uint64_t seed, uint64_t first_read, uint32_t n_frags,
                                      const uint8_t *d_frags, const uint64_t *d_frag_off, const double *d_target,
                                      uint8_t *d_out, size_t out_cap, brx_read_stats *d_stats, size_t *out_bytes,
                                      void *hip_stream) {
    if (!c || !d_out || !d_stats || !d_frags || !d_frag_off || !d_target) return BRX_E_ARG;
    return run_pipeline(c, seed, first_read, n_frags, true, d_frags, d_frag_off, d_target, d_out, out_cap, d_stats, out_bytes,
                        (hipStream_t)hip_stream);
}

extern "C" int brx_align_batch(brx_ctx *c, uint32_t n_pairs, const uint8_t *d_queries, const uint64_t *d_q_off,
                               const uint8_t *d_targets, const uint64_t *d_t_off, const int32_t *d_k_hint,
                               int32_t *d_dist, uint32_t *d_ncols, uint32_t *d_nmatch, uint8_t *d_ops,
                               const uint64_t *d_ops_off, void *hip_stream) {
    if (!c || !d_q_off || !d_t_off || !d_k_hint || !d_dist || !d_ncols || !d_nmatch) return BRX_E_ARG;
    if (!c->scratch) return fail(c, BRX_E_STATE, "scratch arena not set");
    if (n_pairs == 0) return BRX_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    HIPCHK(c, hipSetDevice(c->device));
    DBG("align_batch: %u pairs", n_pairs);
    std::vector<uint64_t> qo(n_pairs + 1), to(n_pairs + 1);
    std::vector<int32_t> kh(n_pairs);
    HIPCHK(c, hipMemcpyAsync(qo.data(), d_q_off, (size_t)(n_pairs + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(to.data(), d_t_off, (size_t)(n_pairs + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(kh.data(), d_k_hint, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    DBG("align_batch: offsets copied");
    Arena A; A.base = c->scratch; A.cap = c->scratch_bytes; A.used = 0;
    uint64_t *scr_off = (uint64_t *)A.take((size_t)n_pairs * 8);
    uint64_t *scr_bytes = (uint64_t *)A.take((size_t)n_pairs * 8);
    uint32_t *counters = (uint32_t *)A.take(4096 * 4);
    if (!A.ok()) return scratch_short(c, A.used + ((size_t)1 << 26));
    size_t at = (A.used + 255) & ~(size_t)255;
    size_t cap = c->scratch_bytes - at;
    std::vector<uint64_t> h_off(n_pairs), h_bytes(n_pairs);
    std::vector<std::pair<uint32_t, uint32_t>> chunks;
    uint32_t begin = 0; uint64_t used = 0, biggest = 0;
    for (uint32_t i = 0; i < n_pairs; ++i) {
        uint64_t Q = qo[i + 1] - qo[i], T = to[i + 1] - to[i];
        if (Q >= ((uint64_t)1 << 30) || T >= ((uint64_t)1 << 30)) return fail(c, BRX_E_ARG, "sequence %u too long", i);
        /* the band-doubling rounds of the kernel (k = 64, 128, ... capped at max(Q,T)) do not need
           monotonically more traceback store: size the pair for the largest of them */
        uint64_t units = 0;
        if (Q && T) {
            const int maxk = (int)std::max(Q, T);
            int k = kh[i] >= 0 ? kh[i] : std::min(maxk, 64);
            for (;;) {
                BrxGeom g = brx_make_geom((int)Q, (int)T, k);
                if (g.G) units = std::max(units, brx_align_units(g));
                if (kh[i] >= 0 || k >= maxk) break;
                k = k * 2 > maxk ? maxk : k * 2;
            }
        }
        uint64_t need = (((Q + 31) & ~15ull) + ((T + 31) & ~15ull) + (units + 8) * 8 + 255) & ~255ull;
        biggest = std::max(biggest, need);
        if (used + need > cap) { chunks.push_back({begin, i}); begin = i; used = 0; }
        h_off[i] = used; h_bytes[i] = need; used += need;
    }
    chunks.push_back({begin, n_pairs});
    if (biggest > cap) return scratch_short(c, at + (size_t)biggest);
    if (chunks.size() > 4000) return scratch_short(c, at + (size_t)biggest * 64);
    HIPCHK(c, hipMemcpyAsync(scr_off, h_off.data(), (size_t)n_pairs * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(scr_bytes, h_bytes.data(), (size_t)n_pairs * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemsetAsync(counters, 0, 4096 * 4, st));
    HIPCHK(c, hipEventRecord(c->ev_b[BRX_STAGE_FINAL], st));
    for (size_t ci = 0; ci < chunks.size(); ++ci) {
        uint32_t b = chunks[ci].first, e = chunks[ci].second;
        if (e == b) continue;
        uint32_t waves = std::min<uint64_t>(e - b, (uint64_t)c->n_cu * (uint64_t)c->waves_per_cu);
        DBG("align_batch: chunk %zu pairs [%u,%u) waves %u", ci, b, e, waves);
        hipLaunchKernelGGL(k_align_batch, dim3(waves), dim3(64), 0, st, n_pairs, b, e, counters + ci, d_queries, d_q_off,
                           d_targets, d_t_off, d_k_hint, d_dist, d_ncols, d_nmatch, d_ops, d_ops_off,
                           c->scratch + at, scr_off, scr_bytes, brx_debug() ? c->d_prog : (uint32_t *)nullptr);
    }
    HIPCHK(c, hipEventRecord(c->ev_e[BRX_STAGE_FINAL], st));
    DBG("align_batch: launched, waiting");
    { int rcw = wait_stream(c, st, "k_align_batch"); if (rcw) return rcw; }
    for (int i = 0; i < BRX_STAGE_COUNT; ++i) c->stage_ms[i] = 0.f;
    (void)hipEventElapsedTime(&c->stage_ms[BRX_STAGE_FINAL], c->ev_b[BRX_STAGE_FINAL], c->ev_e[BRX_STAGE_FINAL]);
    c->final_launches = (uint32_t)chunks.size();
    HIPCHK(c, hipGetLastError());
    DBG("align_batch: done");
    return BRX_OK;
}

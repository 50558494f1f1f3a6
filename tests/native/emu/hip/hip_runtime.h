/*
 * TEST INFRASTRUCTURE -- a CPU stand-in for <hip/hip_runtime.h>, used only by tests/test_emulated_device.py.
 *
 * tests/ compile the PRODUCT SOURCES (badread_amd/csrc/brx_hip.hip with brx_kernels.h, brx_mutate.h, brx_align.h)
 * with g++ and this directory first on the include path.  Every lane of a wavefront becomes a fiber (ucontext) of
 * one OS thread; lanes run until they reach a cross-lane operation (__ballot, __shfl*, DPP wave_ror, readfirstlane,
 * __syncthreads), deposit their operand and yield; the operation completes when all live lanes of the wave have
 * arrived.  Workgroups of a launch run one after another, launches are synchronous, "device memory" is host memory,
 * streams and events are clocks.  This is an interpreter for the wave-level semantics the kernels rely on -- NOT a
 * fallback: nothing under badread_amd/ can load the resulting library, and it is ~1000x slower than the oracle.
 * What it buys: the parity tests of the real kernel source run in the CPU test suite, so a logic error in a kernel
 * is caught without a GPU (integer overflow, LDS ring indexing, traceback addressing ...).
 *
 * Limits: workgroups of up to 16 waves (64 x W threads); cross-lane operations must be reached by all live lanes of the wave
 * (true for these kernels by construction; a lane that never arrives is reported as a deadlock); no timing fidelity.
 */
#ifndef BRX_HIP_EMU_H
#define BRX_HIP_EMU_H

#include <execinfo.h>
#include <signal.h>
#include <sys/mman.h>
#include <ucontext.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <vector>

#define __HIP_EMU__ 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define BRX_GLOBAL                 /* address-space qualifier of the product's explicit global-memory stores: one flat memory here */
#define __shared__ static thread_local   /* one workgroup runs at a time per host thread: a static IS its LDS */

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 v; v.x = x; v.y = y; return v; }
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }

namespace emu {

constexpr int WAVE = 64;
constexpr int MAXW = 16;                          /* waves per workgroup (1024 threads) */
constexpr size_t STACK_BYTES = 8u << 20;          /* per lane; mmap'ed, so only touched pages cost memory */

/* One workgroup = up to MAXW waves.  Cross-lane operations rendezvous the live lanes of ONE wave; __syncthreads
 * rendezvous every live lane of the workgroup.  The scheduler round-robins over all lanes of all waves, so waves of a
 * workgroup interleave at their cross-lane operations (a wave spinning on LDS written by another wave makes progress
 * as long as it reaches cross-lane operations or calls emu::spin_yield() inside the spin). */
struct Wave {
    int live = 0, arrived = 0;
    unsigned long long gen = 0;                     /* completed cross-lane operations of this wave */
    uint64_t slot[2][WAVE];                         /* operands, double-buffered by operation parity */
    uint64_t part[2] = {0, 0};                      /* lanes that took part in the operation of each buffer: a lane that has
                                                       left the kernel since then still counts for the slower readers */
};
struct State {
    ucontext_t sched;
    ucontext_t lane_ctx[MAXW * WAVE];
    char *stacks = nullptr;
    int stacks_for = 0;
    bool done[MAXW * WAVE];
    Wave wave[MAXW];
    int n_threads = WAVE, n_waves = 1;
    int block_live = 0, bar_arrived = 0;
    unsigned long long bar_gen = 0;
    int cur = 0;                                    /* running thread of the workgroup */
    unsigned block = 0, grid = 1;
    const std::function<void()> *body = nullptr;
    unsigned long long clock = 0;
    unsigned long long yields = 0;
};
inline State &S() { static thread_local State s; return s; }     /* host threads (contexts in flight) do not share it */

inline void yield_to_scheduler() { State &s = S(); swapcontext(&s.lane_ctx[s.cur], &s.sched); }
inline void spin_yield() { State &s = S(); s.yields += 1; yield_to_scheduler(); }

/* all live lanes of the running lane's wave deposit `v`; returns the buffer holding every lane's operand */
inline const uint64_t *exchange(uint64_t v) {
    State &s = S();
    Wave &w = s.wave[s.cur / WAVE];
    const unsigned long long g = w.gen;
    uint64_t *buf = w.slot[g & 1];
    if (w.arrived == 0) w.part[g & 1] = 0;
    w.part[g & 1] |= 1ull << (s.cur % WAVE);
    buf[s.cur % WAVE] = v;
    if (++w.arrived >= w.live) { w.arrived = 0; w.gen = g + 1; }
    else while (w.gen == g) yield_to_scheduler();
    return buf;
}
/* the lanes that deposited into `buf` (the value exchange() returned) */
inline uint64_t participants(const uint64_t *buf) {
    State &s = S();
    Wave &w = s.wave[s.cur / WAVE];
    return w.part[buf == w.slot[1] ? 1 : 0];
}
inline void block_barrier() {
    State &s = S();
    const unsigned long long g = s.bar_gen;
    if (++s.bar_arrived >= s.block_live) { s.bar_arrived = 0; s.bar_gen = g + 1; }
    else while (s.bar_gen == g) yield_to_scheduler();
}

inline void lane_entry() {
    State &s = S();
    (*s.body)();
    s.done[s.cur] = true;                            /* its operand slots keep their last values: slower lanes may still be
                                                        reading the operation this lane has already left */
    Wave &w = s.wave[s.cur / WAVE];
    w.live -= 1;
    if (w.live > 0 && w.arrived >= w.live) { w.arrived = 0; w.gen += 1; }      /* the others were waiting for this lane */
    s.block_live -= 1;
    if (s.block_live > 0 && s.bar_arrived >= s.block_live) { s.bar_arrived = 0; s.bar_gen += 1; }
    yield_to_scheduler();
}

inline void run_block(unsigned block, unsigned grid, int n_threads, const std::function<void()> &body) {
    State &s = S();
    if (!s.stacks || s.stacks_for < n_threads) {
        if (s.stacks) munmap(s.stacks, STACK_BYTES * (size_t)s.stacks_for);
        s.stacks = (char *)mmap(nullptr, STACK_BYTES * (size_t)n_threads, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (s.stacks == (char *)MAP_FAILED) { perror("[hip_emu] mmap"); abort(); }
        s.stacks_for = n_threads;
    }
    s.block = block; s.grid = grid; s.body = &body;
    s.n_threads = n_threads; s.n_waves = n_threads / WAVE;
    s.block_live = n_threads; s.bar_arrived = 0; s.bar_gen = 0;
    for (int w = 0; w < s.n_waves; ++w) { s.wave[w].live = WAVE; s.wave[w].arrived = 0; s.wave[w].gen = 0; memset(s.wave[w].slot, 0, sizeof(s.wave[w].slot)); }
    for (int l = 0; l < n_threads; ++l) {
        s.done[l] = false;
        getcontext(&s.lane_ctx[l]);
        s.lane_ctx[l].uc_stack.ss_sp = s.stacks + STACK_BYTES * l;
        s.lane_ctx[l].uc_stack.ss_size = STACK_BYTES;
        s.lane_ctx[l].uc_link = &s.sched;
        makecontext(&s.lane_ctx[l], (void (*)())lane_entry, 0);
    }
    unsigned long long idle_rounds = 0, last_prog = ~0ull;
    while (s.block_live > 0) {
        for (int l = 0; l < n_threads; ++l) {
            if (s.done[l]) continue;
            s.cur = l;
            swapcontext(&s.sched, &s.lane_ctx[l]);
        }
        unsigned long long prog = s.bar_gen * 1315423911ull + (unsigned long long)s.block_live + s.yields * 7919ull;
        for (int w = 0; w < s.n_waves; ++w) prog = prog * 31ull + s.wave[w].gen;
        if (prog == last_prog) {
            if (++idle_rounds > 4) {
                fprintf(stderr, "[hip_emu] deadlock in block %u: %d live threads;", block, s.block_live);
                for (int w = 0; w < s.n_waves; ++w) fprintf(stderr, " wave %d: %d of %d at a cross-lane operation;", w, s.wave[w].arrived, s.wave[w].live);
                fprintf(stderr, " %d at __syncthreads\n", s.bar_arrived);
                abort();
            }
        } else { idle_rounds = 0; last_prog = prog; }
    }
}

inline void on_segv(int, siginfo_t *info, void *) {
    State &s = S();
    fprintf(stderr, "[hip_emu] SIGSEGV at %p in thread %d of block %u (stack of that lane: %p..%p)\n", info->si_addr, s.cur, s.block,
            (void *)(s.stacks + STACK_BYTES * s.cur), (void *)(s.stacks + STACK_BYTES * (s.cur + 1)));
    void *frames[48];
    const int n = backtrace(frames, 48);
    backtrace_symbols_fd(frames, n, 2);
    _exit(139);
}
inline void install_segv_trace() {
    static bool done = false;
    if (done || !getenv("BRX_EMU_TRACE")) return;
    done = true;
    static char alt[1 << 16];
    stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof(alt); ss.ss_flags = 0;
    sigaltstack(&ss, nullptr);
    struct sigaction sa; memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_segv; sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, nullptr);
}

template <class F> inline void launch(dim3 grid, dim3 block, F &&f) {
    install_segv_trace();
    if (block.x % WAVE != 0 || block.x == 0 || block.x > (unsigned)(MAXW * WAVE) || block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1) {
        fprintf(stderr, "[hip_emu] only (64 x W) x 1 x 1 workgroups, W <= %d\n", MAXW); abort();
    }
    std::function<void()> body(std::forward<F>(f));
    for (unsigned b = 0; b < grid.x; ++b) run_block(b, grid.x, (int)block.x, body);
}

struct Idx { unsigned x, y, z; };
inline Idx tidx() { return Idx{(unsigned)S().cur, 0u, 0u}; }
inline Idx bidx() { return Idx{S().block, 0u, 0u}; }
inline Idx bdim() { return Idx{(unsigned)S().n_threads, 1u, 1u}; }
inline Idx gdim() { return Idx{S().grid, 1u, 1u}; }

}  // namespace emu

#define threadIdx (emu::tidx())
#define blockIdx (emu::bidx())
#define blockDim (emu::bdim())
#define gridDim (emu::gdim())

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })

/* ---- cross-lane operations ------------------------------------------------------------------------------- */
static inline unsigned long long __ballot(int pred) {
    const uint64_t *v = emu::exchange(pred ? 1u : 0u);
    unsigned long long m = 0;
    const uint64_t part = emu::participants(v);                /* lanes that had left the kernel before this ballot: EXEC off */
    for (int l = 0; l < emu::WAVE; ++l) if ((part >> l) & 1ull) m |= (unsigned long long)(v[l] & 1u) << l;
    return m;
}
static inline int __shfl(int v, int src, int width = 64) { (void)width; const int me = emu::S().cur % emu::WAVE; const uint64_t *a = emu::exchange((uint32_t)v); (void)me; return (int)(uint32_t)a[src & 63]; }
static inline int __shfl_xor(int v, int mask, int width = 64) { (void)width; const int me = emu::S().cur % emu::WAVE; const uint64_t *a = emu::exchange((uint32_t)v); return (int)(uint32_t)a[(me ^ mask) & 63]; }
static inline int __shfl_up(int v, unsigned delta, int width = 64) { (void)width; const int me = emu::S().cur % emu::WAVE; const uint64_t *a = emu::exchange((uint32_t)v); return me >= (int)delta ? (int)(uint32_t)a[me - (int)delta] : v; }
static inline int __shfl_down(int v, unsigned delta, int width = 64) { (void)width; const int me = emu::S().cur % emu::WAVE; const uint64_t *a = emu::exchange((uint32_t)v); return me + (int)delta < 64 ? (int)(uint32_t)a[me + (int)delta] : v; }
static inline unsigned __shfl(unsigned v, int src, int width = 64) { return (unsigned)__shfl((int)v, src, width); }
static inline unsigned __shfl_xor(unsigned v, int mask, int width = 64) { return (unsigned)__shfl_xor((int)v, mask, width); }
static inline unsigned __shfl_up(unsigned v, unsigned delta, int width = 64) { return (unsigned)__shfl_up((int)v, delta, width); }
static inline void __syncthreads() { emu::block_barrier(); }
namespace emu {
inline int dpp(int v, int ctrl) {
    const int me = S().cur % WAVE;
    if ((ctrl & 0x1F0) == 0x120 && (ctrl & 15)) {                              /* row_ror:n: rotation inside rows of 16 lanes */
        const uint64_t *a = exchange((uint32_t)v);
        return (int)(uint32_t)a[(me & ~15) | ((me - (ctrl & 15)) & 15)];
    }
    if (ctrl != 0x13C) { fprintf(stderr, "[hip_emu] DPP control %#x not modelled\n", ctrl); abort(); }
    const uint64_t *a = exchange((uint32_t)v);
    return (int)(uint32_t)a[(me + 63) & 63];                                  /* wave_ror:1 */
}
inline int readfirstlane(int v) {
    const uint64_t *a = exchange((uint32_t)v);
    const uint64_t part = participants(a);
    for (int l = 0; l < WAVE; ++l) if ((part >> l) & 1ull) return (int)(uint32_t)a[l];
    return v;
}
}  // namespace emu
#define __builtin_amdgcn_update_dpp(old, v, ctrl, row_mask, bank_mask, bound_ctrl) emu::dpp((v), (ctrl))
#define __builtin_amdgcn_mov_dpp(v, ctrl, row_mask, bank_mask, bound_ctrl) emu::dpp((v), (ctrl))
#define __builtin_amdgcn_readfirstlane(v) emu::readfirstlane((int)(v))
#define __builtin_amdgcn_alignbit(hi, lo, sh) ((uint32_t)((((uint64_t)(uint32_t)(hi) << 32) | (uint64_t)(uint32_t)(lo)) >> ((sh) & 31)))   /* v_alignbit_b32 */
/* s_waitcnt separates "every lane has loaded" from "any lane stores" in lockstep code (the in-place shift of the ops in
   k_align_batch); all uses sit in wave-uniform control flow, so here it is a rendezvous as well */
#define __builtin_amdgcn_s_waitcnt(x) ((void)emu::exchange(0))
/* A fence is where the kernels publish one lane's stores to the other lanes of the wave (the hardware runs the lanes in
   lockstep; here a lane runs ahead until its next cross-lane operation), so the fence is a rendezvous. */
#define __builtin_amdgcn_fence(order, scope) ((void)emu::exchange(0))
/* a scheduling barrier of the compiler on the hardware (LDS operations of a wave run in order); here the point where the lanes
   that wrote LDS entries and the lanes that read them meet */
#define __builtin_amdgcn_wave_barrier() ((void)emu::exchange(0))
#define __builtin_amdgcn_s_memtime() (++emu::S().clock)
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __hip_atomic_store(ptr, val, order, scope) (*(ptr) = (val))
#define __HIP_MEMORY_SCOPE_AGENT 1
#define __hip_atomic_load(ptr, order, scope) (*(ptr))

/* ---- scalar helpers ------------------------------------------------------------------------------------- */
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
template <class T, class U> static inline T atomicAdd(T *p, U v) { T old = *p; *p = (T)(old + (T)v); return old; }
template <class T, class U> static inline T atomicOr(T *p, U v) { T old = *p; *p = (T)(old | (T)v); return old; }
template <class T, class U, class V> static inline T atomicCAS(T *p, U expected, V desired) { T old = *p; if (old == (T)expected) *p = (T)desired; return old; }
template <class T, class U> static inline T atomicMin(T *p, U v) { T old = *p; if ((T)v < old) *p = (T)v; return old; }
template <class T, class U> static inline T atomicMax(T *p, U v) { T old = *p; if ((T)v > old) *p = (T)v; return old; }

/* ---- runtime API: host memory, synchronous streams, wall-clock events ---------------------------------------- */
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorNotReady = 600, hipErrorInvalidValue = 1 };
typedef struct emu_stream { int id; } *hipStream_t;
typedef struct emu_event { double ms; } *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipEventBlockingSync = 1, hipHostMallocMapped = 2, hipHostMallocDefault = 0 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 16 };

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "emulated HIP error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 2; return hipSuccess; }   /* a 2-CU "chip": small grids */
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = new emu_stream{0}; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
static inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(struct emu_event *) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event{0.0}; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) {
    e->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->ms - a->ms); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }

#endif /* BRX_HIP_EMU_H */

// tools/native/valu_bench.hip -- MEASUREMENT TOOL (not part of the product): the wave64 issue rate of the 32-bit integer
// VALU instructions the aligners are made of (v_and/v_or/v_xor/v_add_u32/v_lshl/v_bfi), on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/native/valu_bench.hip -o /tmp/valu_bench && /tmp/valu_bench
// Every wave runs ITER trips of CHAINS independent dependency chains of 8 instructions each; the kernel is launched
// with W waves per SIMD on every SIMD of the chip, and the rate is instructions / (SIMDs x elapsed cycles).
// The result is the denominator of bench.py's `roofline_alu` (profiles/valu_rate.json).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int CHAINS>
__global__ void __launch_bounds__(64) k_valu(uint32_t *out, int iters, uint32_t a0, uint32_t b0) {
    uint32_t x[CHAINS], a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = a * (c + 1) + b;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            uint32_t v = x[c];
            // one Myers-like step: and, add, xor, or, or-not, and, shift-or, bfi
            uint32_t e = v & a;
            uint32_t s = e + b;
            uint32_t xh = (s ^ b) | v;
            uint32_t ph = a | ~(xh | b);
            uint32_t mh = b & xh;
            uint32_t phs = (ph << 1) | (v >> 31);
            uint32_t r = (phs & mh) | (v & ~mh);
            x[c] = r + phs;
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc ^= x[c];
    if (acc == 0x12345678u) out[blockIdx.x] = acc;          // keep the work alive
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[1] = (uint32_t)(t1 - t0); out[2] = (uint32_t)((t1 - t0) >> 32); }
}

// Control: the same harness on v_fma_f32, the instruction MI355X_MICROARCH.md quotes a two-cycle issue for.
// 8 dependent fmas per step and chain; if a wave64 v_fma_f32 really issued in 2 cycles, the chip would sustain twice the
// instructions per second of the integer kernel above.
template <int CHAINS>
__global__ void __launch_bounds__(64) k_fma(float *out, int iters, float a0, float b0) {
    float x[CHAINS], a = a0 + 1e-6f * threadIdx.x, b = b0 - 1e-6f * threadIdx.x;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = a * (c + 1) + b;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            float v = x[c];
            v = __builtin_fmaf(v, a, b); v = __builtin_fmaf(v, b, a); v = __builtin_fmaf(v, a, b); v = __builtin_fmaf(v, b, a);
            v = __builtin_fmaf(v, a, b); v = __builtin_fmaf(v, b, a); v = __builtin_fmaf(v, a, b); v = __builtin_fmaf(v, b, a);
            x[c] = v;
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc += x[c];
    if (acc == 1234.5f) out[blockIdx.x + 8] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { ((uint32_t *)out)[1] = (uint32_t)(t1 - t0); ((uint32_t *)out)[2] = (uint32_t)((t1 - t0) >> 32); }
}

template <int CHAINS>
static void run_fma(int waves_per_simd, int n_cu) {
    float *d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
    const int iters = 20000;
    const int blocks = n_cu * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_fma<CHAINS>), dim3(blocks), dim3(64), 0, 0, d, 100, 0.999f, 0.001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_fma<CHAINS>), dim3(blocks), dim3(64), 0, 0, d, iters, 0.999f, 0.001f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    uint32_t h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double cyc = (double)(((uint64_t)h[2] << 32) | h[1]);
    const double insts = (double)iters * CHAINS * 8.0;
    printf("{\"control\": \"v_fma_f32\", \"chains\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"wave_cycles_per_inst\": %.2f, \"insts_per_s_chip\": %.4g}\n",
           CHAINS, waves_per_simd, ms, cyc / insts, insts * blocks / (ms * 1e-3));
    hipFree(d);
}

template <int CHAINS>
static void run(int waves_per_simd, int n_cu) {
    uint32_t *d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
    const int iters = 20000;
    const int blocks = n_cu * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_valu<CHAINS>), dim3(blocks), dim3(64), 0, 0, d, 100, 3u, 5u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_valu<CHAINS>), dim3(blocks), dim3(64), 0, 0, d, iters, 3u, 5u);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    uint32_t h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double cyc = (double)(((uint64_t)h[2] << 32) | h[1]);
    // instruction count per trip per chain, from the source: and add xor or (or+not = v_or + v_not or v_nor) and lshl_or and bfi/and-or add  ~ 12-14 VALU; report per "step"
    const double steps = (double)iters * CHAINS;
    printf("{\"chains\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"wave_cycles_per_step\": %.2f, \"steps_per_s_chip\": %.4g}\n",
           CHAINS, waves_per_simd, ms, cyc / steps, steps * blocks / (ms * 1e-3));
    hipFree(d);
}

int main() {
    int dev = 0, n_cu = 256;
    hipSetDevice(dev);
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    printf("{\"n_cu\": %d}\n", n_cu);
    for (int w : {1, 2, 4, 8}) { run<1>(w, n_cu); run<4>(w, n_cu); run<8>(w, n_cu); }
    for (int w : {1, 4, 8}) { run_fma<1>(w, n_cu); run_fma<8>(w, n_cu); }
    return 0;
}

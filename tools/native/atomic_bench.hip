// Throughput of returning device-scope atomics from one lane per wave, as the list appends of the mutate passes issue them:
//   hipcc --offload-arch=gfx950 -O3 tools/native/atomic_bench.hip -o /tmp/atomic_bench && /tmp/atomic_bench
// mode 0: every wave adds to ONE address; 1: six addresses (blockIdx % 6); 2: one address per wave; 3: one address, wave-aggregated (64 per atomic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(64) k(uint32_t *ctr, uint32_t *out, int per_wave, int mode) {
    uint32_t acc = 0;
    for (int i = 0; i < per_wave; ++i) {
        uint32_t *p = mode == 0 ? ctr : mode == 1 ? ctr + 64 * (blockIdx.x % 6) : mode == 2 ? ctr + 64 * blockIdx.x : ctr;
        uint32_t v = atomicAdd(p, threadIdx.x == 0 ? (mode == 3 ? 64u : 1u) : 0u);
        v = __builtin_amdgcn_readfirstlane(v);
        acc += v;
        out[(blockIdx.x * 64 + threadIdx.x) ^ (acc & 1)] = acc;      // something that depends on the result
    }
}
int main() {
    uint32_t *ctr, *out; const int waves = 8192;
    hipMalloc(&ctr, 64 * 4 * (waves + 8)); hipMalloc(&out, waves * 64 * 4 + 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 4; ++mode) for (int per_wave : {1, 4, 16}) {
        hipMemset(ctr, 0, 64 * 4 * (waves + 8));
        hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, ctr, out, per_wave, mode);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, ctr, out, per_wave, mode);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("{\"mode\": %d, \"waves\": %d, \"atomics_per_wave\": %d, \"ms\": %.4f, \"ns_per_atomic\": %.2f}\n", mode, waves, per_wave, ms, ms * 1e6 / (waves * (double)per_wave));
    }
    return 0;
}

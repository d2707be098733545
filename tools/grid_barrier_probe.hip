// Micro-benchmark: what does a grid-wide barrier between phases of ONE persistent kernel cost on an MI355X, against a kernel boundary?
// (sizing of a persistent conv -> BatchNorm chain for the pyramid tail's small blocks)
//   hipcc -O3 --offload-arch=gfx950 tools/grid_barrier_probe.hip -o tools/build/grid_barrier_probe && tools/build/grid_barrier_probe
// Barrier: a monotonically increasing arrival counter (agent scope); workgroup w of G arrives with a release add after a workgroup
// barrier and thread 0 spins (bounded) with acquire loads until the count reaches G * (phase + 1).  Between barriers every workgroup
// touches `bytes_per_wg` of memory written by its neighbour in the previous phase (so that the release / acquire pair is not free).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) phases_kernel(unsigned* counter, float* buf, int floats_per_wg, int phases, unsigned* timed_out) {
    const int G = gridDim.x, w = blockIdx.x;
    float acc = 0.f;
    for (int p = 0; p < phases; ++p) {
        float* mine = buf + (size_t)w * floats_per_wg;
        const float* theirs = buf + (size_t)((w + 1) % G) * floats_per_wg;
        for (int i = threadIdx.x; i < floats_per_wg; i += 256) { acc += theirs[i]; mine[i] = acc + (float)p; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)G * (unsigned)(p + 1);
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) { atomicAdd(timed_out, 1u); break; }
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (acc == 123.456f) buf[0] = acc;
}
__global__ void __launch_bounds__(256) one_phase_kernel(float* buf, int floats_per_wg, int p) {
    const int G = gridDim.x, w = blockIdx.x;
    float acc = 0.f;
    float* mine = buf + (size_t)w * floats_per_wg;
    const float* theirs = buf + (size_t)((w + 1) % G) * floats_per_wg;
    for (int i = threadIdx.x; i < floats_per_wg; i += 256) { acc += theirs[i]; mine[i] = acc + (float)p; }
}
int main() {
    const int phases = 64;
    unsigned *counter, *timed_out;
    float* buf;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&timed_out, 4));
    CK(hipMalloc(&buf, (size_t)1024 * 65536 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("workgroups | floats per wg and phase | persistent kernel, us per phase | separate launches, us per phase | timeouts\n");
    for (int G : {32, 64, 128, 256, 512}) {
        for (int fl : {256, 4096, 65536}) {
            float best_p = 1e9f, best_s = 1e9f;
            unsigned to = 0;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemset(counter, 0, 4)); CK(hipMemset(timed_out, 0, 4));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(phases_kernel, dim3(G), dim3(256), 0, 0, counter, buf, fl, phases, timed_out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best_p = ms < best_p ? ms : best_p;
                CK(hipMemcpy(&to, timed_out, 4, hipMemcpyDeviceToHost));
                CK(hipEventRecord(e0));
                for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(one_phase_kernel, dim3(G), dim3(256), 0, 0, buf, fl, p);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                best_s = ms < best_s ? ms : best_s;
            }
            printf("%10d | %23d | %31.2f | %31.2f | %u\n", G, fl, best_p * 1e3f / phases, best_s * 1e3f / phases, to);
        }
    }
    return 0;
}

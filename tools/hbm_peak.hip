// Micro-benchmark: what a plain streaming kernel reaches on an MI355X at the buffer sizes libssdk's HBM-bound kernels move
// (hnm_rows / loss_bwd / post_select read batch x 8732 x 81 floats = 90.5 MB at batch 32, 181 MB at batch 64), so that their
// fraction of the 8 TB/s peak can be read against what the size itself allows (launch ramp + tail of a 20-40 us kernel).
//   read : float4 grid-stride loads, one float per workgroup written back
//   copy : float4 loads + float4 stores of the same size (pack_dy_kernel's shape: bytes counted = read + written)
//   4B   : the same two with one dword per lane
// Buffers are rotated over > 1 GiB so that nothing is served by the 256 MB infinity cache or the L2s.
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_peak.hip -o tools/build/hbm_peak && tools/build/hbm_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void __launch_bounds__(256) read_kernel(const float4* __restrict__ src, size_t n4, float* __restrict__ out) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {   // four independent loads in flight per thread
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y;
        acc.z += a.z + b.z + c.z + d.z; acc.w += a.w + b.w + c.w + d.w;
    }
    for (; i < n4; i += stride) { const float4 a = src[i]; acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 12345.678f) out[blockIdx.x] = s;   // never true for the zero-filled source: keeps the loads alive
}

__global__ void __launch_bounds__(256) copy_kernel(const float4* __restrict__ src, size_t n4, float4* __restrict__ dst) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

// the same copy with 4-byte accesses (one dword per lane, 256 contiguous bytes per wave instruction)
__global__ void __launch_bounds__(256) copy1_kernel(const float* __restrict__ src, size_t n, float* __restrict__ dst) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const float a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) read1_kernel(const float* __restrict__ src, size_t n, float* __restrict__ out) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) acc += src[i] + src[i + stride] + src[i + 2 * stride] + src[i + 3 * stride];
    for (; i < n; i += stride) acc += src[i];
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const size_t pool = (size_t)3 << 30;   // rotate inside 3 GiB
    char* buf = nullptr;
    float* out = nullptr;
    CK(hipMalloc(&buf, pool));
    CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(buf, 0, pool));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t sizes[] = {(size_t)32 * 8732 * 81 * 4, (size_t)64 * 8732 * 81 * 4, (size_t)1 << 30};   // 1 GiB: large-size reference
    const int grids[] = {2048, 8192, 32768};
    for (size_t bytes : sizes) {
        const size_t n4 = bytes / 16, slot = (bytes + 4095) / 4096 * 4096;
        for (int mode = 0; mode < 4; ++mode) {
            const size_t span = (mode & 1) ? 2 * slot : slot;      // copy: source + destination
            const int slots = (int)(pool / span);
            for (int grid : grids) {
                float best = 1e30f, sum = 0.f;
                const int reps = 20;
                for (int r = 0; r < reps + 3; ++r) {   // 8 back-to-back launches per timing: what a kernel inside a step sees
                    CK(hipEventRecord(e0, 0));
                    for (int k = 0; k < 8; ++k) {
                        char* base = buf + (size_t)((r * 8 + k) % slots) * span;
                        if (mode == 2) hipLaunchKernelGGL(read1_kernel, dim3(grid), dim3(256), 0, 0, (const float*)base, n4 * 4, out);
                        else if (mode == 3) hipLaunchKernelGGL(copy1_kernel, dim3(grid), dim3(256), 0, 0, (const float*)base, n4 * 4, (float*)(base + slot));
                        else if (mode == 0) hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, (const float4*)base, n4, out);
                        else hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, (const float4*)base, n4, (float4*)(base + slot));
                    }
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms = 0.f;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    ms /= 8.f;
                    if (r >= 3) { sum += ms; if (ms < best) best = ms; }
                }
                const double moved = ((mode & 1) ? 2.0 : 1.0) * (double)bytes;
                printf("%s %8.1f MB grid %5d : avg %7.1f us = %5.2f TB/s (%.2f of 8), best %7.1f us = %5.2f TB/s\n", mode == 0 ? "read  16B" : mode == 1 ? "copy  16B" : mode == 2 ? "read   4B" : "copy   4B",
                       bytes / 1e6, grid, sum / reps * 1e3, moved / (sum / reps * 1e-3) / 1e12, moved / (sum / reps * 1e-3) / 8e12, best * 1e3,
                       moved / (best * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}

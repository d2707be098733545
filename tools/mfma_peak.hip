// Micro-benchmark: what the fp32 matrix pipe of an MI355X delivers with v_mfma_f32_32x32x2_f32 (the instruction the head
// GEMMs issue), at one and two waves per SIMD, bare and with LDS fragment reads between the MFMAs.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/build/mfma_peak && tools/build/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef __attribute__((address_space(3))) void* lds_ptr_t;
// 0: bare MFMAs from registers; 1: operands re-read from LDS every 16 MFMAs (1 A + 4 B ds_read_b128);
// 2: mode 1 + one workgroup barrier per 64 MFMAs; 3: mode 2 + 8 LDS-DMA pieces (8 KB per wave) per 64 MFMAs from an
// L2-resident buffer; 4: mode 3 without the barrier
template <int MODE>
__global__ void __launch_bounds__(256, 2) mfma_loop(float* out, int iters, const float* src = nullptr, unsigned src_bytes = 0) {
    __shared__ __attribute__((aligned(16))) float lds[5 * 32 * 36 * 2];
    __shared__ __attribute__((aligned(1024))) float dma_dst[4 * 8 * 256];
    for (int i = threadIdx.x; i < 5 * 32 * 36 * 2; i += 256) lds[i] = (float)(i % 7) * 0.001f;
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const int lane = threadIdx.x & 63;
    const float* rd = lds + (lane & 31) * 36 + 4 * (lane >> 5);
    f32x4 a = *(const f32x4*)rd, b[4];
    for (int j = 0; j < 4; ++j) b[j] = *(const f32x4*)(rd + (j + 1) * 32 * 36);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)src_bytes, 0x00020000);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned voff = ((blockIdx.x * 4 + wave) * 8192u + lane * 16u) % (src_bytes ? src_bytes - 65536u : 1u);
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 3 && (it & 3) == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dma_dst + (wave * 8 + i) * 256), 16, voff + i * 1024u, 0, 0, 0);
            voff = (voff + 4 * 8192u * 512u) % (src_bytes - 65536u);
        }
        if (MODE >= 1) {
            a = *(const f32x4*)(rd + (it & 3) * 8);
            for (int j = 0; j < 4; ++j) b[j] = *(const f32x4*)(rd + (j + 1) * 32 * 36 + (it & 3) * 8);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[j][kk], acc[j], 0, 0, 0);
        if ((MODE == 2 || MODE == 3) && (it & 3) == 3) __syncthreads();
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int blocks, int iters, float* out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    static float* src = nullptr;
    const unsigned src_bytes = 64u << 20;
    if (!src) { (void)hipMalloc(&src, src_bytes); (void)hipMemset(src, 0, src_bytes); }
    hipLaunchKernelGGL(mfma_loop<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, src, src_bytes);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, src, src_bytes);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 /*waves*/ * iters * 16 * 4096.0;
    printf("%-44s blocks %4d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
}

int main() {
    float* out;
    (void)hipMalloc(&out, 4096 * 256 * sizeof(float));
    const int iters = 20000;
    run<0>("bare MFMA, 1 wave/SIMD", 256, iters, out);
    run<0>("bare MFMA, 2 waves/SIMD", 512, iters, out);
    run<1>("MFMA + 5 ds_read_b128 per 16, 1 wave/SIMD", 256, iters, out);
    run<1>("MFMA + 5 ds_read_b128 per 16, 2 waves/SIMD", 512, iters, out);
    run<2>("... + barrier per 64 MFMA, 2 waves/SIMD", 512, iters, out);
    run<3>("... + barrier + 8 DMA pieces per 64, 2 waves/SIMD", 512, iters, out);
    run<4>("... + 8 DMA pieces per 64, no barrier, 2 waves/SIMD", 512, iters, out);
    run<3>("... + barrier + 8 DMA pieces per 64, 1 wave/SIMD", 256, iters, out);
    return 0;
}

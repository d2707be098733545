// Experiment: semantics of buffer_load_dwordx4 ... lds on gfx950 (lane -> LDS placement, out-of-range lanes).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const float* p, int nbytes, float* out) {
    __shared__ __attribute__((aligned(16))) float s[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) s[i] = -7.0f;
    __syncthreads();
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)p), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)p >> 32));
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, nbytes, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // lane l of wave w fetches 16 B at element (w*64 + (63 - l)) * 4  (reversed, to see the lane -> LDS mapping); odd lanes of wave 1 are out of range
    unsigned voff = (unsigned)(wave * 64 + (63 - lane)) * 16u;
    if (wave == 1 && (lane & 1)) voff = 0x80000000u;
    if (wave == 2 && (lane & 1)) voff = 0xFFFFFFF0u;
    if (wave != 3 || lane < 32)   // wave 3: upper half of the lanes inactive (exec = 0)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(s + wave * 256), 16, voff, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) lgkmcnt(0)
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = s[i];
}
int main() {
    float *p, *out, h[1024], src[1024];
    for (int i = 0; i < 1024; ++i) src[i] = (float)i;
    (void)hipMalloc(&p, 4096); (void)hipMalloc(&out, 4096);
    (void)hipMemcpy(p, src, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, p, 4096, out);
    (void)hipMemcpy(h, out, 4096, hipMemcpyDeviceToHost);
    for (int w = 0; w < 4; ++w) {
        printf("wave %d LDS floats [0..15]:", w);
        for (int i = 0; i < 16; ++i) printf(" %g", h[w * 256 + i]);
        printf("  ... [248..255]:");
        for (int i = 248; i < 256; ++i) printf(" %g", h[w * 256 + i]);
        printf("\n");
    }
    return 0;
}

// Establishes the register layout of v_mfma_f32_16x16x1_4b_f32 (four 16 x 16 x 1 blocks) by experiment, for the 16-column remainder tile of
// conv.hip's dma_tile: A one-hot in lane t, B = lane + 1; where does the product land?
//   hipcc --offload-arch=gfx950 tools/mfma_16x16x1_layout.hip -o /tmp/mfma_layout && /tmp/mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(int t, float* out) {
    const int l = threadIdx.x;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(l == t ? 1.0f : 0.0f, (float)(l + 1), acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) out[e * 64 + l] = acc[e];
}

int main() {
    float* d;
    hipMalloc(&d, 16 * 64 * sizeof(float));
    float h[16 * 64];
    int bad = 0;
    for (int t = 0; t < 64; ++t) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, t, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int v = 0; v < 16; ++v)
            for (int l = 0; l < 64; ++l) {
                // hypothesis: VGPR v of lane l holds D[block v / 4][row 4 * (l / 16) + v % 4][column l % 16]; A lane t is block t / 16, row t % 16; B lane is block / column alike
                const bool hit = v / 4 == t / 16 && 4 * (l / 16) + v % 4 == t % 16;
                const float want = hit ? (float)((t / 16) * 16 + l % 16 + 1) : 0.0f;
                if (h[v * 64 + l] != want) {
                    if (bad < 20) printf("t=%d v=%d lane=%d got %g want %g\n", t, v, l, h[v * 64 + l], want);
                    ++bad;
                }
            }
    }
    printf(bad ? "LAYOUT MISMATCH (%d)\n" : "layout as assumed: D[v][l] = block v/4, row 4*(l/16) + v%%4, column l%%16 (%d mismatches)\n", bad);
    return bad != 0;
}

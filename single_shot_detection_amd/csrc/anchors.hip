// anchors.hip -- anchor generation.
//
// Reference: detection/anchor_generators/ssd.py:12-151, retina_net.py:18-54, _anchor_generator.py:7-20.
// The reference builds anchors on the CPU with torch ops every step (lru-cached per level) and re-uploads the
// concatenated [A,4] tensor each step (detection/init.py:117).  Here the per-level (w,h) table is a handful of
// host scalars (same fp32/double promotion sequence as the reference's mixed tensor/python arithmetic) and the
// [H,W,nb,4] grid is written once, directly in HBM, by one small kernel; the Python layer caches the result.
#include <math.h>

#include "common.h"

namespace ssdk {

constexpr int kMaxBoxesPerCell = 32;
struct HwTable {
    float w[kMaxBoxesPerCell];
    float h[kMaxBoxesPerCell];
};

// torch.linspace fp32 element (aten RangeFactoriesKernel.cpp): the shipped kernels contract the multiply-add,
// so this is an explicit fmaf (the TU itself is compiled with -ffp-contract=off).
__host__ __device__ inline float linspace_at(float start, float end, float step, int n, int i) {
    if (n == 1) return start;
    return i < n / 2 ? fmaf(step, (float)i, start) : fmaf(-step, (float)(n - i - 1), end);
}

__global__ void __launch_bounds__(256) anchors_level_kernel(float4* __restrict__ out, int H, int W, int nb, HwTable hw,
                                                            float xs, float xe, float xstep, float ys, float ye, float ystep) {
    const int total = H * W * nb;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int k = t % nb, cell = t / nb;
        const int x = cell % W, y = cell / W;
        out[t] = make_float4(linspace_at(xs, xe, xstep, W, x), linspace_at(ys, ye, ystep, H, y), hw.w[k], hw.h[k]);
    }
}

}  // namespace ssdk

using namespace ssdk;

extern "C" int ssdk_linspace_f32(float start, float end, int steps, float* out) {
    SSDK_REQUIRE(steps >= 1 && out, SSDK_E_INVALID, "ssdk_linspace_f32: steps=%d", steps);
    const float step = steps > 1 ? (end - start) / (float)(steps - 1) : 0.0f;
    for (int i = 0; i < steps; ++i) out[i] = linspace_at(start, end, step, steps, i);
    return SSDK_OK;
}

extern "C" int ssdk_anchor_sizes_ssd(const double* ratios, int nratio, float min_scale, float max_scale, int img_w,
                                     int img_h, float* hws, int hws_cap) {
    SSDK_REQUIRE(ratios && hws && nratio > 0, SSDK_E_INVALID, "ssdk_anchor_sizes_ssd: null/empty ratios");
    int nb = 0;
    // ssd.py:125 sizes = scales * img  (fp32 tensor * python int)
    const float min_w = min_scale * (float)img_w, min_h = min_scale * (float)img_h;
    const float max_w = max_scale * (float)img_w, max_h = max_scale * (float)img_h;
    for (int k = 0; k < nratio; ++k) {
        SSDK_REQUIRE(ratios[k] >= 1.0, SSDK_E_INVALID, "ssdk_anchor_sizes_ssd: aspect ratio %g < 1 with flip (ssd.py:88)", ratios[k]);
        const int reps = ratios[k] > 1.0 ? 2 : 1;  // ssd.py:87-92 flip
        for (int r = 0; r < reps; ++r) {
            SSDK_REQUIRE(nb + 1 < hws_cap, SSDK_E_INVALID, "ssdk_anchor_sizes_ssd: hws_cap too small");
            const double ar = r == 0 ? ratios[k] : 1.0 / ratios[k];
            const float sr = (float)sqrt(ar);  // python float scalar joins an fp32 tensor op as fp32
            hws[2 * nb + 0] = min_w * sr;      // ssd.py:132
            hws[2 * nb + 1] = min_h / sr;      // ssd.py:133
            ++nb;
        }
    }
    hws[2 * nb + 0] = (float)sqrt((double)(min_w * max_w));  // ssd.py:135 math.sqrt of an fp32 product
    hws[2 * nb + 1] = (float)sqrt((double)(min_h * max_h));  // ssd.py:136
    return nb + 1;
}

extern "C" int ssdk_anchor_sizes_retina(const double* ratios, int nratio, int level, double scale, int scales_per_level,
                                        float* hws, int hws_cap) {
    SSDK_REQUIRE(ratios && hws && nratio > 0 && scales_per_level > 0, SSDK_E_INVALID, "ssdk_anchor_sizes_retina: bad args");
    SSDK_REQUIRE(nratio * scales_per_level <= hws_cap, SSDK_E_INVALID, "ssdk_anchor_sizes_retina: hws_cap too small");
    for (int j = 0; j < scales_per_level; ++j) {
        const double size = scale * pow(2.0, (double)level + (double)j / scales_per_level);  // retina_net.py:26
        for (int k = 0; k < nratio; ++k) {
            hws[2 * (j * nratio + k) + 0] = (float)(size * sqrt(ratios[k]));  // :42
            hws[2 * (j * nratio + k) + 1] = (float)(size / sqrt(ratios[k]));  // :43
        }
    }
    return nratio * scales_per_level;
}

extern "C" int ssdk_anchors_level_ex(float* out, int layer_h, int layer_w, int nb, const float* hws_host, double step_w, double step_h,
                                     double offset_x, double offset_y, void* stream) {
    SSDK_REQUIRE(out && hws_host, SSDK_E_INVALID, "ssdk_anchors_level: null pointer");
    SSDK_REQUIRE(layer_h > 0 && layer_w > 0 && nb > 0 && nb <= kMaxBoxesPerCell, SSDK_E_INVALID,
                 "ssdk_anchors_level: bad shape H=%d W=%d nb=%d (nb <= %d)", layer_h, layer_w, nb, kMaxBoxesPerCell);
    HwTable hw;
    for (int k = 0; k < nb; ++k) { hw.w[k] = hws_host[2 * k]; hw.h[k] = hws_host[2 * k + 1]; }
    // ssd.py:138-139: python-float products become the fp32 start / end of torch.linspace
    const float xs = (float)(offset_x * step_w), xe = (float)((offset_x + layer_w - 1) * step_w);
    const float ys = (float)(offset_y * step_h), ye = (float)((offset_y + layer_h - 1) * step_h);
    const float xstep = layer_w > 1 ? (xe - xs) / (float)(layer_w - 1) : 0.0f;
    const float ystep = layer_h > 1 ? (ye - ys) / (float)(layer_h - 1) : 0.0f;
    const int total = layer_h * layer_w * nb;
    const int blocks = cdiv(total, 256) < 1024 ? cdiv(total, 256) : 1024;
    hipLaunchKernelGGL(anchors_level_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float4*)out, layer_h, layer_w,
                       nb, hw, xs, xe, xstep, ys, ye, ystep);
    SSDK_CHECK_LAUNCH("anchors_level_kernel");
    return SSDK_OK;
}

extern "C" int ssdk_anchors_level(float* out, int layer_h, int layer_w, int nb, const float* hws_host, int img_w,
                                  int img_h, void* stream) {
    SSDK_REQUIRE(layer_h > 0 && layer_w > 0, SSDK_E_INVALID, "ssdk_anchors_level: bad shape H=%d W=%d", layer_h, layer_w);
    // ssd.py:117-118 step = img / n (python division), :138-139 offset .5
    return ssdk_anchors_level_ex(out, layer_h, layer_w, nb, hws_host, (double)img_w / layer_w, (double)img_h / layer_h, 0.5, 0.5, stream);
}

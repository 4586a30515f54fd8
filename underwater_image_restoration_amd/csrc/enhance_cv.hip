// The RL adapter's OpenCV-backed operators and scores on the device (remainder):
//   histogram_equalization, clahe_equalization          rl/uiep_cwgr_ops.py:24-44
//   compute_uism, compute_uiconm, compute_uciqe          rl/uiqm_uciqe.py:24-65
// PARITY UNPINNED: the reference computes these through cv2 (split / merge, cvtColor RGB <-> Lab on 8-bit images,
// createCLAHE, Sobel), and OpenCV is not installed in the build container, so no golden vector of the reference could be
// generated.  What is restated here is OpenCV's documented arithmetic: the float CIE L*a*b* formulas of cvtColor (sRGB
// gamma, D65 white point, 8-bit scaling L * 255 / 100, a + 128, b + 128, round to nearest) -- OpenCV's own 8-bit path goes
// through interpolated fixed-point tables and may differ from these by one level in some pixels --, CLAHE as in
// modules/imgproc/src/clahe.cpp (integer clip limit, excess redistributed in a batch plus a strided residual, LUT =
// round(cumulative * 255 / tileArea), bilinear blend of the four neighbouring tile LUTs, BORDER_REFLECT_101 padding to a
// multiple of the grid), Sobel 3x3 with BORDER_REFLECT_101.  histogram_equalization is numpy arithmetic in the reference
// (cv2 only splits and merges the channels) and is reproduced operation for operation.
// The CPU restatement the tests compare against is oracle/enhance_ref.py (same formulas in numpy, written independently).
#include "common.h"

typedef unsigned char u8;

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// (img * 255).astype(np.uint8) of the clamped image (uiep_cwgr_ops.py:14-16, uiqm_uciqe.py:8-10): fp32 product, truncation
__device__ __forceinline__ u8 quant_u8(float v) {
    v = fminf(fmaxf(v, 0.f), 1.f);   // NaN -> 0 (clamp(0, 1) of NaN stays NaN in torch; astype(uint8) of NaN is unspecified)
    return (u8)(int)(v * 255.f);
}

// ---- quantise (+ per-channel histograms)
__global__ void __launch_bounds__(256) cv_quant_hist_kernel(const float* __restrict__ img, int C, size_t hw, u8* __restrict__ q,
                                                            unsigned* __restrict__ hist) {
    __shared__ unsigned sh[256];
    const int c = blockIdx.y;
    sh[threadIdx.x] = 0;
    __syncthreads();
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < hw; i += (size_t)gridDim.x * 256) {
        const u8 v = quant_u8(img[c * hw + i]);
        q[c * hw + i] = v;
        if (hist) atomicAdd(&sh[v], 1u);
    }
    __syncthreads();
    if (hist && sh[threadIdx.x]) atomicAdd(&hist[c * 256 + threadIdx.x], sh[threadIdx.x]);
}

// histogram_equalization's LUT: cdf = hist.cumsum(); (cdf - cdf.min()) * 255 / (cdf.max() - cdf.min() + 1e-6) in float64;
// np.interp at the integer sample points returns the table value; astype(np.uint8) truncates
__global__ void __launch_bounds__(256) cv_histeq_lut_kernel(const unsigned* __restrict__ hist, u8* __restrict__ lut) {
    __shared__ long long cdf[256];
    const int c = blockIdx.x;
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int i = 0; i < 256; ++i) { run += hist[c * 256 + i]; cdf[i] = run; }
    }
    __syncthreads();
    const long long lo = cdf[0], hi = cdf[255];          // a cumulative sum is non-decreasing
    const double v = (double)(cdf[threadIdx.x] - lo) * 255.0 / ((double)(hi - lo) + 1e-6);
    lut[c * 256 + threadIdx.x] = (u8)(int)v;
}

__global__ void __launch_bounds__(256) cv_lut_apply_kernel(const u8* __restrict__ q, const u8* __restrict__ lut, int C, size_t hw,
                                                           float* __restrict__ out) {
    const int c = blockIdx.y;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < hw; i += (size_t)gridDim.x * 256)
        out[c * hw + i] = (float)lut[c * 256 + q[c * hw + i]] / 255.0f;
}

// ---- 8-bit RGB <-> Lab (cvtColor COLOR_RGB2LAB / COLOR_LAB2RGB, float formulas, D65)
__device__ __forceinline__ float srgb_to_lin(float c) { return c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f); }
__device__ __forceinline__ float lin_to_srgb(float c) { return c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.f / 2.4f) - 0.055f; }
__device__ __forceinline__ float lab_f(float t) { return t > 0.008856f ? cbrtf(t) : 7.787f * t + 16.f / 116.f; }
__device__ __forceinline__ u8 sat_round_u8(float v) {
    const float r = rintf(v);   // cvRound: to nearest, ties to even
    return (u8)(r < 0.f ? 0.f : r > 255.f ? 255.f : r);
}
__device__ __forceinline__ void rgb8_to_lab8(u8 r8, u8 g8, u8 b8, u8& L8, u8& a8, u8& bb8) {
    const float R = srgb_to_lin(r8 / 255.f), G = srgb_to_lin(g8 / 255.f), B = srgb_to_lin(b8 / 255.f);
    const float X = (0.412453f * R + 0.357580f * G + 0.180423f * B) / 0.950456f;
    const float Y = 0.212671f * R + 0.715160f * G + 0.072169f * B;
    const float Z = (0.019334f * R + 0.119193f * G + 0.950227f * B) / 1.088754f;
    const float fx = lab_f(X), fy = lab_f(Y), fz = lab_f(Z);
    const float L = Y > 0.008856f ? 116.f * fy - 16.f : 903.3f * Y;
    L8 = sat_round_u8(L * 255.f / 100.f);
    a8 = sat_round_u8(500.f * (fx - fy) + 128.f);
    bb8 = sat_round_u8(200.f * (fy - fz) + 128.f);
}
__device__ __forceinline__ void lab8_to_rgb8(u8 L8, u8 a8, u8 b8, u8& r8, u8& g8, u8& bb8) {
    const float L = L8 * 100.f / 255.f, a = (float)a8 - 128.f, b = (float)b8 - 128.f;
    float fy, Y;
    if (L <= 7.9996248f) { Y = L / 903.3f; fy = 7.787f * Y + 16.f / 116.f; }
    else { fy = (L + 16.f) / 116.f; Y = fy * fy * fy; }
    const float fx = a / 500.f + fy, fz = fy - b / 200.f;
    const float X = (fx > 0.20689303f ? fx * fx * fx : (fx - 16.f / 116.f) / 7.787f) * 0.950456f;
    const float Z = (fz > 0.20689303f ? fz * fz * fz : (fz - 16.f / 116.f) / 7.787f) * 1.088754f;
    const float R = 3.240479f * X - 1.53715f * Y - 0.498535f * Z;
    const float G = -0.969256f * X + 1.875991f * Y + 0.041556f * Z;
    const float B = 0.055648f * X - 0.204043f * Y + 1.057311f * Z;
    r8 = sat_round_u8(lin_to_srgb(fminf(fmaxf(R, 0.f), 1.f)) * 255.f);
    g8 = sat_round_u8(lin_to_srgb(fminf(fmaxf(G, 0.f), 1.f)) * 255.f);
    bb8 = sat_round_u8(lin_to_srgb(fminf(fmaxf(B, 0.f), 1.f)) * 255.f);
}

__global__ void __launch_bounds__(256) cv_rgb2lab_kernel(const u8* __restrict__ q, size_t hw, u8* __restrict__ lab) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < hw; i += (size_t)gridDim.x * 256)
        rgb8_to_lab8(q[i], q[hw + i], q[2 * hw + i], lab[i], lab[hw + i], lab[2 * hw + i]);
}
// Lab (with the CLAHE'd L plane) -> RGB float image
__global__ void __launch_bounds__(256) cv_lab2rgb_kernel(const u8* __restrict__ lab, const u8* __restrict__ Lnew, size_t hw,
                                                         float* __restrict__ out) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < hw; i += (size_t)gridDim.x * 256) {
        u8 r, g, b;
        lab8_to_rgb8(Lnew[i], lab[hw + i], lab[2 * hw + i], r, g, b);
        out[i] = (float)r / 255.0f;
        out[hw + i] = (float)g / 255.0f;
        out[2 * hw + i] = (float)b / 255.0f;
    }
}

// ---- CLAHE (clahe.cpp): one block per tile builds the tile's clipped histogram LUT
__global__ void __launch_bounds__(256) cv_clahe_lut_kernel(const u8* __restrict__ L, int H, int W, int th, int tw, int tiles_x,
                                                           int clip, u8* __restrict__ luts) {
    __shared__ int hist[256];
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < th * tw; i += 256) {
        const int y = reflect101(ty * th + i / tw, H), x = reflect101(tx * tw + i % tw, W);   // the padded image mirrors (101) the source
        atomicAdd(&hist[L[(size_t)y * W + x]], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (clip > 0) {
            int clipped = 0;
            for (int i = 0; i < 256; ++i)
                if (hist[i] > clip) { clipped += hist[i] - clip; hist[i] = clip; }
            const int batch = clipped / 256;
            int residual = clipped - batch * 256;
            for (int i = 0; i < 256; ++i) hist[i] += batch;
            if (residual != 0) {
                const int step = max(256 / residual, 1);
                for (int i = 0; i < 256 && residual > 0; i += step, --residual) hist[i]++;
            }
        }
        const float scale = 255.f / (float)(th * tw);
        int sum = 0;
        for (int i = 0; i < 256; ++i) {
            sum += hist[i];
            luts[(size_t)blockIdx.x * 256 + i] = sat_round_u8((float)sum * scale);
        }
    }
}
__global__ void __launch_bounds__(256) cv_clahe_interp_kernel(const u8* __restrict__ L, int H, int W, int th, int tw, int tiles_y,
                                                              int tiles_x, const u8* __restrict__ luts, u8* __restrict__ out) {
    const float inv_th = 1.0f / (float)th, inv_tw = 1.0f / (float)tw;
    const size_t hw = (size_t)H * W;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < hw; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
        const float tyf = y * inv_th - 0.5f, txf = x * inv_tw - 0.5f;
        int ty1 = (int)floorf(tyf), tx1 = (int)floorf(txf);
        int ty2 = ty1 + 1, tx2 = tx1 + 1;
        const float ya = tyf - ty1, xa = txf - tx1;
        ty1 = max(ty1, 0); tx1 = max(tx1, 0);
        ty2 = min(ty2, tiles_y - 1); tx2 = min(tx2, tiles_x - 1);
        const int v = L[i];
        const float p11 = luts[((size_t)ty1 * tiles_x + tx1) * 256 + v], p12 = luts[((size_t)ty1 * tiles_x + tx2) * 256 + v];
        const float p21 = luts[((size_t)ty2 * tiles_x + tx1) * 256 + v], p22 = luts[((size_t)ty2 * tiles_x + tx2) * 256 + v];
        const float res = (p11 * (1.f - xa) + p12 * xa) * (1.f - ya) + (p21 * (1.f - xa) + p22 * xa) * ya;
        out[i] = sat_round_u8(res);
    }
}

// ---- scores: per-block partial sums in double, combined in block order by one thread (deterministic)
#define CV_PBLK 256
__device__ __forceinline__ double block_sum(double v, double* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}
// partials[blk][0..2] = sum over the block's pixels of sqrt(gx^2 + gy^2) per channel (cv2.Sobel ksize 3 on the float32 channel)
__global__ void __launch_bounds__(256) cv_sobel_partial_kernel(const u8* __restrict__ q, int H, int W, double* __restrict__ part) {
    __shared__ double red[256];
    const size_t hw = (size_t)H * W;
    double s[3] = {0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < hw; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
        const int ym = reflect101(y - 1, H), yp = reflect101(y + 1, H), xm = reflect101(x - 1, W), xp = reflect101(x + 1, W);
        for (int c = 0; c < 3; ++c) {
            const u8* p = q + c * hw;
            const float a = p[(size_t)ym * W + xm], b = p[(size_t)ym * W + x], cc = p[(size_t)ym * W + xp];
            const float d = p[(size_t)y * W + xm], f = p[(size_t)y * W + xp];
            const float g = p[(size_t)yp * W + xm], h = p[(size_t)yp * W + x], k = p[(size_t)yp * W + xp];
            const float gx = (cc + 2.f * f + k) - (a + 2.f * d + g), gy = (g + 2.f * h + k) - (a + 2.f * b + cc);
            s[c] += (double)sqrtf(gx * gx + gy * gy);
        }
    }
    for (int c = 0; c < 3; ++c) {
        const double r = block_sum(s[c], red);
        if (threadIdx.x == 0) part[blockIdx.x * 3 + c] = r;
    }
}
// partials[blk][0..6]: sum L, sum L^2, sum chroma, sum chroma^2, sum chroma / (L + 1e-6), -min L, max L (on the 8-bit Lab values
// as float32, a and b WITH their +128 offset: uiqm_uciqe.py:53-61 uses them as cvtColor returns them)
__global__ void __launch_bounds__(256) cv_lab_partial_kernel(const u8* __restrict__ lab, size_t hw, double* __restrict__ part) {
    __shared__ double red[256];
    double s[5] = {0, 0, 0, 0, 0};
    float mn = 1e30f, mx = -1e30f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < hw; i += (size_t)gridDim.x * 256) {
        const float L = lab[i], a = lab[hw + i], b = lab[2 * hw + i];
        const float ch = sqrtf(a * a + b * b);
        s[0] += L; s[1] += (double)L * L; s[2] += ch; s[3] += (double)ch * ch; s[4] += (double)(ch / (L + 1e-6f));
        mn = fminf(mn, L); mx = fmaxf(mx, L);
    }
    for (int k = 0; k < 5; ++k) {
        const double r = block_sum(s[k], red);
        if (threadIdx.x == 0) part[blockIdx.x * 7 + k] = r;
    }
    red[threadIdx.x] = -mn;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x * 7 + 5] = red[0];
    __syncthreads();
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x * 7 + 6] = red[0];
}
// out3 = (uism, uiconm, uciqe)
__global__ void cv_scores_finalize_kernel(const double* __restrict__ sob, const double* __restrict__ labp, int nblk, double npix,
                                          double* __restrict__ out3) {
    double g[3] = {0, 0, 0}, s[5] = {0, 0, 0, 0, 0}, nmn = -1e300, mx = -1e300;
    for (int b = 0; b < nblk; ++b) {
        for (int c = 0; c < 3; ++c) g[c] += sob[b * 3 + c];
        for (int k = 0; k < 5; ++k) s[k] += labp[b * 7 + k];
        nmn = fmax(nmn, labp[b * 7 + 5]);
        mx = fmax(mx, labp[b * 7 + 6]);
    }
    out3[0] = (g[0] / npix + g[1] / npix + g[2] / npix) / 3.0;                       // compute_uism: mean of the per-channel means
    const float max_l = (float)mx, min_l = (float)(-nmn);
    out3[1] = (double)((max_l - min_l) / (max_l + min_l + 1e-6f));                   // compute_uiconm (float32 arithmetic)
    const double mL = s[0] / npix, vL = fmax(s[1] / npix - mL * mL, 0.0);
    const double mC = s[2] / npix, vC = fmax(s[3] / npix - mC * mC, 0.0);
    out3[2] = 0.4680 * sqrt(vC) + 0.2745 * sqrt(vL) + 0.2576 * (s[4] / npix);         // compute_uciqe
}

static inline int cv_grid(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g > 1024 ? 1024 : (g < 1 ? 1 : g));
}
static inline size_t cv_al(size_t n) { return (n + 255) & ~(size_t)255; }

extern "C" size_t uwir_cv_workspace_bytes(int H, int W, int tiles_y, int tiles_x) {
    if (H <= 0 || W <= 0) return 0;
    const size_t hw = (size_t)H * W;
    const size_t t = (size_t)(tiles_y > 0 ? tiles_y : 8) * (tiles_x > 0 ? tiles_x : 8);
    // q (3 planes), lab (3 planes), L' (1 plane), histograms / LUTs (3 x 256), tile LUTs, partial sums
    return cv_al(3 * hw) + cv_al(3 * hw) + cv_al(hw) + cv_al(3 * 256 * sizeof(unsigned)) + cv_al(3 * 256) + cv_al(t * 256) +
           cv_al((size_t)CV_PBLK * 10 * sizeof(double));
}

struct CvWs {
    u8 *q, *lab, *L2, *lut, *tluts;
    unsigned* hist;
    double* part;
};
static CvWs cv_carve(void* ws, int H, int W, int tiles_y, int tiles_x) {
    const size_t hw = (size_t)H * W;
    char* p = (char*)ws;
    CvWs c;
    c.q = (u8*)p; p += cv_al(3 * hw);
    c.lab = (u8*)p; p += cv_al(3 * hw);
    c.L2 = (u8*)p; p += cv_al(hw);
    c.hist = (unsigned*)p; p += cv_al(3 * 256 * sizeof(unsigned));
    c.lut = (u8*)p; p += cv_al(3 * 256);
    c.tluts = (u8*)p; p += cv_al((size_t)tiles_y * tiles_x * 256);
    c.part = (double*)p;
    return c;
}

extern "C" int uwir_histogram_equalization(const float* img, int C, int H, int W, float* out, void* workspace, size_t workspace_bytes,
                                           uwir_stream_t stream) {
    UWIR_CHECK(img && out && workspace, "histogram_equalization: null pointer");
    UWIR_CHECK(C >= 1 && C <= 3 && H > 0 && W > 0, "histogram_equalization: bad shape C=%d H=%d W=%d (1 to 3 channels)", C, H, W);
    UWIR_CHECK(((uintptr_t)workspace & 255) == 0 && workspace_bytes >= uwir_cv_workspace_bytes(H, W, 8, 8), "histogram_equalization: workspace");
    hipStream_t st = (hipStream_t)stream;
    const size_t hw = (size_t)H * W;
    CvWs w = cv_carve(workspace, H, W, 8, 8);
    hipError_t e = hipMemsetAsync(w.hist, 0, 3 * 256 * sizeof(unsigned), st);
    if (e != hipSuccess) return uwir_set_error(UWIR_ELAUNCH, "histogram_equalization: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(cv_quant_hist_kernel, dim3(cv_grid(hw), C), dim3(256), 0, st, img, C, hw, w.q, w.hist);
    hipLaunchKernelGGL(cv_histeq_lut_kernel, dim3(C), dim3(256), 0, st, w.hist, w.lut);
    hipLaunchKernelGGL(cv_lut_apply_kernel, dim3(cv_grid(hw), C), dim3(256), 0, st, w.q, w.lut, C, hw, out);
    UWIR_LAUNCH_CHECK("histogram_equalization");
    return UWIR_OK;
}

extern "C" int uwir_clahe_equalization(const float* img, int H, int W, float clip_limit, int tiles_y, int tiles_x, float* out,
                                       void* workspace, size_t workspace_bytes, uwir_stream_t stream) {
    UWIR_CHECK(img && out && workspace, "clahe_equalization: null pointer");
    UWIR_CHECK(H > 0 && W > 0 && tiles_y > 0 && tiles_x > 0 && tiles_y <= 64 && tiles_x <= 64, "clahe_equalization: bad shape / grid");
    UWIR_CHECK(((uintptr_t)workspace & 255) == 0 && workspace_bytes >= uwir_cv_workspace_bytes(H, W, tiles_y, tiles_x), "clahe_equalization: workspace");
    hipStream_t st = (hipStream_t)stream;
    const size_t hw = (size_t)H * W;
    CvWs w = cv_carve(workspace, H, W, tiles_y, tiles_x);
    // tiles of the image padded (reflect 101) up to a multiple of the grid (clahe.cpp: copyMakeBorder when not divisible)
    const int th = (H + tiles_y - 1) / tiles_y, tw = (W + tiles_x - 1) / tiles_x;
    int clip = 0;
    if (clip_limit > 0.0f) {
        clip = (int)(clip_limit * (float)(th * tw) / 256.f);
        clip = clip < 1 ? 1 : clip;
    }
    hipLaunchKernelGGL(cv_quant_hist_kernel, dim3(cv_grid(hw), 3), dim3(256), 0, st, img, 3, hw, w.q, (unsigned*)nullptr);
    hipLaunchKernelGGL(cv_rgb2lab_kernel, dim3(cv_grid(hw)), dim3(256), 0, st, w.q, hw, w.lab);
    hipLaunchKernelGGL(cv_clahe_lut_kernel, dim3(tiles_y * tiles_x), dim3(256), 0, st, w.lab, H, W, th, tw, tiles_x, clip, w.tluts);
    hipLaunchKernelGGL(cv_clahe_interp_kernel, dim3(cv_grid(hw)), dim3(256), 0, st, w.lab, H, W, th, tw, tiles_y, tiles_x, w.tluts, w.L2);
    hipLaunchKernelGGL(cv_lab2rgb_kernel, dim3(cv_grid(hw)), dim3(256), 0, st, w.lab, w.L2, hw, out);
    UWIR_LAUNCH_CHECK("clahe_equalization");
    return UWIR_OK;
}

extern "C" int uwir_cv_scores(const float* img, int H, int W, double* out3, void* workspace, size_t workspace_bytes,
                              uwir_stream_t stream) {
    UWIR_CHECK(img && out3 && workspace, "cv_scores: null pointer");
    UWIR_CHECK(H > 0 && W > 0, "cv_scores: bad shape");
    UWIR_CHECK(((uintptr_t)workspace & 255) == 0 && workspace_bytes >= uwir_cv_workspace_bytes(H, W, 8, 8), "cv_scores: workspace");
    hipStream_t st = (hipStream_t)stream;
    const size_t hw = (size_t)H * W;
    CvWs w = cv_carve(workspace, H, W, 8, 8);
    int nblk = cv_grid(hw);
    if (nblk > CV_PBLK) nblk = CV_PBLK;
    hipLaunchKernelGGL(cv_quant_hist_kernel, dim3(cv_grid(hw), 3), dim3(256), 0, st, img, 3, hw, w.q, (unsigned*)nullptr);
    hipLaunchKernelGGL(cv_rgb2lab_kernel, dim3(cv_grid(hw)), dim3(256), 0, st, w.q, hw, w.lab);
    hipLaunchKernelGGL(cv_sobel_partial_kernel, dim3(nblk), dim3(256), 0, st, w.q, H, W, w.part);
    hipLaunchKernelGGL(cv_lab_partial_kernel, dim3(nblk), dim3(256), 0, st, w.lab, hw, w.part + (size_t)CV_PBLK * 3);
    hipLaunchKernelGGL(cv_scores_finalize_kernel, dim3(1), dim3(1), 0, st, w.part, w.part + (size_t)CV_PBLK * 3, nblk, (double)hw, out3);
    UWIR_LAUNCH_CHECK("cv_scores");
    return UWIR_OK;
}

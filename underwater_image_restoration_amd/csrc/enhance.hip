// RL enhancement operators and state features that the reference computes in torch / numpy:
// rl/uiep_cwgr_ops.py gamma_correction, lf_hf_mix, wavelet_denoise, guided_filter, multi_scale_retinex;
// rl/uiqm_uciqe.py compute_uicm; rl/infer_adapt.py compute_sharpness and the histogram mean / std of build_state.
// Images are planar CHW fp32, as the reference passes them.  Everything here is HBM- or latency-bound byte work on one
// image: no MFMA.  Reductions are deterministic: integer atomics only (exact), floating-point partial sums are written
// per block and added in a fixed order.  The OpenCV-backed operators of the same files (CLAHE, 8-bit Lab, cv2.Sobel)
// are not built: see DESIGN.md section 7.
#include "common.h"

static inline int grid_for(size_t n, int block = 256, int cap = 4096) {
    size_t g = ceil_div_sz(n, block);
    return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g));
}

// ------------------------------------------------------------------------------------------
// pointwise
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gamma_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n, float gamma) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = powf(fminf(fmaxf(x[i], 0.f), 1.f), gamma);
}

extern "C" int uwir_gamma_correction(const float* x, float* out, size_t n, float gamma, uwir_stream_t stream) {
    UWIR_CHECK(x && out, "gamma_correction: null pointer");
    if (n == 0) return UWIR_OK;
    hipLaunchKernelGGL(gamma_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, out, n, fmaxf(gamma, 1e-3f));
    UWIR_LAUNCH_CHECK("gamma_correction");
    return UWIR_OK;
}

// two rounded products and a rounded sum, as the reference's tensor expression evaluates (no contraction into an FMA)
__global__ void __launch_bounds__(256) mix_kernel(const float* __restrict__ lo, const float* __restrict__ hi, float* __restrict__ out,
                                                  size_t n, float a_hi, float a_lo) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = fminf(fmaxf(__fadd_rn(__fmul_rn(a_hi, hi[i]), __fmul_rn(a_lo, lo[i])), 0.f), 1.f);
}

extern "C" int uwir_lf_hf_mix(const float* low, const float* high, double alpha, float* out, size_t n, uwir_stream_t stream) {
    UWIR_CHECK(low && high && out, "lf_hf_mix: null pointer");
    if (n == 0) return UWIR_OK;
    // python: alpha * high + (1 - alpha) * low with a double alpha; each scalar reaches the fp32 tensor op rounded once
    hipLaunchKernelGGL(mix_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, low, high, out, n, (float)alpha,
                       (float)(1.0 - alpha));
    UWIR_LAUNCH_CHECK("lf_hf_mix");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// wavelet_denoise: Haar level -> per band t = scale * (median|c| / 0.6745)^2 / (sqrt(mean c^2) + 1e-6) -> soft
// threshold -> inverse Haar.  The median (torch.median: the LOWER middle element) is an exact radix select over the
// bit patterns of |c| (non-negative floats order like their bits): four 8-bit passes, most significant byte first,
// each a histogram launch (LDS bins -> integer atomics) and a one-block pick.
// workspace: [4][n] floats (ll, lh, hl, hh) | WdState | [3][WD_PBLK] doubles (sum c^2 partials)
// ------------------------------------------------------------------------------------------
#define WD_PBLK 256
struct WdState {
    unsigned hist[3][4][256];
    unsigned prefix[3], k[3];
    float thr[3];
};
static __device__ __forceinline__ float haar_k() {
    const float s = 0.70710678118654752440f;
    return s * s;   // fl32(2^-0.5)^2, S/dwt_fft.py:10-21
}

__global__ void __launch_bounds__(256) wd_dwt_kernel(const float* __restrict__ x, int C, int H, int W, int h, int w,
                                                     float* __restrict__ bands, double* __restrict__ part, WdState* st) {
    const size_t n = (size_t)C * h * w;
    if (blockIdx.x == 0 && threadIdx.x < 3) st->k[threadIdx.x] = (unsigned)((n - 1) / 2);   // rank of the lower median
    const float k = haar_k();
    double s[3] = {0.0, 0.0, 0.0};
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int ox = (int)(i % w), oy = (int)((i / w) % h), c = (int)(i / ((size_t)w * h));
        const float* r0 = x + ((size_t)c * H + 2 * oy) * W + 2 * ox;
        const float a = r0[0], b = r0[1], cc = r0[W], d = r0[W + 1];
        const float ll = k * a + k * b + k * cc + k * d;
        const float lh = k * a - k * b + k * cc - k * d;
        const float hl = k * a + k * b - k * cc - k * d;
        const float hh = k * a - k * b - k * cc + k * d;
        bands[i] = ll; bands[n + i] = lh; bands[2 * n + i] = hl; bands[3 * n + i] = hh;
        s[0] += (double)(lh * lh); s[1] += (double)(hl * hl); s[2] += (double)(hh * hh);   // c**2 is rounded to fp32 first
    }
    __shared__ double red[3][256];
    for (int b = 0; b < 3; ++b) red[b][threadIdx.x] = s[b];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off)
            for (int b = 0; b < 3; ++b) red[b][threadIdx.x] += red[b][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x < 3) part[threadIdx.x * WD_PBLK + blockIdx.x] = red[threadIdx.x][0];
}

__global__ void __launch_bounds__(256) wd_hist_kernel(const float* __restrict__ bands, size_t n, WdState* st, int pass) {
    const int band = blockIdx.y, shift = 24 - 8 * pass;
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned prefix = st->prefix[band];
    const unsigned mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    const float* c = bands + (size_t)(band + 1) * n;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned u = __float_as_uint(fabsf(c[i]));
        if ((u & mask) == prefix) atomicAdd(&h[(u >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&st->hist[band][pass][threadIdx.x], h[threadIdx.x]);
}

// one block per band: the bin that holds the element of rank k; after the last pass also the threshold
__global__ void __launch_bounds__(256) wd_pick_kernel(WdState* st, int pass, size_t n, float scale, const double* __restrict__ part,
                                                      int nblk) {
    const int band = blockIdx.x, t = threadIdx.x, shift = 24 - 8 * pass;
    __shared__ unsigned cum[256];
    __shared__ double red[256];
    cum[t] = st->hist[band][pass][t];
    __syncthreads();
    if (t == 0) {
        unsigned k = st->k[band], run = 0;
        int bin = 255;
        for (int b = 0; b < 256; ++b) {
            if (run + cum[b] > k) { bin = b; break; }
            run += cum[b];
        }
        st->k[band] = k - run;
        st->prefix[band] |= (unsigned)bin << shift;
    }
    if (pass != 3) return;
    red[t] = t < nblk ? part[band * WD_PBLK + t] : 0.0;   // nblk <= WD_PBLK = 256: a fixed-shape tree, fixed order
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) red[t] += red[t + off];
        __syncthreads();
    }
    if (t == 0) {
        const float med = __uint_as_float(st->prefix[band]);
        const float sigma = med / 0.6745f;
        const float var = (float)(red[0] / (double)n);
        st->thr[band] = scale * (sigma * sigma / (sqrtf(var) + 1e-6f));
    }
}

__global__ void __launch_bounds__(256) wd_idwt_kernel(const float* __restrict__ bands, const WdState* __restrict__ st, int C,
                                                      int h, int w, float* __restrict__ out) {
    const size_t n = (size_t)C * h * w;
    const float k = haar_k();
    const float t0 = st->thr[0], t1 = st->thr[1], t2 = st->thr[2];
    auto soft = [](float v, float thr) {
        const float m = fmaxf(fabsf(v) - thr, 0.f);   // sign(v) * relu(|v| - t)
        return v > 0.f ? m : (v < 0.f ? -m : 0.f * m);
    };
    const int W2 = 2 * w;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int ox = (int)(i % w), oy = (int)((i / w) % h), c = (int)(i / ((size_t)w * h));
        const float ll = bands[i], lh = soft(bands[n + i], t0), hl = soft(bands[2 * n + i], t1), hh = soft(bands[3 * n + i], t2);
        float* r0 = out + ((size_t)c * 2 * h + 2 * oy) * W2 + 2 * ox;
        r0[0] = k * ll + k * lh + k * hl + k * hh;
        r0[1] = k * ll - k * lh + k * hl - k * hh;
        r0[W2] = k * ll + k * lh - k * hl - k * hh;
        r0[W2 + 1] = k * ll - k * lh - k * hl + k * hh;
    }
}

static size_t wd_align(size_t v) { return (v + 255) & ~(size_t)255; }
extern "C" size_t uwir_wavelet_denoise_workspace_bytes(int C, int H, int W) {
    if (C <= 0 || H < 2 || W < 2) return 0;
    const size_t n = (size_t)C * (H / 2) * (W / 2);
    return wd_align(4 * n * sizeof(float)) + wd_align(sizeof(WdState)) + 3 * WD_PBLK * sizeof(double);
}

extern "C" int uwir_wavelet_denoise(const float* x, int C, int H, int W, float threshold_scale, float* out, void* workspace,
                                    size_t workspace_bytes, uwir_stream_t stream) {
    UWIR_CHECK(x && out && workspace, "wavelet_denoise: null pointer");
    UWIR_CHECK(C > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "wavelet_denoise: H and W must be even (got %d x %d)", H, W);
    UWIR_CHECK(((uintptr_t)workspace & 255) == 0 && workspace_bytes >= uwir_wavelet_denoise_workspace_bytes(C, H, W),
               "wavelet_denoise: workspace too small / misaligned");
    const int h = H / 2, w = W / 2;
    const size_t n = (size_t)C * h * w;
    UWIR_CHECK(n < 0xffffffffULL, "wavelet_denoise: image too large");
    hipStream_t s = (hipStream_t)stream;
    float* bands = (float*)workspace;
    WdState* st = (WdState*)((char*)workspace + wd_align(4 * n * sizeof(float)));
    double* part = (double*)((char*)st + wd_align(sizeof(WdState)));
    if (hipMemsetAsync(st, 0, sizeof(WdState), s) != hipSuccess) return uwir_set_error(UWIR_ELAUNCH, "wavelet_denoise: memset failed");
    const int nblk = grid_for(n, 256, WD_PBLK);
    hipLaunchKernelGGL(wd_dwt_kernel, dim3(nblk), dim3(256), 0, s, x, C, H, W, h, w, bands, part, st);
    for (int pass = 0; pass < 4; ++pass) {
        hipLaunchKernelGGL(wd_hist_kernel, dim3(grid_for(n, 256, 1024), 3), dim3(256), 0, s, bands, n, st, pass);
        hipLaunchKernelGGL(wd_pick_kernel, dim3(3), dim3(256), 0, s, st, pass, n, threshold_scale, part, nblk);
    }
    hipLaunchKernelGGL(wd_idwt_kernel, dim3(grid_for(n)), dim3(256), 0, s, bands, st, C, h, w, out);
    UWIR_LAUNCH_CHECK("wavelet_denoise");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// guided_filter (one channel): zero-padded (2r+1)^2 box means.  Two tiled kernels: (I, P) -> (a, b), then
// (a, b, I) -> q.  16 x 16 outputs per block, the (16 + 2r)^2 inputs in LDS.
// ------------------------------------------------------------------------------------------
#define GF_T 16
#define GF_RMAX 16
template <bool SECOND>
__global__ void __launch_bounds__(256) gf_kernel(const float* __restrict__ in0, const float* __restrict__ in1,
                                                 const float* __restrict__ guide, int H, int W, int r, float eps,
                                                 float* __restrict__ out0, float* __restrict__ out1) {
    extern __shared__ float sm[];
    const int L = GF_T + 2 * r;
    float* s0 = sm;
    float* s1 = sm + L * L;
    const int x0 = blockIdx.x * GF_T, y0 = blockIdx.y * GF_T;
    for (int i = threadIdx.x; i < L * L; i += 256) {
        const int ly = i / L, lx = i % L, y = y0 + ly - r, x = x0 + lx - r;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        s0[i] = in ? in0[(size_t)y * W + x] : 0.f;
        s1[i] = in ? in1[(size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    const int tx = threadIdx.x % GF_T, ty = threadIdx.x / GF_T;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= W || y >= H) return;
    const float inv = 1.0f / (float)((2 * r + 1) * (2 * r + 1));
    if (!SECOND) {   // in0 = I (guidance), in1 = P
        float si = 0.f, sp = 0.f, sip = 0.f, sii = 0.f;
        for (int dy = 0; dy <= 2 * r; ++dy)
            for (int dx = 0; dx <= 2 * r; ++dx) {
                const float i = s0[(ty + dy) * L + tx + dx], p = s1[(ty + dy) * L + tx + dx];
                si += i * inv; sp += p * inv; sip += (i * p) * inv; sii += (i * i) * inv;   // the tap weight is 1 / k^2
            }
        const float cov = sip - si * sp, var = sii - si * si;
        const float a = cov / (var + eps);
        out0[(size_t)y * W + x] = a;
        out1[(size_t)y * W + x] = sp - a * si;
    } else {         // in0 = a, in1 = b
        float sa = 0.f, sb = 0.f;
        for (int dy = 0; dy <= 2 * r; ++dy)
            for (int dx = 0; dx <= 2 * r; ++dx) {
                sa += s0[(ty + dy) * L + tx + dx] * inv;
                sb += s1[(ty + dy) * L + tx + dx] * inv;
            }
        out0[(size_t)y * W + x] = sa * guide[(size_t)y * W + x] + sb;
    }
}

extern "C" size_t uwir_guided_filter_workspace_bytes(int H, int W) {
    return H > 0 && W > 0 ? 2 * (size_t)H * W * sizeof(float) : 0;
}

extern "C" int uwir_guided_filter(const float* input, const float* guidance, int H, int W, int radius, float eps, float* out,
                                  void* workspace, size_t workspace_bytes, uwir_stream_t stream) {
    UWIR_CHECK(input && guidance && out && workspace, "guided_filter: null pointer");
    UWIR_CHECK(H > 0 && W > 0 && radius >= 0 && radius <= GF_RMAX, "guided_filter: bad shape / radius (0..%d)", GF_RMAX);
    UWIR_CHECK(workspace_bytes >= uwir_guided_filter_workspace_bytes(H, W) && ((uintptr_t)workspace & 3) == 0,
               "guided_filter: workspace too small");
    float* a = (float*)workspace;
    float* b = a + (size_t)H * W;
    const int L = GF_T + 2 * radius;
    const size_t lds = 2 * (size_t)L * L * sizeof(float);
    dim3 grid(ceil_div(W, GF_T), ceil_div(H, GF_T));
    hipLaunchKernelGGL(gf_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, guidance, input, guidance, H, W, radius, eps, a, b);
    hipLaunchKernelGGL(gf_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, (const float*)a, (const float*)b, guidance, H, W,
                       radius, eps, out, (float*)nullptr);
    UWIR_LAUNCH_CHECK("guided_filter");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// multi_scale_retinex: per sigma a zero-padded Gaussian of size 2 ceil(3 sigma) + 1, normalised in 2-D -- which
// factorises exactly into two normalised 1-D kernels: a row pass (row + weight table in LDS) and a column pass
// (8 outputs per thread, coalesced global reads) that also accumulates log x - log(blur + 1e-6) and, on the last
// sigma, finishes exp(mean) - 1 clamped to [0, 1].  Only the taps that fall inside the image are visited, so the
// 1501-tap kernel of sigma = 250 costs W + H taps per pixel, not 1501^2.
// workspace: [C][H][W] row-pass result | [C][H][W] accumulator
// ------------------------------------------------------------------------------------------
#define MSR_TY 8
__global__ void __launch_bounds__(256) msr_rows_kernel(const float* __restrict__ x, int H, int W, int r, float neg_inv_2s2,
                                                       float inv_norm, float* __restrict__ tmp) {
    extern __shared__ float sm[];
    float* row = sm;                 // [W]
    float* wt = sm + W;              // [min(r, W - 1) + 1]
    const int re = min(r, W - 1);
    const float* src = x + (size_t)blockIdx.x * W;   // blockIdx.x = c * H + y
    for (int i = threadIdx.x; i < W; i += 256) row[i] = fmaxf(src[i], 1e-3f);
    for (int d = threadIdx.x; d <= re; d += 256) wt[d] = expf((float)(d * d) * neg_inv_2s2) * inv_norm;
    __syncthreads();
    for (int xo = threadIdx.x; xo < W; xo += 256) {
        const int lo = max(0, xo - re), hi = min(W - 1, xo + re);
        float s = 0.f;
        for (int xi = lo; xi <= hi; ++xi) s = fmaf(row[xi], wt[abs(xo - xi)], s);
        tmp[(size_t)blockIdx.x * W + xo] = s;
    }
}

__global__ void __launch_bounds__(256) msr_cols_kernel(const float* __restrict__ x, const float* __restrict__ tmp, int H, int W,
                                                       int r, float neg_inv_2s2, float inv_norm, float* __restrict__ acc,
                                                       float* __restrict__ out, int first, int last, float nsig) {
    extern __shared__ float wt[];    // [min(r, H - 1) + 1]
    const int re = min(r, H - 1);
    for (int d = threadIdx.x; d <= re; d += 256) wt[d] = expf((float)(d * d) * neg_inv_2s2) * inv_norm;
    __syncthreads();
    const int xo = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * MSR_TY, c = blockIdx.z;
    if (xo >= W) return;
    const float* plane = tmp + (size_t)c * H * W;
    float s[MSR_TY];
#pragma unroll
    for (int t = 0; t < MSR_TY; ++t) s[t] = 0.f;
    const int lo = max(0, y0 - re), hi = min(H - 1, y0 + MSR_TY - 1 + re);
    for (int yi = lo; yi <= hi; ++yi) {
        const float v = plane[(size_t)yi * W + xo];
#pragma unroll
        for (int t = 0; t < MSR_TY; ++t) {
            const int d = abs(y0 + t - yi);
            if (d <= re) s[t] = fmaf(v, wt[d], s[t]);
        }
    }
#pragma unroll
    for (int t = 0; t < MSR_TY; ++t) {
        const int y = y0 + t;
        if (y >= H) break;
        const size_t o = ((size_t)c * H + y) * W + xo;
        const float term = logf(fmaxf(x[o], 1e-3f)) - logf(s[t] + 1e-6f);
        const float a = first ? (0.f + term) : acc[o] + term;
        if (last) out[o] = fminf(fmaxf(expf(a / nsig) - 1.0f, 0.f), 1.f);
        else acc[o] = a;
    }
}

extern "C" size_t uwir_multi_scale_retinex_workspace_bytes(int C, int H, int W) {
    return C > 0 && H > 0 && W > 0 ? 2 * (size_t)C * H * W * sizeof(float) : 0;
}

extern "C" int uwir_multi_scale_retinex(const float* x, int C, int H, int W, const double* sigmas, int nsig, float* out,
                                        void* workspace, size_t workspace_bytes, uwir_stream_t stream) {
    UWIR_CHECK(x && out && workspace && sigmas, "multi_scale_retinex: null pointer");
    UWIR_CHECK(C > 0 && H > 0 && W > 0 && nsig > 0 && (long long)C * H < 0x7fffffffLL && C <= 65535,
               "multi_scale_retinex: bad shape");
    UWIR_CHECK(workspace_bytes >= uwir_multi_scale_retinex_workspace_bytes(C, H, W) && ((uintptr_t)workspace & 3) == 0,
               "multi_scale_retinex: workspace too small");
    UWIR_CHECK((size_t)(2 * W) * sizeof(float) <= 60 * 1024 && (size_t)H * sizeof(float) <= 60 * 1024,
               "multi_scale_retinex: image too large for the LDS row (%d x %d)", H, W);
    float* tmp = (float*)workspace;
    float* acc = tmp + (size_t)C * H * W;
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < nsig; ++i) {
        const double sg = sigmas[i];
        UWIR_CHECK(sg > 0.0, "multi_scale_retinex: sigma must be positive");
        const int size = (int)(2 * ceil(3 * sg) + 1), r = size / 2;
        double norm = 0.0;
        for (int u = -r; u <= r; ++u) norm += exp(-(double)u * u / (2 * sg * sg));
        const float n2 = (float)(-1.0 / (2 * sg * sg)), inv = (float)(1.0 / norm);
        const size_t lds_r = ((size_t)W + (size_t)(r < W - 1 ? r : W - 1) + 1) * sizeof(float);
        const size_t lds_c = ((size_t)(r < H - 1 ? r : H - 1) + 1) * sizeof(float);
        hipLaunchKernelGGL(msr_rows_kernel, dim3(C * H), dim3(256), lds_r, s, x, H, W, r, n2, inv, tmp);
        hipLaunchKernelGGL(msr_cols_kernel, dim3(ceil_div(W, 256), ceil_div(H, MSR_TY), C), dim3(256), lds_c, s, x, (const float*)tmp,
                           H, W, r, n2, inv, acc, out, i == 0, i == nsig - 1, (float)nsig);
    }
    UWIR_LAUNCH_CHECK("multi_scale_retinex");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// state features of build_state (rl/infer_adapt.py:62-68) that need no OpenCV:
//   out[0] UICM (rl/uiqm_uciqe.py:13-21, on the uint8-quantised image: exact integer sums; C must be 3, else NaN)
//   out[1], out[2] mean and unbiased std over every element;  out[3] mean Sobel magnitude of the channel mean
// workspace: ST_PBLK x 3 doubles of block partials | 4 int64 (sum rg, sum rg^2, sum 2yb, sum (2yb)^2)
// ------------------------------------------------------------------------------------------
#define ST_PBLK 256
__global__ void __launch_bounds__(256) state_partial_kernel(const float* __restrict__ img, int C, int H, int W,
                                                            double* __restrict__ part, long long* __restrict__ isum) {
    const size_t npix = (size_t)H * W;
    double sx = 0.0, sxx = 0.0, sg = 0.0;
    long long irg = 0, irg2 = 0, iyb = 0, iyb2 = 0;
    auto gray = [&](int y, int x) -> float {
        if (y < 0 || y >= H || x < 0 || x >= W) return 0.f;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += img[((size_t)c * H + y) * W + x];
        return s / (float)C;
    };
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i % W);
        int q[3] = {0, 0, 0};
        for (int c = 0; c < C; ++c) {
            const float v = img[(size_t)c * npix + i];
            sx += (double)v;
            sxx += (double)v * (double)v;
            if (c < 3) q[c] = (int)__fmul_rn(fminf(fmaxf(v, 0.f), 1.f), 255.f);   // (clamp * 255).astype(uint8): truncation
        }
        const int rg = q[0] - q[1], yb2 = q[0] + q[1] - 2 * q[2];
        irg += rg; irg2 += rg * rg; iyb += yb2; iyb2 += yb2 * yb2;
        const float g00 = gray(y - 1, x - 1), g01 = gray(y - 1, x), g02 = gray(y - 1, x + 1);
        const float g10 = gray(y, x - 1), g12 = gray(y, x + 1);
        const float g20 = gray(y + 1, x - 1), g21 = gray(y + 1, x), g22 = gray(y + 1, x + 1);
        const float gx = (g02 - g00) + 2.f * (g12 - g10) + (g22 - g20);
        const float gy = (g20 - g00) + 2.f * (g21 - g01) + (g22 - g02);
        sg += (double)sqrtf(gx * gx + gy * gy);
    }
    __shared__ double red[3][256];
    __shared__ long long ired[4][256];
    red[0][threadIdx.x] = sx; red[1][threadIdx.x] = sxx; red[2][threadIdx.x] = sg;
    ired[0][threadIdx.x] = irg; ired[1][threadIdx.x] = irg2; ired[2][threadIdx.x] = iyb; ired[3][threadIdx.x] = iyb2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
            for (int k = 0; k < 4; ++k) ired[k][threadIdx.x] += ired[k][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x < 3) part[threadIdx.x * ST_PBLK + blockIdx.x] = red[threadIdx.x][0];
    if (threadIdx.x < 4) atomicAdd((unsigned long long*)&isum[threadIdx.x], (unsigned long long)ired[threadIdx.x][0]);
}

__global__ void __launch_bounds__(256) state_final_kernel(const double* __restrict__ part, const long long* __restrict__ isum,
                                                          int nblk, int C, int H, int W, double* __restrict__ out) {
    __shared__ double red[3][256];
    const int t = threadIdx.x;
    for (int k = 0; k < 3; ++k) red[k][t] = t < nblk ? part[k * ST_PBLK + t] : 0.0;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off)
            for (int k = 0; k < 3; ++k) red[k][t] += red[k][t + off];
        __syncthreads();
    }
    if (t != 0) return;
    const double npix = (double)H * W, n = npix * C;
    const double mean = red[0][0] / n;
    out[1] = mean;
    out[2] = n > 1 ? sqrt(fmax(red[1][0] - n * mean * mean, 0.0) / (n - 1)) : nan("");
    out[3] = red[2][0] / npix;
    if (C == 3) {
        const double mrg = (double)isum[0] / npix, myb = 0.5 * (double)isum[2] / npix;
        const double srg = sqrt(fmax((double)isum[1] / npix - mrg * mrg, 0.0));
        const double syb = sqrt(fmax(0.25 * (double)isum[3] / npix - myb * myb, 0.0));
        out[0] = (-0.0268 * mrg + 0.1586 * srg) - (0.0504 * myb - 0.164 * syb);
    } else {
        out[0] = nan("");
    }
}

extern "C" size_t uwir_enhance_state_workspace_bytes(void) { return 3 * ST_PBLK * sizeof(double) + 4 * sizeof(long long); }

extern "C" int uwir_enhance_state(const float* img, int C, int H, int W, double* out4, void* workspace, size_t workspace_bytes,
                                  uwir_stream_t stream) {
    UWIR_CHECK(img && out4 && workspace, "enhance_state: null pointer");
    UWIR_CHECK(C > 0 && H > 0 && W > 0, "enhance_state: bad shape");
    UWIR_CHECK(workspace_bytes >= uwir_enhance_state_workspace_bytes() && ((uintptr_t)workspace & 7) == 0,
               "enhance_state: workspace too small / misaligned");
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)workspace;
    long long* isum = (long long*)(part + 3 * ST_PBLK);
    if (hipMemsetAsync(isum, 0, 4 * sizeof(long long), s) != hipSuccess) return uwir_set_error(UWIR_ELAUNCH, "enhance_state: memset failed");
    const int nblk = grid_for((size_t)H * W, 256, ST_PBLK);
    hipLaunchKernelGGL(state_partial_kernel, dim3(nblk), dim3(256), 0, s, img, C, H, W, part, isum);
    hipLaunchKernelGGL(state_final_kernel, dim3(1), dim3(256), 0, s, (const double*)part, (const long long*)isum, nblk, C, H, W, out4);
    UWIR_LAUNCH_CHECK("enhance_state");
    return UWIR_OK;
}

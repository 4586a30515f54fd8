// Token-row kernels for the transformer parts of the path (DA-CLIP ViT towers, IR-SDE UNet attention):
// per-row LayerNorm over channels and the ViT patch gather.  HBM-bound, one wave per row, float4 lanes.
#include "common.h"

// ------------------------------------------------------------------------------------------
// LayerNorm over the C channels of each row (F.layer_norm semantics: biased variance, eps inside
// the sqrt), optionally of (x + add[row % period]) -- the ViT's "+ positional/class embedding"
// (L/open_clip/transformer.py:527-536) rides along.  beta may be NULL (gain-only LayerNorm of the
// IR-SDE UNet, U/models/modules/module_util.py:77-86).  One wave per row; the row is read three
// times but only the first read leaves the cache hierarchy.
// ------------------------------------------------------------------------------------------
// G = lanes that share one row (64, 32, 16 or 8): narrow rows (C = 64 -> 16 float4) would leave most of a wave idle,
// so a wave handles 64 / G rows and the reductions stay inside the G-lane group.
template <typename TX, typename TY, int G>
__global__ void __launch_bounds__(256) ln_rows_kernel(const TX* __restrict__ x, size_t ldx, int M, int C4,
                                                      const float* __restrict__ add, int add_period, int add_ld,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, const float* res, size_t ldr, TY* y, size_t ldy) {
    constexpr int RPW = UWIR_WAVE / G;  // rows per wave
    const int lane = threadIdx.x % UWIR_WAVE, sub = lane % G;
    const int wave = blockIdx.x * (blockDim.x / UWIR_WAVE) + threadIdx.x / UWIR_WAVE;
    const int row_raw = wave * RPW + lane / G;
    const bool live = row_raw < M;
    const int row = live ? row_raw : M - 1;  // dead groups shadow the last row (no stores) so shuffles stay convergent
    const TX* xr = x + (size_t)row * ldx;
    const float* ar = add ? add + (size_t)(row % add_period) * add_ld : nullptr;
    const float invC = 1.f / (float)(4 * C4);
    auto ld = [&](int c4) {
        float4 v = load4(xr + 4 * c4);
        if (ar) {
            float4 a = load4(ar + 4 * c4);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        return v;
    };
    float s = 0.f;
    for (int c4 = sub; c4 < C4; c4 += G) {
        float4 v = ld(c4);
        s += (v.x + v.y) + (v.z + v.w);
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s * invC;
    float q = 0.f;
    for (int c4 = sub; c4 < C4; c4 += G) {
        float4 v = ld(c4);
        float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) q += __shfl_xor(q, off);
    if (!live) return;
    const float rstd = 1.f / sqrtf(q * invC + eps);
    TY* yr = y + (size_t)row * ldy;
    const float* rr = res ? res + (size_t)row * ldr : nullptr;
    for (int c4 = sub; c4 < C4; c4 += G) {
        float4 v = ld(c4), g = load4(gamma + 4 * c4);
        float4 b = beta ? load4(beta + 4 * c4) : make_float4(0, 0, 0, 0);
        float4 o;
        o.x = (v.x - mean) * rstd * g.x + b.x;
        o.y = (v.y - mean) * rstd * g.y + b.y;
        o.z = (v.z - mean) * rstd * g.z + b.z;
        o.w = (v.w - mean) * rstd * g.w + b.w;
        if (rr) {  // Residual(PreNorm(...)) tail: LN(out) + x (U/models/modules/module_util.py:27-33,167,185)
            float4 r = load4(rr + 4 * c4);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        store4(yr + 4 * c4, o);
    }
}

template <typename TX, typename TY>
static void launch_ln_rows(const TX* x, size_t ldx, int M, int C, const float* add, int add_period, int add_ld,
                           const float* gamma, const float* beta, float eps, const float* res, size_t ldr, TY* y,
                           size_t ldy, hipStream_t s) {
    const int C4 = C / 4;
#define LN_LAUNCH(G)                                                                                                   \
    hipLaunchKernelGGL((ln_rows_kernel<TX, TY, G>), dim3(ceil_div(M, 4 * (UWIR_WAVE / G))), dim3(256), 0, s, x, ldx, M, C4, \
                       add, add_period, add_ld, gamma, beta, eps, res, ldr, y, ldy)
    if (C4 <= 8) LN_LAUNCH(8);
    else if (C4 <= 16) LN_LAUNCH(16);
    else if (C4 <= 32) LN_LAUNCH(32);
    else LN_LAUNCH(64);
#undef LN_LAUNCH
}

extern "C" int uwir_ln_rows(const void* x, int x_dtype, size_t ldx, int M, int C, const float* add, int add_period, int add_ld,
                            const float* gamma, const float* beta, float eps, const float* res, size_t ldr, void* y,
                            int y_dtype, size_t ldy, uwir_stream_t stream) {
    UWIR_CHECK(x && gamma && y, "ln_rows: null pointer");
    UWIR_CHECK(M > 0 && C > 0 && C % 4 == 0 && ldx >= (size_t)C && ldx % 4 == 0 && ldy >= (size_t)C && ldy % 4 == 0,
               "ln_rows: bad shape M=%d C=%d", M, C);
    UWIR_CHECK(!add || (add_period > 0 && add_ld >= C && add_ld % 4 == 0), "ln_rows: bad add table");
    UWIR_CHECK(!res || (ldr >= (size_t)C && ldr % 4 == 0), "ln_rows: bad residual stride");
    UWIR_CHECK(y_dtype == UWIR_F32 || y_dtype == UWIR_BF16, "ln_rows: bad output dtype");
    UWIR_CHECK(x_dtype == UWIR_F32 || x_dtype == UWIR_BF16, "ln_rows: bad input dtype");
    UWIR_CHECK((const void*)x != (const void*)y || (y_dtype == UWIR_F32 && x_dtype == UWIR_F32), "ln_rows: in-place needs fp32 rows");
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == UWIR_F32 && y_dtype == UWIR_F32)
        launch_ln_rows<float, float>((const float*)x, ldx, M, C, add, add_period, add_ld, gamma, beta, eps, res, ldr, (float*)y, ldy, st);
    else if (x_dtype == UWIR_F32)
        launch_ln_rows<float, bf16_t>((const float*)x, ldx, M, C, add, add_period, add_ld, gamma, beta, eps, res, ldr, (bf16_t*)y, ldy, st);
    else if (y_dtype == UWIR_F32)
        launch_ln_rows<bf16_t, float>((const bf16_t*)x, ldx, M, C, add, add_period, add_ld, gamma, beta, eps, res, ldr, (float*)y, ldy, st);
    else
        launch_ln_rows<bf16_t, bf16_t>((const bf16_t*)x, ldx, M, C, add, add_period, add_ld, gamma, beta, eps, res, ldr, (bf16_t*)y, ldy, st);
    UWIR_LAUNCH_CHECK("ln_rows");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// ViT patch gather (the im2col of Conv2d(3, width, k=P, stride=P), L/open_clip/transformer.py:409,522):
// image NCHW fp32 -> rows [B*(G+1), Cin*P*P] with k = (c*P + ky)*P + kx, the layout of
// conv1.weight.reshape(width, -1).  Row 0 of every image is the (zero) class-token slot, so the patch
// GEMM output already has the (B, G+1, width) token layout.
// ------------------------------------------------------------------------------------------
template <typename TY>
__global__ void __launch_bounds__(256) patchify_kernel(const float* __restrict__ img, int Cin, int H, int W, int P,
                                                       int gh, int gw, TY* __restrict__ out, size_t ldo, size_t total) {
    const int P4 = P / 4, K4 = Cin * P * P4, L = gh * gw + 1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int k4 = (int)(i % K4);
        size_t row = i / K4;
        int l = (int)(row % L), b = (int)(row / L);
        int kx4 = k4 % P4, ky = (k4 / P4) % P, c = k4 / (P4 * P);
        float4 v = make_float4(0, 0, 0, 0);
        if (l > 0) {
            int py = (l - 1) / gw, px = (l - 1) % gw;
            v = load4(img + (((size_t)b * Cin + c) * H + py * P + ky) * W + px * P + 4 * kx4);
        }
        store4(out + row * ldo + ((size_t)c * P + ky) * P + 4 * kx4, v);
    }
}

extern "C" int uwir_patchify(const float* img_nchw, int B, int Cin, int H, int W, int P, void* out, int out_dtype,
                             size_t ldo, uwir_stream_t stream) {
    UWIR_CHECK(img_nchw && out, "patchify: null pointer");
    UWIR_CHECK(B > 0 && Cin > 0 && P > 0 && P % 4 == 0 && H >= P && W >= P && W % 4 == 0, "patchify: bad shape");
    UWIR_CHECK(((uintptr_t)img_nchw & 15) == 0, "patchify: image must be 16-byte aligned");
    UWIR_CHECK(ldo >= (size_t)Cin * P * P && ldo % 4 == 0, "patchify: ldo too small");
    UWIR_CHECK(out_dtype == UWIR_F32 || out_dtype == UWIR_BF16, "patchify: bad output dtype");
    int gh = H / P, gw = W / P;
    size_t total = (size_t)B * (gh * gw + 1) * Cin * P * (P / 4);
    size_t g = ceil_div_sz(total, 256);
    if (g > 8192) g = 8192;
    if (out_dtype == UWIR_F32)
        hipLaunchKernelGGL(patchify_kernel<float>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, img_nchw, Cin, H,
                           W, P, gh, gw, (float*)out, ldo, total);
    else
        hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, img_nchw, Cin,
                           H, W, P, gh, gw, (bf16_t*)out, ldo, total);
    UWIR_LAUNCH_CHECK("patchify");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// out[m,:] = a[m,:] + b[m,:]  (fp32 row matrices with independent strides; out may alias a or b).
// The explicit form of the main tower's `x += control.pop()` (L/open_clip/transformer.py:367-368) when the
// caller hands in materialised hiddens.
// ------------------------------------------------------------------------------------------
template <typename TB, typename TO>
__global__ void __launch_bounds__(256) add_rows_kernel(const float* a, size_t lda, const TB* b, size_t ldb, TO* out,
                                                       size_t ldo, int C4, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int c4 = (int)(i % C4);
        size_t m = i / C4;
        float4 u = load4(a + m * lda + 4 * c4), v = load4(b + m * ldb + 4 * c4);
        store4(out + m * ldo + 4 * c4, make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w));
    }
}

extern "C" int uwir_add_rows(const float* a, size_t lda, const void* b, int b_dtype, size_t ldb, void* out, int out_dtype,
                             size_t ldo, int M, int C, uwir_stream_t stream) {
    UWIR_CHECK(a && b && out, "add_rows: null pointer");
    UWIR_CHECK(M > 0 && C > 0 && C % 4 == 0 && lda >= (size_t)C && ldb >= (size_t)C && ldo >= (size_t)C &&
                   lda % 4 == 0 && ldb % 4 == 0 && ldo % 4 == 0,
               "add_rows: bad shape");
    size_t total = (size_t)M * (C / 4);
    size_t g = ceil_div_sz(total, 256);
    if (g > 8192) g = 8192;
    UWIR_CHECK((b_dtype == UWIR_F32 || b_dtype == UWIR_BF16) && (out_dtype == UWIR_F32 || out_dtype == UWIR_BF16), "add_rows: bad dtype");
    UWIR_CHECK((const void*)a != (const void*)out || out_dtype == UWIR_F32, "add_rows: out may alias a only as fp32");
#define ADD_LAUNCH(TB_, TO_)                                                                                            \
    hipLaunchKernelGGL((add_rows_kernel<TB_, TO_>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a, lda,       \
                       (const TB_*)b, ldb, (TO_*)out, ldo, C / 4, total)
    if (b_dtype == UWIR_F32 && out_dtype == UWIR_F32) ADD_LAUNCH(float, float);
    else if (b_dtype == UWIR_F32) ADD_LAUNCH(float, bf16_t);
    else if (out_dtype == UWIR_F32) ADD_LAUNCH(bf16_t, float);
    else ADD_LAUNCH(bf16_t, bf16_t);
#undef ADD_LAUNCH
    UWIR_LAUNCH_CHECK("add_rows");
    return UWIR_OK;
}

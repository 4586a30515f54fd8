// HBM-bound kernels: Haar DWT/IDWT, LayerNorm2d statistics -> folded affine, tiny linears,
// bilinear mask resize, stem/head 3x3 convs, fused IR-SDE steps.  All NHWC, fp32, float4 lanes.
#include "common.h"
#include <string.h>
#include <mutex>

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
int uwir_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char* uwir_last_error(void) { return g_err; }
extern "C" int uwir_version(void) { return 200; }

// ------------------------------------------------------------------------------------------
// Haar DWT / IDWT.  The reference tap is fl32(2^-0.5)^2 (S/dwt_fft.py:10-21), not 0.5.
// One thread = one output pixel x 4 channels: reads 4 float4, writes 4 float4.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float haar_tap() {
    const float s = 0.70710678118654752440f;
    return s * s;
}

template <typename TX, typename TS>   // feature rows in / sub-band rows out: fp32, or bf16 (the student's bf16 mode)
__global__ void __launch_bounds__(256) dwt2_kernel(const TX* __restrict__ x, int ldx, int H, int W, int C4, int h,
                                                   int w, TS* __restrict__ ll, int ld_ll, TS* __restrict__ hf,
                                                   int ld_hf, size_t total) {
    const float k = haar_tap();
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int c4 = (int)(i % C4);
        size_t p = i / C4;  // output pixel (b, y, x)
        int ox = (int)(p % w);
        size_t q = p / w;
        int oy = (int)(q % h);
        int b = (int)(q / h);
        const TX* r0 = x + (((size_t)b * H + 2 * oy) * W + 2 * ox) * ldx + 4 * c4;
        const TX* r1 = r0 + (size_t)W * ldx;
        float4 a = load4(r0), bb = load4(r0 + ldx), c = load4(r1), d = load4(r1 + ldx);
        float4 o_ll, o_lh, o_hl, o_hh;
#define HAAR(F)                                             \
    o_ll.F = k * a.F + k * bb.F + k * c.F + k * d.F;        \
    o_lh.F = k * a.F - k * bb.F + k * c.F - k * d.F;        \
    o_hl.F = k * a.F + k * bb.F - k * c.F - k * d.F;        \
    o_hh.F = k * a.F - k * bb.F - k * c.F + k * d.F;
        HAAR(x) HAAR(y) HAAR(z) HAAR(w)
#undef HAAR
        int C = 4 * C4;
        store4(ll + p * ld_ll + 4 * c4, o_ll);
        TS* hp = hf + p * ld_hf + 4 * c4;
        store4(hp, o_lh);
        store4(hp + C, o_hl);
        store4(hp + 2 * C, o_hh);
    }
}

template <typename TH, typename TO>   // high-frequency rows in / feature rows out: fp32 or bf16
__global__ void __launch_bounds__(256) idwt2_kernel(const float* __restrict__ ll, int ld_ll,
                                                    const TH* __restrict__ hf, int ld_hf, int h, int w, int C4,
                                                    TO* __restrict__ out, int ldo, size_t total) {
    const float k = haar_tap();
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int c4 = (int)(i % C4);
        size_t p = i / C4;
        int ox = (int)(p % w);
        size_t q = p / w;
        int oy = (int)(q % h);
        int b = (int)(q / h);
        int C = 4 * C4;
        float4 vll = load4(ll + p * ld_ll + 4 * c4);
        const TH* hp = hf + p * ld_hf + 4 * c4;
        float4 vlh = load4(hp), vhl = load4(hp + C), vhh = load4(hp + 2 * C);
        float4 o00, o01, o10, o11;
#define IHAAR(F)                                                        \
    o00.F = k * vll.F + k * vlh.F + k * vhl.F + k * vhh.F;              \
    o01.F = k * vll.F - k * vlh.F + k * vhl.F - k * vhh.F;              \
    o10.F = k * vll.F + k * vlh.F - k * vhl.F - k * vhh.F;              \
    o11.F = k * vll.F - k * vlh.F - k * vhl.F + k * vhh.F;
        IHAAR(x) IHAAR(y) IHAAR(z) IHAAR(w)
#undef IHAAR
        int W2 = 2 * w;
        TO* r0 = out + (((size_t)b * 2 * h + 2 * oy) * W2 + 2 * ox) * ldo + 4 * c4;
        TO* r1 = r0 + (size_t)W2 * ldo;
        store4(r0, o00);
        store4(r0 + ldo, o01);
        store4(r1, o10);
        store4(r1 + ldo, o11);
    }
}

static inline int ew_grid(size_t total, int block = 256) {
    size_t g = ceil_div_sz(total, block);
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int uwir_dwt2(const void* x, int x_dtype, int ldx, int B, int H, int W, int C, void* ll, int ld_ll, void* hf,
                         int ld_hf, int sub_dtype, uwir_stream_t stream) {
    UWIR_CHECK(x && ll && hf, "dwt2: null pointer");
    UWIR_CHECK((sub_dtype == UWIR_F32 || sub_dtype == UWIR_BF16) && (x_dtype == UWIR_F32 || x_dtype == UWIR_BF16), "dwt2: bad dtype");
    UWIR_CHECK(B > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0, "dwt2: bad shape B=%d H=%d W=%d C=%d", B, H, W, C);
    UWIR_CHECK(ldx >= C && ld_ll >= C && ld_hf >= 3 * C && ldx % 4 == 0 && ld_ll % 4 == 0 && ld_hf % 4 == 0,
               "dwt2: bad strides");
    int h = H / 2, w = W / 2;
    size_t total = (size_t)B * h * w * (C / 4);
#define DWT_LAUNCH(TX_, TS_)                                                                                            \
    hipLaunchKernelGGL((dwt2_kernel<TX_, TS_>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const TX_*)x,   \
                       ldx, H, W, C / 4, h, w, (TS_*)ll, ld_ll, (TS_*)hf, ld_hf, total)
    if (x_dtype == UWIR_F32 && sub_dtype == UWIR_F32) DWT_LAUNCH(float, float);
    else if (x_dtype == UWIR_F32) DWT_LAUNCH(float, bf16_t);
    else if (sub_dtype == UWIR_F32) DWT_LAUNCH(bf16_t, float);
    else DWT_LAUNCH(bf16_t, bf16_t);
#undef DWT_LAUNCH
    UWIR_LAUNCH_CHECK("dwt2");
    return UWIR_OK;
}

extern "C" int uwir_idwt2(const float* ll, int ld_ll, const void* hf, int hf_dtype, int ld_hf, int B, int h, int w, int C,
                          void* out, int out_dtype, int ldo, uwir_stream_t stream) {
    UWIR_CHECK(ll && hf && out, "idwt2: null pointer");
    UWIR_CHECK((hf_dtype == UWIR_F32 || hf_dtype == UWIR_BF16) && (out_dtype == UWIR_F32 || out_dtype == UWIR_BF16), "idwt2: bad dtype");
    UWIR_CHECK(B > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0, "idwt2: bad shape");
    UWIR_CHECK(ldo >= C && ld_ll >= C && ld_hf >= 3 * C && ldo % 4 == 0 && ld_ll % 4 == 0 && ld_hf % 4 == 0,
               "idwt2: bad strides");
    size_t total = (size_t)B * h * w * (C / 4);
#define IDWT_LAUNCH(TH_, TO_)                                                                                           \
    hipLaunchKernelGGL((idwt2_kernel<TH_, TO_>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, ll, ld_ll,      \
                       (const TH_*)hf, ld_hf, h, w, C / 4, (TO_*)out, ldo, total)
    if (hf_dtype == UWIR_F32 && out_dtype == UWIR_F32) IDWT_LAUNCH(float, float);
    else if (hf_dtype == UWIR_F32) IDWT_LAUNCH(float, bf16_t);
    else if (out_dtype == UWIR_F32) IDWT_LAUNCH(bf16_t, float);
    else IDWT_LAUNCH(bf16_t, bf16_t);
#undef IDWT_LAUNCH
    UWIR_LAUNCH_CHECK("idwt2");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// LayerNorm2d statistics: per (b, c) mean / biased variance over the n rows of x [B, n, C].
// Pass 1: block (chunk of rows, channel slab, b) accumulates shifted sums sum(x-K), sum((x-K)^2)
//         (K = x[b,0,c], keeps fp32 partials well conditioned) -> fp64 partials in workspace.
// Pass 2: one thread per (b,c) combines partials in a FIXED order (deterministic) in fp64 and
//         emits the folded affine a*x + s.
// ------------------------------------------------------------------------------------------
#define LN_ROWS_PER_BLOCK 128
#define LN_SLAB_Q 64  // channel quads per block (256 channels)

template <typename TX>
__global__ void __launch_bounds__(256) ln_partial_kernel(const TX* __restrict__ x, int ldx, int n, int C4,
                                                         double* __restrict__ part, int nchunks) {
    // threads: tx = quad within the slab (0..qs-1), ty = row lane (0..rl-1); rl = 256 / qs uses (nearly) every thread
    // whatever the width (C = 192: 48 quads x 5 row lanes), 4 independent 16-byte loads in flight per thread
    const int qs = min(C4 - blockIdx.y * LN_SLAB_Q, LN_SLAB_Q);
    const int rl = 256 / qs;
    const int tx = threadIdx.x % qs, ty = threadIdx.x / qs;
    const int b = blockIdx.z, chunk = blockIdx.x;
    const int q = blockIdx.y * LN_SLAB_Q + tx;
    __shared__ float red[2][256][4];  // [sum|sq][ty * qs + tx][4]
    float4 s = make_float4(0, 0, 0, 0), ss = make_float4(0, 0, 0, 0);
    const bool active = ty < rl;
    if (active) {
        const TX* base = x + (size_t)b * n * ldx + 4 * q;
        const float4 K = load4(base);
        const int r0 = chunk * LN_ROWS_PER_BLOCK, r1 = min(n, r0 + LN_ROWS_PER_BLOCK);
#define LN_ACC(v)                                                                   \
    {                                                                               \
        float dx = v.x - K.x, dy = v.y - K.y, dz = v.z - K.z, dw = v.w - K.w;       \
        s.x += dx; s.y += dy; s.z += dz; s.w += dw;                                 \
        ss.x += dx * dx; ss.y += dy * dy; ss.z += dz * dz; ss.w += dw * dw;         \
    }
        int r = r0 + ty;
        for (; r + 3 * rl < r1; r += 4 * rl) {
            const float4 v0 = load4(base + (size_t)r * ldx), v1 = load4(base + (size_t)(r + rl) * ldx);
            const float4 v2 = load4(base + (size_t)(r + 2 * rl) * ldx), v3 = load4(base + (size_t)(r + 3 * rl) * ldx);
            LN_ACC(v0) LN_ACC(v1) LN_ACC(v2) LN_ACC(v3)
        }
        for (; r < r1; r += rl) {
            const float4 v = load4(base + (size_t)r * ldx);
            LN_ACC(v)
        }
#undef LN_ACC
        *reinterpret_cast<float4*>(red[0][ty * qs + tx]) = s;
        *reinterpret_cast<float4*>(red[1][ty * qs + tx]) = ss;
    }
    __syncthreads();
    if (ty == 0) {
        int C = 4 * C4;
        double* o = part + (((size_t)b * nchunks + chunk) * C + 4 * q) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double a0 = 0, a1 = 0;
            for (int l = 0; l < rl; ++l) {   // fixed order: deterministic
                a0 += (double)red[0][l * qs + tx][j];
                a1 += (double)red[1][l * qs + tx][j];
            }
            o[2 * j] = a0;
            o[2 * j + 1] = a1;
        }
    }
}

// bf16 rows, C % 8 == 0: the same partial sums from 16-byte loads (8 channels per thread; the 4-channel form above moves
// 8 bytes per load and ran the bf16 sub-band rows at 2.6 TB/s).  Same shift K, same per-thread row order r0 + ty + k * rl
// and the same fixed-order fp64 combination, but rl differs from the 4-channel form, so the two round differently in the
// last bits of the fp32 partials (both are deterministic).
#define LN_SLAB_O 32  // channel octets per block (256 channels)
__global__ void __launch_bounds__(256) ln_partial8_kernel(const bf16_t* __restrict__ x, int ldx, int n, int C8,
                                                          double* __restrict__ part, int nchunks) {
    const int qs = min(C8 - blockIdx.y * LN_SLAB_O, LN_SLAB_O);
    const int rl = 256 / qs;
    const int tx = threadIdx.x % qs, ty = threadIdx.x / qs;
    const int b = blockIdx.z, chunk = blockIdx.x;
    const int q = blockIdx.y * LN_SLAB_O + tx;
    __shared__ float red[2][256][8];
    float s[8], ss[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
    const bool active = ty < rl;
    if (active) {
        const bf16_t* base = x + (size_t)b * n * ldx + 8 * q;
        float K[8];
        {
            const bf16x8 k8 = *reinterpret_cast<const bf16x8*>(base);
#pragma unroll
            for (int j = 0; j < 8; ++j) K[j] = (float)k8[j];
        }
        const int r0 = chunk * LN_ROWS_PER_BLOCK, r1 = min(n, r0 + LN_ROWS_PER_BLOCK);
        auto acc8 = [&](const bf16x8& v) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = (float)v[j] - K[j];
                s[j] += d;
                ss[j] += d * d;
            }
        };
        int r = r0 + ty;
        for (; r + 3 * rl < r1; r += 4 * rl) {
            const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(base + (size_t)r * ldx);
            const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(base + (size_t)(r + rl) * ldx);
            const bf16x8 v2 = *reinterpret_cast<const bf16x8*>(base + (size_t)(r + 2 * rl) * ldx);
            const bf16x8 v3 = *reinterpret_cast<const bf16x8*>(base + (size_t)(r + 3 * rl) * ldx);
            acc8(v0); acc8(v1); acc8(v2); acc8(v3);
        }
        for (; r < r1; r += rl) acc8(*reinterpret_cast<const bf16x8*>(base + (size_t)r * ldx));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[0][ty * qs + tx][j] = s[j];
            red[1][ty * qs + tx][j] = ss[j];
        }
    }
    __syncthreads();
    // 8 * qs (channel) outputs: thread t < 8 * qs combines channel (t / 8 -> octet, t % 8) in a fixed order
    for (int t = threadIdx.x; t < 8 * qs; t += 256) {
        const int o8 = t >> 3, j = t & 7;
        double a0 = 0, a1 = 0;
        for (int l = 0; l < rl; ++l) {
            a0 += (double)red[0][l * qs + o8][j];
            a1 += (double)red[1][l * qs + o8][j];
        }
        const int C = 8 * C8, c = 8 * (blockIdx.y * LN_SLAB_O + o8) + j;
        double* o = part + (((size_t)b * nchunks + chunk) * C + c) * 2;
        o[0] = a0;
        o[1] = a1;
    }
}

template <typename TX>
__global__ void __launch_bounds__(256) ln_finalize_kernel(const TX* __restrict__ x, int ldx, int n, int C,
                                                          const double* __restrict__ part, int nchunks, float eps,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ scale_shift, int ss_ld,
                                                          float* __restrict__ a_out, float* __restrict__ s_out,
                                                          float* __restrict__ mean_out,
                                                          float* __restrict__ pooled_out, int B) {
    // 16 lanes per (b, c): lane l adds chunks l, l+16, ... then the 16 partial sums are combined by a fixed butterfly
    // (bit-reproducible); a single thread walking 128 chunk partials was latency-bound (12 us per launch)
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gi >> 4, l = gi & 15;
    const bool valid = i < B * C;
    const int b = valid ? i / C : 0, c = valid ? i % C : 0;
    double s = 0, ss = 0;
    for (int k = l; k < nchunks; k += 16) {
        const double* p = part + (((size_t)b * nchunks + k) * C + c) * 2;
        s += p[0];
        ss += p[1];
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        ss += __shfl_xor(ss, off);
    }
    if (!valid || l != 0) return;
    double K = (double)(float)x[(size_t)b * n * ldx + c];
    double md = s / n;
    double mean = K + md;
    double var = ss / n - md * md;
    if (var < 0) var = 0;
    if (mean_out) mean_out[i] = (float)mean;
    if (a_out) {
        double rstd = 1.0 / sqrt(var + (double)eps);
        double g = gamma[c], bt = beta[c];
        double sc = 1.0, sh = 0.0;
        if (scale_shift) {
            sc = 1.0 + (double)scale_shift[(size_t)b * ss_ld + c];
            sh = (double)scale_shift[(size_t)b * ss_ld + C + c];
        }
        a_out[i] = (float)(rstd * g * sc);
        s_out[i] = (float)((bt - mean * rstd * g) * sc + sh);
        // spatial mean of the normalised map: mean((x-mu)*rstd) == 0, so it is beta*(1+scale)+shift
        if (pooled_out) pooled_out[i] = (float)(bt * sc + sh);
    }
}

extern "C" size_t uwir_ln2d_workspace_bytes(int B, int n, int C) {
    size_t nchunks = ceil_div(n, LN_ROWS_PER_BLOCK);
    return (size_t)B * nchunks * C * 2 * sizeof(double);
}

extern "C" int uwir_ln2d_affine(const void* x, int x_dtype, int ldx, int B, int n, int C, float eps, const float* gamma,
                                const float* beta, const float* scale_shift, int ss_ld, float* a_out, float* s_out,
                                float* mean_out, float* pooled_out, void* workspace, size_t workspace_bytes,
                                uwir_stream_t stream) {
    UWIR_CHECK(x && workspace, "ln2d: null pointer");
    UWIR_CHECK(x_dtype == UWIR_F32 || x_dtype == UWIR_BF16, "ln2d: bad input dtype");
    UWIR_CHECK(B > 0 && n > 0 && C > 0 && C % 4 == 0 && ldx >= C && ldx % 4 == 0, "ln2d: bad shape B=%d n=%d C=%d ldx=%d",
               B, n, C, ldx);
    UWIR_CHECK((a_out && s_out && gamma && beta) || (!a_out && !s_out && mean_out), "ln2d: inconsistent outputs");
    UWIR_CHECK(!scale_shift || ss_ld >= 2 * C, "ln2d: ss_ld too small");
    UWIR_CHECK(workspace_bytes >= uwir_ln2d_workspace_bytes(B, n, C), "ln2d: workspace too small");
    UWIR_CHECK(B <= 65535, "ln2d: batch too large");
    int nchunks = ceil_div(n, LN_ROWS_PER_BLOCK);
    int C4 = C / 4;
    dim3 grid(nchunks, ceil_div(C4, LN_SLAB_Q), B);
#define LN2D_LAUNCH(TX_)                                                                                                  \
    hipLaunchKernelGGL(ln_partial_kernel<TX_>, grid, dim3(256), 0, (hipStream_t)stream, (const TX_*)x, ldx, n, C4,        \
                       (double*)workspace, nchunks);                                                                      \
    UWIR_LAUNCH_CHECK("ln_partial");                                                                                      \
    hipLaunchKernelGGL(ln_finalize_kernel<TX_>, dim3(ceil_div(B * C * 16, 256)), dim3(256), 0, (hipStream_t)stream,       \
                       (const TX_*)x, ldx, n, C, (const double*)workspace, nchunks, eps, gamma, beta, scale_shift, ss_ld, \
                       a_out, s_out, mean_out, pooled_out, B);
    if (x_dtype == UWIR_BF16 && C % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0) {
        hipLaunchKernelGGL(ln_partial8_kernel, dim3(nchunks, ceil_div(C / 8, LN_SLAB_O), B), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, ldx, n, C / 8, (double*)workspace, nchunks);
        UWIR_LAUNCH_CHECK("ln_partial8");
        hipLaunchKernelGGL(ln_finalize_kernel<bf16_t>, dim3(ceil_div(B * C * 16, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, ldx, n, C, (const double*)workspace, nchunks, eps, gamma, beta, scale_shift, ss_ld,
                           a_out, s_out, mean_out, pooled_out, B);
    } else if (x_dtype == UWIR_BF16) { LN2D_LAUNCH(bf16_t) } else { LN2D_LAUNCH(float) }
#undef LN2D_LAUNCH
    UWIR_LAUNCH_CHECK("ln_finalize");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// Allocation-free variant (include/uwir.h: uwir_ln2d_affine_noalloc).  Same two kernels; the fp64 partial sums live in
// a file-scope device array instead of a caller workspace.  Every GPU holds its own copy of the array at its own
// address: the address is looked up once per device (with that device current) and only that pointer is ever handed
// to kernels -- the array's NAME is a host-side stand-in, not a device address.
// ------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char ln2d_scratch[UWIR_LN2D_NOALLOC_SCRATCH_BYTES];

#define UWIR_MAX_DEVICES 64
static void* g_ln2d_scratch_ptr[UWIR_MAX_DEVICES];   // per device; written once under the mutex, then read-only
static std::mutex g_ln2d_scratch_mu;

static int ln2d_scratch_for_current_device(void** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return uwir_set_error(UWIR_ELAUNCH, "ln2d_noalloc: hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= UWIR_MAX_DEVICES) return uwir_set_error(UWIR_EUNSUPPORTED, "ln2d_noalloc: device index %d", dev);
    std::lock_guard<std::mutex> lock(g_ln2d_scratch_mu);
    if (!g_ln2d_scratch_ptr[dev]) {
        void* ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(ln2d_scratch));   // this device's copy (device `dev` is current)
        if (e != hipSuccess || !ptr)
            return uwir_set_error(UWIR_ELAUNCH, "ln2d_noalloc: hipGetSymbolAddress: %s", hipGetErrorString(e));
        g_ln2d_scratch_ptr[dev] = ptr;
    }
    *out = g_ln2d_scratch_ptr[dev];
    return UWIR_OK;
}

extern "C" size_t uwir_ln2d_noalloc_max_bytes(void) { return UWIR_LN2D_NOALLOC_SCRATCH_BYTES; }

extern "C" int uwir_ln2d_affine_noalloc(const void* x, int x_dtype, int ldx, int B, int n, int C, float eps, const float* gamma,
                                        const float* beta, const float* scale_shift, int ss_ld, float* a_out,
                                        float* s_out, float* mean_out, float* pooled_out, uwir_stream_t stream) {
    UWIR_CHECK(B > 0 && n > 0 && C > 0, "ln2d_noalloc: bad shape B=%d n=%d C=%d", B, n, C);
    const size_t need = uwir_ln2d_workspace_bytes(B, n, C);
    UWIR_CHECK(need <= (size_t)UWIR_LN2D_NOALLOC_SCRATCH_BYTES,
               "ln2d_noalloc: B=%d n=%d C=%d needs %zu bytes of partial sums, the built-in scratch holds %zu "
               "(use uwir_ln2d_affine with a caller workspace)", B, n, C, need, (size_t)UWIR_LN2D_NOALLOC_SCRATCH_BYTES);
    void* ptr = nullptr;
    int rc = ln2d_scratch_for_current_device(&ptr);
    if (rc != UWIR_OK) return rc;
    // every partial that ln_finalize reads is written by ln_partial of THIS call first: nothing needs clearing
    return uwir_ln2d_affine(x, x_dtype, ldx, B, n, C, eps, gamma, beta, scale_shift, ss_ld, a_out, s_out, mean_out, pooled_out,
                            ptr, need, stream);
}

// ------------------------------------------------------------------------------------------
// Apply a folded per-(b,c) norm affine and cast: out[m,c] = a[b,c]*x[m,c] + s[b,c] (bf16 or fp32 out).
// Used where the consumer GEMM is better fed by the LDS-DMA bf16 path than by normalise-on-load:
// one extra HBM pass (6 B / element) buys the 16-byte DMA pipeline for the whole GEMM.
// ------------------------------------------------------------------------------------------
template <typename TX, typename TO>
__global__ void __launch_bounds__(256) affine_cast_kernel(const TX* __restrict__ x, int ldx, int n, int C4,
                                                          const float* __restrict__ a, const float* __restrict__ sh,
                                                          TO* __restrict__ out, int ldo, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int c4 = (int)(i % C4);
        size_t m = i / C4;
        int b = (int)(m / n);
        float4 v = load4(x + m * ldx + 4 * c4);
        float4 sc = load4(a + ((size_t)b * C4 + c4) * 4), h = load4(sh + ((size_t)b * C4 + c4) * 4);
        store4(out + m * ldo + 4 * c4,
               make_float4(fmaf(v.x, sc.x, h.x), fmaf(v.y, sc.y, h.y), fmaf(v.z, sc.z, h.z), fmaf(v.w, sc.w, h.w)));
    }
}

extern "C" int uwir_affine_cast(const void* x, int x_dtype, int ldx, int B, int n, int C, const float* a, const float* s,
                                void* out, int out_dtype, int ldo, uwir_stream_t stream) {
    UWIR_CHECK(x && a && s && out, "affine_cast: null pointer");
    UWIR_CHECK(x_dtype == UWIR_F32 || x_dtype == UWIR_BF16, "affine_cast: bad input dtype");
    UWIR_CHECK(B > 0 && n > 0 && C > 0 && C % 4 == 0 && ldx >= C && ldx % 4 == 0 && ldo >= C && ldo % 4 == 0,
               "affine_cast: bad shape");
    UWIR_CHECK(out_dtype == UWIR_F32 || out_dtype == UWIR_BF16, "affine_cast: bad output dtype");
    size_t total = (size_t)B * n * (C / 4);
#define AC_LAUNCH(TX_, TO_)                                                                                             \
    hipLaunchKernelGGL((affine_cast_kernel<TX_, TO_>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,           \
                       (const TX_*)x, ldx, n, C / 4, a, s, (TO_*)out, ldo, total)
    if (x_dtype == UWIR_F32 && out_dtype == UWIR_F32) AC_LAUNCH(float, float);
    else if (x_dtype == UWIR_F32) AC_LAUNCH(float, bf16_t);
    else if (out_dtype == UWIR_F32) AC_LAUNCH(bf16_t, float);
    else AC_LAUNCH(bf16_t, bf16_t);
#undef AC_LAUNCH
    UWIR_LAUNCH_CHECK("affine_cast");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// The per-sample affine of LayerNorm2d / AdaLayerNorm2d (S/utils.py:18-36) folded into the NEXT 1x1 conv's weights:
//   W (a_b * x + s_b) + bias = (W diag(a_b)) x + (W s_b + bias)
// -> one bf16 weight panel Wb[b] = W * a_b (column-wise) and one bias row bb[b] = W s_b + bias per image, so the GEMM reads
// the un-normalised rows directly (LDS-DMA path) and no normalised copy of the activation is written or read.
// One block per (b, 4 output rows): thread = 4-column piece; the bias row is a block reduction in fp32.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fold_affine_weights_kernel(const float* __restrict__ W, int N, int K,
                                                                  const float* __restrict__ a, const float* __restrict__ s,
                                                                  const float* __restrict__ bias, bf16_t* __restrict__ Wb,
                                                                  float* __restrict__ bb) {
    const int b = blockIdx.y, n0 = blockIdx.x * 4, tid = threadIdx.x;
    __shared__ float red[4][4];
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k4 = tid; k4 < K / 4; k4 += 256) {
        const float4 a4 = load4(a + (size_t)b * K + 4 * k4), s4 = load4(s + (size_t)b * K + 4 * k4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (n0 + r < N) {
                const float4 w4 = load4(W + (size_t)(n0 + r) * K + 4 * k4);
                store4(Wb + ((size_t)b * N + n0 + r) * K + 4 * k4, make_float4(w4.x * a4.x, w4.y * a4.y, w4.z * a4.z, w4.w * a4.w));
                part[r] += w4.x * s4.x + w4.y * s4.y + w4.z * s4.z + w4.w * s4.w;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = part[r];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((tid & 63) == 0) red[r][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < 4 && n0 + tid < N)
        bb[(size_t)b * N + n0 + tid] = ((red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3])) + (bias ? bias[n0 + tid] : 0.f);
}

extern "C" int uwir_fold_affine_weights(const float* W, int N, int K, int B, const float* a, const float* s, const float* bias,
                                        void* Wb, float* bb, uwir_stream_t stream) {
    UWIR_CHECK(W && a && s && Wb && bb, "fold_affine_weights: null pointer");
    UWIR_CHECK(N > 0 && K > 0 && K % 8 == 0 && B > 0 && B <= 65535, "fold_affine_weights: bad shape (K %% 8 == 0)");
    UWIR_CHECK((((uintptr_t)W | (uintptr_t)a | (uintptr_t)s | (uintptr_t)Wb) & 15) == 0, "fold_affine_weights: 16-byte alignment");
    hipLaunchKernelGGL(fold_affine_weights_kernel, dim3(ceil_div(N, 4), B), dim3(256), 0, (hipStream_t)stream, W, N, K, a, s, bias,
                       (bf16_t*)Wb, bb);
    UWIR_LAUNCH_CHECK("fold_affine_weights");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// GroupNorm (U/models/modules/attention.py:76-77,236,254: 32 groups, eps 1e-6, affine) as a folded
// per-(b,c) affine a*x + s for the consumer GEMM's normalise-on-load.  Statistics: the per-channel
// shifted partial sums of ln_partial_kernel, merged per group in fp64 (parallel-variance formula).
// ------------------------------------------------------------------------------------------
// One wave per (b, group): lane = (channel j of the group, chunk lane kk); the chunk sums of a channel are spread over
// its 64 / cg lanes and combined by xor shuffles (a single thread per group walking cg x nchunks fp64 records took 23 us
// at B = 8, C = 512, n = 1024 -- longer than the statistics pass itself).
__global__ void __launch_bounds__(256) gn_finalize_kernel(const float* __restrict__ x, int ldx, int n, int C, int groups,
                                                          const double* __restrict__ part, int nchunks, float eps,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ a_out,
                                                          float* __restrict__ s_out, int B) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= B * groups) return;                                   // wave-uniform
    const int b = i / groups, g = i % groups, cg = C / groups;
    // cgp = lanes along the channel axis (power of two >= cg, at most 64); channels beyond 64 per group loop
    int cgp = 1;
    while (cgp < cg && cgp < 64) cgp <<= 1;
    const int kl = 64 / cgp, jl = lane % cgp, kk = lane / cgp;
    double sum_mean = 0, sum_m2 = 0, sum_mean2 = 0;
    for (int j0 = 0; j0 < cg; j0 += cgp) {
        const int j = j0 + jl;
        const bool ok = j < cg;
        const int c = g * cg + (ok ? j : 0);
        double s = 0, ss = 0;
        if (ok)
            for (int k = kk; k < nchunks; k += kl) {
                const double* p = part + (((size_t)b * nchunks + k) * C + c) * 2;
                s += p[0];
                ss += p[1];
            }
        for (int off = cgp; off < 64; off <<= 1) {                  // over the chunk lanes of this channel
            s += __shfl_xor(s, off);
            ss += __shfl_xor(ss, off);
        }
        if (ok && kk == 0) {
            const double md = s / n;
            const double mean = (double)x[(size_t)b * n * ldx + c] + md;
            sum_mean += mean;
            sum_mean2 += mean * mean;
            sum_m2 += ss - s * md;
        }
    }
    for (int off = 1; off < 64; off <<= 1) {                        // over the channels (lanes with kk != 0 hold zeros)
        sum_mean += __shfl_xor(sum_mean, off);
        sum_mean2 += __shfl_xor(sum_mean2, off);
        sum_m2 += __shfl_xor(sum_m2, off);
    }
    const double gmean = sum_mean / cg;
    double var = (sum_m2 + (double)n * (sum_mean2 - cg * gmean * gmean)) / ((double)n * cg);
    if (var < 0) var = 0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    for (int j = lane; j < cg; j += 64) {
        const int c = g * cg + j;
        const double a = rstd * (double)gamma[c];
        a_out[(size_t)b * C + c] = (float)a;
        s_out[(size_t)b * C + c] = (float)((double)beta[c] - gmean * a);
    }
}

extern "C" int uwir_groupnorm_affine(const float* x, int ldx, int B, int n, int C, int groups, float eps,
                                     const float* gamma, const float* beta, float* a_out, float* s_out, void* workspace,
                                     size_t workspace_bytes, uwir_stream_t stream) {
    UWIR_CHECK(x && gamma && beta && a_out && s_out && workspace, "groupnorm: null pointer");
    UWIR_CHECK(B > 0 && B <= 65535 && n > 0 && C > 0 && C % 4 == 0 && groups > 0 && C % groups == 0 && ldx >= C &&
                   ldx % 4 == 0,
               "groupnorm: bad shape B=%d n=%d C=%d groups=%d", B, n, C, groups);
    UWIR_CHECK(workspace_bytes >= uwir_ln2d_workspace_bytes(B, n, C), "groupnorm: workspace too small");
    int nchunks = ceil_div(n, LN_ROWS_PER_BLOCK);
    int C4 = C / 4;
    dim3 grid(nchunks, ceil_div(C4, LN_SLAB_Q), B);
    hipLaunchKernelGGL(ln_partial_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, n, C4, (double*)workspace,
                       nchunks);
    UWIR_LAUNCH_CHECK("gn_partial");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(ceil_div(B * groups, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, n,
                       C, groups, (const double*)workspace, nchunks, eps, gamma, beta, a_out, s_out, B);
    UWIR_LAUNCH_CHECK("gn_finalize");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// ChannelAttention fold (S/wtb.py:24-28,118-119): x + xn*w with xn = a*x + s  ==  x*(1 + a*w) + s*w,
// i.e. the residual x enters the proj GEMM epilogue through a per-(b,c) affine.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) chan_attn_fold_kernel(const float* __restrict__ a, const float* __restrict__ s,
                                                             const float* __restrict__ w, float* __restrict__ os,
                                                             float* __restrict__ oh, int count) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        os[i] = 1.f + a[i] * w[i];
        oh[i] = s[i] * w[i];
    }
}

extern "C" int uwir_chan_attn_fold(const float* a, const float* s, const float* w, float* out_scale, float* out_shift,
                                   int count, uwir_stream_t stream) {
    UWIR_CHECK(a && s && w && out_scale && out_shift && count > 0, "chan_attn_fold: bad arguments");
    hipLaunchKernelGGL(chan_attn_fold_kernel, dim3(ceil_div(count, 256)), dim3(256), 0, (hipStream_t)stream, a, s, w,
                       out_scale, out_shift, count);
    UWIR_LAUNCH_CHECK("chan_attn_fold");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// tiny linear y[b, o] = act(x[b, :] . W[o, :] + bias[o]) for a handful of rows b.
// linear_kernel: one wave per output element (b, o) -- the launch-latency-sized calls (a few thousand outputs).
// linear_wide_kernel: cond / time embeddings -> all the AdaLN / gate / ResBlock vectors of a forward in one call
// (B = 16, O = 19200, I = 512 in the student).  There the weight matrix is the only real traffic (39 MB), so it is read
// ONCE: a wave owns 4 output columns and keeps their weight elements in registers while it walks up to 16 rows of x
// staged in LDS (one wave per element re-reads every weight row B times: 0.14-0.21 ms; this form 0.063 ms, now bound by
// its 64 butterfly reductions per wave, not by the stream; on a small call it costs 25 us instead of 4, hence two kernels).  Per element both accumulate
// in the same order (lane-strided partial sums, then the xor butterfly): bit-identical results.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) linear_kernel(const float* __restrict__ x, int ldx, int B, int I,
                                                     const float* __restrict__ W, const float* __restrict__ bias,
                                                     int O, int act, float* __restrict__ y, int ldy) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) / UWIR_WAVE;
    int lane = threadIdx.x % UWIR_WAVE;
    if (wave >= B * O) return;  // whole wave exits together
    int b = wave / O, o = wave % O;
    const float* xr = x + (size_t)b * ldx;
    const float* wr = W + (size_t)o * I;
    float acc = 0.f;
    for (int i = lane; i < I; i += UWIR_WAVE) acc = fmaf(xr[i], wr[i], acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
        if (bias) acc += bias[o];
        y[(size_t)b * ldy + o] = act_apply(acc, act);
    }
}

#define LIN_OW 4    // output columns per wave
template <int LIN_BT>   // rows of x per pass (LDS: LIN_BT x I floats)
__global__ void __launch_bounds__(256) linear_wide_kernel(const float* __restrict__ x, int ldx, int B, int I,
                                                          const float* __restrict__ W, const float* __restrict__ bias,
                                                          int O, int act, float* __restrict__ y, int ldy) {
    extern __shared__ float sx[];   // [LIN_BT][I]
    const int wave = threadIdx.x / UWIR_WAVE, lane = threadIdx.x % UWIR_WAVE;
    const int o0 = (blockIdx.x * 4 + wave) * LIN_OW;
    const float* wr[LIN_OW];
#pragma unroll
    for (int k = 0; k < LIN_OW; ++k) wr[k] = W + (size_t)min(o0 + k, O - 1) * I;
    for (int b0 = 0; b0 < B; b0 += LIN_BT) {
        const int nb = min(LIN_BT, B - b0);
        __syncthreads();
        for (int ii = threadIdx.x; ii < I; ii += 256) {   // LIN_BT independent loads in flight, then the LDS stores
            float v[LIN_BT];
#pragma unroll
            for (int t = 0; t < LIN_BT; ++t) v[t] = t < nb ? x[(size_t)(b0 + t) * ldx + ii] : 0.f;
#pragma unroll
            for (int t = 0; t < LIN_BT; ++t) sx[t * I + ii] = v[t];
        }
        __syncthreads();
        float acc[LIN_OW][LIN_BT];
#pragma unroll
        for (int k = 0; k < LIN_OW; ++k)
#pragma unroll
            for (int t = 0; t < LIN_BT; ++t) acc[k][t] = 0.f;
        for (int i = lane; i < I; i += UWIR_WAVE) {
            float w[LIN_OW];
#pragma unroll
            for (int k = 0; k < LIN_OW; ++k) w[k] = wr[k][i];
#pragma unroll
            for (int t = 0; t < LIN_BT; ++t) {
                const float xv = sx[t * I + i];
#pragma unroll
                for (int k = 0; k < LIN_OW; ++k) acc[k][t] = fmaf(xv, w[k], acc[k][t]);
            }
        }
#pragma unroll
        for (int k = 0; k < LIN_OW; ++k)
#pragma unroll
            for (int t = 0; t < LIN_BT; ++t) {
                float a = acc[k][t];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
                if (lane == 0 && t < nb && o0 + k < O) {
                    if (bias) a += bias[o0 + k];
                    y[(size_t)(b0 + t) * ldy + o0 + k] = act_apply(a, act);
                }
            }
    }
}

extern "C" int uwir_linear(const float* x, int ldx, int B, int I, const float* W, const float* bias, int O, int act,
                           float* y, int ldy, uwir_stream_t stream) {
    UWIR_CHECK(x && W && y, "linear: null pointer");
    UWIR_CHECK(B > 0 && I > 0 && O > 0 && ldx >= I && ldy >= O, "linear: bad shape");
    hipStream_t s = (hipStream_t)stream;
    if (B >= 8 && I <= 768 && (size_t)O * I >= (1u << 20)) {   // >= 4 MB of weights, <= 48 KiB of LDS
        hipLaunchKernelGGL(linear_wide_kernel<16>, dim3(ceil_div(O, 4 * LIN_OW)), dim3(256), (size_t)16 * I * sizeof(float), s, x,
                           ldx, B, I, W, bias, O, act, y, ldy);
    } else {
        size_t waves = (size_t)B * O;
        UWIR_CHECK(waves < (1u << 28), "linear: too many outputs");
        hipLaunchKernelGGL(linear_kernel, dim3((unsigned)ceil_div_sz(waves, 4)), dim3(256), 0, s, x, ldx, B, I, W, bias, O, act,
                           y, ldy);
    }
    UWIR_LAUNCH_CHECK("linear");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// y[b,:] = softmax(x[b,:]) * mul  -- the prompt weighting of the IR-SDE UNet
// (U/models/modules/DenoisingUNet_arch.py:135).  One wave per row.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* __restrict__ x, int ldx, int B, int C,
                                                           const float* __restrict__ mul, float* __restrict__ y,
                                                           int ldy) {
    int row = (blockIdx.x * blockDim.x + threadIdx.x) / UWIR_WAVE;
    int lane = threadIdx.x % UWIR_WAVE;
    if (row >= B) return;
    const float* xr = x + (size_t)row * ldx;
    float m = -INFINITY;
    for (int c = lane; c < C; c += UWIR_WAVE) m = fmaxf(m, xr[c]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    float s = 0.f;
    for (int c = lane; c < C; c += UWIR_WAVE) s += expf(xr[c] - m);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float inv = 1.f / s;
    for (int c = lane; c < C; c += UWIR_WAVE) y[(size_t)row * ldy + c] = expf(xr[c] - m) * inv * (mul ? mul[c] : 1.f);
}

extern "C" int uwir_softmax_rows(const float* x, int ldx, int B, int C, const float* mul, float* y, int ldy,
                                 uwir_stream_t stream) {
    UWIR_CHECK(x && y && B > 0 && C > 0 && ldx >= C && ldy >= C, "softmax_rows: bad arguments");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, B, C, mul, y,
                       ldy);
    UWIR_LAUNCH_CHECK("softmax_rows");
    return UWIR_OK;
}

// A window descriptor (device memory, caller-built) is trusted only after this check: every kernel that takes `win`
// evaluates it per block / element and LEAVES without touching memory when it fails, so a bad descriptor handed through
// the public C ABI writes nothing instead of reading or writing outside the frames.  NF = number of frames.
__device__ __forceinline__ bool window_ok(const uwir_window& w, int NF, int FH, int FW, int T) {
    return w.frame >= 0 && w.frame < NF && w.y >= 0 && w.x >= 0 && w.y <= FH - T && w.x <= FW - T;
}
__device__ __forceinline__ bool window_core_ok(const uwir_window& w, int T) {
    return w.cy0 >= w.y && w.cy1 <= w.y + T && w.cx0 >= w.x && w.cx1 <= w.x + T && w.cy0 <= w.cy1 && w.cx0 <= w.cx1;
}

// ------------------------------------------------------------------------------------------
// bilinear resize of feature ROWS (NHWC -> NHWC, align_corners=False): the decoder's F.interpolate of an up-sampled map to
// its skip's size (S/model.py:150-151), which the reference takes whenever a level's side is odd.  One thread per pixel and
// 4-channel group; same source-index rule and product order as bilinear_kernel below.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) bilinear_rows_kernel(const T* __restrict__ src, int lds_, int B, int hs, int ws, int C,
                                                            T* __restrict__ dst, int ldd, int hd, int wd, float sy, float sx) {
    const int c4 = C / 4;
    const size_t total = (size_t)B * hd * wd * c4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4) * 4;
        const size_t p = i / c4;
        const int ox = (int)(p % wd);
        const size_t q = p / wd;
        const int oy = (int)(q % hd), b = (int)(q / hd);
        const float fy = fmaxf(sy * (oy + 0.5f) - 0.5f, 0.f), fx = fmaxf(sx * (ox + 0.5f) - 0.5f, 0.f);
        const int y0 = min((int)fy, hs - 1), x0 = min((int)fx, ws - 1);
        const int y1 = min(y0 + 1, hs - 1), x1 = min(x0 + 1, ws - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const T* s = src + (size_t)b * hs * ws * lds_ + c;
        const float4 v00 = load4(s + ((size_t)y0 * ws + x0) * lds_), v01 = load4(s + ((size_t)y0 * ws + x1) * lds_);
        const float4 v10 = load4(s + ((size_t)y1 * ws + x0) * lds_), v11 = load4(s + ((size_t)y1 * ws + x1) * lds_);
        float4 o;
        o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
        o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
        o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
        o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
        store4(dst + p * ldd + c, o);
    }
}

extern "C" int uwir_bilinear_rows(const void* src, int lds_, int B, int hs, int ws, int C, void* dst, int ldd, int hd, int wd,
                                  int dtype, uwir_stream_t stream) {
    UWIR_CHECK(src && dst, "bilinear_rows: null pointer");
    UWIR_CHECK(B > 0 && hs > 0 && ws > 0 && hd > 0 && wd > 0 && C > 0 && C % 4 == 0 && lds_ >= C && ldd >= C &&
                   lds_ % 4 == 0 && ldd % 4 == 0, "bilinear_rows: bad shape (C and the pixel strides must be multiples of 4)");
    UWIR_CHECK(dtype == UWIR_F32 || dtype == UWIR_BF16, "bilinear_rows: dtype must be fp32 or bf16");
    const size_t total = (size_t)B * hd * wd * (C / 4);
    const float sy = (float)hs / (float)hd, sx = (float)ws / (float)wd;
    if (dtype == UWIR_BF16)
        hipLaunchKernelGGL(bilinear_rows_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)src, lds_, B, hs, ws, C, (bf16_t*)dst, ldd, hd, wd, sy, sx);
    else
        hipLaunchKernelGGL(bilinear_rows_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)src, lds_, B, hs, ws, C, (float*)dst, ldd, hd, wd, sy, sx);
    UWIR_LAUNCH_CHECK("bilinear_rows");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// bilinear resize (align_corners=False, no antialias), NCHW source -> NHWC destination columns
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bilinear_kernel(const float* __restrict__ src, int B, int M, int hs, int ws,
                                                       float* __restrict__ dst, int ldd, int hd, int wd, float sy,
                                                       float sx, const uwir_window* __restrict__ win, int FH, int FW,
                                                       int NF) {
    size_t total = (size_t)B * hd * wd * M;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int m = (int)(i % M);
        size_t p = i / M;
        int ox = (int)(p % wd);
        size_t q = p / wd;
        int oy = (int)(q % hd);
        int b = (int)(q / hd);
        // ATen area_pixel_compute_source_index, align_corners=False, cubic=False: clamp below at 0
        float fy = fmaxf(sy * (oy + 0.5f) - 0.5f, 0.f);
        float fx = fmaxf(sx * (ox + 0.5f) - 0.5f, 0.f);
        int y0 = min((int)fy, hs - 1), x0 = min((int)fx, ws - 1);
        int y1 = min(y0 + 1, hs - 1), x1 = min(x0 + 1, ws - 1);
        float ly = fy - (float)y0, lx = fx - (float)x0;
        // plain batch: image b is contiguous; windows: image b is the hs x ws window of frame win[b].frame at (y, x)
        if (win && !window_ok(win[b], NF, FH, FW, hs)) continue;
        const float* s = win ? src + (((size_t)win[b].frame * M + m) * FH + win[b].y) * FW + win[b].x
                             : src + ((size_t)b * M + m) * hs * ws;
        const int rs = win ? FW : ws;
        float v00 = s[(size_t)y0 * rs + x0], v01 = s[(size_t)y0 * rs + x1];
        float v10 = s[(size_t)y1 * rs + x0], v11 = s[(size_t)y1 * rs + x1];
        float hy = 1.f - ly, hx = 1.f - lx;
        dst[p * ldd + m] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    }
}

extern "C" int uwir_bilinear_nchw_to_nhwc(const float* src, int B, int M, int hs, int ws, float* dst, int ldd, int hd,
                                          int wd, uwir_stream_t stream) {
    UWIR_CHECK(src && dst, "bilinear: null pointer");
    UWIR_CHECK(B > 0 && M > 0 && hs > 0 && ws > 0 && hd > 0 && wd > 0 && ldd >= M, "bilinear: bad shape");
    size_t total = (size_t)B * hd * wd * M;
    hipLaunchKernelGGL(bilinear_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, src, B, M, hs, ws,
                       dst, ldd, hd, wd, (float)hs / (float)hd, (float)ws / (float)wd, (const uwir_window*)nullptr, 0, 0, 0);
    UWIR_LAUNCH_CHECK("bilinear");
    return UWIR_OK;
}

extern "C" int uwir_bilinear_win(const float* frames, int n_frames, int M, int FH, int FW, const uwir_window* win, int nwin,
                                 int T, float* dst, int ldd, int hd, int wd, uwir_stream_t stream) {
    UWIR_CHECK(frames && win && dst, "bilinear_win: null pointer");
    UWIR_CHECK(n_frames > 0 && nwin > 0 && M > 0 && T > 0 && T <= FH && T <= FW && hd > 0 && wd > 0 && ldd >= M, "bilinear_win: bad shape");
    size_t total = (size_t)nwin * hd * wd * M;
    hipLaunchKernelGGL(bilinear_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, frames, nwin, M, T, T, dst,
                       ldd, hd, wd, (float)T / (float)hd, (float)T / (float)wd, win, FH, FW, n_frames);
    UWIR_LAUNCH_CHECK("bilinear_win");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// stem conv 3x3 (tiny Cin) from NCHW input to NHWC features: thread = pixel x 4 output channels
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) stem_kernel(const float* __restrict__ x, int B, int Cin, int H, int W,
                                                   const float* __restrict__ w, const float* __restrict__ bias,
                                                   int Co4, float* __restrict__ out, int ldo) {
    size_t total = (size_t)B * H * W * Co4;
    int Cout = 4 * Co4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int c4 = (int)(i % Co4);
        size_t p = i / Co4;
        int ox = (int)(p % W);
        size_t q = p / W;
        int oy = (int)(q % H);
        int b = (int)(q / H);
        float4 acc = bias ? load4(bias + 4 * c4) : make_float4(0, 0, 0, 0);
        for (int ci = 0; ci < Cin; ++ci) {
            const float* xp = x + ((size_t)b * Cin + ci) * H * W;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                int iy = oy + ky - 1;
                if (iy < 0 || iy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    int ix = ox + kx - 1;
                    if (ix < 0 || ix >= W) continue;
                    float v = xp[(size_t)iy * W + ix];
                    float4 wv = load4(w + ((size_t)(ky * 3 + kx) * Cin + ci) * Cout + 4 * c4);
                    acc.x = fmaf(v, wv.x, acc.x);
                    acc.y = fmaf(v, wv.y, acc.y);
                    acc.z = fmaf(v, wv.z, acc.z);
                    acc.w = fmaf(v, wv.w, acc.w);
                }
            }
        }
        store4(out + p * ldo + 4 * c4, acc);
    }
}

// Thread-per-pixel form (Cin <= 4, Cout % 16 == 0): the 9 * Cin inputs of a pixel sit in registers, the weights are
// block-uniform (scalar loads, SGPR operands), 16 output channels at a time.  The form above spends most of its
// instructions on 27 scalar + 27 vector loads per 4 outputs (0.9 TB/s on the 268 MB fp32 output of the student stem).
template <typename TO>
__global__ void __launch_bounds__(256) stem_pixel_kernel(const float* __restrict__ x, int B, int Cin, int H, int W,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         int Cout, TO* __restrict__ out, int ldo,
                                                         const uwir_window* __restrict__ win, int FH, int FW, int NF) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (ox >= W || oy >= H) return;
    if (win && !window_ok(win[b], NF, FH, FW, H)) return;
    // plain batch: image b is contiguous NCHW; windows: image b is the H x W window of frame win[b].frame at (y, x) --
    // the conv still pads with zeros at the WINDOW border (every window is an image of its own to the network)
    const size_t plane = win ? (size_t)FH * FW : (size_t)H * W;
    const int rs = win ? FW : W;
    const float* xb = win ? x + ((size_t)win[b].frame * Cin * FH + win[b].y) * FW + win[b].x : x + (size_t)b * Cin * plane;
    float in[36];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
            in[t * 4 + ci] = (ok && ci < Cin) ? xb[ci * plane + (size_t)iy * rs + ix] : 0.f;
    }
    TO* o = out + (((size_t)b * H + oy) * W + ox) * ldo;
    for (int c0 = 0; c0 < Cout; c0 += 16) {
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = bias ? bias[c0 + j] : 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            for (int ci = 0; ci < Cin; ++ci) {
                const float* wr = w + ((size_t)t * Cin + ci) * Cout + c0;   // uniform address
                const float v = in[t * 4 + ci];
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = fmaf(v, wr[j], acc[j]);
            }
#pragma unroll
        for (int j = 0; j < 16; j += 4) store4(o + c0 + j, make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]));
    }
}

// The student's stem (3x3, Cin <= 4 -> Cout = 64, fp32 NCHW image -> NHWC rows) on the matrix cores.  K = 9 taps x 4
// channel slots (the 4th, and the 3rd of a 2-channel input, zero) = 36, padded to 64: two k-steps of
// v_mfma_f32_16x16x32_bf16 per 16 pixels and 16-channel tile; a lane's 8-element piece of a k-step is two taps of ONE pixel
// column, gathered from the NCHW planes (the image is 4 % of the bytes this kernel moves) and split hi + lo, as are the
// weights (hi x hi + lo x hi + hi x lo: the fp32 product to ~2^-16, the output rows are the only rounding in bf16 mode).
// Output as in the 7x7 stem kernel of the UNet: a lane owns 16 consecutive channels of its pixel.  The thread-per-pixel
// form wrote its 134 MB of bf16 rows at 0.7 TB/s (B = 16, 256 x 256).
__device__ __forceinline__ int chan_perm16(int rho) {   // gemm.hip's chan_perm: LDS row j * 16 + r holds channel (r >> 2) * 16 + j * 4 + (r & 3)
    return (rho & 64) | (((rho >> 2) & 3) << 4) | (((rho >> 4) & 3) << 2) | (rho & 3);
}
template <typename TO>
__global__ void __launch_bounds__(256) stem3_mfma_kernel(const float* __restrict__ x, int Cin, int H, int W,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         TO* __restrict__ out, int ldo, const uwir_window* __restrict__ win,
                                                         int FH, int FW, int NF) {
    constexpr int NFR = 2 * 4;                                     // fragments: (k-step s, channel tile j)
    __shared__ __attribute__((aligned(16))) bf16_t s_w[2 * NFR * 64 * 8];   // [hi | lo][fragment][lane][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int b = blockIdx.z;
    if (win && !window_ok(win[b], NF, FH, FW, H)) return;          // block-uniform, before the barrier
    // k = 32 s + 8 fq + e  ->  tap t = 8 s + 2 fq + (e >> 2), channel slot e & 3; taps >= 9 and slots >= Cin are zero
    for (int i = tid; i < NFR * 64; i += 256) {
        const int f = i >> 6, l = i & 63, j = f & 3, s2 = f >> 2;
        const int ch = chan_perm16(j * 16 + (l & 15)), q = l >> 4;
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int t = 8 * s2 + 2 * q + (e >> 2), ci = e & 3;
            const float v = (t < 9 && ci < Cin) ? w[((size_t)t * Cin + ci) * 64 + ch] : 0.f;   // w is [3, 3, Cin, Cout]
            const bf16_t h = (bf16_t)v;
            hi[e] = h;
            lo[e] = (bf16_t)(v - (float)h);
        }
        *reinterpret_cast<bf16x8*>(&s_w[(size_t)i * 8]) = hi;
        *reinterpret_cast<bf16x8*>(&s_w[(size_t)(NFR * 64 + i) * 8]) = lo;
    }
    __syncthreads();
    const int x0 = blockIdx.x * 16, ybase = blockIdx.y * 32 + wave * 8;
    const size_t plane = win ? (size_t)FH * FW : (size_t)H * W;
    const int rs = win ? FW : W;
    const float* xb = win ? x + ((size_t)win[b].frame * Cin * FH + win[b].y) * FW + win[b].x : x + (size_t)b * Cin * plane;
    float bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[j][e] = bias ? bias[fq * 16 + j * 4 + e] : 0.f;
    const int px = x0 + fr;
#pragma unroll 1
    for (int r = 0; r < 8; ++r) {
        const int y = ybase + r;
        if (y >= H) break;                                          // wave-uniform
        bf16x8 ah[2], al[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int t = 8 * s2 + 2 * fq + half;
                const int iy = y + t / 3 - 1, ix = px + t % 3 - 1;
                const bool ok = t < 9 && iy >= 0 && iy < H && ix >= 0 && ix < W;   // zero padding at the WINDOW border
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    const float v = (ok && ci < Cin) ? xb[ci * plane + (size_t)iy * rs + ix] : 0.f;
                    const bf16_t h = (bf16_t)v;
                    ah[s2][4 * half + ci] = h;
                    al[s2][4 * half + ci] = (bf16_t)(v - (float)h);
                }
            }
        }
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = (f32x4){bv[j][0], bv[j][1], bv[j][2], bv[j][3]};
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&s_w[(size_t)((s2 * 4 + j) * 64 + lane) * 8]);
                const bf16x8 wl = *reinterpret_cast<const bf16x8*>(&s_w[(size_t)((NFR + s2 * 4 + j) * 64 + lane) * 8]);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah[s2], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al[s2], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[s2], acc[j], 0, 0, 0);
            }
        if (px < W) {
            TO* op = out + (((size_t)b * H + y) * W + px) * ldo + fq * 16;
            if constexpr (sizeof(TO) == 2) {
#pragma unroll
                for (int jp = 0; jp < 4; jp += 2) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)acc[jp][e]; o[4 + e] = (bf16_t)acc[jp + 1][e]; }
                    *reinterpret_cast<bf16x8*>(op + jp * 4) = o;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(op + j * 4) = acc[j];
            }
        }
    }
}
static int g_stem_mfma = 1;   // tools/ and tests: uwir_head_variant bit 1 set keeps the thread-per-pixel stem
static bool stem3_ok(int B, int Cin, int H, int Cout, const void* out, int ldo) {
    return g_stem_mfma && Cin >= 1 && Cin <= 4 && Cout == 64 && ldo % 8 == 0 && ((uintptr_t)out & 15) == 0 && B <= 65535 &&
           ceil_div(H, 32) <= 65535;
}

extern "C" int uwir_stem_conv3x3(const float* x_nchw, int B, int Cin, int H, int W, const float* w, const float* bias,
                                 int Cout, void* outv, int out_dtype, int ldo, uwir_stream_t stream) {
    UWIR_CHECK(x_nchw && w && outv, "stem: null pointer");
    UWIR_CHECK(out_dtype == UWIR_F32 || out_dtype == UWIR_BF16, "stem: bad output dtype");
    float* out = (float*)outv;
    UWIR_CHECK(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout % 4 == 0 && ldo >= Cout && ldo % 4 == 0, "stem: bad shape");
    if (out_dtype == UWIR_BF16 && stem3_ok(B, Cin, H, Cout, outv, ldo)) {   // (fp32 rows keep the exact fp32 kernel)
        hipLaunchKernelGGL(stem3_mfma_kernel<bf16_t>, dim3(ceil_div(W, 16), ceil_div(H, 32), B), dim3(256), 0, (hipStream_t)stream,
                           x_nchw, Cin, H, W, w, bias, (bf16_t*)outv, ldo, (const uwir_window*)nullptr, 0, 0, 0);
        UWIR_LAUNCH_CHECK("stem(mfma)");
        return UWIR_OK;
    }
    if (Cin <= 4 && Cout % 16 == 0 && B <= 65535 && ceil_div(H, 4) <= 65535) {
        if (out_dtype == UWIR_BF16)
            hipLaunchKernelGGL(stem_pixel_kernel<bf16_t>, dim3(ceil_div(W, 64), ceil_div(H, 4), B), dim3(256), 0, (hipStream_t)stream,
                               x_nchw, B, Cin, H, W, w, bias, Cout, (bf16_t*)outv, ldo, (const uwir_window*)nullptr, 0, 0, 0);
        else
            hipLaunchKernelGGL(stem_pixel_kernel<float>, dim3(ceil_div(W, 64), ceil_div(H, 4), B), dim3(256), 0, (hipStream_t)stream,
                               x_nchw, B, Cin, H, W, w, bias, Cout, out, ldo, (const uwir_window*)nullptr, 0, 0, 0);
        UWIR_LAUNCH_CHECK("stem(pixel)");
        return UWIR_OK;
    }
    UWIR_CHECK(out_dtype == UWIR_F32, "stem: bf16 rows need Cin <= 4 and Cout %% 16 == 0");
    size_t total = (size_t)B * H * W * (Cout / 4);
    hipLaunchKernelGGL(stem_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x_nchw, B, Cin, H, W, w,
                       bias, Cout / 4, out, ldo);
    UWIR_LAUNCH_CHECK("stem");
    return UWIR_OK;
}

extern "C" int uwir_stem_conv3x3_win(const float* frames, int n_frames, int Cin, int FH, int FW, const uwir_window* win,
                                     int nwin, int T, const float* w, const float* bias, int Cout, void* out, int out_dtype, int ldo,
                                     uwir_stream_t stream) {
    UWIR_CHECK(frames && win && w && out, "stem_win: null pointer");
    UWIR_CHECK(out_dtype == UWIR_F32 || out_dtype == UWIR_BF16, "stem_win: bad output dtype");
    UWIR_CHECK(n_frames > 0 && nwin > 0 && nwin <= 65535 && T > 0 && T <= FH && T <= FW && ldo >= Cout && ldo % 4 == 0,
               "stem_win: bad shape");
    UWIR_CHECK(Cin >= 1 && Cin <= 4 && Cout % 16 == 0, "stem_win: needs Cin <= 4 and Cout %% 16 == 0");
    if (out_dtype == UWIR_BF16 && stem3_ok(nwin, Cin, T, Cout, out, ldo)) {
        hipLaunchKernelGGL(stem3_mfma_kernel<bf16_t>, dim3(ceil_div(T, 16), ceil_div(T, 32), nwin), dim3(256), 0, (hipStream_t)stream,
                           frames, Cin, T, T, w, bias, (bf16_t*)out, ldo, win, FH, FW, n_frames);
        UWIR_LAUNCH_CHECK("stem_win(mfma)");
        return UWIR_OK;
    }
    if (out_dtype == UWIR_BF16)
        hipLaunchKernelGGL(stem_pixel_kernel<bf16_t>, dim3(ceil_div(T, 64), ceil_div(T, 4), nwin), dim3(256), 0, (hipStream_t)stream,
                           frames, nwin, Cin, T, T, w, bias, Cout, (bf16_t*)out, ldo, win, FH, FW, n_frames);
    else
        hipLaunchKernelGGL(stem_pixel_kernel<float>, dim3(ceil_div(T, 64), ceil_div(T, 4), nwin), dim3(256), 0, (hipStream_t)stream,
                           frames, nwin, Cin, T, T, w, bias, Cout, (float*)out, ldo, win, FH, FW, n_frames);
    UWIR_LAUNCH_CHECK("stem_win");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// head: normalise-on-load (final LayerNorm2d affine) -> conv3x3 (zero pad AFTER the norm)
//       -> + input residual -> clamp.  One wave per output pixel: lanes split (tap, channel-quad)
//       work, Cout (3) accumulators per lane, wave reduction.  x is NHWC, outputs NCHW.
// ------------------------------------------------------------------------------------------
#define HEAD_MAX_COUT 4
__global__ void __launch_bounds__(256) head_kernel(const float* __restrict__ x, int ldx, int B, int H, int W, int Cin,
                                                   const float* __restrict__ a_scale,
                                                   const float* __restrict__ a_shift, const float* __restrict__ w,
                                                   const float* __restrict__ bias, int Cout,
                                                   const float* __restrict__ res, float* __restrict__ out,
                                                   float* __restrict__ preclamp, int Ho, int Wo, int do_clamp) {
    // input map H x W (zero padded conv), outputs only the top-left Ho x Wo crop (NCHW [B,Cout,Ho,Wo])
    const int lane = threadIdx.x % UWIR_WAVE;
    const int C4 = Cin / 4;
    const int items = 9 * C4;  // (tap, quad)
    size_t npix = (size_t)B * Ho * Wo;
    size_t wave0 = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) / UWIR_WAVE;
    size_t nwaves = ((size_t)gridDim.x * blockDim.x) / UWIR_WAVE;
    for (size_t p = wave0; p < npix; p += nwaves) {
        int ox = (int)(p % Wo);
        size_t q = p / Wo;
        int oy = (int)(q % Ho);
        int b = (int)(q / Ho);
        float acc[HEAD_MAX_COUT] = {0.f, 0.f, 0.f, 0.f};
        for (int it = lane; it < items; it += UWIR_WAVE) {
            int tap = it / C4, c4 = it % C4;
            int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            float4 v = load4(x + (((size_t)b * H + iy) * W + ix) * ldx + 4 * c4);
            if (a_scale) {
                float4 sc = load4(a_scale + (size_t)b * Cin + 4 * c4), sh = load4(a_shift + (size_t)b * Cin + 4 * c4);
                v.x = fmaf(v.x, sc.x, sh.x);
                v.y = fmaf(v.y, sc.y, sh.y);
                v.z = fmaf(v.z, sc.z, sh.z);
                v.w = fmaf(v.w, sc.w, sh.w);
            }
#pragma unroll
            for (int co = 0; co < HEAD_MAX_COUT; ++co) {
                if (co < Cout) {
                    float4 wv = load4(w + ((size_t)co * 9 + tap) * Cin + 4 * c4);
                    acc[co] += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
                }
            }
        }
#pragma unroll
        for (int co = 0; co < HEAD_MAX_COUT; ++co) {
            float a = acc[co];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
            acc[co] = a;
        }
        if (lane < Cout) {
            float v = acc[0];
#pragma unroll
            for (int co = 1; co < HEAD_MAX_COUT; ++co) v = (lane == co) ? acc[co] : v;
            size_t o = (((size_t)b * Cout + lane) * Ho + oy) * Wo + ox;
            v += (bias ? bias[lane] : 0.f);
            if (res) v += res[o];
            if (preclamp) preclamp[o] = v;
            out[o] = do_clamp ? fminf(fmaxf(v, 0.f), 1.f) : v;
        }
    }
}

// LDS-tiled form (Cin % 16 == 0): a block owns an 8 x 32 pixel tile; the (8+2) x (32+2) halo tile is staged through
// LDS 16 channels at a time (normalise-on-load affine applied on the way in, zero padding stays zero) and every thread
// accumulates the Cout outputs of ONE pixel from 16-byte LDS reads.  The wave-per-pixel form above re-reads every
// input pixel 9 times from L2 and leaves half its lanes idle (335 GB/s on the student head); here the map is read
// once (+ halo) and the stores / residual loads are coalesced NCHW rows.
#define HT_H 8
#define HT_W 32
#define HT_CC 16
#define HT_PS 20  // LDS pixel stride in floats: 16 channels + 4 pad (conflict-free 16-byte reads across a row of lanes)
template <typename TX>
__global__ void __launch_bounds__(256) head_tile_kernel(const TX* __restrict__ x, int ldx, int H, int W, int Cin,
                                                        const float* __restrict__ a_scale,
                                                        const float* __restrict__ a_shift, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int Cout,
                                                        const float* __restrict__ res, float* __restrict__ out,
                                                        float* __restrict__ preclamp, int Ho, int Wo, int do_clamp,
                                                        const uwir_window* __restrict__ win, int FH, int FW, int NF) {
    constexpr int PH = HT_H + 2, PW = HT_W + 2, NPIX = PH * PW;
    __shared__ __attribute__((aligned(16))) float tile[NPIX * HT_PS];
    const int tid = threadIdx.x, tx = tid % HT_W, ty = tid / HT_W;
    const int x0 = blockIdx.x * HT_W, y0 = blockIdx.y * HT_H, b = blockIdx.z;
    // windows: image b is a window of a frame; only its OWNED core [cy0,cy1) x [cx0,cx1) (frame coordinates) is written,
    // straight into the output frame, and the residual is read from the input frame.  Tiles without an owned pixel leave.
    uwir_window wd = {0, 0, 0, 0, 0, 0, 0, 0};
    if (win) {
        wd = win[b];
        if (!window_ok(wd, NF, FH, FW, H) || !window_core_ok(wd, H)) return;
        if (wd.y + y0 >= wd.cy1 || wd.y + y0 + HT_H <= wd.cy0 || wd.x + x0 >= wd.cx1 || wd.x + x0 + HT_W <= wd.cx0) return;
    }
    float acc[HEAD_MAX_COUT] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < Cin; c0 += HT_CC) {
        if (c0) __syncthreads();
        for (int i = tid; i < NPIX * (HT_CC / 4); i += 256) {
            const int pix = i / (HT_CC / 4), qd = i % (HT_CC / 4);
            const int iy = y0 + pix / PW - 1, ix = x0 + pix % PW - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                v = load4(x + (((size_t)b * H + iy) * W + ix) * ldx + c0 + 4 * qd);
                if (a_scale) {
                    const float4 sc = load4(a_scale + (size_t)b * Cin + c0 + 4 * qd), sh = load4(a_shift + (size_t)b * Cin + c0 + 4 * qd);
                    v.x = fmaf(v.x, sc.x, sh.x);
                    v.y = fmaf(v.y, sc.y, sh.y);
                    v.z = fmaf(v.z, sc.z, sh.z);
                    v.w = fmaf(v.w, sc.w, sh.w);
                }
            }
            *reinterpret_cast<float4*>(&tile[pix * HT_PS + 4 * qd]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* tp = &tile[((ty + tap / 3) * PW + tx + tap % 3) * HT_PS];
#pragma unroll
            for (int qd = 0; qd < HT_CC / 4; ++qd) {
                const float4 v = *reinterpret_cast<const float4*>(tp + 4 * qd);
#pragma unroll
                for (int co = 0; co < HEAD_MAX_COUT; ++co) {
                    if (co < Cout) {
                        const float* wp = w + ((size_t)co * 9 + tap) * Cin + c0 + 4 * qd;  // block-uniform: scalar loads
                        acc[co] = fmaf(v.x, wp[0], acc[co]);
                        acc[co] = fmaf(v.y, wp[1], acc[co]);
                        acc[co] = fmaf(v.z, wp[2], acc[co]);
                        acc[co] = fmaf(v.w, wp[3], acc[co]);
                    }
                }
            }
        }
    }
    const int oy = y0 + ty, ox = x0 + tx;
    if (oy >= Ho || ox >= Wo) return;
    if (win && (wd.y + oy < wd.cy0 || wd.y + oy >= wd.cy1 || wd.x + ox < wd.cx0 || wd.x + ox >= wd.cx1)) return;
#pragma unroll
    for (int co = 0; co < HEAD_MAX_COUT; ++co) {
        if (co < Cout) {
            const size_t o = win ? (((size_t)wd.frame * Cout + co) * FH + wd.y + oy) * FW + wd.x + ox
                                 : (((size_t)b * Cout + co) * Ho + oy) * Wo + ox;
            float v = acc[co] + (bias ? bias[co] : 0.f);
            if (res) v += res[o];
            if (preclamp) preclamp[o] = v;
            out[o] = do_clamp ? fminf(fmaxf(v, 0.f), 1.f) : v;
        }
    }
}

// The student's head (final LayerNorm2d affine -> conv3x3 64 -> <= 4 -> + input -> clamp, S/model.py:158-161) on bf16
// rows on the matrix cores, same scheme as the tail kernel above: output channels as MFMA rows, 16 pixels of an image row
// as columns, a wave walks 8 rows of a 16-pixel strip with three input rows of fragments live.  The norm's per-(image,
// channel) affine a x + s is folded into the block's weights (a block serves one image): w' = w a enters LDS as hi + lo
// bf16 pairs (two MFMAs per fragment: the bf16 rows are exact operands, nothing of the last layer is rounded to bf16), and
// the shift becomes one constant per (tap, output channel), T[t][co] = sum_c w[co][t][c] s[c], added for the taps that lie
// inside the image -- the conv pads with zeros AFTER the norm, so border pixels take fewer of them.  Window mode as in
// head_tile_kernel.  The per-pixel tile kernel read its bf16 rows at 0.75 TB/s.
template <int NCH>
__global__ void __launch_bounds__(256) head_mfma_kernel(const bf16_t* __restrict__ x, int ldx, int H, int W,
                                                        const float* __restrict__ a_scale, const float* __restrict__ a_shift,
                                                        const float* __restrict__ w, const float* __restrict__ bias, int Cout,
                                                        const float* __restrict__ res, float* __restrict__ out, int do_clamp,
                                                        const uwir_window* __restrict__ win, int FH, int FW, int NF) {
    constexpr int NF9 = 9 * NCH, Cin = 32 * NCH;
    __shared__ __attribute__((aligned(16))) bf16_t s_w[2 * NF9 * 64 * 8];   // [hi | lo][fragment = tap * NCH + chunk][lane][8]
    __shared__ float s_tap[9][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int b = blockIdx.z;
    uwir_window wd = {0, 0, 0, 0, 0, 0, 0, 0};
    if (win) {                                                      // block-uniform exits come before the barrier
        wd = win[b];
        if (!window_ok(wd, NF, FH, FW, H) || !window_core_ok(wd, H)) return;
        const int x0 = blockIdx.x * 16, y0 = blockIdx.y * 32;
        if (wd.y + y0 >= wd.cy1 || wd.y + y0 + 32 <= wd.cy0 || wd.x + x0 >= wd.cx1 || wd.x + x0 + 16 <= wd.cx0) return;
    }
    for (int i = tid; i < NF9 * 64; i += 256) {
        const int f = i >> 6, l = i & 63, co = l & 15, kq = l >> 4, tap = f / NCH, c = f % NCH;
        bf16x8 hi, lo;
        float wv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (co < Cout) {
            const float* wp = w + ((size_t)co * 9 + tap) * Cin + c * 32 + kq * 8;
            const float4 w0 = load4(wp), w1 = load4(wp + 4);
            wv[0] = w0.x; wv[1] = w0.y; wv[2] = w0.z; wv[3] = w0.w; wv[4] = w1.x; wv[5] = w1.y; wv[6] = w1.z; wv[7] = w1.w;
            if (a_scale) {
                const float* ap = a_scale + (size_t)b * Cin + c * 32 + kq * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) wv[j] *= ap[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = wv[j];
            const bf16_t h = (bf16_t)v;
            hi[j] = h;
            lo[j] = (bf16_t)(v - (float)h);
        }
        *reinterpret_cast<bf16x8*>(&s_w[(size_t)i * 8]) = hi;
        *reinterpret_cast<bf16x8*>(&s_w[(size_t)(NF9 * 64 + i) * 8]) = lo;
    }
    if (tid < 36) {                                                 // T[tap][co]: the shift's share of every tap
        const int tap = tid >> 2, co = tid & 3;
        float t = 0.f;
        if (a_shift && co < Cout) {
            const float* wp = w + ((size_t)co * 9 + tap) * Cin;
            const float* sp = a_shift + (size_t)b * Cin;
            for (int c = 0; c < Cin; ++c) t = fmaf(wp[c], sp[c], t);
        }
        s_tap[tap][co] = t;
    }
    __syncthreads();
    const int px = blockIdx.x * 16 + fr, ybase = blockIdx.y * 32 + wave * 8;
    const bf16_t* xb = x + (size_t)b * H * W * ldx + fq * 8;
    float bv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = (bias && e < Cout) ? bias[e] : 0.f;
    auto load_row = [&](int iy, bf16x8 (&dst)[3 * NCH]) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = px + kx - 1;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const bf16_t* src = xb + ((size_t)(ok ? iy : 0) * W + (ok ? ix : 0)) * ldx;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                bf16x8 v = *reinterpret_cast<const bf16x8*>(src + c * 32);
                if (!ok) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)0.f;
                }
                dst[kx * NCH + c] = v;
            }
        }
    };
    bf16x8 rows[3][3 * NCH];
    load_row(ybase - 1, rows[0]);
    load_row(ybase, rows[1]);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int y = ybase + r;
        if (y < H) {                                                // wave-uniform
            load_row(y + 1, rows[(r + 2) % 3]);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};   // two chains: hi and lo weights
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int t = 0; t < 3 * NCH; ++t) {
                    const int f = ky * 3 * NCH + t;
                    const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&s_w[(size_t)(f * 64 + lane) * 8]);
                    const bf16x8 wl = *reinterpret_cast<const bf16x8*>(&s_w[(size_t)((NF9 + f) * 64 + lane) * 8]);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, rows[(r + ky) % 3][t], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, rows[(r + ky) % 3][t], acc1, 0, 0, 0);
                }
            bool own = fq == 0 && px < W;
            if (win) own = own && wd.y + y >= wd.cy0 && wd.y + y < wd.cy1 && wd.x + px >= wd.cx0 && wd.x + px < wd.cx1;
            if (own) {
                float tb[4] = {bv[0], bv[1], bv[2], bv[3]};
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int iy = y + t / 3 - 1, ix = px + t % 3 - 1;
                    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) tb[e] += s_tap[t][e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < Cout) {
                        const size_t o = win ? (((size_t)wd.frame * Cout + e) * FH + wd.y + y) * FW + wd.x + px
                                             : (((size_t)b * Cout + e) * H + y) * W + px;
                        float v = (acc0[e] + acc1[e]) + tb[e];
                        if (res) v += res[o];
                        out[o] = do_clamp ? fminf(fmaxf(v, 0.f), 1.f) : v;
                    }
            }
        }
    }
}

static bool head_mfma_ok(const void* x, int ldx, int B, int H, int Cin, const float* w, const float* preclamp) {
    return (Cin == 32 || Cin == 64) && !preclamp && ldx % 8 == 0 && (((uintptr_t)x | (uintptr_t)w) & 15) == 0 && B <= 65535 &&
           ceil_div(H, 32) <= 65535;
}
static int g_head_mfma = 1;   // tools/: uwir_head_variant(0) keeps the per-pixel tile kernel
extern "C" void uwir_head_variant(int v) {
    g_head_mfma = v & 1;
    g_stem_mfma = (v & 2) ? 0 : 1;
}

static bool head_tiled_ok(int B, int Ho, int Wo, int Cin) {
    return Cin % HT_CC == 0 && B <= 65535 && ceil_div(Ho, HT_H) <= 65535;
}

extern "C" int uwir_head_conv3x3(const void* xv, int x_dtype, int ldx, int B, int H, int W, int Cin, const float* a_scale,
                                 const float* a_shift, const float* w, const float* bias, int Cout,
                                 const float* residual_nchw, float* out_nchw, float* preclamp_nchw,
                                 uwir_stream_t stream) {
    UWIR_CHECK(xv && w && out_nchw, "head: null pointer");
    UWIR_CHECK(x_dtype == UWIR_F32 || x_dtype == UWIR_BF16, "head: bad input dtype");
    const float* x = (const float*)xv;
    UWIR_CHECK(B > 0 && H > 0 && W > 0 && Cin % 4 == 0 && Cout >= 1 && Cout <= HEAD_MAX_COUT && ldx >= Cin &&
                   ldx % 4 == 0,
               "head: bad shape (Cout<=%d)", HEAD_MAX_COUT);
    UWIR_CHECK((a_scale == nullptr) == (a_shift == nullptr), "head: a_scale/a_shift must come together");
    if (g_head_mfma && x_dtype == UWIR_BF16 && head_mfma_ok(xv, ldx, B, H, Cin, w, preclamp_nchw)) {
        const dim3 grid(ceil_div(W, 16), ceil_div(H, 32), B);
        if (Cin == 64)
            hipLaunchKernelGGL((head_mfma_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)xv, ldx, H, W,
                               a_scale, a_shift, w, bias, Cout, residual_nchw, out_nchw, 1, (const uwir_window*)nullptr, 0, 0, 0);
        else
            hipLaunchKernelGGL((head_mfma_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)xv, ldx, H, W,
                               a_scale, a_shift, w, bias, Cout, residual_nchw, out_nchw, 1, (const uwir_window*)nullptr, 0, 0, 0);
        UWIR_LAUNCH_CHECK("head(mfma)");
        return UWIR_OK;
    }
    if (head_tiled_ok(B, H, W, Cin)) {
        if (x_dtype == UWIR_BF16)
            hipLaunchKernelGGL(head_tile_kernel<bf16_t>, dim3(ceil_div(W, HT_W), ceil_div(H, HT_H), B), dim3(256), 0,
                               (hipStream_t)stream, (const bf16_t*)xv, ldx, H, W, Cin, a_scale, a_shift, w, bias, Cout,
                               residual_nchw, out_nchw, preclamp_nchw, H, W, 1, (const uwir_window*)nullptr, 0, 0, 0);
        else
            hipLaunchKernelGGL(head_tile_kernel<float>, dim3(ceil_div(W, HT_W), ceil_div(H, HT_H), B), dim3(256), 0,
                               (hipStream_t)stream, x, ldx, H, W, Cin, a_scale, a_shift, w, bias, Cout, residual_nchw, out_nchw,
                               preclamp_nchw, H, W, 1, (const uwir_window*)nullptr, 0, 0, 0);
        UWIR_LAUNCH_CHECK("head(tile)");
        return UWIR_OK;
    }
    UWIR_CHECK(x_dtype == UWIR_F32, "head: bf16 rows need Cin %% 16 == 0");
    size_t npix = (size_t)B * H * W;
    size_t blocks = ceil_div_sz(npix, 4);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(head_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, B, H, W, Cin,
                       a_scale, a_shift, w, bias, Cout, residual_nchw, out_nchw, preclamp_nchw, H, W, 1);
    UWIR_LAUNCH_CHECK("head");
    return UWIR_OK;
}

extern "C" int uwir_head_conv3x3_win(const void* x, int x_dtype, int ldx, int nwin, int T, int Cin, const float* a_scale,
                                     const float* a_shift, const float* w, const float* bias, int Cout,
                                     const float* frames_in, float* frames_out, int n_frames, int FH, int FW,
                                     const uwir_window* win, uwir_stream_t stream) {
    UWIR_CHECK(x && w && frames_in && frames_out && win, "head_win: null pointer");
    UWIR_CHECK(n_frames > 0 && nwin > 0 && nwin <= 65535 && T > 0 && T <= FH && T <= FW && Cin % HT_CC == 0 && Cout >= 1 &&
                   Cout <= HEAD_MAX_COUT && ldx >= Cin && ldx % 4 == 0,
               "head_win: bad shape (Cin %% 16 == 0, Cout <= %d)", HEAD_MAX_COUT);
    UWIR_CHECK((a_scale == nullptr) == (a_shift == nullptr), "head_win: a_scale/a_shift must come together");
    UWIR_CHECK(x_dtype == UWIR_F32 || x_dtype == UWIR_BF16, "head_win: bad input dtype");
    if (g_head_mfma && x_dtype == UWIR_BF16 && head_mfma_ok(x, ldx, nwin, T, Cin, w, nullptr)) {
        const dim3 grid(ceil_div(T, 16), ceil_div(T, 32), nwin);
        if (Cin == 64)
            hipLaunchKernelGGL((head_mfma_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, T, T,
                               a_scale, a_shift, w, bias, Cout, frames_in, frames_out, 1, win, FH, FW, n_frames);
        else
            hipLaunchKernelGGL((head_mfma_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, T, T,
                               a_scale, a_shift, w, bias, Cout, frames_in, frames_out, 1, win, FH, FW, n_frames);
        UWIR_LAUNCH_CHECK("head_win(mfma)");
        return UWIR_OK;
    }
    if (x_dtype == UWIR_BF16)
        hipLaunchKernelGGL(head_tile_kernel<bf16_t>, dim3(ceil_div(T, HT_W), ceil_div(T, HT_H), nwin), dim3(256), 0,
                           (hipStream_t)stream, (const bf16_t*)x, ldx, T, T, Cin, a_scale, a_shift, w, bias, Cout, frames_in,
                           frames_out, (float*)nullptr, T, T, 1, win, FH, FW, n_frames);
    else
        hipLaunchKernelGGL(head_tile_kernel<float>, dim3(ceil_div(T, HT_W), ceil_div(T, HT_H), nwin), dim3(256), 0,
                           (hipStream_t)stream, (const float*)x, ldx, T, T, Cin, a_scale, a_shift, w, bias, Cout, frames_in,
                           frames_out, (float*)nullptr, T, T, 1, win, FH, FW, n_frames);
    UWIR_LAUNCH_CHECK("head_win");
    return UWIR_OK;
}

// The UNet's tail conv on bf16 rows (Cin = 32 or 64 -> Cout <= 4) on the matrix cores: the per-pixel FMA form above spends
// 110 us on a map it could read in 15 (B = 8, 256 x 256, Cin 64).  v_mfma_f32_16x16x32_bf16 with the (<= 4 used of 16)
// output channels as rows and 16 pixels of an image row as columns; the fp32 weights enter as hi + lo bf16 pairs (two
// MFMAs per fragment: 16 mantissa bits, the last layer's weights are not rounded to bf16), staged once per block in LDS in
// fragment order; the activation fragments are 16-byte global loads (pixel (y + ky - 1, x + kx - 1), 8 channels), zero
// outside the image.  A wave walks 8 rows of a 16-pixel column strip; lanes 0-15 hold the pixel's channels 0-3.
// fp32 rows (TX = float: the precise tail path of the UNet's bf16 mode): each 8-channel piece is split into hi + lo bf16
// fragments in registers and takes three MFMAs (w_hi x_hi + w_lo x_hi + w_hi x_lo: the product to ~2^-16).
template <int NCH, typename TX>
__global__ void __launch_bounds__(256) tail_mfma_kernel(const TX* __restrict__ x, int ldx, int H, int W,
                                                        const float* __restrict__ w, const float* __restrict__ bias, int Cout,
                                                        float* __restrict__ out, int Ho, int Wo) {
    constexpr int NF = 9 * NCH, Cin = 32 * NCH;
    __shared__ __attribute__((aligned(16))) bf16_t s_w[2 * NF * 64 * 8];   // [hi | lo][fragment = tap * NCH + chunk][lane][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    for (int i = tid; i < NF * 64; i += 256) {
        const int f = i >> 6, l = i & 63, co = l & 15, kq = l >> 4, tap = f / NCH, c = f % NCH;
        bf16x8 hi, lo;
        float wv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (co < Cout) {
            const float* wp = w + ((size_t)co * 9 + tap) * Cin + c * 32 + kq * 8;
            const float4 w0 = load4(wp), w1 = load4(wp + 4);
            wv[0] = w0.x; wv[1] = w0.y; wv[2] = w0.z; wv[3] = w0.w; wv[4] = w1.x; wv[5] = w1.y; wv[6] = w1.z; wv[7] = w1.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = wv[j];
            const bf16_t h = (bf16_t)v;
            hi[j] = h;
            lo[j] = (bf16_t)(v - (float)h);
        }
        *reinterpret_cast<bf16x8*>(&s_w[(size_t)i * 8]) = hi;
        *reinterpret_cast<bf16x8*>(&s_w[(size_t)(NF * 64 + i) * 8]) = lo;
    }
    __syncthreads();
    const int b = blockIdx.z, px = blockIdx.x * 16 + fr, ybase = blockIdx.y * 32 + wave * 8;
    constexpr bool X32 = sizeof(TX) == 4;
    constexpr int NR = X32 ? 2 : 1;                                  // fragments per piece: hi (+ lo)
    const TX* xb = x + (size_t)b * H * W * ldx + fq * 8;
    float bv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = (bias && e < Cout) ? bias[e] : 0.f;
    // rolling window over the input rows: fragments of rows y - 1, y, y + 1 are live, each new output row loads one row
    auto load_row = [&](int iy, bf16x8 (&dst)[3 * NCH * NR]) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = px + kx - 1;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const TX* src = xb + ((size_t)(ok ? iy : 0) * W + (ok ? ix : 0)) * ldx;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if constexpr (X32) {
                    const float4 a0 = load4(src + c * 32), a1 = load4(src + c * 32 + 4);
                    const float f[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    bf16x8 hi, lo;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float v = ok ? f[j] : 0.f;
                        const bf16_t h = (bf16_t)v;
                        hi[j] = h;
                        lo[j] = (bf16_t)(v - (float)h);
                    }
                    dst[(kx * NCH + c) * 2] = hi;
                    dst[(kx * NCH + c) * 2 + 1] = lo;
                } else {
                    bf16x8 v = *reinterpret_cast<const bf16x8*>(src + c * 32);
                    if (!ok) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = (bf16_t)0.f;
                    }
                    dst[kx * NCH + c] = v;
                }
            }
        }
    };
    bf16x8 rows[3][3 * NCH * NR];
    load_row(ybase - 1, rows[0]);
    load_row(ybase, rows[1]);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int y = ybase + r;
        if (y < Ho) {                                               // wave-uniform
            load_row(y + 1, rows[(r + 2) % 3]);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};   // two chains: hi weights; the small terms
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int t = 0; t < 3 * NCH; ++t) {
                    const int f = ky * 3 * NCH + t;                 // fragment = (ky * 3 + kx) * NCH + c
                    const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&s_w[(size_t)(f * 64 + lane) * 8]);
                    const bf16x8 wl = *reinterpret_cast<const bf16x8*>(&s_w[(size_t)((NF + f) * 64 + lane) * 8]);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, rows[(r + ky) % 3][t * NR], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, rows[(r + ky) % 3][t * NR], acc1, 0, 0, 0);
                    if constexpr (X32)
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, rows[(r + ky) % 3][t * NR + 1], acc1, 0, 0, 0);
                }
            if (fq == 0 && px < Wo) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < Cout) out[(((size_t)b * Cout + e) * Ho + y) * Wo + px] = (acc0[e] + acc1[e]) + bv[e];
            }
        }
    }
}

static int tail_conv3x3_impl(const void* xv, int x_dtype, int ldx, int B, int H, int W, int Cin, const float* w,
                             const float* bias, int Cout, float* out_nchw, int Ho, int Wo, bool split, uwir_stream_t stream) {
    UWIR_CHECK(xv && w && out_nchw, "tail: null pointer");
    UWIR_CHECK(x_dtype == UWIR_F32 || x_dtype == UWIR_BF16, "tail: bad input dtype");
    const float* x = (const float*)xv;
    UWIR_CHECK(B > 0 && H > 0 && W > 0 && Cin % 4 == 0 && Cout >= 1 && Cout <= HEAD_MAX_COUT && ldx >= Cin &&
                   ldx % 4 == 0 && Ho > 0 && Wo > 0 && Ho <= H && Wo <= W,
               "tail: bad shape (Cout<=%d, crop within the map)", HEAD_MAX_COUT);
    if ((x_dtype == UWIR_BF16 || split) && (Cin == 32 || Cin == 64) && ldx % 8 == 0 && (((uintptr_t)xv | (uintptr_t)w) & 15) == 0 &&
        B <= 65535 && ceil_div(Ho, 32) <= 65535) {
        const dim3 grid(ceil_div(Wo, 16), ceil_div(Ho, 32), B);
        if (x_dtype == UWIR_BF16 && Cin == 64)
            hipLaunchKernelGGL((tail_mfma_kernel<2, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)xv, ldx, H, W, w,
                               bias, Cout, out_nchw, Ho, Wo);
        else if (x_dtype == UWIR_BF16)
            hipLaunchKernelGGL((tail_mfma_kernel<1, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)xv, ldx, H, W, w,
                               bias, Cout, out_nchw, Ho, Wo);
        else if (Cin == 64)
            hipLaunchKernelGGL((tail_mfma_kernel<2, float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)xv, ldx, H, W, w,
                               bias, Cout, out_nchw, Ho, Wo);
        else
            hipLaunchKernelGGL((tail_mfma_kernel<1, float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)xv, ldx, H, W, w,
                               bias, Cout, out_nchw, Ho, Wo);
        UWIR_LAUNCH_CHECK("tail(mfma)");
        return UWIR_OK;
    }
    if (head_tiled_ok(B, Ho, Wo, Cin)) {
        if (x_dtype == UWIR_BF16)
            hipLaunchKernelGGL(head_tile_kernel<bf16_t>, dim3(ceil_div(Wo, HT_W), ceil_div(Ho, HT_H), B), dim3(256), 0,
                               (hipStream_t)stream, (const bf16_t*)xv, ldx, H, W, Cin, (const float*)nullptr, (const float*)nullptr,
                               w, bias, Cout, (const float*)nullptr, out_nchw, (float*)nullptr, Ho, Wo, 0,
                               (const uwir_window*)nullptr, 0, 0, 0);
        else
            hipLaunchKernelGGL(head_tile_kernel<float>, dim3(ceil_div(Wo, HT_W), ceil_div(Ho, HT_H), B), dim3(256), 0,
                               (hipStream_t)stream, x, ldx, H, W, Cin, (const float*)nullptr, (const float*)nullptr, w, bias,
                               Cout, (const float*)nullptr, out_nchw, (float*)nullptr, Ho, Wo, 0, (const uwir_window*)nullptr, 0, 0, 0);
        UWIR_LAUNCH_CHECK("tail(tile)");
        return UWIR_OK;
    }
    UWIR_CHECK(x_dtype == UWIR_F32, "tail: bf16 rows need Cin %% 16 == 0");
    size_t npix = (size_t)B * Ho * Wo;
    size_t blocks = ceil_div_sz(npix, 4);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(head_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, B, H, W, Cin,
                       (const float*)nullptr, (const float*)nullptr, w, bias, Cout, (const float*)nullptr, out_nchw,
                       (float*)nullptr, Ho, Wo, 0);
    UWIR_LAUNCH_CHECK("tail");
    return UWIR_OK;
}

extern "C" int uwir_tail_conv3x3(const void* xv, int x_dtype, int ldx, int B, int H, int W, int Cin, const float* w,
                                 const float* bias, int Cout, float* out_nchw, int Ho, int Wo, uwir_stream_t stream) {
    return tail_conv3x3_impl(xv, x_dtype, ldx, B, H, W, Cin, w, bias, Cout, out_nchw, Ho, Wo, false, stream);
}

extern "C" int uwir_tail_conv3x3_split(const float* x, int ldx, int B, int H, int W, int Cin, const float* w, const float* bias,
                                       int Cout, float* out_nchw, int Ho, int Wo, uwir_stream_t stream) {
    return tail_conv3x3_impl(x, UWIR_F32, ldx, B, H, W, Cin, w, bias, Cout, out_nchw, Ho, Wo, true, stream);
}

// ------------------------------------------------------------------------------------------
// fused IR-SDE reverse steps (20 B / element: x, mu, noise, eps in; x out), float4 lanes.
// Operation order follows the reference expressions so fp32 results agree to rounding.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float euler_one(float x, float mu, float nz, float e, float theta, float sigma,
                                           float sigma_bar, float dt, float sqrt_dt, float sw, bool has_eps) {
    float score = -nz / sigma_bar;                                       // sde_utils.py:186-187
    float drift = (theta * (mu - x) - sw * (sigma * sigma) * score) * dt;  // :177-181 (sw = 1 SDE, 0.5 ODE)
    float r = x - drift;
    if (has_eps) r -= sigma * (e * sqrt_dt);                              // :183-184
    return r;
}

__global__ void __launch_bounds__(256) sde_euler_kernel(const float* __restrict__ x, const float* __restrict__ mu,
                                                        const float* __restrict__ noise,
                                                        const float* __restrict__ eps, float* __restrict__ xo,
                                                        size_t n4, size_t numel, float theta, float sigma,
                                                        float sigma_bar, float dt, float sqrt_dt, float sw) {
    const bool he = eps != nullptr;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = load4(x + 4 * i), m = load4(mu + 4 * i), z = load4(noise + 4 * i);
        float4 e = he ? load4(eps + 4 * i) : make_float4(0, 0, 0, 0);
        float4 r;
        r.x = euler_one(a.x, m.x, z.x, e.x, theta, sigma, sigma_bar, dt, sqrt_dt, sw, he);
        r.y = euler_one(a.y, m.y, z.y, e.y, theta, sigma, sigma_bar, dt, sqrt_dt, sw, he);
        r.z = euler_one(a.z, m.z, z.z, e.z, theta, sigma, sigma_bar, dt, sqrt_dt, sw, he);
        r.w = euler_one(a.w, m.w, z.w, e.w, theta, sigma, sigma_bar, dt, sqrt_dt, sw, he);
        store4(xo + 4 * i, r);
    }
    // tail (numel % 4)
    size_t t = 4 * n4 + blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t < numel)
        xo[t] = euler_one(x[t], mu[t], noise[t], he ? eps[t] : 0.f, theta, sigma, sigma_bar, dt, sqrt_dt, sw, he);
}

__device__ __forceinline__ float post_one(float x, float mu, float nz, float e, float sigma_bar, float expA, float t1,
                                          float t2, float sd, bool has_eps) {
    float x0 = (x - mu - sigma_bar * nz) * expA + mu;          // sde_utils.py:245-247
    float mean = t1 * (x - mu) + t2 * (x0 - mu) + mu;         // :205-214
    return has_eps ? mean + sd * e : mean;                     // :227-231
}

__global__ void __launch_bounds__(256) sde_post_kernel(const float* __restrict__ x, const float* __restrict__ mu,
                                                       const float* __restrict__ noise, const float* __restrict__ eps,
                                                       float* __restrict__ xo, size_t n4, size_t numel,
                                                       float sigma_bar, float expA, float t1, float t2, float sd) {
    const bool he = eps != nullptr;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = load4(x + 4 * i), m = load4(mu + 4 * i), z = load4(noise + 4 * i);
        float4 e = he ? load4(eps + 4 * i) : make_float4(0, 0, 0, 0);
        float4 r;
        r.x = post_one(a.x, m.x, z.x, e.x, sigma_bar, expA, t1, t2, sd, he);
        r.y = post_one(a.y, m.y, z.y, e.y, sigma_bar, expA, t1, t2, sd, he);
        r.z = post_one(a.z, m.z, z.z, e.z, sigma_bar, expA, t1, t2, sd, he);
        r.w = post_one(a.w, m.w, z.w, e.w, sigma_bar, expA, t1, t2, sd, he);
        store4(xo + 4 * i, r);
    }
    size_t t = 4 * n4 + blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t < numel) xo[t] = post_one(x[t], mu[t], noise[t], he ? eps[t] : 0.f, sigma_bar, expA, t1, t2, sd, he);
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int uwir_sde_euler_step(const float* x, const float* mu, const float* noise, const float* eps, float* x_out,
                                   size_t numel, float theta, float sigma, float sigma_bar, float dt,
                                   float score_weight, uwir_stream_t stream) {
    UWIR_CHECK(x && mu && noise && x_out && numel > 0, "sde_euler: null pointer / empty");
    UWIR_CHECK(aligned16(x) && aligned16(mu) && aligned16(noise) && aligned16(x_out) && (!eps || aligned16(eps)),
               "sde_euler: pointers must be 16-byte aligned");
    size_t n4 = numel / 4;
    int grid = ew_grid(n4 ? n4 : 1);
    hipLaunchKernelGGL(sde_euler_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, mu, noise, eps, x_out, n4,
                       numel, theta, sigma, sigma_bar, dt, sqrtf(dt), score_weight);
    UWIR_LAUNCH_CHECK("sde_euler");
    return UWIR_OK;
}

extern "C" int uwir_sde_posterior_step(const float* x, const float* mu, const float* noise, const float* eps,
                                       float* x_out, size_t numel, float sigma_bar, float expA, float term1,
                                       float term2, float std, uwir_stream_t stream) {
    UWIR_CHECK(x && mu && noise && x_out && numel > 0, "sde_posterior: null pointer / empty");
    UWIR_CHECK(aligned16(x) && aligned16(mu) && aligned16(noise) && aligned16(x_out) && (!eps || aligned16(eps)),
               "sde_posterior: pointers must be 16-byte aligned");
    size_t n4 = numel / 4;
    int grid = ew_grid(n4 ? n4 : 1);
    hipLaunchKernelGGL(sde_post_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, mu, noise, eps, x_out, n4,
                       numel, sigma_bar, expA, term1, term2, std);
    UWIR_LAUNCH_CHECK("sde_posterior");
    return UWIR_OK;
}

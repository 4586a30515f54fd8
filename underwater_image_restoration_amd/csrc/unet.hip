// Kernels specific to the IR-SDE score network (ConditionalUNet): input assembly with reflect padding,
// LinearAttention (two passes over the tokens, HBM-bound), GEGLU gating.
#include "common.h"

// ------------------------------------------------------------------------------------------
// U/models/modules/DenoisingUNet_arch.py:123-127,111-116: x = cat([xt - cond, cond]) padded on the right /
// bottom by reflection to a multiple of 2^depth.  NCHW fp32 in -> NHWC fp32 out with the channel count
// rounded up to Cp (zero filled) so the 7x7 stem runs through the implicit-GEMM kernel.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) unet_input_kernel(const float* __restrict__ xt, const float* __restrict__ cond,
                                                         int Cin, int H, int W, int Hp, int Wp, int Cp,
                                                         float* __restrict__ out, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(i % Cp);
        size_t p = i / Cp;
        int x = (int)(p % Wp);
        size_t q = p / Wp;
        int y = (int)(q % Hp);
        int b = (int)(q / Hp);
        int sy = y < H ? y : 2 * (H - 1) - y;   // F.pad(..., 'reflect'): pad < size is checked on the host
        int sx = x < W ? x : 2 * (W - 1) - x;
        float v = 0.f;
        if (c < 2 * Cin) {
            int cc = c < Cin ? c : c - Cin;
            size_t o = (((size_t)b * Cin + cc) * H + sy) * W + sx;
            v = c < Cin ? xt[o] - cond[o] : cond[o];
        }
        out[i] = v;
    }
}

extern "C" int uwir_unet_input(const float* xt_nchw, const float* cond_nchw, int B, int Cin, int H, int W, int Hp, int Wp,
                               int Cp, float* out, uwir_stream_t stream) {
    UWIR_CHECK(xt_nchw && cond_nchw && out, "unet_input: null pointer");
    UWIR_CHECK(B > 0 && Cin > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W && Hp - H < H && Wp - W < W && Cp >= 2 * Cin,
               "unet_input: bad shape (reflect padding must be smaller than the image)");
    size_t total = (size_t)B * Hp * Wp * Cp;
    size_t g = ceil_div_sz(total, 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(unet_input_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, xt_nchw, cond_nchw, Cin, H,
                       W, Hp, Wp, Cp, out, total);
    UWIR_LAUNCH_CHECK("unet_input");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// LinearAttention, U/models/modules/module_util.py:157-185.  qkv rows [B*n, 3*heads*D] = [q | k | v],
// channel (head, d) at head*D + d, D = 32:
//   ctx[b,h,d,e] = sum_n softmax_n(k)[n,d] * v[n,e] / n ;  out[n,h,e] = sum_d ctx[d,e] * softmax_d(q[n,h,:])[d] * D^-0.5
// Three passes over the token rows, all HBM-bound:
//   (1) kmax:  column max of k over each chunk of rows                  -> pmax [B, chunks, heads*D]
//   (2) ctx:   per (chunk, head, b): E = exp(k - max_d), S[d] += E, A[d][e] += E * v[e]   -> partials
//       fin:   fixed-order sum over chunks, A / S / n * D^-0.5                              -> ctx [B, heads, D, D]
//   (3) apply: per (row, head): softmax over the D q-values in registers, 32x32 mat-vec against ctx (LDS broadcast)
// ------------------------------------------------------------------------------------------
#define LA_D 32
#define LA_CHUNK 512   // rows per block in passes 1 and 2

template <typename T>
__global__ void __launch_bounds__(256) la_kmax_kernel(const T* __restrict__ k, int ld, int n, int HD, int nchunks,
                                                      float* __restrict__ pmax) {
    // block (chunk, b); thread -> (column quad cq, row lane rl)
    const int Q = HD / 4, RL = 256 / Q;
    const int cq = threadIdx.x % Q, rl = threadIdx.x / Q;
    const int chunk = blockIdx.x, b = blockIdx.y;
    __shared__ float red[256][4];
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (rl < RL) {
        int r0 = chunk * LA_CHUNK, r1 = min(n, r0 + LA_CHUNK);
        for (int r = r0 + rl; r < r1; r += RL) {
            float4 v = load4(k + ((size_t)b * n + r) * ld + 4 * cq);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    *reinterpret_cast<float4*>(red[threadIdx.x]) = m;
    __syncthreads();
    if (rl == 0) {
        for (int l = 1; l < RL; ++l) {
            float4 o = *reinterpret_cast<float4*>(red[l * Q + cq]);
            m.x = fmaxf(m.x, o.x); m.y = fmaxf(m.y, o.y); m.z = fmaxf(m.z, o.z); m.w = fmaxf(m.w, o.w);
        }
        store4(pmax + ((size_t)b * nchunks + chunk) * HD + 4 * cq, m);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) la_ctx_kernel(const T* __restrict__ k, const T* __restrict__ v, int ld, int n,
                                                     int heads, int nchunks, const float* __restrict__ pmax,
                                                     float* __restrict__ part) {
    // block (chunk, head, b); thread -> (d = tid/8, e-quad = tid%8): A[d][4 e] + S[d]
    constexpr int D = LA_D, TR = 64;  // rows staged per LDS tile
    const int chunk = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const int HD = heads * D;
    const int tid = threadIdx.x, d = tid >> 3, eq = tid & 7;
    __shared__ float sE[TR][D + 1];
    __shared__ __attribute__((aligned(16))) float sV[TR][D];
    __shared__ float sMax[D];
    if (tid < D) {
        float m = -INFINITY;
        for (int c = 0; c < nchunks; ++c) m = fmaxf(m, pmax[((size_t)b * nchunks + c) * HD + head * D + tid]);
        sMax[tid] = m;
    }
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    float ssum = 0.f;
    const int r0 = chunk * LA_CHUNK, r1 = min(n, r0 + LA_CHUNK);
    for (int t0 = r0; t0 < r1; t0 += TR) {
        // stage: 64 rows x 32 k (-> exp) and 32 v; thread -> (row = tid/4, 8 columns)
        {
            int rr = tid >> 2, c8 = (tid & 3) * 8;
            int r = t0 + rr;
            bool ok = r < r1;
            size_t off = ((size_t)b * n + (ok ? r : r0)) * ld + head * D + c8;
            float4 k0 = load4(k + off), k1 = load4(k + off + 4), v0 = load4(v + off), v1 = load4(v + off + 4);
            float kk[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) sE[rr][c8 + j] = ok ? expf(kk[j] - sMax[c8 + j]) : 0.f;
            if (!ok) { v0 = make_float4(0, 0, 0, 0); v1 = v0; }
            *reinterpret_cast<float4*>(&sV[rr][c8]) = v0;
            *reinterpret_cast<float4*>(&sV[rr][c8 + 4]) = v1;
        }
        __syncthreads();
#pragma unroll 8
        for (int rr = 0; rr < TR; ++rr) {
            float e = sE[rr][d];
            float4 vv = *reinterpret_cast<const float4*>(&sV[rr][4 * eq]);
            acc.x = fmaf(e, vv.x, acc.x); acc.y = fmaf(e, vv.y, acc.y);
            acc.z = fmaf(e, vv.z, acc.z); acc.w = fmaf(e, vv.w, acc.w);
            ssum += e;
        }
        __syncthreads();
    }
    float* o = part + (((size_t)b * heads + head) * nchunks + chunk) * (D * D + D);
    store4(o + d * D + 4 * eq, acc);
    if (eq == 0) o[D * D + d] = ssum;
}

__global__ void __launch_bounds__(256) la_fin_kernel(const float* __restrict__ part, int nchunks, float inv_n_scale,
                                                     float* __restrict__ ctx, int total) {
    // one thread per ctx element (bh, d, e); fixed-order chunk sum in fp64 -> deterministic
    constexpr int D = LA_D;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int e = i % D, d = (i / D) % D, bh = i / (D * D);
    double a = 0, s = 0;
    for (int c = 0; c < nchunks; ++c) {
        const float* p = part + ((size_t)bh * nchunks + c) * (D * D + D);
        a += (double)p[d * D + e];
        s += (double)p[D * D + d];
    }
    ctx[i] = (float)(a / s * (double)inv_n_scale);
}

template <typename T, typename TO>
__global__ void __launch_bounds__(256) la_apply_kernel(const T* __restrict__ q, int ld, int n, int heads,
                                                       const float* __restrict__ ctx, TO* __restrict__ out, int ldo,
                                                       size_t rows_total) {
    // block: 256 rows of one (b, head); ctx[b,head] in LDS; thread = one row
    constexpr int D = LA_D;
    const int head = blockIdx.y, b = blockIdx.z;
    __shared__ __attribute__((aligned(16))) float sC[D][D];
    for (int i = threadIdx.x; i < D * D; i += 256) (&sC[0][0])[i] = ctx[((size_t)b * heads + head) * D * D + i];
    __syncthreads();
    int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    size_t row = (size_t)b * n + r;
    const T* qr = q + row * ld + head * D;
    float qv[D];
#pragma unroll
    for (int j = 0; j < D; j += 4) {
        float4 t = load4(qr + j);
        qv[j] = t.x; qv[j + 1] = t.y; qv[j + 2] = t.z; qv[j + 3] = t.w;
    }
    float m = qv[0];
#pragma unroll
    for (int j = 1; j < D; ++j) m = fmaxf(m, qv[j]);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) { qv[j] = expf(qv[j] - m); s += qv[j]; }
    const float inv = 1.f / s;
    TO* orow = out + row * ldo + head * D;
#pragma unroll
    for (int e4 = 0; e4 < D; e4 += 4) {
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
            float4 c4 = *reinterpret_cast<const float4*>(&sC[dd][e4]);
            acc.x = fmaf(c4.x, qv[dd], acc.x); acc.y = fmaf(c4.y, qv[dd], acc.y);
            acc.z = fmaf(c4.z, qv[dd], acc.z); acc.w = fmaf(c4.w, qv[dd], acc.w);
        }
        store4(orow + e4, make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv));
    }
}

// ------------------------------------------------------------------------------------------
// bf16 form of passes (2) and (3) on the matrix cores.  The VALU kernels above spend 1024 FMAs per (row, head) in each
// pass (2.1 GFMA per pass at B = 8, 256^2: 3x the time the HBM traffic needs).  Both contractions are 32-wide GEMMs:
//   ctx:   A = E^T (d x rows), B = V (rows x e): the contracted index is the ROW, i.e. the slow index of the row-major
//          inputs, so the 64-row tiles are transposed on their way into LDS ([d][row] / [e][row]) and every wave feeds
//          16 rows per v_mfma_f32_32x32x16_bf16; the row sums S[d] ride along as A x ones.
//   apply: A = softmax_d(q) (rows x d) is row-major as stored (16-byte fragments straight from global), B = ctx from an
//          LDS copy kept transposed; the softmax normalisation is folded into A.
// E, V, softmax(q) and ctx enter the MFMAs rounded to bf16 (fp32 accumulation); the fp32 path keeps the VALU kernels.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) la_ctx_mfma_kernel(const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                          int ld, int n, int heads, int nchunks,
                                                          const float* __restrict__ pmax, float* __restrict__ part) {
    constexpr int D = LA_D, TR = 64, STR = TR + 8;
    const int chunk = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const int HD = heads * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane & 31, kg = lane >> 5;
    __shared__ __attribute__((aligned(16))) bf16_t sEt[D * STR];
    __shared__ __attribute__((aligned(16))) bf16_t sVt[D * STR];
    __shared__ float sMax[D];
    __shared__ float red[4][D][D + 1];
    __shared__ float redS[4][D];
    if (tid < D) {
        float m = -INFINITY;
        for (int c = 0; c < nchunks; ++c) m = fmaxf(m, pmax[((size_t)b * nchunks + c) * HD + head * D + tid]);
        sMax[tid] = m;
    }
    __syncthreads();
    f32x16 acc, accS;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accS[r] = 0.f; }
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.f;
    const int r0 = chunk * LA_CHUNK, r1 = min(n, r0 + LA_CHUNK);
    const int rr = tid >> 2, c8 = (tid & 3) * 8;
    for (int t0 = r0; t0 < r1; t0 += TR) {
        {   // stage 64 rows: E = exp(k - max_d) and V, transposed to [column][row]
            const int r = t0 + rr;
            const bool ok = r < r1;
            const size_t off = ((size_t)b * n + (ok ? r : r0)) * ld + head * D + c8;
            const bf16x8 kk = *reinterpret_cast<const bf16x8*>(k + off);
            const bf16x8 vv = *reinterpret_cast<const bf16x8*>(v + off);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float e = ok ? __builtin_amdgcn_exp2f(((float)kk[j] - sMax[c8 + j]) * 1.4426950408889634f) : 0.f;
                sEt[(c8 + j) * STR + rr] = (bf16_t)e;
                sVt[(c8 + j) * STR + rr] = ok ? vv[j] : (bf16_t)0.f;
            }
        }
        __syncthreads();
        {   // wave w contracts rows 16w .. 16w+15 of the tile
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(&sEt[lq * STR + 16 * wave + 8 * kg]);
            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(&sVt[lq * STR + 16 * wave + 8 * kg]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bv, acc, 0, 0, 0);
            accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ones, accS, 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {   // accumulator: column e = lq, rows d = (r&3) + 8(r>>2) + 4kg
        const int d = (r & 3) + 8 * (r >> 2) + 4 * kg;
        red[wave][d][lq] = acc[r];
        if (lq == 0) redS[wave][d] = accS[r];
    }
    __syncthreads();
    const int d = tid >> 3, eq = tid & 7;
    float* o = part + (((size_t)b * heads + head) * nchunks + chunk) * (D * D + D);
    float4 t;
    t.x = red[0][d][4 * eq] + red[1][d][4 * eq] + red[2][d][4 * eq] + red[3][d][4 * eq];
    t.y = red[0][d][4 * eq + 1] + red[1][d][4 * eq + 1] + red[2][d][4 * eq + 1] + red[3][d][4 * eq + 1];
    t.z = red[0][d][4 * eq + 2] + red[1][d][4 * eq + 2] + red[2][d][4 * eq + 2] + red[3][d][4 * eq + 2];
    t.w = red[0][d][4 * eq + 3] + red[1][d][4 * eq + 3] + red[2][d][4 * eq + 3] + red[3][d][4 * eq + 3];
    store4(o + d * D + 4 * eq, t);
    if (eq == 0) o[D * D + d] = redS[0][d] + redS[1][d] + redS[2][d] + redS[3][d];
}

template <typename TO>
__global__ void __launch_bounds__(256) la_apply_mfma_kernel(const bf16_t* __restrict__ q, int ld, int n, int heads,
                                                            const float* __restrict__ ctx, TO* __restrict__ out, int ldo) {
    constexpr int D = LA_D, CS = D + 8;
    const int head = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane & 31, kg = lane >> 5;
    __shared__ __attribute__((aligned(16))) bf16_t sCt[D * CS];   // ctx transposed: [e][d]
    for (int i = tid; i < D * D; i += 256) sCt[(i % D) * CS + i / D] = (bf16_t)ctx[((size_t)b * heads + head) * D * D + i];
    __syncthreads();
    bf16x8 cb[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) cb[st] = *reinterpret_cast<const bf16x8*>(&sCt[lq * CS + 16 * st + 8 * kg]);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int base = blockIdx.x * 256 + wave * 64 + g * 32;
        if (base >= n) break;   // wave-uniform
        const int r = min(base + lq, n - 1);
        const bf16_t* qr = q + ((size_t)b * n + r) * ld + head * D + 8 * kg;
        float f[16];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(qr + 16 * st);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[8 * st + j] = (float)t[j];
        }
        float m = f[0];
#pragma unroll
        for (int j = 1; j < 16; ++j) m = fmaxf(m, f[j]);
        {   // the other 16 values of the row sit in lane ^ 32
            unsigned u = __float_as_uint(m);
            auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        float ssum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) { f[j] = __builtin_amdgcn_exp2f((f[j] - m) * 1.4426950408889634f); ssum += f[j]; }
        {
            unsigned u = __float_as_uint(ssum);
            auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            ssum = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        const float inv = 1.f / ssum;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8 pa;
#pragma unroll
            for (int j = 0; j < 8; ++j) pa[j] = (bf16_t)(f[8 * st + j] * inv);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, cb[st], acc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {   // accumulator: column e = lq, rows = (i&3) + 8(i>>2) + 4kg of this 32-row group
            const int row = base + (i & 3) + 8 * (i >> 2) + 4 * kg;
            if (row < n) out[((size_t)b * n + row) * ldo + head * D + lq] = (TO)acc[i];
        }
    }
}

extern "C" size_t uwir_linattn_workspace_bytes(int B, int n, int heads) {
    size_t nchunks = ceil_div(n, LA_CHUNK);
    size_t HD = (size_t)heads * LA_D;
    return ((size_t)B * nchunks * HD + (size_t)B * heads * nchunks * (LA_D * LA_D + LA_D) +
            (size_t)B * heads * LA_D * LA_D) * sizeof(float);
}

template <typename T>
static int linattn_launch(const void* qkv, int ld, int B, int n, int heads, void* out, int out_dtype, int ldo, float* ws,
                          hipStream_t s) {
    const int D = LA_D, HD = heads * D;
    const int nchunks = ceil_div(n, LA_CHUNK);
    const T* q = (const T*)qkv;
    const T* k = q + HD;
    const T* v = q + 2 * HD;
    float* pmax = ws;
    float* part = pmax + (size_t)B * nchunks * HD;
    float* ctx = part + (size_t)B * heads * nchunks * (D * D + D);
    hipLaunchKernelGGL(la_kmax_kernel<T>, dim3(nchunks, B), dim3(256), 0, s, k, ld, n, HD, nchunks, pmax);
    UWIR_LAUNCH_CHECK("linattn(kmax)");
    constexpr bool BF = sizeof(T) == 2;
    const bool mfma = BF && ld % 8 == 0;   // 16-byte fragments
    if constexpr (BF) {
        if (mfma) hipLaunchKernelGGL(la_ctx_mfma_kernel, dim3(nchunks, heads, B), dim3(256), 0, s, (const bf16_t*)k,
                                     (const bf16_t*)v, ld, n, heads, nchunks, pmax, part);
    }
    if (!mfma) hipLaunchKernelGGL(la_ctx_kernel<T>, dim3(nchunks, heads, B), dim3(256), 0, s, k, v, ld, n, heads, nchunks, pmax, part);
    UWIR_LAUNCH_CHECK("linattn(ctx)");
    const int total = B * heads * D * D;
    // q * D^-0.5 and v / n (module_util.py:176-177) are folded into the context
    const float inv_n_scale = (float)(1.0 / ((double)n * sqrt((double)D)));
    hipLaunchKernelGGL(la_fin_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, part, nchunks, inv_n_scale, ctx, total);
    UWIR_LAUNCH_CHECK("linattn(fin)");
    dim3 grid(ceil_div(n, 256), heads, B);
    if constexpr (BF) {
        if (mfma) {
            if (out_dtype == UWIR_F32)
                hipLaunchKernelGGL(la_apply_mfma_kernel<float>, grid, dim3(256), 0, s, (const bf16_t*)q, ld, n, heads, ctx,
                                   (float*)out, ldo);
            else
                hipLaunchKernelGGL(la_apply_mfma_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)q, ld, n, heads, ctx,
                                   (bf16_t*)out, ldo);
            UWIR_LAUNCH_CHECK("linattn(apply, mfma)");
            return UWIR_OK;
        }
    }
    if (out_dtype == UWIR_F32)
        hipLaunchKernelGGL((la_apply_kernel<T, float>), grid, dim3(256), 0, s, q, ld, n, heads, ctx, (float*)out, ldo,
                           (size_t)B * n);
    else
        hipLaunchKernelGGL((la_apply_kernel<T, bf16_t>), grid, dim3(256), 0, s, q, ld, n, heads, ctx, (bf16_t*)out, ldo,
                           (size_t)B * n);
    UWIR_LAUNCH_CHECK("linattn(apply)");
    return UWIR_OK;
}

extern "C" int uwir_linear_attention(const void* qkv, int dtype, int ld, int B, int n, int heads, void* out, int out_dtype,
                                     int ldo, void* workspace, size_t workspace_bytes, uwir_stream_t stream) {
    UWIR_CHECK(qkv && out && workspace, "linear_attention: null pointer");
    UWIR_CHECK(B > 0 && B <= 65535 && n > 0 && heads > 0 && heads <= 8, "linear_attention: bad shape (heads <= 8)");
    UWIR_CHECK(ld >= 3 * heads * LA_D && ld % 4 == 0 && ldo >= heads * LA_D && ldo % 4 == 0,
               "linear_attention: bad leading dimensions");
    UWIR_CHECK((dtype == UWIR_F32 || dtype == UWIR_BF16) && (out_dtype == UWIR_F32 || out_dtype == UWIR_BF16),
               "linear_attention: bad dtype");
    UWIR_CHECK(((uintptr_t)qkv & 15) == 0, "linear_attention: qkv must be 16-byte aligned");
    UWIR_CHECK(workspace_bytes >= uwir_linattn_workspace_bytes(B, n, heads), "linear_attention: workspace too small");
    if (dtype == UWIR_F32)
        return linattn_launch<float>(qkv, ld, B, n, heads, out, out_dtype, ldo, (float*)workspace, (hipStream_t)stream);
    return linattn_launch<bf16_t>(qkv, ld, B, n, heads, out, out_dtype, ldo, (float*)workspace, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// GEGLU gate, U/models/modules/attention.py:37-44: h [M, 2F] = [value | gate] -> out [M, F] = value * gelu(gate)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) geglu_kernel(const T* __restrict__ h, size_t ldh, int F4, T* __restrict__ out,
                                                    size_t ldo, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int c4 = (int)(i % F4);
        size_t m = i / F4;
        float4 u = load4(h + m * ldh + 4 * c4), g = load4(h + m * ldh + 4 * (size_t)F4 + 4 * c4);
        float4 o = make_float4(u.x * act_apply(g.x, UWIR_ACT_GELU), u.y * act_apply(g.y, UWIR_ACT_GELU),
                               u.z * act_apply(g.z, UWIR_ACT_GELU), u.w * act_apply(g.w, UWIR_ACT_GELU));
        store4(out + m * ldo + 4 * c4, o);
    }
}

extern "C" int uwir_geglu(const void* h, int dtype, size_t ldh, int M, int F, void* out, size_t ldo, uwir_stream_t stream) {
    UWIR_CHECK(h && out, "geglu: null pointer");
    UWIR_CHECK(M > 0 && F > 0 && F % 4 == 0 && ldh >= (size_t)2 * F && ldh % 4 == 0 && ldo >= (size_t)F && ldo % 4 == 0,
               "geglu: bad shape");
    UWIR_CHECK(dtype == UWIR_F32 || dtype == UWIR_BF16, "geglu: bad dtype");
    size_t total = (size_t)M * (F / 4);
    size_t g = ceil_div_sz(total, 256);
    if (g > 8192) g = 8192;
    if (dtype == UWIR_F32)
        hipLaunchKernelGGL(geglu_kernel<float>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float*)h, ldh,
                           F / 4, (float*)out, ldo, total);
    else
        hipLaunchKernelGGL(geglu_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h,
                           ldh, F / 4, (bf16_t*)out, ldo, total);
    UWIR_LAUNCH_CHECK("geglu");
    return UWIR_OK;
}

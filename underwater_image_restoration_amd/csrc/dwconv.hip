// Multi-scale depthwise convolution 3x3 + 5x5 + 7x7 (zero padding) fused with the channel concat
// that feeds the 1x1 merge conv (S/wtb.py:53-62,91; S/sffb.py:21-31,46).
//
// NHWC.  A block owns an 8x16 pixel tile x 32-channel slab: the (8+6)x(16+6) halo tile is staged
// once in LDS as fp32 (input affine = normalise-on-load applied here; out-of-image pixels are 0,
// i.e. padding happens AFTER the norm like the reference), the 83 taps x 32 channels of weights
// sit beside it.  A thread produces 4 consecutive x pixels of one channel quad for all three
// kernel sizes from a sliding 10-wide register row, so every staged value is read from LDS once
// per row and feeds up to 15 taps x 4 pixels.
#include "common.h"

#define DW_TH 8
#define DW_TW 16
#define DW_CS 32                       // channels per block
#define DW_HALO 3
#define DW_LW (DW_TW + 2 * DW_HALO)    // 22
#define DW_LH (DW_TH + 2 * DW_HALO)    // 14
#define DW_ROWSTRIDE (DW_LW * DW_CS + 32)  // floats; +32 makes consecutive rows hit opposite bank halves
#define DW_NTAPS (9 + 25 + 49)

template <typename TIN, typename TOUT>
__global__ void __launch_bounds__(256) dwconv_ms_kernel(const TIN* __restrict__ x, int ldx, int H, int W, int C,
                                                        const float* __restrict__ in_scale,
                                                        const float* __restrict__ in_shift,
                                                        const float* __restrict__ w3, const float* __restrict__ w5,
                                                        const float* __restrict__ w7, const float* __restrict__ b3,
                                                        const float* __restrict__ b5, const float* __restrict__ b7,
                                                        TOUT* __restrict__ out, int ldo, int tiles_x) {
    __shared__ __attribute__((aligned(16))) float s_in[DW_LH * DW_ROWSTRIDE];
    __shared__ __attribute__((aligned(16))) float s_w[DW_NTAPS * DW_CS];

    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * DW_CS;
    const int ty0 = (blockIdx.x / tiles_x) * DW_TH, tx0 = (blockIdx.x % tiles_x) * DW_TW;
    const int nq = min(DW_CS, C - c0) / 4;  // valid quads in this slab

    // ---- stage weights: s_w[tap][32], taps ordered [3x3 | 5x5 | 7x7]
    for (int i = tid; i < DW_NTAPS * (DW_CS / 4); i += 256) {
        int tap = i / (DW_CS / 4), q = i % (DW_CS / 4);
        float4 v = make_float4(0, 0, 0, 0);
        if (q < nq) {
            const float* src = tap < 9 ? w3 + (size_t)tap * C : (tap < 34 ? w5 + (size_t)(tap - 9) * C : w7 + (size_t)(tap - 34) * C);
            v = load4(src + c0 + 4 * q);
        }
        *reinterpret_cast<float4*>(&s_w[tap * DW_CS + 4 * q]) = v;
    }
    // ---- stage halo tile
    for (int i = tid; i < DW_LH * DW_LW * (DW_CS / 4); i += 256) {
        int q = i % (DW_CS / 4);
        int p = i / (DW_CS / 4);
        int lx = p % DW_LW, ly = p / DW_LW;
        int iy = ty0 + ly - DW_HALO, ix = tx0 + lx - DW_HALO;
        float4 v = make_float4(0, 0, 0, 0);
        if (q < nq && iy >= 0 && iy < H && ix >= 0 && ix < W) {
            v = load4(x + (((size_t)b * H + iy) * W + ix) * ldx + c0 + 4 * q);
            if (in_scale) {
                float4 sc = load4(in_scale + (size_t)b * C + c0 + 4 * q);
                float4 sh = load4(in_shift + (size_t)b * C + c0 + 4 * q);
                v.x = fmaf(v.x, sc.x, sh.x);
                v.y = fmaf(v.y, sc.y, sh.y);
                v.z = fmaf(v.z, sc.z, sh.z);
                v.w = fmaf(v.w, sc.w, sh.w);
            }
        }
        *reinterpret_cast<float4*>(&s_in[ly * DW_ROWSTRIDE + lx * DW_CS + 4 * q]) = v;
    }
    __syncthreads();

    const int q = tid % 8, y = (tid / 8) % 8, xg = tid / 64;  // 8 quads x 8 rows x 4 x-groups
    if (q >= nq) return;
    float4 a3[4], a5[4], a7[4];
    {
        float4 z = make_float4(0, 0, 0, 0);
        float4 v3 = b3 ? load4(b3 + c0 + 4 * q) : z, v5 = b5 ? load4(b5 + c0 + 4 * q) : z,
               v7 = b7 ? load4(b7 + c0 + 4 * q) : z;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a3[j] = v3; a5[j] = v5; a7[j] = v7; }
    }
#define FMA4(acc, wv, iv)                 \
    acc.x = fmaf(wv.x, iv.x, acc.x);      \
    acc.y = fmaf(wv.y, iv.y, acc.y);      \
    acc.z = fmaf(wv.z, iv.z, acc.z);      \
    acc.w = fmaf(wv.w, iv.w, acc.w);
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
        float4 row[10];
        const float* rp = &s_in[(y + ky) * DW_ROWSTRIDE + (xg * 4) * DW_CS + 4 * q];
#pragma unroll
        for (int i = 0; i < 10; ++i) row[i] = *reinterpret_cast<const float4*>(rp + i * DW_CS);
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            float4 wv = *reinterpret_cast<const float4*>(&s_w[(34 + ky * 7 + kx) * DW_CS + 4 * q]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { FMA4(a7[j], wv, row[j + kx]) }
        }
        if (ky >= 1 && ky <= 5) {
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                float4 wv = *reinterpret_cast<const float4*>(&s_w[(9 + (ky - 1) * 5 + kx) * DW_CS + 4 * q]);
#pragma unroll
                for (int j = 0; j < 4; ++j) { FMA4(a5[j], wv, row[j + kx + 1]) }
            }
        }
        if (ky >= 2 && ky <= 4) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float4 wv = *reinterpret_cast<const float4*>(&s_w[((ky - 2) * 3 + kx) * DW_CS + 4 * q]);
#pragma unroll
                for (int j = 0; j < 4; ++j) { FMA4(a3[j], wv, row[j + kx + 2]) }
            }
        }
    }
#undef FMA4
    const int oy = ty0 + y;
    if (oy >= H) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int ox = tx0 + xg * 4 + j;
        if (ox < W) {
            TOUT* o = out + (((size_t)b * H + oy) * W + ox) * ldo + c0 + 4 * q;
            store4(o, a3[j]);
            store4(o + C, a5[j]);
            store4(o + 2 * C, a7[j]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Second form (C % 32 == 0): the kernel above reads 83 weight quads + 70 input quads from LDS per thread and is bound
// by LDS bandwidth, not by its FMAs.  Here a WAVE owns one channel quad, so the 83 x 4 weights of the quad are
// wave-uniform and arrive through scalar loads (SGPR operands of the packed FMAs): LDS carries the input rows only.
// Block = 8 waves = 32 channels over a 16 x 16 pixel tile; lane = (row y, 4-pixel group xg); the 22 x 22 halo tile sits
// in LDS as one fp32 plane per quad with an odd row pitch (23 quads), which makes the 16 row-lanes of a 16-byte LDS
// read hit 16 distinct bank groups.  A lane's results are 8-byte pieces scattered over 12 (pixel, scale) rows, so bf16
// outputs are transposed through LDS (the halo tile's storage, reused) and leave as 64-byte runs per (pixel, scale).
// ------------------------------------------------------------------------------------------
#define D2_T 16
#define D2_L (D2_T + 2 * DW_HALO)   // 22
#define D2_PITCH 23                 // quads per LDS row (odd)
#define D2_PLANE (D2_L * D2_PITCH)  // quads per channel-quad plane
#define D2_Q 8                      // quads (waves) per block
#define D2_OPS 104                  // bf16 elements per staged output pixel: 3 x 32 + 8 pad (208 B: 16-byte aligned, and
                                    // 16 row-lanes x 208 B land on 16 distinct bank groups)

template <typename TIN, typename TOUT>
__global__ void __launch_bounds__(512) dwconv_ms32_kernel(const TIN* __restrict__ x, int ldx, int H, int W, int C,
                                                          const float* __restrict__ in_scale,
                                                          const float* __restrict__ in_shift,
                                                          const float* __restrict__ w3, const float* __restrict__ w5,
                                                          const float* __restrict__ w7, const float* __restrict__ b3,
                                                          const float* __restrict__ b5, const float* __restrict__ b7,
                                                          TOUT* __restrict__ out, int ldo, int tiles_x) {
    __shared__ __attribute__((aligned(16))) float4 s_in[D2_Q * D2_PLANE];
    static_assert(D2_T * D2_T * D2_OPS * 2 <= D2_Q * D2_PLANE * 16, "the output staging must fit in the halo tile");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);   // this wave's channel quad (scalar)
    const int b = blockIdx.z;
    const int c0 = blockIdx.x * (4 * D2_Q);
    const int ty0 = (blockIdx.y / tiles_x) * D2_T, tx0 = (blockIdx.y % tiles_x) * D2_T;

    // ---- stage the halo tile: item = (pixel, quad), the quads of a pixel on adjacent lanes (64 contiguous bytes of bf16).
    // All of a thread's loads are issued before the first LDS store (a rolled load -> store loop exposes one HBM
    // round trip per item: 8 in a row per block, which made this kernel latency-bound at 4x its FMA time); the quad
    // index of an item is tid % 8 for every item, so the normalise-on-load affine is loaded once.
    {
        constexpr int NITEM = D2_L * D2_L * D2_Q, NIT = (NITEM + 511) / 512;
        const int q = tid % D2_Q;
        float4 raw[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pix = (tid + 512 * it) / D2_Q;
            const int lx = pix % D2_L, ly = pix / D2_L;
            const int iy = ty0 + ly - DW_HALO, ix = tx0 + lx - DW_HALO;
            const bool ok = pix < D2_L * D2_L && iy >= 0 && iy < H && ix >= 0 && ix < W;
            raw[it] = make_float4(0, 0, 0, 0);
            if (ok) raw[it] = load4(x + (((size_t)b * H + iy) * W + ix) * ldx + c0 + 4 * q);
        }
        float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
        if (in_scale) {
            sc = load4(in_scale + (size_t)b * C + c0 + 4 * q);
            sh = load4(in_shift + (size_t)b * C + c0 + 4 * q);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pix = (tid + 512 * it) / D2_Q;
            if (pix >= D2_L * D2_L) continue;
            const int lx = pix % D2_L, ly = pix / D2_L;
            const int iy = ty0 + ly - DW_HALO, ix = tx0 + lx - DW_HALO;
            float4 v = raw[it];
            if (in_scale && iy >= 0 && iy < H && ix >= 0 && ix < W) {   // padding stays zero: it pads the NORMALISED image
                v.x = fmaf(v.x, sc.x, sh.x);
                v.y = fmaf(v.y, sc.y, sh.y);
                v.z = fmaf(v.z, sc.z, sh.z);
                v.w = fmaf(v.w, sc.w, sh.w);
            }
            s_in[q * D2_PLANE + ly * D2_PITCH + lx] = v;
        }
    }
    __syncthreads();

    const int y = lane & 15, xg = lane >> 4;
    const int cq = c0 + 4 * wq;                     // first channel of this wave's quad (scalar)
    const float* __restrict__ w3q = w3 + cq;
    const float* __restrict__ w5q = w5 + cq;
    const float* __restrict__ w7q = w7 + cq;
    float4 a3[4], a5[4], a7[4];
    {
        const float4 z = make_float4(0, 0, 0, 0);
        const float4 v3 = b3 ? load4(b3 + cq) : z, v5 = b5 ? load4(b5 + cq) : z, v7 = b7 ? load4(b7 + cq) : z;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a3[j] = v3; a5[j] = v5; a7[j] = v7; }
    }
#define FMA4(acc, wv, iv)                 \
    acc.x = fmaf(wv.x, iv.x, acc.x);      \
    acc.y = fmaf(wv.y, iv.y, acc.y);      \
    acc.z = fmaf(wv.z, iv.z, acc.z);      \
    acc.w = fmaf(wv.w, iv.w, acc.w);
    const float4* plane = s_in + wq * D2_PLANE + y * D2_PITCH + xg * 4;
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
        float4 row[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) row[i] = plane[ky * D2_PITCH + i];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            const float4 wv = load4(w7q + (size_t)(ky * 7 + kx) * C);   // wave-uniform address: scalar load
#pragma unroll
            for (int j = 0; j < 4; ++j) { FMA4(a7[j], wv, row[j + kx]) }
        }
        if (ky >= 1 && ky <= 5) {
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                const float4 wv = load4(w5q + (size_t)((ky - 1) * 5 + kx) * C);
#pragma unroll
                for (int j = 0; j < 4; ++j) { FMA4(a5[j], wv, row[j + kx + 1]) }
            }
        }
        if (ky >= 2 && ky <= 4) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 wv = load4(w3q + (size_t)((ky - 2) * 3 + kx) * C);
#pragma unroll
                for (int j = 0; j < 4; ++j) { FMA4(a3[j], wv, row[j + kx + 2]) }
            }
        }
    }
#undef FMA4
    if constexpr (sizeof(TOUT) == 2) {
        __syncthreads();   // every wave is done reading the halo tile: its storage becomes the output staging area
        bf16_t* s_out = reinterpret_cast<bf16_t*>(s_in);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16_t* o = s_out + ((xg * 4 + j) * D2_T + y) * D2_OPS + 4 * wq;   // slot = x * 16 + y: row-lanes adjacent
            store4(o, a3[j]);
            store4(o + 32, a5[j]);
            store4(o + 64, a7[j]);
        }
        __syncthreads();
        for (int i = tid; i < D2_T * D2_T * 12; i += 512) {   // 16-byte pieces: 4 per (pixel, scale)
            const int piece = i & 3, sc = (i >> 2) % 3, pxl = i / 12;
            const int py = pxl / D2_T, px = pxl % D2_T;
            const int oy = ty0 + py, ox = tx0 + px;
            if (oy < H && ox < W) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(s_out + (px * D2_T + py) * D2_OPS + sc * 32 + piece * 8);
                *reinterpret_cast<bf16x8*>(out + (((size_t)b * H + oy) * W + ox) * ldo + (size_t)sc * C + c0 + piece * 8) = v;
            }
        }
    } else {
        const int oy = ty0 + y;
        if (oy >= H) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ox = tx0 + xg * 4 + j;
            if (ox < W) {
                TOUT* o = out + (((size_t)b * H + oy) * W + ox) * ldo + cq;
                store4(o, a3[j]);
                store4(o + C, a5[j]);
                store4(o + 2 * C, a7[j]);
            }
        }
    }
}

template <typename TIN, typename TOUT>
static int launch_dw(const void* x, int ldx, int B, int H, int W, int C, const float* in_scale, const float* in_shift,
                     const float* w3, const float* w5, const float* w7, const float* b3, const float* b5,
                     const float* b7, void* out, int ldo, hipStream_t stream) {
    // bf16 outputs with 16-byte-aligned rows take the wave-per-quad kernel
    if (C % 32 == 0 && sizeof(TOUT) == 2 && ldo % 8 == 0 && ((uintptr_t)out & 15) == 0 &&
        (long long)ceil_div(W, D2_T) * ceil_div(H, D2_T) <= 65535) {
        const int tx = ceil_div(W, D2_T), ty = ceil_div(H, D2_T);
        hipLaunchKernelGGL((dwconv_ms32_kernel<TIN, TOUT>), dim3(C / 32, tx * ty, B), dim3(512), 0, stream, (const TIN*)x, ldx,
                           H, W, C, in_scale, in_shift, w3, w5, w7, b3, b5, b7, (TOUT*)out, ldo, tx);
        UWIR_LAUNCH_CHECK("dwconv_ms(32)");
        return UWIR_OK;
    }
    int tiles_x = ceil_div(W, DW_TW), tiles_y = ceil_div(H, DW_TH);
    dim3 grid(tiles_x * tiles_y, ceil_div(C, DW_CS), B);
    hipLaunchKernelGGL((dwconv_ms_kernel<TIN, TOUT>), grid, dim3(256), 0, stream, (const TIN*)x, ldx, H, W, C, in_scale,
                       in_shift, w3, w5, w7, b3, b5, b7, (TOUT*)out, ldo, tiles_x);
    UWIR_LAUNCH_CHECK("dwconv_ms");
    return UWIR_OK;
}

extern "C" int uwir_dwconv_ms(const void* x, int x_dtype, int ldx, int B, int H, int W, int C, const float* in_scale,
                              const float* in_shift, const float* w3, const float* w5, const float* w7,
                              const float* b3, const float* b5, const float* b7, void* out, int out_dtype, int ldo,
                              uwir_stream_t stream) {
    UWIR_CHECK(x && out && w3 && w5 && w7, "dwconv_ms: null pointer");
    UWIR_CHECK(B > 0 && B <= 65535 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "dwconv_ms: bad shape");
    UWIR_CHECK(ldx >= C && ldx % 4 == 0 && ldo >= 3 * C && ldo % 4 == 0, "dwconv_ms: bad strides");
    UWIR_CHECK((in_scale == nullptr) == (in_shift == nullptr), "dwconv_ms: in_scale/in_shift must come together");
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == UWIR_F32 && out_dtype == UWIR_F32)
        return launch_dw<float, float>(x, ldx, B, H, W, C, in_scale, in_shift, w3, w5, w7, b3, b5, b7, out, ldo, s);
    if (x_dtype == UWIR_F32 && out_dtype == UWIR_BF16)
        return launch_dw<float, bf16_t>(x, ldx, B, H, W, C, in_scale, in_shift, w3, w5, w7, b3, b5, b7, out, ldo, s);
    if (x_dtype == UWIR_BF16 && out_dtype == UWIR_BF16)
        return launch_dw<bf16_t, bf16_t>(x, ldx, B, H, W, C, in_scale, in_shift, w3, w5, w7, b3, b5, b7, out, ldo, s);
    if (x_dtype == UWIR_BF16 && out_dtype == UWIR_F32)
        return launch_dw<bf16_t, float>(x, ldx, B, H, W, C, in_scale, in_shift, w3, w5, w7, b3, b5, b7, out, ldo, s);
    return uwir_set_error(UWIR_EUNSUPPORTED, "dwconv_ms: unsupported dtype combination");
}

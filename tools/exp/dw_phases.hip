// Experiment: where does dwconv_ms32 spend its time?  MODE bit0: no global loads in staging, bit1: one filter row
// instead of seven, bit2: no output stores.   python tools/exp/make_dw_phases.py regenerates this file from csrc/dwconv.hip
#include "common.h"
#include <cstdio>
#include <cstdlib>
#define DW_HALO 3
#define D2_T 16
#define D2_L (D2_T + 2 * DW_HALO)   // 22
#define D2_PITCH 23                 // quads per LDS row (odd)
#define D2_PLANE (D2_L * D2_PITCH)  // quads per channel-quad plane
#define D2_Q 8                      // quads (waves) per block
#define D2_OPS 104                  // bf16 elements per staged output pixel: 3 x 32 + 8 pad (208 B: 16-byte aligned, and
                                    // 16 row-lanes x 208 B land on 16 distinct bank groups)

// Measured on MI355X (tools/exp/dw_phases.hip, level-0 shape 16 x 128 x 128 x 576): loads only 0.12 ms, FMAs only
// 0.24 ms, stores only 0.16 ms above the empty kernel -- and the one-tile-per-block form took their SUM (0.62 ms): the
// resident blocks run in lock step chip-wide, so the memory phases (HBM-bound while they last) and the FMA phase never
// overlap.  Hence a block walks `nt` consecutive tiles, software-pipelined: the next tile's halo is fetched into
// registers before the current tile's FMAs, and the output stores are never waited for.
//   vmcnt counts a wave's loads AND stores, and the two complete out of order with respect to each other, so a wave
// that waits for its prefetch also waits for every store it has in flight.  The roles are therefore split by wave:
// waves 0..NLW-1 issue all of the block's halo loads (and wait only for those), the remaining waves all of its output
// stores (and never wait for them); all eight waves do the FMAs.  Loads and stores are branch-free buffer accesses
// (out-of-image lanes get an offset beyond the descriptor's range: loads return zeros -- the padding --, stores are
// dropped), so the prefetch wait is a counted s_waitcnt and nothing diverges.
typedef unsigned int dw_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int dw_u32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct DwRaw;
template <> struct DwRaw<bf16_t> {
    typedef dw_u32x2 type;
    static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0); }
    static __device__ __forceinline__ float4 cvt(type v) {
        return make_float4(__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16),
                           __uint_as_float(v[1] & 0xffff0000u));
    }
};
template <> struct DwRaw<float> {
    typedef dw_u32x4 type;
    static __device__ __forceinline__ type load(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }
    static __device__ __forceinline__ float4 cvt(type v) {
        return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    }
};
#define DW_OOB 0x7fffffff   // a byte offset no tensor of this path reaches (the host checks bytes per image < 2^31)

// The same value, but not loop-invariant to the optimiser: per-thread geometry re-derived from it inside the tile loop
// costs a few integer instructions per tile; hoisted out of the loop it sat in ~35 registers across the FMA phase and
// spilled (a scratch reload in the FMA loop is a vmcnt(0): the end of the overlap).
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ int opaque_s(int v) {   // same for a wave-uniform value
    asm volatile("" : "+s"(v));
    return v;
}

template <typename TIN, typename TOUT, bool AFF, int MODE>
__global__ void __launch_bounds__(512, 4) dwconv_ms32_kernel(const TIN* __restrict__ x, int ldx, int H, int W, int C,
                                                             const float* __restrict__ in_scale,
                                                             const float* __restrict__ in_shift,
                                                             const float* __restrict__ w3, const float* __restrict__ w5,
                                                             const float* __restrict__ w7, const float* __restrict__ b3,
                                                             const float* __restrict__ b5, const float* __restrict__ b7,
                                                             TOUT* __restrict__ out, int ldo, int tiles_x, int tiles, int nt) {
    static_assert(sizeof(TOUT) == 2, "the wave-per-quad kernel writes bf16");
    __shared__ __attribute__((aligned(16))) float4 s_in[D2_Q * D2_PLANE];
    static_assert(D2_T * D2_T * D2_OPS * 2 <= D2_Q * D2_PLANE * 16, "the output staging must fit in the halo tile");
    typedef DwRaw<TIN> R;
    // fp32 input: the prefetch would hold 44 registers and cost the second block per CU -- no roles, no prefetch
    constexpr bool PREFETCH = sizeof(TIN) == 2;
    constexpr int NLW = PREFETCH ? 6 : 8;            // loader waves
    constexpr int NLT = 64 * NLW;                    // threads that stage the halo tile
    constexpr int NST = PREFETCH ? 512 - NLT : 512;  // threads that store the outputs
    const int tid = threadIdx.x, lane = tid & 63;
    const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);   // this wave's channel quad (scalar)
    const bool loader = wq < NLW, storer = !PREFETCH || wq >= NLW;
    const int b = blockIdx.z;
    const int c0 = blockIdx.x * (4 * D2_Q);
    const size_t img = (size_t)H * W;
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)b * img * ldx), 0,
                                                                   (int)(img * ldx * sizeof(TIN)), 0x00020000);
    __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (size_t)b * img * ldo), 0,
                                                                   (int)(img * ldo * sizeof(TOUT)), 0x00020000);

    // ---- halo staging: item = (pixel, quad), the quads of a pixel on adjacent lanes (64 contiguous bytes of bf16); the
    // quad of an item is tid % 8 for every item, so the normalise-on-load affine is loaded once
    constexpr int NITEM = D2_L * D2_L * D2_Q, NIT = (NITEM + NLT - 1) / NLT;
    const int q = tid % D2_Q;
    typename R::type raw[NIT];
    auto item_ok = [&](int tid_o, int it, int ty0, int tx0, int& ly, int& lx, int& iy, int& ix) {
        const int pix = (tid_o >> 3) + (NLT / D2_Q) * it;
        ly = (pix * 2979) >> 16;   // pix / 22 for pix < 1024
        lx = pix - D2_L * ly;
        iy = ty0 + ly - DW_HALO, ix = tx0 + lx - DW_HALO;
        return pix < D2_L * D2_L && iy >= 0 && iy < H && ix >= 0 && ix < W;
    };
    auto load_raw = [&](int t, bool live) {
        const int ty0 = (t / tiles_x) * D2_T, tx0 = (t % tiles_x) * D2_T;
        const int tid_o = opaque(tid);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int ly, lx, iy, ix;
            const bool ok = item_ok(tid_o, it, ty0, tx0, ly, lx, iy, ix) && live;
            const int off = ((iy * W + ix) * ldx + c0 + 4 * q) * (int)sizeof(TIN);
            raw[it] = R::load(xrs, (ok && !(MODE & 1)) ? off : DW_OOB);
        }
    };
    float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
    if (AFF) {
        sc = load4(in_scale + (size_t)b * C + c0 + 4 * q);
        sh = load4(in_shift + (size_t)b * C + c0 + 4 * q);
    }
    auto stage = [&](int t) {
        const int ty0 = (t / tiles_x) * D2_T, tx0 = (t % tiles_x) * D2_T;
        const int tid_o = opaque(tid);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int ly, lx, iy, ix;
            const bool ok = item_ok(tid_o, it, ty0, tx0, ly, lx, iy, ix);
            float4 v = R::cvt(raw[it]);
            if (AFF && ok) {   // padding stays zero: it pads the NORMALISED image
                v.x = fmaf(v.x, sc.x, sh.x);
                v.y = fmaf(v.y, sc.y, sh.y);
                v.z = fmaf(v.z, sc.z, sh.z);
                v.w = fmaf(v.w, sc.w, sh.w);
            }
            if ((tid_o >> 3) + (NLT / D2_Q) * it < D2_L * D2_L) s_in[q * D2_PLANE + ly * D2_PITCH + lx] = v;
        }
    };

    const int cq = c0 + 4 * wq;                     // first channel of this wave's quad (scalar)
    const float* __restrict__ w3q = w3 + cq;
    const float* __restrict__ w5q = w5 + cq;
    const float* __restrict__ w7q = w7 + cq;
    bf16_t* s_out = reinterpret_cast<bf16_t*>(s_in);

    int t = blockIdx.y * nt;
    const int t_end = min(t + nt, tiles);
    if (PREFETCH && loader) load_raw(t, true);
    for (; t < t_end; ++t) {
        if (!PREFETCH) load_raw(t, true);
        if (loader) stage(t);
        __syncthreads();
        if (PREFETCH && loader) load_raw(t + 1, t + 1 < t_end);   // in flight during the FMAs below
        const int lane_o = opaque(lane);
        const int y = lane_o & 15, xg = lane_o >> 4;
        const float4* plane = s_in + wq * D2_PLANE + y * D2_PITCH + xg * 4;
        float4 a3[4], a5[4], a7[4];
        {
            const int cq_o = opaque_s(cq);
            const float4 z4 = make_float4(0, 0, 0, 0);
            const float4 v3 = b3 ? load4(b3 + cq_o) : z4, v5 = b5 ? load4(b5 + cq_o) : z4, v7 = b7 ? load4(b7 + cq_o) : z4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a3[j] = v3; a5[j] = v5; a7[j] = v7; }
        }
#define FMA4(acc, wv, iv)                 \
    acc.x = fmaf(wv.x, iv.x, acc.x);      \
    acc.y = fmaf(wv.y, iv.y, acc.y);      \
    acc.z = fmaf(wv.z, iv.z, acc.z);      \
    acc.w = fmaf(wv.w, iv.w, acc.w);
#pragma unroll 1
        for (int ky = 0; ky < ((MODE & 2) ? 1 : 7); ++ky) {
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
        // a lane's results are 8-byte pieces scattered over 12 (pixel, scale) rows: transposed through LDS (the halo tile's
        // storage, reused) they leave as 64-byte runs per (pixel, scale)
        __syncthreads();   // every wave is done reading the halo tile
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16_t* o = s_out + ((xg * 4 + j) * D2_T + y) * D2_OPS + 4 * wq;   // slot = x * 16 + y: row-lanes adjacent
            store4(o, a3[j]);
            store4(o + 32, a5[j]);
            store4(o + 64, a7[j]);
        }
        __syncthreads();
        if (storer) {
            // 16-byte pieces, 4 per (pixel, scale): piece = i & 3, pixel = (i >> 2) % 256, scale = i >> 10, so a store
            // instruction covers 16 whole 64-byte runs and everything but the thread's base offsets is a constant
            const int ty0 = (t / tiles_x) * D2_T, tx0 = (t % tiles_x) * D2_T;
            const int ts = opaque(tid) & (NST - 1);
            const int piece = ts & 3, px = (ts >> 2) & 15, py0 = ts >> 6;
            const int ox = tx0 + px;
            const bf16_t* src = s_out + (px * D2_T + py0) * D2_OPS + piece * 8;
            const int off0 = (((ty0 + py0) * W + ox) * ldo + c0 + piece * 8) * 2;
            constexpr int RPS = 1024 / NST;   // stores per scale: each covers NST / 64 pixel rows
#pragma unroll
            for (int r = 0; r < RPS; ++r) {
                const int oy = ty0 + py0 + (NST / 64) * r;
                const int voff = (oy < H && ox < W && !(MODE & 4)) ? off0 + (NST / 64) * r * W * ldo * 2 : DW_OOB;
#pragma unroll
                for (int scl = 0; scl < 3; ++scl) {
                    const dw_u32x4 v = *reinterpret_cast<const dw_u32x4*>(src + (NST / 64) * r * D2_OPS + scl * 32);
                    __builtin_amdgcn_raw_buffer_store_b128(v, ors, voff, scl * C * 2, 0);
                }
            }
        }
        __syncthreads();   // the staging area is the next tile's halo tile
    }
}


template <int MODE> static float run(const bf16_t* x, bf16_t* out, const float* w, int B, int H, int W, int C, int nt) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int tx = (W + 15) / 16, ty = (H + 15) / 16, tiles = tx * ty;
    auto go = [&]() { hipLaunchKernelGGL((dwconv_ms32_kernel<bf16_t, bf16_t, false, MODE>), dim3(C / 32, (tiles + nt - 1) / nt, B), dim3(512), 0, 0, x, C, H, W, C,
        (const float*)nullptr, (const float*)nullptr, w, w + 9 * C, w + 34 * C, w + 83 * C, w + 84 * C, w + 85 * C, out, 3 * C, tx, tiles, nt); };
    go(); go(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for (int i = 0; i < 10; ++i) go(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}
int main(int argc, char** argv) {
    const int B = 16, H = 128, W = 128, C = 576;
    size_t nx = (size_t)B * H * W * C;
    bf16_t *x, *out; float* w;
    (void)hipMalloc(&x, nx * 2); (void)hipMalloc(&out, nx * 6); (void)hipMalloc(&w, 86 * C * 4);
    (void)hipMemset(x, 0, nx * 2); (void)hipMemset(w, 0, 86 * C * 4);
    for (int a = 1; a < (argc > 1 ? argc : 2); ++a) {
        const int nt = argc > 1 ? atoi(argv[a]) : 4;
        printf("tiles per block %d\n", nt);
        printf("  full            %.3f ms\n", run<0>(x, out, w, B, H, W, C, nt));
        printf("  no loads        %.3f ms\n", run<1>(x, out, w, B, H, W, C, nt));
        printf("  1/7 compute     %.3f ms\n", run<2>(x, out, w, B, H, W, C, nt));
        printf("  no stores       %.3f ms\n", run<4>(x, out, w, B, H, W, C, nt));
        printf("  loads only      %.3f ms\n", run<6>(x, out, w, B, H, W, C, nt));
        printf("  compute only    %.3f ms\n", run<5>(x, out, w, B, H, W, C, nt));
        printf("  stores only     %.3f ms\n", run<3>(x, out, w, B, H, W, C, nt));
        printf("  nothing         %.3f ms\n", run<7>(x, out, w, B, H, W, C, nt));
    }
    return 0;
}

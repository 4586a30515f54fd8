// Large-K bf16 x bf16 GEMM on 256 x 288 block tiles (the dw_merge 1x1 convs of the student: N = 576 / 1152 / 2304,
// K = 1728 / 3456 / 6912; wtb.py:91), one block of 8 waves per CU.
//
// Why this tile.  Ablations of the 256 x 192 kernel (gemm_wide.hip) in a lock-step, 64-deep-stage form, M262144 N576
// K1728, one process, interleaved rounds (uwir_gemm_wide_probe below runs the same ablations of THIS kernel):
//     whole kernel                                         0.539 ms  (967 TF/s)
//     no MFMAs (LDS-DMA fill + fragment reads + barriers)  0.517 ms
//     no LDS-DMA inside the loop (MFMAs + fragment reads)  0.367 ms
//     every block stages block 0's A rows (all L2 hits)    0.453 ms
//     fill only, free-running, 1 / 2 / 3 stages in flight  0.398 / 0.403 / 0.403 ms
// The L2 -> LDS fill alone takes 0.40 ms however deep it is pipelined: 4.76 GB staged per launch = 12 TB/s = 47 GB/s per
// CU, which is what the programming guide's per-CU fill rates give for this mix (66-73 GB/s from the XCD's L2, 24-33 GB/s
// beyond it; 29 % of the staged bytes are first touches of A).  The kernel is bound by the bytes it stages, not by MFMA
// issue or latency, so the lever is bytes per FLOP = (1 / BM + 1 / BN): 256 x 288 stages 19 % fewer than 256 x 192 (and A
// is re-read from L2 twice instead of three times at N = 576).  288 = 2 x 144 is the widest tile whose accumulators fit
// beside the fragments (8 waves of 64 x 144 = 4 x 9 tiles of v_mfma_f32_16x16x32_bf16: 144 + 52 + addressing = 214 VGPRs
// of the 256 a wave gets at two waves per SIMD), and it divides every merge conv of the student.
//
// Schedule.  K stages of 64 (128-byte LDS rows = whole L2 lines per row piece), double-buffered (2 x 68 KiB), filled by
// LDS-DMA in 1 KiB pieces of 8 rows x 128 B with the chunk index XOR (row & 7) applied on the SOURCE address and again on
// the fragment read (conflict-free ds_read_b128).  One barrier per stage: wait (own pieces of stage t) | barrier | issue
// stage t + 1 | k-half 0: 13 fragment reads, 36 MFMAs | k-half 1: the same.  A wave stalls in the issue of an LDS-DMA
// instruction while the memory pipeline is backed up (100-150 cycles per instruction here), and the two waves of a SIMD
// stalled together leave its matrix pipe idle; so waves 0-3 issue their pieces right after the barrier and waves 4-7
// (their SIMD mates) between the two k-halves.  Measured against the 256 x 192 kernel on the three merge shapes, same
// box, bit-identical outputs: 0.571 -> 0.454 ms (915 -> 1149 TF/s), 0.517 -> 0.404 (1010 -> 1298), 0.497 -> 0.376
// (1049 -> 1391); everyone issuing early 0.471 / 0.424 / 0.398; everyone mid-stage (another box) 4 % behind that; waves
// grouped by (w & 1) or (w & 2) instead 0.467 / 0.420 / 0.394 and 0.463 / 0.412 / 0.391; fragment reads pinned ahead of
// the MFMAs by sched_barrier 3 % slower than the compiler's own interleave.
// Tried on the 192-wide tile and not kept: fragments double-buffered in registers over the k-halves with the fill issued
// two stages ahead (same time: the fill is bandwidth-bound, not latency-bound), LDS-DMA instructions spread among the
// MFMAs (3 % slower).  Tried on this tile and not kept: a second barrier after the last fragment reads of a stage, so that
// its buffer is restaged for stage t + 2 while the k-half-1 MFMAs still run (two stages in flight): 0.484 -> 0.505-0.525,
// 0.425 -> 0.457-0.472, 0.398 -> 0.415-0.445 ms -- the barrier costs more than the earlier fill start gains.
// What is left (uwir_gemm_wide_probe, M262144 N576 K1728): fill alone 0.342 ms, MFMAs + fragment reads alone 0.350, fill +
// reads + barriers without MFMAs 0.404, whole kernel 0.459 = 1.31 x the larger floor.
#include "common.h"

#define G3_BM 256
#define G3_BN 288
#define G3_BK 64
#define G3_ROWB 128
#define G3_STAGE ((G3_BM + G3_BN) * G3_ROWB)   // 69632 bytes
#define G3_EP 152                               // bf16 per staged epilogue row: 144 + 8 pad
#define G3_SMEM (8 * 64 * G3_EP * 2 > 2 * G3_STAGE ? 8 * 64 * G3_EP * 2 : 2 * G3_STAGE)

struct Gw3P {
    const bf16_t* A;
    const bf16_t* W;
    bf16_t* out;
    const float* bias;
    const float* col_scale;
    const float* batch_bias;        // per-image bias rows [B, batch_ld] or null
    long long w_batch_stride;       // elements between the weight panels of consecutive images, 0 = one shared panel
    int M, N, K, lda, ldo, act, tiles_n, nblocks, rows_per_image, batch_ld;
};

__device__ __forceinline__ void g3_dma16(const void* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase)
                 : "memory");
}
template <int N> __device__ __forceinline__ void g3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// VAR 0 = the product schedule.  Timing-only ablations (results wrong or absent): 1 = every wave issues its fill right
// after the barrier, 2 = every wave mid-stage, 3 = no MFMAs, 4 = no fill inside the loop, 5 = fill only (no fragment reads,
// no MFMAs, no barriers; one stage in flight behind the one waited for).
template <int VAR>
__global__ void __launch_bounds__(512, 1) gemm_w288_kernel(Gw3P p) {
    __shared__ __attribute__((aligned(1024))) char smem[G3_SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    {
        const int nwg = p.nblocks, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int m0 = (bid / p.tiles_n) * G3_BM, n0 = (bid % p.tiles_n) * G3_BN;
    const int M = p.M, K = p.K;

    // staging: A = 32 pieces (wave w: 4w .. 4w + 3), W = 36 pieces (wave w: 4w .. 4w + 3; waves 0-3 also piece 32 + w);
    // lane -> (row in piece = lane / 8, LDS chunk = lane % 8) reads GLOBAL chunk (lane % 8) ^ (row & 7)
    const char* a_base = (const char*)(p.A + (size_t)m0 * p.lda);
    // per-image weight panels / bias rows (the folded LayerNorm2d affine): a tile lies inside one image (host check)
    const int img = p.rows_per_image ? m0 / p.rows_per_image : 0;
    const char* w_base = (const char*)(p.W + (size_t)img * (size_t)p.w_batch_stride + (size_t)n0 * K);
    unsigned a_off[4], w_off[5];
    {
        const int prow = lane >> 3, pch = lane & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (4 * wave + i) * 8 + prow;
            const int rr = min(m0 + r, M - 1) - m0;   // rows past M repeat the last one (their outputs are not stored)
            a_off[i] = (unsigned)rr * (unsigned)p.lda * 2u + (pch ^ (r & 7)) * 16;
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int piece = i < 4 ? 4 * wave + i : 32 + (wave & 3);
            const int r = piece * 8 + prow;
            w_off[i] = (unsigned)r * (unsigned)K * 2u + (pch ^ (r & 7)) * 16;
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    auto issue = [&](int kt, int buf) {
        const char* ab = a_base + (size_t)kt * (G3_BK * 2);
        const char* wb = w_base + (size_t)kt * (G3_BK * 2);
        const unsigned la = lds0 + buf * G3_STAGE, lb = la + G3_BM * G3_ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) g3_dma16(ab, a_off[i], la + (4 * wave + i) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) g3_dma16(wb, w_off[i], lb + (4 * wave + i) * 1024);
        if (wave < 4) g3_dma16(wb, w_off[4], lb + (32 + wave) * 1024);
    };

    f32x4 acc[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 144;
    const int fr = lane & 15, fq = lane >> 4;
    const int sw = fr & 7;
    const int fa_row = (wm0 + fr) * G3_ROWB, fb_row = G3_BM * G3_ROWB + (wn0 + fr) * G3_ROWB;

    // Stage t is read after a barrier that follows every wave's vmcnt(0) on its own pieces of it; buffer (t + 1) & 1 is
    // restaged after the barrier of stage t, which every wave passes with its reads of stage t - 1 retired (their MFMAs
    // consumed them).
    const int nk = K / G3_BK;
    issue(0, 0);
    if constexpr (VAR == 5) {
        for (int t = 0; t < nk; ++t) {
            if (t + 1 < nk) {
                issue(t + 1, (t + 1) & 1);
                if (wave < 4) g3_wait_vm<9>(); else g3_wait_vm<8>();
            } else g3_wait_vm<0>();
        }
    } else {
        const bool early = VAR == 1 || VAR == 3 || ((VAR == 0 || VAR == 4) && wave < 4);
        for (int t = 0; t < nk; ++t) {
            g3_wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            const bool more = VAR != 4 && t + 1 < nk;
            if (early && more) issue(t + 1, (t + 1) & 1);
            const char* st = smem + (t & 1) * G3_STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (kk == 1 && !early && more) issue(t + 1, (t + 1) & 1);
                bf16x8 fa[4], fb[9];
                const int ch = ((kk * 4 + fq) ^ sw) * 16;
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(st + fa_row + i * (16 * G3_ROWB) + ch);
#pragma unroll
                for (int j = 0; j < 9; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(st + fb_row + j * (16 * G3_ROWB) + ch);
                if constexpr (VAR == 3) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
                    for (int j = 0; j < 9; ++j) asm volatile("" ::"v"(fb[j]));
                } else {
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int j = 0; j < 9; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_s_setprio(0);
                }
            }
        }
    }

    // epilogue through LDS: lane (pixel fr of tile i, quarter fq) holds channels 16 j + 4 fq .. + 3; rows leave as whole
    // 288-byte segments
    bf16_t* stg = reinterpret_cast<bf16_t*>(smem) + wave * (64 * G3_EP);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int nl = 16 * j + 4 * fq, n = n0 + wn0 + nl;
        f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.batch_bias) bv += *reinterpret_cast<const f32x4*>(p.batch_bias + (size_t)img * p.batch_ld + n);
        const f32x4 cs = p.col_scale ? *reinterpret_cast<const f32x4*>(p.col_scale + n) : (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = acc[i][j] + bv;
            if (p.act != UWIR_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], p.act);
            }
            if (p.col_scale) v *= cs;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
            *reinterpret_cast<bf16x4*>(stg + (16 * i + fr) * G3_EP + nl) = o;
        }
    }
    bf16_t* outp = p.out + (size_t)(m0 + wm0) * p.ldo + n0 + wn0;
#pragma unroll 6
    for (int k = 0; k < 64 * 18 / 64; ++k) {
        const int pc = lane + 64 * k, row = pc / 18, c16 = pc - 18 * row;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + row * G3_EP + c16 * 8);
        if (m0 + wm0 + row < M) *reinterpret_cast<bf16x8*>(outp + (size_t)row * p.ldo + c16 * 8) = v;
    }
}

static int g3_fill(Gw3P& p, const uwir_gemm_args& g, int M, int K) {
    // K >= 1024: the merge GEMMs this kernel was built for.  192 <= K < 1024 (round 3: the WTB's qkv GEMM, three to
    // twelve stages per tile): measured against the 128 x 128 LDS-DMA kernel at M 262144 N 576 K 192 122 vs 141 us (158 with
    // per-image panels and bias rows), M 16384 N 2304 K 768 54 vs 65, M 65536 N 1152 K 384 70 vs 70.
    if (g.N % G3_BN != 0 || K % G3_BK != 0 || K < 192 || M < 4096) return 0;
    if (g.res || g.blend_out || g.blend_t || g.batch_scale || g.store_mode != 0 || !g.out || g.a_scale || g.ups) return 0;
    const int rpi = g.OH * g.OW;
    const int bld = g.batch_ld ? g.batch_ld : g.N;
    if (g.w_batch_stride || g.batch_bias) {      // per-image operands: whole tiles inside one image, 16-byte aligned rows
        if (rpi <= 0 || rpi % G3_BM != 0 || M % rpi != 0) return 0;
        if (g.w_batch_stride % 8 != 0 || g.w_batch_stride < 0 || ((uintptr_t)g.batch_bias & 15) || bld % 4 != 0) return 0;
    }
    if (g.out_dtype != UWIR_BF16) return 0;
    if (((uintptr_t)g.A & 15) || ((uintptr_t)g.Wt & 15) || g.lda % 8 != 0 || ((uintptr_t)g.out & 15) || g.ldo % 8 != 0) return 0;
    if (((uintptr_t)g.bias & 15) || ((uintptr_t)g.col_scale & 15)) return 0;
    if ((size_t)G3_BM * g.lda * 2 >= 0xffffffffULL || (size_t)G3_BN * K * 2 >= 0xffffffffULL) return 0;   // 32-bit piece offsets
    p.A = (const bf16_t*)g.A;
    p.W = (const bf16_t*)g.Wt;
    p.out = (bf16_t*)g.out;
    p.bias = g.bias;
    p.col_scale = g.col_scale;
    p.batch_bias = g.batch_bias;
    p.w_batch_stride = g.w_batch_stride;
    p.rows_per_image = (g.w_batch_stride || g.batch_bias) ? rpi : 0;
    p.batch_ld = bld;
    p.M = M; p.N = g.N; p.K = K; p.lda = g.lda; p.ldo = g.ldo; p.act = g.act;
    p.tiles_n = g.N / G3_BN;
    const long long nb = (long long)ceil_div(M, G3_BM) * p.tiles_n;
    if (nb > 0x7fffffffLL) return 0;
    p.nblocks = (int)nb;
    return 1;
}

// Called by launch_gemm (gemm.hip) for plain 1x1 bf16 GEMMs ahead of the 256 x 192 kernel; returns 1 if it took the
// launch, 0 if the shape is not its own, < 0 on error.
int uwir_gemm_wide288_try(const uwir_gemm_args& g, int M, int K, hipStream_t stream) {
    Gw3P p;
    if (!g3_fill(p, g, M, K)) return 0;
    hipLaunchKernelGGL((gemm_w288_kernel<0>), dim3(p.nblocks), dim3(512), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return uwir_set_error(UWIR_ELAUNCH, "conv_gemm(wide 288): %s", hipGetErrorString(e));
    return 1;
}

// uwir.h: measurement hook (the ablations listed above the kernel; every variant but 0 leaves `out` undefined).
extern "C" int uwir_gemm_wide_probe(const void* A, const void* Wt, void* out, const float* bias, int M, int N, int K, int variant,
                                    uwir_stream_t stream) {
    UWIR_CHECK(A && Wt && out, "gemm_wide_probe: null pointer");
    UWIR_CHECK(variant >= 0 && variant <= 5, "gemm_wide_probe: variant %d is not one of 0 .. 5", variant);
    uwir_gemm_args g = {};
    g.A = A; g.Wt = Wt; g.out = out; g.bias = bias; g.N = N; g.lda = K; g.ldo = N; g.out_dtype = UWIR_BF16; g.act = UWIR_ACT_NONE;
    Gw3P p;
    UWIR_CHECK(g3_fill(p, g, M, K), "gemm_wide_probe: M %d N %d K %d is not a shape of the 256 x 288 kernel (N %% 288 == 0, K %% 64 == 0, "
               "K >= 192, M >= 4096, 16-byte aligned pointers)", M, N, K);
#define G3_GO(V) hipLaunchKernelGGL((gemm_w288_kernel<V>), dim3(p.nblocks), dim3(512), 0, (hipStream_t)stream, p)
    switch (variant) {
        case 1: G3_GO(1); break;
        case 2: G3_GO(2); break;
        case 3: G3_GO(3); break;
        case 4: G3_GO(4); break;
        case 5: G3_GO(5); break;
        default: G3_GO(0);
    }
#undef G3_GO
    UWIR_LAUNCH_CHECK("gemm_wide_probe");
    return UWIR_OK;
}

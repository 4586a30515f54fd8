// Large-K bf16 x bf16 GEMM (the dw_merge 1x1 convs of the student, K = 1728 .. 6912: wtb.py:91 / sffb.py:46) on
// 256 x 192 block tiles, one block of 8 waves per CU.
//
// Why another kernel: the 128 x 128 x 64 LDS-DMA kernel (gemm.hip) pulls 32 KiB through the CU's vector-memory path per
// 512 MFMA cycles per SIMD and keeps four blocks per CU to hide its load -> barrier -> MFMA chain; it sits at 890-1000
// TF/s whatever is done inside that structure (MI355X programming guide: "the step-3 structure's ceiling").  Here a block
// owns 256 x 192 outputs (28 KiB per 768 MFMA cycles per SIMD: 0.57x the L2 traffic per FLOP), its 8 waves are 2 (M) x
// 4 (N) of 128 x 48 (8 x 3 tiles of v_mfma_f32_16x16x32_bf16, 96 accumulator registers), and the K loop is a ring of
// NS stages of 32 k filled by LDS-DMA (global_load_lds_dwordx4 issued from inline asm: hipcc drains a __builtin LDS-DMA
// with vmcnt(0) before the next LDS read) NS - 1 stages ahead of use and retired with COUNTED s_waitcnt, raw s_barrier.
// The two wave rows run half a stage apart (see the loop), so one row's MFMAs cover the other's barrier, DMA issue and
// fragment reads.  bf16 outputs leave through LDS as whole 96-byte row segments.
//
// Measured (MI355X, tools/exp/abab.py: alternating A/B/A/B after a warm-up -- the first configuration timed in a process
// reads 10-15 % slow, which made an earlier A-then-B figure for K = 1728 look like a gain): M262144 N576 K1728 0.565 vs
// 0.565 ms (equal, 925 TF/s), M65536 N1152 K3456 0.516 -> 0.512, M16384 N2304 K6912 0.543 -> 0.488 (1070 TF/s): the
// wide tile pays where K is long enough to amortise its prologue/epilogue at one block per CU.  What was tried on the
// way, same shapes: 4 waves of 128 x 96 (32x32x16 MFMAs, AGPR accumulators, one wave per SIMD, register-double-buffered
// fragments) 0.64 / 0.56 ms -- with the MFMAs or the DMA compiled out in turn it took 0.53 / 0.53 of a 0.30 ms
// skeleton, i.e. the two did not overlap at all; 8 waves in lock step 0.62 / 0.55; ring depth 3 / 4 / 5: 0.75 / 0.64 /
// 0.64 (latency is covered from depth 4 on).  Not built: a persistent form that keeps the ring full across output tiles
// (prologue and epilogue are exposed at one block per CU: ~10 % at K = 1728).
//
// LDS image of a stage: [row][4 chunks of 16 B] (64-byte rows), chunk index XOR gw_swz(row); the swizzle is applied on
// the SOURCE address of the lane-linear DMA and again on the fragment read.
#include "common.h"

#define GW_BM 256
#define GW_BN 192
#define GW_BK 32
#define GW_ROWB 64
#define GW_STAGE ((GW_BM + GW_BN) * GW_ROWB)   // 28672 bytes
#define GW_NS 4                                // ring depth (114 KiB); 5 measured the same

struct GwP {
    const bf16_t* A;
    const bf16_t* W;
    void* out;
    const float* bias;
    const float* col_scale;
    int M, N, K, lda, ldo, act, tiles_n, nblocks;
};

__device__ __forceinline__ void gw_dma16(const void* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase)
                 : "memory");
}
template <int N> __device__ __forceinline__ void gw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Fragment read: lane = (row l % 16, chunk l / 16).  ds_read_b128 is served in the 16-lane groups {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31} (+32 for the upper half-wave) over 16 slots of 16 bytes; a group therefore holds row quads
// {0, 3} of chunk c and {1, 2} of chunk c ^ 1 (or the reverse), each quad covering the four 64-byte sub-rows:
// chunk ^= (-q) & 3 on the row quad q = (row >> 2) & 3 makes the 16 slots distinct.
__device__ __forceinline__ int gw_swz(int row) { return (-(row >> 2)) & 3; }
#define GW_DMA 4   // LDS-DMA instructions per thread and stage (2 for A, 2 for W: waves 4-7 repeat their one W piece)

template <int NS>
__global__ void __launch_bounds__(512, 1) gemm_wide_kernel(GwP p) {
    __shared__ __attribute__((aligned(1024))) char smem[NS * GW_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    {
        const int nwg = p.nblocks, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int m0 = (bid / p.tiles_n) * GW_BM, n0 = (bid % p.tiles_n) * GW_BN;
    const int M = p.M, K = p.K;

    // staging: A = 16 pieces of 16 rows (wave w: pieces 2w, 2w + 1), W = 12 pieces (waves 0-3: 2w, 2w + 1; waves 4-7:
    // piece 4 + w twice)
    const char* a_base = (const char*)(p.A + (size_t)m0 * p.lda);
    const char* w_base = (const char*)(p.W + (size_t)n0 * K);
    unsigned a_off[2], w_off[2];
    int w_piece[2];
    {
        const int prow = lane >> 2, pch = lane & 3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (2 * wave + i) * 16 + prow;
            const int rr = min(m0 + r, M - 1) - m0;
            a_off[i] = (unsigned)rr * (unsigned)p.lda * 2u + (pch ^ gw_swz(r)) * 16;
            w_piece[i] = wave < 4 ? 2 * wave + i : 4 + wave;
            const int rw = w_piece[i] * 16 + prow;
            w_off[i] = (unsigned)rw * (unsigned)K * 2u + (pch ^ gw_swz(rw)) * 16;
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    auto issue = [&](int kt, int buf) {
        const char* ab = a_base + (size_t)kt * (GW_BK * 2);
        const char* wb = w_base + (size_t)kt * (GW_BK * 2);
        const unsigned la = lds0 + buf * GW_STAGE, lb = la + GW_BM * GW_ROWB;
#pragma unroll
        for (int i = 0; i < 2; ++i) gw_dma16(ab, a_off[i], la + (2 * wave + i) * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) gw_dma16(wb, w_off[i], lb + w_piece[i] * 1024);
    };

    f32x4 acc[8][3];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wm0 = (wave >> 2) * 128, wn0 = (wave & 3) * 48;
    const int fr = lane & 15, fq = lane >> 4;
    const int ch = (fq ^ gw_swz(fr)) * 16;
    const int fa_off = (wm0 + fr) * GW_ROWB + ch, fb_off = GW_BM * GW_ROWB + (wn0 + fr) * GW_ROWB + ch;

    // The two wave rows (waves 0-3 / 4-7: one of each per SIMD) run HALF A STAGE APART: a stage is
    //   phase A: wait (own DMA of stage t + 1 landed) | barrier | issue DMA of stage t + NS - 1 | fragment reads of stage t | wait
    //   phase B: barrier | 24 MFMAs
    // and the second row passes one extra barrier first, so while one row's MFMAs run the other row is in phase A.
    // Stage t is read after a barrier that follows EVERY wave's wait for it (the waits run one stage ahead for the row
    // that lags); buffer (t - 1) % NS is restaged after a barrier that follows every wave's lgkmcnt(0) on its reads.
    const int nk = K / GW_BK;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s, s);
    if (wave >= 4) {   // the lagging row's share of stage 0 must have landed before the leading row's first read
        gw_wait_vm<GW_DMA * (NS - 2)>();
        __builtin_amdgcn_s_barrier();
    }
    for (int t = 0; t < nk; ++t) {
        const int ahead = max(min(nk - 1, t + NS - 2) - (t + 1), 0);   // stages issued after stage t + 1
        if (ahead >= NS - 3) gw_wait_vm<GW_DMA * (NS - 3)>();
        else if (NS > 4 && ahead == NS - 4) gw_wait_vm<GW_DMA * (NS > 4 ? NS - 4 : 0)>();
        else if (NS > 5 && ahead == NS - 5) gw_wait_vm<GW_DMA * (NS > 5 ? NS - 5 : 0)>();
        else gw_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        const char* st = smem + (t % NS) * GW_STAGE;
        bf16x8 fa[8], fb[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(st + fb_off + j * (16 * GW_ROWB));
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(st + fa_off + i * (16 * GW_ROWB));
        __builtin_amdgcn_sched_barrier(0);
        if (t + NS - 1 < nk) issue(t + NS - 1, (t + NS - 1) % NS);   // behind the reads: their latency covers the issue
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();

    // epilogue through LDS, as above: lane (pixel fr of tile i, quarter fq) holds channels 16 j + 4 fq .. + 3
    constexpr int EP = 56;   // bf16 per staged row: 48 + 8 pad
    static_assert(8 * 128 * EP * 2 <= NS * GW_STAGE, "epilogue staging must fit in the stage ring");
    bf16_t* stg = reinterpret_cast<bf16_t*>(smem) + wave * (128 * EP);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int nl = 16 * j + 4 * fq, n = n0 + wn0 + nl;
        const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 cs = p.col_scale ? *reinterpret_cast<const f32x4*>(p.col_scale + n) : (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x4 v = acc[i][j] + bv;
            if (p.act != UWIR_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], p.act);
            }
            if (p.col_scale) v *= cs;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
            *reinterpret_cast<bf16x4*>(stg + (16 * i + fr) * EP + nl) = o;
        }
    }
    bf16_t* outp = (bf16_t*)p.out + (size_t)(m0 + wm0) * p.ldo + n0 + wn0;
#pragma unroll 4
    for (int k = 0; k < 128 * 6 / 64; ++k) {
        const int pc = lane + 64 * k, row = pc / 6, c16 = pc - 6 * row;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + row * EP + c16 * 8);
        if (m0 + wm0 + row < M) *reinterpret_cast<bf16x8*>(outp + (size_t)row * p.ldo + c16 * 8) = v;
    }
}


// Called by launch_gemm (gemm.hip) for plain 1x1 bf16 GEMMs; returns 1 if it took the launch, 0 if the shape is not its
// own, < 0 on error.
int uwir_gemm_wide_try(const uwir_gemm_args& g, int M, int K, hipStream_t stream) {
    if (g.N % GW_BN != 0 || K % GW_BK != 0 || K < 1024 || K / GW_BK < GW_NS || M < 4096) return 0;
    if (g.res || g.blend_out || g.blend_t || g.batch_scale || g.batch_bias || g.store_mode != 0 || !g.out || g.a_scale || g.ups)
        return 0;
    if (g.out_dtype != UWIR_BF16) return 0;
    if (((uintptr_t)g.A & 15) || ((uintptr_t)g.Wt & 15) || g.lda % 8 != 0 || ((uintptr_t)g.out & 15) || g.ldo % 8 != 0) return 0;
    if (((uintptr_t)g.bias & 15) || ((uintptr_t)g.col_scale & 15)) return 0;
    if ((size_t)GW_BM * g.lda * 2 >= 0xffffffffULL || (size_t)GW_BN * K * 2 >= 0xffffffffULL) return 0;
    GwP p;
    p.A = (const bf16_t*)g.A;
    p.W = (const bf16_t*)g.Wt;
    p.out = g.out;
    p.bias = g.bias;
    p.col_scale = g.col_scale;
    p.M = M; p.N = g.N; p.K = K; p.lda = g.lda; p.ldo = g.ldo; p.act = g.act;
    p.tiles_n = g.N / GW_BN;
    const long long nb = (long long)ceil_div(M, GW_BM) * p.tiles_n;
    if (nb > 0x7fffffffLL) return 0;
    p.nblocks = (int)nb;
    hipLaunchKernelGGL((gemm_wide_kernel<GW_NS>), dim3(p.nblocks), dim3(512), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return uwir_set_error(UWIR_ELAUNCH, "conv_gemm(wide): %s", hipGetErrorString(e));
    return 1;
}

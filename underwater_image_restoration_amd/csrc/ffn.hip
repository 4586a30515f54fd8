// The WTB's feed-forward block in one launch (bf16 mode of the student; wtb.py:117-124):
//     out = W2 gelu(W0_b x + bias0_b) + b2 + x
// x [M, C] bf16 rows (the residual stream x1), W0_b [B][2C][C] bf16 = ffn.0.weight with the norm2 affine of image b folded
// in (uwir_fold_affine_weights), bias0_b [B][2C], W2 [C][2C] bf16 = ffn.2.weight, out [M, C] bf16.
//
// As two launches the [M, 2C] hidden tensor is written and read back: 2.7 KB of HBM traffic per row at C = 192 against the
// 0.77 KB of x in, out out.  Here it never leaves registers.  Both GEMMs run on v_mfma_f32_16x16x32_bf16 in the
// "channels x pixels" orientation of gemm.hip (a lane holds 4 consecutive output channels of one pixel), and two 16-channel
// accumulator tiles side by side are exactly the 8 k-values a lane must hold for the pixel operand of the next MFMA when
// the k axis is read in the order pi(32 s + 8 q + e) = 32 s + 4 q + e (e < 4), 32 s + 16 + 4 q + e - 4 (e >= 4): W2 is
// stored with its columns in that order at load time (student.py: ffn2_wp) -- the same trick as gate.hip and as the
// P -> P V step of the attention kernels.
//
// A block of 8 waves owns 16 * PT * 8 rows of ONE image per pass (the weights are per image) and walks the hidden
// channels in chunks of HC: per chunk the HC x C slice of W0_b and the C x HC slice of W2 (48 KiB together) arrive by
// LDS-DMA into a double buffer, one barrier per chunk, the waves that share a SIMD issuing their DMA half a chunk apart
// (gemm_wide288.hip); a wave keeps its rows' x fragments (the k operand of the first GEMM) and its C-wide output
// accumulators in registers for the whole pass.  The output leaves through LDS as whole rows.
//
// Measured (MI355X, tools/exp/ffn_bench.py, B = 16): C 192, n 16384 (level 0) 250 -> 139 us for the pair of launches it
// replaces, C 384, n 4096 (level 1) 120 -> 108 us; student forward 47.4 -> 46.7 ms A/B/A/B, PSNR against the oracle unchanged
// (71.2 dB).  Where the 139 us go (VAR ablations, timing only): without the GELU 102 us (16 VALU instructions, two of
// them transcendental, per hidden value: 37 us), with one output tile of GEMM 2 instead of twelve 120 us, with neither
// 87 us -- i.e. GEMM 1 (20 us of MFMA at the pipe rate), the weight stream (295 MB through L2 -> LDS per launch), the x
// loads and the output stores take 87 us: per pass the x loads, the first chunk's DMA and the output staging are exposed
// (about 8 of 22 us; four passes per CU).  Next: issue the following pass's x loads and first chunk behind the last
// chunk's GEMM 2 (needs the output staged per pixel tile beside the ring instead of in it).  A static s_setprio 1 for the
// second wave half (VAR & 4) changed nothing.
#include "common.h"

template <int CW> struct FfnCfg;
template <> struct FfnCfg<192> { static constexpr int PT = 2, HC = 64; };
template <> struct FfnCfg<384> { static constexpr int PT = 1, HC = 32; };

struct FfnP {
    const bf16_t* x;       // [M, ldx]
    const bf16_t* w0;      // [B][2C][C]
    const float* bias0;    // [B][2C]
    const bf16_t* w2p;     // [C][2C], columns permuted by pi
    const float* b2;       // [C]
    bf16_t* out;           // [M, ldo]
    long long w0_stride;   // elements between images
    int ldx, ldo, n, npass;
};

__device__ __forceinline__ void ffn_dma16(const void* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase)
                 : "memory");
}
__device__ __forceinline__ int ffn_swz4(int row) { return (-(row >> 2)) & 3; }   // 64-byte LDS rows (gemm_wide.hip)

template <int CW, int VAR>
__global__ void __launch_bounds__(512, 1) ffn_fused_kernel(FfnP p) {
    constexpr int PT = FfnCfg<CW>::PT, HC = FfnCfg<CW>::HC;
    constexpr int NKS = CW / 32;            // k-steps of the first GEMM
    constexpr int NCH = 2 * CW / HC;        // hidden chunks
    constexpr int HT = HC / 16;             // hidden tiles per chunk
    constexpr int KS2 = HC / 32;            // k-steps of the second GEMM per chunk
    constexpr int OT = CW / 16;             // output tiles
    constexpr int A_BYTES = HC * CW * 2, B_BYTES = CW * HC * 2, STAGE = A_BYTES + B_BYTES;   // 24 + 24 KiB
    constexpr int PXB = 8 * 16 * PT;        // rows per block and pass
    constexpr int EP = CW * 2 + 16;         // bytes per staged output row
    static_assert(STAGE == 49152 && (A_BYTES / 1024) == 24 && (B_BYTES / 1024) == 24, "6 DMA pieces per wave and chunk");
    constexpr int RING = 8 * 16 * PT * EP > 2 * STAGE ? 8 * 16 * PT * EP : 2 * STAGE;   // chunk ring, reused as output staging
    // one LDS object: [ring | this pass's bias0 row (2C floats)]
    __shared__ __attribute__((aligned(1024))) char smem[RING + 2 * CW * 4];
    float* sbias = reinterpret_cast<float*>(smem + RING);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;

    // ---- DMA descriptors: 48 pieces of 1 KiB per chunk, wave w issues A pieces 3w .. 3w + 2 and B pieces 3w .. 3w + 2
    // A (W0 slice) image: [seg][row][128 B], 16-byte chunk index XOR (row & 7); piece = (seg, 8 rows)
    // B (W2 slice) image: HC = 64: [row][128 B], chunk XOR (row & 7), piece = 8 rows; HC = 32: [row][64 B], chunk XOR swz4(row),
    //                     piece = 16 rows
    unsigned a_off[3], b_off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int pc = 3 * wave + i;
        {
            const int seg = pc / (HC / 8), rg = pc - seg * (HC / 8);
            const int r = 8 * rg + (lane >> 3), c = lane & 7;
            a_off[i] = (unsigned)r * (CW * 2) + seg * 128 + ((c ^ (r & 7)) * 16);
        }
        if (HC == 64) {
            const int r = 8 * pc + (lane >> 3), c = lane & 7;
            b_off[i] = (unsigned)r * (2 * CW * 2) + ((c ^ (r & 7)) * 16);
        } else {
            const int r = 16 * pc + (lane >> 2), c = lane & 3;
            b_off[i] = (unsigned)r * (2 * CW * 2) + ((c ^ ffn_swz4(r)) * 16);
        }
    }
    // fragment read offsets (bytes inside a stage)
    int fa_off[HT];      // + (s >> 1) * (HC * 128) + chunk term, see wa()
#pragma unroll
    for (int j = 0; j < HT; ++j) fa_off[j] = (16 * j + fr) * 128;
    auto wa = [&](const char* st, int j, int s) {   // W0 fragment: hidden tile j of the chunk, k-step s
        const int c16 = 4 * (s & 1) + fq;
        return *reinterpret_cast<const bf16x8*>(st + (s >> 1) * (HC * 128) + fa_off[j] + ((c16 ^ (fr & 7)) * 16));
    };
    auto wb = [&](const char* st, int jo, int s) {  // W2 fragment: output tile jo, k-step s of the chunk
        const int row = 16 * jo + fr;
        if (HC == 64) return *reinterpret_cast<const bf16x8*>(st + A_BYTES + row * 128 + (((4 * s + fq) ^ (fr & 7)) * 16));
        return *reinterpret_cast<const bf16x8*>(st + A_BYTES + row * 64 + ((fq ^ ffn_swz4(row)) * 16));
    };

    if ((VAR & 4) && wave >= 4) __builtin_amdgcn_s_setprio(1);   // (measured: no effect, see the header)
    for (int pass = blockIdx.x; pass < p.npass; pass += gridDim.x) {
        const int row0 = pass * PXB;                 // PXB divides n: the pass lies inside one image
        const int b = row0 / p.n;
        const char* w0b = (const char*)(p.w0 + (size_t)b * p.w0_stride);
        const char* w2b = (const char*)p.w2p;
        const float* bias0 = p.bias0 + (size_t)b * (2 * CW);
        auto issue = [&](int ch, int buf) {
            const char* ab = w0b + (size_t)ch * (HC * CW * 2);
            const char* bb = w2b + (size_t)ch * (HC * 2);
            const unsigned la = lds0 + buf * STAGE + (3 * wave) * 1024, lb = la + A_BYTES;
#pragma unroll
            for (int i = 0; i < 3; ++i) ffn_dma16(ab, a_off[i], la + i * 1024);
#pragma unroll
            for (int i = 0; i < 3; ++i) ffn_dma16(bb, b_off[i], lb + i * 1024);
        };
        issue(0, 0);
        // the bias row goes through LDS: an ordinary global load inside the chunk loop would make hipcc wait vmcnt(0) at its
        // first use and drain the LDS-DMA that was just issued
        for (int i = tid; i < 2 * CW / 4; i += 512)
            *reinterpret_cast<float4*>(sbias + 4 * i) = *reinterpret_cast<const float4*>(bias0 + 4 * i);

        // ---- this wave's rows: x fragments (k operand of GEMM 1) and the output accumulators = b2 + x (the residual)
        const int wrow0 = row0 + wave * (16 * PT);
        bf16x8 xf[PT][NKS];
        f32x4 acc2[PT][OT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const bf16_t* xr = p.x + (size_t)(wrow0 + 16 * pt + fr) * p.ldx;
#pragma unroll
            for (int s = 0; s < NKS; ++s) xf[pt][s] = *reinterpret_cast<const bf16x8*>(xr + 32 * s + 8 * fq);
#pragma unroll
            for (int jo = 0; jo < OT; ++jo) {
                const float4 r = load4(xr + 16 * jo + 4 * fq);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(p.b2 + 16 * jo + 4 * fq);
                acc2[pt][jo] = (f32x4){bv[0] + r.x, bv[1] + r.y, bv[2] + r.z, bv[3] + r.w};
            }
        }

        for (int ch = 0; ch < NCH; ++ch) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of chunk ch (and, first time, its x rows)
            __builtin_amdgcn_s_barrier();                         // everyone's pieces; everyone is done with chunk ch - 1
            const bool more = ch + 1 < NCH;
            if (wave < 4 && more) issue(ch + 1, (ch + 1) & 1);
            const char* st = smem + (ch & 1) * STAGE;
            // GEMM 1: hidden tiles of the chunk, accumulators start as the per-image bias
            f32x4 acc1[PT][HT];
#pragma unroll
            for (int j = 0; j < HT; ++j) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(sbias + ch * HC + 16 * j + 4 * fq);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) acc1[pt][j] = bv;
            }
#pragma unroll
            for (int s = 0; s < NKS; ++s)
#pragma unroll
                for (int j = 0; j < HT; ++j) {
                    const bf16x8 wf = wa(st, j, s);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) acc1[pt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[pt][s], acc1[pt][j], 0, 0, 0);
                }
            if (wave >= 4 && more) issue(ch + 1, (ch + 1) & 1);
            // GELU -> the pixel operand of GEMM 2 (k order pi)
            bf16x8 hf[PT][KS2];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int j = 0; j < HT; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc1[pt][j][e];
                        hf[pt][j >> 1][4 * (j & 1) + e] = (VAR & 1) ? (bf16_t)v : (bf16_t)(0.5f * v * (1.f + erf_fast(v * 0.70710678118654752440f)));
                    }
            // GEMM 2
#pragma unroll
            for (int s = 0; s < KS2; ++s)
#pragma unroll
                for (int jo = 0; jo < ((VAR & 2) ? 1 : OT); ++jo) {
                    const bf16x8 wf = wb(st, jo, s);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) acc2[pt][jo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, hf[pt][s], acc2[pt][jo], 0, 0, 0);
                }
        }

        // ---- output through LDS (the chunk ring, now idle) as whole rows
        __syncthreads();
        char* stg = smem + wave * (16 * PT * EP);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int jo = 0; jo < OT; ++jo) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)acc2[pt][jo][e];
                *reinterpret_cast<bf16x4*>(stg + (16 * pt + fr) * EP + (16 * jo + 4 * fq) * 2) = o;
            }
        __builtin_amdgcn_wave_barrier();
        constexpr int PPR = CW * 2 / 16;     // 16-byte pieces per row
#pragma unroll 4
        for (int k = 0; k < 16 * PT * PPR / 64; ++k) {
            const int id = lane + 64 * k, r = id / PPR, c = id - r * PPR;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + r * EP + c * 16);
            *reinterpret_cast<bf16x8*>(p.out + (size_t)(wrow0 + r) * p.ldo + c * 8) = v;
        }
        __syncthreads();                     // the next pass's first DMA lands in this memory
    }
}

extern "C" int uwir_ffn_fused_supported(int C, int n) {
    return (C == 192 && n % 256 == 0) || (C == 384 && n % 128 == 0);
}

extern "C" int uwir_ffn_fused(const void* x, int ldx, const void* w0_batched, long long w0_batch_stride, const float* bias0_batched,
                              const void* w2_permuted, const float* b2, int B, int n, int C, void* out, int ldo, uwir_stream_t stream) {
    UWIR_CHECK(x && w0_batched && bias0_batched && w2_permuted && b2 && out, "ffn_fused: null pointer");
    UWIR_CHECK(B > 0 && n > 0 && uwir_ffn_fused_supported(C, n),
               "ffn_fused: C = %d, n = %d is not a built shape (C = 192 with n %% 256 == 0, C = 384 with n %% 128 == 0)", C, n);
    UWIR_CHECK(ldx >= C && ldx % 8 == 0 && ldo >= C && ldo % 8 == 0 && w0_batch_stride >= 2LL * C * C && w0_batch_stride % 8 == 0,
               "ffn_fused: bad strides");
    UWIR_CHECK((((uintptr_t)x | (uintptr_t)w0_batched | (uintptr_t)w2_permuted | (uintptr_t)out | (uintptr_t)bias0_batched |
                 (uintptr_t)b2) & 15) == 0, "ffn_fused: pointers must be 16-byte aligned");
    UWIR_CHECK((long long)B * n < 0x7fffffffLL, "ffn_fused: too many rows");
    FfnP p;
    p.x = (const bf16_t*)x; p.w0 = (const bf16_t*)w0_batched; p.bias0 = bias0_batched; p.w2p = (const bf16_t*)w2_permuted; p.b2 = b2;
    p.out = (bf16_t*)out; p.w0_stride = w0_batch_stride; p.ldx = ldx; p.ldo = ldo; p.n = n;
    const int pxb = C == 192 ? 256 : 128;
    p.npass = (int)((long long)B * n / pxb);
    const int grid = p.npass < 256 ? p.npass : 256;
    if (C == 192) hipLaunchKernelGGL((ffn_fused_kernel<192, 0>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((ffn_fused_kernel<384, 0>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
    UWIR_LAUNCH_CHECK("ffn_fused");
    return UWIR_OK;
}

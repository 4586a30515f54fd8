// WFIGate + inverse Haar transform in one pass (bf16 mode of the student): wfi_gate.py:29-45 followed by model.py's
// `idwt2(merged, hf)` (dwt_fft.py:24-37).
//
//   hidden = relu(W1 [t | f | masks] + bias_b)        64 channels      (bias_b = net.0.bias + W1_cond cond_proj(z_d): per image)
//   alpha  = sigmoid(W2 hidden + b2)                  C channels
//   merged = alpha t + (1 - alpha) f
//   out    = idwt2(merged, hf)                        4 pixels x C channels per sub-band pixel
//
// As separate launches (conv_gemm -> conv_gemm with the blend epilogue -> idwt2) every sub-band pixel moved 3.0 KB through
// HBM at C = 64 (hidden written and read back, t and f read twice, merged written and read back); here it moves 1.7 KB: the
// gate input row and the hf row in, alpha (an output of the model) and the four output pixels out.  A wave owns 16 pixels.
// Both small GEMMs run on v_mfma_f32_16x16x32_bf16 in the "channels x pixels" orientation of gemm.hip, in which a lane
// holds 4 consecutive output channels of one pixel; two such 16-channel tiles side by side are exactly the 8 k-values a lane
// must hold for the next MFMA's pixel operand if the k axis is read in the order pi(32 s + 8 q + e) = 32 s + 4 q + e (e < 4),
// 32 s + 16 + 4 q + e - 4 (e >= 4) -- so W2 is staged into LDS with its columns in that order and the hidden tensor never
// leaves registers.  t and f come back from the gate row in the accumulator layout for the blend (L1 / L2 hits: the same
// wave loaded those lines for the first GEMM a few hundred cycles earlier).
//
// Measured (MI355X, tools/exp/gate_bench.py, B = 16, 256 x 256 student, three launches -> one): C 64 at 128 x 128 178 -> 104
// us (4.35 TB/s of the 453 MB it moves), C 128 at 64 x 64 80 -> 57, C 256 at 32 x 32 52 -> 40; student forward 47.65 ->
// 47.2 ms A/B/A/B, outputs bit-identical to the three launches (same MFMAs in the same k order, same epilogue expressions).
// The first version stored straight from the accumulator layout (8 bytes per lane: 32-byte pieces of an output line per
// lane group) and took 160 us at C 64: 62 us with no stores at all, +24 us for alpha, +71 us for the output pixels
// (1.9 TB/s).  Rows now leave through a per-wave LDS region as whole 128-byte (out) and 256-byte (alpha) runs.
#include "common.h"

template <int CC, int MP>
struct GateCfg {
    static constexpr int KG = 2 * CC + MP;           // gate input columns
    static constexpr int NK = (KG + 31) / 32;        // k-steps of the first GEMM
    static constexpr int P1 = NK * 32 + 8;           // LDS row pitch of W1 (bf16 elements): 16 rows hit 16 distinct 16-byte bank groups
    static constexpr int P2 = 64 + 8;                // ... of W2
    static constexpr int W_BYTES = (64 * P1 + CC * P2) * 2;
    // output staging, one region per wave, used twice per 64-channel group: 16 pixels x 64 alpha values (fp32 rows of
    // 256 + 16 bytes), then 64 output pixels x 64 channels (bf16 rows of 128 + 8 bytes)
    static constexpr int SA = 272, SO = 136;
    static constexpr int STG = 64 * SO;              // 8704 bytes (>= 16 * SA)
    static constexpr int NW = CC <= 128 ? 8 : 4;     // waves per block (weights + NW staging regions within 160 KiB)
    static constexpr int LDS_BYTES = W_BYTES + NW * STG;
};

struct GateP {
    const float* gin;        // [M, ldg]: t (C) | f (C) | masks (MP)
    const bf16_t* w1;        // [64, KG]
    const bf16_t* w2;        // [C, 64]
    const float* bias1;      // [B, 64]
    const float* b2;         // [C]
    const bf16_t* hf;        // [M, ld_hf]: lh | hl | hh
    bf16_t* out;             // [B, 2h, 2w, ldo]
    float* alpha;            // [M, C]
    float* merged;           // [M, C] or null
    bf16_t* merged16;        // blend-only form (out == null): merged as bf16 rows [M, ld_m16]
    int ld_m16;
    int ldg, ld_hf, ldo, M, n, h, w, tiles, waves_total;
};

template <int CC, int MP>
__global__ void __launch_bounds__((GateCfg<CC, MP>::NW * 64)) gate_idwt_kernel(GateP p) {
    using G = GateCfg<CC, MP>;
    constexpr int KG = G::KG, NK = G::NK, P1 = G::P1, P2 = G::P2, NW = G::NW, NT = NW * 64, SA = G::SA, SO = G::SO;
    __shared__ __attribute__((aligned(16))) char smem[G::LDS_BYTES];
    bf16_t* w1s = reinterpret_cast<bf16_t*>(smem);
    bf16_t* w2s = w1s + 64 * P1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- weights -> LDS (once per block): W1 zero-padded to NK * 32 columns, W2 with its columns in the order pi
    for (int i = tid; i < 64 * (NK * 4); i += NT) {            // 8-element pieces
        const int r = i / (NK * 4), c8 = (i - r * (NK * 4)) * 8;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = c8 + e < KG ? p.w1[(size_t)r * KG + c8 + e] : (bf16_t)0.f;
        *reinterpret_cast<bf16x8*>(w1s + r * P1 + c8) = v;
    }
    for (int i = tid; i < CC * 8; i += NT) {
        const int r = i >> 3, pos0 = (i & 7) * 8;                // positions pos0 .. pos0 + 7 = (s, q): 32 s + 8 q
        const int s = pos0 >> 5, q = (pos0 >> 3) & 3;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p.w2[(size_t)r * 64 + 32 * s + (e < 4 ? 4 * q + e : 16 + 4 * q + e - 4)];
        *reinterpret_cast<bf16x8*>(w2s + r * P2 + pos0) = v;
    }
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    const float kh = 0.70710678118654752440f * 0.70710678118654752440f;   // the reference's Haar tap, fl32(2^-0.5)^2
    char* stg = smem + G::W_BYTES + wave * G::STG;
    for (int tile = blockIdx.x * NW + wave; tile < p.tiles; tile += p.waves_total) {
        const int row = tile * 16 + fr;
        const bool live = row < p.M;
        const int rr = live ? row : p.M - 1;
        const float* grow = p.gin + (size_t)rr * p.ldg;
        const int b = rr / p.n;

        // ---- hidden = relu(W1 x + bias_b): the pixel operand of k-step s is x[32 s + 8 fq .. + 7] of this lane's pixel
        f32x4 acc1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        constexpr int CH = 5;                                    // k-steps whose loads are in flight together
#pragma unroll
        for (int s0 = 0; s0 < NK; s0 += CH) {
            float4 xa[CH], xb[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int s = s0 + u;
                if (s < NK) {
                    const int k0 = 32 * s + 8 * fq;              // (KG % 8 == 0: a lane's 8 values are all inside or all outside)
                    const bool in = (32 * s + 32 <= KG) || k0 < KG;
                    xa[u] = in ? *reinterpret_cast<const float4*>(grow + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
                    xb[u] = in ? *reinterpret_cast<const float4*>(grow + k0 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int s = s0 + u;
                if (s < NK) {
                    bf16x8 xf;
                    xf[0] = (bf16_t)xa[u].x; xf[1] = (bf16_t)xa[u].y; xf[2] = (bf16_t)xa[u].z; xf[3] = (bf16_t)xa[u].w;
                    xf[4] = (bf16_t)xb[u].x; xf[5] = (bf16_t)xb[u].y; xf[6] = (bf16_t)xb[u].z; xf[7] = (bf16_t)xb[u].w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(w1s + (16 * j + fr) * P1 + 32 * s + 8 * fq);
                        acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc1[j], 0, 0, 0);
                    }
                }
            }
        }
        bf16x8 hfrag[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias1 + (size_t)b * 64 + 16 * j + 4 * fq);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = acc1[j][e] + bv[e];
                hfrag[j >> 1][4 * (j & 1) + e] = (bf16_t)(v > 0.f ? v : 0.f);
            }
        }

        const bool inverse = p.out != nullptr;                   // block-uniform: false = blend-only form (no idwt2)
        // ---- alpha, blend, inverse Haar, 64 channels (4 MFMA tiles) at a time; this lane: channels 16 j + 4 fq .. + 3 of
        // pixel fr.  Results leave through the wave's LDS region as whole rows: direct stores from this layout are 32-byte
        // (out) and 64-byte (alpha) pieces of a line per lane group, and ran the output at 1.9 TB/s.
        const bf16_t* hrow = p.hf + (size_t)rr * p.ld_hf;
        // write-out geometry: lane handles staged rows r = lane / 8 + 8 (k & 1) (out) and lane / 16 + 4 k (alpha)
        bf16_t* obase[2];
        bool olive[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int orow = tile * 16 + (lane >> 3) + 8 * u;
            olive[u] = inverse && orow < p.M;
            const int q = olive[u] ? orow : p.M - 1;
            const int ob = q / p.n, pix = q - ob * p.n, y = pix / p.w, x = pix - y * p.w;
            obase[u] = p.out + (((size_t)ob * 2 * p.h + 2 * y) * (2 * p.w) + 2 * x) * p.ldo + (lane & 7) * 8;
        }
#pragma unroll
        for (int g = 0; g < CC / 64; ++g) {
            float4 t[4], f[4], vlh[4], vhl[4], vhh[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 64 * g + 16 * j + 4 * fq;
                t[j] = *reinterpret_cast<const float4*>(grow + c);
                f[j] = *reinterpret_cast<const float4*>(grow + CC + c);
                if (inverse) { vlh[j] = load4(hrow + c); vhl[j] = load4(hrow + CC + c); vhh[j] = load4(hrow + 2 * CC + c); }
            }
            float4 m[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 64 * g + 16 * j + 4 * fq;
                f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(w2s + (64 * g + 16 * j + fr) * P2 + 32 * s + 8 * fq);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, hfrag[s], acc2, 0, 0, 0);
                }
                const f32x4 b2 = *reinterpret_cast<const f32x4*>(p.b2 + c);
                float4 a;
                a.x = sigmoid_fast(acc2[0] + b2[0]); a.y = sigmoid_fast(acc2[1] + b2[1]);
                a.z = sigmoid_fast(acc2[2] + b2[2]); a.w = sigmoid_fast(acc2[3] + b2[3]);
                m[j].x = a.x * t[j].x + (1.f - a.x) * f[j].x; m[j].y = a.y * t[j].y + (1.f - a.y) * f[j].y;
                m[j].z = a.z * t[j].z + (1.f - a.z) * f[j].z; m[j].w = a.w * t[j].w + (1.f - a.w) * f[j].w;
                *reinterpret_cast<float4*>(stg + fr * SA + (16 * j + 4 * fq) * 4) = a;
                if (p.merged && live) *reinterpret_cast<float4*>(p.merged + (size_t)row * CC + c) = m[j];
            }
            __builtin_amdgcn_wave_barrier();
            // alpha: 16 rows x 16 pieces of 16 bytes; 16 lanes write one pixel's 256 contiguous bytes
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = (lane >> 4) + 4 * k, arow = tile * 16 + r;
                const float4 v = *reinterpret_cast<const float4*>(stg + r * SA + (lane & 15) * 16);
                if (arow < p.M) *reinterpret_cast<float4*>(p.alpha + (size_t)arow * CC + 64 * g + (lane & 15) * 4) = v;
            }
            __builtin_amdgcn_wave_barrier();
            if (!inverse) {
                // blend-only: merged as bf16, 16 rows x 128 bytes through the same region, 8 lanes per row
#pragma unroll
                for (int j = 0; j < 4; ++j) store4(reinterpret_cast<bf16_t*>(stg + fr * SO) + 16 * j + 4 * fq, m[j]);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int r = (lane >> 3) + 8 * k, mrow = tile * 16 + r;
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + r * SO + (lane & 7) * 16);
                    if (mrow < p.M) *reinterpret_cast<bf16x8*>(p.merged16 + (size_t)mrow * p.ld_m16 + 64 * g + (lane & 7) * 8) = v;
                }
                __builtin_amdgcn_wave_barrier();
                continue;
            }
            // out: staged row 16 v + fr = output pixel (2y + v / 2, 2x + v % 2) of sub-band pixel fr, 64 channels as bf16
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 q00, q01, q10, q11;
#define IHAAR(F)                                                                    \
    q00.F = kh * m[j].F + kh * vlh[j].F + kh * vhl[j].F + kh * vhh[j].F;            \
    q01.F = kh * m[j].F - kh * vlh[j].F + kh * vhl[j].F - kh * vhh[j].F;            \
    q10.F = kh * m[j].F + kh * vlh[j].F - kh * vhl[j].F - kh * vhh[j].F;            \
    q11.F = kh * m[j].F - kh * vlh[j].F - kh * vhl[j].F + kh * vhh[j].F;
                IHAAR(x) IHAAR(y) IHAAR(z) IHAAR(w)
#undef IHAAR
                bf16_t* d = reinterpret_cast<bf16_t*>(stg + fr * SO) + 16 * j + 4 * fq;
                store4(d, q00);
                store4(d + 16 * (SO / 2), q01);
                store4(d + 32 * (SO / 2), q10);
                store4(d + 48 * (SO / 2), q11);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 8; ++k) {   // piece lane + 64 k: staged row lane / 8 + 8 k = 16 v + r, v = k / 2, r = lane / 8 + 8 (k % 2)
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + ((lane >> 3) + 8 * k) * SO + (lane & 7) * 16);
                bf16_t* dst = obase[k & 1] + ((k >> 1) & 1) * p.ldo + (size_t)(k >> 2) * (2 * p.w) * p.ldo + 64 * g;
                if (olive[k & 1]) *reinterpret_cast<bf16x8*>(dst) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int CC, int MP>
static int gate_launch(GateP& p, hipStream_t stream) {
    using G = GateCfg<CC, MP>;
    // one block per CU holds the LDS (weights + staging); two rounds of blocks amortise the weight staging over >= 4 tiles per wave
    int grid = ceil_div(p.tiles, G::NW);
    if (grid > 512) grid = 512;
    p.waves_total = grid * G::NW;
    hipLaunchKernelGGL((gate_idwt_kernel<CC, MP>), dim3(grid), dim3(G::NW * 64), 0, stream, p);
    UWIR_LAUNCH_CHECK("gate_idwt");
    return UWIR_OK;
}

extern "C" int uwir_gate_idwt_supported(int C, int Kg) {
    return (C == 64 || C == 128 || C == 256) && (Kg == 2 * C || Kg == 2 * C + 16);
}

extern "C" int uwir_gate_idwt(const float* gate_in, int ldg, int Kg, const void* w1, const void* w2, const float* bias1,
                              const float* b2, const void* hf, int ld_hf, int B, int h, int w, int C, void* out, int ldo,
                              float* alpha, float* merged, uwir_stream_t stream) {
    UWIR_CHECK(gate_in && w1 && w2 && bias1 && b2 && hf && out && alpha, "gate_idwt: null pointer");
    UWIR_CHECK(uwir_gate_idwt_supported(C, Kg), "gate_idwt: C = %d, Kg = %d is not a built shape (C in {64, 128, 256}, Kg = 2 C or 2 C + 16)", C, Kg);
    UWIR_CHECK(B > 0 && h > 0 && w > 0 && (long long)B * h * w < 0x7fffffffLL, "gate_idwt: bad shape B=%d h=%d w=%d", B, h, w);
    UWIR_CHECK(ldg >= Kg && ldg % 4 == 0 && ld_hf >= 3 * C && ld_hf % 4 == 0 && ldo >= C && ldo % 4 == 0, "gate_idwt: bad strides");
    UWIR_CHECK((((uintptr_t)gate_in | (uintptr_t)bias1 | (uintptr_t)b2 | (uintptr_t)alpha | (uintptr_t)merged) & 15) == 0 &&
               (((uintptr_t)hf & 7) | ((uintptr_t)out & 15)) == 0 && ldo % 8 == 0, "gate_idwt: misaligned pointer or ldo %% 8 != 0");
    GateP p;
    p.gin = gate_in; p.w1 = (const bf16_t*)w1; p.w2 = (const bf16_t*)w2; p.bias1 = bias1; p.b2 = b2;
    p.hf = (const bf16_t*)hf; p.out = (bf16_t*)out; p.alpha = alpha; p.merged = merged; p.merged16 = nullptr; p.ld_m16 = 0;
    p.ldg = ldg; p.ld_hf = ld_hf; p.ldo = ldo; p.M = B * h * w; p.n = h * w; p.h = h; p.w = w;
    p.tiles = ceil_div(p.M, 16);
    hipStream_t st = (hipStream_t)stream;
    const bool masks = Kg != 2 * C;
    if (C == 64) return masks ? gate_launch<64, 16>(p, st) : gate_launch<64, 0>(p, st);
    if (C == 128) return masks ? gate_launch<128, 16>(p, st) : gate_launch<128, 0>(p, st);
    return masks ? gate_launch<256, 16>(p, st) : gate_launch<256, 0>(p, st);
}

// Blend-only form: alpha and merged, no inverse transform.  Between two blocks of one stage the reference runs idwt2 and then
// the next block's dwt2 on its result (S/model.py:40, :31), which returns (merged, lh, hl, hh) again up to the factor
// 4 fl32(2^-1/2)^4 = 1 - 1.2e-7: the student (bf16 mode) hands the next block merged (bf16 rows, here) and the hf rows it
// already holds, and neither the full-resolution map nor its transform touch HBM.  Same kernel, same blend expressions.
extern "C" int uwir_gate_blend(const float* gate_in, int ldg, int Kg, const void* w1, const void* w2, const float* bias1,
                               const float* b2, int B, int h, int w, int C, float* alpha, void* merged_bf16, int ld_merged,
                               uwir_stream_t stream) {
    UWIR_CHECK(gate_in && w1 && w2 && bias1 && b2 && alpha && merged_bf16, "gate_blend: null pointer");
    UWIR_CHECK(uwir_gate_idwt_supported(C, Kg), "gate_blend: C = %d, Kg = %d is not a built shape (C in {64, 128, 256}, Kg = 2 C or 2 C + 16)", C, Kg);
    UWIR_CHECK(B > 0 && h > 0 && w > 0 && (long long)B * h * w < 0x7fffffffLL, "gate_blend: bad shape B=%d h=%d w=%d", B, h, w);
    UWIR_CHECK(ldg >= Kg && ldg % 4 == 0 && ld_merged >= C && ld_merged % 8 == 0, "gate_blend: bad strides");
    UWIR_CHECK((((uintptr_t)gate_in | (uintptr_t)bias1 | (uintptr_t)b2 | (uintptr_t)alpha | (uintptr_t)merged_bf16) & 15) == 0,
               "gate_blend: misaligned pointer");
    GateP p;
    p.gin = gate_in; p.w1 = (const bf16_t*)w1; p.w2 = (const bf16_t*)w2; p.bias1 = bias1; p.b2 = b2;
    p.hf = (const bf16_t*)merged_bf16;   // never read in this form (a valid address keeps the row pointer arithmetic defined)
    p.out = nullptr; p.alpha = alpha; p.merged = nullptr; p.merged16 = (bf16_t*)merged_bf16; p.ld_m16 = ld_merged;
    p.ldg = ldg; p.ld_hf = 0; p.ldo = 0; p.M = B * h * w; p.n = h * w; p.h = h; p.w = w;
    p.tiles = ceil_div(p.M, 16);
    hipStream_t st = (hipStream_t)stream;
    const bool masks = Kg != 2 * C;
    if (C == 64) return masks ? gate_launch<64, 16>(p, st) : gate_launch<64, 0>(p, st);
    if (C == 128) return masks ? gate_launch<128, 16>(p, st) : gate_launch<128, 0>(p, st);
    return masks ? gate_launch<256, 16>(p, st) : gate_launch<256, 0>(p, st);
}

// Implicit-GEMM convolution on the CDNA4 matrix cores (1x1 convs, k x k convs, linears).
//
//   out[m, n] = epi( sum_k A[m, k] * Wt[n, k] ),  m = output pixel (b,oy,ox), k = (ky,kx,c)
//
// Tile: 128 (M) x BN (N, 64 or 128) x BK per block of 4 waves; operands go global -> registers
// (normalise-on-load affine + dtype conversion happen here) -> LDS (padded rows, conflict-free
// 16-byte fragment reads) -> MFMA.  Register prefetch of tile k+1 overlaps the MFMAs of tile k,
// one barrier per K tile.  bf16 path: v_mfma_f32_16x16x32_bf16 (fp32 accumulate); fp32 path:
// v_mfma_f32_16x16x4_f32 (exact fp32, fmaf-chain numerics).  The MFMA is issued "transposed"
// (weights in the A slot, activations in the B slot) so that each lane ends up holding 4
// CONSECUTIVE output channels of one pixel: the epilogue (bias, per-sample bias, activation,
// affine residual, gate blend, depth-to-space scatter) works on 16-byte vectors.
#include "common.h"
#include <type_traits>

struct GemmP {
    uwir_gemm_args a;
    int M, K, tiles_n, nblocks;  // nblocks = output tiles; the grid holds nblocks * splitk blocks
    int splitk;                  // > 1: K is cut into `splitk` slices, one block each (few-tile launches)
};

// ---- deterministic split-K.  A launch with fewer output tiles than CUs (DA-CLIP: 800 token rows -> 42 tiles on 256
// CUs, 48 K tiles each) is bound by the per-K-tile load latency of its few blocks.  Each slice block dumps its fp32
// accumulators to the caller's workspace ([slice][tile][reg][thread] float4: coalesced) and a second small launch adds
// the slices IN SLICE ORDER (bit-reproducible, unlike atomics) and runs the epilogue.  Two launches rather than a
// "last block combines" scheme: cross-block hand-over inside one kernel needs agent-scope release / acquire, which
// on this multi-XCD part writes back and invalidates whole L2s (measured 2x slower than not splitting at all).
#define SPLITK_MAX 8
template <int TM, int TN>
__device__ __forceinline__ void splitk_dump(const GemmP& p, const f32x4 (&acc)[TM][TN], int tile, int slice) {
    f32x4* mine = (f32x4*)p.a.workspace + ((size_t)slice * p.nblocks + tile) * ((size_t)TM * TN * 256);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mine[(size_t)(i * TN + j) * 256 + threadIdx.x] = acc[i][j];
}

template <typename T> struct MmaCfg;
template <> struct MmaCfg<bf16_t> {
    static constexpr int BK = 64, EPC = 8, KSTEP = 32, PAD = 16;   // 160-byte rows: conflict-free ds_read_b128 fragments (144-byte rows were 2-way conflicted)
};
template <> struct MmaCfg<float> {
    static constexpr int BK = 16, EPC = 4, KSTEP = 4, PAD = 4;
};

// raw (un-converted) 16-byte-ish chunk as loaded from global
template <typename TA, int EPC> struct RawChunk;
template <> struct RawChunk<float, 8> { float4 lo, hi; };
template <> struct RawChunk<float, 4> { float4 lo; };
template <> struct RawChunk<bf16_t, 8> { bf16x8 v; };

// Output-channel permutation inside each 64-wide wave column block.  LDS row rho = j*16 + r of the weight tile holds
// channel pi(rho) = (r>>2)*16 + j*4 + (r&3) (an involution: it swaps the two 2-bit fields), so that after the
// transposed MFMA the lane with fq = lane>>4 owns the 16 CONSECUTIVE channels fq*16 .. fq*16+15 of its pixel
// (acc[i][j][e] <-> channel fq*16 + j*4 + e): bias / residual loads and the stores are 32-64 contiguous bytes per lane.
__device__ __forceinline__ int chan_perm(int rho) {
    return (rho & 64) | (((rho >> 2) & 3) << 4) | (((rho >> 4) & 3) << 2) | (rho & 3);
}

// ---- shared epilogue: the lane holds, per (i, j) MFMA tile, pixel m = m_base + i*16 + fr and the 4
//      consecutive channels n = n_base + fq*16 + j*4 .. +3
// mrow[i] = output row (pixel index (b*OH + oy)*OW + ox) of the FIRST of the 16 pixels of MFMA tile i, mcnt[i] = how many
// of the 16 exist (<= 0: none): linear M tiles pass consecutive runs, spatial tiles one image row segment each.
template <int TM>
__device__ __forceinline__ void linear_rows(int (&mrow)[TM], int (&mcnt)[TM], int m_base, int M) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        mrow[i] = m_base + i * 16;
        mcnt[i] = M - mrow[i];
    }
}

// bias, per-sample scale / bias, activation, column scale, affine residual, gate blend for the lane's 16 channels of
// output row `out_row` (image b).  Feature-outer, channel-inner: every `if (g.x)` is one block-uniform branch around a
// group of INDEPENDENT loads that fly together, and the bias is loaded once per lane (epi_cols), not per row group.  (Channel-outer with the feature tests inside, every (i, j) piece exposed the
// latency of each of its loads in turn -- 16 dependent L2 round trips per wave for a plain bias: with them a block's
// epilogue took as long as its whole K loop at K = 192.)  Per element the operations and their order are unchanged.
template <int TN>
struct EpiCols {
    f32x4 bias[TN];
};
template <int TN>
__device__ __forceinline__ void epi_cols(const uwir_gemm_args& g, int n_base, int fq, EpiCols<TN>& c) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n_base + fq * 16 + j * 4;
        const bool in = n < g.N;
        c.bias[j] = (g.bias && in) ? *reinterpret_cast<const f32x4*>(g.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}
// EPI (compile-time epilogue class, chosen by epi_class on the host): EPI_FULL everything; EPI_LEAN no residual / blend /
// per-sample scale / scatter, and the bias is ALREADY in the accumulators (the kernel initialised them with it: its load
// overlaps the first K tile instead of standing between the last MFMA and the stores); EPI_RES = EPI_LEAN + an fp32
// residual (plain or affine) with nothing non-additive between the bias and it (no activation, no column scale), which
// therefore ALSO starts out in the accumulators.  The lean classes keep the plain 1x1 GEMMs at 4 resident blocks per CU (<= 128 VGPRs).
#define EPI_FULL 0
#define EPI_LEAN 1
#define EPI_RES 2
template <int TN, int EPI>
__device__ __forceinline__ void epi_values(const uwir_gemm_args& g, const f32x4 (&acc)[TN], const EpiCols<TN>& c, size_t out_row,
                                           int b, int n_base, int fq, f32x4 (&v)[TN]) {
    const int N = g.N;
    const int bld = g.batch_ld ? g.batch_ld : N;
    const int nq = n_base + fq * 16;   // + j * 4
    auto ld = [&](const float* ptr, int j) {
        return nq + j * 4 < N ? *reinterpret_cast<const f32x4*>(ptr + nq + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
#pragma unroll
    for (int j = 0; j < TN; ++j) v[j] = acc[j];
    if (EPI == EPI_FULL && g.bias) {
#pragma unroll
        for (int j = 0; j < TN; ++j) v[j] += c.bias[j];
    }
    if (EPI == EPI_FULL && g.batch_scale) {
        f32x4 t[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) t[j] = ld(g.batch_scale + (size_t)b * bld, j);
#pragma unroll
        for (int j = 0; j < TN; ++j) v[j] *= t[j];
    }
    if (g.batch_bias) {
        f32x4 t[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) t[j] = ld(g.batch_bias + (size_t)b * bld, j);
#pragma unroll
        for (int j = 0; j < TN; ++j) v[j] += t[j];
    }
    // one block-uniform dispatch per call, not one per element (act_apply's switch inside the 16-element loop compiled to
    // a chain of scalar branches around every element: it tripled the epilogue of the resident-weight conv kernel)
#define UWIR_EPI_ACT(EXPR)                                                 \
    _Pragma("unroll") for (int j = 0; j < TN; ++j)                         \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) { const float x = v[j][e]; v[j][e] = (EXPR); }
    switch (g.act) {
        case UWIR_ACT_RELU: UWIR_EPI_ACT(x > 0.f ? x : 0.f) break;
        case UWIR_ACT_GELU: UWIR_EPI_ACT(act_apply(x, UWIR_ACT_GELU)) break;
        case UWIR_ACT_SIGMOID: UWIR_EPI_ACT(sigmoid_fast(x)) break;
        case UWIR_ACT_SILU: UWIR_EPI_ACT(x * sigmoid_fast(x)) break;
        default: break;
    }
#undef UWIR_EPI_ACT
    if (g.col_scale) {
        f32x4 t[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) t[j] = ld(g.col_scale, j);
#pragma unroll
        for (int j = 0; j < TN; ++j) v[j] *= t[j];
    }
    if (EPI == EPI_FULL && g.res) {
        f32x4 r[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (g.res_dtype == UWIR_BF16) {
                const float4 t = nq + j * 4 < N ? load4((const bf16_t*)g.res + out_row * g.ldr + nq + j * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                r[j] = (f32x4){t.x, t.y, t.z, t.w};
            } else {
                r[j] = ld((const float*)g.res + out_row * g.ldr, j);
            }
        }
        if (g.res_scale) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const f32x4 s4 = ld(g.res_scale + (size_t)b * N, j), h4 = ld(g.res_shift + (size_t)b * N, j);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[j][e] = fmaf(r[j][e], s4[e], h4[e]);
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) v[j] += r[j];
    }
    if (EPI == EPI_FULL && g.blend_out) {
        f32x4 t[TN], f[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            t[j] = ld(g.blend_t + out_row * g.ld_blend_in, j);
            f[j] = ld(g.blend_f + out_row * g.ld_blend_in, j);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (nq + j * 4 >= N) continue;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[j][e] * t[j][e] + (1.f - v[j][e]) * f[j][e];
            *reinterpret_cast<f32x4*>(g.blend_out + out_row * g.ld_blend_out + nq + j * 4) = o;
        }
    }
}

static int epi_class(const uwir_gemm_args& g) {
    if (g.blend_out || g.batch_scale || g.store_mode != 0 || !g.out) return EPI_FULL;
    if (!g.res) return EPI_LEAN;
    // (an affine residual's per-image scale / shift must be one vector per 128-row tile)
    return (g.act == UWIR_ACT_NONE && !g.col_scale && (!g.res_scale || (g.OH * g.OW) % 128 == 0)) ? EPI_RES : EPI_FULL;
}
// Lean classes: the accumulators START as bias (+ residual).  epi_start_issue issues the loads branch-free (addresses
// clamped into the tensor: lanes past an edge compute garbage they never store) BEFORE the first K tile's loads;
// epi_start_apply folds them into the accumulators after that tile's barrier, so their latency hides behind the tile
// instead of preceding or following the K loop.  Rows are linear (m_base + i*16 + fr); an affine residual's per-image
// vectors are taken at row m_tile0 (epi_class guarantees one image per 128-row tile).
template <int TN>
struct EpiStart {
    f32x4 b4[TN], rs4[TN], rh4[TN];
};
template <int TM, int TN, int EPI>
__device__ __forceinline__ void epi_start_issue(const uwir_gemm_args& g, f32x4 (&acc)[TM][TN], EpiStart<TN>& st, int m_base,
                                                int M, int m_tile0, int n_base, int fr, int fq) {
    if constexpr (EPI != EPI_FULL) {
        const int N = g.N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = min(n_base + fq * 16 + j * 4, N - 4);
            st.b4[j] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI == EPI_RES) {
                if (g.res_scale) {
                    const size_t bo = (size_t)(m_tile0 / (g.OH * g.OW)) * N + n;
                    st.rs4[j] = *reinterpret_cast<const f32x4*>(g.res_scale + bo);
                    st.rh4[j] = *reinterpret_cast<const f32x4*>(g.res_shift + bo);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const size_t ro = (size_t)min(m_base + i * 16 + fr, M - 1) * g.ldr + n;
                    if (g.res_dtype == UWIR_BF16) {              // block-uniform
                        const float4 t = load4((const bf16_t*)g.res + ro);
                        acc[i][j] = (f32x4){t.x, t.y, t.z, t.w};
                    } else {
                        acc[i][j] = *reinterpret_cast<const f32x4*>((const float*)g.res + ro);
                    }
                }
            }
        }
    }
}
template <int TM, int TN, int EPI>
__device__ __forceinline__ void epi_start_apply(const uwir_gemm_args& g, f32x4 (&acc)[TM][TN], const EpiStart<TN>& st) {
    if constexpr (EPI != EPI_FULL) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (EPI == EPI_RES) {
                    if (g.res_scale) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][e] = fmaf(acc[i][j][e], st.rs4[j][e], st.rh4[j][e]);
                    }
                    acc[i][j] += st.b4[j];
                } else {
                    acc[i][j] = st.b4[j];
                }
            }
    }
}
template <int TM, int TN, int EPI = EPI_FULL>
__device__ __forceinline__ void gemm_epilogue(const uwir_gemm_args& g, const f32x4 (&acc)[TM][TN], const int (&mrow)[TM],
                                              const int (&mcnt)[TM], int n_base, int fr, int fq) {
    static_assert(TN == 4, "the channel permutation assumes 4 MFMA tiles per 64-channel wave block");
    const int N = g.N;
    const int OHW = g.OH * g.OW;
    const int Cout4 = N / 4;  // depth-to-space: channels per sub-pixel
    if constexpr (EPI != EPI_FULL) {
        // Fast path of the lean classes (block-uniform test): rows in the plain layout, 8-channel groups whole.  The general
        // loop below spends ~700 VALU instructions per wave and tile on row / image index arithmetic, edge handling and
        // layout dispatch -- as much issue time as the 96 MFMAs of a K = 192 tile; here a tile's stores cost ~150.
        // Same operations in the same order on every value, so results are bit-identical to the general loop.
        const bool o16 = g.out_dtype == UWIR_BF16;
        if (g.out && g.store_mode == 0 && N % 8 == 0 && g.ldo % 8 == 0 && ((uintptr_t)g.out & 15) == 0) {
            const int nq = n_base + fq * 16;
            const int bld = g.batch_ld ? g.batch_ld : N;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const size_t orow = (size_t)(mrow[i] + fr) * g.ldo + nq;
                const bool row_ok = fr < mcnt[i];
                // (per-sample bias: the image of THIS lane's row; lanes past the tile's rows read image 0 and store nothing)
                const float* bb = g.batch_bias ? g.batch_bias + (size_t)((row_ok ? mrow[i] + fr : 0) / OHW) * bld + nq : nullptr;
#pragma unroll
                for (int jp = 0; jp < TN; jp += 2) {
                    f32x4 v[2] = {acc[i][jp], acc[i][jp + 1]};
                    if (g.batch_bias && nq + jp * 4 < N) {
                        v[0] += *reinterpret_cast<const f32x4*>(bb + jp * 4);
                        v[1] += *reinterpret_cast<const f32x4*>(bb + jp * 4 + 4);
                    }
#define UWIR_EPI_ACT(EXPR)                                                 \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                          \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) { const float x = v[j][e]; v[j][e] = (EXPR); }
                    switch (g.act) {
                        case UWIR_ACT_RELU: UWIR_EPI_ACT(x > 0.f ? x : 0.f) break;
                        case UWIR_ACT_GELU: UWIR_EPI_ACT(act_apply(x, UWIR_ACT_GELU)) break;
                        case UWIR_ACT_SIGMOID: UWIR_EPI_ACT(sigmoid_fast(x)) break;
                        case UWIR_ACT_SILU: UWIR_EPI_ACT(x * sigmoid_fast(x)) break;
                        default: break;
                    }
#undef UWIR_EPI_ACT
                    const bool col_ok = nq + jp * 4 < N;   // N % 8 == 0: the pair of quads is whole or absent
                    if (g.col_scale && col_ok) {
                        v[0] *= *reinterpret_cast<const f32x4*>(g.col_scale + nq + jp * 4);
                        v[1] *= *reinterpret_cast<const f32x4*>(g.col_scale + nq + jp * 4 + 4);
                    }
                    if (row_ok && col_ok) {
                        if (o16) {
                            bf16x8 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)v[0][e]; o[4 + e] = (bf16_t)v[1][e]; }
                            *reinterpret_cast<bf16x8*>((bf16_t*)g.out + orow + jp * 4) = o;
                        } else {
                            *reinterpret_cast<f32x4*>((float*)g.out + orow + jp * 4) = v[0];
                            *reinterpret_cast<f32x4*>((float*)g.out + orow + jp * 4 + 4) = v[1];
                        }
                    }
                }
            }
            return;
        }
    }
    EpiCols<TN> cols;
    if (EPI == EPI_FULL) epi_cols<TN>(g, n_base, fq, cols);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        f32x4 vv[TN];
        {
            const int m = fr < mcnt[i] ? mrow[i] + fr : 0;
            if (fr < mcnt[i]) epi_values<TN, EPI>(g, acc[i], cols, (size_t)m, m / OHW, n_base, fq, vv);
        }
        if (fr >= mcnt[i]) continue;
        int m = mrow[i] + fr;
        int b = m / OHW;
        size_t out_row = (size_t)m;
        int rem = m - b * OHW, oy = rem / g.OW, ox = rem - oy * g.OW;
        if (!g.out) continue;
        // bf16 rows, plain layout: two 16-byte stores cover the lane's 16 channels (when 8-channel groups are whole
        // and the row base is 16-byte aligned); everything else goes out as 4-channel pieces
        if (EPI == EPI_FULL && g.store_mode == 2) {
            // GEGLU pairs: the lane's 16 channels are 8 values then their 8 gates (weight rows ordered that way by the
            // caller): out[m, (n_base + fq*16) / 2 + jj*4 + e] = value * gelu(gate), exact (erf) GELU
            const int n = n_base + fq * 16;
            if (n < N) {
                float o8[8];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o8[jj * 4 + e] = vv[jj][e] * act_apply(vv[jj + 2][e], UWIR_ACT_GELU);
                const size_t off = out_row * g.ldo + n / 2;
                if (g.out_dtype == UWIR_BF16) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)o8[e];
                    *reinterpret_cast<bf16x8*>((bf16_t*)g.out + off) = o;
                } else {
                    store4((float*)g.out + off, make_float4(o8[0], o8[1], o8[2], o8[3]));
                    store4((float*)g.out + off + 4, make_float4(o8[4], o8[5], o8[6], o8[7]));
                }
            }
            continue;
        }
        const bool wide = g.out_dtype == UWIR_BF16 && g.store_mode == 0 && (N % 8 == 0) && (g.ldo % 8 == 0) &&
                          (((uintptr_t)g.out & 15) == 0);
#pragma unroll
        for (int jp = 0; jp < TN; jp += 2) {
            int n = n_base + fq * 16 + jp * 4;
            if (wide) {
                if (n < N) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)vv[jp][e]; o[4 + e] = (bf16_t)vv[jp + 1][e]; }
                    *reinterpret_cast<bf16x8*>((bf16_t*)g.out + out_row * g.ldo + n) = o;
                }
                continue;
            }
#pragma unroll
            for (int j = jp; j < jp + 2; ++j) {
                int nj = n_base + fq * 16 + j * 4;
                if (nj >= N) continue;
                size_t off;
                if (EPI == EPI_FULL && g.store_mode == 1) {
                    int qd = nj / Cout4, co = nj - qd * Cout4;
                    int dy = qd >> 1, dx = qd & 1;
                    off = (((size_t)b * 2 * g.OH + 2 * oy + dy) * (2 * g.OW) + 2 * ox + dx) * g.ldo + co;
                } else {
                    off = out_row * g.ldo + nj;
                }
                float4 o = make_float4(vv[j][0], vv[j][1], vv[j][2], vv[j][3]);
                if (g.out_dtype == UWIR_F32) store4((float*)g.out + off, o);
                else store4((bf16_t*)g.out + off, o);
            }
        }
    }
}


template <typename TC, typename TA, int BN, int NB, int EPI = EPI_FULL>
__global__ void __launch_bounds__(256) conv_gemm_kernel(GemmP p) {
    using Cfg = MmaCfg<TC>;
    constexpr int BM = 128, BK = Cfg::BK, EPC = Cfg::EPC, BKP = BK + Cfg::PAD, CPR = BK / EPC;
    constexpr int ACH = BM * CPR / 256, BCH = BN * CPR / 256;
    constexpr int WAVES_N = BN / 64, WAVES_M = 4 / WAVES_N, WM = BM / WAVES_M, TM = WM / 16, TN = 4;
    constexpr int KSTEPS = BK / Cfg::KSTEP;

    __shared__ __attribute__((aligned(16))) TC sA[NB][BM * BKP];
    __shared__ __attribute__((aligned(16))) TC sB[NB][BN * BKP];

    const uwir_gemm_args& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware remap (bijective): blocks sharing an XCD (bid % 8) get a contiguous run of tiles,
    // so the N-tiles of one M-panel reuse the activation rows out of one L2.
    int bid = blockIdx.x;
    {
        int nwg = p.nblocks, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int m0 = (bid / p.tiles_n) * BM, n0 = (bid % p.tiles_n) * BN;
    const int M = p.M, K = p.K, N = g.N, C = g.C;
    const int OHW = g.OH * g.OW;
    const bool is1x1 = (g.KH == 1 && g.KW == 1 && g.stride == 1 && g.pad == 0);

    const TA* __restrict__ Aptr = (const TA*)g.A;
    const TC* __restrict__ Wptr = (const TC*)g.Wt;

    // ---- per-thread staging descriptors
    int a_b[ACH], a_y[ACH], a_x[ACH];
    bool a_ok[ACH];
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        int c = tid + 256 * i;
        int m = m0 + c / CPR;
        a_ok[i] = m < M;
        int mm = a_ok[i] ? m : 0;
        int b = mm / OHW, rem = mm - b * OHW;
        int oy = rem / g.OW, ox = rem - oy * g.OW;
        a_b[i] = b;
        a_y[i] = oy * g.stride - g.pad;
        a_x[i] = ox * g.stride - g.pad;
    }
    RawChunk<TA, EPC> ra[ACH];
    bool ra_in[ACH];
    int ra_c[ACH];
    RawChunk<TC, EPC> rb[BCH];

    // k -> (tap, channel): when C % BK == 0 a whole K tile lies inside ONE filter tap, so the two integer divisions
    // are wave-uniform scalar work done once per tile instead of per chunk on the vector ALU (the im2col address
    // arithmetic used to outweigh the tile's 32 MFMAs for k x k convs).
    const bool tap_uniform = !is1x1 && (C % BK == 0);
    auto load_regs = [&](int kt) {
        int tap_s = 0, ky_s = 0, kx_s = 0, ch_s = 0;
        if (tap_uniform) {
            const int k0 = kt * BK;
            tap_s = __builtin_amdgcn_readfirstlane(k0 / C);
            ch_s = k0 - tap_s * C;
            ky_s = __builtin_amdgcn_readfirstlane(tap_s / g.KW);
            kx_s = tap_s - ky_s * g.KW;
        }
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            int c = tid + 256 * i;
            int k = kt * BK + (c % CPR) * EPC;
            bool ok = a_ok[i] && k < K;
            int tap = 0, ch = k, ky, kx;
            if (tap_uniform) {
                ch = ch_s + (c % CPR) * EPC; ky = ky_s; kx = kx_s;
            } else {
                if (!is1x1) { tap = k / C; ch = k - tap * C; }
                ky = tap / g.KW; kx = tap - ky * g.KW;
            }
            int iy = a_y[i] + ky, ix = a_x[i] + kx;
            // ups = 1: the conv runs on the nearest-neighbour 2x upsampled image (never materialised)
            ok = ok && iy >= 0 && iy < (g.H << g.ups) && ix >= 0 && ix < (g.W << g.ups);
            ra_in[i] = ok;
            ra_c[i] = ch;
            if (ok) {
                const TA* src = Aptr + (((size_t)a_b[i] * g.H + (iy >> g.ups)) * g.W + (ix >> g.ups)) * g.lda + ch;
                if constexpr (sizeof(TA) == 4) {
                    ra[i].lo = *reinterpret_cast<const float4*>(src);
                    if constexpr (EPC == 8) ra[i].hi = *reinterpret_cast<const float4*>(src + 4);
                } else {
                    ra[i].v = *reinterpret_cast<const bf16x8*>(src);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            int c = tid + 256 * i;
            int n = n0 + c / CPR;
            int k = kt * BK + (c % CPR) * EPC;
            if (n < N && k < K) {
                const TC* src = Wptr + (size_t)n * K + k;
                if constexpr (sizeof(TC) == 4) rb[i].lo = *reinterpret_cast<const float4*>(src);
                else rb[i].v = *reinterpret_cast<const bf16x8*>(src);
            } else {
                if constexpr (sizeof(TC) == 4) rb[i].lo = make_float4(0, 0, 0, 0);
                else { bf16x8 z; for (int j = 0; j < 8; ++j) z[j] = (bf16_t)0.f; rb[i].v = z; }
            }
        }
    };

    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            int c = tid + 256 * i;
            float v[EPC];
            if (ra_in[i]) {
                if constexpr (sizeof(TA) == 4) {
                    v[0] = ra[i].lo.x; v[1] = ra[i].lo.y; v[2] = ra[i].lo.z; v[3] = ra[i].lo.w;
                    if constexpr (EPC == 8) { v[4] = ra[i].hi.x; v[5] = ra[i].hi.y; v[6] = ra[i].hi.z; v[7] = ra[i].hi.w; }
                } else {
#pragma unroll
                    for (int j = 0; j < EPC; ++j) v[j] = (float)ra[i].v[j];
                }
                if (g.a_scale) {
                    const float* sc = g.a_scale + (size_t)a_b[i] * C + ra_c[i];
                    const float* sh = g.a_shift + (size_t)a_b[i] * C + ra_c[i];
#pragma unroll
                    for (int j = 0; j < EPC; j += 4) {
                        float4 s4 = load4(sc + j), h4 = load4(sh + j);
                        v[j] = fmaf(v[j], s4.x, h4.x);
                        v[j + 1] = fmaf(v[j + 1], s4.y, h4.y);
                        v[j + 2] = fmaf(v[j + 2], s4.z, h4.z);
                        v[j + 3] = fmaf(v[j + 3], s4.w, h4.w);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < EPC; ++j) v[j] = 0.f;
            }
            TC* dst = &sA[buf][(c / CPR) * BKP + (c % CPR) * EPC];
            if constexpr (sizeof(TC) == 4) {
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
                *reinterpret_cast<bf16x8*>(dst) = o;
            }
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            int c = tid + 256 * i;
            TC* dst = &sB[buf][chan_perm(c / CPR) * BKP + (c % CPR) * EPC];  // channel nn lives in LDS row pi(nn)
            if constexpr (sizeof(TC) == 4) *reinterpret_cast<float4*>(dst) = rb[i].lo;
            else *reinterpret_cast<bf16x8*>(dst) = rb[i].v;
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * 64;
    const int fr = lane & 15, fq = lane >> 4;
    EpiStart<TN> est;
    epi_start_issue<TM, TN, EPI>(g, acc, est, m0 + wm0, M, m0, n0 + wn0, fr, fq);   // (lean classes are linear-row launches)

    auto compute = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            if constexpr (sizeof(TC) == 2) {
                bf16x8 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i] = *reinterpret_cast<const bf16x8*>(&sA[buf][(wm0 + i * 16 + fr) * BKP + kk * 32 + fq * 8]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[j] = *reinterpret_cast<const bf16x8*>(&sB[buf][(wn0 + j * 16 + fr) * BKP + kk * 32 + fq * 8]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            } else {
                float fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = sA[buf][(wm0 + i * 16 + fr) * BKP + kk * 4 + fq];
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = sB[buf][(wn0 + j * 16 + fr) * BKP + kk * 4 + fq];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
        }
    };

    const int nk = (K + BK - 1) / BK;
    if constexpr (NB == 2) {
        load_regs(0);
        store_lds(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_regs(kt + 1);
            compute(kt & 1);
            if (kt + 1 < nk) store_lds((kt + 1) & 1);
            __syncthreads();
        }
    } else {  // one LDS buffer, register prefetch of the next tile across the compute
        load_regs(0);
        for (int kt = 0; kt < nk; ++kt) {
            store_lds(0);
            __syncthreads();
            if (kt + 1 < nk) load_regs(kt + 1);
            if (kt == 0) epi_start_apply<TM, TN, EPI>(g, acc, est);
            compute(0);
            __syncthreads();
        }
    }

    int mrow[TM], mcnt[TM];
    linear_rows<TM>(mrow, mcnt, m0 + wm0, M);
    gemm_epilogue<TM, TN, EPI>(g, acc, mrow, mcnt, n0 + wn0, fr, fq);
}

// ------------------------------------------------------------------------------------------
// Fast path: plain bf16 x bf16 GEMM (1x1 conv, bf16 activations, no prologue affine, K % 64 == 0).
// Both operands stream global -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4): no staging
// registers, no conversion pass.  The LDS image is lane-linear ([row][8 chunks of 16 B], 128-B rows),
// so the bank-conflict swizzle (chunk ^= row & 7) is applied on the per-lane SOURCE address and
// again on the fragment read.  One barrier per K tile: tile t+1 is in flight while tile t feeds
// 32 v_mfma_f32_16x16x32_bf16 per wave.
// ------------------------------------------------------------------------------------------
// NBUF = 2: tile t+1 streams in while tile t computes (64 KiB LDS, 2 blocks / CU).
// NBUF = 1: load -> compute per tile (32 KiB LDS, 4 blocks / CU): twice the resident blocks hide more latency than
// the intra-block overlap does (see launch_gemm) -- this is the variant in use.
template <int NBUF, int EPI = EPI_FULL>
__global__ void __launch_bounds__(256, (EPI == EPI_LEAN ? 4 : EPI == EPI_RES ? 3 : 2)) gemm_bf16_glds_kernel(GemmP p) {
    constexpr int BM = 128, BN = 128, BK = 64, TM = 4, TN = 4;
    constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile
    __shared__ __attribute__((aligned(1024))) char smem[2 * NBUF * TILE_BYTES];  // [buf][A|B]

    const uwir_gemm_args& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x, slice = 0;
    if (p.splitk > 1) {
        slice = bid % p.splitk;
        bid /= p.splitk;
    } else {
        int nwg = p.nblocks, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int m0 = (bid / p.tiles_n) * BM, n0 = (bid % p.tiles_n) * BN;
    const int M = p.M, K = p.K, N = g.N;
    const bf16_t* __restrict__ A = (const bf16_t*)g.A;
    // per-image weight panels (uwir_fold_affine_weights): a 128-row tile lies inside ONE image (checked on the host)
    const bf16_t* __restrict__ Wt = (const bf16_t*)g.Wt + (g.w_batch_stride ? (size_t)(m0 / (g.OH * g.OW)) * (size_t)g.w_batch_stride : 0);

    // staging: wave w owns rows [32w, 32w+32) of both tiles, 4 LDS-DMA pieces of 8 rows each per operand.
    // lane -> (row-in-piece = lane/8, LDS chunk = lane%8) reads GLOBAL chunk (lane%8) ^ (row & 7).
    const int prow = lane >> 3, pch = lane & 7;
    const bf16_t* a_src[4];
    const bf16_t* b_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = wave * 32 + i * 8 + prow;
        int gch = pch ^ (r & 7);
        int am = min(m0 + r, M - 1), bn = min(n0 + chan_perm(r), N - 1);  // LDS row r holds channel pi(r); clamp: rows
                                                                          // past the edge are never stored
        a_src[i] = A + (size_t)am * g.lda + gch * 8;
        b_src[i] = Wt + (size_t)bn * K + gch * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* la = smem + buf * 2 * TILE_BYTES + (wave * 32) * 128;
        char* lb = la + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(la + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(lb + i * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int fr = lane & 15, fq = lane >> 4;
    EpiStart<TN> est;
    epi_start_issue<TM, TN, EPI>(g, acc, est, m0 + wm0, M, m0, n0 + wn0, fr, fq);
    // fragment byte offsets inside a tile: row*128 + ((chunk ^ (row&7)) * 16); row&7 == fr&7 for every i
    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_off[i] = (wm0 + i * 16 + fr) * 128;
#pragma unroll
    for (int j = 0; j < TN; ++j) b_off[j] = (wn0 + j * 16 + fr) * 128;
    const int sw = fr & 7;

    const int nk = K / BK;
    auto compute = [&](int buf) {
        const char* la = smem + buf * 2 * TILE_BYTES;
        const char* lb = la + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[TM], fb[TN];
            const int ch = ((kk * 4 + fq) ^ sw) * 16;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(la + a_off[i] + ch);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(lb + b_off[j] + ch);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    };
    int kt0 = 0, kt1 = nk;
    if (p.splitk > 1) {
        kt0 = (int)((long long)nk * slice / p.splitk);
        kt1 = (int)((long long)nk * (slice + 1) / p.splitk);
    }
    if constexpr (NBUF == 2) {
        stage(0, kt0);
        __syncthreads();  // hipcc drains the LDS-DMA (vmcnt(0)) ahead of the barrier
        for (int kt = kt0; kt < kt1; ++kt) {
            const int buf = (kt - kt0) & 1;
            if (kt + 1 < kt1) stage(buf ^ 1, kt + 1);
            compute(buf);
            __syncthreads();
        }
    } else {
        stage(0, kt0);
        __syncthreads();
        epi_start_apply<TM, TN, EPI>(g, acc, est);
        compute(0);
        __syncthreads();
        for (int kt = kt0 + 1; kt < kt1; ++kt) {
            stage(0, kt);
            __syncthreads();
            compute(0);
            __syncthreads();
        }
    }
    if (p.splitk > 1) {
        splitk_dump<TM, TN>(p, acc, bid, slice);
        return;
    }
    int mrow[TM], mcnt[TM];
    linear_rows<TM>(mrow, mcnt, m0 + wm0, M);
    gemm_epilogue<TM, TN, EPI>(g, acc, mrow, mcnt, n0 + wn0, fr, fq);
}

// second launch of a split-K GEMM: one block per (output tile, 16-row MFMA group i), same thread <-> element mapping as
// the glds kernel; 4 x splitk independent 16-byte loads per thread
__global__ void __launch_bounds__(256) gemm_splitk_epilogue_kernel(GemmP p) {
    constexpr int TM = 4, TN = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x >> 2, i = blockIdx.x & 3;
    const int m0 = (tile / p.tiles_n) * 128, n0 = (tile % p.tiles_n) * 128;
    const f32x4* part = (const f32x4*)p.a.workspace;
    const size_t per_tile = (size_t)TM * TN * 256;
    f32x4 v[SPLITK_MAX][TN];
#pragma unroll
    for (int sl = 0; sl < SPLITK_MAX; ++sl)
#pragma unroll
        for (int j = 0; j < TN; ++j)
            v[sl][j] = sl < p.splitk ? part[((size_t)sl * p.nblocks + tile) * per_tile + (size_t)(i * TN + j) * 256 + tid]
                                     : (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 acc[1][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        f32x4 sum = v[0][j];
#pragma unroll
        for (int sl = 1; sl < SPLITK_MAX; ++sl) sum += v[sl][j];  // slice order; absent slices add +0
        acc[0][j] = sum;
    }
    int mrow[1], mcnt[1];
    linear_rows<1>(mrow, mcnt, m0 + (wave >> 1) * 64 + i * 16, p.M);
    gemm_epilogue<1, TN>(p.a, acc, mrow, mcnt, n0 + (wave & 1) * 64, lane & 15, lane >> 4);
}

static bool glds_eligible(const uwir_gemm_args& g, int K) {
    return g.compute == UWIR_BF16 && g.a_dtype == UWIR_BF16 && g.KH == 1 && g.KW == 1 && g.stride == 1 && g.pad == 0 &&
           !g.a_scale && K % 64 == 0 && g.N > 64 && g.lda % 8 == 0 && !g.ups;
}


// ------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolutions (the IR-SDE UNet's ResBlocks: 236 of its 266 GF per step) as an implicit GEMM over
// LDS-staged HALO tiles.  The generic kernel above gathers an im2col K tile per filter tap, i.e. it pulls every input
// pixel through L2 -> registers -> (convert) -> LDS nine times; at N = 64 that traffic, not the MFMA, bounds it.  Here
// a block owns an 8 x 16 pixel output tile: per 32-channel chunk the (8+2) x (16+2) input tile is staged ONCE (affine +
// bf16 conversion on the way in, zero padding, nearest-2x upsampling by source addressing), and the nine taps are nine
// shifted views of it: the B operand of tap (ky, kx) for pixel (y, x) is LDS row (y+ky)*18 + x+kx.  Weights stream
// through LDS three taps (one filter row) at a time.  Register prefetch of the next stage overlaps the MFMAs.
// ------------------------------------------------------------------------------------------
#define HC_TW 16
#define HC_PW (HC_TW + 2)
#define HC_BK 32
#define HC_STR 32                        // LDS row = one pixel's (or channel's) 32-channel chunk: 64 bytes, four 16-byte pieces
#define HC_PWL 24                        // LDS rows per halo-tile image row (18 used): a multiple of 8, so the swizzle of
                                         // row (y + ky) * 24 + x + kx depends on x + kx only and (y + ky) stays an immediate
// piece p of LDS row r sits at piece position p ^ hc_swz(r).  ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31} (+32): with lane = 16 * piece + row this placement is conflict-free from any first row; the padded
// 80-byte rows it replaces were 2-way conflicted on every fragment read (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE)
__device__ __forceinline__ int hc_swz(int row) { return (row >> 1) & 3; }
#define HC_TG 3                          // taps per weight stage

template <typename TA, int BN, int HC_TH>
__global__ void __launch_bounds__(256) conv3x3_halo_kernel(GemmP p) {
    constexpr int HC_NPIX = (HC_TH + 2) * HC_PW;               // 180 (8-row tile) or 324 (16-row tile)
    constexpr int WAVES_N = BN / 64, WAVES_M = 4 / WAVES_N, TM = HC_TH / WAVES_M, TN = 4;
    constexpr int A_ITEMS = HC_NPIX * (HC_BK / 8);                    // 16-byte (8 x bf16) pieces of the halo tile
    constexpr int A_IT = (A_ITEMS + 255) / 256;
    constexpr int W_ITEMS = HC_TG * BN * (HC_BK / 8);
    constexpr int W_IT = (W_ITEMS + 255) / 256;
    __shared__ __attribute__((aligned(16))) bf16_t sA[(HC_TH + 2) * HC_PWL * HC_STR];
    __shared__ __attribute__((aligned(16))) bf16_t sW[HC_TG * BN * HC_STR];

    const uwir_gemm_args& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    {
        int nwg = p.nblocks, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int OH = g.OH, OW = g.OW, C = g.C, N = g.N, K = p.K;
    const int tiles_x = (OW + HC_TW - 1) / HC_TW, tiles_y = (OH + HC_TH - 1) / HC_TH;
    const int n0 = (bid % p.tiles_n) * BN;
    int mt = bid / p.tiles_n;
    const int tx = mt % tiles_x; mt /= tiles_x;
    const int ty = mt % tiles_y;
    const int b = mt / tiles_y;
    const int y0 = ty * HC_TH, x0 = tx * HC_TW;
    const TA* __restrict__ Aptr = (const TA*)g.A;
    const bf16_t* __restrict__ Wptr = (const bf16_t*)g.Wt;

    // ---- per-thread staging descriptors, fixed over the stage loop.  Everything that does not depend on the stage is
    // computed ONCE (the first version redid the index arithmetic and 64-bit address products per stage and predicated
    // every load: ~180 VALU instructions and 28 exec-mask branches around 24 MFMAs per stage, VALU-active 59 % of the
    // kernel).  Loads are unconditional from clamped addresses; zero padding is a select at the LDS store.
    static_assert(W_ITEMS % 256 == 0, "weight stage must be whole iterations");
    const TA* a_ptr[A_IT];
    int a_lds[A_IT];          // LDS element offset of the piece, or -1 when this thread has no item in iteration `it`
    unsigned a_ok = 0;        // bit it: the pixel is inside the image (else the piece is zero padding)
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int item = tid + 256 * it;
        const int pix = item / (HC_BK / 8), piece = item % (HC_BK / 8);
        const int iy = y0 + pix / HC_PW - 1, ix = x0 + pix % HC_PW - 1;   // coordinates in the (upsampled) input image
        const bool ok = item < A_ITEMS && iy >= 0 && iy < OH && ix >= 0 && ix < OW;
        if (ok) a_ok |= 1u << it;
        const int cy = ok ? iy : 0, cx = ok ? ix : 0;
        a_ptr[it] = Aptr + (((size_t)b * g.H + (cy >> g.ups)) * g.W + (cx >> g.ups)) * g.lda + piece * 8;
        const int lrow = (pix / HC_PW) * HC_PWL + pix % HC_PW;
        a_lds[it] = item < A_ITEMS ? lrow * HC_STR + (piece ^ hc_swz(lrow)) * 8 : -1;
    }
    const bf16_t* w_ptr[W_IT];
    int w_lds[W_IT];
    unsigned w_ok = 0;
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int item = tid + 256 * it;
        const int row = item / (HC_BK / 8), piece = item % (HC_BK / 8);   // row = tap_local * BN + channel
        const int tapl = row / BN, nl = row % BN, nn = n0 + nl;
        if (nn < N) w_ok |= 1u << it;
        w_ptr[it] = Wptr + (size_t)min(nn, N - 1) * K + tapl * C + piece * 8;
        const int lrow = tapl * BN + chan_perm(nl);                        // LDS row pi(n)
        w_lds[it] = lrow * HC_STR + (piece ^ hc_swz(lrow)) * 8;
    }
    RawChunk<TA, 8> ra[A_IT];
    bf16x8 rw[W_IT];

    const int nchunks = C / HC_BK;
    auto load_a = [&](int c0) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const TA* src = a_ptr[it] + c0;
            if constexpr (sizeof(TA) == 4) {
                ra[it].lo = *reinterpret_cast<const float4*>(src);
                ra[it].hi = *reinterpret_cast<const float4*>(src + 4);
            } else {
                ra[it].v = *reinterpret_cast<const bf16x8*>(src);
            }
        }
    };
    auto load_w = [&](int woff) {   // woff = (filter row * 3) * C + c0: block-uniform
#pragma unroll
        for (int it = 0; it < W_IT; ++it) rw[it] = *reinterpret_cast<const bf16x8*>(w_ptr[it] + woff);
    };
    auto store_a = [&](int c0) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            if (a_lds[it] < 0) continue;
            float v[8];
            if constexpr (sizeof(TA) == 4) {
                v[0] = ra[it].lo.x; v[1] = ra[it].lo.y; v[2] = ra[it].lo.z; v[3] = ra[it].lo.w;
                v[4] = ra[it].hi.x; v[5] = ra[it].hi.y; v[6] = ra[it].hi.z; v[7] = ra[it].hi.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (float)ra[it].v[j];
            }
            if (g.a_scale) {   // block-uniform
                const int ch = c0 + (tid & 3) * 8;   // piece = item % 4 = tid % 4
                const float* sc = g.a_scale + (size_t)b * C + ch;
                const float* sh = g.a_shift + (size_t)b * C + ch;
#pragma unroll
                for (int j = 0; j < 8; j += 4) {
                    float4 s4 = load4(sc + j), h4 = load4(sh + j);
                    v[j] = fmaf(v[j], s4.x, h4.x);
                    v[j + 1] = fmaf(v[j + 1], s4.y, h4.y);
                    v[j + 2] = fmaf(v[j + 2], s4.z, h4.z);
                    v[j + 3] = fmaf(v[j + 3], s4.w, h4.w);
                }
            }
            const bool ok = (a_ok >> it) & 1;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(ok ? v[j] : 0.f);   // padding pads the NORMALISED image
            *reinterpret_cast<bf16x8*>(&sA[a_lds[it]]) = o;
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            bf16x8 o = rw[it];
            if (!((w_ok >> it) & 1)) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16_t)0.f;
            }
            *reinterpret_cast<bf16x8*>(&sW[w_lds[it]]) = o;
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    const int wr0 = (wave / WAVES_N) * TM, wn0 = (wave % WAVES_N) * 64;   // first output row / channel of this wave
    const bf16_t* a_rd[HC_TG];                                            // + (i + ky) * HC_PWL * HC_STR: immediates
#pragma unroll
    for (int t = 0; t < HC_TG; ++t) {
        const int r = wr0 * HC_PWL + fr + t;
        a_rd[t] = &sA[r * HC_STR + (fq ^ hc_swz(r)) * 8];
    }
    const bf16_t* w_rd = &sW[(wn0 + fr) * HC_STR + (fq ^ hc_swz(wn0 + fr)) * 8];   // + (kx * BN + j * 16) * HC_STR

    // filter row ky = grp (compile-time): taps grp*3 .. grp*3+2
    auto compute = [&](auto grp_tag) {
        constexpr int grp = decltype(grp_tag)::value;
#pragma unroll
        for (int tapl = 0; tapl < HC_TG; ++tapl) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const bf16x8*>(a_rd[tapl] + (i + grp) * HC_PWL * HC_STR);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(w_rd + (tapl * BN + j * 16) * HC_STR);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    };

    load_a(0);
    load_w(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int c0 = chunk * HC_BK;
        // ---- filter row 0 (+ the chunk's halo tile)
        if (chunk) __syncthreads();       // every wave is done with the previous stage's LDS image
        store_a(c0);
        store_w();
        __syncthreads();
        load_w(1 * HC_TG * C + c0);
        compute(std::integral_constant<int, 0>{});
        // ---- filter row 1
        __syncthreads();
        store_w();
        __syncthreads();
        load_w(2 * HC_TG * C + c0);
        if (chunk + 1 < nchunks) load_a(c0 + HC_BK);   // two compute stages ahead of its use
        compute(std::integral_constant<int, 1>{});
        // ---- filter row 2
        __syncthreads();
        store_w();
        __syncthreads();
        if (chunk + 1 < nchunks) load_w(c0 + HC_BK);
        compute(std::integral_constant<int, 2>{});
    }

    int mrow[TM], mcnt[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int oy = y0 + wr0 + i;
        mrow[i] = (b * OH + oy) * OW + x0;
        mcnt[i] = oy < OH ? OW - x0 : 0;
    }
    gemm_epilogue<TM, TN>(g, acc, mrow, mcnt, n0 + wn0, fr, fq);
}

// ------------------------------------------------------------------------------------------
// The same halo-tile convolution for bf16 activations with no prologue affine (every 3x3 conv of the IR-SDE UNet once its
// activation stream is bf16), with BOTH operands streamed global -> LDS by LDS-DMA.  The register-staged kernel above
// holds the next stage's operands in VGPRs across a __syncthreads pair (whose vmcnt(0) ends every prefetch at the next
// barrier): compiled without its MFMAs it still took 70 % of its time (32 x 32 maps, C = 512: 39 of 55 us), the load ->
// LDS store -> barrier chain of each filter row being exposed by the 2 resident blocks per CU of the deep levels.  Here
//   * weights: a ring of NS stages (one filter row x 32 channels x 64 output channels = 12 KiB, 3 DMA instructions per
//     thread) filled NS - 1 stages ahead;
//   * activations: the chunk's (TH + 2) x 18 halo tile (64 B per pixel), double buffered, filled one chunk (3 stages) ahead;
//     padding pixels load a clamped address and are zeroed in LDS by the lane that owns them once its DMA has landed;
//   * ONE raw s_barrier per stage, behind a COUNTED s_waitcnt vmcnt (DMA completes in issue order; the counts below are
//     the instructions issued after the stage that is about to be read);
// LDS image: 64-byte rows, piece p of row r at position p ^ hc_swz(r), image rows 20 LDS rows apart (row + 20 k flips
// bit 1 of the swizzle for odd k: two read pointers per tap).  Block = 4 waves, TH x 16 pixels x 64 channels.
// ------------------------------------------------------------------------------------------
#define CD_PWL 20
__device__ __forceinline__ void cd_dma16(const void* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase)
                 : "memory");
}
template <int N> __device__ __forceinline__ void cd_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PWL = LDS rows per image row of the halo tile: 20 (swizzle period 2 along the tile rows, every wave issues the same
// number of halo pieces) or 18 (no padding rows: period 4, 21 pieces for 16 + 2 rows -- wave 0 issues one more than the
// others, so its counted waits differ by one -- which is what lets 16-row tiles keep a ring of 3 at two blocks per CU).
template <int TH, int NS, int PWL>
__global__ void __launch_bounds__(256) conv3x3_dma_kernel(GemmP p) {
    constexpr int BN = 64, TM = TH / 4, TN = 4, KV = PWL == 20 ? 2 : 4;
    constexpr int A_PIECES = PWL == 20 ? ((TH + 2) * PWL + 63) / 64 * 4 : ((TH + 2) * PWL + 15) / 16;   // 16-row DMA pieces
    constexpr int A_IT = (A_PIECES + 3) / 4, W_IT = 3;           // DMA instructions per thread (wave 0): per chunk / per stage
    constexpr bool RAGGED = A_PIECES % 4 != 0;                   // waves 1-3 issue A_IT - 1 pieces
    static_assert(!RAGGED || A_PIECES % 4 == 1, "only wave 0 may carry the extra piece");
    constexpr int A_BYTES = A_PIECES * 1024, W_BYTES = 3 * BN * 64;
    __shared__ __attribute__((aligned(1024))) char smem[2 * A_BYTES + NS * W_BYTES];

    const uwir_gemm_args& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    {
        int nwg = p.nblocks, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int OH = g.OH, OW = g.OW, C = g.C, K = p.K;
    const int tiles_x = (OW + HC_TW - 1) / HC_TW, tiles_y = (OH + TH - 1) / TH;
    const int n0 = (bid % p.tiles_n) * BN;
    int mt = bid / p.tiles_n;
    const int tx = mt % tiles_x; mt /= tiles_x;
    const int ty = mt % tiles_y;
    const int b = mt / tiles_y;
    const int y0 = ty * TH, x0 = tx * HC_TW;

    // ---- DMA descriptors.  Instruction `it` of wave w fills LDS rows (it * 4 + w) * 16 .. + 15, lane = row * 4 + position
    unsigned a_off[A_IT], a_pad = 0;   // bit it of a_pad: this lane's piece is padding of a pixel that is read
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int rho = (it * 4 + wave) * 16 + (lane >> 2);       // (a piece past A_PIECES is never issued)
        const int py = rho / PWL, px = rho % PWL;
        const int iy = y0 + py - 1, ix = x0 + px - 1;
        const bool used = py < TH + 2 && px < HC_PW && it * 4 + wave < A_PIECES;
        const bool in = iy >= 0 && iy < OH && ix >= 0 && ix < OW;
        if (used && !in) a_pad |= 1u << it;
        const int cy = used && in ? iy : 0, cx = used && in ? ix : 0;
        a_off[it] = ((unsigned)((b * g.H + (cy >> g.ups)) * g.W + (cx >> g.ups)) * (unsigned)g.lda +
                     (unsigned)(((lane & 3) ^ hc_swz(rho)) * 8)) * 2u;
    }
    unsigned w_off[W_IT];
    {
        const int r = wave * 16 + (lane >> 2);                     // LDS row of the 64-channel block: holds channel pi(r)
#pragma unroll
        for (int j = 0; j < W_IT; ++j)
            w_off[j] = ((unsigned)(n0 + chan_perm(r)) * (unsigned)K + (unsigned)(j * C + ((lane & 3) ^ hc_swz(r)) * 8)) * 2u;
    }
    const char* a_base = (const char*)g.A;
    const char* w_base = (const char*)g.Wt;
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    const unsigned lds_w = lds0 + 2 * A_BYTES;
    const int nchunks = C / HC_BK, S = 3 * nchunks;
    auto issue_w = [&](int s) {                                    // stage s = chunk s / 3, filter row s % 3
        const int c = s / 3, gr = s - 3 * c;
        const char* wb = w_base + (size_t)(gr * 3 * C + c * HC_BK) * 2;
        const unsigned dst = lds_w + (s % NS) * W_BYTES + wave * 1024;
#pragma unroll
        for (int j = 0; j < W_IT; ++j) cd_dma16(wb, w_off[j], dst + j * 4096);
    };
    auto issue_a = [&](int c) {
        const char* ab = a_base + (size_t)c * (HC_BK * 2);
        const unsigned dst = lds0 + (c & 1) * A_BYTES + wave * 1024;
#pragma unroll
        for (int it = 0; it < A_IT; ++it)
            if (!RAGGED || it + 1 < A_IT || wave == 0) cd_dma16(ab, a_off[it], dst + it * 4096);   // wave-uniform
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    const int wr0 = wave * TM;
    int a_rd[HC_TG][KV];                                           // byte offsets into an activation buffer, by (tap, (i + ky) % KV)
#pragma unroll
    for (int t = 0; t < HC_TG; ++t) {
        const int r = wr0 * PWL + fr + t;
#pragma unroll
        for (int kk = 0; kk < KV; ++kk) a_rd[t][kk] = r * 64 + (fq ^ hc_swz(r + PWL * kk)) * 16;   // swizzle of row r + PWL k: period KV
    }
    const int w_rd = fr * 64 + (fq ^ hc_swz(fr)) * 16;

    // The three taps of a stage are software-pipelined by hand: tap t + 1's fragments are read (into the other half of
    // fa / fb) between tap t's first MFMAs -- one ds_read_b128 behind each MFMA, pinned with sched_group_barrier -- so
    // that only the first tap of a stage waits for LDS.  Left to itself the scheduler put most reads directly in front
    // of the MFMA that consumes them (lgkmcnt(0) a dozen times per stage), which at two waves per SIMD showed: without
    // any DMA in the loop the kernel still ran at 45 % of the MFMA rate.
    auto compute = [&](auto grp_tag, const char* sa, const char* sw, auto&& refill) {
        constexpr int grp = decltype(grp_tag)::value;
        bf16x8 fa[2][TM], fb[2][TN];
        auto read_tap = [&](int tapl, int h) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[h][i] = *reinterpret_cast<const bf16x8*>(sa + a_rd[tapl][(i + grp) % KV] + (i + grp) * (PWL * 64));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[h][j] = *reinterpret_cast<const bf16x8*>(sw + w_rd + (tapl * BN + j * 16) * 64);
        };
        read_tap(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        refill();                                  // the stage's DMA issue (slow: ~100 cycles each) under the first reads' latency
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tapl = 0; tapl < HC_TG; ++tapl) {
            const int h = tapl & 1;
            if (tapl + 1 < HC_TG) read_tap(tapl + 1, h ^ 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[h][j], fa[h][i], acc[i][j], 0, 0, 0);
            if (tapl + 1 < HC_TG) {
#pragma unroll
                for (int k = 0; k < TM + TN; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
            }
        }
    };

    // one stage: wait (counted) for this stage's operands, barrier, refill the ring slot / buffer that the barrier freed,
    // compute.  WN = DMA instructions that may still be in flight when stage s is read (issued after it), steady state
    // and last chunk.
    auto stage = [&](auto grp_tag, int c, bool last) {
        constexpr int grp = decltype(grp_tag)::value;
        constexpr bool with_a = grp != 0 && (grp == 1 || NS >= 3);   // the next chunk's halo pieces are younger than stage s
        constexpr int steady = W_IT * (NS - 2) + (with_a ? A_IT : 0);
        constexpr int tail = grp == 0 ? W_IT * (NS - 2 < 2 ? NS - 2 : 2) : grp == 1 ? W_IT * (NS - 2 < 1 ? NS - 2 : 1) : 0;
        constexpr int steady_w = steady - ((with_a && RAGGED) ? 1 : 0) < 0 ? 0 : steady - ((with_a && RAGGED) ? 1 : 0);
        const int s = 3 * c + grp;
        if (last) cd_wait_vm<tail>();
        else if (wave == 0) cd_wait_vm<steady>();
        else cd_wait_vm<steady_w>();
        if (grp == 0 && a_pad) {                                   // zero the padding pieces this lane's DMA just wrote
            char* mine = smem + (c & 1) * A_BYTES + wave * 1024 + lane * 16;
#pragma unroll
            for (int it = 0; it < A_IT; ++it)
                if ((a_pad >> it) & 1) *reinterpret_cast<float4*>(mine + it * 4096) = make_float4(0.f, 0.f, 0.f, 0.f);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#if !defined(CD_ABL) || CD_ABL != 3
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
        compute(grp_tag, smem + (c & 1) * A_BYTES, smem + 2 * A_BYTES + (s % NS) * W_BYTES, [&]() {
#if !defined(CD_ABL) || CD_ABL != 2
            if (s + NS - 1 < S) issue_w(s + NS - 1);
            if (grp == 0 && !last) issue_a(c + 1);
#endif
        });
        __builtin_amdgcn_sched_barrier(0);
    };

    issue_a(0);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < S) issue_w(s);
    for (int c = 0; c < nchunks; ++c) {
        const bool last = c + 1 == nchunks;
        stage(std::integral_constant<int, 0>{}, c, last);
        stage(std::integral_constant<int, 1>{}, c, last);
        stage(std::integral_constant<int, 2>{}, c, last);
    }

    int mrow[TM], mcnt[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int oy = y0 + wr0 + i;
        mrow[i] = (b * OH + oy) * OW + x0;
        mcnt[i] = oy < OH ? OW - x0 : 0;
    }
    gemm_epilogue<TM, TN>(g, acc, mrow, mcnt, n0, fr, fq);
}

// ------------------------------------------------------------------------------------------
// Short-K form (C <= 64: the 64-channel levels of the IR-SDE UNet, K = 576): six filter-row stages per tile never fill a
// pipeline, and a block's prologue (first operands' latency) and epilogue (residual loads, stores) outweigh its 1.1 us of
// MFMAs -- 515 TF/s on the register-staged kernel at 4 blocks per CU, no better on the ring above.  Here the block is
// PERSISTENT (one per CU) and keeps the whole 64-channel weight slice in LDS (72 KiB, loaded once); only halo tiles
// stream.  Its 8 waves are two groups of 4 that walk separate lists of 8 x 16 pixel tiles with their own double-buffered
// halo tiles; one block-wide s_barrier per 32-channel chunk.  Group 0 runs compute -> epilogue inside a barrier interval,
// group 1 runs (previous tile's epilogue) -> compute, so each SIMD always has one wave issuing MFMAs while the other
// loads residuals / stores.  Operand waits: the next chunk's DMA is issued right after the barrier and awaited with
// vmcnt(0) at the END of the chunk's MFMAs, BEFORE the epilogue issues its stores (loads and stores share vmcnt and
// complete out of order between them: a counted wait with stores in flight proves nothing).
// ------------------------------------------------------------------------------------------
template <int NCH, bool RES32 = false>   // 32-channel chunks per tile: C = 32 * NCH; RES32: the residual rows are fp32
__global__ void __launch_bounds__(512, 1) conv3x3_resw_kernel(GemmP p) {
    constexpr int BN = 64, TH = 8, TM = 2, TN = 4, NB = 3;         // NB halo buffers per group: DMA runs two chunks ahead
    constexpr int A_PIECES = ((TH + 2) * CD_PWL + 15) / 16;        // 13 wave-instructions of 16 LDS rows (the last is partial)
    constexpr int A_BYTES = A_PIECES * 1024;
    constexpr int W_BYTES = NCH * 9 * BN * 64;                     // [chunk][tap][64 rows][64 B]
    constexpr int W_IT = W_BYTES / (512 * 16);
    static_assert(W_BYTES % (512 * 16) == 0, "weight image must be whole DMA rounds of the block");
    static_assert(W_BYTES + 2 * NB * A_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) char smem[W_BYTES + 2 * NB * A_BYTES];

    const uwir_gemm_args& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave8 >> 2, wave = wave8 & 3;
    const int OH = g.OH, OW = g.OW, C = g.C, K = p.K;
    const int tiles_x = (OW + HC_TW - 1) / HC_TW, tiles_y = (OH + TH - 1) / TH;
    const int mtiles = g.B * tiles_y * tiles_x;
    const int nt = blockIdx.x % p.tiles_n, mblk = blockIdx.x / p.tiles_n, nmblk = gridDim.x / p.tiles_n;
    const int n0 = nt * BN;
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    const unsigned lds_a = lds0 + W_BYTES + grp * NB * A_BYTES;
    char* sa_base = smem + W_BYTES + grp * NB * A_BYTES;

    // ---- the weight slice: LDS row r of (chunk c, tap t) holds channel n0 + pi(r), pieces swizzled by the row
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int slot = it * 8 + wave8;                           // 1 KiB = 16 rows
        const int img = slot >> 2, r = (slot & 3) * 16 + (lane >> 2);   // img = c * 9 + t
        const int c = img / 9, t = img - 9 * c;
        const unsigned off = ((unsigned)(n0 + chan_perm(r)) * (unsigned)K + (unsigned)(t * C + c * HC_BK + ((lane & 3) ^ hc_swz(r)) * 8)) * 2u;
        cd_dma16(g.Wt, off, lds0 + slot * 1024);
    }

    const int fr = lane & 15, fq = lane >> 4;
    const int wr0 = wave * TM;
    int a_rd[HC_TG][2];
#pragma unroll
    for (int t = 0; t < HC_TG; ++t) {
        const int r = wr0 * CD_PWL + fr + t;
        a_rd[t][0] = r * 64 + (fq ^ hc_swz(r)) * 16;
        a_rd[t][1] = r * 64 + (fq ^ hc_swz(r) ^ 2) * 16;
    }
    const int w_rd = fr * 64 + (fq ^ hc_swz(fr)) * 16;

    f32x4 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto compute_chunk = [&](const char* sa, const char* sw) {      // 3 filter rows x 3 taps of one 32-channel chunk
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int tapl = 0; tapl < HC_TG; ++tapl) {
                bf16x8 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i] = *reinterpret_cast<const bf16x8*>(sa + a_rd[tapl][(i + ky) & 1] + (i + ky) * (CD_PWL * 64));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[j] = *reinterpret_cast<const bf16x8*>(sw + w_rd + ((ky * 3 + tapl) * BN + j * 16) * 64);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
    };

    // this group's tile list: (mblk * 2 + grp) + k * 2 * nmblk, k = 0, 1, ..; chunk sequence number q = k * NCH + c
    auto tile_of = [&](int k) {
        const long long t = (long long)(mblk * 2 + grp) + (long long)k * 2 * nmblk;
        return t < mtiles ? (int)t : -1;
    };
    struct Tile { int b, y0, x0; };
    auto coords = [&](int t) {
        Tile r;
        r.x0 = (t % tiles_x) * HC_TW;
        t /= tiles_x;
        r.y0 = (t % tiles_y) * TH;
        r.b = t / tiles_y;
        return r;
    };
    // wave w issues pieces w, w + 4, ..: wave 0 one more than the others (13 pieces)
    // DMA of chunk q (tile list position q / NCH) into buffer q % NB; returns the mask of this lane's padding pieces
    auto issue_a = [&](int q) -> unsigned {
        const int t = tile_of(q / NCH), c = q % NCH;
        if (t < 0) return 0u;
        const Tile tc = coords(t);
        unsigned pad = 0;
        const char* ab = (const char*)g.A + (size_t)c * (HC_BK * 2);
        const unsigned dst = lds_a + (q % NB) * A_BYTES;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int piece = it * 4 + wave;
            if (piece < A_PIECES) {                                // wave-uniform
                const int rho = piece * 16 + (lane >> 2);
                const int py = rho / CD_PWL, px = rho % CD_PWL;
                const int iy = tc.y0 + py - 1, ix = tc.x0 + px - 1;
                const bool used = py < TH + 2 && px < HC_PW;
                const bool in = iy >= 0 && iy < OH && ix >= 0 && ix < OW;
                if (used && !in) pad |= 1u << it;
                const int cy = used && in ? iy : 0, cx = used && in ? ix : 0;
                const unsigned off = ((unsigned)((tc.b * g.H + (cy >> g.ups)) * g.W + (cx >> g.ups)) * (unsigned)g.lda +
                                      (unsigned)(((lane & 3) ^ hc_swz(rho)) * 8)) * 2u;
                cd_dma16(ab, off, dst + piece * 1024);
            }
        }
        return pad;
    };
    auto zero_pad = [&](unsigned pad, int q) {                      // after the chunk's DMA has landed, before the barrier
        if (pad) {
            char* mine = sa_base + (q % NB) * A_BYTES + wave * 1024 + lane * 16;
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if ((pad >> it) & 1) *reinterpret_cast<float4*>(mine + it * 4096) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // chunk q + 1 has landed: at most the pieces of chunk q + 2 (the only younger LOADS) may be in flight.  Loads retire in
    // order among themselves, so stores still in flight can only make this wait longer, never shorter.
    auto wait_next = [&](bool two_ahead_issued) {
        if (!two_ahead_issued) cd_wait_vm<0>();
        else if (wave == 0) cd_wait_vm<4>();
        else cd_wait_vm<3>();
    };
    // ---- epilogue with its operands PREFETCHED.  The shared gemm_epilogue loads per-sample scale, per-sample bias and the
    // residual in turn, each behind a block-uniform branch: three dependent memory round trips per tile row, which the
    // other kernels hide behind their many resident blocks and this one cannot (256^2, C 64 -> 64, B = 8: 41 us without an
    // epilogue, 105 with the shared one, 58 with this one; the register-staged kernel takes 65).  Here everything a tile's
    // epilogue reads is requested before the tile's last chunk of MFMAs, and the activation is dispatched once per tile.
    // Supported (checked on the host): bias, batch_scale, batch_bias, activation, bf16 residual, plain bf16 / fp32 rows.
    const int nq0 = n0 + fq * 16;                                  // this lane's 16 channels: nq0 + j * 4 .. + 3
    f32x4 e_bias[TN], e_bs[TN], e_bb[TN];
    uint2 e_res[TM][TN];
    f32x4 e_res32[RES32 ? TM : 1][TN];
    long long e_row[TM];                                           // output row of (tile row i, pixel fr), or -1
#pragma unroll
    for (int j = 0; j < TN; ++j)
        e_bias[j] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + nq0 + j * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const int bld = g.batch_ld ? g.batch_ld : g.N;
    const bool silu = g.act == UWIR_ACT_SILU;
    const float* e_bsp = g.batch_scale;
    const float* e_bbp = g.batch_bias;
    const void* e_resp = g.res;
    auto epi_prefetch = [&](int t) {
        const Tile tc = coords(t);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (e_bsp) e_bs[j] = *reinterpret_cast<const f32x4*>(e_bsp + (size_t)tc.b * bld + nq0 + j * 4);
            if (e_bbp) e_bb[j] = *reinterpret_cast<const f32x4*>(e_bbp + (size_t)tc.b * bld + nq0 + j * 4);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int oy = tc.y0 + wr0 + i, ox = tc.x0 + fr;
            const bool ok = oy < OH && ox < OW;
            e_row[i] = ok ? ((long long)tc.b * OH + oy) * OW + ox : -1;
            if (e_resp) {
                if constexpr (RES32) {
                    const float* rp = (const float*)e_resp + (size_t)(ok ? e_row[i] : 0) * g.ldr + nq0;
#pragma unroll
                    for (int j = 0; j < TN; ++j) e_res32[i][j] = *reinterpret_cast<const f32x4*>(rp + j * 4);
                } else {
                    const bf16_t* rp = (const bf16_t*)e_resp + (size_t)(ok ? e_row[i] : 0) * g.ldr + nq0;
#pragma unroll
                    for (int j = 0; j < TN; ++j) e_res[i][j] = *reinterpret_cast<const uint2*>(rp + j * 4);
                }
            }
        }
    };
    auto epilogue = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            f32x4 v[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                v[j] = acc[i][j] + e_bias[j];
                if (e_bsp) v[j] *= e_bs[j];
                if (e_bbp) v[j] += e_bb[j];
                if (silu) {                                        // (block-uniform; the host admits NONE and SILU only)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[j][e] = v[j][e] * sigmoid_fast(v[j][e]);
                }
                if (e_resp) {
                    if constexpr (RES32) {
                        v[j] += e_res32[i][j];
                    } else {
                        const unsigned lo = e_res[i][j].x, hi = e_res[i][j].y;
                        v[j][0] += __uint_as_float(lo << 16);
                        v[j][1] += __uint_as_float(lo & 0xffff0000u);
                        v[j][2] += __uint_as_float(hi << 16);
                        v[j][3] += __uint_as_float(hi & 0xffff0000u);
                    }
                }
            }
            if (e_row[i] < 0) continue;
            if (g.out_dtype == UWIR_BF16) {
                bf16_t* op = (bf16_t*)g.out + (size_t)e_row[i] * g.ldo + nq0;
#pragma unroll
                for (int jp = 0; jp < TN; jp += 2) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)v[jp][e]; o[4 + e] = (bf16_t)v[jp + 1][e]; }
                    *reinterpret_cast<bf16x8*>(op + jp * 4) = o;
                }
            } else {
                float* op = (float*)g.out + (size_t)e_row[i] * g.ldo + nq0;
#pragma unroll
                for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(op + j * 4) = v[j];
            }
        }
    };

    const int iters = (int)(((long long)mtiles + 2LL * nmblk - 1) / (2LL * nmblk));   // group 0's tile count (>= group 1's)
    const int nq = iters * NCH;
    unsigned pad1 = issue_a(0), pad2 = 0;                          // masks of the chunks in flight: q + 1 and q + 2
    cd_wait_vm<0>();                                               // weights + chunk 0
    zero_pad(pad1, 0);
    pad1 = nq > 1 ? issue_a(1) : 0u;
    int t_prev = -1;                                               // group 1: tile whose epilogue is still owed
    zero_acc();
    for (int q = 0; q < nq; ++q) {
        const int c = q % NCH, t_cur = tile_of(q / NCH);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                              // chunk q is complete in LDS; buffer (q + 2) % NB is free
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 1 && c == 0 && t_prev >= 0) {                   // the deferred epilogue of the previous tile
            epilogue();
            zero_acc();
            t_prev = -1;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c == NCH - 1 && t_cur >= 0) epi_prefetch(t_cur);       // lands behind this chunk's MFMAs
        const bool two = q + 2 < nq && tile_of((q + 2) / NCH) >= 0;
        pad2 = two ? issue_a(q + 2) : 0u;
        __builtin_amdgcn_sched_barrier(0);
        if (t_cur >= 0) compute_chunk(sa_base + (q % NB) * A_BYTES, smem + c * (9 * BN * 64));
        __builtin_amdgcn_sched_barrier(0);
        wait_next(two);
        zero_pad(pad1, q + 1);
        pad1 = pad2;
        if (c == NCH - 1 && t_cur >= 0) {
            if (grp == 0) {
                epilogue();
                zero_acc();
            } else {
                t_prev = t_cur;
            }
        }
    }
    if (grp == 1 && t_prev >= 0) epilogue();
}

// ------------------------------------------------------------------------------------------
// The UNet's 7x7 stem (cat([xt - cond, cond]) padded to 8 channels -> 64, DenoisingUNet_arch.py:60,127) on fp32 rows of
// 8 channels: the generic im2col loader spends its time on per-tap index arithmetic (100 us, 260 TF/s at B = 8, 256^2).
// Here one filter row is two MFMA k-steps -- 8 taps (the 8th with zero weights) x 8 channels, so a lane's 8-element
// fragment piece is exactly ONE input pixel -- the 56 weight fragments live in LDS in fragment order, and a wave walks 8
// output rows of a 16-pixel column strip with a rolling window of 7 input rows in registers (2 fragment loads per new row).
// ------------------------------------------------------------------------------------------
template <typename TO>   // bf16 rows, or fp32 rows (the precise tail path of the UNet's bf16 mode keeps the stem features fp32)
__global__ void __launch_bounds__(256) stem7_mfma_kernel(const float* __restrict__ x, int lda, int H, int W,
                                                         const bf16_t* __restrict__ wt, TO* __restrict__ out, int ldo) {
    constexpr int NF = 7 * 2 * 4;                                  // fragments: (ky, k-step s, channel tile j)
    __shared__ __attribute__((aligned(16))) bf16_t s_w[NF * 64 * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    for (int i = tid; i < NF * 64; i += 256) {
        const int f = i >> 6, l = i & 63, j = f & 3, s2 = (f >> 2) & 1, ky = f >> 3;
        const int ch = chan_perm(j * 16 + (l & 15)), kx = 4 * s2 + (l >> 4);
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.f;
        if (kx < 7) v = *reinterpret_cast<const bf16x8*>(wt + (size_t)ch * 392 + (ky * 7 + kx) * 8);
        *reinterpret_cast<bf16x8*>(&s_w[(size_t)i * 8]) = v;
    }
    __syncthreads();
    const int b = blockIdx.z, x0 = blockIdx.x * 16, ybase = blockIdx.y * 32 + wave * 8;
    const float* xb = x + (size_t)b * H * W * lda;
    auto load_row = [&](int iy, bf16x8 (&dst)[2]) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int ix = x0 + fr + 4 * s2 + fq - 3;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float* src = xb + ((size_t)(ok ? iy : 0) * W + (ok ? ix : 0)) * lda;
            const float4 lo = load4(src), hi = load4(src + 4);
            const float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16_t)(ok ? f[e] : 0.f);
            dst[s2] = v;
        }
    };
    bf16x8 rows[7][2];
#pragma unroll
    for (int k = 0; k < 6; ++k) load_row(ybase - 3 + k, rows[k]);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int y = ybase + r;
        if (y < H) {                                                // wave-uniform
            load_row(y + 3, rows[(r + 6) % 7]);
            f32x4 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(&s_w[(size_t)(((ky * 2 + s2) * 4 + j) * 64 + lane) * 8]);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, rows[(r + ky) % 7][s2], acc[j], 0, 0, 0);
                    }
            if (x0 + fr < W) {
                TO* op = out + ((size_t)(b * H + y) * W + x0 + fr) * ldo + fq * 16;
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
}

static bool stem7_eligible(const uwir_gemm_args& g) {
    return g.compute == UWIR_BF16 && g.KH == 7 && g.KW == 7 && g.stride == 1 && g.pad == 3 && g.C == 8 && g.N == 64 &&
           g.a_dtype == UWIR_F32 && g.lda % 4 == 0 && g.ups == 0 && g.OH == g.H && g.OW == g.W && !g.a_scale && !g.bias &&
           !g.batch_bias && !g.batch_scale && g.act == UWIR_ACT_NONE && !g.res && !g.col_scale && !g.blend_out &&
           g.store_mode == 0 && g.out && (g.out_dtype == UWIR_BF16 || g.out_dtype == UWIR_F32) && g.ldo % 8 == 0 &&
           ((uintptr_t)g.out & 15) == 0 &&
           ((uintptr_t)g.A & 15) == 0 && g.B <= 65535 && ceil_div(g.H, 32) <= 65535;
}

static int g_conv_dma = 1;    // tools/: uwir_conv_variant bits 3-5: 0 default, 1 DMA conv off, 2.. forced tile / ring shapes
static bool conv_dma_eligible(const uwir_gemm_args& g, int K) {
    // 32-bit byte offsets from the tensor bases; weights in whole 64-channel blocks; no prologue affine.  C = 64 (two
    // chunks: the pipeline never fills) stays on the register-staged kernel and its four blocks per CU: 256^2 C64->64
    // 62.6 us there, 70.8 - 77 us here.
    return g.a_dtype == UWIR_BF16 && !g.a_scale && g.N % 64 == 0 && g.C >= 64 && ((uintptr_t)g.A & 15) == 0 &&
           (unsigned long long)g.B * g.H * g.W * g.lda * 2ull < 0xffffffffull && (unsigned long long)g.N * K * 2ull < 0xffffffffull;
}

static bool halo_eligible(const uwir_gemm_args& g) {
    return g.compute == UWIR_BF16 && g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad == 1 && g.C % HC_BK == 0 &&
           g.OH == (g.H << g.ups) && g.OW == (g.W << g.ups) && g.lda % 8 == 0;
}

static int g_gemm_wide = 1;   // tools/: uwir_conv_variant(2) keeps the large-K GEMMs on the 128 x 128 kernel
static int g_gemm_288 = 1;    // tools/: uwir_conv_variant(64) keeps them off the 256 x 288 kernel (256 x 192 where N % 192 == 0)
static int g_halo_th16 = 0;   // tools/: uwir_conv_variant(1) takes 16-row tiles at N <= 64 (A/B timing: 85 vs 88 us at 256^2
                              // C 64 -> 64, equal elsewhere -- three resident 8-row blocks hide more than the bigger tile saves)
void uwir_fft_set_radix4(int on);   // fft.hip
extern "C" void uwir_conv_variant(int v) {
    g_halo_th16 = v & 1;
    g_conv_dma = ((v >> 3) & 7) == 0 ? 1 : ((v >> 3) & 7) == 1 ? 0 : ((v >> 3) & 7);
    g_gemm_wide = (v & 2) ? 0 : 1;
    g_gemm_288 = (v & 64) ? 0 : 1;
    uwir_fft_set_radix4((v & 4) ? 0 : 1);
}

template <typename TA>
static int launch_halo(const GemmP& p0, hipStream_t stream) {
    GemmP p = p0;
    // optional 16-row tiles (256 pixels per block: TM doubles, i.e. half the weight re-staging and barriers per pixel and
    // 0.5 instead of 0.75 LDS fragment reads per MFMA at N = 64): measured no faster, off by default
    const long long mt16 = (long long)p.a.B * ceil_div(p.a.OH, 16) * ceil_div(p.a.OW, HC_TW);
    const bool th16 = g_halo_th16 && p.a.N <= 64 && mt16 >= 1024;
    const int th = th16 ? 16 : 8;
    const long long mtiles = (long long)p.a.B * ceil_div(p.a.OH, th) * ceil_div(p.a.OW, HC_TW);
    int bn = p.a.N <= 64 ? 64 : 128;
    if (bn == 128 && mtiles * ceil_div(p.a.N, 128) < 512) bn = 64;
    p.tiles_n = ceil_div(p.a.N, bn);
    const long long nb = mtiles * p.tiles_n;
    if (nb > 0x7fffffffLL) return uwir_set_error(UWIR_EINVAL, "conv_gemm: grid too large");
    p.nblocks = (int)nb;
    if (bn == 64 && th16) hipLaunchKernelGGL((conv3x3_halo_kernel<TA, 64, 16>), dim3(p.nblocks), dim3(256), 0, stream, p);
    else if (bn == 64) hipLaunchKernelGGL((conv3x3_halo_kernel<TA, 64, 8>), dim3(p.nblocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv3x3_halo_kernel<TA, 128, 8>), dim3(p.nblocks), dim3(256), 0, stream, p);
    UWIR_LAUNCH_CHECK("conv_gemm(halo)");
    return UWIR_OK;
}

static int launch_conv_dma(const GemmP& p0, hipStream_t stream) {
    GemmP p = p0;
    const uwir_gemm_args& ga = p.a;
    const bool resw_epi = (ga.act == UWIR_ACT_NONE || ga.act == UWIR_ACT_SILU) && !ga.col_scale && !ga.blend_out && ga.store_mode == 0 && !ga.res_scale && ga.out &&
                          (!ga.res || ((ga.res_dtype == UWIR_BF16 || ga.res_dtype == UWIR_F32) && ga.ldr % 4 == 0)) &&
                          ((uintptr_t)ga.out & 15) == 0 && ga.ldo % 8 == 0 &&
                          (((uintptr_t)ga.bias | (uintptr_t)ga.batch_scale | (uintptr_t)ga.batch_bias) & 15) == 0;
    if (p.a.C == 64 && g_conv_dma != 7 && resw_epi) {   // short K: persistent blocks with the weight slice resident in LDS
        p.tiles_n = p.a.N / 64;
        const long long mtiles = (long long)p.a.B * ceil_div(p.a.OH, 8) * ceil_div(p.a.OW, HC_TW);
        long long nmblk = 256 / p.tiles_n;                     // one block per CU
        if (nmblk < 1) nmblk = 1;
        if (nmblk > (mtiles + 1) / 2) nmblk = (mtiles + 1) / 2;
        p.nblocks = (int)(nmblk * p.tiles_n);
        if (ga.res && ga.res_dtype == UWIR_F32)
            hipLaunchKernelGGL((conv3x3_resw_kernel<2, true>), dim3(p.nblocks), dim3(512), 0, stream, p);
        else
            hipLaunchKernelGGL((conv3x3_resw_kernel<2, false>), dim3(p.nblocks), dim3(512), 0, stream, p);
        UWIR_LAUNCH_CHECK("conv_gemm(resident weights)");
        return UWIR_OK;
    }
    // Tile / ring shape (tools/bench_conv.py, B = 8): 16-row tiles (half the weight traffic and LDS fragment reads per
    // MFMA, ring of 2) win wherever they still give two blocks per CU -- 64^2 C384->256 66.7 (register-staged) / 60.2
    // (8 rows, ring 3) / 54.5 us; 128^2 C192->128 75.4 / 67.8 / 63.3; nearest-2x 32^2 -> 64^2 C512->256 77 / 70 / 60.6 --
    // below that 8-row tiles with a ring of 3: 32^2 C512->512 50.3 / 41.7 / 44.9, 32^2 C256->256 22.4 / 17.4 / 27.
    int shape = g_conv_dma == 7 ? 1 : g_conv_dma; // forced by tools/: 2 = 8 rows, ring 4; 3 = 8 rows, ring 3; 4 = 16 rows, ring 2;
                                                  // 5 = 16 rows, ring 3 on unpadded (18-row pitch) halo tiles;
                                                  // 7 = automatic, but C = 64 on the ring kernel instead of the resident-weight one
    if (shape < 2) {
        const long long nb16 = (long long)p.a.B * ceil_div(p.a.OH, 16) * ceil_div(p.a.OW, HC_TW) * (p.a.N / 64);
        shape = nb16 >= 512 ? 5 : 3;   // 16-row tiles: ring of 3 on 18-row-pitch halo tiles (1-4 % over the ring of 2 on pitch 20)
    }
    const int th = shape >= 4 ? 16 : 8;
    p.tiles_n = p.a.N / 64;
    const long long nb = (long long)p.a.B * ceil_div(p.a.OH, th) * ceil_div(p.a.OW, HC_TW) * p.tiles_n;
    if (nb > 0x7fffffffLL) return uwir_set_error(UWIR_EINVAL, "conv_gemm: grid too large");
    p.nblocks = (int)nb;
    if (shape == 2) hipLaunchKernelGGL((conv3x3_dma_kernel<8, 4, 20>), dim3(p.nblocks), dim3(256), 0, stream, p);
    else if (shape == 3) hipLaunchKernelGGL((conv3x3_dma_kernel<8, 3, 20>), dim3(p.nblocks), dim3(256), 0, stream, p);
    else if (shape == 5) hipLaunchKernelGGL((conv3x3_dma_kernel<16, 3, 18>), dim3(p.nblocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv3x3_dma_kernel<16, 2, 20>), dim3(p.nblocks), dim3(256), 0, stream, p);
    UWIR_LAUNCH_CHECK("conv_gemm(dma)");
    return UWIR_OK;
}

// split-K factor of a plain bf16 GEMM with 128 x 128 tiles (1 = none) and the workspace it needs
static int splitk_want(int M, int N, int K) {
    const long long tiles = (long long)ceil_div(M, 128) * ceil_div(N, 128);
    if (tiles >= 128) return 1;
    int s = (int)(256 / tiles);
    s = s < K / 64 / 3 ? s : K / 64 / 3;  // at least 3 K tiles per slice
    return s > SPLITK_MAX ? SPLITK_MAX : (s < 1 ? 1 : s);
}
static size_t splitk_bytes(int M, int N, int s) {
    return (size_t)s * ceil_div(M, 128) * ceil_div(N, 128) * (128 * 128 * sizeof(float));
}
static int splitk_factor(int M, int N, int K, const void* ws, size_t ws_bytes) {
    int s = splitk_want(M, N, K);
    if (s <= 1 || !ws || ((uintptr_t)ws & 15) != 0) return 1;
    while (s > 1 && splitk_bytes(M, N, s) > ws_bytes) --s;
    return s;
}

int uwir_gemm_wide_try(const uwir_gemm_args& g, int M, int K, hipStream_t stream);      // gemm_wide.hip
int uwir_gemm_wide288_try(const uwir_gemm_args& g, int M, int K, hipStream_t stream);   // gemm_wide288.hip

template <typename TC, typename TA>
static int launch_gemm(const GemmP& p0, hipStream_t stream) {
    GemmP p = p0;
    if (p.a.w_batch_stride) {
        if (!glds_eligible(p.a, p.K) || (p.a.OH * p.a.OW) % 128 != 0 || p.a.w_batch_stride < (long long)p.a.N * p.K ||
            p.a.w_batch_stride % 8 != 0)
            return uwir_set_error(UWIR_EUNSUPPORTED, "conv_gemm: w_batch_stride needs a plain bf16 1x1 GEMM (bf16 rows, K %% 64 == 0, "
                                                     "N > 64), OH * OW %% 128 == 0 and a stride >= N * K, a multiple of 8");
    }
    if (g_gemm_wide && glds_eligible(p.a, p.K)) {
        int took = g_gemm_288 ? uwir_gemm_wide288_try(p.a, p.M, p.K, stream) : 0;   // (takes per-image weight panels too)
        if (took == 0 && !p.a.w_batch_stride) took = uwir_gemm_wide_try(p.a, p.M, p.K, stream);
        if (took != 0) return took < 0 ? took : UWIR_OK;
    }
    int bn = p.a.N <= 64 ? 64 : 128;
    // few-block launches (deep, narrow levels: M = 8192 rows): halve the N tile to double the blocks on the 256 CUs
    if (bn == 128 && (long long)ceil_div(p.M, 128) * ceil_div(p.a.N, 128) < 512 && !glds_eligible(p.a, p.K)) bn = 64;
    p.tiles_n = ceil_div(p.a.N, bn);
    long long nb = (long long)ceil_div(p.M, 128) * p.tiles_n;
    if (nb > 0x7fffffffLL) return uwir_set_error(UWIR_EINVAL, "conv_gemm: grid too large");
    p.nblocks = (int)nb;
    // (one block per CU and a long K: M8192 N512 K2048 42 -> 33 us double-buffered; shorter K gains nothing)
    if (glds_eligible(p.a, p.K) && (p.nblocks < 256 || (p.nblocks == 256 && p.K >= 1536))) {
        // fewer tiles than CUs: per-tile latency is what matters.  Double-buffered staging (57 -> 44 us at M800 N768
        // K3072) and, when the caller lent a workspace, split-K (see splitk_dump)
        p.splitk = splitk_factor(p.M, p.a.N, p.K, p.a.workspace, p.a.workspace_bytes);
        hipLaunchKernelGGL((gemm_bf16_glds_kernel<2, EPI_FULL>), dim3(p.nblocks * p.splitk), dim3(256), 0, stream, p);
        if (p.splitk > 1) hipLaunchKernelGGL(gemm_splitk_epilogue_kernel, dim3(p.nblocks * 4), dim3(256), 0, stream, p);
        UWIR_LAUNCH_CHECK("conv_gemm(glds, few tiles)");
        return UWIR_OK;
    }
    if (glds_eligible(p.a, p.K)) {
        // measured on MI355X (tools/bench_gemm.py): the single-buffered form wins at every K of this path, e.g.
        // M262144 N384: K192 0.155 -> 0.124 ms, K768 0.303 -> 0.256; M262144 N576 K1728 0.796 -> 0.689;
        // M16384 N2304 K6912 0.641 -> 0.609.  Four resident blocks per CU hide more than intra-block overlap.
        switch (epi_class(p.a)) {
            case EPI_LEAN: hipLaunchKernelGGL((gemm_bf16_glds_kernel<1, EPI_LEAN>), dim3(p.nblocks), dim3(256), 0, stream, p); break;
            case EPI_RES: hipLaunchKernelGGL((gemm_bf16_glds_kernel<1, EPI_RES>), dim3(p.nblocks), dim3(256), 0, stream, p); break;
            default: hipLaunchKernelGGL((gemm_bf16_glds_kernel<1, EPI_FULL>), dim3(p.nblocks), dim3(256), 0, stream, p);
        }
        UWIR_LAUNCH_CHECK("conv_gemm(glds)");
        return UWIR_OK;
    }
    // one LDS buffer + register prefetch (NB = 1): half the LDS, more resident blocks; measured 13.0 -> 11.5 ms on the
    // IR-SDE UNet step (B=8, 256^2, bf16) against the double-buffered form, neutral on the student
// (no EPI_RES variants here: the register-staged loop plus residual start values reach 257 VGPRs at BN = 128)
#define UWIR_LAUNCH_GENERIC(BN_)                                                                                         \
    if (epi_class(p.a) == EPI_LEAN)                                                                                      \
        hipLaunchKernelGGL((conv_gemm_kernel<TC, TA, BN_, 1, EPI_LEAN>), dim3(p.nblocks), dim3(256), 0, stream, p);      \
    else                                                                                                                 \
        hipLaunchKernelGGL((conv_gemm_kernel<TC, TA, BN_, 1, EPI_FULL>), dim3(p.nblocks), dim3(256), 0, stream, p);
    if (bn == 64) { UWIR_LAUNCH_GENERIC(64) } else { UWIR_LAUNCH_GENERIC(128) }
#undef UWIR_LAUNCH_GENERIC
    UWIR_LAUNCH_CHECK("conv_gemm");
    return UWIR_OK;
}

extern "C" size_t uwir_conv_gemm_workspace_bytes(const uwir_gemm_args* args) {
    if (!args) return 0;
    const uwir_gemm_args& g = *args;
    if (g.B <= 0 || g.OH <= 0 || g.OW <= 0 || g.C <= 0 || g.N <= 0) return 0;
    const long long M = (long long)g.B * g.OH * g.OW;
    const int K = g.KH * g.KW * g.C;
    if (M >= 0x7fffffffLL || !glds_eligible(g, K)) return 0;
    const int s = splitk_want((int)M, g.N, K);
    return s > 1 ? splitk_bytes((int)M, g.N, s) : 0;
}

extern "C" int uwir_conv_gemm(const uwir_gemm_args* args, uwir_stream_t stream) {
    UWIR_CHECK(args, "conv_gemm: null args");
    const uwir_gemm_args& g = *args;
    UWIR_CHECK(g.A && g.Wt && (g.out || g.blend_out), "conv_gemm: null pointer");
    UWIR_CHECK(g.B > 0 && g.H > 0 && g.W > 0 && g.C > 0 && g.N > 0 && g.KH > 0 && g.KW > 0 && g.stride > 0 &&
                   g.pad >= 0 && g.OH > 0 && g.OW > 0,
               "conv_gemm: bad geometry");
    UWIR_CHECK(g.ups == 0 || g.ups == 1, "conv_gemm: ups must be 0 or 1");
    UWIR_CHECK(g.batch_ld == 0 || (g.batch_ld >= g.N && g.batch_ld % 4 == 0), "conv_gemm: bad batch_ld");
    UWIR_CHECK((((uintptr_t)g.batch_bias | (uintptr_t)g.batch_scale) & 15) == 0, "conv_gemm: batch_bias/batch_scale must be 16-byte aligned");
    const int Hv = g.H << g.ups, Wv = g.W << g.ups;
    UWIR_CHECK(Hv + 2 * g.pad >= g.KH && Wv + 2 * g.pad >= g.KW &&
                   g.OH <= (Hv + 2 * g.pad - g.KH) / g.stride + 1 && g.OW <= (Wv + 2 * g.pad - g.KW) / g.stride + 1,
               "conv_gemm: output size exceeds the convolution's extent");
    UWIR_CHECK(g.compute == UWIR_F32 || g.compute == UWIR_BF16, "conv_gemm: bad compute dtype");
    UWIR_CHECK(g.w_dtype == g.compute, "conv_gemm: weight dtype must equal compute dtype");
    const int epc = g.compute == UWIR_BF16 ? 8 : 4;
    UWIR_CHECK(g.C % epc == 0 && g.lda % epc == 0 && g.lda >= g.C, "conv_gemm: C and lda must be multiples of %d", epc);
    UWIR_CHECK(((uintptr_t)g.A & 15) == 0 && ((uintptr_t)g.Wt & 15) == 0, "conv_gemm: A / Wt must be 16-byte aligned");
    UWIR_CHECK(g.N % 4 == 0, "conv_gemm: N must be a multiple of 4");
    UWIR_CHECK((g.a_scale == nullptr) == (g.a_shift == nullptr), "conv_gemm: a_scale/a_shift must come together");
    UWIR_CHECK((g.res_scale == nullptr) == (g.res_shift == nullptr), "conv_gemm: res_scale/res_shift must come together");
    UWIR_CHECK(!g.res_scale || g.res, "conv_gemm: res_scale without res");
    UWIR_CHECK(!g.res || (g.ldr >= g.N && g.ldr % 4 == 0), "conv_gemm: bad ldr");
    UWIR_CHECK(!g.res || g.res_dtype == UWIR_F32 || g.res_dtype == UWIR_BF16, "conv_gemm: bad res dtype");
    UWIR_CHECK(!g.res || ((uintptr_t)g.res & (g.res_dtype == UWIR_BF16 ? 7 : 15)) == 0, "conv_gemm: res must be 16-byte (fp32) / 8-byte (bf16) aligned");
    UWIR_CHECK(!g.blend_out || (g.blend_t && g.blend_f && g.ld_blend_in % 4 == 0 && g.ld_blend_out % 4 == 0),
               "conv_gemm: bad blend arguments");
    UWIR_CHECK(g.store_mode >= 0 && g.store_mode <= 2, "conv_gemm: bad store_mode");
    if (g.out) {
        UWIR_CHECK(g.out_dtype == UWIR_F32 || g.out_dtype == UWIR_BF16, "conv_gemm: bad out dtype");
        UWIR_CHECK(g.ldo % 4 == 0, "conv_gemm: ldo must be a multiple of 4");
        if (g.store_mode == 1) {
            UWIR_CHECK(g.N % 16 == 0 && g.ldo >= g.N / 4 && !g.res && !g.blend_out,
                       "conv_gemm: depth-to-space needs N %% 16 == 0 and no residual/blend");
        } else if (g.store_mode == 2) {
            UWIR_CHECK(g.N % 16 == 0 && g.ldo >= g.N / 2 && g.ldo % 8 == 0 && ((uintptr_t)g.out & 15) == 0 && !g.res &&
                           !g.blend_out && g.act == UWIR_ACT_NONE && !g.col_scale,
                       "conv_gemm: GEGLU store needs N %% 16 == 0, 16-byte aligned rows of N / 2 and a plain bias epilogue");
        } else {
            UWIR_CHECK(g.ldo >= g.N, "conv_gemm: ldo < N");
        }
    }
    GemmP p;
    p.a = g;
    long long M = (long long)g.B * g.OH * g.OW;
    UWIR_CHECK(M < 0x7fffffffLL, "conv_gemm: M too large");
    p.M = (int)M;
    p.K = g.KH * g.KW * g.C;
    p.tiles_n = 0;
    p.nblocks = 0;
    p.splitk = 1;
    hipStream_t s = (hipStream_t)stream;
    if (stem7_eligible(g)) {
        if (g.out_dtype == UWIR_BF16)
            hipLaunchKernelGGL(stem7_mfma_kernel<bf16_t>, dim3(ceil_div(g.W, 16), ceil_div(g.H, 32), g.B), dim3(256), 0, s,
                               (const float*)g.A, g.lda, g.H, g.W, (const bf16_t*)g.Wt, (bf16_t*)g.out, g.ldo);
        else
            hipLaunchKernelGGL(stem7_mfma_kernel<float>, dim3(ceil_div(g.W, 16), ceil_div(g.H, 32), g.B), dim3(256), 0, s,
                               (const float*)g.A, g.lda, g.H, g.W, (const bf16_t*)g.Wt, (float*)g.out, g.ldo);
        UWIR_LAUNCH_CHECK("conv_gemm(stem 7x7)");
        return UWIR_OK;
    }
    if (halo_eligible(g)) {
        if (g_conv_dma && conv_dma_eligible(g, p.K)) return launch_conv_dma(p, s);
        if (g.a_dtype == UWIR_F32) return launch_halo<float>(p, s);
        if (g.a_dtype == UWIR_BF16) return launch_halo<bf16_t>(p, s);
    }
    if (g.compute == UWIR_BF16) {
        if (g.a_dtype == UWIR_F32) return launch_gemm<bf16_t, float>(p, s);
        if (g.a_dtype == UWIR_BF16) return launch_gemm<bf16_t, bf16_t>(p, s);
    } else {
        if (g.a_dtype == UWIR_F32) return launch_gemm<float, float>(p, s);
    }
    return uwir_set_error(UWIR_EUNSUPPORTED, "conv_gemm: unsupported (compute, a_dtype) combination");
}

// Fused LinearAttention block of the IR-SDE score network (bf16 MFMA mode):
//     out = LN_gout( W_out . LA( W_qkv . LN_gpre(x) ) + b_out ) + x
// = Residual(PreNorm(dim, LinearAttention(dim))), U/models/modules/module_util.py:27-33,77-97,157-185 as wired by
// DenoisingUNet_arch.py:81-88.  The unfused launch sequence (ln_rows -> qkv GEMM -> kmax / ctx / fin / apply ->
// to_out GEMM -> ln_rows + residual) moves ~2.1 GB through HBM at level 0 (B = 8, 256^2); every step after the context
// reduction is ROW-LOCAL, so two passes over x suffice (x read twice, out written once: 0.4 GB):
//   pass 1 (la_kv):    x tile -> channel LayerNorm -> K_h, V_h = xn . W^T on MFMA -> E = exp(K - running max_d) ->
//                      ctx^T[e][d] += V^T E, S[d] += sum E, per (chunk of rows, b, head); online max over the rows
//                      with the accumulator rescaled like flash attention (any shift gives the same softmax_n(k)).
//   fin:               chunks combined (max-aligned) -> ctx^T * D^-0.5 / n, bf16
//   pass 2 (la_apply): x rows -> LayerNorm -> Q^T_h -> softmax over d -> Y^T_h = ctx^T P -> Z^T = W_out Y^T + b ->
//                      LayerNorm over the output channels -> + x -> out.
// No LDS transposition anywhere: every product is oriented so that the NEXT product sums over the ROW index of the
// previous 32x32 accumulator tile (column on the lane, rows in the 16 registers), which can then be fed straight back
// as an MFMA operand after a bf16 pack (k order inside a step: row 16s + 8(j>>2) + 4h + (j&3) for element j of lane
// half h -- the other operand is gathered in the same order).
#include "common.h"

#define LF_D 32          // dim_head (fixed by the reference, DenoisingUNet_arch.py:31-32)
#define LF_HEADS 4       // heads (ditto); one wave per head in pass 1
#define LF_TR 64         // rows per LDS tile in pass 1
#define LOG2E_F 1.4426950408889634f

// v_permlane32_swap hands every lane its own value and that of lane ^ 32 (in an order that depends on the half): the
// symmetric combinations below need no case split.
__device__ __forceinline__ float half_max(float v) {
    const unsigned u = __float_as_uint(v);
    auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}
__device__ __forceinline__ float half_sum(float v) {
    const unsigned u = __float_as_uint(v);
    auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& x, int s) {   // registers 8s .. 8s+7 -> bf16 fragment of k-step s
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)x[8 * s + j];
    return o;
}

// ------------------------------------------------------------------------------------------
// pass 1
// ------------------------------------------------------------------------------------------
template <int DIM, typename TX>
__global__ void __launch_bounds__(256) la_kv_fused_kernel(const TX* __restrict__ x, int ldx, int n, int rows_per_chunk,
                                                          int nchunks, const float* __restrict__ g_pre, float eps,
                                                          const bf16_t* __restrict__ w_qkv, float* __restrict__ part) {
    constexpr int D = LF_D, KS = DIM / 16, PITCH = DIM + 8;          // bf16 elements per LDS row (16-byte pad)
    constexpr int LPR = DIM / 4;                                      // lanes per row when loading float4 pieces
    constexpr int RPI = 256 / LPR;                                    // rows per block-wide load instruction
    constexpr int NLD = LF_TR / RPI;                                  // float4 loads per thread and tile
    __shared__ __attribute__((aligned(16))) bf16_t s_xn[2][LF_TR * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane & 31, lh = lane >> 5;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int r_begin = chunk * rows_per_chunk, r_end = min(n, r_begin + rows_per_chunk);
    const int head = wave;
    const TX* xb = x + (size_t)b * n * ldx;

    // weights of this head as B operands: B[k = c][n = d] = Wk[head*32 + d][c]  (rows 128.. of w_qkv are k, 256.. v)
    bf16x8 wk[KS], wv[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        wk[ks] = *reinterpret_cast<const bf16x8*>(w_qkv + (size_t)(LF_HEADS * D + head * D + lq) * DIM + 16 * ks + 8 * lh);
        wv[ks] = *reinterpret_cast<const bf16x8*>(w_qkv + (size_t)(2 * LF_HEADS * D + head * D + lq) * DIM + 16 * ks + 8 * lh);
    }
    const int lrow = tid / LPR, lcol = (tid % LPR) * 4;              // this thread's (row in instruction, first channel)
    const float4 gp = load4(g_pre + lcol);

    f32x16 acc;                                                       // ctx^T[e][d]: column (lane) = d, rows = e
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float run_max = -INFINITY, run_sum = 0.f;                         // per lane: column d = lq (both halves agree on max)

    float4 raw[NLD];
    auto load_tile = [&](int r0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int r = r0 + i * RPI + lrow;
            raw[i] = r < r_end ? load4(xb + (size_t)r * ldx + lcol) : make_float4(0, 0, 0, 0);
        }
    };
    auto norm_store = [&](int buf) {                                  // channel LayerNorm (g only, biased var) -> bf16 tile
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const float4 v = raw[i];
            float s = v.x + v.y + v.z + v.w;
#pragma unroll
            for (int off = LPR / 2; off > 0; off >>= 1) s += __shfl_xor(s, off);
            const float mean = s * (1.f / DIM);
            const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
            float q = dx * dx + dy * dy + dz * dz + dw * dw;
#pragma unroll
            for (int off = LPR / 2; off > 0; off >>= 1) q += __shfl_xor(q, off);
            const float rstd = rsqrtf(q * (1.f / DIM) + eps);
            store4(&s_xn[buf][(i * RPI + lrow) * PITCH + lcol],
                   make_float4(dx * rstd * gp.x, dy * rstd * gp.y, dz * rstd * gp.z, dw * rstd * gp.w));
        }
    };

    load_tile(r_begin);
    int buf = 0;
    for (int r0 = r_begin; r0 < r_end; r0 += LF_TR, buf ^= 1) {
        norm_store(buf);
        __syncthreads();
        if (r0 + LF_TR < r_end) load_tile(r0 + LF_TR);
#pragma unroll
        for (int rg = 0; rg < LF_TR / 32; ++rg) {
            const int rbase = r0 + rg * 32;
            if (rbase >= r_end) break;                                // wave-uniform
            f32x16 xk, xv;
#pragma unroll
            for (int r = 0; r < 16; ++r) { xk[r] = 0.f; xv[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {                         // A[row][c] = xn tile, rows on lanes
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&s_xn[buf][(rg * 32 + lq) * PITCH + 16 * ks + 8 * lh]);
                xk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wk[ks], xk, 0, 0, 0);
                xv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wv[ks], xv, 0, 0, 0);
            }
            // xk[reg] = K[row = rbase + (reg&3) + 8(reg>>2) + 4 lh][d = lq] ; rows past the end contribute exp = 0
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
                xk[r] = row < r_end ? xk[r] * LOG2E_F : -INFINITY;
                m = fmaxf(m, xk[r]);
            }
            m = half_max(m);
            const float new_max = fmaxf(run_max, m);                  // finite: the group has at least one valid row
            const float f = __builtin_amdgcn_exp2f(run_max - new_max);   // first group: exp2(-inf) = 0
            run_max = new_max;
            float ssum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                xk[r] = __builtin_amdgcn_exp2f(xk[r] - new_max);
                ssum += xk[r];
            }
            run_sum = run_sum * f + ssum;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] *= f;
#pragma unroll
            for (int s = 0; s < 2; ++s)                                // ctx^T += V^T (A: m = e) x E (B: n = d), k = row
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(xv, s), pack8(xk, s), acc, 0, 0, 0);
        }
    }
    run_sum = half_sum(run_sum);
    // partial record of (b, head, chunk): [max(32) | sum(32) | ctx^T 32 x 32 (e-major)]
    float* o = part + (((size_t)b * LF_HEADS + head) * nchunks + chunk) * (2 * D + D * D);
    if (lh == 0) {
        o[lq] = run_max;
        o[D + lq] = run_sum;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) o[2 * D + ((r & 3) + 8 * (r >> 2) + 4 * lh) * D + lq] = acc[r];
}

// chunks -> ctx^T[b][head][e][d] * (D^-0.5 / n), bf16.  One block per (b, head, group of 4 e); thread = (d, chunk lane cl):
// 32 chunk lanes, so a thread walks nchunks / 32 records (one block per (b, head) with 4 chunk lanes was a chain of
// 2 x 32 dependent L2 round trips: 16 us per call, six calls per UNet step); the lanes' partial sums are combined through
// LDS in a fixed order (deterministic).
#define LF_FIN_CL 32
__global__ void __launch_bounds__(1024) la_fin_fused_kernel(const float* __restrict__ part, int nchunks, float inv_n_scale,
                                                            bf16_t* __restrict__ ctxT) {
    constexpr int D = LF_D, REC = 2 * D + D * D, CL = LF_FIN_CL;
    __shared__ float s_m[CL][D];
    __shared__ float s_s[CL][D];
    __shared__ float s_a[CL][4][D];
    const int bh = blockIdx.x >> 3, eg = blockIdx.x & 7, d = threadIdx.x & 31, cl = threadIdx.x >> 5;
    const float* p0 = part + (size_t)bh * nchunks * REC;
    float m = -INFINITY;
    for (int c = cl; c < nchunks; c += CL) m = fmaxf(m, p0[(size_t)c * REC + d]);
    s_m[cl][d] = m;
    __syncthreads();
    m = s_m[0][d];
#pragma unroll
    for (int k = 1; k < CL; ++k) m = fmaxf(m, s_m[k][d]);
    float a[4] = {0.f, 0.f, 0.f, 0.f}, sum = 0.f;
    for (int c = cl; c < nchunks; c += CL) {
        const float* p = p0 + (size_t)c * REC;
        const float w = __builtin_amdgcn_exp2f(p[d] - m);
        sum = fmaf(w, p[D + d], sum);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = fmaf(w, p[2 * D + (eg * 4 + i) * D + d], a[i]);
    }
    s_s[cl][d] = sum;
#pragma unroll
    for (int i = 0; i < 4; ++i) s_a[cl][i][d] = a[i];
    __syncthreads();
    if (cl < 4) {                                                  // thread (d, i = cl): fixed-order sums over the chunk lanes
        float st = 0.f, at = 0.f;
#pragma unroll
        for (int k = 0; k < CL; ++k) { st += s_s[k][d]; at += s_a[k][cl][d]; }
        ctxT[(size_t)bh * D * D + (eg * 4 + cl) * D + d] = (bf16_t)(at / st * inv_n_scale);
    }
}

// ------------------------------------------------------------------------------------------
// pass 2: one wave = rgw groups of 32 rows, all heads.  Weights, ctx^T and the gain / bias vectors sit in LDS for the
// block (padded pitches: conflict-free 16-byte fragment reads); the operands that multiply an accumulator-derived
// fragment (ctx^T, W_out) are stored with their k index already in accumulator order, so every fragment is ONE
// ds_read_b128: LDS column st*16 + h*8 + j of a 32-wide group holds source column 16 st + 8 (j>>2) + 4 h + (j&3).
// ------------------------------------------------------------------------------------------

template <int DIM, typename TX>
__global__ void __launch_bounds__(256) la_apply_fused_kernel(const TX* __restrict__ x, int ldx, int n,
                                                             const float* __restrict__ g_pre, float eps,
                                                             const bf16_t* __restrict__ w_qkv,
                                                             const bf16_t* __restrict__ ctxT,
                                                             const bf16_t* __restrict__ w_out,
                                                             const float* __restrict__ b_out,
                                                             const float* __restrict__ g_out, TX* __restrict__ out,
                                                             int ldo, int rgw) {
    constexpr int D = LF_D, KS = DIM / 16, HD = LF_HEADS * D, OT = DIM / 32;
    constexpr int PQ = DIM + 8, PO = HD + 8, PC = D + 8;             // LDS pitches in bf16 elements
    constexpr int SM_BYTES = (HD * PQ + DIM * PO + LF_HEADS * D * PC) * 2 + 3 * DIM * 4;
    __shared__ __attribute__((aligned(16))) char smem[SM_BYTES];
    bf16_t* s_wq = reinterpret_cast<bf16_t*>(smem);                   // [HD][PQ]      Wq rows (d), natural k = c
    bf16_t* s_wo = s_wq + HD * PQ;                                    // [DIM][PO]     W_out rows (o), k in accumulator order
    bf16_t* s_ct = s_wo + DIM * PO;                                   // [4][D][PC]    ctx^T rows (e), k = d in accumulator order
    float* s_vec = reinterpret_cast<float*>(s_ct + LF_HEADS * D * PC);   // [g_pre | b_out | g_out]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y;

    for (int i = tid; i < HD * (DIM / 8); i += 256) {                 // 16-byte pieces
        const int r = i / (DIM / 8), c8 = (i % (DIM / 8)) * 8;
        *reinterpret_cast<bf16x8*>(&s_wq[r * PQ + c8]) = *reinterpret_cast<const bf16x8*>(w_qkv + (size_t)r * DIM + c8);
    }
    // accumulator-order images: LDS piece (group of 32 k, st, h) = source columns 16 st + 4 h + {0..3} and + 8 + {0..3}
    auto gather_piece = [](const bf16_t* src_group32, int st, int h) {
        const bf16x4 c0 = *reinterpret_cast<const bf16x4*>(src_group32 + 16 * st + 4 * h);
        const bf16x4 c1 = *reinterpret_cast<const bf16x4*>(src_group32 + 16 * st + 4 * h + 8);
        bf16x8 a;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = c0[j]; a[4 + j] = c1[j]; }
        return a;
    };
    for (int i = tid; i < DIM * (HD / 8); i += 256) {
        const int r = i / (HD / 8), pc = i % (HD / 8);                // pc = hd * 4 + st * 2 + h
        *reinterpret_cast<bf16x8*>(&s_wo[r * PO + pc * 8]) =
            gather_piece(w_out + (size_t)r * HD + (pc >> 2) * 32, (pc >> 1) & 1, pc & 1);
    }
    for (int i = tid; i < LF_HEADS * D * (D / 8); i += 256) {
        const int r = i / (D / 8), pc = i % (D / 8);                  // r = head * 32 + e ; pc = st * 2 + h
        *reinterpret_cast<bf16x8*>(&s_ct[r * PC + pc * 8]) =
            gather_piece(ctxT + (size_t)b * LF_HEADS * D * D + (size_t)r * D, pc >> 1, pc & 1);
    }
    for (int i = tid; i < DIM; i += 256) {
        s_vec[i] = g_pre[i];
        s_vec[DIM + i] = b_out[i];
        s_vec[2 * DIM + i] = g_out[i];
    }
    __syncthreads();

#pragma unroll 1
    for (int rg = 0; rg < rgw; ++rg) {
        const int rbase = ((blockIdx.x * 4 + wave) * rgw + rg) * 32;
        if (rbase >= n) break;                                         // wave-uniform; no barrier below
        const int row = min(rbase + lq, n - 1);                        // clamped rows recompute the last row (not stored)
        const bool row_ok = rbase + lq < n;
        const TX* xr = x + ((size_t)b * n + row) * ldx;

        // ---- this lane's half of its row, in B-operand order: channels 16 ks + 8 lh + j ; LayerNorm across the lane pair
        float xv[KS][8];
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float4 lo = load4(xr + 16 * ks + 8 * lh), hi = load4(xr + 16 * ks + 8 * lh + 4);
            xv[ks][0] = lo.x; xv[ks][1] = lo.y; xv[ks][2] = lo.z; xv[ks][3] = lo.w;
            xv[ks][4] = hi.x; xv[ks][5] = hi.y; xv[ks][6] = hi.z; xv[ks][7] = hi.w;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += xv[ks][j];
        }
        const float mean = half_sum(s) * (1.f / DIM);
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) { xv[ks][j] -= mean; q = fmaf(xv[ks][j], xv[ks][j], q); }
        const float rstd = rsqrtf(half_sum(q) * (1.f / DIM) + eps);
        bf16x8 xn[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float4 g0 = load4(s_vec + 16 * ks + 8 * lh), g1 = load4(s_vec + 16 * ks + 8 * lh + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) xn[ks][j] = (bf16_t)(xv[ks][j] * rstd * gg[j]);
        }

        // ---- per head: Q^T (rows d, column = token row) -> softmax over d -> Y^T_h = ctx^T_h P ; packed for the to_out GEMM
        bf16x8 yf[LF_HEADS][2];
#pragma unroll
        for (int hd = 0; hd < LF_HEADS; ++hd) {
            f32x16 qt;
#pragma unroll
            for (int r = 0; r < 16; ++r) qt[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {                          // A[m = d][k = c] = Wq[hd*32 + d][c]
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&s_wq[(hd * D + lq) * PQ + 16 * ks + 8 * lh]);
                qt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xn[ks], qt, 0, 0, 0);
            }
            float m = qt[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, qt[r]);
            m = half_max(m);
            float ssum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { qt[r] = __builtin_amdgcn_exp2f((qt[r] - m) * LOG2E_F); ssum += qt[r]; }
            const float inv = 1.f / half_sum(ssum);
#pragma unroll
            for (int r = 0; r < 16; ++r) qt[r] *= inv;
            f32x16 y;
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] = 0.f;
#pragma unroll
            for (int st = 0; st < 2; ++st) {                           // A[m = e][k = d] = ctx^T[e][d], d in accumulator order
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&s_ct[(hd * D + lq) * PC + 16 * st + 8 * lh]);
                y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pack8(qt, st), y, 0, 0, 0);
            }
            yf[hd][0] = pack8(y, 0);
            yf[hd][1] = pack8(y, 1);
        }

        // ---- Z^T[o][row] = sum_e W_out[o][e] Y[e][row] + b[o] ; k = head*32 + (accumulator order of e)
        f32x16 z[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
#pragma unroll
            for (int r = 0; r < 16; ++r) z[ot][r] = 0.f;
#pragma unroll
            for (int hd = 0; hd < LF_HEADS; ++hd)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(&s_wo[(ot * 32 + lq) * PO + hd * D + 16 * st + 8 * lh]);
                    z[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, yf[hd][st], z[ot], 0, 0, 0);
                }
        }
        // z[ot][r]: output channel o = 32 ot + (r&3) + 8 (r>>2) + 4 lh of token `row`; LayerNorm across the lane pair
        float zs = 0.f;
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 bb = load4(s_vec + DIM + 32 * ot + 8 * g4 + 4 * lh);
                z[ot][4 * g4] += bb.x; z[ot][4 * g4 + 1] += bb.y; z[ot][4 * g4 + 2] += bb.z; z[ot][4 * g4 + 3] += bb.w;
                zs += z[ot][4 * g4] + z[ot][4 * g4 + 1] + z[ot][4 * g4 + 2] + z[ot][4 * g4 + 3];
            }
        const float zmean = half_sum(zs) * (1.f / DIM);
        float zq = 0.f;
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
            for (int r = 0; r < 16; ++r) { z[ot][r] -= zmean; zq = fmaf(z[ot][r], z[ot][r], zq); }
        const float zrstd = rsqrtf(half_sum(zq) * (1.f / DIM) + eps);
        if (row_ok) {
            TX* orow = out + ((size_t)b * n + row) * ldo;
#pragma unroll
            for (int ot = 0; ot < OT; ++ot)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int o = 32 * ot + 8 * g4 + 4 * lh;
                    const float4 gg = load4(s_vec + 2 * DIM + o), res = load4(xr + o);
                    store4(orow + o, make_float4(fmaf(z[ot][4 * g4] * zrstd, gg.x, res.x),
                                                 fmaf(z[ot][4 * g4 + 1] * zrstd, gg.y, res.y),
                                                 fmaf(z[ot][4 * g4 + 2] * zrstd, gg.z, res.z),
                                                 fmaf(z[ot][4 * g4 + 3] * zrstd, gg.w, res.w)));
                }
        }
    }
}

// ------------------------------------------------------------------------------------------
static int la_block_chunks(int B, int n, int* rows_per_chunk) {
    // at most 768 blocks over the batch (3 resident blocks per CU at 138 VGPRs: ONE round of blocks -- 1024 equal blocks
    // on 768 slots take two), chunks a multiple of the 64-row tile, at most 256 chunks
    int want = 768 / B;
    if (want < 1) want = 1;
    int tiles = ceil_div(n, LF_TR);
    int nchunks = want < tiles ? want : tiles;
    if (nchunks > 256) nchunks = 256;
    if (nchunks < 1) nchunks = 1;
    int tiles_per = ceil_div(tiles, nchunks);
    *rows_per_chunk = tiles_per * LF_TR;
    return ceil_div(n, *rows_per_chunk);
}

extern "C" size_t uwir_linattn_block_workspace_bytes(int B, int n) {
    if (B <= 0 || n <= 0) return 0;
    int rpc;
    const size_t nchunks = la_block_chunks(B, n, &rpc);
    return (size_t)B * LF_HEADS * nchunks * (2 * LF_D + LF_D * LF_D) * sizeof(float) +
           (size_t)B * LF_HEADS * LF_D * LF_D * sizeof(bf16_t);
}

extern "C" int uwir_linattn_block(const void* x, int io_dtype, int ldx, int B, int n, int dim, int heads, const float* g_pre,
                                  const void* w_qkv, const void* w_out, const float* b_out, const float* g_out, float eps,
                                  void* out, int ldo, void* workspace, size_t workspace_bytes, uwir_stream_t stream) {
    UWIR_CHECK(x && g_pre && w_qkv && w_out && b_out && g_out && out && workspace, "linattn_block: null pointer");
    UWIR_CHECK(io_dtype == UWIR_F32 || io_dtype == UWIR_BF16, "linattn_block: x / out must be fp32 or bf16");
    UWIR_CHECK(heads == LF_HEADS, "linattn_block: heads must be 4 (the reference's LinearAttention)");
    UWIR_CHECK(dim == 64 || dim == 128 || dim == 256, "linattn_block: dim must be 64, 128 or 256 (got %d)", dim);
    const int ldq = io_dtype == UWIR_BF16 ? 8 : 4;    // rows start on 16-byte boundaries
    UWIR_CHECK(B > 0 && B <= 65535 && n > 0 && ldx >= dim && ldx % ldq == 0 && ldo >= dim && ldo % ldq == 0,
               "linattn_block: bad shape");
    UWIR_CHECK((((uintptr_t)x | (uintptr_t)out | (uintptr_t)w_qkv | (uintptr_t)w_out | (uintptr_t)g_pre | (uintptr_t)g_out |
                 (uintptr_t)b_out | (uintptr_t)workspace) & 15) == 0, "linattn_block: pointers must be 16-byte aligned");
    UWIR_CHECK(workspace_bytes >= uwir_linattn_block_workspace_bytes(B, n), "linattn_block: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    int rpc;
    const int nchunks = la_block_chunks(B, n, &rpc);
    float* part = (float*)workspace;
    bf16_t* ctxT = (bf16_t*)(part + (size_t)B * LF_HEADS * nchunks * (2 * LF_D + LF_D * LF_D));
    const bf16_t* wq = (const bf16_t*)w_qkv;
    const bf16_t* wo = (const bf16_t*)w_out;
    const float inv_n_scale = (float)(1.0 / ((double)n * sqrt((double)LF_D)));   // q * D^-0.5 and v / n folded into ctx
    // row groups per wave in pass 2: amortise the block's LDS weight image, but keep >= ~768 blocks on the 256 CUs
    // (rounded UP: one round of at most 768 blocks, not a full round and a nearly empty one)
    long long quads = (long long)B * ceil_div(n, 128);
    int rgw = (int)((quads + 767) / 768);
    rgw = rgw < 1 ? 1 : (rgw > 8 ? 8 : rgw);
#define LA_LAUNCH(DIM_, TX_)                                                                                               \
    hipLaunchKernelGGL((la_kv_fused_kernel<DIM_, TX_>), dim3(nchunks, B), dim3(256), 0, s, (const TX_*)x, ldx, n, rpc,     \
                       nchunks, g_pre, eps, wq, part);                                                                     \
    UWIR_LAUNCH_CHECK("linattn_block(kv)");                                                                                \
    hipLaunchKernelGGL(la_fin_fused_kernel, dim3(B * LF_HEADS * 8), dim3(1024), 0, s, part, nchunks, inv_n_scale, ctxT);          \
    UWIR_LAUNCH_CHECK("linattn_block(fin)");                                                                               \
    hipLaunchKernelGGL((la_apply_fused_kernel<DIM_, TX_>), dim3(ceil_div(n, 128 * rgw), B), dim3(256), 0, s,               \
                       (const TX_*)x, ldx, n, g_pre, eps, wq, ctxT, wo, b_out, g_out, (TX_*)out, ldo, rgw);                \
    UWIR_LAUNCH_CHECK("linattn_block(apply)");
    if (io_dtype == UWIR_BF16) {
        if (dim == 64) { LA_LAUNCH(64, bf16_t) } else if (dim == 128) { LA_LAUNCH(128, bf16_t) } else { LA_LAUNCH(256, bf16_t) }
    } else {
        if (dim == 64) { LA_LAUNCH(64, float) } else if (dim == 128) { LA_LAUNCH(128, float) } else { LA_LAUNCH(256, float) }
    }
#undef LA_LAUNCH
    return UWIR_OK;
}

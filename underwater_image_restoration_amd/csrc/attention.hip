// Global multi-head attention, flash style (online softmax, no n x n matrix in memory).
//   S/wtb.py:94-115  (one V)   and   S/cfc.py:24-39 (one softmax(QK^T/sqrt(hd)), two V's)
//
// Block = 4 waves x 32 queries; keys/values stream through LDS in 64-key tiles (register
// prefetch of the next tile overlaps the MFMAs of the current one).  Per wave and 32-key block:
//   S^T[key][query] = K . Q^T          -- "swapped" QK^T: the query sits on the lane, so the row
//                                         max / sum are register reductions + ONE half exchange
//   P = exp2(c*S - m)                  -- deferred-max online softmax (rescale only when needed)
//   O^T[d][query] += V^T[d][key] . P   -- the 32x32 accumulator of S^T IS the B operand of this
//                                         MFMA (rows = keys = the summed index): no LDS round trip
// bf16 path: v_mfma_f32_32x32x16_bf16, V^T is staged with the key permutation that the
// accumulator-as-operand register order implies; fp32 path: v_mfma_f32_32x32x2_f32 (exact f32).
#include "common.h"
#include <type_traits>

struct AttnP {
    const void *q, *k, *v[2];
    void* o[2];
    int ldq, ldk, ldv[2], ldo[2];
    int n, heads, hd, out_dtype;
    float c;  // scale * log2(e); 1 when q is pre-scaled
    const float* kmax2;  // optional [B*heads]: max over keys of |k|^2 (bf16 fast path: a-priori score bound)
    int norm_bound_only;   // tools/: uwir_attn_variant bit 5 -- the Cauchy-Schwarz bound alone decides
    const unsigned* kext;  // packed path: [B*heads][2][32] order-preserving keys of max_k k_d and max_k (-k_d)
    int* flags;          // split launch: per-block "bound too large, redo with the online-max kernel" marks
};

#define AT_KT 64

template <typename T> struct Chunk4;
template <> struct Chunk4<bf16_t> { typedef bf16x4 type; };
template <> struct Chunk4<float> { typedef f32x4 type; };

template <int G> struct GIdx { static constexpr int value = G; };

// QG = 32-query groups per wave (1 or 2).  With 2, every staged K / V^T tile and every LDS fragment read feeds two
// independent S^T / O^T chains: the per-tile staging + barrier cost per (query, key) pair halves and each wave has
// two dependency chains to interleave.  Block = 4 waves x 32*QG queries.
// MODE 0: one launch holding both loops (per-wave choice).  Split launch for the big problems, so that the hot
// kernel carries only the fast loop's registers: MODE 1 = fast loop only, a block whose bound is too large marks
// itself in p.flags and leaves; MODE 2 = online-max loop only, run by the marked blocks (all others exit at once).
template <typename T, int HD, int NV, int QG, int MODE>
__global__ void __launch_bounds__(256) flash_attn_kernel(AttnP p) {
    constexpr bool BF = sizeof(T) == 2;
    // marks: one int per 128 queries of a (b, head); this block covers QG of them
    const int nq128 = (p.n + 127) / 128;
    int* const marks = p.flags ? p.flags + (size_t)(blockIdx.z * gridDim.y + blockIdx.y) * nq128 + blockIdx.x * QG : nullptr;
    if constexpr (MODE == 2) {
        int any = 0;
#pragma unroll
        for (int i = 0; i < QG; ++i)
            if (blockIdx.x * QG + i < nq128) any |= marks[i];
        if (any == 0) return;  // block-uniform, ahead of every barrier
    }
    constexpr int KT = AT_KT;
    constexpr int KSTR = HD + (BF ? 8 : 1);   // sK row stride (elements)
    constexpr int VSTR = KT + (BF ? 8 : 1);   // sVt row stride
    constexpr int DV = HD * NV;
    constexpr int DG = (DV + 31) / 32;         // 32-row d groups of O^T
    constexpr int NKC = (KT * (HD / 4)) / 256; // K chunks (4 elements) per thread per tile
    constexpr int QPR = HD / 4;                // chunks per key row
    // spare O^T row: a ones-row in V^T makes the PV MFMA sum P for free.  bf16 path only: the fp32 parity path
    // keeps the pairwise VALU sums (a 4096-long sequential fp32 chain costs ~3e-5 of accuracy).
    constexpr bool SUMROW = BF && (DV % 32) != 0;
    using C4 = typename Chunk4<T>::type;

    __shared__ __attribute__((aligned(16))) T sK[2][KT * KSTR];
    __shared__ __attribute__((aligned(16))) T sVt[2][DG * 32 * VSTR];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int n = p.n, hd = p.hd;
    const size_t row0 = (size_t)b * n;
    const int qbase = blockIdx.x * (128 * QG) + wave * (32 * QG) + lq;  // this lane's query in group g: qbase + 32 g
    const float c = p.c;

    const T* __restrict__ Kp = (const T*)p.k + (size_t)head * hd;
    const T* __restrict__ Vp0 = (const T*)p.v[0] + (size_t)head * hd;
    const T* __restrict__ Vp1 = NV > 1 ? (const T*)p.v[1] + (size_t)head * hd : nullptr;

    // ---- zero the V^T image once (rows / columns never written must read as 0, not NaN garbage)
    for (int i = tid; i < 2 * DG * 32 * VSTR; i += 256) (&sVt[0][0])[i] = (T)0.f;
    for (int i = tid; i < 2 * KT * KSTR; i += 256) (&sK[0][0])[i] = (T)0.f;

    // ---- Q fragments straight from global, pre-multiplied by c = scale*log2(e): the QK^T MFMA then
    //      yields scores directly in the log2 domain and exp2 needs no per-element multiply.
    constexpr int KS = BF ? HD / 16 : HD / 2;
    typedef typename std::conditional<BF, bf16x8, float>::type QF;
    QF qf[QG][KS];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const T* Qp = (const T*)p.q + (size_t)head * hd;
        const int q_idx = qbase + 32 * g;
        const bool qok = q_idx < n;
        const T* qrow = Qp + (row0 + (qok ? q_idx : 0)) * p.ldq;
        if constexpr (BF) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 f;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    int d = 16 * ks + 8 * lh + 4 * half;
                    bf16x4 v;
                    if (qok && d < hd) v = *reinterpret_cast<const bf16x4*>(qrow + d);
                    else { v[0] = v[1] = v[2] = v[3] = (bf16_t)0.f; }
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[4 * half + e] = c == 1.f ? v[e] : (bf16_t)((float)v[e] * c);
                }
                qf[g][ks] = f;
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                int d = 2 * ks + lh;
                qf[g][ks] = (qok && d < hd) ? (float)qrow[d] * c : 0.f;
            }
        }
    }

    // ---- staging registers.  Rows past n are CLAMPED to the last key, not zero-filled: every consumer masks keys
    //      >= n itself (P = 0 there), and a real row is finite, so no predicate / zero-fill sits in the loop.
    C4 rk[NKC], rv[NV][NKC];
    auto zero4 = [](C4& v) { v[0] = (T)0.f; v[1] = (T)0.f; v[2] = (T)0.f; v[3] = (T)0.f; };
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NKC; ++i) {
            int cidx = tid + 256 * i;
            int key = min(k0 + cidx / QPR, n - 1), d = 4 * (cidx % QPR);
            size_t r = row0 + key;
            if (HD == hd || d < hd) {
                rk[i] = *reinterpret_cast<const C4*>(Kp + r * p.ldk + d);
                rv[0][i] = *reinterpret_cast<const C4*>(Vp0 + r * p.ldv[0] + d);
                if constexpr (NV > 1) rv[1][i] = *reinterpret_cast<const C4*>(Vp1 + r * p.ldv[1] + d);
            } else {
                zero4(rk[i]);
                zero4(rv[0][i]);
                if constexpr (NV > 1) zero4(rv[1][i]);
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NKC; ++i) {
            int cidx = tid + 256 * i;
            int key = cidx / QPR, d = 4 * (cidx % QPR);
            if constexpr (BF) {
                *reinterpret_cast<bf16x4*>(&sK[buf][key * KSTR + d]) = rk[i];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) sK[buf][key * KSTR + d + e] = rk[i][e];
            }
            int pos = key;
            if constexpr (BF) {
                int kk = key & 15;
                pos = (key & ~15) + 8 * ((kk >> 2) & 1) + 4 * (kk >> 3) + (kk & 3);
            }
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) sVt[buf][(v * HD + d + e) * VSTR + pos] = rv[v][i][e];
        }
    };

    f32x16 O[QG][DG];
#pragma unroll
    for (int g = 0; g < QG; ++g)
#pragma unroll
        for (int dg = 0; dg < DG; ++dg)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[g][dg][r] = 0.f;
    f32x16 Lacc[QG];  // fast path without a spare V^T row: every row accumulates sum_k P (ones-operand MFMA)
#pragma unroll
    for (int g = 0; g < QG; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) Lacc[g][r] = 0.f;
    // Online-max loop: the running max m (log2 units) enters the scores through ONE extra QK^T MFMA k-step instead
    // of 16 VALU subtractions per block: K is extended by constant columns [1, 1, mask] and Q by
    // [-m_hi, -m_lo, -BIG] (m split in two bf16 terms, exact in the fp32 accumulate), so the MFMA
    // emits c*S - m directly, keys >= n are pushed to -BIG (exp2 -> 0) and the numerator is a bare exp2.
    float m_eff[QG], l_run[QG];
    bool first[QG];        // the first block adopts its own max unconditionally (wave-uniform)
    QF qext[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        m_eff[g] = 0.f;
        l_run[g] = 0.f;    // this lane's partial denominator (unused when the ones-row provides it)
        first[g] = true;
        if constexpr (BF) {
#pragma unroll
            for (int j = 0; j < 8; ++j) qext[g][j] = (bf16_t)0.f;
            qext[g][2] = (bf16_t)(-30000.f);
        } else {
            qext[g] = lh == 0 ? 0.f : -1e30f;
        }
    }
    constexpr float THR = BF ? 4.f : 0.f;
    const int kbase = lq * KSTR + (BF ? 8 : 1) * lh;
    const int vbase = lq * VSTR + (BF ? 8 * lh : 0);

    auto block = [&](const int kb, const int buf, const int k0, auto gtag) {
        constexpr int g = decltype(gtag)::value;
        const bool masked = (k0 + kb * 32 + lq) >= n;
        // ---- S^T = K . Q^T - m
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        if constexpr (BF) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(&sK[buf][kb * 32 * KSTR + kbase + 16 * ks]);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[g][ks], s, 0, 0, 0);
            }
            bf16x8 kext;
            const bf16_t one = (bf16_t)(lh == 0 ? 1.f : 0.f);
#pragma unroll
            for (int j = 0; j < 8; ++j) kext[j] = (bf16_t)0.f;
            kext[0] = one;
            kext[1] = one;
            kext[2] = (bf16_t)((lh == 0 && masked) ? 1.f : 0.f);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kext, qext[g], s, 0, 0, 0);
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float kf = sK[buf][kb * 32 * KSTR + kbase + 2 * ks];
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf, qf[g][ks], s, 0, 0, 0);
            }
            const float kext = lh == 0 ? 1.f : (masked ? 1.f : 0.f);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kext, qext[g], s, 0, 0, 0);
        }
        // ---- online softmax, deferred max: rescale only when the block max exceeds the running max by > THR
        float mxa = fmaxf(s[0], s[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) mxa = fmaxf(fmaxf(mxa, s[r]), s[r + 1]);
        {
            unsigned u = __float_as_uint(mxa);
            auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            mxa = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        if (first[g] || __any(mxa > THR)) {
            const float delta = first[g] ? mxa : fmaxf(mxa, 0.f);
            float m_new = m_eff[g] + delta;
            if constexpr (BF) {
                const bf16_t hi = (bf16_t)m_new;
                const bf16_t lo = (bf16_t)(m_new - (float)hi);
                m_new = (float)hi + (float)lo;  // exactly what the next MFMA will subtract
                qext[g][0] = (bf16_t)(-(float)hi);
                qext[g][1] = (bf16_t)(-(float)lo);
            } else {
                if (lh == 0) qext[g] = -m_new;
            }
            const float dlt = m_new - m_eff[g];
            m_eff[g] = m_new;
            const float alpha = first[g] ? 1.f : __builtin_amdgcn_exp2f(-dlt);
            first[g] = false;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] -= dlt;
            l_run[g] *= alpha;
#pragma unroll
            for (int dg = 0; dg < DG; ++dg)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[g][dg][r] *= alpha;
        }
        float pr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) pr[r] = __builtin_amdgcn_exp2f(s[r]);
        if constexpr (!SUMROW) {
            float l0 = 0.f, l1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) { l0 += pr[r]; l1 += pr[r + 1]; }
            l_run[g] += l0 + l1;
        }
        // ---- O^T += V^T . P   (with SUMROW, row DV of V^T is all ones, so O^T[DV][q] accumulates sum_k P)
        if constexpr (BF) {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                bf16x8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (bf16_t)pr[8 * ss + j];
#pragma unroll
                for (int dg = 0; dg < DG; ++dg) {
                    bf16x8 vf = *reinterpret_cast<const bf16x8*>(&sVt[buf][dg * 32 * VSTR + vbase + kb * 32 + 16 * ss]);
                    O[g][dg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, O[g][dg], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
                for (int dg = 0; dg < DG; ++dg) {
                    float vf = sVt[buf][dg * 32 * VSTR + vbase + key];
                    O[g][dg] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, pr[r], O[g][dg], 0, 0, 0);
                }
            }
        }
    };

    // ---- bf16 fast path: a-priori score bound.  |c*q.k| <= |c*q| * max_k|k| =: ub (Cauchy-Schwarz on the exact bf16
    // operands).  The softmax shift is a per-query constant that cancels in O / l, so it only has to keep exp2 in
    // range: with ub <= 90 every exp2 argument lies in [-90, 90], P = 2^s (bf16, 8-bit exponent) is normal, the fp32
    // row sums l <= n * 2^90 <= 2^107 (n <= 2^17) and O <= l * max|v| (finite for |v| < 2^20) stay in range, and NO
    // shift is applied at all -- no running max, no rescale, no extra MFMA k-step,
    // same relative accuracy as the online-max form.  Waves with a larger bound (wave-uniform test) take the
    // online-max loop; both loops stage the same tiles behind the same barriers.
    bool fast = false;
    if constexpr (BF && MODE != 2) {
        if (MODE == 1 || p.kmax2) {
            bool ok = true;
            const float km2 = p.kmax2[b * p.heads + head];
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                float q2 = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) { float t = (float)qf[g][ks][j]; q2 = fmaf(t, t, q2); }
                unsigned u = __float_as_uint(q2);
                auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                q2 = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                const float ub = sqrtf(q2 * km2) * 1.002f + 1e-3f;
                ok = ok && ub <= 90.f;
            }
            if constexpr (MODE == 1) {
                if (__syncthreads_or(!ok)) {  // block-uniform: leave the whole block to the online-max launch
                    if (tid < QG && blockIdx.x * QG + tid < nq128) marks[tid] = 1;
                    return;
                }
                fast = true;
            } else {
                fast = __all(ok);
            }
        }
    }
    // one 32-key block of the fast path for all QG query groups; TAIL: keys >= n get P = 0 explicitly
    auto block_fast = [&](const int kb, const int buf, const int k0, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        if constexpr (BF) {
            f32x16 s[QG];
#pragma unroll
            for (int g = 0; g < QG; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[g][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(&sK[buf][kb * 32 * KSTR + kbase + 16 * ks]);
#pragma unroll
                for (int g = 0; g < QG; ++g) s[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[g][ks], s[g], 0, 0, 0);
            }
            bf16x8 pb[QG][2];
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                float pr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) pr[r] = __builtin_amdgcn_exp2f(s[g][r]);
                if constexpr (TAIL) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        pr[r] = key < n ? pr[r] : 0.f;
                    }
                }
#pragma unroll
                for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pb[g][ss][j] = (bf16_t)pr[8 * ss + j];
            }
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
#pragma unroll
                for (int dg = 0; dg < DG; ++dg) {
                    bf16x8 vf = *reinterpret_cast<const bf16x8*>(&sVt[buf][dg * 32 * VSTR + vbase + kb * 32 + 16 * ss]);
#pragma unroll
                    for (int g = 0; g < QG; ++g)
                        O[g][dg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[g][ss], O[g][dg], 0, 0, 0);
                }
                if constexpr (!SUMROW) {  // no spare V^T row: the denominators come from a constant all-ones A operand
                    bf16x8 ones;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.f;
#pragma unroll
                    for (int g = 0; g < QG; ++g)
                        Lacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pb[g][ss], Lacc[g], 0, 0, 0);
                }
            }
        }
    };

    const int ntiles = (n + KT - 1) / KT;
    __syncthreads();
    if constexpr (SUMROW) {  // ones row (after the zero fill, before the first tile lands)
        for (int i = tid; i < 2 * VSTR; i += 256) sVt[i / VSTR][DV * VSTR + (i % VSTR)] = (T)1.f;
    }
    load_tile(0);
    store_tile(0);
    __syncthreads();

    if (MODE == 1 || (MODE == 0 && fast)) {
        // two tiles per trip with compile-time buffer indices: halves the loop-carried accumulator copies the
        // register allocator leaves at the back edge and folds the LDS buffer offsets into immediates
        auto tile_fast = [&](const int t, const int buf) {
            if (t + 1 < ntiles) load_tile((t + 1) * KT);
            const int k0 = t * KT;
            if (k0 + KT <= n) {
                block_fast(0, buf, k0, std::false_type{});
                block_fast(1, buf, k0, std::false_type{});
            } else {
                block_fast(0, buf, k0, std::true_type{});
                if (k0 + 32 < n) block_fast(1, buf, k0, std::true_type{});
            }
            if (t + 1 < ntiles) store_tile(buf ^ 1);
            __syncthreads();
        };
        int t = 0;
        for (; t + 1 < ntiles; t += 2) {
            tile_fast(t, 0);
            tile_fast(t + 1, 1);
        }
        if (t < ntiles) tile_fast(t, 0);
    } else if constexpr (MODE != 1) {
        for (int t = 0; t < ntiles; ++t) {
            const int buf = t & 1;
            if (t + 1 < ntiles) load_tile((t + 1) * KT);
            const int k0 = t * KT;
            auto all_groups = [&](const int kb) {
                block(kb, buf, k0, GIdx<0>{});
                if constexpr (QG > 1) block(kb, buf, k0, GIdx<1>{});
                if constexpr (QG > 2) block(kb, buf, k0, GIdx<2>{});
                if constexpr (QG > 3) block(kb, buf, k0, GIdx<3>{});
                static_assert(QG <= 4, "at most 4 query groups per wave");
            };
            all_groups(0);
            if (k0 + 32 < n) all_groups(1);
            if (t + 1 < ntiles) store_tile(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- finish: denominators (ones-row, ones-operand accumulator or summed halves), normalise, store
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        float l;
        if constexpr (SUMROW) {
            constexpr int lr = DV % 32, rl = (lr & 3) + 4 * (lr >> 3), hl = (lr >> 2) & 1;
            unsigned u = __float_as_uint(O[g][DG - 1][rl]);
            auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            l = __uint_as_float(sw[hl]);
        } else if (MODE != 2 && fast) {
            l = Lacc[g][0];  // every row of Lacc holds sum_k P[k][q] for this lane's query
        } else {
            unsigned u = __float_as_uint(l_run[g]);
            auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        const int q_idx = qbase + 32 * g;
        if (q_idx >= n) continue;
        const float inv = 1.f / l;
#pragma unroll
        for (int dg = 0; dg < DG; ++dg) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int d = dg * 32 + 8 * rq + 4 * lh;  // rows d..d+3 live in regs 4*rq..4*rq+3
                const int v = d / HD, dd = d % HD;
                if (v < NV && dd < hd) {
                    float4 o = make_float4(O[g][dg][4 * rq] * inv, O[g][dg][4 * rq + 1] * inv, O[g][dg][4 * rq + 2] * inv,
                                           O[g][dg][4 * rq + 3] * inv);
                    size_t off = (row0 + q_idx) * p.ldo[v] + (size_t)head * hd + dd;
                    if (p.out_dtype == UWIR_F32) store4((float*)p.o[v] + off, o);
                    else store4((bf16_t*)p.o[v] + off, o);
                }
            }
        }
    }
}

// max over the keys of one (b, head) of |k|^2, for the fast path's score bound.  grid (chunks, heads, B); positive
// floats order like their bit patterns, so the cross-block max is an integer atomicMax on a zeroed word.
#define KMAX_ROWS 2048
__global__ void __launch_bounds__(256) attn_kmax_kernel(const bf16_t* __restrict__ k, int ldk, int n, int heads, int hd,
                                                        unsigned* __restrict__ out) {
    const int head = blockIdx.y, b = blockIdx.z;
    const int r0 = blockIdx.x * KMAX_ROWS, r1 = min(n, r0 + KMAX_ROWS);
    float m = 0.f;
    for (int r = r0 + threadIdx.x; r < r1; r += 256) {
        const bf16_t* kr = k + ((size_t)b * n + r) * ldk + (size_t)head * hd;
        float s = 0.f;
        for (int d = 0; d < hd; d += 4) {
            float4 v = load4(kr + d);
            s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
        }
        m = fmaxf(m, s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out + b * heads + head, __float_as_uint(m));
}

// ------------------------------------------------------------------------------------------
// Packed-stream fast path (bf16, hd 16 / 32, long sequences).  The staging work of the LDS kernel above -- strided
// K / V gathers, the V transposition through 2-byte LDS stores, a barrier per 64 keys -- is per (block, tile); at
// hd = 16 the loop is bound by exp issue and every other instruction in it costs throughput.  Here a pre-pass
// (attn_pack_kernel, one read of K and V per call, it also yields max|k|^2 for the score bound) lays both operands out
// exactly as the MFMA fragments want them:
//   Kp  [b*heads+head][npad keys][HD]                      one 32-key block = one contiguous run
//   Vtp [b*heads+head][npad/16 key groups][ROWS][16 keys]  V^T rows (d of V1, then V2, then a ones row when a row of
//                                                          the 32-row MFMA tile is spare), keys in accumulator order
// and the main kernel has no LDS and no barrier: every wave streams 16-byte fragments straight into registers with
// buffer loads (per-lane offset fixed, the block offset is a scalar), two key blocks ahead of their use, for 32*QG
// queries.  Keys in [n, npad) are packed as K = 0, V = 0, ones = 0: they add exp2(0) * 0 to every sum (without a
// ones row the denominators would count them as 1 each: the last block of a ragged sequence runs with their
// exponentials masked to 0).
// ------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define PK_KEYS 256  // keys per block of the pack pre-pass

template <int HD, int NV> struct PackCfg {
    static constexpr int DV = HD * NV, DG = (DV + 31) / 32;
    static constexpr bool SUMROW = (DV % 32) != 0;
    static constexpr int ROWS = DV + (SUMROW ? 1 : 0);
};

__host__ __device__ static inline size_t pk_align256(size_t v) { return (v + 255) & ~(size_t)255; }

// order-preserving map float -> unsigned (an integer atomicMax on a zeroed word then is a float max) and back
__device__ __forceinline__ unsigned attn_okey(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float attn_okey_inv(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

#define PK_TILES 4   // 256-key tiles per block of the pack pre-pass (the next tile's loads fly during the current gather)

template <int HD, int NV>
__global__ void __launch_bounds__(256) attn_pack_kernel(const bf16_t* __restrict__ k, int ldk, const bf16_t* __restrict__ v1,
                                                        int ldv1, const bf16_t* __restrict__ v2, int ldv2, int n, int npad,
                                                        int heads, bf16_t* __restrict__ Kp, bf16_t* __restrict__ Vtp,
                                                        unsigned* __restrict__ kmax2, unsigned* __restrict__ kext) {
    using P = PackCfg<HD, NV>;
    constexpr int DV = P::DV, ROWS = P::ROWS, SV = DV + 8;  // LDS row stride (elements): rows stay 16-byte aligned
    constexpr int PPK = HD / 8;                             // 16-byte pieces per key row
    constexpr int NIT = PK_KEYS * PPK / 256;
    __shared__ __attribute__((aligned(16))) bf16_t sV[PK_KEYS * SV];
    const int tid = threadIdx.x, head = blockIdx.y, b = blockIdx.z;
    const int bh = b * heads + head;
    float m = 0.f;
    // per-dimension extremes of this thread's 8 dimensions (its piece of a key row is the same in every item: 256 % PPK
    // == 0); keys in [n, npad) count as the zeros they are packed as
    float dmax[8], dmin[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { dmax[e] = -3.0e38f; dmin[e] = 3.0e38f; }
    bf16x8 kv[NIT], vv[NIT][NV];
    // 16-byte pieces, adjacent lanes = adjacent pieces of one key row
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it;
            const int key = k0 + idx / PPK, piece = idx % PPK;
#pragma unroll
            for (int e = 0; e < 8; ++e) kv[it][e] = (bf16_t)0.f;
#pragma unroll
            for (int t = 0; t < NV; ++t) vv[it][t] = kv[it];
            if (key < n) {
                const size_t r = (size_t)b * n + key;
                kv[it] = *reinterpret_cast<const bf16x8*>(k + r * ldk + (size_t)head * HD + piece * 8);
                vv[it][0] = *reinterpret_cast<const bf16x8*>(v1 + r * ldv1 + (size_t)head * HD + piece * 8);
                if constexpr (NV > 1) vv[it][1] = *reinterpret_cast<const bf16x8*>(v2 + r * ldv2 + (size_t)head * HD + piece * 8);
            }
        }
    };
    const int kb0 = blockIdx.x * (PK_KEYS * PK_TILES);
    load_tile(kb0);
    for (int t = 0; t < PK_TILES; ++t) {
        const int k0 = kb0 + t * PK_KEYS;
        if (k0 >= npad) break;
        // ---- phase 1: K straight to its packed place, V rows into LDS for the transposition
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it;
            const int kl = idx / PPK, piece = idx % PPK, key = k0 + kl;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = (float)kv[it][e];
                s = fmaf(f, f, s);
                if (key < npad) { dmax[e] = fmaxf(dmax[e], f); dmin[e] = fminf(dmin[e], f); }
            }
#pragma unroll
            for (int off = 1; off < PPK; off <<= 1) s += __shfl_xor(s, off);   // the PPK lanes of one key
            m = fmaxf(m, s);
            if (key < npad) *reinterpret_cast<bf16x8*>(Kp + ((size_t)bh * npad + key) * HD + piece * 8) = kv[it];
#pragma unroll
            for (int u = 0; u < NV; ++u) *reinterpret_cast<bf16x8*>(&sV[kl * SV + u * HD + piece * 8]) = vv[it][u];
        }
        __syncthreads();
        if (t + 1 < PK_TILES && k0 + PK_KEYS < npad) load_tile(k0 + PK_KEYS);
        // ---- phase 2: one item = one V^T row of one 16-key group: 16 keys gathered in accumulator order, one 32-byte
        //      store; adjacent lanes = adjacent rows, so a key group's ROWS x 32 bytes leave as one contiguous run
        for (int item = tid; item < (PK_KEYS / 16) * ROWS; item += 256) {
            const int kg = item / ROWS, row = item - kg * ROWS;
            const int kbase = k0 + kg * 16;
            if (kbase >= npad) continue;
            bf16x8 o[2];
#pragma unroll
            for (int pos = 0; pos < 16; ++pos) {
                const int kk = 8 * ((pos >> 2) & 1) + 4 * (pos >> 3) + (pos & 3);  // bits 2 and 3 swapped (an involution)
                bf16_t val;
                if (row < DV) val = sV[(kg * 16 + kk) * SV + row];
                else val = (bf16_t)((kbase + kk) < n ? 1.f : 0.f);
                o[pos >> 3][pos & 7] = val;
            }
            bf16_t* dst = Vtp + (((size_t)bh * (npad / 16) + kbase / 16) * ROWS + row) * 16;
            *reinterpret_cast<bf16x8*>(dst) = o[0];
            *reinterpret_cast<bf16x8*>(dst + 8) = o[1];
        }
        __syncthreads();
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((tid & 63) == 0 && m > 0.f) atomicMax(kmax2 + bh, __float_as_uint(m));
    // lanes with the same piece (tid % PPK) hold the same dimensions
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int off = PPK; off < 64; off <<= 1) {
            dmax[e] = fmaxf(dmax[e], __shfl_xor(dmax[e], off));
            dmin[e] = fminf(dmin[e], __shfl_xor(dmin[e], off));
        }
    }
    if ((tid & 63) < PPK && dmax[0] >= dmin[0]) {   // (a block wholly past npad saw no key)
        const int d0 = (tid & 63) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            atomicMax(kext + (size_t)bh * 64 + d0 + e, attn_okey(dmax[e]));
            atomicMax(kext + (size_t)bh * 64 + 32 + d0 + e, attn_okey(-dmin[e]));
        }
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Denominators when the V^T tile has no spare row for a ones row (LSV): 0 = a ones-operand 32x32x16 MFMA per 16 keys
// (every one of its 32 output rows is the same sum: 64 MFMA cycles per 32 x 32 block); 1 = packed fp32 adds of the
// exponentials on the VALU (8 per block, on the pipe that bounds this kernel); 2 = a 16x16x32 MFMA whose A operand is
// a 0/1 selector: fed the 32x32x16 B fragment as it is, lane group G of its k dimension holds queries 16 (G & 1) + n,
// key half G >> 1, so row 0 = groups {0, 2} and row 1 = groups {1, 3} give the sums of queries n and n + 16: half the
// MFMA cycles of (0) and no VALU work.
template <int HD, int NV, int QG, int PF, int LSV>
__global__ void __launch_bounds__(256) attn_stream_kernel(AttnP p, const bf16_t* __restrict__ Kp,
                                                          const bf16_t* __restrict__ Vtp, int npad, int nqb, int nblocks) {
    using P = PackCfg<HD, NV>;
    constexpr int DV = P::DV, DG = P::DG, ROWS = P::ROWS, KS = HD / 16;
    constexpr bool SUMROW = P::SUMROW;
    constexpr int NSLOT = PF + 1;
    // XCD-aware bijective remap: the blocks that share an XCD (bid % 8) walk a contiguous range of (b, head, query
    // block), so the K / V^T stream of one (b, head) is served by one L2
    int vid = blockIdx.x;
    {
        int q = nblocks / 8, r = nblocks % 8, xcd = vid % 8;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + vid / 8;
    }
    const int bh = vid / nqb, qb = vid - bh * nqb;
    const int b = bh / p.heads, head = bh - b * p.heads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, lh = lane >> 5;
    const int n = p.n;
    const size_t row0 = (size_t)b * n;
    const int qbase = qb * (128 * QG) + wave * (32 * QG) + lq;
    const float c = p.c;

    // ---- Q fragments (pre-multiplied by c: scores come out of the MFMA in the log2 domain) and the score bound
    // Two a-priori bounds on |score| of a query against EVERY key of its (b, head), the smaller one decides:
    //   |c q| max_k |k|                                               (Cauchy-Schwarz)
    //   max(sum_d max(t_d kmax_d, t_d kmin_d), -sum_d min(...))       (per dimension: k_d lies in [kmin_d, kmax_d])
    // The second is the tight one when a few keys carry a large norm in a direction the query does not share: on the
    // CFC attentions of the 512 x 512 student it lowers the share of queries above 90 from 36 - 73 % to 0.2 - 2 %
    // (tools/exp/attn_bound_tight.py).
    bf16x8 qf[QG][KS];
    bool ok = true;
    const float km2 = p.kmax2[bh];
    float kdmax[KS][8], kdmin[KS][8];   // this lane's dimensions: 16 ks + 8 lh + 0..7
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const unsigned* ke = p.kext + (size_t)bh * 64 + 16 * ks + 8 * lh;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            kdmax[ks][e] = attn_okey_inv(ke[e]);
            kdmin[ks][e] = -attn_okey_inv(ke[32 + e]);
        }
    }
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const int q_idx = qbase + 32 * g;
        const bool qok = q_idx < n;
        const bf16_t* qrow = (const bf16_t*)p.q + (size_t)head * HD + (row0 + (qok ? q_idx : 0)) * p.ldq;
        float q2 = 0.f, hi = 0.f, lo = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 f;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                bf16x4 v = *reinterpret_cast<const bf16x4*>(qrow + 16 * ks + 8 * lh + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bf16_t t = qok ? (c == 1.f ? v[e] : (bf16_t)((float)v[e] * c)) : (bf16_t)0.f;
                    f[4 * half + e] = t;
                    const float tf = (float)t;
                    q2 = fmaf(tf, tf, q2);
                    const float a = tf * kdmax[ks][4 * half + e], b2 = tf * kdmin[ks][4 * half + e];
                    hi += fmaxf(a, b2);
                    lo += fminf(a, b2);
                }
            }
            qf[g][ks] = f;
        }
        // the other half-wave holds the other 8 dimensions of every 16 of the same query
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(q2), __float_as_uint(q2), false, false);
        q2 = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        auto sh = __builtin_amdgcn_permlane32_swap(__float_as_uint(hi), __float_as_uint(hi), false, false);
        hi = __uint_as_float(sh[0]) + __uint_as_float(sh[1]);
        auto sl = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(lo), false, false);
        lo = __uint_as_float(sl[0]) + __uint_as_float(sl[1]);
        const float ub = p.norm_bound_only ? sqrtf(q2 * km2) : fminf(sqrtf(q2 * km2), fmaxf(hi, -lo));
        ok = ok && (ub * 1.002f + 1e-3f <= 90.f);
    }
    // wave-uniform (the kernel has no barrier): a wave whose 32 * QG queries do not all pass marks their 128-query
    // group(s) for the online-max launch and leaves; the other waves of the block carry on
    if (__any(!ok)) {
        const int q0 = qb * (128 * QG) + wave * (32 * QG);   // first query of this wave
        if (lane < (QG + 3) / 4 && q0 + 128 * lane < n)
            p.flags[(size_t)bh * ((n + 127) / 128) + q0 / 128 + lane] = 1;
        return;
    }

    // ---- fragment streams
    const size_t kbytes = (size_t)npad * HD * 2, vbytes = (size_t)(npad / 16) * ROWS * 32;
    __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)Kp + (size_t)bh * kbytes), 0,
                                                                   (int)kbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)Vtp + (size_t)bh * vbytes), 0,
                                                                   (int)vbytes, 0x00020000);
    const int kvo = lq * (HD * 2) + lh * 16;
    int vvo[DG];
#pragma unroll
    for (int dg = 0; dg < DG; ++dg) vvo[dg] = min(lq + 32 * dg, ROWS - 1) * 32 + lh * 16;

    u32x4 kf[NSLOT][KS], vf[NSLOT][2][DG];
    auto issue = [&](const int blk, auto slot_tag) {
        constexpr int S = decltype(slot_tag)::value;
        const int ko = blk * (32 * HD * 2), vo = blk * (2 * ROWS * 32);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[S][ks] = __builtin_amdgcn_raw_buffer_load_b128(krs, kvo + ks * 32, ko, 0);
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
            for (int dg = 0; dg < DG; ++dg)
                vf[S][ss][dg] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vvo[dg], vo + ss * (ROWS * 32), 0);
    };

    f32x16 O[QG][DG], Lacc[QG];
    f32x2 lsum[QG];
    f32x4 Lsel[QG];
    bf16x8 sel;   // row 0 of A lives in lanes 0 / 32 (groups 0, 2), row 1 in lanes 17 / 49 (groups 1, 3)
    {
        const bf16_t sv = (bf16_t)((lane == 0 || lane == 32 || lane == 17 || lane == 49) ? 1.f : 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) sel[j] = sv;
    }
#pragma unroll
    for (int g = 0; g < QG; ++g) {
#pragma unroll
        for (int dg = 0; dg < DG; ++dg)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[g][dg][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) Lacc[g][r] = 0.f;
        lsum[g] = f32x2{0.f, 0.f};
        Lsel[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.f;

    // MASK (the last key block of a ragged sequence, forms without a ones row): the exponentials of the keys in
    // [n, npad) are set to 0.  Those keys are packed as zeros, so each would add exp2(0) = 1 to a row sum; subtracting
    // their count afterwards cancels catastrophically when every real score of a query is very negative.
    auto compute = [&](auto slot_tag, auto mask_tag, const int keys_left) {
        constexpr int S = decltype(slot_tag)::value;
        constexpr bool MASK = decltype(mask_tag)::value != 0;
        f32x16 s[QG];
#pragma unroll
        for (int g = 0; g < QG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[g][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 kk = __builtin_bit_cast(bf16x8, kf[S][ks]);
#pragma unroll
            for (int g = 0; g < QG; ++g) s[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kk, qf[g][ks], s[g], 0, 0, 0);
        }
        bf16x8 pb[QG][2];
#pragma unroll
        for (int g = 0; g < QG; ++g) {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    f32x2 e = {__builtin_amdgcn_exp2f(s[g][8 * ss + j]), __builtin_amdgcn_exp2f(s[g][8 * ss + j + 1])};
                    if constexpr (MASK) {   // accumulator register r of the 32 x 32 tile = key 8 (r / 4) + 4 lh + r % 4
                        const int r = 8 * ss + j, key = 8 * (r >> 2) + 4 * lh + (r & 3);
                        e[0] = key < keys_left ? e[0] : 0.f;
                        e[1] = key + 1 < keys_left ? e[1] : 0.f;
                    }
                    if constexpr (LSV == 1 && !SUMROW) lsum[g] += e;
                    pb[g][ss][j] = (bf16_t)e[0];
                    pb[g][ss][j + 1] = (bf16_t)e[1];
                }
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
#pragma unroll
            for (int dg = 0; dg < DG; ++dg) {
                const bf16x8 vv = __builtin_bit_cast(bf16x8, vf[S][ss][dg]);
#pragma unroll
                for (int g = 0; g < QG; ++g)
                    O[g][dg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vv, pb[g][ss], O[g][dg], 0, 0, 0);
            }
            if constexpr (!SUMROW && LSV == 0) {
#pragma unroll
                for (int g = 0; g < QG; ++g)
                    Lacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pb[g][ss], Lacc[g], 0, 0, 0);
            }
            if constexpr (!SUMROW && LSV == 2) {
#pragma unroll
                for (int g = 0; g < QG; ++g)
                    Lsel[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel, pb[g][ss], Lsel[g], 0, 0, 0);
            }
        }
    };

    const bool ragged = !SUMROW && npad != n;   // uniform: the last block then runs on its own, masked
    const int nblk = npad / 32 - (ragged ? 1 : 0);
    int i = 0;
    if constexpr (PF == 2) {
        issue(0, GIdx<0>{});
        issue(1, GIdx<1>{});
        for (; i + 3 <= nblk; i += 3) {  // loads past the last block return zeros (buffer bounds) and are never used
            issue(i + 2, GIdx<2>{});
            compute(GIdx<0>{}, GIdx<0>{}, 32);
            issue(i + 3, GIdx<0>{});
            compute(GIdx<1>{}, GIdx<0>{}, 32);
            issue(i + 4, GIdx<1>{});
            compute(GIdx<2>{}, GIdx<0>{}, 32);
        }
        if (i < nblk) compute(GIdx<0>{}, GIdx<0>{}, 32);
        if (i + 1 < nblk) compute(GIdx<1>{}, GIdx<0>{}, 32);
    } else {
        issue(0, GIdx<0>{});
        for (; i + 2 <= nblk; i += 2) {
            issue(i + 1, GIdx<1>{});
            compute(GIdx<0>{}, GIdx<0>{}, 32);
            issue(i + 2, GIdx<0>{});
            compute(GIdx<1>{}, GIdx<0>{}, 32);
        }
        if (i < nblk) compute(GIdx<0>{}, GIdx<0>{}, 32);
    }
    if constexpr (!SUMROW) {
        if (ragged) {
            issue(nblk, GIdx<0>{});
            compute(GIdx<0>{}, GIdx<1>{}, n - nblk * 32);
        }
    }

    // ---- denominators, normalise, store (same accumulator layout as flash_attn_kernel)
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        float l;
        if constexpr (SUMROW) {
            constexpr int lr = DV % 32, rl = (lr & 3) + 4 * (lr >> 3), hl = (lr >> 2) & 1;
            unsigned u = __float_as_uint(O[g][DG - 1][rl]);
            auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            l = __uint_as_float(sw[hl]);
        } else if constexpr (LSV == 2) {   // lane n < 16 holds the sums of queries n (row 0) and n + 16 (row 1)
            const float l0 = __shfl(Lsel[g][0], lq & 15), l1 = __shfl(Lsel[g][1], lq & 15);
            l = lq < 16 ? l0 : l1;
        } else if constexpr (LSV == 1) {   // this lane's keys, then the other half-wave's (same query, the other 16 of 32)
            unsigned u = __float_as_uint(lsum[g][0] + lsum[g][1]);
            auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        } else {
            l = Lacc[g][0];
        }
        const int q_idx = qbase + 32 * g;
        if (q_idx >= n) continue;
        const float inv = 1.f / l;
#pragma unroll
        for (int dg = 0; dg < DG; ++dg) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int d = dg * 32 + 8 * rq + 4 * lh;
                const int v = d / HD, dd = d % HD;
                if (v < NV) {
                    float4 o = make_float4(O[g][dg][4 * rq] * inv, O[g][dg][4 * rq + 1] * inv, O[g][dg][4 * rq + 2] * inv,
                                           O[g][dg][4 * rq + 3] * inv);
                    size_t off = (row0 + q_idx) * p.ldo[v] + (size_t)head * HD + dd;
                    if (p.out_dtype == UWIR_F32) store4((float*)p.o[v] + off, o);
                    else store4((bf16_t*)p.o[v] + off, o);
                }
            }
        }
    }
}

// workspace: [B*heads] max|k|^2, one int per 128-query group, [B*heads][2][32] per-dimension key extremes (packed path),
// then (packed path) Kp and Vtp, 256-byte aligned
static size_t attn_hdr_bytes(int B, int n, int heads) {
    return ((size_t)B * heads + (size_t)ceil_div(n, 128) * heads * B + (size_t)B * heads * 64) * sizeof(float);
}
static bool attn_packed_shape(int B, int n, int heads, int hd, int nv) {
    // (n = 1024, hd 32, B = 8: 60.4 us on the LDS kernel, 44.8 us packed at 16 heads; 37.5 / 33.9 at 8 heads)
    return (hd == 16 || hd == 32) && (nv == 1 || nv == 2) && n >= 1024 &&
           (long long)ceil_div(n, 256) * heads * B >= 256;
}
static size_t attn_packed_bytes(int B, int n, int heads, int hd, int nv) {
    const int npad = ceil_div(n, 32) * 32, dv = hd * nv, rows = dv + ((dv % 32) ? 1 : 0);
    const size_t bh = (size_t)B * heads;
    return pk_align256(attn_hdr_bytes(B, n, heads)) + pk_align256(bh * npad * hd * 2) + pk_align256(bh * (npad / 16) * rows * 32);
}

// Measurement hook (bench.py's roofline leg): a pair of HIP events armed for the NEXT uwir_flash_attn call of this
// thread is recorded immediately before and after the launch of the call's MAIN kernel only (not the pack / bound
// pre-pass, not the fallback launch), on the call's stream.  Thread-local like the error string; cleared by use.
static thread_local hipEvent_t g_probe[2] = {nullptr, nullptr};
extern "C" int uwir_flash_attn_probe(void* ev_before, void* ev_after) {
    g_probe[0] = (hipEvent_t)ev_before;
    g_probe[1] = (hipEvent_t)ev_after;
    return UWIR_OK;
}
static inline void probe_mark(int which, hipStream_t stream) {
    if (g_probe[which]) {
        (void)hipEventRecord(g_probe[which], stream);
        g_probe[which] = nullptr;
    }
}

static int g_attn_lsv = -1;   // tools/: -1 = default; bits 0-1: denominator form (see attn_stream_kernel) unless bit 3 is set, bit 2: (QG, PF) = (2, 1) instead of (4, 2) for hd16 2V, bit 4: at most two stream blocks per CU (56 KiB of unused dynamic LDS per block), bit 5: score bound from max|k| alone, bit 6: 256-query fallback blocks, bit 7: (QG, PF) = (4, 2) for hd16 2V, bit 8: (2, 1) for hd32 1V
extern "C" void uwir_attn_variant(int lsv) { g_attn_lsv = lsv; }
static inline bool attn_lsv_forced() { return g_attn_lsv >= 0 && !(g_attn_lsv & 8); }
static inline unsigned attn_cap_lds() { return (g_attn_lsv >= 0 && (g_attn_lsv & 16)) ? 56u * 1024u : 0u; }

template <int HD, int NV, int QG, int PF>
static int launch_attn_packed(const AttnP& p, int B, void* workspace, hipStream_t stream) {
    using P = PackCfg<HD, NV>;
    const int n = p.n, npad = ceil_div(n, 32) * 32;
    const size_t bh = (size_t)B * p.heads;
    char* base = (char*)workspace + pk_align256(attn_hdr_bytes(B, n, p.heads));
    bf16_t* Kp = (bf16_t*)base;
    bf16_t* Vtp = (bf16_t*)(base + pk_align256(bh * npad * HD * 2));
    if ((size_t)npad * HD * 2 >= 0x7fffffffULL || (size_t)(npad / 16) * P::ROWS * 32 >= 0x7fffffffULL)
        return uwir_set_error(UWIR_EINVAL, "flash_attn: sequence too long for the packed stream");
    hipLaunchKernelGGL((attn_pack_kernel<HD, NV>), dim3(ceil_div(npad, PK_KEYS * PK_TILES), p.heads, B), dim3(256), 0, stream,
                       (const bf16_t*)p.k, p.ldk, (const bf16_t*)p.v[0], p.ldv[0], (const bf16_t*)p.v[1], p.ldv[1], n, npad,
                       p.heads, Kp, Vtp, (unsigned*)workspace, (unsigned*)p.kext);
    UWIR_LAUNCH_CHECK("flash_attn(pack)");
    const int nqb = ceil_div(n, 128 * QG);
    const long long nb = (long long)nqb * p.heads * B;
    if (nb > 0x7fffffffLL) return uwir_set_error(UWIR_EINVAL, "flash_attn: grid too large");
    probe_mark(0, stream);
    // measured (tools/bench_attn.py lsv): hd16 2V 2.18 / 2.02 / 1.83 ms for forms 0 / 1 / 2, hd32 1V 0.53 / 0.51 / 0.47,
    // hd32 2V 0.258 / 0.223 / 0.234
    const int lsv = P::SUMROW ? 0 : (!attn_lsv_forced() ? ((HD == 32 && NV == 2) ? 1 : 2) : (g_attn_lsv & 3));
    const unsigned cap = attn_cap_lds();
    const dim3 grid((unsigned)nb);
    if (lsv == 2)
        hipLaunchKernelGGL((attn_stream_kernel<HD, NV, QG, PF, P::SUMROW ? 0 : 2>), grid, dim3(256), cap, stream, p, Kp, Vtp, npad, nqb, (int)nb);
    else if (lsv == 1)
        hipLaunchKernelGGL((attn_stream_kernel<HD, NV, QG, PF, P::SUMROW ? 0 : 1>), grid, dim3(256), cap, stream, p, Kp, Vtp, npad, nqb, (int)nb);
    else
        hipLaunchKernelGGL((attn_stream_kernel<HD, NV, QG, PF, 0>), grid, dim3(256), cap, stream, p, Kp, Vtp, npad, nqb, (int)nb);
    probe_mark(1, stream);
    // blocks whose score bound is too large for a fixed shift marked themselves: the online-max loop redoes them
    // (128-query blocks: only the marked groups are redone; 256-query blocks redo both halves of a pair when either is
    // marked -- uwir_attn_variant bit 6 keeps them for A/B timing)
    if (g_attn_lsv >= 0 && (g_attn_lsv & 64))
        hipLaunchKernelGGL((flash_attn_kernel<bf16_t, HD, NV, 2, 2>), dim3(ceil_div(n, 256), p.heads, B), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((flash_attn_kernel<bf16_t, HD, NV, 1, 2>), dim3(ceil_div(n, 128), p.heads, B), dim3(256), 0, stream, p);
    UWIR_LAUNCH_CHECK("flash_attn(stream)");
    return UWIR_OK;
}

template <typename T, int HD>
static int launch_attn_nv(const AttnP& p, int nv, int B, hipStream_t stream) {
    // big bf16 problems with hd <= 16: two query groups per wave (256 queries per block) and the split launch
    if constexpr (sizeof(T) == 2 && HD <= 16) {
        // (the two-V variant measured no faster this way -- 202 VGPRs, 2 waves/SIMD -- and stays on the single launch)
        if (nv == 1 && p.flags && (long long)ceil_div(p.n, 256) * p.heads * B >= 1024) {
            dim3 grid(ceil_div(p.n, 256), p.heads, B);
            probe_mark(0, stream);
            hipLaunchKernelGGL((flash_attn_kernel<T, HD, 1, 2, 1>), grid, dim3(256), 0, stream, p);
            probe_mark(1, stream);
            hipLaunchKernelGGL((flash_attn_kernel<T, HD, 1, 2, 2>), grid, dim3(256), 0, stream, p);
            UWIR_LAUNCH_CHECK("flash_attn(split)");
            return UWIR_OK;
        }
    }
    dim3 grid(ceil_div(p.n, 128), p.heads, B);
    probe_mark(0, stream);
    if (nv == 1) hipLaunchKernelGGL((flash_attn_kernel<T, HD, 1, 1, 0>), grid, dim3(256), 0, stream, p);
    else if constexpr (HD <= 64) hipLaunchKernelGGL((flash_attn_kernel<T, HD, 2, 1, 0>), grid, dim3(256), 0, stream, p);
    else return uwir_set_error(UWIR_EUNSUPPORTED, "flash_attn: two value streams at head_dim > 64 run as two calls");
    probe_mark(1, stream);
    UWIR_LAUNCH_CHECK("flash_attn");
    return UWIR_OK;
}

template <typename T>
static int launch_attn(const AttnP& p, int nv, int B, hipStream_t stream) {
    if (p.hd <= 16) return launch_attn_nv<T, 16>(p, nv, B, stream);
    if (p.hd <= 32) return launch_attn_nv<T, 32>(p, nv, B, stream);
    if (p.hd <= 64) return launch_attn_nv<T, 64>(p, nv, B, stream);
    return launch_attn_nv<T, 128>(p, nv, B, stream);   // one value stream only (the V^T tiles of two would not fit LDS in fp32)
}

extern "C" size_t uwir_flash_attn_workspace_bytes(int B, int n, int heads) { return attn_hdr_bytes(B, n, heads); }
extern "C" size_t uwir_flash_attn_packed_workspace_bytes(int B, int n, int heads, int hd, int nv) {
    if (B <= 0 || n <= 0 || heads <= 0 || !attn_packed_shape(B, n, heads, hd, nv)) return attn_hdr_bytes(B, n, heads);
    return attn_packed_bytes(B, n, heads, hd, nv);
}

extern "C" int uwir_flash_attn(const void* q, int ldq, const void* k, int ldk, const void* v1, int ldv1,
                               const void* v2, int ldv2, int dtype, int B, int n, int heads, int hd, float scale,
                               void* o1, int ldo1, void* o2, int ldo2, int out_dtype, void* workspace,
                               size_t workspace_bytes, uwir_stream_t stream) {
    UWIR_CHECK(q && k && v1 && o1, "flash_attn: null pointer");
    UWIR_CHECK((v2 == nullptr) == (o2 == nullptr), "flash_attn: v2 and o2 must come together");
    UWIR_CHECK(B > 0 && B <= 65535 && heads > 0 && heads <= 65535 && n > 0, "flash_attn: bad shape");
    UWIR_CHECK(hd > 0 && hd <= 128 && hd % 4 == 0, "flash_attn: head_dim must be a multiple of 4, <= 128 (got %d)", hd);
    if (hd > 64 && v2) {
        // wide heads (a student wider / deeper than the reference's default): the shared-softmax pair runs as two
        // single-V attentions over the same q, k (same results; QK^T and the softmax are computed twice)
        int rc = uwir_flash_attn(q, ldq, k, ldk, v1, ldv1, nullptr, 0, dtype, B, n, heads, hd, scale, o1, ldo1, nullptr, 0,
                                 out_dtype, workspace, workspace_bytes, stream);
        if (rc != UWIR_OK) return rc;
        return uwir_flash_attn(q, ldq, k, ldk, v2, ldv2, nullptr, 0, dtype, B, n, heads, hd, scale, o2, ldo2, nullptr, 0,
                               out_dtype, workspace, workspace_bytes, stream);
    }
    UWIR_CHECK(dtype == UWIR_F32 || dtype == UWIR_BF16, "flash_attn: bad dtype");
    UWIR_CHECK(out_dtype == UWIR_F32 || out_dtype == UWIR_BF16, "flash_attn: bad out dtype");
    int w = heads * hd;
    UWIR_CHECK(ldq >= w && ldk >= w && ldv1 >= w && ldo1 >= w && (!v2 || (ldv2 >= w && ldo2 >= w)),
               "flash_attn: leading dimension smaller than heads*hd");
    UWIR_CHECK(ldq % 4 == 0 && ldk % 4 == 0 && ldv1 % 4 == 0 && ldo1 % 4 == 0 && ldv2 % 4 == 0 && ldo2 % 4 == 0,
               "flash_attn: leading dimensions must be multiples of 4");
    const uintptr_t am = dtype == UWIR_BF16 ? 7 : 15;
    UWIR_CHECK(((uintptr_t)q & am) == 0 && ((uintptr_t)k & am) == 0 && ((uintptr_t)v1 & am) == 0 &&
                   ((uintptr_t)v2 & am) == 0,
               "flash_attn: misaligned input pointer");
    AttnP p;
    p.q = q; p.k = k; p.v[0] = v1; p.v[1] = v2;
    p.o[0] = o1; p.o[1] = o2;
    p.ldq = ldq; p.ldk = ldk; p.ldv[0] = ldv1; p.ldv[1] = ldv2; p.ldo[0] = ldo1; p.ldo[1] = ldo2;
    p.n = n; p.heads = heads; p.hd = hd; p.out_dtype = out_dtype;
    UWIR_CHECK(scale >= 0.f, "flash_attn: negative scale");
    p.c = scale == 0.f ? 1.f : scale * 1.4426950408889634f;
    int nv = v2 ? 2 : 1;
    p.kmax2 = nullptr;
    p.kext = nullptr;
    p.norm_bound_only = (g_attn_lsv >= 0 && (g_attn_lsv & 32)) ? 1 : 0;
    p.flags = nullptr;
    // (short sequences -- ViT towers, 50 tokens -- gain nothing from the bound pre-pass: it would triple their launches)
    if (dtype == UWIR_BF16 && workspace && n >= 256) {
        const size_t hdr = attn_hdr_bytes(B, n, heads);
        UWIR_CHECK(workspace_bytes >= hdr && ((uintptr_t)workspace & 3) == 0, "flash_attn: workspace too small / misaligned");
        hipStream_t s = (hipStream_t)stream;
        if (hipMemsetAsync(workspace, 0, hdr, s) != hipSuccess)
            return uwir_set_error(UWIR_ELAUNCH, "flash_attn: hipMemsetAsync failed");
        p.kmax2 = (const float*)workspace;
        p.flags = (int*)workspace + (size_t)B * heads;
        p.kext = (const unsigned*)workspace + (size_t)B * heads + (size_t)ceil_div(n, 128) * heads * B;
        if (attn_packed_shape(B, n, heads, hd, nv) && workspace_bytes >= attn_packed_bytes(B, n, heads, hd, nv) &&
            ((uintptr_t)workspace & 255) == 0) {
            // the pack pre-pass also produces max|k|^2
            // (QG, PF) per shape measured on MI355X (tools/bench_attn.py): hd16 1V 4.84 -> 4.72 ms with 4 query groups,
            // hd16 2V 2.29 -> 2.24 and hd32 1V 0.54 -> 0.53 with one block of prefetch (with the selector-MFMA denominators
            // hd16 2V runs 1.88 at (2, 1) and 1.83 at (4, 2)), hd32 2V 0.31 -> 0.26 with (2, 1)
            // (end of round 2, A/B/A/B with the pre-pass: hd16 2V at (4, 1) 1.80 / 1.80 ms against 1.87 / 1.83 at (4, 2); hd16 1V at
            // (4, 1) 5.00 / 5.01 against 4.80 / 4.67 at (4, 2).  uwir_attn_variant bit 7 keeps (4, 2) for the two-V kernel.)
            if (hd == 16) return nv == 1 ? launch_attn_packed<16, 1, 4, 2>(p, B, workspace, s)
                                         : ((g_attn_lsv >= 0 && (g_attn_lsv & 4)) ? launch_attn_packed<16, 2, 2, 1>(p, B, workspace, s)
                                            : (g_attn_lsv >= 0 && (g_attn_lsv & 128)) ? launch_attn_packed<16, 2, 4, 2>(p, B, workspace, s)
                                                            : launch_attn_packed<16, 2, 4, 1>(p, B, workspace, s));
            // (hd32 1V, A/B/A/B with the pre-pass: (4, 2) 0.474 - 0.493 ms against 0.492 - 0.526 at (2, 1) and 0.55 - 0.59 at (4, 1) for
            // B16 n4096, 1.63 against 1.70 for B4 n16384; hd32 2V stays at (2, 1): 0.233 against 0.238 at (4, 1).  Bit 8 keeps (2, 1).)
            if (nv == 1 && !(g_attn_lsv >= 0 && (g_attn_lsv & 256))) return launch_attn_packed<32, 1, 4, 2>(p, B, workspace, s);
            return nv == 1 ? launch_attn_packed<32, 1, 2, 1>(p, B, workspace, s)
                           : launch_attn_packed<32, 2, 2, 1>(p, B, workspace, s);
        }
        hipLaunchKernelGGL(attn_kmax_kernel, dim3(ceil_div(n, KMAX_ROWS), heads, B), dim3(256), 0, s, (const bf16_t*)k, ldk,
                           n, heads, hd, (unsigned*)workspace);
        UWIR_LAUNCH_CHECK("flash_attn(kmax)");
    }
    if (dtype == UWIR_BF16) return launch_attn<bf16_t>(p, nv, B, (hipStream_t)stream);
    return launch_attn<float>(p, nv, B, (hipStream_t)stream);
}

// 2-D FFT amplitude/phase modulation (S/dwt_fft.py:61-72, S/sffb.py:49-54), NHWC, fp32.
//
// Each pass transforms "lines" of L points for a slab of CS consecutive channels entirely in LDS
// (re/im planes, channel fastest => conflict-free, float4 global traffic):
//   fwd rows : real x            -> complex workspace          (along W)
//   fwd cols : complex workspace -> amp = |F|, phase = atan2   (along H)  [self-conjugate bins forced real]
//   inv cols : amp,phase (polar) -> complex workspace          (along H)
//   inv rows : complex workspace -> addend + Re(.)             (along W)
// Powers of two: in-place radix-2 DIT (bit-reversed load, log2 L barrier-separated stages);
// other lengths: direct DFT against a full twiddle table (O(L^2), correctness path).
// norm="ortho": each pass scales by 1/sqrt(L).
#include "common.h"

enum { FFT_FWD_ROWS = 0, FFT_FWD_COLS = 1, FFT_INV_COLS = 2, FFT_INV_ROWS = 3 };

struct FftP {
    const float* in0;  // x | wre | amp   | wre
    const float* in1;  // - | wim | phase | wim
    float* out0;       // wre | amp   | wre | out
    float* out1;       // wim | phase | wim | -
    const float* addend;
    int ld_in, ld_out, ld_add;
    int H, W, C, L, CS, logL;
    float norm;
    int half;          // fft_lines4_kernel only.  Forward (real input): columns 0 .. W/2 of the row pass are kept, the column
                       // pass transforms those and writes each bin's mirror as its conjugate.  Inverse (real part
                       // wanted): the column pass transforms the Hermitian part of the spectrum on columns 0 .. W/2,
                       // the row pass reads the other columns as conjugates
};

__device__ __forceinline__ int bitrev(int v, int bits) {
    return bits == 0 ? 0 : (int)(__brev((unsigned)v) >> (32 - bits));
}

template <int MODE>
__global__ void __launch_bounds__(256) fft_lines_kernel(FftP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = p.L, CS = p.CS, tid = threadIdx.x;
    constexpr bool ROWS = (MODE == FFT_FWD_ROWS || MODE == FFT_INV_ROWS);
    constexpr bool INV = (MODE == FFT_INV_COLS || MODE == FFT_INV_ROWS);
    const bool pow2 = p.logL >= 0;
    float* re = smem;
    float* im = re + L * CS;
    float* re2 = pow2 ? re : im + L * CS;  // second buffer only for the direct DFT
    float* im2 = pow2 ? im : re2 + L * CS;
    float* twr = (pow2 ? im + L * CS : im2 + L * CS);
    float* twi = twr + L;

    // line -> (b, fixed coordinate); point j -> pixel
    const int line = blockIdx.x;
    const int c0 = blockIdx.y * CS;
    const int ncs = min(CS, p.C - c0);  // valid channels in slab (multiple of 4)
    int b, fixed;
    if (ROWS) { b = line / p.H; fixed = line % p.H; } else { b = line / p.W; fixed = line % p.W; }
    auto pixel = [&](int j) -> size_t {
        return ROWS ? ((size_t)b * p.H + fixed) * p.W + j : ((size_t)b * p.H + j) * p.W + fixed;
    };

    // twiddles: tw[k] = exp(-+ 2 pi i k / L), k in [0, L)
    for (int k = tid; k < L; k += 256) {
        float s, c;
        sincospif(2.0f * (float)k / (float)L, &s, &c);
        twr[k] = c;
        twi[k] = INV ? s : -s;
    }
    // load
    const int q4 = CS / 4;
    for (int i = tid; i < L * q4; i += 256) {
        int j = i / q4, q = i % q4;
        float4 vr = make_float4(0, 0, 0, 0), vi = make_float4(0, 0, 0, 0);
        if (4 * q < ncs) {
            size_t px = pixel(j);
            if (MODE == FFT_FWD_ROWS) {
                vr = load4(p.in0 + px * p.ld_in + c0 + 4 * q);
            } else if (MODE == FFT_INV_COLS) {
                float4 a = load4(p.in0 + px * p.ld_in + c0 + 4 * q), ph = load4(p.in1 + px * p.ld_in + c0 + 4 * q);
                float s, c;
                sincosf(ph.x, &s, &c); vr.x = a.x * c; vi.x = a.x * s;
                sincosf(ph.y, &s, &c); vr.y = a.y * c; vi.y = a.y * s;
                sincosf(ph.z, &s, &c); vr.z = a.z * c; vi.z = a.z * s;
                sincosf(ph.w, &s, &c); vr.w = a.w * c; vi.w = a.w * s;
            } else {
                vr = load4(p.in0 + px * p.ld_in + c0 + 4 * q);
                vi = load4(p.in1 + px * p.ld_in + c0 + 4 * q);
            }
        }
        int dst = pow2 ? bitrev(j, p.logL) : j;
        *reinterpret_cast<float4*>(&re[dst * CS + 4 * q]) = vr;
        *reinterpret_cast<float4*>(&im[dst * CS + 4 * q]) = vi;
    }
    __syncthreads();

    if (pow2) {
        for (int s = 0; s < p.logL; ++s) {
            const int half = 1 << s;
            const int tstep = L >> (s + 1);
            for (int i = tid; i < (L / 2) * CS; i += 256) {
                int ch = i % CS, t = i / CS;
                int grp = t >> s, pos = t & (half - 1);
                int i0 = ((grp << (s + 1)) + pos) * CS + ch, i1 = i0 + half * CS;
                float wr = twr[pos * tstep], wi = twi[pos * tstep];
                float br = re[i1], bi = im[i1];
                float tr = wr * br - wi * bi, ti = wr * bi + wi * br;
                float ar = re[i0], ai = im[i0];
                re[i0] = ar + tr; im[i0] = ai + ti;
                re[i1] = ar - tr; im[i1] = ai - ti;
            }
            __syncthreads();
        }
    } else {
        for (int i = tid; i < L * CS; i += 256) {
            int ch = i % CS, k = i / CS;
            float sr = 0.f, si = 0.f;
            int idx = 0;  // (j*k) mod L, updated incrementally
            for (int j = 0; j < L; ++j) {
                float wr = twr[idx], wi = twi[idx];
                float xr = re[j * CS + ch], xi = im[j * CS + ch];
                sr += wr * xr - wi * xi;
                si += wr * xi + wi * xr;
                idx += k;
                if (idx >= L) idx -= L;
            }
            re2[i] = sr;
            im2[i] = si;
        }
        __syncthreads();
    }

    // store
    const float nrm = p.norm;
    const bool fx_self = ROWS ? false : (fixed == 0 || (p.W % 2 == 0 && fixed == p.W / 2));
    for (int i = tid; i < L * q4; i += 256) {
        int j = i / q4, q = i % q4;
        if (4 * q >= ncs) continue;
        float4 vr = *reinterpret_cast<const float4*>(&re2[j * CS + 4 * q]);
        float4 vi = *reinterpret_cast<const float4*>(&im2[j * CS + 4 * q]);
        vr.x *= nrm; vr.y *= nrm; vr.z *= nrm; vr.w *= nrm;
        vi.x *= nrm; vi.y *= nrm; vi.z *= nrm; vi.w *= nrm;
        size_t px = pixel(j);
        if (MODE == FFT_FWD_COLS) {
            // bins with ky in {0,H/2} and kx in {0,W/2} of a real signal are exactly real: pin Im = +0
            if (fx_self && (j == 0 || (p.H % 2 == 0 && j == p.H / 2))) vi = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 a, ph;
            a.x = hypotf(vr.x, vi.x); ph.x = atan2f(vi.x, vr.x);
            a.y = hypotf(vr.y, vi.y); ph.y = atan2f(vi.y, vr.y);
            a.z = hypotf(vr.z, vi.z); ph.z = atan2f(vi.z, vr.z);
            a.w = hypotf(vr.w, vi.w); ph.w = atan2f(vi.w, vr.w);
            store4(p.out0 + px * p.ld_out + c0 + 4 * q, a);
            store4(p.out1 + px * p.ld_out + c0 + 4 * q, ph);
        } else if (MODE == FFT_INV_ROWS) {
            float4 ad = load4(p.addend + px * p.ld_add + c0 + 4 * q);
            vr.x += ad.x; vr.y += ad.y; vr.z += ad.z; vr.w += ad.w;
            store4(p.out0 + px * p.ld_out + c0 + 4 * q, vr);
        } else {
            store4(p.out0 + px * p.ld_out + c0 + 4 * q, vr);
            store4(p.out1 + px * p.ld_out + c0 + 4 * q, vi);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Power-of-two lines, second form.  The radix-2 kernel above moves every point through LDS seven times (L = 128) as
// 4-byte accesses with a barrier per stage, and is bound by that (LDS 64-128 B/clk for b32 traffic), not by HBM.  Here:
// in-place decimation-in-frequency with radix-4 stages (one radix-2 stage when log2 L is odd: 4 stages for 128, 3 for
// 64), every LDS access a 16-byte channel quad (lane = (butterfly, quad): 16 consecutive lanes cover one 256-byte bank
// row), natural-order load, digit-reversed read at the store.  A block takes as many lines as fill 256 threads with
// radix-4 butterflies (2 at L = 128, 4 at 64, 8 at 32).  The 1/sqrt(L) scaling, polar conversions and the addend are
// fused into the load / store as before.
// ------------------------------------------------------------------------------------------
struct Cx4 { float4 r, i; };
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4neg(float4 a) { return make_float4(-a.x, -a.y, -a.z, -a.w); }
__device__ __forceinline__ Cx4 cadd(Cx4 a, Cx4 b) { return {f4add(a.r, b.r), f4add(a.i, b.i)}; }
__device__ __forceinline__ Cx4 csub(Cx4 a, Cx4 b) { return {f4sub(a.r, b.r), f4sub(a.i, b.i)}; }
__device__ __forceinline__ Cx4 cmulw(Cx4 a, float wr, float wi) {   // a * (wr + i wi)
    Cx4 o;
    o.r = make_float4(a.r.x * wr - a.i.x * wi, a.r.y * wr - a.i.y * wi, a.r.z * wr - a.i.z * wi, a.r.w * wr - a.i.w * wi);
    o.i = make_float4(a.r.x * wi + a.i.x * wr, a.r.y * wi + a.i.y * wr, a.r.z * wi + a.i.z * wr, a.r.w * wi + a.i.w * wr);
    return o;
}

#define FFT4_CS 16   // channels per block slab: 64 bytes per pixel, 4 quads

template <int MODE>
__global__ void __launch_bounds__(256) fft_lines4_kernel(FftP p, int nlines, int nlb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CS = FFT4_CS, Q4 = CS / 4;
    constexpr bool ROWS = (MODE == FFT_FWD_ROWS || MODE == FFT_INV_ROWS);
    constexpr bool INV = (MODE == FFT_INV_COLS || MODE == FFT_INV_ROWS);
    const int L = p.L, tid = threadIdx.x;
    float* re = smem;                 // [nlb][L][CS]
    float* im = re + nlb * L * CS;
    float* twr = im + nlb * L * CS;   // [L]
    float* twi = twr + L;
    const int c0 = blockIdx.y * CS;
    const int ncs = min(CS, p.C - c0);
    const int line0 = blockIdx.x * nlb;

    for (int k = tid; k < L; k += 256) {
        float s, c;
        sincospif(2.0f * (float)k / (float)L, &s, &c);
        twr[k] = c;
        twi[k] = INV ? s : -s;
    }
    // column lines of a half spectrum: W/2 + 1 per image (FWD_COLS with p.half), else W
    const int WL = ((MODE == FFT_FWD_COLS || MODE == FFT_INV_COLS) && p.half) ? p.W / 2 + 1 : p.W;
    auto pixel = [&](int line, int j) -> size_t {
        const int b = ROWS ? line / p.H : line / WL, fixed = ROWS ? line % p.H : line % WL;
        return ROWS ? ((size_t)b * p.H + fixed) * p.W + j : ((size_t)b * p.H + j) * p.W + fixed;
    };
    // ---- load (natural order).  nlb * L == 256 (host), so a thread has exactly 4 (pixel, quad) items: all of their
    // global loads are issued before the first LDS store (a rolled load -> store loop exposes one HBM round trip per item)
    // Inverse transform whose REAL part is wanted (p.half): Re ifft2(S) = ifft2(Sh), Sh[k] = (S[k] + conj S[-k]) / 2, and
    // Sh is Hermitian.  The column pass forms Sh for columns 0 .. W/2 only (mr / mi: the mirrored bin ((H - ky) % H,
    // (W - kx) % W) of the input), the row pass reads column j > W/2 of its line as the conjugate of column W - j.
    float4 lr[4], li[4], mr[4], mi[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = tid + 256 * it;
        const int q = i % Q4, j = (i / Q4) % L, ll = i / (Q4 * L), line = line0 + ll;
        lr[it] = make_float4(0, 0, 0, 0);
        li[it] = make_float4(0, 0, 0, 0);
        mr[it] = make_float4(0, 0, 0, 0);
        mi[it] = make_float4(0, 0, 0, 0);
        if (4 * q < ncs && line < nlines) {
            const bool conj_src = MODE == FFT_INV_ROWS && p.half && j > p.W / 2;
            const size_t px = pixel(line, conj_src ? p.W - j : j);
            lr[it] = load4(p.in0 + px * p.ld_in + c0 + 4 * q);
            if (MODE != FFT_FWD_ROWS) li[it] = load4(p.in1 + px * p.ld_in + c0 + 4 * q);
            if (conj_src) li[it] = f4neg(li[it]);
            if (MODE == FFT_INV_COLS && p.half) {
                const int b = line / WL, fixed = line % WL;
                const size_t pm = ((size_t)b * p.H + (j ? p.H - j : 0)) * p.W + (fixed ? p.W - fixed : 0);
                mr[it] = load4(p.in0 + pm * p.ld_in + c0 + 4 * q);
                mi[it] = load4(p.in1 + pm * p.ld_in + c0 + 4 * q);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = tid + 256 * it;
        const int q = i % Q4, j = (i / Q4) % L, ll = i / (Q4 * L);
        float4 vr = lr[it], vi = li[it];
        if (MODE == FFT_INV_COLS) {   // (amp, phase) -> complex
            const float4 a = lr[it], ph = li[it];
            float s, c;
            sincosf(ph.x, &s, &c); vr.x = a.x * c; vi.x = a.x * s;
            sincosf(ph.y, &s, &c); vr.y = a.y * c; vi.y = a.y * s;
            sincosf(ph.z, &s, &c); vr.z = a.z * c; vi.z = a.z * s;
            sincosf(ph.w, &s, &c); vr.w = a.w * c; vi.w = a.w * s;
            if (p.half) {
                const float4 a2 = mr[it], p2 = mi[it];
                sincosf(p2.x, &s, &c); vr.x = 0.5f * (vr.x + a2.x * c); vi.x = 0.5f * (vi.x - a2.x * s);
                sincosf(p2.y, &s, &c); vr.y = 0.5f * (vr.y + a2.y * c); vi.y = 0.5f * (vi.y - a2.y * s);
                sincosf(p2.z, &s, &c); vr.z = 0.5f * (vr.z + a2.z * c); vi.z = 0.5f * (vi.z - a2.z * s);
                sincosf(p2.w, &s, &c); vr.w = 0.5f * (vr.w + a2.w * c); vi.w = 0.5f * (vi.w - a2.w * s);
            }
        }
        *reinterpret_cast<float4*>(&re[(ll * L + j) * CS + 4 * q]) = vr;
        *reinterpret_cast<float4*>(&im[(ll * L + j) * CS + 4 * q]) = vi;
    }
    __syncthreads();

    // ---- in-place DIF stages: sub-transform size ns, radix r; butterfly k of group g touches g*ns + k + i*(ns/r)
    auto ldc = [&](int idx) { Cx4 v; v.r = *reinterpret_cast<const float4*>(&re[idx]); v.i = *reinterpret_cast<const float4*>(&im[idx]); return v; };
    auto stc = [&](int idx, Cx4 v) { *reinterpret_cast<float4*>(&re[idx]) = v.r; *reinterpret_cast<float4*>(&im[idx]) = v.i; };
    int ns = L;
    while (ns >= 4) {
        const int qs = ns >> 2, tstep = L / ns;
        for (int i = tid; i < nlb * (L >> 2) * Q4; i += 256) {
            const int q = i % Q4, t = (i / Q4) % (L >> 2), ll = i / (Q4 * (L >> 2));
            const int g = t / qs, k = t - g * qs;
            const int base = ((ll * L + g * ns + k) * CS) + 4 * q, st = qs * CS;
            const Cx4 a0 = ldc(base), a1 = ldc(base + st), a2 = ldc(base + 2 * st), a3 = ldc(base + 3 * st);
            const Cx4 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), d = csub(a1, a3);
            // forward: (-i) d = (d.i, -d.r); inverse: (+i) d = (-d.i, d.r)
            const Cx4 t3 = INV ? Cx4{f4neg(d.i), d.r} : Cx4{d.i, f4neg(d.r)};
            const Cx4 y0 = cadd(t0, t2), y2 = csub(t0, t2), y1 = cadd(t1, t3), y3 = csub(t1, t3);
            const int w1 = k * tstep, w2 = 2 * w1, w3 = 3 * w1;   // < L: k < ns/4
            stc(base, y0);
            stc(base + st, cmulw(y1, twr[w1], twi[w1]));
            stc(base + 2 * st, cmulw(y2, twr[w2], twi[w2]));
            stc(base + 3 * st, cmulw(y3, twr[w3], twi[w3]));
        }
        __syncthreads();
        ns = qs;
    }
    if (ns == 2) {   // last stage of an odd log2 L: k = 0 only, twiddle 1
        for (int i = tid; i < nlb * (L >> 1) * Q4; i += 256) {
            const int q = i % Q4, t = (i / Q4) % (L >> 1), ll = i / (Q4 * (L >> 1));
            const int base = ((ll * L + 2 * t) * CS) + 4 * q;
            const Cx4 a0 = ldc(base), a1 = ldc(base + CS);
            stc(base, cadd(a0, a1));
            stc(base + CS, csub(a0, a1));
        }
        __syncthreads();
    }

    // ---- store: frequency (or sample) j sits at the digit-reversed position
    const float nrm = p.norm;
    float4 ad[4];
    if (MODE == FFT_INV_ROWS) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = tid + 256 * it;
            const int q = i % Q4, j = (i / Q4) % L, ll = i / (Q4 * L), line = line0 + ll;
            ad[it] = make_float4(0, 0, 0, 0);
            if (4 * q < ncs && line < nlines) ad[it] = load4(p.addend + pixel(line, j) * p.ld_add + c0 + 4 * q);
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = tid + 256 * it;
        const int q = i % Q4, j = (i / Q4) % L, ll = i / (Q4 * L), line = line0 + ll;
        if (4 * q >= ncs || line >= nlines) continue;
        if (MODE == FFT_FWD_ROWS && p.half && j > p.W / 2) continue;   // X[W - k] = conj X[k]: the column pass restores them
        int pos = 0, kk = j, sz = L;
        while (sz >= 4) { pos += (kk & 3) * (sz >> 2); kk >>= 2; sz >>= 2; }
        if (sz == 2) pos += kk & 1;
        float4 vr = *reinterpret_cast<const float4*>(&re[(ll * L + pos) * CS + 4 * q]);
        float4 vi = *reinterpret_cast<const float4*>(&im[(ll * L + pos) * CS + 4 * q]);
        vr.x *= nrm; vr.y *= nrm; vr.z *= nrm; vr.w *= nrm;
        vi.x *= nrm; vi.y *= nrm; vi.z *= nrm; vi.w *= nrm;
        const size_t px = pixel(line, j);
        if (MODE == FFT_FWD_COLS) {
            // bins with ky in {0,H/2} and kx in {0,W/2} of a real signal are exactly real: pin Im = +0
            const int fixed = line % WL;
            const bool fx_self = fixed == 0 || (p.W % 2 == 0 && fixed == p.W / 2);
            if (fx_self && (j == 0 || (p.H % 2 == 0 && j == p.H / 2))) vi = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 a, ph;
            a.x = hypotf(vr.x, vi.x); ph.x = atan2f(vi.x, vr.x);
            a.y = hypotf(vr.y, vi.y); ph.y = atan2f(vi.y, vr.y);
            a.z = hypotf(vr.z, vi.z); ph.z = atan2f(vi.z, vr.z);
            a.w = hypotf(vr.w, vi.w); ph.w = atan2f(vi.w, vr.w);
            store4(p.out0 + px * p.ld_out + c0 + 4 * q, a);
            store4(p.out1 + px * p.ld_out + c0 + 4 * q, ph);
            if (p.half && !fx_self) {
                // F[(H - ky) % H, W - kx] = conj F[ky, kx] for a real map: same amplitude, phase negated (atan2 is odd in
                // its first argument, signed zeros included)
                const int b = line / WL;
                const size_t pm = ((size_t)b * p.H + (j ? p.H - j : 0)) * p.W + (p.W - fixed);
                store4(p.out0 + pm * p.ld_out + c0 + 4 * q, a);
                store4(p.out1 + pm * p.ld_out + c0 + 4 * q, make_float4(-ph.x, -ph.y, -ph.z, -ph.w));
            }
        } else if (MODE == FFT_INV_ROWS) {
            vr.x += ad[it].x; vr.y += ad[it].y; vr.z += ad[it].z; vr.w += ad[it].w;
            store4(p.out0 + px * p.ld_out + c0 + 4 * q, vr);
        } else {
            store4(p.out0 + px * p.ld_out + c0 + 4 * q, vr);
            store4(p.out1 + px * p.ld_out + c0 + 4 * q, vi);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Lines whose length is 2^a 3^b 5^c 7^d 11^e 13^f but not a power of two (the sub-band maps of 1080p / 720p / VGA-sized
// frames: 540, 270, 135, 480, 240, 120, 360, 180, 90 ...; of 224-pixel frames: 112, 56, 28): the same in-place
// decimation-in-frequency scheme with a host-chosen radix schedule out of {4, 2, 3, 5, 7, 11, 13}, generic loops instead
// of the fixed one-butterfly-per-thread mapping.  Before this kernel such lines took the O(L^2) direct DFT (120 points:
// 15x the work).  Lengths with a prime factor above 13 still do.
// ------------------------------------------------------------------------------------------
#define FFT_MAXST 12
struct FftSched { int n; int radix[FFT_MAXST]; };

template <int MODE>
__global__ void __launch_bounds__(256) fft_linesmr_kernel(FftP p, FftSched sc, int nlines, int nlb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool ROWS = (MODE == FFT_FWD_ROWS || MODE == FFT_INV_ROWS);
    constexpr bool INV = (MODE == FFT_INV_COLS || MODE == FFT_INV_ROWS);
    const int L = p.L, CS = p.CS, Q4 = CS / 4, tid = threadIdx.x;
    float* re = smem;                 // [nlb][L][CS]
    float* im = re + nlb * L * CS;
    float* twr = im + nlb * L * CS;   // [L]
    float* twi = twr + L;
    const int c0 = blockIdx.y * CS;
    const int ncs = min(CS, p.C - c0);
    const int line0 = blockIdx.x * nlb;

    for (int k = tid; k < L; k += 256) {
        float s, c;
        sincospif(2.0f * (float)k / (float)L, &s, &c);
        twr[k] = c;
        twi[k] = INV ? s : -s;
    }
    auto pixel = [&](int line, int j) -> size_t {
        const int b = ROWS ? line / p.H : line / p.W, fixed = ROWS ? line % p.H : line % p.W;
        return ROWS ? ((size_t)b * p.H + fixed) * p.W + j : ((size_t)b * p.H + j) * p.W + fixed;
    };
    for (int i = tid; i < nlb * L * Q4; i += 256) {
        const int q = i % Q4, j = (i / Q4) % L, ll = i / (Q4 * L), line = line0 + ll;
        float4 vr = make_float4(0, 0, 0, 0), vi = make_float4(0, 0, 0, 0);
        if (4 * q < ncs && line < nlines) {
            const size_t px = pixel(line, j);
            vr = load4(p.in0 + px * p.ld_in + c0 + 4 * q);
            if (MODE != FFT_FWD_ROWS) vi = load4(p.in1 + px * p.ld_in + c0 + 4 * q);
            if (MODE == FFT_INV_COLS) {
                const float4 a = vr, ph = vi;
                float s, c;
                sincosf(ph.x, &s, &c); vr.x = a.x * c; vi.x = a.x * s;
                sincosf(ph.y, &s, &c); vr.y = a.y * c; vi.y = a.y * s;
                sincosf(ph.z, &s, &c); vr.z = a.z * c; vi.z = a.z * s;
                sincosf(ph.w, &s, &c); vr.w = a.w * c; vi.w = a.w * s;
            }
        }
        *reinterpret_cast<float4*>(&re[(ll * L + j) * CS + 4 * q]) = vr;
        *reinterpret_cast<float4*>(&im[(ll * L + j) * CS + 4 * q]) = vi;
    }
    __syncthreads();

    auto ldc = [&](int idx) { Cx4 v; v.r = *reinterpret_cast<const float4*>(&re[idx]); v.i = *reinterpret_cast<const float4*>(&im[idx]); return v; };
    auto stc = [&](int idx, Cx4 v) { *reinterpret_cast<float4*>(&re[idx]) = v.r; *reinterpret_cast<float4*>(&im[idx]) = v.i; };
    // multiply by -i (forward) / +i (inverse): the quarter turn of this transform direction
    auto rot = [&](Cx4 d) { return INV ? Cx4{f4neg(d.i), d.r} : Cx4{d.i, f4neg(d.r)}; };
    auto scl = [&](Cx4 a, float f) { return Cx4{make_float4(a.r.x * f, a.r.y * f, a.r.z * f, a.r.w * f), make_float4(a.i.x * f, a.i.y * f, a.i.z * f, a.i.w * f)}; };
    // butterfly of a prime radix R (7, 11, 13) as a direct R-point DFT: y_i = (sum_j a_j w_R^(ij)) tw[i w1], with
    // w_R^m = tw[m L / R] out of the line's own table (R divides L).  R^2 complex products per butterfly; such a factor
    // occurs once or twice in a line (224-pixel frames: sub-bands of 112 = 4 * 4 * 7), the 2 / 3 / 4 / 5 stages carry the rest.
    auto prime_bfly = [&](auto r_tag, int base, int st, int w1) {
        constexpr int R = decltype(r_tag)::value;
        const int rs = L / R;
        Cx4 a[R];
#pragma unroll
        for (int j = 0; j < R; ++j) a[j] = ldc(base + j * st);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            Cx4 y = a[0];
#pragma unroll
            for (int j = 1; j < R; ++j) {
                const int m = (i * j) % R;
                y = m ? cadd(y, cmulw(a[j], twr[m * rs], twi[m * rs])) : cadd(y, a[j]);
            }
            stc(base + i * st, i ? cmulw(y, twr[i * w1], twi[i * w1]) : y);
        }
    };
    int ns = L;
    for (int stg = 0; stg < sc.n; ++stg) {
        const int r = sc.radix[stg], qs = ns / r, tstep = L / ns;
        for (int i = tid; i < nlb * (L / r) * Q4; i += 256) {
            const int q = i % Q4, t = (i / Q4) % (L / r), ll = i / (Q4 * (L / r));
            const int g = t / qs, k = t - g * qs;
            const int base = ((ll * L + g * ns + k) * CS) + 4 * q, st = qs * CS;
            const int w1 = k * tstep;   // output i is twiddled by tw[i * w1]: i * k < ns, so the index stays below L
            if (r == 4) {
                const Cx4 a0 = ldc(base), a1 = ldc(base + st), a2 = ldc(base + 2 * st), a3 = ldc(base + 3 * st);
                const Cx4 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = rot(csub(a1, a3));
                stc(base, cadd(t0, t2));
                stc(base + st, cmulw(cadd(t1, t3), twr[w1], twi[w1]));
                stc(base + 2 * st, cmulw(csub(t0, t2), twr[2 * w1], twi[2 * w1]));
                stc(base + 3 * st, cmulw(csub(t1, t3), twr[3 * w1], twi[3 * w1]));
            } else if (r == 2) {
                const Cx4 a0 = ldc(base), a1 = ldc(base + st);
                stc(base, cadd(a0, a1));
                stc(base + st, cmulw(csub(a0, a1), twr[w1], twi[w1]));
            } else if (r == 3) {
                const Cx4 a0 = ldc(base), a1 = ldc(base + st), a2 = ldc(base + 2 * st);
                const Cx4 t1 = cadd(a1, a2), t2 = csub(a0, scl(t1, 0.5f)), t3 = rot(scl(csub(a1, a2), 0.86602540378443864676f));
                stc(base, cadd(a0, t1));
                stc(base + st, cmulw(cadd(t2, t3), twr[w1], twi[w1]));
                stc(base + 2 * st, cmulw(csub(t2, t3), twr[2 * w1], twi[2 * w1]));
            } else if (r == 7) {
                prime_bfly(std::integral_constant<int, 7>{}, base, st, w1);
            } else if (r == 11) {
                prime_bfly(std::integral_constant<int, 11>{}, base, st, w1);
            } else if (r == 13) {
                prime_bfly(std::integral_constant<int, 13>{}, base, st, w1);
            } else {   // 5
                const Cx4 a0 = ldc(base), a1 = ldc(base + st), a2 = ldc(base + 2 * st), a3 = ldc(base + 3 * st), a4 = ldc(base + 4 * st);
                const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
                const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
                const Cx4 t1 = cadd(a1, a4), t2 = cadd(a2, a3), t3 = csub(a1, a4), t4 = csub(a2, a3);
                const Cx4 m1 = cadd(a0, cadd(scl(t1, c1), scl(t2, c2))), m2 = cadd(a0, cadd(scl(t1, c2), scl(t2, c1)));
                const Cx4 n1 = rot(cadd(scl(t3, s1), scl(t4, s2))), n2 = rot(csub(scl(t3, s2), scl(t4, s1)));
                stc(base, cadd(a0, cadd(t1, t2)));
                stc(base + st, cmulw(cadd(m1, n1), twr[w1], twi[w1]));
                stc(base + 2 * st, cmulw(cadd(m2, n2), twr[2 * w1], twi[2 * w1]));
                stc(base + 3 * st, cmulw(csub(m2, n2), twr[3 * w1], twi[3 * w1]));
                stc(base + 4 * st, cmulw(csub(m1, n1), twr[4 * w1], twi[4 * w1]));
            }
        }
        __syncthreads();
        ns = qs;
    }

    const float nrm = p.norm;
    for (int i = tid; i < nlb * L * Q4; i += 256) {
        const int q = i % Q4, j = (i / Q4) % L, ll = i / (Q4 * L), line = line0 + ll;
        if (4 * q >= ncs || line >= nlines) continue;
        int pos = 0, kk = j, sz = L;
        for (int stg = 0; stg < sc.n; ++stg) {
            const int r = sc.radix[stg];
            pos += (kk % r) * (sz / r);
            kk /= r;
            sz /= r;
        }
        float4 vr = *reinterpret_cast<const float4*>(&re[(ll * L + pos) * CS + 4 * q]);
        float4 vi = *reinterpret_cast<const float4*>(&im[(ll * L + pos) * CS + 4 * q]);
        vr.x *= nrm; vr.y *= nrm; vr.z *= nrm; vr.w *= nrm;
        vi.x *= nrm; vi.y *= nrm; vi.z *= nrm; vi.w *= nrm;
        const size_t px = pixel(line, j);
        if (MODE == FFT_FWD_COLS) {
            const int fixed = line % p.W;
            const bool fx_self = fixed == 0 || (p.W % 2 == 0 && fixed == p.W / 2);
            if (fx_self && (j == 0 || (p.H % 2 == 0 && j == p.H / 2))) vi = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 a, ph;
            a.x = hypotf(vr.x, vi.x); ph.x = atan2f(vi.x, vr.x);
            a.y = hypotf(vr.y, vi.y); ph.y = atan2f(vi.y, vr.y);
            a.z = hypotf(vr.z, vi.z); ph.z = atan2f(vi.z, vr.z);
            a.w = hypotf(vr.w, vi.w); ph.w = atan2f(vi.w, vr.w);
            store4(p.out0 + px * p.ld_out + c0 + 4 * q, a);
            store4(p.out1 + px * p.ld_out + c0 + 4 * q, ph);
        } else if (MODE == FFT_INV_ROWS) {
            float4 ad = load4(p.addend + px * p.ld_add + c0 + 4 * q);
            vr.x += ad.x; vr.y += ad.y; vr.z += ad.z; vr.w += ad.w;
            store4(p.out0 + px * p.ld_out + c0 + 4 * q, vr);
        } else {
            store4(p.out0 + px * p.ld_out + c0 + 4 * q, vr);
            store4(p.out1 + px * p.ld_out + c0 + 4 * q, vi);
        }
    }
}

// radix schedule for L = 2^a 3^b 5^c 7^d 11^e 13^f (4s first, then a 2, then the odd primes); n = 0 if L has another prime factor
static FftSched fft_schedule(int L) {
    FftSched s = {};
    int v = L;
    while (v % 4 == 0 && s.n < FFT_MAXST) { s.radix[s.n++] = 4; v /= 4; }
    while (v % 2 == 0 && s.n < FFT_MAXST) { s.radix[s.n++] = 2; v /= 2; }
    while (v % 3 == 0 && s.n < FFT_MAXST) { s.radix[s.n++] = 3; v /= 3; }
    while (v % 5 == 0 && s.n < FFT_MAXST) { s.radix[s.n++] = 5; v /= 5; }
    for (int pr : {7, 11, 13})
        while (v % pr == 0 && s.n < FFT_MAXST) { s.radix[s.n++] = pr; v /= pr; }
    if (v != 1) s.n = 0;
    return s;
}

static int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

static int pick_cs(int L, bool pow2) {
    // LDS floats: pow2: 2*L*CS + 2*L ; direct: 4*L*CS + 2*L.  Keep under ~64 KiB.
    int cs = 16;
    while (cs > 4 && (size_t)(pow2 ? 2 : 4) * L * cs * 4 + 8 * (size_t)L > 60 * 1024) cs >>= 1;
    return cs;
}

static int g_fft_radix4 = 1;   // tools/: uwir_conv_variant bit 2 keeps the radix-2 kernel
void uwir_fft_set_radix4(int on) { g_fft_radix4 = on; }

template <int MODE>
static int launch_fft(FftP p, int B, hipStream_t stream) {
    const bool rows = (MODE == FFT_FWD_ROWS || MODE == FFT_INV_ROWS);
    p.L = rows ? p.W : p.H;
    p.logL = ilog2_exact(p.L);
    const bool pow2 = p.logL >= 0;
    p.CS = pick_cs(p.L, pow2);
    p.norm = 1.0f / sqrtf((float)p.L);
    const int nlines_all = rows ? B * p.H : B * (p.half ? p.W / 2 + 1 : p.W);
    if (pow2 && p.L >= 4 && p.L <= 256 && g_fft_radix4) {
        const int nlb = 256 / p.L;   // nlb * L == 256: one radix-4 butterfly and 4 load / store items per thread
        const size_t lds4 = ((size_t)2 * nlb * p.L * FFT4_CS + 2 * (size_t)p.L) * sizeof(float);
        if (lds4 <= 64 * 1024) {
            p.CS = FFT4_CS;
            dim3 grid4(ceil_div(nlines_all, nlb), ceil_div(p.C, FFT4_CS));
            hipLaunchKernelGGL((fft_lines4_kernel<MODE>), grid4, dim3(256), lds4, stream, p, nlines_all, nlb);
            UWIR_LAUNCH_CHECK("fft_lines4");
            return UWIR_OK;
        }
    }
    size_t lds = ((size_t)(pow2 ? 2 : 4) * p.L * p.CS + 2 * (size_t)p.L) * sizeof(float);
    if (lds > 64 * 1024) return uwir_set_error(UWIR_EUNSUPPORTED, "fft: line length %d too large", p.L);
    int nlines = rows ? B * p.H : B * p.W;
    if (!pow2 && p.L >= 6 && g_fft_radix4) {
        const FftSched sc = fft_schedule(p.L);
        int cs = 16;
        while (cs > 4 && ((size_t)2 * p.L * cs + 2 * (size_t)p.L) * sizeof(float) > 60 * 1024) cs >>= 1;
        const size_t per_line = (size_t)2 * p.L * cs * sizeof(float);
        if (sc.n > 0 && per_line + 8 * (size_t)p.L <= 60 * 1024) {
            // lines per block: fill 256 threads with the widest stage's butterflies, within the LDS budget
            int nlb = max(1, 256 / max(1, (p.L / 2) * (cs / 4)));
            while (nlb > 1 && nlb * per_line + 8 * (size_t)p.L > 60 * 1024) --nlb;
            p.CS = cs;
            const size_t ldsm = nlb * per_line + 8 * (size_t)p.L;
            dim3 gridm(ceil_div(nlines_all, nlb), ceil_div(p.C, cs));
            hipLaunchKernelGGL((fft_linesmr_kernel<MODE>), gridm, dim3(256), ldsm, stream, p, sc, nlines_all, nlb);
            UWIR_LAUNCH_CHECK("fft_lines(mixed radix)");
            return UWIR_OK;
        }
    }
    dim3 grid(nlines, ceil_div(p.C, p.CS));
    hipLaunchKernelGGL((fft_lines_kernel<MODE>), grid, dim3(256), lds, stream, p);
    UWIR_LAUNCH_CHECK("fft_lines");
    return UWIR_OK;
}

extern "C" size_t uwir_fft_workspace_bytes(int B, int H, int W, int C) {
    return 2 * (size_t)B * H * W * C * sizeof(float);
}

static int fft_check(int B, int H, int W, int C, const void* ws, size_t ws_bytes) {
    UWIR_CHECK(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "fft: bad shape");
    UWIR_CHECK(H <= 1024 && W <= 1024, "fft: H, W must be <= 1024");
    UWIR_CHECK((long long)B * H < 0x7fffffffLL && (long long)B * W < 0x7fffffffLL, "fft: too many lines");
    UWIR_CHECK(ws && ws_bytes >= uwir_fft_workspace_bytes(B, H, W, C), "fft: workspace too small");
    return UWIR_OK;
}

extern "C" int uwir_fft2_amp_phase(const float* x, int ldx, int B, int H, int W, int C, float* amp, float* phase,
                                   int ld_out, void* workspace, size_t workspace_bytes, uwir_stream_t stream) {
    UWIR_CHECK(x && amp && phase, "fft2: null pointer");
    int rc = fft_check(B, H, W, C, workspace, workspace_bytes);
    if (rc) return rc;
    UWIR_CHECK(ldx >= C && ldx % 4 == 0 && ld_out >= C && ld_out % 4 == 0, "fft2: bad strides");
    float* wre = (float*)workspace;
    float* wim = wre + (size_t)B * H * W * C;
    FftP p = {};
    p.H = H; p.W = W; p.C = C;
    // real input: when both passes take the radix-4 line kernel, half of the intermediate spectrum is neither written nor
    // transformed (Hermitian symmetry); the other line kernels keep the full form
    p.half = (g_fft_radix4 && ilog2_exact(W) >= 2 && ilog2_exact(H) >= 2 && W <= 256 && H <= 256) ? 1 : 0;
    p.in0 = x; p.in1 = nullptr; p.ld_in = ldx; p.out0 = wre; p.out1 = wim; p.ld_out = C;
    rc = launch_fft<FFT_FWD_ROWS>(p, B, (hipStream_t)stream);
    if (rc) return rc;
    p.in0 = wre; p.in1 = wim; p.ld_in = C; p.out0 = amp; p.out1 = phase; p.ld_out = ld_out;
    return launch_fft<FFT_FWD_COLS>(p, B, (hipStream_t)stream);
}

extern "C" int uwir_ifft2_polar_add(const float* amp, const float* phase, int ld_in, const float* addend, int ld_add,
                                    int B, int H, int W, int C, float* out, int ldo, void* workspace,
                                    size_t workspace_bytes, uwir_stream_t stream) {
    UWIR_CHECK(amp && phase && addend && out, "ifft2: null pointer");
    int rc = fft_check(B, H, W, C, workspace, workspace_bytes);
    if (rc) return rc;
    UWIR_CHECK(ld_in >= C && ld_in % 4 == 0 && ld_add >= C && ld_add % 4 == 0 && ldo >= C && ldo % 4 == 0,
               "ifft2: bad strides");
    float* wre = (float*)workspace;
    float* wim = wre + (size_t)B * H * W * C;
    FftP p = {};
    p.H = H; p.W = W; p.C = C;
    // only the real part of the inverse is kept (sffb.py:54): the Hermitian part of the spectrum on half the columns
    p.half = (g_fft_radix4 && ilog2_exact(W) >= 2 && ilog2_exact(H) >= 2 && W <= 256 && H <= 256) ? 1 : 0;
    p.in0 = amp; p.in1 = phase; p.ld_in = ld_in; p.out0 = wre; p.out1 = wim; p.ld_out = C;
    rc = launch_fft<FFT_INV_COLS>(p, B, (hipStream_t)stream);
    if (rc) return rc;
    p.in0 = wre; p.in1 = wim; p.ld_in = C; p.out0 = out; p.out1 = nullptr; p.ld_out = ldo;
    p.addend = addend; p.ld_add = ld_add;
    return launch_fft<FFT_INV_ROWS>(p, B, (hipStream_t)stream);
}

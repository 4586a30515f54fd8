// Teacher heads on the device: the step between the DA-CLIP tokens and the student's
// conditioning, so teacher -> student runs without the .pt / .npy detour.
//   underwater_ir/teacher/mask_head.py:96-164          MaskHead.forward
//   underwater_ir/teacher/differentiable_crf.py:44-80  DifferentiableGuidedCRF.forward (mean-field iterations)
//   underwater_ir/teacher/deg_coder.py:31-59           DegradationCoder.forward
// The maps are tiny (7x7 ... 32x32 tokens, K <= 64 prompts): one block per image runs every phase, the [K, n] maps
// live in the output buffers / a caller-provided workspace (L2-resident), phases are separated by block barriers.
// The whole head chain is ONE launch instead of ~60 framework ops.
#include "common.h"

struct TeacherP {
    const float* tokens;  // [B*n, ld_tok] rows (mode 0)
    const float* text;    // [K, C]
    const float* probs_in;  // mode 1 (CRF alone): [B, K, h, w]
    const float* conf_in;   // mode 1: optional [B, K]
    const float* thr;     // [K]
    const float* image;   // [B, img_c, img_h, img_w] or null
    int ld_tok, B, n, h, w, C, K, mode;
    float tau, alpha, delta, eps;
    int morph, use_crf, crf_iters, crf_ks, img_c, img_h, img_w;
    float crf_sigma, crf_edge_w;
    float *scores, *probs, *masks, *intensity, *raw_probs;  // [B, K, h, w]
    float *avg_intensity, *global_prob, *confidence;        // [B, K]
    float *logits, *code, *conf_scale;                      // [B, K], [B, 2K], [B]
    float* ws;                                              // per image: 2*K*n + 2*n + K floats
};

#define TH_MAXK 64
#define TH_MAXKS 15

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ void __launch_bounds__(256) teacher_heads_kernel(TeacherP p) {
    __shared__ float s_tinv[TH_MAXK];                 // 1 / max(|text_k|, 1e-12)
    __shared__ float s_conf[TH_MAXK];                 // per-class message weight of the CRF
    __shared__ float s_kern[TH_MAXKS * TH_MAXKS];
    __shared__ float s_red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = p.n, K = p.K, C = p.C, h = p.h, w = p.w;
    const size_t mo = (size_t)b * K * n;               // offset of this image's [K, n] maps
    float* q = p.probs + mo;
    float* wsA = p.ws + (size_t)b * (2 * (size_t)K * n + 2 * n + K);  // [K, n] scratch (new q / dilated masks)
    float* wsU = wsA + (size_t)K * n;                                  // [K, n] unary
    float* wsG = wsU + (size_t)K * n;                                  // [n] gray, then [n] edge
    float* wsE = wsG + n;

    if (p.mode == 0) {
        // ---- cosine similarity of every token with every prompt (F.normalize: x / max(|x|, 1e-12))
        for (int k = wave; k < K; k += 4) {
            float s = 0.f;
            for (int c = lane; c < C; c += 64) { float v = p.text[(size_t)k * C + c]; s = fmaf(v, v, s); }
            s = wave_sum(s);
            if (lane == 0) s_tinv[k] = 1.f / fmaxf(sqrtf(s), 1e-12f);
        }
        __syncthreads();
        for (int t = wave; t < n; t += 4) {
            const float* tr = p.tokens + ((size_t)b * n + t) * p.ld_tok;
            float s = 0.f;
            for (int c = lane; c < C; c += 64) { float v = tr[c]; s = fmaf(v, v, s); }
            const float inv = 1.f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
            for (int k = 0; k < K; ++k) {
                float d = 0.f;
                for (int c = lane; c < C; c += 64) d = fmaf(tr[c], p.text[(size_t)k * C + c], d);
                d = wave_sum(d);
                if (lane == 0) p.scores[mo + (size_t)k * n + t] = d * inv * s_tinv[k];
            }
        }
        __syncthreads();
        // ---- softmax over the prompts (temperature), sigmoid intensity
        for (int t = tid; t < n; t += 256) {
            float mx = -INFINITY;
            for (int k = 0; k < K; ++k) mx = fmaxf(mx, p.scores[mo + (size_t)k * n + t] * p.tau);
            float sum = 0.f;
            for (int k = 0; k < K; ++k) sum += expf(p.scores[mo + (size_t)k * n + t] * p.tau - mx);
            const float inv = 1.f / sum;
            for (int k = 0; k < K; ++k) {
                const float sc = p.scores[mo + (size_t)k * n + t];
                const float pr = expf(sc * p.tau - mx) * inv;
                q[(size_t)k * n + t] = pr;
                p.raw_probs[mo + (size_t)k * n + t] = pr;
                p.intensity[mo + (size_t)k * n + t] = 1.f / (1.f + expf(-sc * p.alpha));
            }
        }
        __syncthreads();
        // ---- per-class means: avg intensity, the pre-CRF confidence band
        for (int k = wave; k < K; k += 4) {
            double si = 0.0, sc = 0.0;
            const float th = p.thr[k], up = th + p.delta;
            for (int t = lane; t < n; t += 64) {
                si += (double)p.intensity[mo + (size_t)k * n + t];
                const float pr = q[(size_t)k * n + t];
                sc += (pr >= th && pr <= up) ? 1.0 : 0.0;
            }
            si = wave_sum(si);
            sc = wave_sum(sc);
            if (lane == 0) {
                p.avg_intensity[(size_t)b * K + k] = (float)(si / n);
                s_conf[k] = (float)(sc / n);
            }
        }
        __syncthreads();
    } else {
        for (int i = tid; i < K * n; i += 256) q[i] = p.probs_in[mo + i];
        for (int k = tid; k < K; k += 256) s_conf[k] = p.conf_in ? p.conf_in[(size_t)b * K + k] : 1.f;
        __syncthreads();
    }

    if (p.use_crf) {
        // ---- guidance: gray = channel mean of the (bilinearly resized, align_corners=False) image
        const float sy = (float)p.img_h / h, sx = (float)p.img_w / w;
        const bool same = p.img_h == h && p.img_w == w;
        for (int t = tid; t < n; t += 256) {
            const int y = t / w, x = t - y * w;
            float acc = 0.f;
            for (int c = 0; c < p.img_c; ++c) {
                const float* im = p.image + ((size_t)b * p.img_c + c) * p.img_h * p.img_w;
                if (same) {
                    acc += im[y * w + x];
                } else {
                    const float fy = fmaxf((y + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf((x + 0.5f) * sx - 0.5f, 0.f);
                    const int y0 = min((int)fy, p.img_h - 1), x0 = min((int)fx, p.img_w - 1);
                    const int y1 = min(y0 + 1, p.img_h - 1), x1 = min(x0 + 1, p.img_w - 1);
                    const float ly = fy - y0, lx = fx - x0;
                    const float top = im[y0 * p.img_w + x0] * (1.f - lx) + im[y0 * p.img_w + x1] * lx;
                    const float bot = im[y1 * p.img_w + x0] * (1.f - lx) + im[y1 * p.img_w + x1] * lx;
                    acc += top * (1.f - ly) + bot * ly;
                }
            }
            wsG[t] = acc / p.img_c;
        }
        // Gaussian smoothing taps, normalised (differentiable_crf.py:10-15)
        const int ks = p.crf_ks, r = ks / 2;
        if (tid < ks * ks) {
            const int iy = tid / ks - r, ix = tid % ks - r;
            s_kern[tid] = expf(-(float)(ix * ix + iy * iy) / (2.f * p.crf_sigma * p.crf_sigma));
        }
        __syncthreads();
        if (wave == 0) {
            float s = 0.f;
            for (int i = lane; i < ks * ks; i += 64) s += s_kern[i];
            s = wave_sum(s);
            if (lane == 0) s_red[0] = fmaxf(s, 1e-6f);
        }
        __syncthreads();
        if (tid < ks * ks) s_kern[tid] /= s_red[0];
        for (int t = tid; t < n; t += 256) {
            const int y = t / w, x = t - y * w;
            const float g0 = wsG[t];
            const float gx = x + 1 < w ? wsG[t + 1] - g0 : 0.f, gy = y + 1 < h ? wsG[t + w] - g0 : 0.f;
            wsE[t] = expf(-p.crf_edge_w * sqrtf(gx * gx + gy * gy + 1e-6f));
        }
        for (int i = tid; i < K * n; i += 256) wsU[i] = -logf(fmaxf(q[i], 1e-6f));
        __syncthreads();
        // ---- mean-field iterations: q <- softmax_k( -(unary + conf_k * edge * (G*q - q)) ), ping-pong q <-> wsA
        float* cur = q;
        float* nxt = wsA;
        for (int it = 0; it < p.crf_iters; ++it) {
            for (int t = tid; t < n; t += 256) {
                const int y = t / w, x = t - y * w;
                const float e = wsE[t];
                float mx = -INFINITY;
                for (int k = 0; k < K; ++k) {
                    const float* qk = cur + (size_t)k * n;
                    float sm = 0.f;
                    for (int iy = 0; iy < ks; ++iy) {
                        const int yy = y + iy - r;
                        if (yy < 0 || yy >= h) continue;
                        for (int ix = 0; ix < ks; ++ix) {
                            const int xx = x + ix - r;
                            if (xx >= 0 && xx < w) sm = fmaf(s_kern[iy * ks + ix], qk[yy * w + xx], sm);
                        }
                    }
                    const float z = -(wsU[(size_t)k * n + t] + s_conf[k] * (e * (sm - qk[t])));
                    nxt[(size_t)k * n + t] = z;
                    mx = fmaxf(mx, z);
                }
                float sum = 0.f;
                for (int k = 0; k < K; ++k) {
                    const float ez = expf(nxt[(size_t)k * n + t] - mx);
                    nxt[(size_t)k * n + t] = ez;
                    sum += ez;
                }
                const float inv = 1.f / sum;
                for (int k = 0; k < K; ++k) nxt[(size_t)k * n + t] *= inv;
            }
            __syncthreads();
            float* tmp = cur; cur = nxt; nxt = tmp;
        }
        if (cur != q) {
            for (int i = tid; i < K * n; i += 256) q[i] = cur[i];
            __syncthreads();
        }
    }
    if (p.mode != 0) return;

    // ---- masks = close(probs >= thr): dilation then erosion over the in-bounds window (max_pool2d pads with -inf)
    const int mr = p.morph / 2;
    for (int i = tid; i < K * n; i += 256) {
        const int k = i / n, t = i - k * n, y = t / w, x = t - y * w;
        const float th = p.thr[k];
        float d = 0.f;
        for (int yy = max(0, y - mr); yy <= min(h - 1, y + mr); ++yy)
            for (int xx = max(0, x - mr); xx <= min(w - 1, x + mr); ++xx)
                d = fmaxf(d, q[(size_t)k * n + yy * w + xx] >= th ? 1.f : 0.f);
        wsA[i] = d;
    }
    __syncthreads();
    for (int i = tid; i < K * n; i += 256) {
        const int k = i / n, t = i - k * n, y = t / w, x = t - y * w;
        float e = 1.f;
        for (int yy = max(0, y - mr); yy <= min(h - 1, y + mr); ++yy)
            for (int xx = max(0, x - mr); xx <= min(w - 1, x + mr); ++xx) e = fminf(e, wsA[(size_t)k * n + yy * w + xx]);
        p.masks[mo + i] = e;
    }
    // ---- global statistics and the degradation code z_d = [logit(global_prob) | avg_intensity]
    for (int k = wave; k < K; k += 4) {
        double sp = 0.0, sc = 0.0;
        const float th = p.thr[k], up = th + p.delta;
        for (int t = lane; t < n; t += 64) {
            const float pr = q[(size_t)k * n + t];
            sp += (double)pr;
            sc += (pr >= th && pr <= up) ? 1.0 : 0.0;
        }
        sp = wave_sum(sp);
        sc = wave_sum(sc);
        if (lane == 0) {
            const float gp = (float)(sp / n), cf = (float)(sc / n);
            p.global_prob[(size_t)b * K + k] = gp;
            p.confidence[(size_t)b * K + k] = cf;
            s_conf[k] = cf;
            const float cl = fminf(fmaxf(gp, p.eps), 1.f - p.eps);
            const float lg = logf(cl / (1.f - cl));
            p.logits[(size_t)b * K + k] = lg;
            p.code[(size_t)b * 2 * K + k] = lg;
            p.code[(size_t)b * 2 * K + K + k] = p.avg_intensity[(size_t)b * K + k];
        }
    }
    __syncthreads();
    if (wave == 0) {
        float s = 0.f;
        for (int k = lane; k < K; k += 64) s += s_conf[k];
        s = wave_sum(s);
        if (lane == 0) p.conf_scale[b] = 1.f - s / K;
    }
}

extern "C" size_t uwir_teacher_heads_workspace_bytes(int B, int n, int K) {
    if (B <= 0 || n <= 0 || K <= 0) return 0;
    return (size_t)B * (2 * (size_t)K * n + 2 * n + K) * sizeof(float);
}

static int check_crf(const uwir_teacher_args& a) {
    UWIR_CHECK(a.crf_iters >= 0 && a.crf_ks >= 1 && a.crf_ks % 2 == 1 && a.crf_ks <= TH_MAXKS,
               "teacher_heads: crf kernel must be odd and <= %d", TH_MAXKS);
    UWIR_CHECK(a.crf_sigma > 0.f, "teacher_heads: crf sigma must be positive");
    UWIR_CHECK(a.image && a.img_c > 0 && a.img_h > 0 && a.img_w > 0, "teacher_heads: the CRF needs the guidance image");
    return UWIR_OK;
}

static void fill_common(TeacherP& p, const uwir_teacher_args& a) {
    p.thr = a.thresholds;
    p.image = a.image;
    p.B = a.B; p.n = a.h * a.w; p.h = a.h; p.w = a.w; p.C = a.C; p.K = a.K;
    p.tau = a.temperature; p.alpha = a.alpha; p.delta = a.confidence_delta; p.eps = a.eps;
    p.morph = a.morphology_kernel; p.use_crf = a.use_crf; p.crf_iters = a.crf_iters; p.crf_ks = a.crf_ks;
    p.img_c = a.img_c; p.img_h = a.img_h; p.img_w = a.img_w; p.crf_sigma = a.crf_sigma; p.crf_edge_w = a.crf_edge_weight;
    p.ws = (float*)a.workspace;
}

extern "C" int uwir_teacher_heads(const uwir_teacher_args* args, uwir_stream_t stream) {
    UWIR_CHECK(args, "teacher_heads: null args");
    const uwir_teacher_args& a = *args;
    UWIR_CHECK(a.tokens && a.text && a.thresholds, "teacher_heads: null input");
    UWIR_CHECK(a.scores && a.probs && a.masks && a.intensity && a.raw_probs && a.avg_intensity && a.global_prob &&
                   a.confidence && a.logits && a.code && a.confidence_scale,
               "teacher_heads: null output");
    UWIR_CHECK(a.B > 0 && a.B <= 65535 && a.h > 0 && a.w > 0 && a.C > 0 && a.K > 0 && a.K <= TH_MAXK,
               "teacher_heads: bad shape (K <= %d)", TH_MAXK);
    UWIR_CHECK((long long)a.h * a.w * a.K < (1 << 24), "teacher_heads: grid too large");
    UWIR_CHECK(a.ld_tokens >= a.C, "teacher_heads: ld_tokens < C");
    UWIR_CHECK(a.morphology_kernel >= 1 && a.morphology_kernel % 2 == 1, "teacher_heads: morphology kernel must be odd");
    if (a.use_crf && check_crf(a) != UWIR_OK) return UWIR_EINVAL;
    UWIR_CHECK(a.workspace && a.workspace_bytes >= uwir_teacher_heads_workspace_bytes(a.B, a.h * a.w, a.K) &&
                   ((uintptr_t)a.workspace & 3) == 0,
               "teacher_heads: workspace too small / misaligned");
    TeacherP p = {};
    fill_common(p, a);
    p.mode = 0;
    p.tokens = a.tokens; p.ld_tok = a.ld_tokens; p.text = a.text;
    p.scores = a.scores; p.probs = a.probs; p.masks = a.masks; p.intensity = a.intensity; p.raw_probs = a.raw_probs;
    p.avg_intensity = a.avg_intensity; p.global_prob = a.global_prob; p.confidence = a.confidence;
    p.logits = a.logits; p.code = a.code; p.conf_scale = a.confidence_scale;
    hipLaunchKernelGGL(teacher_heads_kernel, dim3(a.B), dim3(256), 0, (hipStream_t)stream, p);
    UWIR_LAUNCH_CHECK("teacher_heads");
    return UWIR_OK;
}

extern "C" int uwir_guided_crf(const uwir_teacher_args* args, const float* probs_in, const float* confidence_in,
                               uwir_stream_t stream) {
    UWIR_CHECK(args && probs_in, "guided_crf: null args");
    const uwir_teacher_args& a = *args;
    UWIR_CHECK(a.probs && a.probs != probs_in, "guided_crf: output must be a separate buffer");
    UWIR_CHECK(a.B > 0 && a.B <= 65535 && a.h > 0 && a.w > 0 && a.K > 0 && a.K <= TH_MAXK, "guided_crf: bad shape");
    UWIR_CHECK((long long)a.h * a.w * a.K < (1 << 24), "guided_crf: grid too large");
    if (check_crf(a) != UWIR_OK) return UWIR_EINVAL;
    UWIR_CHECK(a.workspace && a.workspace_bytes >= uwir_teacher_heads_workspace_bytes(a.B, a.h * a.w, a.K) &&
                   ((uintptr_t)a.workspace & 3) == 0,
               "guided_crf: workspace too small / misaligned");
    TeacherP p = {};
    fill_common(p, a);
    p.mode = 1;
    p.use_crf = 1;
    p.probs_in = probs_in;
    p.conf_in = confidence_in;
    p.probs = a.probs;
    hipLaunchKernelGGL(teacher_heads_kernel, dim3(a.B), dim3(256), 0, (hipStream_t)stream, p);
    UWIR_LAUNCH_CHECK("guided_crf");
    return UWIR_OK;
}

/*
 * uwir.h -- C ABI of libuwir_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * restoration forward pass of anbinh93/Underwater-image-restoration.
 *
 * The reference has NO native / FFI layer (pure Python over torch ops), so
 * each entry point names the reference *Python* function whose device work it replaces
 * (paths relative to /root/reference).  S/ = underwater_ir/student/naf_unet_wfi/,
 * L/ = legacy/third_party/universal-image-restoration/.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch caching allocator);
 *     the library allocates nothing and keeps no mutable global state except the
 *     thread-local last-error string;
 *   - activations are NHWC ("token major"): pixel (b,y,x) is row m = (b*H + y)*W + x of a
 *     row-major [M, ld] matrix, ld >= channels (lets callers write into concat buffers);
 *   - every call only ENQUEUES work on `stream` (a hipStream_t) and never synchronises;
 *   - return value: 0 = OK, negative = uwir_status; uwir_last_error() gives the message.
 */
#ifndef UWIR_H
#define UWIR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* uwir_stream_t; /* hipStream_t */

enum uwir_status { UWIR_OK = 0, UWIR_EINVAL = -1, UWIR_EUNSUPPORTED = -2, UWIR_ELAUNCH = -3 };
enum uwir_dtype { UWIR_F32 = 0, UWIR_BF16 = 1 };
enum uwir_act { UWIR_ACT_NONE = 0, UWIR_ACT_RELU = 1, UWIR_ACT_GELU = 2, UWIR_ACT_SIGMOID = 3, UWIR_ACT_SILU = 4 };

int uwir_version(void);
/* Measurement switch with no counterpart in the reference, for A/B timing in one process (tools/): bit 0 selects the
 * 3x3 halo-conv tile shape (0 = 8-row tiles, the default; 1 = 16-row tiles, identical results), bit 1 keeps the
 * large-K 1x1 GEMMs on the 128 x 128 kernel instead of the 256 x 192 one (same sums in another order), bit 2 keeps
 * the radix-2 FFT line kernel instead of the radix-4 one (same transform, another summation order), bits 3-5 pick the
 * 3x3 conv kernel for bf16 rows (0 = by shape, the default; 1 = the register-staged halo kernel; 2 / 3 / 4 = the LDS-DMA
 * kernel with 8-row tiles and a ring of 4 / 3, or 16-row tiles and a ring of 2; 5 = 16-row tiles, ring of 3; 7 = by shape but never the
 * resident-weight kernel; identical results for every value), bit 6 keeps the large-K 1x1 GEMMs off the 256 x 288 kernel
 * (N % 288 == 0: the 256 x 192 one where N % 192 == 0, else 128 x 128; same sums in the same order). */
void uwir_conv_variant(int variant);
/* Measurement hook with no counterpart in the reference: the 256 x 288 large-K bf16 GEMM (out[M, N] = A[M, K] Wt[N, K]^T + bias,
 * dense rows; N % 288 == 0, K % 64 == 0, K >= 192, M >= 4096) or one of its timing-only ablations, which leave `out`
 * undefined: 0 = the kernel as uwir_conv_gemm launches it, 1 = every wave issues its LDS-DMA right after the stage barrier,
 * 2 = every wave between the two k-halves, 3 = no MFMAs, 4 = no LDS-DMA inside the K loop, 5 = the LDS-DMA fill alone. */
int uwir_gemm_wide_probe(const void* A, const void* Wt, void* out, const float* bias, int M, int N, int K, int variant,
                         uwir_stream_t stream);
/* Test / measurement switch, no counterpart in the reference: tiles walked by one block of the pipelined depthwise
 * kernel (0 = chosen from the grid size, the default).  Results are identical for every value. */
void uwir_dwconv_tiles_per_block(int nt);
/* Measurement switch, no counterpart in the reference: where the packed-stream attention kernel sums its softmax
 * denominators when the V^T tile has no spare row (-1 = per-shape default, 0 = ones-operand MFMA, 1 = packed fp32 adds). */
void uwir_attn_variant(int lsv);
/* Measurement switch, no counterpart in the reference: bit 0 clear keeps the student's head conv on bf16 rows on the
 * per-pixel tile kernel, bit 1 set keeps its stem conv into bf16 rows on the thread-per-pixel kernel; 1 (default) takes the
 * MFMA kernels for both (hi + lo operand splits: they agree with the exact forms to ~2^-16). */
void uwir_head_variant(int variant);
const char* uwir_last_error(void);

/* ---- S/dwt_fft.py:33-39 dwt2 -------------------------------------------------------------
 * x [B,H,W,C] fp32 (pixel stride ldx) -> ll [B,H/2,W/2,C] (stride ld_ll) and
 * hf [B,H/2,W/2,3C] = [lh|hl|hh] (stride ld_hf); odd trailing row/column dropped like the
 * reference's stride-2 valid conv.  C % 4 == 0.  ll and hf are written as sub_dtype: fp32, or bf16 (the student's
 * bf16 mode keeps its sub-band stream -- ll, hf and the WTB's residual rows -- in bf16; the Haar sums are fp32). */
int uwir_dwt2(const void* x, int x_dtype, int ldx, int B, int H, int W, int C, void* ll, int ld_ll, void* hf, int ld_hf,
              int sub_dtype, uwir_stream_t stream);

/* ---- S/dwt_fft.py:42-58 idwt2 ------------------------------------------------------------
 * ll [B,h,w,C] fp32, hf [B,h,w,3C]=[lh|hl|hh] of hf_dtype (fp32 or bf16) -> out [B,2h,2w,C] of out_dtype. */
int uwir_idwt2(const float* ll, int ld_ll, const void* hf, int hf_dtype, int ld_hf, int B, int h, int w, int C, void* out,
               int out_dtype, int ldo, uwir_stream_t stream);

/* ---- S/wfi_gate.py:29-45 WFIGate + S/dwt_fft.py:42-58 idwt2 in one pass (bf16 mode) ----------------------------
 * gate_in [M = B*h*w, ldg] fp32 rows = [t_out (C) | f_out (C) | masks resized to (h, w) (Kg - 2C columns, zero-padded)],
 * w1 [64, Kg] bf16 = net.0.weight without its cond columns, bias1 [B, 64] fp32 = net.0.bias + the cond columns applied to
 * cond_proj(z_d) (per image), w2 [C, 64] bf16 = net.2.weight, b2 [C] fp32:
 *   alpha = sigmoid(w2 relu(w1 gate_in + bias1) + b2)      -> alpha [M, C] fp32 (the model's alpha map, NHWC)
 *   merged = alpha t_out + (1 - alpha) f_out               -> merged [M, C] fp32 if not NULL (diagnostic taps)
 *   out = idwt2(merged, hf)                                -> out [B, 2h, 2w, C] bf16 (pixel stride ldo)
 * with hf [M, ld_hf] bf16 = [lh | hl | hh].  The same values as uwir_conv_gemm (relu) -> uwir_conv_gemm (sigmoid, blend)
 * -> uwir_idwt2 up to the summation order inside the bf16 MFMAs; built for C in {64, 128, 256} and Kg = 2C or 2C + 16
 * (uwir_gate_idwt_supported), other shapes take the three calls. */
int uwir_gate_idwt_supported(int C, int Kg);
int uwir_gate_idwt(const float* gate_in, int ldg, int Kg, const void* w1, const void* w2, const float* bias1, const float* b2,
                   const void* hf, int ld_hf, int B, int h, int w, int C, void* out, int ldo, float* alpha, float* merged,
                   uwir_stream_t stream);
/* The same gate WITHOUT the inverse transform, for a block that is followed by another block of the same stage
 * (S/model.py:40 then :31 of the next block): dwt2(idwt2(merged, lh, hl, hh)) returns (merged, lh, hl, hh) up to the factor
 * 4 fl32(2^-1/2)^4 = 1 - 1.2e-7, so the next block takes merged (written here as bf16 rows, pixel stride ld_merged) as its
 * ll and keeps the hf rows it already holds; the full-resolution map is never formed.  alpha as above. */
int uwir_gate_blend(const float* gate_in, int ldg, int Kg, const void* w1, const void* w2, const float* bias1, const float* b2,
                    int B, int h, int w, int C, float* alpha, void* merged_bf16, int ld_merged, uwir_stream_t stream);

/* ---- S/wtb.py:117-124 the WTB's feed-forward block in one pass (bf16 mode) -----------------------------------------
 *   out = W2 gelu(W0_b x + bias0_b) + b2 + x                 (exact erf GELU, as nn.GELU())
 * x [B*n, ldx] bf16 rows with C channels (the block's residual stream), w0_batched [B][2C][C] bf16 and bias0_batched [B][2C]
 * fp32 = ffn.0 with the norm2 affine of image b folded in (uwir_fold_affine_weights; w0_batch_stride elements between
 * images), w2_permuted [C][2C] bf16 = ffn.2.weight with the columns of every 32-block in the order
 * 4q + e, 16 + 4q + e (q = 0..3, e = 0..3) -- position 8q + e' holds column 4q + e' (e' < 4) or 16 + 4q + e' - 4 --, b2 [C],
 * out [B*n, ldo] bf16 (may alias nothing else).  The [B*n, 2C] hidden tensor of the two-launch form (uwir_conv_gemm with
 * GELU, uwir_conv_gemm with the residual) never reaches HBM.  Built for C = 192 (n % 256 == 0) and C = 384 (n % 128 == 0):
 * uwir_ffn_fused_supported; other shapes take the two launches. */
int uwir_ffn_fused_supported(int C, int n);
int uwir_ffn_fused(const void* x, int ldx, const void* w0_batched, long long w0_batch_stride, const float* bias0_batched,
                   const void* w2_permuted, const float* b2, int B, int n, int C, void* out, int ldo, uwir_stream_t stream);

/* ---- S/utils.py:10-36 LayerNorm2d / AdaLayerNorm2d ------------------------------------------
 * Per-(b,c) statistics over the n = H*W rows of x [B,n,C]; emits the folded affine
 *   LN(x)*(1+scale)+shift == a[b,c]*x + s[b,c]
 * (a = rstd*gamma*(1+scale), s = (beta - mean*rstd*gamma)*(1+scale) + shift) that the
 * consumer kernel applies while loading x, optionally mean[b,c] (global average pool,
 * S/deg_head.py:20) and pooled[b,c] = spatial mean of the normalised map (= beta*(1+scale)+shift).  scale_shift [B, ss_ld] holds [scale(C)|shift(C)] per sample or is NULL
 * (plain LayerNorm2d).  gamma/beta NULL => statistics only (mean_out must be given).
 * workspace: >= uwir_ln2d_workspace_bytes(B,n,C) bytes.  C % 4 == 0.  x is fp32 or bf16 (x_dtype); sums are fp32 per
 * block and fp64 across blocks either way. */
size_t uwir_ln2d_workspace_bytes(int B, int n, int C);
int uwir_ln2d_affine(const void* x, int x_dtype, int ldx, int B, int n, int C, float eps, const float* gamma, const float* beta,
                     const float* scale_shift, int ss_ld, float* a_out, float* s_out, float* mean_out,
                     float* pooled_out, void* workspace, size_t workspace_bytes, uwir_stream_t stream);

/* ---- allocation-free variant of uwir_ln2d_affine.  NO COUNTERPART IN THE REFERENCE: it exists for callers that
 * cannot allocate device memory or hand over a workspace -- code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  Same arguments and bit-identical results, minus the workspace: the fp64 partial
 * sums go to a built-in static device array of UWIR_LN2D_NOALLOC_SCRATCH_BYTES per GPU (held until the process
 * ends).  Shapes whose partials exceed it (uwir_ln2d_workspace_bytes(B,n,C) > uwir_ln2d_noalloc_max_bytes()) are
 * rejected with UWIR_EINVAL.  Calls no hipMalloc / hipFree / allocating routine.  All calls on one GPU share that GPU's
 * array: calls that may overlap on one GPU (two streams, two threads) race -- order them, or use uwir_ln2d_affine
 * with separate workspaces.  The device is the one current at the call (the stream must belong to it). */
#define UWIR_LN2D_NOALLOC_SCRATCH_BYTES (16u << 20)
size_t uwir_ln2d_noalloc_max_bytes(void);
int uwir_ln2d_affine_noalloc(const void* x, int x_dtype, int ldx, int B, int n, int C, float eps, const float* gamma,
                             const float* beta, const float* scale_shift, int ss_ld, float* a_out, float* s_out,
                             float* mean_out, float* pooled_out, uwir_stream_t stream);

/* ---- materialised form of the same norm (S/utils.py:18-36): out[m,c] = a[b,c]*x[m,c] + s[b,c], b = m / n, out fp32
 * or bf16.  For consumers that are better fed a ready bf16 operand (LDS-DMA GEMM path) than normalise-on-load. */
int uwir_affine_cast(const void* x, int x_dtype, int ldx, int B, int n, int C, const float* a, const float* s, void* out,
                     int out_dtype, int ldo, uwir_stream_t stream);

/* ---- the same norm folded into the NEXT 1x1 conv's weights (S/utils.py:18-36 feeding S/wtb.py:50,71):
 *   W (a_b x + s_b) + bias = (W diag(a_b)) x + (W s_b + bias)
 * W fp32 [N, K] -> Wb bf16 [B, N, K] (per-image weight panels, see uwir_gemm_args.w_batch_stride), bb fp32 [B, N] (per-image
 * bias rows, bias may be NULL); a, s fp32 [B, K] as written by uwir_ln2d_affine.  K % 8 == 0.  With it the consumer GEMM
 * reads the un-normalised rows and uwir_affine_cast's round trip of the activation disappears. */
int uwir_fold_affine_weights(const float* W, int N, int K, int B, const float* a, const float* s, const float* bias, void* Wb,
                             float* bb, uwir_stream_t stream);

/* ---- S/wtb.py:13-28,118-119 ChannelAttention folded into the residual ------------------------
 * With xn = a*x + s (the affine above) and w = sigmoid(MLP(mean(xn))) [B,C]:
 *   x + xn*w == x*(1 + a*w) + s*w   ->  out_scale = 1 + a*w, out_shift = s*w   (count = B*C).
 * mean(xn) needs no reduction: it is pooled_out of uwir_ln2d_affine. */
int uwir_chan_attn_fold(const float* a, const float* s, const float* w, float* out_scale, float* out_shift, int count,
                        uwir_stream_t stream);

/* ---- nn.Linear on tiny batches (cond projections S/utils.py:31, S/wfi_gate.py:36; channel
 * attention MLP S/wtb.py:16-21; degradation head S/deg_head.py:13-17) ------------------------
 * y[b,o] = act(sum_i x[b,i]*W[o,i] + bias[o]); x stride ldx, y stride ldy. */
int uwir_linear(const float* x, int ldx, int B, int I, const float* W, const float* bias, int O, int act, float* y,
                int ldy, uwir_stream_t stream);

/* ---- implicit-GEMM convolution / 1x1 conv / linear on MFMA ----------------------------------
 * Replaces nn.Conv2d(k=1) (S/wtb.py:50,63,65,71-73; S/sffb.py:32-35; S/cfc.py:19-22;
 * S/wfi_gate.py:23-25; S/model.py:108), Conv2d(k=2,s=2) / ConvTranspose2d(k=2,s=2)
 * (S/model.py:93-104), and the k x k convs of the IR-SDE UNet.
 *   out[m, n] = epilogue( sum_{ky,kx,c} A[b, oy*stride+ky-pad, ox*stride+kx-pad, c] * Wt[n, (ky*KW+kx)*C + c] )
 * m = (b*OH + oy)*OW + ox.  Out-of-image taps contribute 0 (zero padding is applied AFTER
 * the optional input affine, like conv(pad)(norm(x))). */
typedef struct uwir_gemm_args {
    const void* A;        /* activation, NHWC, a_dtype, pixel stride lda */
    int a_dtype, lda;
    int B, H, W, C;       /* input geometry; C % 8 == 0 */
    int KH, KW, stride, pad;
    int OH, OW;
    const float* a_scale; /* optional [B, C] prologue affine (both or neither) */
    const float* a_shift;
    const void* Wt;       /* [N, KH*KW*C] row-major, w_dtype == compute dtype */
    int w_dtype, N;
    const float* bias;       /* [N] or NULL */
    const float* batch_bias; /* [B, N] or NULL (per-sample bias, e.g. folded cond_proj of the gate) */
    int act;                 /* uwir_act, applied to (acc + bias) * batch_scale + batch_bias */
    const void* res;         /* optional residual [M, ldr] of res_dtype (fp32 unless set) added after act */
    int ldr;
    const float* res_scale;  /* optional [B, N]: residual enters as res*res_scale + res_shift */
    const float* res_shift;
    void* out;               /* [M, ldo] out_dtype (may be NULL when only blend_out is wanted) */
    int out_dtype, ldo;
    int store_mode;          /* 0: row m; 1: depth-to-space 2x2 (ConvTranspose2d k=2,s=2): N = 4*Cout,
                                n = (dy*2+dx)*Cout + co goes to pixel (b, 2*oy+dy, 2*ox+dx), channel co;
                                2: GEGLU (U/attention.py:37-44 fused into its projection): weight rows come in groups
                                of 16 = 8 value rows then their 8 gate rows, out [M, N/2] = value * gelu(gate) */
    const float* blend_t;    /* optional gate blend (S/wfi_gate.py:44): blend_out = v*t + (1-v)*f, */
    const float* blend_f;    /* v = epilogue value, t/f/blend_out fp32 with strides ld_blend_in / ld_blend_out */
    int ld_blend_in;
    float* blend_out;
    int ld_blend_out;
    int compute;             /* UWIR_F32: exact-fp32 MFMA; UWIR_BF16: bf16 MFMA, fp32 accumulate */
    const float* col_scale;  /* optional [N]: value *= col_scale[n] after act, before the residual (used to hand
                                attention a Q that already carries scale*log2(e), rounded to bf16 only once) */
    const float* batch_scale; /* optional [B, N]: (acc + bias) *= batch_scale[b,n] BEFORE batch_bias and act -- the
                                 IR-SDE ResBlock's x*(scale+1)+shift (U/models/modules/module_util.py:121-128) */
    int ups;                  /* 1: A is read through a nearest-neighbour 2x upsample (nn.Upsample(scale_factor=2) +
                                 Conv2d, module_util.py:100-104); H, W stay the SOURCE size, OH/OW refer to 2H x 2W */
    int batch_ld;             /* row stride of batch_bias / batch_scale (0 = N): lets one [B, total] matrix hold the
                                 per-sample vectors of many layers */
    void* workspace;          /* optional, 16-byte aligned, >= uwir_conv_gemm_workspace_bytes(args) bytes: lets a plain
                                 bf16 GEMM with fewer output tiles than CUs (ViT-B/32: 50 tokens x batch rows) cut K
                                 into slices, one block each, summed in slice order by a second small launch
                                 (bit-reproducible).  Calls sharing a workspace must be ordered on one stream. */
    size_t workspace_bytes;
    int res_dtype;            /* dtype of res: UWIR_F32 (0, the default) or UWIR_BF16 (the IR-SDE UNet's bf16 activation
                                 stream); an affine residual (res_scale) and the lean epilogue classes take fp32 rows */
    long long w_batch_stride; /* 0: one weight matrix.  > 0 (elements): image b multiplies by Wt + b * w_batch_stride (the
                                 panels of uwir_fold_affine_weights).  Plain bf16 1x1 GEMMs only (bf16 rows, K % 64 == 0,
                                 N > 64) and OH * OW % 128 == 0 (a 128-row tile never straddles two images); anything else
                                 is rejected with UWIR_EUNSUPPORTED */
} uwir_gemm_args;
size_t uwir_conv_gemm_workspace_bytes(const uwir_gemm_args* args); /* 0 when the launch would not use one */
int uwir_conv_gemm(const uwir_gemm_args* args, uwir_stream_t stream);

/* ---- multi-scale depthwise conv 3x3 + 5x5 + 7x7 (zero pad) fused with the concat -----------
 * S/wtb.py:53-62,91 and S/sffb.py:21-31,46 (depthwise halves).  x [B,H,W,C] (x_dtype) ->
 * out [B,H,W,3C] = [dw3|dw5|dw7] (out_dtype).  Weights are tap-major: w3 [9,C], w5 [25,C],
 * w7 [49,C]; biases [C] (NULL = 0).  Optional input affine [B,C] (normalise-on-load; padding
 * stays 0).  C % 4 == 0. */
int uwir_dwconv_ms(const void* x, int x_dtype, int ldx, int B, int H, int W, int C, const float* in_scale,
                   const float* in_shift, const float* w3, const float* w5, const float* w7, const float* b3,
                   const float* b5, const float* b7, void* out, int out_dtype, int ldo, uwir_stream_t stream);

/* ---- global multi-head attention, flash style (no n x n matrix) ----------------------------
 * S/wtb.py:94-115 (one V) and S/cfc.py:24-39 (one softmax(QK^T/sqrt(hd)) applied to two V's).
 * q,k,v1,v2: [B*n, ld*] matrices (dtype), head h occupies columns [h*hd, (h+1)*hd).
 * o1/o2 likewise (out_dtype).  v2/o2 may be NULL.  hd % 4 == 0, hd <= 128 (above 64 a v2 / o2 pair runs as two passes).
 * scale = 1/sqrt(hd) normally; scale == 0 means "q is already multiplied by (1/sqrt(hd))*log2(e)" (the producing
 * GEMM applied it through col_scale), which saves one bf16 rounding of q.
 * workspace (optional, >= uwir_flash_attn_workspace_bytes(B, n, heads), 4-byte aligned): enables the bf16 fast path --
 * a pre-pass stores max_k |k|^2 per (b, head) there; where the a-priori score bound ub = |c q| * max|k| is <= 90 (log2
 * units) exp2 cannot leave the fp32/bf16 range (P = 2^s in [2^-90, 2^90] is a normal bf16; the fp32 row sum l is
 * <= n 2^90 <= 2^107 for n <= 2^17; O <= l max|v|, finite for |v| < 2^20), the (mathematically cancelling) softmax
 * shift is dropped and no running max / rescale is computed; blocks with a larger bound run the online-max loop.
 * With >= uwir_flash_attn_packed_workspace_bytes(B, n, heads, hd, nv) bytes, 256-byte aligned, long bf16 sequences with
 * hd 16 or 32 (n >= 1024, at least 256 query blocks) take the packed-stream kernels: the pre-pass also rewrites K and
 * V^T into MFMA-fragment order inside the workspace and the main kernel streams them with no LDS and no barrier; its
 * pre-pass also stores the per-dimension extremes of the keys, and the bound in use there is the smaller of ub and
 * max(sum_d max(c q_d kmax_d, c q_d kmin_d), -sum_d min(...)) (every k_d lies in [kmin_d, kmax_d]).  The results are the
 * same function of the inputs to bf16 rounding; shapes outside that range return the small size. */
/* Measurement hook with no counterpart in the reference: arms two hipEvent_t for the NEXT uwir_flash_attn call of the
 * calling thread; they are recorded on that call's stream right before / after the launch of its main kernel only
 * (not the pre-pass, not the fallback launch), so a benchmark can time the dominant kernel in place. */
int uwir_flash_attn_probe(void* hip_event_before, void* hip_event_after);
size_t uwir_flash_attn_workspace_bytes(int B, int n, int heads);
size_t uwir_flash_attn_packed_workspace_bytes(int B, int n, int heads, int hd, int nv);
int uwir_flash_attn(const void* q, int ldq, const void* k, int ldk, const void* v1, int ldv1, const void* v2,
                    int ldv2, int dtype, int B, int n, int heads, int hd, float scale, void* o1, int ldo1,
                    void* o2, int ldo2, int out_dtype, void* workspace, size_t workspace_bytes, uwir_stream_t stream);

/* ---- S/dwt_fft.py:61-72 fft2 / ifft2 (norm="ortho", full complex spectrum) ------------------
 * fwd: x [B,H,W,C] fp32 -> amp = |F|, phase = atan2(Im,Re) (both [B,H,W,C], stride ld_out).
 * inv: out = addend + Re(ifft2(amp*exp(i*phase)))  (S/sffb.py:52-54: spatial + freq).
 * workspace: >= uwir_fft_workspace_bytes(B,H,W,C).  H, W <= 1024; line lengths that are powers of two (<= 256) use an
 * in-LDS radix-4 FFT, other lengths of the form 2^a 3^b 5^c a mixed-radix one, anything else a direct DFT.  C % 4 == 0. */
size_t uwir_fft_workspace_bytes(int B, int H, int W, int C);
int uwir_fft2_amp_phase(const float* x, int ldx, int B, int H, int W, int C, float* amp, float* phase, int ld_out,
                        void* workspace, size_t workspace_bytes, uwir_stream_t stream);
int uwir_ifft2_polar_add(const float* amp, const float* phase, int ld_in, const float* addend, int ld_add, int B,
                         int H, int W, int C, float* out, int ldo, void* workspace, size_t workspace_bytes,
                         uwir_stream_t stream);

/* ---- F.interpolate(masks, size, mode="bilinear", align_corners=False), S/wfi_gate.py:39-40 --
 * src NCHW [B,M,hs,ws] fp32 -> dst NHWC rows (pixel stride ldd, M channels written at dst). */
int uwir_bilinear_nchw_to_nhwc(const float* src, int B, int M, int hs, int ws, float* dst, int ldd, int hd, int wd,
                               uwir_stream_t stream);

/* ---- F.interpolate(x, size=skip.shape[-2:], mode="bilinear", align_corners=False), S/model.py:150-151 --
 * The decoder's resize of an up-sampled map to its skip's size, taken whenever a level's side is odd.
 * src rows [B,hs,ws,C] (pixel stride ld_src) -> dst rows [B,hd,wd,C] (pixel stride ldd), both of dtype (fp32 or bf16);
 * C and both strides multiples of 4. */
int uwir_bilinear_rows(const void* src, int ld_src, int B, int hs, int ws, int C, void* dst, int ldd, int hd, int wd,
                       int dtype, uwir_stream_t stream);

/* ---- S/model.py:81,133 stem: Conv2d(Cin,Cout,3,padding=1) reading the NCHW input image -------
 * w repacked [3,3,Cin,Cout]; out NHWC rows of out_dtype (bf16 needs Cin <= 4, Cout % 16 == 0).  Cout % 4 == 0. */
int uwir_stem_conv3x3(const float* x_nchw, int B, int Cin, int H, int W, const float* w, const float* bias, int Cout,
                      void* out, int out_dtype, int ldo, uwir_stream_t stream);

/* ---- S/model.py:158-161 final_norm -> head conv3x3 -> clamp(out + input, 0, 1) ----------------
 * x NHWC rows of x_dtype (bf16 needs Cin % 16 == 0) with the final LayerNorm2d folded as a_scale/a_shift [B,Cin]; w
 * repacked [Cout,3,3,Cin]; residual/out NCHW fp32; preclamp (optional) receives out+input before the clamp. */
int uwir_head_conv3x3(const void* x, int x_dtype, int ldx, int B, int H, int W, int Cin, const float* a_scale,
                      const float* a_shift, const float* w, const float* bias, int Cout, const float* residual_nchw,
                      float* out_nchw, float* preclamp_nchw, uwir_stream_t stream);

/* ---- windowed forms for halo tiling (NEW semantics, no reference function -- the
 * reference runs whole frames only; defined in underwater_image_restoration_amd/tiling.py).  "Image b" of the batch the
 * network sees is the T x T window of frame win[b].frame with top-left pixel (y, x); frames are NCHW [F, C, FH, FW].
 * The stem reads its window in place (zero padding at the WINDOW border: each window is an image of its own), the mask
 * resize samples the window of the frame-sized masks, and the head adds the window of the input frame and writes only
 * the pixels the window OWNS -- frame rows [cy0, cy1) x columns [cx0, cx1) -- straight into the output frame: no
 * gather of windows before the network and no stitch copy after it.  `win` is a DEVICE array of nwin descriptors.
 * Descriptor contract (n_frames = F): 0 <= frame < F, 0 <= y <= FH - T, 0 <= x <= FW - T, and the owned core inside the
 * window (y <= cy0 <= cy1 <= y + T, same for x).  The descriptors live in device memory, so the entry points cannot
 * reject a bad one with an error code without a synchronisation; instead every kernel checks each descriptor it uses
 * and skips a window that breaks the contract (nothing is read or written for it).  Build them on the host with the
 * same checks (tiling.TiledRestorer.window_descriptors / ops.check_windows do). */
typedef struct uwir_window { int frame, y, x, cy0, cy1, cx0, cx1, reserved; } uwir_window;
int uwir_stem_conv3x3_win(const float* frames, int n_frames, int Cin, int FH, int FW, const uwir_window* win, int nwin,
                          int T, const float* w, const float* bias, int Cout, void* out, int out_dtype, int ldo,
                          uwir_stream_t stream);
int uwir_bilinear_win(const float* frames, int n_frames, int M, int FH, int FW, const uwir_window* win, int nwin, int T,
                      float* dst, int ldd, int hd, int wd, uwir_stream_t stream);
int uwir_head_conv3x3_win(const void* x, int x_dtype, int ldx, int nwin, int T, int Cin, const float* a_scale, const float* a_shift,
                          const float* w, const float* bias, int Cout, const float* frames_in, float* frames_out,
                          int n_frames, int FH, int FW, const uwir_window* win, uwir_stream_t stream);

/* ---- L/utils/sde_utils.py:44-45,177-187,205-231 fused IR-SDE reverse steps --------------------
 * euler:     x <- x - (theta*(mu-x) - sigma^2*score)*dt - sigma*(eps*sqrt(dt)),  score = -noise/sigma_bar
 * posterior: x0 = (x - mu - sigma_bar*noise)*expA + mu;  x <- term1*(x-mu) + term2*(x0-mu) + mu + std*eps
 * All tensors fp32, numel elements, x updated in place (x_out may alias x).  eps may be NULL
 * (mean step / ODE with half = 0.5). */
int uwir_sde_euler_step(const float* x, const float* mu, const float* noise, const float* eps, float* x_out,
                        size_t numel, float theta, float sigma, float sigma_bar, float dt, float score_weight,
                        uwir_stream_t stream);
int uwir_sde_posterior_step(const float* x, const float* mu, const float* noise, const float* eps, float* x_out,
                            size_t numel, float sigma_bar, float expA, float term1, float term2, float std,
                            uwir_stream_t stream);

/* ---- per-token LayerNorm: open_clip LayerNorm (L/open_clip/transformer.py:22-28, used at :441,:208,:215,:475)
 * and the gain-only channel LayerNorm of the IR-SDE UNet (U/models/modules/module_util.py:77-86) ----------------
 * y[m,:] = LN(x[m,:] + add[m % add_period,:]) * gamma + beta (+ res[m,:]) over the C channels of row m (biased
 * variance, eps inside the sqrt).  add may be NULL (else [add_period, add_ld]: the ViT's positional + class
 * embedding, transformer.py:527-536); beta may be NULL; res (fp32, row stride ldr) may be NULL, else it is the
 * skip of Residual(PreNorm(..)) added after the norm (module_util.py:27-33).  x fp32 or bf16 (x_dtype, row stride
 * ldx; statistics in fp32 either way), y fp32 or bf16 (row stride ldy), y may alias x or res when all are fp32.
 * C % 4 == 0. */
int uwir_ln_rows(const void* x, int x_dtype, size_t ldx, int M, int C, const float* add, int add_period, int add_ld,
                 const float* gamma, const float* beta, float eps, const float* res, size_t ldr, void* y, int y_dtype,
                 size_t ldy, uwir_stream_t stream);

/* ---- ViT patch embedding gather: im2col of Conv2d(3,width,k=P,stride=P), L/open_clip/transformer.py:409,522 ------
 * img NCHW fp32 [B,Cin,H,W] -> out rows [B*(G+1), Cin*P*P] (G = (H/P)*(W/P) patches, k = (c*P+ky)*P+kx, i.e. the
 * layout of conv1.weight.reshape(width,-1)); row 0 of each image is zero (class-token slot, :527-530). */
int uwir_patchify(const float* img_nchw, int B, int Cin, int H, int W, int P, void* out, int out_dtype, size_t ldo,
                  uwir_stream_t stream);

/* ---- `x += control.pop()` with materialised hiddens (L/open_clip/transformer.py:367-368) and other plain
 * residual adds: out[m,:] = a[m,:] + b[m,:]; a fp32, b and out fp32 or bf16 row matrices (b_dtype / out_dtype: the
 * IR-SDE UNet's activation stream is bf16 in bf16 mode), out may alias a (as fp32) or b (same dtype).  C % 4 == 0. */
int uwir_add_rows(const float* a, size_t lda, const void* b, int b_dtype, size_t ldb, void* out, int out_dtype, size_t ldo,
                  int M, int C, uwir_stream_t stream);

/* ======================= IR-SDE score network (U = L/config/daclip-sde/models/modules/) ======================= */

/* ---- U/DenoisingUNet_arch.py:123-127 + check_image_size :111-116 -------------------------------------------------
 * out NHWC fp32 [B,Hp,Wp,Cp] = cat([xt - cond, cond]) reflect-padded right/bottom to Hp x Wp, channels zero-filled
 * from 2*Cin up to Cp (so the 7x7 stem conv runs through uwir_conv_gemm).  xt, cond NCHW fp32 [B,Cin,H,W]. */
int uwir_unet_input(const float* xt_nchw, const float* cond_nchw, int B, int Cin, int H, int W, int Hp, int Wp, int Cp,
                    float* out, uwir_stream_t stream);

/* ---- nn.GroupNorm(32, C, eps=1e-6) of SpatialTransformer (U/attention.py:76-77,236,254) as a folded affine ------
 * x [B,n,C] fp32 -> a[b,c], s[b,c] with GroupNorm(x) == a*x + s, consumed by the next GEMM's a_scale/a_shift.
 * workspace >= uwir_ln2d_workspace_bytes(B,n,C). */
int uwir_groupnorm_affine(const float* x, int ldx, int B, int n, int C, int groups, float eps, const float* gamma,
                          const float* beta, float* a_out, float* s_out, void* workspace, size_t workspace_bytes,
                          uwir_stream_t stream);

/* ---- LinearAttention core, U/module_util.py:170-183 ---------------------------------------------------------------
 * qkv [B*n, ld] = [q | k | v], each heads*32 wide (head-major); out[n, h*32+e] =
 * sum_d ( sum_m softmax_m(k)[m,d] v[m,e] / n ) * softmax_d(q[n,h,:])[d] * 32^-0.5.  dim_head is fixed at 32 like
 * the reference (DenoisingUNet_arch.py:31-32).  workspace >= uwir_linattn_workspace_bytes(B,n,heads). */
size_t uwir_linattn_workspace_bytes(int B, int n, int heads);
int uwir_linear_attention(const void* qkv, int dtype, int ld, int B, int n, int heads, void* out, int out_dtype, int ldo,
                          void* workspace, size_t workspace_bytes, uwir_stream_t stream);

/* ---- the whole Residual(PreNorm(dim, LinearAttention(dim))) block, fused (bf16 MFMA operands, fp32 accumulation) ------
 * U/module_util.py:27-33 (Residual), :77-97 (channel LayerNorm, gain only, PreNorm), :157-185 (LinearAttention incl.
 * to_out = Conv2d(128, dim, 1) + LayerNorm), wired by U/DenoisingUNet_arch.py:81-88:
 *   out[m,:] = LN_{g_out}( W_out . LA( W_qkv . LN_{g_pre}(x)[.] )[m] + b_out ) + x[m,:]
 * Everything after the context reduction is row-local, so the block is two passes over x (+ a tiny combine): pass 1
 * builds ctx[b,h] = softmax_n(k) v^T / n with an online max over the rows, pass 2 does q-softmax, ctx product, to_out,
 * the output LayerNorm and the residual.  x, out [B*n, ld] of io_dtype (fp32, or bf16 = the UNet's bf16 activation
 * stream; statistics and accumulation stay fp32); w_qkv bf16 [3*heads*32, dim] (rows q | k | v),
 * w_out bf16 [dim, heads*32]; heads must be 4, dim 64 / 128 / 256 (the reference's LinearAttention sizes); out may not
 * alias x.  workspace >= uwir_linattn_block_workspace_bytes(B, n), 16-byte aligned.  Same function as the unfused
 * sequence uwir_ln_rows -> uwir_conv_gemm -> uwir_linear_attention -> uwir_conv_gemm -> uwir_ln_rows to bf16 rounding. */
size_t uwir_linattn_block_workspace_bytes(int B, int n);
int uwir_linattn_block(const void* x, int io_dtype, int ldx, int B, int n, int dim, int heads, const float* g_pre,
                       const void* w_qkv, const void* w_out, const float* b_out, const float* g_out, float eps, void* out, int ldo,
                       void* workspace, size_t workspace_bytes, uwir_stream_t stream);

/* ---- GEGLU gate, U/attention.py:37-44: h [M, 2F] = [value | gate] -> out [M, F] = value * gelu(gate) (exact GELU);
 * h and out share dtype. */
int uwir_geglu(const void* h, int dtype, size_t ldh, int M, int F, void* out, size_t ldo, uwir_stream_t stream);

/* ---- prompt weighting softmax(text_mlp(ctx), dim=1) * prompt, U/DenoisingUNet_arch.py:135 -------------------------
 * y[b,:] = softmax(x[b,:]) * mul (mul [C] or NULL). */
int uwir_softmax_rows(const float* x, int ldx, int B, int C, const float* mul, float* y, int ldy, uwir_stream_t stream);

/* ---- final_conv Conv2d(nf, out_nc, 3, 1, 1) + crop x[..., :H, :W], U/DenoisingUNet_arch.py:170-172 ----------------
 * x NHWC fp32 or bf16 (x_dtype; bf16 needs Cin % 16 == 0) [B,H,W,Cin] (padded size) -> out NCHW fp32 [B,Cout,Ho,Wo] (top-left crop), w repacked [Cout,3,3,Cin],
 * Cout <= 4. */
int uwir_tail_conv3x3(const void* x, int x_dtype, int ldx, int B, int H, int W, int Cin, const float* w, const float* bias,
                      int Cout, float* out_nchw, int Ho, int Wo, uwir_stream_t stream);
/* The same conv on fp32 rows with "bf16 x 3" products on the matrix cores (same reference lines): inputs and weights are
 * split hi + lo in the kernel, w_hi x_hi + w_lo x_hi + w_hi x_lo, i.e. each product to ~2^-16 -- the UNet's bf16 mode
 * with the precise tail path (unet.py); uwir_tail_conv3x3 on fp32 rows stays exact fp32.  Cin 32 or 64 takes the MFMA
 * kernel, anything else the exact one. */
int uwir_tail_conv3x3_split(const float* x, int ldx, int B, int H, int W, int Cin, const float* w, const float* bias,
                            int Cout, float* out_nchw, int Ho, int Wo, uwir_stream_t stream);

/* ---- evaluation metrics of the student's eval loops, underwater_ir/student/train_student.py:70-95 ----------------
 * x, y NCHW fp32 [B,C,H,W] -> out [B,3] = per image (ssim_loss, mse, psnr): ssim_loss = mean(1 - ssim_map) with the
 * 11x11 uniform window (avg_pool2d, zero padding counted, variances clamped at 0, eps 1e-8; ssim_value = 1 - that),
 * psnr = 10 log10(1/mse) capped at 100 dB for mse <= 1e-10.  c1 = 0.01^2, c2 = 0.03^2 in the reference.
 * workspace >= uwir_image_metrics_workspace_bytes(B,C,H,W), 8-byte aligned. */
size_t uwir_image_metrics_workspace_bytes(int B, int C, int H, int W);
int uwir_image_metrics(const float* x_nchw, const float* y_nchw, int B, int C, int H, int W, float c1, float c2, float* out,
                       void* workspace, size_t workspace_bytes, uwir_stream_t stream);

/* ---- L/utils/img_utils.py:136-163 tensor2img (3-D / 2-D case) and :182-189 calculate_psnr --------------------------
 * x CHW fp32 -> out HWC uint8 = round_half_even(255 * (clamp(x, lo, hi) - lo) / (hi - lo)), channel order reversed
 * when reverse_channels (RGB -> BGR, the reference's [2, 1, 0] gather); same fp32 operation order, identical bytes.
 * uwir_sqdiff_u8: *out = sum_i (a[i] - b[i])^2 over n bytes (exact); PSNR = 20 log10(255 / sqrt(*out / n)). */
int uwir_tensor2img_u8(const float* x_chw, int C, int H, int W, float lo, float hi, int reverse_channels,
                       uint8_t* out_hwc, uwir_stream_t stream);
int uwir_sqdiff_u8(const uint8_t* a, const uint8_t* b, size_t n, unsigned long long* out, uwir_stream_t stream);

/* ---- teacher heads: tokens x prompt embeddings -> masks / statistics -> degradation code z_d ----------------------
 * underwater_ir/teacher/mask_head.py:96-164 (MaskHead.forward), differentiable_crf.py:44-80 (DifferentiableGuidedCRF,
 * mean-field iterations with a normalised Gaussian and an edge mask from the guidance image) and deg_coder.py:31-59
 * (DegradationCoder: code = [logit(clamp(global_prob, eps, 1-eps)) | avg_intensity], confidence_scale = 1 - mean_k
 * confidence).  One launch per call, one block per image.  Maps are NCHW fp32 [B, K, h, w] like the reference's.
 * tokens: [B*h*w, ld_tokens] rows of C features (token t = y*w + x); text [K, C]; thresholds [K] (device);
 * image (only with use_crf): NCHW fp32 [B, img_c, img_h, img_w], resized bilinearly (align_corners=False) to (h, w)
 * when the sizes differ.  K <= 64, crf_ks odd <= 15, morphology_kernel odd (1 = off).
 * workspace >= uwir_teacher_heads_workspace_bytes(B, h*w, K). */
typedef struct uwir_teacher_args {
    const float* tokens;
    int ld_tokens;
    const float* text;
    const float* thresholds;
    const float* image;
    int B, h, w, C, K;
    int img_c, img_h, img_w;
    float temperature, alpha, confidence_delta, eps;
    int morphology_kernel;
    int use_crf, crf_iters, crf_ks;
    float crf_sigma, crf_edge_weight;
    float *scores, *probs, *masks, *intensity, *raw_probs; /* [B, K, h, w] */
    float *avg_intensity, *global_prob, *confidence;       /* [B, K] */
    float *logits, *code, *confidence_scale;               /* [B, K], [B, 2K], [B] */
    void* workspace;
    size_t workspace_bytes;
} uwir_teacher_args;
size_t uwir_teacher_heads_workspace_bytes(int B, int n, int K);
int uwir_teacher_heads(const uwir_teacher_args* args, uwir_stream_t stream);
/* DifferentiableGuidedCRF.forward(probs, image, confidence=None) on its own: refines probs_in [B, K, h, w] into
 * args->probs (a different buffer); confidence_in [B, K] or NULL (= 1).  Uses only the shape / image / crf_* /
 * probs / workspace fields of args. */
int uwir_guided_crf(const uwir_teacher_args* args, const float* probs_in, const float* confidence_in, uwir_stream_t stream);

/* ---- the RL adapter's enhancement operators and state features that need no OpenCV -------
 * Images are planar CHW fp32 device buffers, as the reference passes them.  (The OpenCV-backed functions of the same
 * files -- histogram_equalization, clahe_equalization, compute_uism / uiconm / uciqe / uiqm -- are not provided.)
 * rl/uiep_cwgr_ops.py:114-116 gamma_correction: out = clamp(x, 0, 1) ** max(gamma, 1e-3). */
int uwir_gamma_correction(const float* x, float* out, size_t n, float gamma, uwir_stream_t stream);
/* rl/uiep_cwgr_ops.py:119-120 lf_hf_mix: out = clamp(alpha * high + (1 - alpha) * low, 0, 1), rounded as the fp32
 * tensor expression rounds (bit-exact). */
int uwir_lf_hf_mix(const float* low, const float* high, double alpha, float* out, size_t n, uwir_stream_t stream);
/* rl/uiep_cwgr_ops.py:46-59 wavelet_denoise on a (C, H, W) image, H and W even: one Haar level, per detail band
 * (all channels together) soft threshold scale * (median|c| / 0.6745)^2 / (sqrt(mean c^2) + 1e-6) with torch.median's
 * lower-middle element found exactly, inverse Haar.  workspace 256-byte aligned. */
size_t uwir_wavelet_denoise_workspace_bytes(int C, int H, int W);
int uwir_wavelet_denoise(const float* x, int C, int H, int W, float threshold_scale, float* out, void* workspace,
                         size_t workspace_bytes, uwir_stream_t stream);
/* rl/uiep_cwgr_ops.py:62-86 guided_filter for ONE channel (H, W) -- the reference's (1, 1, k, k) box kernel at groups=1
 * raises on anything else -- zero-padded (2 radius + 1)^2 box means, radius <= 16. */
size_t uwir_guided_filter_workspace_bytes(int H, int W);
int uwir_guided_filter(const float* input, const float* guidance, int H, int W, int radius, float eps, float* out,
                       void* workspace, size_t workspace_bytes, uwir_stream_t stream);
/* rl/uiep_cwgr_ops.py:89-111 multi_scale_retinex on (C, H, W); sigmas is a HOST array of nsig values.  The reference's
 * 2-D normalised Gaussian of size 2 ceil(3 sigma) + 1 is applied as its two exact 1-D factors, zero padded. */
size_t uwir_multi_scale_retinex_workspace_bytes(int C, int H, int W);
int uwir_multi_scale_retinex(const float* x, int C, int H, int W, const double* sigmas, int nsig, float* out,
                             void* workspace, size_t workspace_bytes, uwir_stream_t stream);
/* ---- the OpenCV-backed operators and scores of the RL adapter (PARITY UNPINNED: cv2 is absent from the build container, so
 * no vector of the reference could be generated; csrc/enhance_cv.hip restates OpenCV's documented arithmetic, the tests
 * compare against an independent numpy restatement in oracle/enhance_ref.py).  One workspace size serves all three
 * (tiles = the CLAHE grid, 8 x 8 for the other two), 256-byte aligned.
 * rl/uiep_cwgr_ops.py:24-35 histogram_equalization: (C, H, W) fp32 image, C <= 3 -> equalised image (uint8 levels / 255).
 * rl/uiep_cwgr_ops.py:38-44 clahe_equalization: (3, H, W) RGB -> 8-bit Lab -> CLAHE(clip_limit, tiles) on L -> RGB.
 * rl/uiqm_uciqe.py:24-65: out3 (DEVICE doubles) = compute_uism, compute_uiconm, compute_uciqe of a (3, H, W) image. */
size_t uwir_cv_workspace_bytes(int H, int W, int tiles_y, int tiles_x);
int uwir_histogram_equalization(const float* img, int C, int H, int W, float* out, void* workspace, size_t workspace_bytes,
                                uwir_stream_t stream);
int uwir_clahe_equalization(const float* img, int H, int W, float clip_limit, int tiles_y, int tiles_x, float* out,
                            void* workspace, size_t workspace_bytes, uwir_stream_t stream);
int uwir_cv_scores(const float* img, int H, int W, double* out3, void* workspace, size_t workspace_bytes, uwir_stream_t stream);
/* State features of rl/infer_adapt.py:62-68 build_state on a (C, H, W) image, written to 4 DEVICE doubles:
 * out4[0] = compute_uicm (rl/uiqm_uciqe.py:13-21; on the uint8-quantised image, C must be 3, else NaN),
 * out4[1], out4[2] = mean and unbiased std over every element, out4[3] = compute_sharpness (:52-59). */
size_t uwir_enhance_state_workspace_bytes(void);
int uwir_enhance_state(const float* img, int C, int H, int W, double* out4, void* workspace, size_t workspace_bytes,
                       uwir_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UWIR_H */

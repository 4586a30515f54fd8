// Evaluation metrics of the student's eval loops on the device (the step AFTER the forward):
//   underwater_ir/student/train_student.py:70-84  ssim_loss  (11x11 UNIFORM window = avg_pool2d, zero padding counted)
//   underwater_ir/student/train_student.py:91-95  psnr_value (10 log10(1 / mse), 100 dB cap)
// One pass over x and y (NCHW fp32): a block owns a 32x32 pixel tile of one (b, c) plane, stages the 42x42 halo of
// both images in LDS, forms the five box sums separably in fp64 (sigma = E[x^2] - mu^2 cancels catastrophically in
// fp32 on flat regions), and emits per-block partial sums of (1 - ssim) and (x - y)^2; a second kernel adds the
// partials of each image in a fixed order (deterministic).  HBM-bound: 8 B per pixel.
#include "common.h"

#define MT 32          // tile edge
#define MR 5           // window radius (11x11)
#define MH (MT + 2 * MR)

__global__ void __launch_bounds__(256) ssim_mse_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                               int H, int W, int tiles_x, int tiles_y, float c1,
                                                               float c2, double* __restrict__ part) {
    __shared__ float sx[MH][MH + 1], sy[MH][MH + 1];
    __shared__ double hs[5][MH][MT];   // horizontal 11-sums of x, y, xx, yy, xy at (halo row, tile col)
    __shared__ double red[2][4];
    const int tid = threadIdx.x;
    const int plane = blockIdx.z, ty0 = blockIdx.y * MT, tx0 = blockIdx.x * MT;
    const float* xp = x + (size_t)plane * H * W;
    const float* yp = y + (size_t)plane * H * W;
    for (int i = tid; i < MH * MH; i += 256) {
        int ly = i / MH, lx = i % MH;
        int gy = ty0 + ly - MR, gx = tx0 + lx - MR;
        bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;   // avg_pool2d pads with zeros and divides by 121 regardless
        sx[ly][lx] = in ? xp[(size_t)gy * W + gx] : 0.f;
        sy[ly][lx] = in ? yp[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < MH * MT; i += 256) {
        int ly = i / MT, cx = i % MT;
        double a = 0, b = 0, aa = 0, bb = 0, ab = 0;
#pragma unroll
        for (int k = 0; k < 2 * MR + 1; ++k) {
            double u = sx[ly][cx + k], v = sy[ly][cx + k];
            a += u; b += v; aa += u * u; bb += v * v; ab += u * v;
        }
        hs[0][ly][cx] = a; hs[1][ly][cx] = b; hs[2][ly][cx] = aa; hs[3][ly][cx] = bb; hs[4][ly][cx] = ab;
    }
    __syncthreads();
    double acc_s = 0, acc_e = 0;
    const double inv = 1.0 / ((2 * MR + 1) * (2 * MR + 1));
    for (int i = tid; i < MT * MT; i += 256) {
        int cy = i / MT, cx = i % MT;
        int gy = ty0 + cy, gx = tx0 + cx;
        if (gy >= H || gx >= W) continue;
        double a = 0, b = 0, aa = 0, bb = 0, ab = 0;
#pragma unroll
        for (int k = 0; k < 2 * MR + 1; ++k) {
            a += hs[0][cy + k][cx]; b += hs[1][cy + k][cx]; aa += hs[2][cy + k][cx]; bb += hs[3][cy + k][cx];
            ab += hs[4][cy + k][cx];
        }
        double mx = a * inv, my = b * inv;
        double vx = aa * inv - mx * mx, vy = bb * inv - my * my, cxy = ab * inv - mx * my;
        vx = vx < 0 ? 0 : vx;                       // train_student.py:78-79 clamp(min=0)
        vy = vy < 0 ? 0 : vy;
        double num = (2 * mx * my + (double)c1) * (2 * cxy + (double)c2);
        double den = (mx * mx + my * my + (double)c1) * (vx + vy + (double)c2);
        acc_s += 1.0 - num / (den + 1e-8);
        double d = (double)sx[cy + MR][cx + MR] - (double)sy[cy + MR][cx + MR];
        acc_e += d * d;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        acc_s += __shfl_xor(acc_s, off);
        acc_e += __shfl_xor(acc_e, off);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = acc_s; red[1][tid >> 6] = acc_e; }
    __syncthreads();
    if (tid == 0) {
        size_t o = ((size_t)plane * tiles_y + blockIdx.y) * tiles_x + blockIdx.x;
        part[2 * o] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        part[2 * o + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ void __launch_bounds__(64) ssim_mse_finalize_kernel(const double* __restrict__ part, int per_image,
                                                               double inv_count, int B, float* __restrict__ out) {
    // one wave per image; lanes stride the image's partials, then a fixed-order tree
    int b = blockIdx.x, lane = threadIdx.x;
    double s = 0, e = 0;
    for (int i = lane; i < per_image; i += 64) {
        s += part[2 * ((size_t)b * per_image + i)];
        e += part[2 * ((size_t)b * per_image + i) + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        e += __shfl_xor(e, off);
    }
    if (lane == 0) {
        double mse = e * inv_count;
        out[3 * b] = (float)(s * inv_count);                               // ssim_loss of image b
        out[3 * b + 1] = (float)mse;
        out[3 * b + 2] = mse <= 1e-10 ? 100.f : (float)(10.0 * log10(1.0 / mse));   // psnr_value of image b
    }
}

extern "C" size_t uwir_image_metrics_workspace_bytes(int B, int C, int H, int W) {
    return (size_t)B * C * ceil_div(H, MT) * ceil_div(W, MT) * 2 * sizeof(double);
}

extern "C" int uwir_image_metrics(const float* x_nchw, const float* y_nchw, int B, int C, int H, int W, float c1, float c2,
                                  float* out, void* workspace, size_t workspace_bytes, uwir_stream_t stream) {
    UWIR_CHECK(x_nchw && y_nchw && out && workspace, "image_metrics: null pointer");
    UWIR_CHECK(B > 0 && C > 0 && H > 0 && W > 0 && (long long)B * C <= 65535, "image_metrics: bad shape");
    UWIR_CHECK(workspace_bytes >= uwir_image_metrics_workspace_bytes(B, C, H, W) && ((uintptr_t)workspace & 7) == 0,
               "image_metrics: workspace too small / misaligned");
    int tx = ceil_div(W, MT), ty = ceil_div(H, MT);
    hipLaunchKernelGGL(ssim_mse_partial_kernel, dim3(tx, ty, B * C), dim3(256), 0, (hipStream_t)stream, x_nchw, y_nchw, H,
                       W, tx, ty, c1, c2, (double*)workspace);
    UWIR_LAUNCH_CHECK("image_metrics(partial)");
    hipLaunchKernelGGL(ssim_mse_finalize_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, (const double*)workspace,
                       C * tx * ty, 1.0 / ((double)C * H * W), B, out);
    UWIR_LAUNCH_CHECK("image_metrics(finalize)");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// L/utils/img_utils.py:136-163 tensor2img: clamp to [lo, hi], rescale to [0, 1], x 255, round half to even (numpy
// .round()), uint8, CHW -> HWC with the channel order reversed (RGB -> BGR).  Same fp32 operation order as the
// reference, so the bytes are identical.  4 bytes in, 1 byte out per element: the D2H copy of a result shrinks 4x.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) tensor2img_kernel(const float* __restrict__ x, int C, int H, int W, float lo,
                                                         float hi, int reverse, uint8_t* __restrict__ out) {
    const size_t total = (size_t)C * H * W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);           // output is HWC: i = (y * W + x) * C + c
        const size_t pix = i / C;
        const int cs = reverse ? C - 1 - c : c;
        float v = x[(size_t)cs * H * W + pix];
        v = fminf(fmaxf(v, lo), hi);
        v = (v - lo) / (hi - lo);
        out[i] = (uint8_t)rintf(v * 255.0f);
    }
}

extern "C" int uwir_tensor2img_u8(const float* x_chw, int C, int H, int W, float lo, float hi, int reverse_channels,
                                  uint8_t* out_hwc, uwir_stream_t stream) {
    UWIR_CHECK(x_chw && out_hwc, "tensor2img: null pointer");
    UWIR_CHECK(C > 0 && H > 0 && W > 0 && hi > lo, "tensor2img: bad shape / range");
    const size_t total = (size_t)C * H * W;
    size_t g = ceil_div_sz(total, 256);
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(tensor2img_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x_chw, C, H, W, lo, hi,
                       reverse_channels, out_hwc);
    UWIR_LAUNCH_CHECK("tensor2img");
    return UWIR_OK;
}

// ------------------------------------------------------------------------------------------
// L/utils/img_utils.py:182-189 calculate_psnr on [0, 255] images: the sum of squared uint8 differences, exact in
// integers (integer atomics commute: bit-reproducible); the caller forms 20 log10(255 / sqrt(sum / n)).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sqdiff_u8_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                        size_t n, unsigned long long* __restrict__ out) {
    unsigned long long s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)a[i] - (int)b[i];
        s += (unsigned)(d * d);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

extern "C" int uwir_sqdiff_u8(const uint8_t* a, const uint8_t* b, size_t n, unsigned long long* out, uwir_stream_t stream) {
    UWIR_CHECK(a && b && out && n > 0, "sqdiff_u8: null pointer / empty input");
    UWIR_CHECK(((uintptr_t)out & 7) == 0, "sqdiff_u8: out must be 8-byte aligned");
    if (hipMemsetAsync(out, 0, sizeof(unsigned long long), (hipStream_t)stream) != hipSuccess)
        return uwir_set_error(UWIR_ELAUNCH, "sqdiff_u8: hipMemsetAsync failed");
    size_t g = ceil_div_sz(n, 256 * 16);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(sqdiff_u8_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a, b, n, out);
    UWIR_LAUNCH_CHECK("sqdiff_u8");
    return UWIR_OK;
}

// Build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/store_pattern tools/exp/store_pattern.hip ; run on the GPU box.
// experiment: what do the GEMM epilogue's output stores cost by pattern?  (not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// block = 256 threads writes a tile of 128 rows x seg bytes at row stride `stride`; mode 0 plain, 1 nontemporal
template <int NT>
__global__ void __launch_bounds__(256) tile_store(char* out, int tiles_n, int seg, size_t stride, int xcd_remap, int nblocks) {
    int bid = blockIdx.x;
    if (xcd_remap) { int q = nblocks / 8, r = nblocks % 8, x = bid % 8; bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8; }
    const int mt = bid / tiles_n, nt = bid % tiles_n;
    const int per_row = seg / 16;                 // 16-byte pieces per row segment
    f4 v = {1.f, 2.f, 3.f, (float)bid};
    for (int i = threadIdx.x; i < 128 * per_row; i += 256) {
        const int r = i / per_row, c = i % per_row;
        f4* p = (f4*)(out + ((size_t)mt * 128 + r) * stride + (size_t)nt * seg + c * 16);
        if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
}
// the lane-owned pattern of the old epilogue: wave (wm, wn) owns 64 rows x 128 B; lane (fr, fq): row fr, 32 B at fq*32
__global__ void __launch_bounds__(256) lane_store(char* out, int tiles_n, size_t stride, int nblocks) {
    int bid = blockIdx.x;
    { int q = nblocks / 8, r = nblocks % 8, x = bid % 8; bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8; }
    const int mt = bid / tiles_n, nt = bid % tiles_n;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    f4 v = {1.f, 2.f, 3.f, (float)bid};
    for (int i = 0; i < 4; ++i)
        for (int jp = 0; jp < 2; ++jp) {
            const int r = (wave >> 1) * 64 + i * 16 + fr;
            f4* p = (f4*)(out + ((size_t)mt * 128 + r) * stride + (size_t)nt * 256 + (wave & 1) * 128 + fq * 32 + jp * 16);
            *p = v;
        }
}
int main() {
    const int M = 262144;
    char* out; CK(hipMalloc(&out, (size_t)M * 1152 + 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch, double bytes) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        printf("%-46s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
    };
    const int mt = M / 128;
    double by = (double)M * 1152;
    timeit("memset", [&] { CK(hipMemsetAsync(out, 1, (size_t)M * 1152)); }, by);
    timeit("rows 1152 B whole (1 block / 128 rows)", [&] { tile_store<0><<<mt, 256>>>(out, 1, 1152, 1152, 0, mt); }, by);
    timeit("256-B segments x4.5, n fastest, no remap", [&] { tile_store<0><<<mt * 4, 256>>>(out, 4, 256, 1152, 0, mt * 4); }, by * 8 / 9);
    timeit("256-B segments, xcd remap", [&] { tile_store<0><<<mt * 4, 256>>>(out, 4, 256, 1152, 1, mt * 4); }, by * 8 / 9);
    timeit("256-B segments, xcd remap, nontemporal", [&] { tile_store<1><<<mt * 4, 256>>>(out, 4, 256, 1152, 1, mt * 4); }, by * 8 / 9);
    timeit("256-B segments, lane-owned 16-B pieces", [&] { lane_store<<<mt * 4, 256>>>(out, 4, 1152, mt * 4); }, by * 8 / 9);
    timeit("128 rows x 256 B contiguous tiles (N = 128)", [&] { tile_store<0><<<mt * 4, 256>>>(out, 1, 256, 256, 1, mt * 4); }, by * 8 / 9);
    timeit("576-B segments x2 (BN = 288)", [&] { tile_store<0><<<mt * 2, 256>>>(out, 2, 576, 1152, 1, mt * 2); }, by);
    timeit("rows whole, nontemporal", [&] { tile_store<1><<<mt, 256>>>(out, 1, 1152, 1152, 0, mt); }, by);
    return 0;
}

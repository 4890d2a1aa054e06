// The round-1 corr81 kernel (one thread per (displacement, pixel), no LDS), kept ONLY as the baseline of tools/pwc_bench.py.
#include <hip/hip_runtime.h>
__global__ void corr81_naive_kernel(const float* __restrict__ f1, const float* __restrict__ f2, float* __restrict__ out,
                                    int C, int H, int W) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    const int b = blockIdx.z / 81, d = blockIdx.z % 81;
    if (x >= W) return;
    const int dy = d / 9 - 4, dx = d % 9 - 4;
    const int y2 = y + dy, x2 = x + dx;
    const long HW = (long)H * W;
    float s = 0.f;
    if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) {
        const float* a = f1 + (long)b * C * HW + (long)y * W + x;
        const float* v = f2 + (long)b * C * HW + (long)y2 * W + x2;
        for (int c = 0; c < C; ++c) s = fmaf(a[(long)c * HW], v[(long)c * HW], s);
    }
    out[((long)b * 81 + d) * HW + (long)y * W + x] = s / (float)C;
}
extern "C" int corr81_naive(const float* f1, const float* f2, float* out, int B, int C, int H, int W, void* stream) {
    dim3 grid((W + 63) / 64, H, B * 81);
    corr81_naive_kernel<<<grid, 64, 0, (hipStream_t)stream>>>(f1, f2, out, C, H, W);
    return (int)hipGetLastError();
}

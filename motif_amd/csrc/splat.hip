// Fused reliability-aware soft-splat (forward warping) for gfx950.
//
// One thread per SOURCE pixel: the target corner indices and bilinear weights are computed once
// (the reference recomputes flow+floor for every one of its 131 channels), then the channel loop
// scatters with global_atomic_add_f32 (built with -munsafe-fp-atomics).  A wave's 64 lanes are 64
// consecutive x of one source row, so for smooth flow every per-channel atomic instruction lands on
// 1-2 contiguous cache lines of the accumulator plane.  Sum, max and count share the index math.
//
// Bit-level contract with the kernel text (softsplat_cp.py:27-38 etc.): floor -> int corners, weights
// as (SE - o) products, bounds test >=0 & <size, addend = (value*e^z) rounded, then * weight rounded.
#include "common.h"

struct Corners {
    int x0, y0;
    float wnw, wne, wsw, wse;
};

__device__ __forceinline__ Corners corners_of(int X, int Y, float fx, float fy) {
    Corners c;
    const float ox = (float)X + fx, oy = (float)Y + fy;
    c.x0 = (int)floorf(ox);
    c.y0 = (int)floorf(oy);
    const float xe = (float)(c.x0 + 1), ye = (float)(c.y0 + 1), xw = (float)c.x0, yn = (float)c.y0;
    c.wnw = (xe - ox) * (ye - oy);
    c.wne = (ox - xw) * (ye - oy);
    c.wsw = (xe - ox) * (oy - yn);
    c.wse = (ox - xw) * (oy - yn);
    return c;
}

__device__ __forceinline__ void atomic_max_float(float* addr, float value) {
    if (value >= 0) atomicMax((int*)addr, __float_as_int(value));
    else atomicMin((unsigned int*)addr, __float_as_uint(value));
}

struct Scatter {
    long onw, one, osw, ose;   // plane offsets of the four targets
    bool vnw, vne, vsw, vse;
    float wnw, wne, wsw, wse;
    __device__ __forceinline__ void init(const Corners& c, int H, int W) {
        const bool xl = c.x0 >= 0 && c.x0 < W, xr = c.x0 + 1 >= 0 && c.x0 + 1 < W;
        const bool yt = c.y0 >= 0 && c.y0 < H, yb = c.y0 + 1 >= 0 && c.y0 + 1 < H;
        vnw = xl && yt; vne = xr && yt; vsw = xl && yb; vse = xr && yb;
        onw = (long)c.y0 * W + c.x0; one = onw + 1; osw = onw + W; ose = osw + 1;
        wnw = c.wnw; wne = c.wne; wsw = c.wsw; wse = c.wse;
    }
    __device__ __forceinline__ void add(float* plane, float v) const {
        if (vnw) atomicAdd(plane + onw, v * wnw);
        if (vne) atomicAdd(plane + one, v * wne);
        if (vsw) atomicAdd(plane + osw, v * wsw);
        if (vse) atomicAdd(plane + ose, v * wse);
    }
    __device__ __forceinline__ void max(float* plane, float v) const {
        if (vnw) atomic_max_float(plane + onw, v * wnw);
        if (vne) atomic_max_float(plane + one, v * wne);
        if (vsw) atomic_max_float(plane + osw, v * wsw);
        if (vse) atomic_max_float(plane + ose, v * wse);
    }
    __device__ __forceinline__ void count(float* plane, float v) const {
        if (vnw) atomicAdd(plane + onw, v);
        if (vne) atomicAdd(plane + one, v);
        if (vsw) atomicAdd(plane + osw, v);
        if (vse) atomicAdd(plane + ose, v);
    }
};

// ---------------------------------------------------------------- operator form (module surface)
__global__ __launch_bounds__(256) void splat_plain_kernel(const float* src, const float* flow, const float* z,
                                                         float* out_sum, float* out_norm, float* out_max, float* out_cnt,
                                                         int C, int H, int W, int tiles_x) {
    const int n = blockIdx.z;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int X = tx * 64 + (threadIdx.x & 63), Y = ty * 4 + (threadIdx.x >> 6);
    if (X >= W || Y >= H) return;
    const long Q = (long)H * W, p = (long)Y * W + X;
    const float fx = flow[((long)n * 2 + 0) * Q + p], fy = flow[((long)n * 2 + 1) * Q + p];
    Scatter s;
    s.init(corners_of(X, Y, fx, fy), H, W);
    float e = 1.f;
    if (z) e = expf(z[(long)n * Q + p]);
    if (out_sum)
        for (int c = 0; c < C; ++c) {
            float v = src[((long)n * C + c) * Q + p];
            if (z) v = v * e;
            s.add(out_sum + ((long)n * C + c) * Q, v);
        }
    if (out_norm) s.add(out_norm + (long)n * Q, e);
    if (out_max) s.max(out_max + (long)n * Q, z ? e : src[(long)n * C * Q + p]);
    if (out_cnt) s.count(out_cnt + (long)n * Q, 1.0f);
}

extern "C" int motif_splat_fwd(const float* src, const float* flow, const float* z,
                               float* out_sum, float* out_norm, float* out_max, float* out_cnt,
                               int N, int C, int H, int W, void* stream) {
    if (!flow || N < 1 || C < 1 || H < 1 || W < 1) return MOTIF_EINVAL;
    if ((out_sum || (!z && out_max)) && !src) return MOTIF_EINVAL;
    const int tiles_x = (W + 63) / 64, tiles_y = (H + 3) / 4;
    dim3 grid(tiles_x * tiles_y, 1, N);
    splat_plain_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, flow, z, out_sum, out_norm, out_max, out_cnt, C, H, W, tiles_x);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// ---------------------------------------------------------------- fused MoTIF form
// grid.z = image (d,b,n); grid.y = channel slice; block = 4 rows x 64 columns of source pixels.
// Channel slices: the 130 feature planes are split over grid.y so that more waves are in flight per
// accumulator plane region; slice 0 additionally handles norm / max / count.
#define SPLAT_SLICES 5   /* 130 = 5 * 26 */
__global__ __launch_bounds__(256) void splat_motif_kernel(const float* __restrict__ imnet_out, const float* __restrict__ pred,
                                                         const float* __restrict__ feat_lr, const int32_t* __restrict__ iy,
                                                         const int32_t* __restrict__ ix, const float* __restrict__ alpha,
                                                         float s20, float sr, float* acc,
                                                         int B, int N, int H, int W, int HH, int WW, int tiles_x) {
    const int img = blockIdx.z;                 // (d*B + b)*N + n
    const int n = img % N, db = img / N, b = db % B;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int X = tx * 64 + (threadIdx.x & 63), Y = ty * 4 + (threadIdx.x >> 6);
    if (X >= WW || Y >= HH) return;
    const long Q = (long)HH * WW, p = (long)Y * WW + X;
    const float p0 = pred[((long)img * 3 + 0) * Q + p];
    const float p1 = pred[((long)img * 3 + 1) * Q + p];
    const float p2 = pred[((long)img * 3 + 2) * Q + p];
    const float fx = (p0 * s20) * sr, fy = (p1 * s20) * sr;          // Ours.py:794
    const float zz = (p2 > 0.f ? p2 : 0.f) * alpha[0];
    const float e = expf(zz);
    Scatter s;
    s.init(corners_of(X, Y, fx, fy), HH, WW);
    float* abase = acc + (long)(b * N + n) * 133 * Q;
    const int slice = blockIdx.y;
    const int c_lo = slice * (130 / SPLAT_SLICES), c_hi = c_lo + (130 / SPLAT_SLICES);
    const long lr = (long)iy[Y] * W + ix[X];
    const long HWl = (long)H * W;
    for (int c = c_lo; c < c_hi; ++c) {
        float v;
        if (c < 64) v = imnet_out[((long)db * 64 + c) * Q + p];
        else if (c == 64) v = p0;
        else if (c == 65) v = p1;
        else v = feat_lr[((long)db * 64 + (c - 66)) * HWl + lr];
        s.add(abase + (long)c * Q, v * e);
    }
    if (slice == 0) {
        s.add(abase + 130L * Q, e);
        s.max(abase + 131L * Q, e);
        s.count(abase + 132L * Q, 1.0f);
    }
}

extern "C" int motif_splat_motif_fwd(const float* imnet_out, const float* pred, const float* feat_lr,
                                     const int32_t* iy, const int32_t* ix, const float* alpha, float flow_scale,
                                     float* acc, int B, int N, int H, int W, int HH, int WW, void* stream) {
    if (!imnet_out || !pred || !feat_lr || !iy || !ix || !alpha || !acc) return MOTIF_EINVAL;
    if (B < 1 || N < 1 || H < 1 || W < 1 || HH < 1 || WW < 1) return MOTIF_EINVAL;
    const int tiles_x = (WW + 63) / 64, tiles_y = (HH + 3) / 4;
    dim3 grid(tiles_x * tiles_y, SPLAT_SLICES, 2 * B * N);
    splat_motif_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(imnet_out, pred, feat_lr, iy, ix, alpha, 20.0f, flow_scale,
                                                              acc, B, N, H, W, HH, WW, tiles_x);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

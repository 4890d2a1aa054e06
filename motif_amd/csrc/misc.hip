// Resampling, normalisation and gate kernels of the MoTIF path (gfx950).  All are HBM/L2-bound
// streaming kernels: one element per thread, lanes along x so every access is coalesced.
#include "common.h"
#include <stdint.h>

// ------------------------------------------------------------------ F.interpolate(bilinear)
__device__ __forceinline__ float resize_sample(const float* __restrict__ p, int H, int W, int oy, int ox, float sh, float sw, int align) {
    float sy, sx;
    if (align) { sy = sh * oy; sx = sw * ox; }
    else {
        sy = sh * (oy + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
        sx = sw * (ox + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
    }
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
    return hy * (hx * p[(long)y0 * W + x0] + lx * p[(long)y0 * W + x1]) +
           ly * (hx * p[(long)y1 * W + x0] + lx * p[(long)y1 * W + x1]);
}

// one thread = VEC consecutive outputs of a row (16-byte stores when VEC = 4); threads are dealt over the flattened
// (row, column group) index of a plane, so no lane idles on a ragged row width; blockIdx.y = plane.
// UP2: exactly x2 up-sampling with align_corners = False (the PCD pyramid's 8x64-plane maps: 0.65 ms per clip).  The four outputs
// 4k..4k+3 of a row read input columns 2k-1..2k+2 only: 8 loads (clamped, issued together) instead of 16, the same
// interpolation expression on the same values.
// post = 1: the value is RAFT's input normalisation of the resized frame, 2 * ((v * 255) / 255) - 1 with the roundings of the four torch
// operations the reference spends on it (Ours.py:544 `* 255`, raft.py:90-91 `2 * (image / 255.0) - 1.0`; IEEE division, and 2 t - 1
// contracted to one FMA is the same number because 2 t is exact) -- one pass instead of the resize plus four element-wise kernels.
__device__ __forceinline__ float resize_fin(float v, float mul, int post) {
    v *= mul;
    if (post == 1) { v = v * 255.f; v = v / 255.f; v = 2.f * v - 1.f; }
    return v;
}

template <int VEC, bool UP2>
__global__ __launch_bounds__(256) MOTIF_SCALAR_F32 void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                               int Ho, int Wo, float sh, float sw, int align, float mul, int post) {
    const int gpr = Wo / VEC + (Wo % VEC ? 1 : 0);                    // column groups per row
    const int gi = blockIdx.x * 256 + threadIdx.x;
    if (gi >= Ho * gpr) return;
    const int oy = gi / gpr, ox = (gi - oy * gpr) * VEC;
    const long nc = blockIdx.y;
    const float* p = in + nc * (long)H * W;
    float* o = out + nc * (long)Ho * Wo + (long)oy * Wo + ox;
    if constexpr (UP2) {
        float sy = sh * (oy + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
        const int y0 = (int)sy, y1 = y0 + (y0 < H - 1 ? 1 : 0);
        const float ly = sy - y0, hy = 1.f - ly;
        const int xb = ox / 2 - 1;
        float r0[4], r1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int xc = min(max(xb + i, 0), W - 1);
            r0[i] = p[(long)y0 * W + xc];
            r1[i] = p[(long)y1 * W + xc];
        }
        f32x4 v;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float sx = sw * (ox + u + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
            const int x0 = (int)sx;
            const float lx = sx - x0, hx = 1.f - lx;
            constexpr int i0[4] = {0, 1, 1, 2};                       // x0 - xb of outputs 4k..4k+3 (the left edge clamps to the same value)
            const float a = r0[i0[u]], b = r0[i0[u] + 1], c = r1[i0[u]], d = r1[i0[u] + 1];
            v[u] = resize_fin(hy * (hx * a + lx * b) + ly * (hx * c + lx * d), mul, post);
        }
        *(f32x4*)o = v;
    } else if constexpr (VEC == 4) {
        f32x4 v;
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = resize_fin(resize_sample(p, H, W, oy, ox + u, sh, sw, align), mul, post);
        *(f32x4*)o = v;
    } else {
        *o = resize_fin(resize_sample(p, H, W, oy, ox, sh, sw, align), mul, post);
    }
}

// x2 upsampling (align_corners = false), the wide form (round 6): a thread makes 8 output columns of the TWO output rows 2k, 2k+1 from the three
// input rows they share -- per row one 16-byte load (columns 4j .. 4j+3) and its two neighbours, 9 loads for 16 outputs where the form above
// issues 32 (PCD's 64-channel pyramids, 8 images: 118 MB written per launch at 1.7 TB/s before).  Every output is the SAME expression of the
// same operands as in resize_bilinear_kernel<4, true> (row / column indices, weights and clamps are formed the same way), so the bits are equal;
// option resize_narrow = 1 keeps the form above (the A/B of tests/test_kernels_gpu.py).  W % 4 == 0, H >= 2, 16-byte aligned planes.
__global__ __launch_bounds__(256) MOTIF_SCALAR_F32 void resize_up2_wide_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                                                 float sh, float sw, float mul, int post) {
    const int W8 = W >> 2;                                            // threads per row pair: Wo / 8
    const int gi = blockIdx.x * 256 + threadIdx.x;
    if (gi >= H * W8) return;
    const int k = gi / W8, j = gi - k * W8;
    const long nc = blockIdx.y;
    const float* p = in + nc * (long)H * W;
    const int Wo = 2 * W;
    // the two output rows' source rows and weights, exactly as the per-row form computes them
    int y0[2], y1[2];
    float ly[2], hy[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        float sy = sh * ((2 * k + r) + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
        y0[r] = (int)sy; y1[r] = y0[r] + (y0[r] < H - 1 ? 1 : 0);
        ly[r] = sy - y0[r]; hy[r] = 1.f - ly[r];
    }
    // rows held: R0 = y0[0], R1 = y1[0], R2 = y1[1]; y0[1] is R0 (k = 0) or R1
    const int rows[3] = {y0[0], y1[0], y1[1]};
    float c[3][6];                                                    // columns 4j-1 .. 4j+4, clamped to the row like the narrow form's window
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float* rp = p + (long)rows[r] * W + 4 * j;
        const f32x4 m = *(const f32x4*)rp;
        c[r][0] = rp[j > 0 ? -1 : 0];
        c[r][1] = m[0]; c[r][2] = m[1]; c[r][3] = m[2]; c[r][4] = m[3];
        c[r][5] = rp[4 * j + 4 < W ? 4 : 3];
    }
    const bool second_from_r1 = y0[1] == y1[0];                       // (false only for k = 0 and for H - 1 == y0: then y0[1] == y0[0])
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        float top[6], bot[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            top[i] = r == 0 ? c[0][i] : (second_from_r1 ? c[1][i] : c[0][i]);
            bot[i] = r == 0 ? c[1][i] : c[2][i];
        }
        float* o = out + nc * (long)(2 * H) * Wo + (long)(2 * k + r) * Wo + 8 * j;
#pragma unroll
        for (int g = 0; g < 2; ++g) {                                 // the narrow form's window of output group g: columns 4j + 2g - 1 .. + 2
            f32x4 v;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float sx = sw * ((8 * j + 4 * g + u) + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
                const int x0 = (int)sx;
                const float lx = sx - x0, hx = 1.f - lx;
                constexpr int i0[4] = {0, 1, 1, 2};
                const int i = 2 * g + i0[u];
                const float a = top[i], b = top[i + 1], cc = bot[i], d = bot[i + 1];
                v[u] = resize_fin(hy[r] * (hx * a + lx * b) + ly[r] * (hx * cc + lx * d), mul, post);
            }
            *(f32x4*)(o + 4 * g) = v;
        }
    }
}

extern "C" int motif_resize_bilinear(const float* in, float* out, int NC, int H, int W, int Ho, int Wo,
                                     int align_corners, float mul, int post, void* stream) {
    if (!in || !out || NC < 1 || H < 1 || W < 1 || Ho < 1 || Wo < 1 || post < 0 || post > 1) return MOTIF_EINVAL;
    if (NC > 65535 || (long)Ho * Wo >= (1L << 31)) return MOTIF_ELIMIT;
    float sh, sw;
    if (align_corners) { sh = Ho > 1 ? (float)(H - 1) / (Ho - 1) : 0.f; sw = Wo > 1 ? (float)(W - 1) / (Wo - 1) : 0.f; }
    else { sh = (float)H / Ho; sw = (float)W / Wo; }
    hipStream_t s = (hipStream_t)stream;
    if (!align_corners && Ho == 2 * H && Wo == 2 * W && H >= 2 && (W & 3) == 0 && (((uintptr_t)out | (uintptr_t)in) & 15) == 0 &&
        !motif_opt(MOTIF_OPT_RESIZE_NARROW)) {
        dim3 grid(cdiv((long)H * (W / 4), 256), NC);
        resize_up2_wide_kernel<<<grid, 256, 0, s>>>(in, out, H, W, sh, sw, mul, post);
    } else if (Wo % 4 == 0 && ((uintptr_t)out & 15) == 0) {
        dim3 grid(cdiv((long)Ho * (Wo / 4), 256), NC);
        if (!align_corners && Ho == 2 * H && Wo == 2 * W && W >= 2)
            resize_bilinear_kernel<4, true><<<grid, 256, 0, s>>>(in, out, H, W, Ho, Wo, sh, sw, align_corners, mul, post);
        else
            resize_bilinear_kernel<4, false><<<grid, 256, 0, s>>>(in, out, H, W, Ho, Wo, sh, sw, align_corners, mul, post);
    } else {
        dim3 grid(cdiv((long)Ho * Wo, 256), NC);
        resize_bilinear_kernel<1, false><<<grid, 256, 0, s>>>(in, out, H, W, Ho, Wo, sh, sw, align_corners, mul, post);
    }
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// ------------------------------------------------------------------ grid_sample helpers
// bilinear sample of one plane at unnormalised (ix, iy), corners outside the plane contribute zero
__device__ __forceinline__ float bilinear_zero(const float* p, int H, int W, float ix, float iy) {
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wnw = ((float)x1 - ix) * ((float)y1 - iy), wne = (ix - (float)x0) * ((float)y1 - iy);
    const float wsw = ((float)x1 - ix) * (iy - (float)y0), wse = (ix - (float)x0) * (iy - (float)y0);
    float v = 0.f;
    if (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H) v += p[(long)y0 * W + x0] * wnw;
    if (x1 >= 0 && x1 < W && y0 >= 0 && y0 < H) v += p[(long)y0 * W + x1] * wne;
    if (x0 >= 0 && x0 < W && y1 >= 0 && y1 < H) v += p[(long)y1 * W + x0] * wsw;
    if (x1 >= 0 && x1 < W && y1 >= 0 && y1 < H) v += p[(long)y1 * W + x1] * wse;
    return v;
}

// BackWarp coordinates (Ours.py:908-920): normalise by w (not w-1), sample with align_corners=True, border
__device__ __forceinline__ void backwarp_coord(int x, int y, float u, float v, int H, int W, float* ix, float* iy) {
    float gx = (((float)x + u) / (float)W) * 2.f - 1.f;
    float gy = (((float)y + v) / (float)H) * 2.f - 1.f;
    float fx = ((gx + 1.f) / 2.f) * (float)(W - 1);
    float fy = ((gy + 1.f) / 2.f) * (float)(H - 1);
    fx = fminf(fmaxf(fx, 0.f), (float)(W - 1));
    fy = fminf(fmaxf(fy, 0.f), (float)(H - 1));
    *ix = fx; *iy = fy;
}

__global__ void backwarp_kernel(const float* __restrict__ img, const float* __restrict__ flow, float* __restrict__ out,
                                int C, int H, int W, float sign) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, n = blockIdx.z;
    if (x >= W) return;
    const long HW = (long)H * W, p = (long)y * W + x;
    float ix, iy;
    backwarp_coord(x, y, flow[((long)n * 2) * HW + p], flow[((long)n * 2 + 1) * HW + p], H, W, &ix, &iy);
    for (int c = 0; c < C; ++c) out[((long)n * C + c) * HW + p] = sign * bilinear_zero(img + ((long)n * C + c) * HW, H, W, ix, iy);
}

extern "C" int motif_backwarp(const float* img, const float* flow, float* out, int N, int C, int H, int W, float sign, void* stream) {
    if (!img || !flow || !out || N < 1 || C < 1) return MOTIF_EINVAL;
    dim3 grid(cdiv(W, 256), H, N);
    backwarp_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(img, flow, out, C, H, W, sign);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// PWC Decoder.Backward (PWCNet.py:146-177).  A workgroup = 64 pixels of a row x 4 channel slices of a group of PWC_WARP_CPB channels (round 6: one
// thread per pixel walking ALL channels took 88 us on the 24 x 40 level -- 24 workgroups of 40 live lanes, 128 dependent gathers each; every
// output is formed by the same expression as before, so the bits are unchanged).
#define PWC_WARP_CPB 8
__global__ __launch_bounds__(256) void pwc_warp_kernel(const float* __restrict__ img, const float* __restrict__ flow, const float* __restrict__ gxs,
                                                       const float* __restrict__ gys, float* __restrict__ out, int C, int H, int W, float dx, float dy,
                                                       int cgroups) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y, n = blockIdx.z / cgroups, cg = blockIdx.z - n * cgroups;
    if (x >= W) return;
    const long HW = (long)H * W, p = (long)y * W + x;
    const float gx = gxs[x] + flow[((long)n * 2) * HW + p] / dx;
    const float gy = gys[y] + flow[((long)n * 2 + 1) * HW + p] / dy;
    const float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f;    // align_corners=False
    const float iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    // validity = the sampled ones-channel (PWCNet.py:167,173-175)
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    float m = 0.f;
    if (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H) m += ((float)x1 - ix) * ((float)y1 - iy);
    if (x1 >= 0 && x1 < W && y0 >= 0 && y0 < H) m += (ix - (float)x0) * ((float)y1 - iy);
    if (x0 >= 0 && x0 < W && y1 >= 0 && y1 < H) m += ((float)x1 - ix) * (iy - (float)y0);
    if (x1 >= 0 && x1 < W && y1 >= 0 && y1 < H) m += (ix - (float)x0) * (iy - (float)y0);
    const float mask = m > 0.999f ? 1.f : 0.f;
    const int c1 = min(C, (cg + 1) * PWC_WARP_CPB);
    for (int c = cg * PWC_WARP_CPB + (threadIdx.x >> 6); c < c1; c += 4)
        out[((long)n * C + c) * HW + p] = bilinear_zero(img + ((long)n * C + c) * HW, H, W, ix, iy) * mask;
}

extern "C" int motif_pwc_backward_warp(const float* img, const float* flow, const float* gx_table, const float* gy_table,
                                       float* out, int N, int C, int H, int W, void* stream) {
    if (!img || !flow || !out || !gx_table || !gy_table || N < 1 || C < 1) return MOTIF_EINVAL;
    const int cgroups = cdiv(C, PWC_WARP_CPB);
    if ((long)N * cgroups > 65535) return MOTIF_ELIMIT;
    dim3 grid(cdiv(W, 64), H, N * cgroups);
    pwc_warp_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(img, flow, gx_table, gy_table, out, C, H, W,
                                                           (float)((W - 1.0) / 2.0), (float)((H - 1.0) / 2.0), cgroups);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// ------------------------------------------------------------------ reliability maps + flow encoder input
__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

__global__ void reliability_kernel(const float* __restrict__ fr0, const float* __restrict__ fr1, long fr_bs,
                                   const float* __restrict__ flow, const float* __restrict__ gf,
                                   float* __restrict__ psies, float* __restrict__ flow_feat, int B, int H, int W) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    const int jb = blockIdx.z, j = jb / B, b = jb % B;       // pair j in (00,01,10,11)
    if (x >= W) return;
    const long HW = (long)H * W, p = (long)y * W + x;
    const float* fl = flow + (long)jb * 2 * HW;
    const float u = fl[p], v = fl[HW + p];
    float ix, iy;
    backwarp_coord(x, y, u, v, H, W, &ix, &iy);
    // psi_photo (Ours.py:562-563)
    const float* src = ((j < 2) ? fr0 : fr1) + (long)b * fr_bs;
    const float* dst = ((j & 1) ? fr1 : fr0) + (long)b * fr_bs;
    float ph = 0.f;
    for (int c = 0; c < 3; ++c) ph += fabsf(src[c * HW + p] - bilinear_zero(dst + c * HW, H, W, ix, iy));
    ph = ph / 3.0f;
    // psi_flow (Ours.py:564-571): reverse pair of (00,01,10,11) is (00,10,01,11)
    const int jr = (j == 1) ? 2 : (j == 2 ? 1 : j);
    const float* fr = flow + (long)(jr * B + b) * 2 * HW;
    float pf = fabsf(u - (-bilinear_zero(fr, H, W, ix, iy))) + fabsf(v - (-bilinear_zero(fr + HW, H, W, ix, iy)));
    pf = (pf / 2.0f) / 10.0f;
    // psi_var (Ours.py:572-577): 3x3 Gaussian over reflect-padded f^2 and f
    float pv = 0.f;
    for (int c = 0; c < 2; ++c) {
        float m2 = 0.f, m1 = 0.f;
        for (int dy = 0; dy < 3; ++dy)
            for (int dx = 0; dx < 3; ++dx) {
                const float f = fl[c * HW + (long)reflect1(y + dy - 1, H) * W + reflect1(x + dx - 1, W)];
                const float g = gf[dy * 3 + dx];
                m2 += (f * f) * g;
                m1 += f * g;
            }
        float var = m2 - m1 * m1;
        var = var < 1e-9f ? 1e-9f : var;
        pv += sqrtf(var);
    }
    pv = pv / 2.0f;
    float* ps = psies + (long)jb * 3 * HW + p;
    ps[0] = ph; ps[HW] = pf; ps[2 * HW] = pv;
    // flow_feat[d*B+b][i*7 + c] with j = 2d + i (Ours.py:626-631)
    const int d = j >> 1, i = j & 1;
    float* ff = flow_feat + ((long)(d * B + b) * 14 + i * 7) * HW + p;
    ff[0] = u / 20.0f; ff[HW] = v / 20.0f;
    ff[2 * HW] = ph; ff[3 * HW] = pf; ff[4 * HW] = pv;
    ff[5 * HW] = (float)d; ff[6 * HW] = (float)i;      // ref_start_durations / 8
}

extern "C" int motif_reliability_fwd(const float* fr0, const float* fr1, long fr_bs, const float* flow, const float* g_filter,
                                     float* psies, float* flow_feat, int B, int H, int W, void* stream) {
    if (!fr0 || !fr1 || !flow || !g_filter || !psies || !flow_feat || B < 1) return MOTIF_EINVAL;
    dim3 grid(cdiv(W, 128), H, 4 * B);
    reliability_kernel<<<grid, 128, 0, (hipStream_t)stream>>>(fr0, fr1, fr_bs, flow, g_filter, psies, flow_feat, B, H, W);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// Table-driven form for the 4-frame generators (Ours_4.py:514-592, Ours_44.py:519-593): J flows per batch item, flow j
// goes from frame tab[j][0] to frame tab[j][1], is tab[j][2] in `flow` and its reverse is tab[j][3]; S flows share one
// source frame ("direction"): flow_feat[(j/S)*B + b][(j%S)*7 + c].
struct RelTab { int src[16], dst[16], fwd[16], rev[16]; float d0[16], d1[16]; };

__global__ void reliability_pairs_kernel(const float* __restrict__ frames, long frame_stride, long batch_stride,
                                         const float* __restrict__ flow, const float* __restrict__ gf, RelTab tab,
                                         float* __restrict__ psies, float* __restrict__ flow_feat, int S, int B, int H, int W) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    const int jb = blockIdx.z, j = jb / B, b = jb % B;
    if (x >= W) return;
    const long HW = (long)H * W, p = (long)y * W + x;
    const float* fl = flow + (long)(tab.fwd[j] * B + b) * 2 * HW;
    const float u = fl[p], v = fl[HW + p];
    float ix, iy;
    backwarp_coord(x, y, u, v, H, W, &ix, &iy);
    const float* src = frames + (long)b * batch_stride + (long)tab.src[j] * frame_stride;
    const float* dst = frames + (long)b * batch_stride + (long)tab.dst[j] * frame_stride;
    float ph = 0.f;
    for (int c = 0; c < 3; ++c) ph += fabsf(src[c * HW + p] - bilinear_zero(dst + c * HW, H, W, ix, iy));
    ph = ph / 3.0f;
    const float* fr = flow + (long)(tab.rev[j] * B + b) * 2 * HW;
    float pf = fabsf(u - (-bilinear_zero(fr, H, W, ix, iy))) + fabsf(v - (-bilinear_zero(fr + HW, H, W, ix, iy)));
    pf = (pf / 2.0f) / 10.0f;
    float pv = 0.f;
    for (int c = 0; c < 2; ++c) {
        float m2 = 0.f, m1 = 0.f;
        for (int dy = 0; dy < 3; ++dy)
            for (int dx = 0; dx < 3; ++dx) {
                const float f = fl[c * HW + (long)reflect1(y + dy - 1, H) * W + reflect1(x + dx - 1, W)];
                const float g = gf[dy * 3 + dx];
                m2 += (f * f) * g;
                m1 += f * g;
            }
        float var = m2 - m1 * m1;
        var = var < 1e-9f ? 1e-9f : var;
        pv += sqrtf(var);
    }
    pv = pv / 2.0f;
    float* ps = psies + (long)jb * 3 * HW + p;
    ps[0] = ph; ps[HW] = pf; ps[2 * HW] = pv;
    const int d = j / S, i = j % S;
    float* ff = flow_feat + ((long)(d * B + b) * (S * 7) + i * 7) * HW + p;
    ff[0] = u / 20.0f; ff[HW] = v / 20.0f;
    ff[2 * HW] = ph; ff[3 * HW] = pf; ff[4 * HW] = pv;
    ff[5 * HW] = tab.d0[j]; ff[6 * HW] = tab.d1[j];
}

extern "C" int motif_reliability_pairs_fwd(const float* frames, long frame_stride, long batch_stride, const float* flow,
                                           const float* g_filter, const int32_t* table, const float* durations, int J, int S,
                                           float* psies, float* flow_feat, int B, int H, int W, void* stream) {
    if (!frames || !flow || !g_filter || !table || !durations || !psies || !flow_feat || B < 1) return MOTIF_EINVAL;
    if (J < 1 || J > 16 || S < 1 || J % S) return MOTIF_EINVAL;
    RelTab tab;
    for (int j = 0; j < J; ++j) {           // host tables (a few dozen ints), passed by value
        tab.src[j] = table[4 * j]; tab.dst[j] = table[4 * j + 1]; tab.fwd[j] = table[4 * j + 2]; tab.rev[j] = table[4 * j + 3];
        tab.d0[j] = durations[2 * j]; tab.d1[j] = durations[2 * j + 1];
    }
    dim3 grid(cdiv(W, 128), H, J * B);
    reliability_pairs_kernel<<<grid, 128, 0, (hipStream_t)stream>>>(frames, frame_stride, batch_stride, flow, g_filter, tab, psies, flow_feat, S, B, H, W);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// ------------------------------------------------------------------ InstanceNorm2d (+relu, +residual)
__device__ __forceinline__ float block_sum(float v, float* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += sh[i];
    return t;
}

// gamma / beta (may be NULL; C channels, plane = image * C + channel): InstanceNorm2d(affine=True), y = x_hat * gamma[c] + beta[c]
__global__ __launch_bounds__(1024) void instance_norm_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                            float* __restrict__ out, int HW, int mode,
                                                            const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr, int C = 1) {
    __shared__ float sh[16];
    const long base = (long)blockIdx.x * HW;
    float s = 0.f;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) s += x[base + i];
    const float mean = block_sum(s, sh) / (float)HW;
    float q = 0.f;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) { const float d = x[base + i] - mean; q += d * d; }
    const float var = block_sum(q, sh) / (float)HW;
    const float inv = 1.0f / sqrtf(var + 1e-5f);
    const float ga = gamma ? gamma[blockIdx.x % C] : 1.f, be = gamma ? beta[blockIdx.x % C] : 0.f;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        float v = (x[base + i] - mean) * inv;
        if (gamma) v = v * ga + be;
        if (mode >= 1) v = v > 0.f ? v : 0.f;
        if (mode == 2) { v += res[base + i]; v = v > 0.f ? v : 0.f; }
        out[base + i] = v;
    }
}

// Large planes, few of them (RAFT's 1/2-resolution layers: 8 planes of 230k pixels per image): three
// grid-wide passes with the plane split over `S` blocks; partial sums in fp64 through a tiny workspace so
// the two-pass mean/variance stays as accurate as the single-block kernel.
__global__ __launch_bounds__(256) void in_partial_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                                        double* __restrict__ part, int HW, int S, int pass) {
    __shared__ float sh[16];
    const int plane = blockIdx.y, sb = blockIdx.x;
    const long base = (long)plane * HW;
    const int per = (HW + S - 1) / S, i0 = sb * per, i1 = min(HW, i0 + per);
    float mean = 0.f;
    if (pass == 1) mean = (float)(stats[plane * 2] / (double)HW);
    float s = 0.f;
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const float d = x[base + i] - mean;
        s += pass == 0 ? d : d * d;
    }
    const float t = block_sum(s, sh);
    if (threadIdx.x == 0) part[(long)plane * S + sb] = (double)t;
}

__global__ void in_reduce_kernel(const double* __restrict__ part, double* __restrict__ stats, int NC, int S, int pass) {
    const int plane = blockIdx.x * blockDim.x + threadIdx.x;
    if (plane >= NC) return;
    double t = 0.0;
    for (int i = 0; i < S; ++i) t += part[(long)plane * S + i];
    stats[plane * 2 + pass] = t;
}

// stats: [NC][2] = (sum, centred second moment) of every plane -- or, with part != nullptr, the S partial (sum, sum of squares) pairs the
// moments kernel wrote: thread 0 adds them in slice order (the order the former reduce kernel used: same bits) -- one launch less.
__global__ __launch_bounds__(256) void in_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                      const double* __restrict__ stats, float* __restrict__ out, int HW, int mode,
                                                      const double* __restrict__ part, int S,
                                                      const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr, int C = 1) {
    const int plane = blockIdx.y;
    const long base = (long)plane * HW;
    __shared__ double st[2];
    if (part) {
        if (threadIdx.x == 0) {
            double s_ = 0.0, q_ = 0.0;
            for (int i = 0; i < S; ++i) { s_ += part[((long)plane * S + i) * 2]; q_ += part[((long)plane * S + i) * 2 + 1]; }
            const double m2 = q_ - s_ * s_ / (double)HW;
            st[0] = s_; st[1] = m2 > 0.0 ? m2 : 0.0;
        }
        __syncthreads();
    }
    const float mean = (float)((part ? st[0] : stats[plane * 2]) / (double)HW);
    const float var = (float)((part ? st[1] : stats[plane * 2 + 1]) / (double)HW);
    const float inv = 1.0f / sqrtf(var + 1e-5f);
    const float ga = gamma ? gamma[plane % C] : 1.f, be = gamma ? beta[plane % C] : 0.f;
    // 16 bytes per lane where the plane allows it (4-byte accesses cost the same vector-memory instruction for a quarter of the bytes)
    const bool v4 = (HW & 3) == 0 && ((((unsigned long long)(x + base)) | ((unsigned long long)(out + base)) | (mode == 2 ? (unsigned long long)(res + base) : 0ull)) & 15) == 0;
    if (v4) {
        const f32x4* x4 = (const f32x4*)(x + base); const f32x4* r4 = (const f32x4*)(res + base); f32x4* o4 = (f32x4*)(out + base);
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (HW >> 2); i += gridDim.x * blockDim.x) {
            f32x4 v = x4[i];
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (mode == 2) r = r4[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = (v[e] - mean) * inv;
                if (gamma) t = t * ga + be;
                if (mode >= 1) t = t > 0.f ? t : 0.f;
                if (mode == 2) { t += r[e]; t = t > 0.f ? t : 0.f; }
                v[e] = t;
            }
            o4[i] = v;
        }
        return;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        float v = (x[base + i] - mean) * inv;
        if (gamma) v = v * ga + be;
        if (mode >= 1) v = v > 0.f ? v : 0.f;
        if (mode == 2) { v += res[base + i]; v = v > 0.f ? v : 0.f; }
        out[base + i] = v;
    }
}

extern "C" int motif_instance_norm(const float* x, const float* res, float* out, int NC, int HW, int mode, void* stream) {
    if (!x || !out || NC < 1 || HW < 1 || (mode == 2 && !res)) return MOTIF_EINVAL;
    instance_norm_kernel<<<NC, 1024, 0, (hipStream_t)stream>>>(x, res, out, HW, mode);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// One pass over the plane for both moments: sum and sum of squares accumulated in fp64 (x*x is exact in fp64, so
// M2 = sum(x^2) - sum(x)^2 / n loses nothing an fp32 two-pass would keep) -- one read of the tensor less than
// mean-pass + variance-pass.
__device__ __forceinline__ double block_sum_d(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += sh[i];
    return t;
}

__global__ __launch_bounds__(256) void in_moments_kernel(const float* __restrict__ x, double* __restrict__ part, int HW, int S) {
    __shared__ double sh[8];
    const int plane = blockIdx.y, sb = blockIdx.x;
    const long base = (long)plane * HW;
    const int per = (HW + S - 1) / S, i0 = sb * per, i1 = min(HW, i0 + per);
    double s = 0.0, q = 0.0;
    // 16-byte loads where the slice allows it: a thread then takes 4 consecutive values per step instead of every 256th -- another
    // (equally valid) fp64 summation order; the moments may differ in their last fp64 bits.
    if ((per & 3) == 0 && (HW & 3) == 0 && (((unsigned long long)(x + base)) & 15) == 0) {
        const f32x4* x4 = (const f32x4*)(x + base);
        for (int i = (i0 >> 2) + threadIdx.x; i < (i1 >> 2); i += blockDim.x) {
            const f32x4 v4 = x4[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double v = (double)v4[e]; s += v; q = fma(v, v, q); }
        }
    } else {
        for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
            const double v = (double)x[base + i];
            s += v;
            q = fma(v, v, q);
        }
    }
    const double ts = block_sum_d(s, sh), tq = block_sum_d(q, sh);
    if (threadIdx.x == 0) { part[((long)plane * S + sb) * 2] = ts; part[((long)plane * S + sb) * 2 + 1] = tq; }
}

extern "C" int motif_instance_norm_ws(const float* x, const float* res, float* out, double* workspace, int NC, int HW, int mode, void* stream) {
    if (!x || !out || NC < 1 || HW < 1 || (mode == 2 && !res)) return MOTIF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int S = 1024 / NC;                       // aim at ~1024 blocks
    if (S > HW / 4096) S = HW / 4096;
    if (!workspace || S < 2) return motif_instance_norm(x, res, out, NC, HW, mode, stream);
    if (S > 64) S = 64;
    double* stats = workspace;               // [NC][2]
    double* part = workspace + 2L * NC;      // [NC][S][2]
    in_moments_kernel<<<dim3(S, NC), 256, 0, s>>>(x, part, HW, S);
    in_apply_kernel<<<dim3(S, NC), 256, 0, s>>>(x, res, stats, out, HW, mode, part, S);      // every block adds the S partials itself
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// InstanceNorm2d(C, affine=True) (PWCNet_light's input normalisation, OpticalFlow/PWCNet_light.py:18,259-260): the kernels above with
// y = x_hat * gamma[c] + beta[c]; x [N,C,HW] dense.  workspace as motif_instance_norm_ws (may be NULL: one block per plane).
extern "C" int motif_instance_norm_affine_ws(const float* x, const float* gamma, const float* beta, float* out, double* workspace, int N, int C, int HW, void* stream) {
    if (!x || !gamma || !beta || !out || N < 1 || C < 1 || HW < 1) return MOTIF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int NC = N * C;
    int S = 1024 / NC;
    if (S > HW / 4096) S = HW / 4096;
    if (S > 64) S = 64;
    if (!workspace || S < 2) instance_norm_kernel<<<NC, 1024, 0, s>>>(x, nullptr, out, HW, 0, gamma, beta, C);
    else {
        double* part = workspace + 2L * NC;
        in_moments_kernel<<<dim3(S, NC), 256, 0, s>>>(x, part, HW, S);
        in_apply_kernel<<<dim3(S, NC), 256, 0, s>>>(x, nullptr, workspace, out, HW, 0, part, S, gamma, beta, C);
    }
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// InstanceNorm with statistics that span several processes (one clip tiled over GPUs in row bands): each rank
// sums x and x^2 in fp64 over the rows it OWNS (its band, not the halo around it), the [NC][3] doubles (sum, sum of squares, count) are
// all-reduced by the caller (RCCL), and every rank normalises its whole crop with the global mean / variance.
__global__ __launch_bounds__(256) void in_moments_rows_kernel(const float* __restrict__ x, double* __restrict__ sums, int H, int W,
                                                              int row_lo, int row_hi) {
    __shared__ double sh[8];
    const int plane = blockIdx.x;
    const float* p = x + (long)plane * H * W + (long)row_lo * W;
    const long n = (long)(row_hi - row_lo) * W;
    double s = 0.0, q = 0.0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = (double)p[i];
        s += v;
        q = fma(v, v, q);
    }
    const double ts = block_sum_d(s, sh), tq = block_sum_d(q, sh);
    if (threadIdx.x == 0) { sums[plane * 3] = ts; sums[plane * 3 + 1] = tq; sums[plane * 3 + 2] = (double)n; }
}

__global__ __launch_bounds__(256) void in_apply_sums_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                           const double* __restrict__ sums, float* __restrict__ out, int HW, int mode) {
    const int plane = blockIdx.y;
    const long base = (long)plane * HW;
    const double n = sums[plane * 3 + 2], sx = sums[plane * 3];
    double m2 = sums[plane * 3 + 1] - sx * sx / n;
    m2 = m2 > 0.0 ? m2 : 0.0;
    const float mean = (float)(sx / n);
    const float var = (float)(m2 / n);
    const float inv = 1.0f / sqrtf(var + 1e-5f);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        float v = (x[base + i] - mean) * inv;
        if (mode >= 1) v = v > 0.f ? v : 0.f;
        if (mode == 2) { v += res[base + i]; v = v > 0.f ? v : 0.f; }
        out[base + i] = v;
    }
}

extern "C" int motif_instance_norm_moments(const float* x, double* sums, int NC, int H, int W, int row_lo, int row_hi, void* stream) {
    if (!x || !sums || NC < 1 || H < 1 || W < 1 || row_lo < 0 || row_hi > H || row_lo >= row_hi) return MOTIF_EINVAL;
    in_moments_rows_kernel<<<NC, 256, 0, (hipStream_t)stream>>>(x, sums, H, W, row_lo, row_hi);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

extern "C" int motif_instance_norm_apply(const float* x, const float* res, const double* sums, float* out, int NC, int HW, int mode, void* stream) {
    if (!x || !sums || !out || NC < 1 || HW < 1 || (mode == 2 && !res)) return MOTIF_EINVAL;
    if (NC > 65535) return MOTIF_ELIMIT;
    int gx = HW / 2048; gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    in_apply_sums_kernel<<<dim3(gx, NC), 256, 0, (hipStream_t)stream>>>(x, res, sums, out, HW, mode);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// ------------------------------------------------------------------ small streaming kernels
__global__ void avg_pool2_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int Ho, int Wo) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    const long nc = blockIdx.z;
    if (x >= Wo) return;
    const float* p = in + nc * (long)H * W + (long)(2 * y) * W + 2 * x;
    out[nc * (long)Ho * Wo + (long)y * Wo + x] = (p[0] + p[1] + p[W] + p[W + 1]) * 0.25f;
}

extern "C" int motif_avg_pool2(const float* in, float* out, int NC, int H, int W, void* stream) {
    if (!in || !out || NC < 1 || H < 2 || W < 2) return MOTIF_EINVAL;
    const int Ho = H / 2, Wo = W / 2;
    dim3 grid(cdiv(Wo, 128), Ho, NC);
    avg_pool2_kernel<<<grid, 128, 0, (hipStream_t)stream>>>(in, out, H, W, Ho, Wo);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// NCHW -> NHWC through an LDS tile (both sides coalesced)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 256 threads: 8 rows per pass
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        tile[r][tx] = (c < C && p < HW) ? in[((long)n * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        if (c < C && p < HW) out[((long)n * HW + p) * C + c] = tile[tx][r];
    }
}

extern "C" int motif_nchw_to_nhwc(const float* in, float* out, int N, int C, int HW, void* stream) {
    if (!in || !out || N < 1 || C < 1 || HW < 1) return MOTIF_EINVAL;
    dim3 grid(cdiv(HW, 32), cdiv(C, 32), N);
    nchw_to_nhwc_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(in, out, C, HW);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// four values per thread with 16-byte accesses where the tensors allow it; the last n % 4 values (and unaligned tensors) go one by one
__global__ void gru_update_kernel(const float* __restrict__ z, const float* __restrict__ q, const float* __restrict__ h,
                                  float* __restrict__ out, long n, int v4) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v4) {
        const long i = t * 4;
        if (i + 4 <= n) {
            const f32x4 zz = *(const f32x4*)(z + i), qq = *(const f32x4*)(q + i), hh = *(const f32x4*)(h + i);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (1.f - zz[e]) * hh[e] + zz[e] * qq[e];
            *(f32x4*)(out + i) = o;
        } else
            for (long j = i; j < n; ++j) out[j] = (1.f - z[j]) * h[j] + z[j] * q[j];
    } else if (t < n)
        out[t] = (1.f - z[t]) * h[t] + z[t] * q[t];
}

extern "C" int motif_gru_update(const float* z, const float* q, const float* h, float* out, long n, void* stream) {
    if (!z || !q || !h || !out || n < 1) return MOTIF_EINVAL;
    const int v4 = ((((unsigned long long)z | (unsigned long long)q | (unsigned long long)h | (unsigned long long)out)) & 15) == 0;
    gru_update_kernel<<<cdiv(v4 ? cdiv(n, 4) : n, 256), 256, 0, (hipStream_t)stream>>>(z, q, h, out, n, v4);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// V = values per thread: 4 (16-byte accesses) when HW % 4 == 0 and the tensors are 16-byte aligned, else 1
template <int V>
__global__ void lstm_gates_kernel(const float* __restrict__ cc, const float* __restrict__ c_cur, float* __restrict__ h_next,
                                  float* __restrict__ c_next, int hid, long HW, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * V;   // over B*hid*HW
    if (i >= n) return;
    const long per = (long)hid * HW;
    const long b = i / per, r = i - b * per;
    const float* g4 = cc + b * 4 * per + r;
    float gi[V], gf[V], go[V], gg[V], cc_[V], cn[V], hn[V];
    if constexpr (V == 4) {
        *(f32x4*)gi = *(const f32x4*)g4; *(f32x4*)gf = *(const f32x4*)(g4 + per);
        *(f32x4*)go = *(const f32x4*)(g4 + 2 * per); *(f32x4*)gg = *(const f32x4*)(g4 + 3 * per);
        *(f32x4*)cc_ = *(const f32x4*)(c_cur + i);
    } else { gi[0] = g4[0]; gf[0] = g4[per]; go[0] = g4[2 * per]; gg[0] = g4[3 * per]; cc_[0] = c_cur[i]; }
#pragma unroll
    for (int e = 0; e < V; ++e) {
        cn[e] = sigmoidf_(gf[e]) * cc_[e] + sigmoidf_(gi[e]) * tanhf(gg[e]);
        hn[e] = sigmoidf_(go[e]) * tanhf(cn[e]);
    }
    if constexpr (V == 4) { *(f32x4*)(c_next + i) = *(f32x4*)cn; *(f32x4*)(h_next + i) = *(f32x4*)hn; }
    else { c_next[i] = cn[0]; h_next[i] = hn[0]; }
}

extern "C" int motif_lstm_gates(const float* cc, const float* c_cur, float* h_next, float* c_next, int B, int hid, int HW, void* stream) {
    if (!cc || !c_cur || !h_next || !c_next || B < 1 || hid < 1 || HW < 1) return MOTIF_EINVAL;
    const long n = (long)B * hid * HW;
    const bool v4 = (HW & 3) == 0 && ((((unsigned long long)cc | (unsigned long long)c_cur | (unsigned long long)h_next | (unsigned long long)c_next)) & 15) == 0;
    if (v4) lstm_gates_kernel<4><<<cdiv(n / 4, 256), 256, 0, (hipStream_t)stream>>>(cc, c_cur, h_next, c_next, hid, HW, n);
    else lstm_gates_kernel<1><<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(cc, c_cur, h_next, c_next, hid, HW, n);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

__global__ void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float a, float b, float* __restrict__ out, long n,
                             int v4) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v4) {
        const long i = t * 4;
        if (i + 4 <= n) {
            const f32x4 xx = *(const f32x4*)(x + i);
            f32x4 o;
            if (y) {
                const f32x4 yy = *(const f32x4*)(y + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = a * xx[e] + b * yy[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = a * xx[e];
            }
            *(f32x4*)(out + i) = o;
        } else
            for (long j = i; j < n; ++j) out[j] = y ? a * x[j] + b * y[j] : a * x[j];
    } else if (t < n)
        out[t] = y ? a * x[t] + b * y[t] : a * x[t];
}

extern "C" int motif_axpby(const float* x, const float* y, float a, float b, float* out, long n, void* stream) {
    if (!x || !out || n < 1) return MOTIF_EINVAL;
    const int v4 = ((((unsigned long long)x | (unsigned long long)y | (unsigned long long)out)) & 15) == 0;
    axpby_kernel<<<cdiv(v4 ? cdiv(n, 4) : n, 256), 256, 0, (hipStream_t)stream>>>(x, y, a, b, out, n, v4);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// a*x + b*y of B items of n elements each, written to a batch-strided destination (out + i * out_bs): RAFT's flow = coords1 - coords0
// goes straight into channels 144..145 of the GRU input buffer (raft.py:117-120 concatenates it there)
__global__ void axpby_bs_kernel(const float* __restrict__ x, const float* __restrict__ y, float a, float b, float* __restrict__ out, long n, long out_bs) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long i = blockIdx.y;
    if (t < n) out[i * out_bs + t] = y ? a * x[i * n + t] + b * y[i * n + t] : a * x[i * n + t];
}

extern "C" int motif_axpby_bs(const float* x, const float* y, float a, float b, float* out, int B, long n, long out_bs, void* stream) {
    if (!x || !out || n < 1 || B < 1) return MOTIF_EINVAL;
    if (B > 65535) return MOTIF_ELIMIT;
    axpby_bs_kernel<<<dim3(cdiv(n, 256), B), 256, 0, (hipStream_t)stream>>>(x, y, a, b, out, n, out_bs);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// The flow the generator returns (Ours.py:794, 858): the predicted flow is scaled up for the splat, flow = (p * 20) * ratio, and
// scaled back for the caller, (flow / 20) / ratio -- four separately rounded torch operations there, one pass here with the same four
// roundings (IEEE division; no contraction possible between a multiply and a divide).  pred [N,3,Q] -> out [N,2,Q].
__global__ void flow_roundtrip_kernel(const float* __restrict__ pred, float* __restrict__ out, long Q, float a, float b) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = blockIdx.y;
    if (i >= 2 * Q) return;
    const float f = (pred[n * 3 * Q + i] * a) * b;
    out[n * 2 * Q + i] = (f / a) / b;
}

extern "C" int motif_flow_roundtrip(const float* pred, float* out, int N, long Q, float a, float b, void* stream) {
    if (!pred || !out || N < 1 || Q < 1) return MOTIF_EINVAL;
    if (N > 65535) return MOTIF_ELIMIT;
    flow_roundtrip_kernel<<<dim3(cdiv(2 * Q, 256), N), 256, 0, (hipStream_t)stream>>>(pred, out, Q, a, b);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// ------------------------------------------------------------------ ConvTranspose2d(k=4, s=2, p=1), small Cout
__global__ void deconv4x4s2_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                   float* __restrict__ out, int Cin, int Cout, int H, int W) {
    const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
    const int n = blockIdx.z / Cout, co = blockIdx.z % Cout;
    const int Ho = 2 * H, Wo = 2 * W;
    if (ox >= Wo) return;
    // oy = 2*iy - 1 + ky  ->  ky has the parity of oy+1: an output has 2 x 2 contributing taps.  Their four sums run as independent
    // FMA chains over the channels, two channels per step (eight chains: the single chain of tap-outer loops was pure latency, 0.66 ms
    // for PWC-Net's 565-channel level); taps outside the image read plane offset 0 with weight 0.
    const long HW = (long)H * W;
    long ioff[4];
    long woff[4];
    float on[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ky = ((oy + 1) & 1) + 2 * (t >> 1), kx = ((ox + 1) & 1) + 2 * (t & 1);
        const int iy = (oy + 1 - ky) / 2, ix = (ox + 1 - kx) / 2;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W && (oy + 1 - ky) >= 0 && (ox + 1 - kx) >= 0;
        ioff[t] = ok ? (long)iy * W + ix : 0;
        woff[t] = (long)co * 16 + ky * 4 + kx;
        on[t] = ok ? 1.f : 0.f;
    }
    const float* ip = in + (long)n * Cin * HW;
    const long wstep = (long)Cout * 16;
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
    int ci = 0;
    for (; ci + 1 < Cin; ci += 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a0[t] = fmaf(ip[(long)ci * HW + ioff[t]], w[(long)ci * wstep + woff[t]] * on[t], a0[t]);
            a1[t] = fmaf(ip[(long)(ci + 1) * HW + ioff[t]], w[(long)(ci + 1) * wstep + woff[t]] * on[t], a1[t]);
        }
    }
    if (ci < Cin) {
#pragma unroll
        for (int t = 0; t < 4; ++t) a0[t] = fmaf(ip[(long)ci * HW + ioff[t]], w[(long)ci * wstep + woff[t]] * on[t], a0[t]);
    }
    float acc = bias ? bias[co] : 0.f;
    acc += ((a0[0] + a1[0]) + (a0[1] + a1[1])) + ((a0[2] + a1[2]) + (a0[3] + a1[3]));
    out[((long)(n * Cout + co) * Ho + oy) * Wo + ox] = acc;
}

// Long reductions (PWC-Net's moduleUpfeat: 529 .. 661 channels -> 2): the channels are cut into NS slices, one per 128-thread row of the
// workgroup; the NS partial sums meet in LDS and slice 0 adds them in slice order (fixed order: deterministic).  One thread per output
// pixel walking 600 channels alone was pure load latency (110-177 us per call for a few megabytes).
template <int NS>
__global__ __launch_bounds__(128 * NS) void deconv4x4s2_sliced_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                                                     float* __restrict__ out, int Cin, int Cout, int H, int W) {
    __shared__ float red[NS][128];
    const int lx = threadIdx.x & 127, slice = threadIdx.x >> 7;
    const int ox = blockIdx.x * 128 + lx, oy = blockIdx.y;
    const int n = blockIdx.z / Cout, co = blockIdx.z % Cout;
    const int Ho = 2 * H, Wo = 2 * W;
    const bool live = ox < Wo;
    const long HW = (long)H * W;
    long ioff[4], woff[4];
    float on[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ky = ((oy + 1) & 1) + 2 * (t >> 1), kx = ((ox + 1) & 1) + 2 * (t & 1);
        const int iy = (oy + 1 - ky) / 2, ix = (ox + 1 - kx) / 2;
        const bool ok = live && iy >= 0 && iy < H && ix >= 0 && ix < W && (oy + 1 - ky) >= 0 && (ox + 1 - kx) >= 0;
        ioff[t] = ok ? (long)iy * W + ix : 0;
        woff[t] = (long)co * 16 + ky * 4 + kx;
        on[t] = ok ? 1.f : 0.f;
    }
    const float* ip = in + (long)n * Cin * HW;
    const long wstep = (long)Cout * 16;
    const int per = (Cin + NS - 1) / NS, cbeg = slice * per, cend = cbeg + per < Cin ? cbeg + per : Cin;
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
    int ci = cbeg;
    for (; ci + 1 < cend; ci += 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a0[t] = fmaf(ip[(long)ci * HW + ioff[t]], w[(long)ci * wstep + woff[t]] * on[t], a0[t]);
            a1[t] = fmaf(ip[(long)(ci + 1) * HW + ioff[t]], w[(long)(ci + 1) * wstep + woff[t]] * on[t], a1[t]);
        }
    }
    if (ci < cend) {
#pragma unroll
        for (int t = 0; t < 4; ++t) a0[t] = fmaf(ip[(long)ci * HW + ioff[t]], w[(long)ci * wstep + woff[t]] * on[t], a0[t]);
    }
    red[slice][lx] = ((a0[0] + a1[0]) + (a0[1] + a1[1])) + ((a0[2] + a1[2]) + (a0[3] + a1[3]));
    __syncthreads();
    if (slice == 0 && live) {
        float acc = bias ? bias[co] : 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) acc += red[s][lx];
        out[((long)(n * Cout + co) * Ho + oy) * Wo + ox] = acc;
    }
}

extern "C" int motif_deconv4x4s2(const float* in, const float* weight, const float* bias, float* out,
                                 int N, int Cin, int Cout, int H, int W, void* stream) {
    if (!in || !weight || !out || N < 1 || Cin < 1 || Cout < 1) return MOTIF_EINVAL;
    dim3 grid(cdiv(2 * W, 128), 2 * H, N * Cout);
    if (Cin >= 64) {
        deconv4x4s2_sliced_kernel<8><<<grid, 128 * 8, 0, (hipStream_t)stream>>>(in, weight, bias, out, Cin, Cout, H, W);
        MOTIF_LAUNCH_CHECK();
        return MOTIF_OK;
    }
    deconv4x4s2_kernel<<<grid, 128, 0, (hipStream_t)stream>>>(in, weight, bias, out, Cin, Cout, H, W);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}


// ------------------------------------------------------------------ frame formats either side of the path
// decode: uint8 interleaved frames [N,H,W,3] as cv2.imread yields them (BGR) -> fp32 planar [N,3,H,W] in [0,1], RGB:
// `img.astype(np.float32) / 255.` then `[:, :, :, [2, 1, 0]]` and HWC->CHW (data/Adobe_test_3.py:171-195).
__global__ void frames_u8_to_f32_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, long HW, int swap) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = blockIdx.y;
    if (p >= HW) return;
    const unsigned char* q = in + (n * HW + p) * 3;
    const float c0 = (float)q[0] / 255.0f, c1 = (float)q[1] / 255.0f, c2 = (float)q[2] / 255.0f;
    float* o = out + n * 3 * HW + p;
    o[0] = swap ? c2 : c0;
    o[HW] = c1;
    o[2 * HW] = swap ? c0 : c2;
}

// encode: fp32 planar RGB [N,3,H,W] -> uint8 interleaved [N,H,W,3].  mode 1: clamp(0,1), *255, round half to even,
// RGB->BGR when swap (utils/util.py:105-129 tensor2img); mode 0: clamp, *255, truncate (demo.py:94-99).
__global__ void frames_f32_to_u8_kernel(const float* __restrict__ in, unsigned char* __restrict__ out, long HW, int mode, int swap) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = blockIdx.y;
    if (p >= HW) return;
    const float* q = in + n * 3 * HW + p;
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = q[k * HW];
        v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);           // NaN -> passes through both tests like torch.clamp; cast gives 0
        v = v * 255.0f;
        c[k] = mode ? rintf(v) : truncf(v);
    }
    unsigned char* o = out + (n * HW + p) * 3;
    o[0] = (unsigned char)(swap ? c[2] : c[0]);
    o[1] = (unsigned char)c[1];
    o[2] = (unsigned char)(swap ? c[0] : c[2]);
}

extern "C" int motif_frames_u8_to_f32(const unsigned char* in, float* out, int N, int H, int W, int swap_rb, void* stream) {
    if (!in || !out || N < 1 || H < 1 || W < 1) return MOTIF_EINVAL;
    const long HW = (long)H * W;
    dim3 grid(cdiv(HW, 256), N);
    frames_u8_to_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(in, out, HW, swap_rb);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

extern "C" int motif_frames_f32_to_u8(const float* in, unsigned char* out, int N, int H, int W, int round_mode, int swap_rb, void* stream) {
    if (!in || !out || N < 1 || H < 1 || W < 1) return MOTIF_EINVAL;
    const long HW = (long)H * W;
    dim3 grid(cdiv(HW, 256), N);
    frames_f32_to_u8_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(in, out, HW, round_mode, swap_rb);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

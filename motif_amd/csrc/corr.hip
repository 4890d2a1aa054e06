// Correlation kernels of the flow extractors (gfx950).
#include "common.h"

// ------------------------------------------------------------------ RAFT windowed bilinear lookup
// One wave per query pixel: the 64 lanes are the (2r+2)^2 = 8x8 integer neighbourhood of floor(coords)
// (r = 3, RAFT-small), each lane owns one 128-channel dot product against the channels-last f2 (float4
// loads, f1 broadcast).  The (2r+1)^2 = 49 window values are the bilinear blends of neighbouring lanes
// (three ds_bpermute shuffles), staged per block in LDS as [49][64 queries] and written with coalesced
// 256-byte rows into the [B,196,H,W] correlation tensor the update block consumes.
__global__ __launch_bounds__(256) void raft_lookup_kernel(const float* __restrict__ fmap1, const float* __restrict__ fmap2,
                                                         const float* __restrict__ coords, float coord_scale,
                                                         float* __restrict__ out, int H1, int W1, int H2, int W2, int C,
                                                         int out_C, int ch_off, float div) {
    __shared__ float tile[49][65];
    const int b = blockIdx.y;
    const long HW1 = (long)H1 * W1;
    const long q0 = (long)blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gy = lane >> 3, gx = lane & 7;
    for (int i = 0; i < 16; ++i) {
        const int ql = wave * 16 + i;
        const long q = q0 + ql;
        if (q >= HW1) break;                                    // wave-uniform
        const float x = coords[((long)b * 2) * HW1 + q] * coord_scale;
        const float y = coords[((long)b * 2 + 1) * HW1 + q] * coord_scale;
        const float fx = floorf(x), fy = floorf(y);
        const float dx = x - fx, dy = y - fy;
        const int h2 = (int)fy - 3 + gy, w2 = (int)fx - 3 + gx;
        float s = 0.f;
        if (h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2) {
            const f32x4* f1 = (const f32x4*)(fmap1 + ((long)b * HW1 + q) * C);
            const f32x4* f2 = (const f32x4*)(fmap2 + (((long)b * H2 + h2) * W2 + w2) * C);
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            for (int c = 0; c < C / 4; ++c) {
                const f32x4 a = f1[c], v = f2[c];
                s0 = fmaf(a[0], v[0], s0); s1 = fmaf(a[1], v[1], s1); s2 = fmaf(a[2], v[2], s2); s3 = fmaf(a[3], v[3], s3);
            }
            s = (s0 + s1) + (s2 + s3);
        }
        const float s_e = __shfl(s, (lane + 1) & 63), s_s = __shfl(s, (lane + 8) & 63), s_se = __shfl(s, (lane + 9) & 63);
        if (gy < 7 && gx < 7) {
            float v = s * (1.f - dy) * (1.f - dx);
            v += s_e * (1.f - dy) * dx;
            v += s_s * dy * (1.f - dx);
            v += s_se * dy * dx;
            tile[gx * 7 + gy][ql] = v / div;                    // channel = ix*(2r+1) + iy
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 49 * 64; idx += 256) {
        const int ch = idx >> 6, ql = idx & 63;
        const long q = q0 + ql;
        if (q < HW1) out[((long)b * out_C + ch_off + ch) * HW1 + q] = tile[ch][ql];
    }
}

extern "C" int motif_raft_corr_lookup(const float* fmap1, const float* fmap2, const float* coords, float coord_scale,
                                      float* out, int B, int H1, int W1, int H2, int W2, int C, int r,
                                      int out_C, int ch_off, float div, void* stream) {
    if (!fmap1 || !fmap2 || !coords || !out || B < 1) return MOTIF_EINVAL;
    if (r != 3 || (C & 3)) return MOTIF_ELIMIT;
    dim3 grid(cdiv((long)H1 * W1, 64), B);
    raft_lookup_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(fmap1, fmap2, coords, coord_scale, out, H1, W1, H2, W2, C, out_C, ch_off, div);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// ------------------------------------------------------------------ PWC-Net 9x9 cost volume
// One thread per (displacement, pixel); lanes run along x so f1 and the shifted f2 reads are both
// coalesced, the 81 displacements of a pixel tile reuse f1/f2 lines through L1/L2.
__global__ void corr81_kernel(const float* __restrict__ f1, const float* __restrict__ f2, float* __restrict__ out,
                              int C, int H, int W, int act) {
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
    out[((long)b * 81 + d) * HW + (long)y * W + x] = act_apply(s / (float)C, act);
}

extern "C" int motif_corr81_fwd(const float* first, const float* second, float* out, int B, int C, int H, int W,
                                int act, void* stream) {
    if (!first || !second || !out || B < 1 || C < 1) return MOTIF_EINVAL;
    dim3 grid(cdiv(W, 64), H, B * 81);
    corr81_kernel<<<grid, 64, 0, (hipStream_t)stream>>>(first, second, out, C, H, W, act);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

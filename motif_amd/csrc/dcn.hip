// Modulated deformable convolution v2 forward (gfx950): deformable im2col, then the fp32-MFMA
// implicit-GEMM engine as a 1x1 convolution over the C*kh*kw column channels (bias + activation fused).
// The sampling position, its four corner offsets and bilinear weights are computed once per
// (deformable group, tap, pixel) and reused for the group's channels; the reference recomputes them per
// channel (dcn_v2_im2col_cuda.cu:125-194).
#include "common.h"

__global__ __launch_bounds__(256) void dcn_im2col_kernel(const float* __restrict__ im, const float* __restrict__ offset,
                                                        const float* __restrict__ mask, float* __restrict__ col,
                                                        int C, int H, int W, int Ho, int Wo, int kh, int kw,
                                                        int stride, int pad, int dil, int dg, long offset_bs, long mask_bs) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int T = kh * kw;
    const int z = blockIdx.z;                 // (b, g, tap)
    const int tap = z % T, g = (z / T) % dg, b = z / (T * dg);
    if (x >= Wo) return;
    const int i = tap / kw, j = tap % kw;
    const long HWo = (long)Ho * Wo, p = (long)y * Wo + x;
    const float* op = offset + (long)b * offset_bs + (long)g * 2 * T * HWo;
    const float offset_h = op[(long)(2 * tap) * HWo + p];
    const float offset_w = op[(long)(2 * tap + 1) * HWo + p];
    const float m = mask[(long)b * mask_bs + ((long)g * T + tap) * HWo + p];
    const float h_im = (float)(y * stride - pad + i * dil) + offset_h;
    const float w_im = (float)(x * stride - pad + j * dil) + offset_w;
    const int cpg = C / dg;
    const long HW = (long)H * W;
    const bool inside = h_im > -1 && w_im > -1 && h_im < H && w_im < W;
    int h_low = 0, w_low = 0;
    float w1 = 0, w2 = 0, w3 = 0, w4 = 0;
    bool v1 = false, v2 = false, v3 = false, v4 = false;
    if (inside) {
        h_low = (int)floorf(h_im); w_low = (int)floorf(w_im);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - h_low, lw = w_im - w_low, hh = 1 - lh, hw = 1 - lw;
        v1 = h_low >= 0 && w_low >= 0;
        v2 = h_low >= 0 && w_high <= W - 1;
        v3 = h_high <= H - 1 && w_low >= 0;
        v4 = h_high <= H - 1 && w_high <= W - 1;
        w1 = hh * hw; w2 = hh * lw; w3 = lh * hw; w4 = lh * lw;
    }
    const long o1 = (long)h_low * W + w_low;
    for (int cc = 0; cc < cpg; ++cc) {
        const int c = g * cpg + cc;
        float val = 0.f;
        if (inside) {
            const float* ip = im + ((long)b * C + c) * HW;
            const float a1 = v1 ? ip[o1] : 0.f, a2 = v2 ? ip[o1 + 1] : 0.f;
            const float a3 = v3 ? ip[o1 + W] : 0.f, a4 = v4 ? ip[o1 + W + 1] : 0.f;
            val = (w1 * a1 + w2 * a2 + w3 * a3 + w4 * a4);
        }
        col[(((long)b * C + c) * T + tap) * HWo + p] = val * m;
    }
}

extern "C" int motif_dcn_v2_fwd(const float* input, const float* offset, const float* mask, const float* packed,
                                const float* bias, float* columns, float* out,
                                int B, int C, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil,
                                int deformable_groups, long offset_bs, long mask_bs, int act, void* stream) {
    if (!input || !offset || !mask || !packed || !columns || !out) return MOTIF_EINVAL;
    if (B < 1 || C < 1 || deformable_groups < 1 || C % deformable_groups) return MOTIF_EINVAL;
    const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
    const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    const int T = kh * kw;
    const long HWo = (long)Ho * Wo;
    if (!offset_bs) offset_bs = (long)deformable_groups * 2 * T * HWo;
    if (!mask_bs) mask_bs = (long)deformable_groups * T * HWo;
    dim3 grid(cdiv(Wo, 64), Ho, B * deformable_groups * T);
    dcn_im2col_kernel<<<grid, 64, 0, (hipStream_t)stream>>>(input, offset, mask, columns, C, H, W, Ho, Wo, kh, kw,
                                                            stride, pad, dil, deformable_groups, offset_bs, mask_bs);
    MOTIF_LAUNCH_CHECK();
    MotifConvDesc d = {};
    d.N = B; d.H = Ho; d.W = Wo; d.C0 = C * T; d.C1 = 0; d.Cout = Cout; d.KH = 1; d.KW = 1;
    d.stride = 1; d.pad = 0; d.dil = 1; d.groups = 1; d.pad_mode = 0; d.act = act; d.act2 = 0; d.act_split = 0; d.res_mode = 0;
    return motif_conv2d_fwd(&d, columns, nullptr, packed, bias, nullptr, out, stream);
}

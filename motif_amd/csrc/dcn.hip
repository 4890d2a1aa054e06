// Modulated deformable convolution v2 forward (gfx950): deformable im2col, then the fp32-MFMA
// implicit-GEMM engine as a 1x1 convolution over the C*kh*kw column channels (bias + activation fused).
// The sampling position, its four corner offsets and bilinear weights are computed once per
// (deformable group, tap, pixel) and reused for the group's channels; the reference recomputes them per
// channel (dcn_v2_im2col_cuda.cu:125-194).
#include "common.h"

struct DcnArgs {
    const float* im[4]; const float* offset[4]; const float* mask[4];
    long im_bs[4];
    float* col;
    int B, C, H, W, Ho, Wo, kh, kw, stride, pad, dil, dg;
    long offset_bs, mask_bs;
};

__global__ __launch_bounds__(64) void dcn_im2col_kernel(DcnArgs a) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int T = a.kh * a.kw;
    int z = blockIdx.z;                       // (problem, b, g, tap)
    const int tap = z % T; z /= T;
    const int g = z % a.dg; z /= a.dg;
    const int b = z % a.B, pz = z / a.B;
    if (x >= a.Wo) return;
    const int i = tap / a.kw, j = tap % a.kw;
    const long HWo = (long)a.Ho * a.Wo, p = (long)y * a.Wo + x;
    const float* op = a.offset[pz] + (long)b * a.offset_bs + (long)g * 2 * T * HWo;
    const float offset_h = op[(long)(2 * tap) * HWo + p];
    const float offset_w = op[(long)(2 * tap + 1) * HWo + p];
    const float m = a.mask[pz][(long)b * a.mask_bs + ((long)g * T + tap) * HWo + p];
    const float h_im = (float)(y * a.stride - a.pad + i * a.dil) + offset_h;
    const float w_im = (float)(x * a.stride - a.pad + j * a.dil) + offset_w;
    const int cpg = a.C / a.dg, H = a.H, W = a.W;
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
    const float* imb = a.im[pz] + (long)b * a.im_bs[pz];
    float* colb = a.col + (long)(pz * a.B + b) * a.C * T * HWo;
    for (int cc = 0; cc < cpg; ++cc) {
        const int c = g * cpg + cc;
        float val = 0.f;
        if (inside) {
            const float* ip = imb + (long)c * HW;
            const float a1 = v1 ? ip[o1] : 0.f, a2 = v2 ? ip[o1 + 1] : 0.f;
            const float a3 = v3 ? ip[o1 + W] : 0.f, a4 = v4 ? ip[o1 + W + 1] : 0.f;
            val = (w1 * a1 + w2 * a2 + w3 * a3 + w4 * a4);
        }
        colb[((long)c * T + tap) * HWo + p] = val * m;
    }
}

extern "C" int motif_dcn_v2_fwd_multi(int P, const float* const* input, const long* input_bs, const float* const* offset,
                                      const float* const* mask, const float* const* packed, const float* const* bias,
                                      float* columns, float* const* out, int B, int C, int H, int W, int Cout, int kh, int kw,
                                      int stride, int pad, int dil, int deformable_groups, long offset_bs, long mask_bs,
                                      int act, void* stream) {
    if (P < 1 || P > 4 || !input || !offset || !mask || !packed || !columns || !out) return MOTIF_EINVAL;
    if (B < 1 || C < 1 || deformable_groups < 1 || C % deformable_groups) return MOTIF_EINVAL;
    const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
    const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    const int T = kh * kw;
    const long HWo = (long)Ho * Wo;
    if (!offset_bs) offset_bs = (long)deformable_groups * 2 * T * HWo;
    if (!mask_bs) mask_bs = (long)deformable_groups * T * HWo;
    DcnArgs a;
    for (int i = 0; i < 4; ++i) {
        const int j = i < P ? i : 0;
        if (!input[j] || !offset[j] || !mask[j] || !packed[j] || !out[j]) return MOTIF_EINVAL;
        a.im[i] = input[j]; a.offset[i] = offset[j]; a.mask[i] = mask[j];
        a.im_bs[i] = (input_bs && input_bs[j]) ? input_bs[j] : (long)C * H * W;
    }
    a.col = columns; a.B = B; a.C = C; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.kh = kh; a.kw = kw;
    a.stride = stride; a.pad = pad; a.dil = dil; a.dg = deformable_groups; a.offset_bs = offset_bs; a.mask_bs = mask_bs;
    dim3 grid(cdiv(Wo, 64), Ho, P * B * deformable_groups * T);
    dcn_im2col_kernel<<<grid, 64, 0, (hipStream_t)stream>>>(a);
    MOTIF_LAUNCH_CHECK();
    MotifConvDesc d = {};
    d.N = B; d.H = Ho; d.W = Wo; d.C0 = C * T; d.C1 = 0; d.Cout = Cout; d.KH = 1; d.KW = 1;
    d.stride = 1; d.pad = 0; d.dil = 1; d.groups = 1; d.pad_mode = 0; d.act = act; d.act2 = 0; d.act_split = 0; d.res_mode = 0;
    const float* cols[4];
    for (int i = 0; i < P; ++i) cols[i] = columns + (long)i * B * C * T * HWo;
    return motif_conv2d_fwd_multi(&d, P, cols, nullptr, packed, bias, nullptr, out, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int motif_dcn_v2_fwd(const float* input, const float* offset, const float* mask, const float* packed,
                                const float* bias, float* columns, float* out,
                                int B, int C, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil,
                                int deformable_groups, long offset_bs, long mask_bs, int act, void* stream) {
    return motif_dcn_v2_fwd_multi(1, &input, nullptr, &offset, &mask, &packed, &bias, columns, &out, B, C, H, W, Cout, kh, kw,
                                  stride, pad, dil, deformable_groups, offset_bs, mask_bs, act, stream);
}

// Modulated deformable convolution v2 forward (gfx950): deformable im2col, then the fp32-MFMA
// implicit-GEMM engine as a 1x1 convolution over the C*kh*kw column channels (bias + activation fused).
// The sampling position, its four corner offsets and bilinear weights are computed once per
// (deformable group, tap, pixel) and reused for the group's channels; the reference recomputes them per
// channel (dcn_v2_im2col_cuda.cu:125-194).
#include "common.h"
#include <stdlib.h>

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

// ================================================================================================
// Fused DCNv2 forward (3x3, stride 1, pad 1, dilation 1 -- the only configuration on the MoTIF path):
// the deformable im2col is produced straight into LDS, chunk by chunk, and consumed by the fp32 MFMA
// loop; the [B, C*9, H*W] `columns` tensor (1 GB per launch at the LSTM's L1 level) never exists.
//
// Block = 8 waves = 8 output rows x 32 columns x 64 output channels.  A thread owns ONE output pixel and
// every second tap (pairs e = tid + 512 j): per deformable group it computes the sampling geometry of its
// pairs once (offsets, mask, corner validity, bilinear weights -- dcn_v2_im2col_cuda.cu:166-187,25-54), then
// for each half-group (4 channels = one reduction chunk of 36 K-rows) prefetches the 4 corner values per
// (pair, channel) into registers while the previous chunk is multiplied, blends them after the MFMA phase and
// writes col[(channel pair, tap, half)][pixel] to the other LDS buffer -- the same K order as the conv
// engine's packed 3x3 weights, so B operands are read with immediate offsets.
// ================================================================================================
struct DcnFusedArgs {
    const float* im[4]; const float* offset[4]; const float* mask[4]; const float* wp[4]; const float* bias[4]; float* out[4];
    long im_bs[4];
    long offset_bs, mask_bs;
    int B, C, H, W, Cout, dg, act, ncg, Kpad, tiles_x;
    int front_pad;
};

#define DF_PAIRS 5          // ceil(9 taps * 256 pixels / 512 threads)
#define DF_CH 4             // channels per chunk
#define DF_ROWS (DF_CH * 9) // K rows per chunk

// WAVES = output rows per block (one wave per row of 32 pixels): 8 -> one 92 KB block per CU; 4 -> two 55 KB blocks per
// CU whose gather latencies and MFMA stretches overlap each other.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void dcn_fused_kernel(DcnFusedArgs a) {
    constexpr int NPX = 32 * WAVES, NT = 64 * WAVES;       // pixels per tile, threads
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int WN = 64;
    float* col0 = smem + a.front_pad;                    // [2][DF_ROWS][NPX]  (front_pad: debugging aid)
    float* wl0 = col0 + 2 * DF_ROWS * NPX;               // [2][DF_ROWS][WN]
    float* bias_s = wl0 + 2 * DF_ROWS * WN;              // [WN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
    const int cg = blockIdx.y;
    const int pz = blockIdx.z / a.B, b = blockIdx.z - pz * a.B;
    const int H = a.H, W = a.W;
    const long HW = (long)H * W;
    const float* imb = a.im[pz] + (long)b * a.im_bs[pz];
    const float* offb = a.offset[pz] + (long)b * a.offset_bs;
    const float* mskb = a.mask[pz] + (long)b * a.mask_bs;
    const float* wbase = a.wp[pz] + (long)cg * a.Kpad * WN;
    const int cpg = a.C / a.dg;

    // this thread's pixel and taps
    const int pxl = tid % NPX;                           // pixel index in the tile: row pxl>>5, column pxl&31
    const int oy = ty * WAVES + (pxl >> 5), ox = tx * 32 + (pxl & 31);
    const bool pix_ok = oy < H && ox < W;
    const long p = (long)oy * W + ox;
    const int tap0 = tid / NPX;                          // taps tap0, tap0+2, ...

    if (tid < WN) {
        const int col = cg * WN + tid;
        bias_s[tid] = (a.bias[pz] && col < a.Cout) ? a.bias[pz][col] : 0.f;
    }

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // geometry of the current deformable group
    int go1[DF_PAIRS];                                   // offset of the (h_low, w_low) corner, or -1 if the tap is dead
    int gfl[DF_PAIRS];                                   // corner validity bits
    float gw1[DF_PAIRS], gw2[DF_PAIRS], gw3[DF_PAIRS], gw4[DF_PAIRS], gm[DF_PAIRS];
    auto geometry = [&](int g) {
#pragma unroll
        for (int j = 0; j < DF_PAIRS; ++j) {
            const int tap = tap0 + 2 * j;
            go1[j] = -1; gfl[j] = 0; gw1[j] = gw2[j] = gw3[j] = gw4[j] = 0.f; gm[j] = 0.f;
            if (tap < 9 && pix_ok) {
                const float* op = offb + (long)g * 18 * HW;
                const float offset_h = op[(long)(2 * tap) * HW + p];
                const float offset_w = op[(long)(2 * tap + 1) * HW + p];
                gm[j] = mskb[((long)g * 9 + tap) * HW + p];
                const float h_im = (float)(oy - 1 + tap / 3) + offset_h;
                const float w_im = (float)(ox - 1 + tap % 3) + offset_w;
                if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
                    const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
                    const int h_high = h_low + 1, w_high = w_low + 1;
                    const float lh = h_im - h_low, lw = w_im - w_low, hh = 1 - lh, hw = 1 - lw;
                    gfl[j] = (h_low >= 0 && w_low >= 0 ? 1 : 0) | (h_low >= 0 && w_high <= W - 1 ? 2 : 0) |
                             (h_high <= H - 1 && w_low >= 0 ? 4 : 0) | (h_high <= H - 1 && w_high <= W - 1 ? 8 : 0);
                    gw1[j] = hh * hw; gw2[j] = hh * lw; gw3[j] = lh * hw; gw4[j] = lh * lw;
                    go1[j] = h_low * W + w_low;          // may be "negative-ish" only where the flag is off
                }
            }
        }
    };

    float pre[DF_PAIRS][DF_CH][4];
    constexpr int NWR = (DF_ROWS * 64 / 4 + NT - 1) / NT;
    f32x4 wreg[NWR];
    auto issue = [&](int c0) {                           // corner values of channels c0..c0+3 for my pairs; weight rows
#pragma unroll
        for (int j = 0; j < DF_PAIRS; ++j) {
#pragma unroll
            for (int cl = 0; cl < DF_CH; ++cl) {
                const float* ip = imb + (long)(c0 + cl) * HW + go1[j];
                const int fl = gfl[j];
                pre[j][cl][0] = (fl & 1) ? ip[0] : 0.f;
                pre[j][cl][1] = (fl & 2) ? ip[1] : 0.f;
                pre[j][cl][2] = (fl & 4) ? ip[W] : 0.f;
                pre[j][cl][3] = (fl & 8) ? ip[W + 1] : 0.f;
            }
        }
        const f32x4* src = (const f32x4*)(wbase + (long)c0 * 9 * WN);       // rows (c0/2*9*2 ...) = c0*9
#pragma unroll
        for (int j = 0; j < NWR; ++j) {
            const int i = tid + NT * j;
            if (i < DF_ROWS * WN / 4) wreg[j] = src[i];
        }
    };
    auto commit = [&](int buf) {
        float* col = col0 + buf * DF_ROWS * NPX;
#pragma unroll
        for (int j = 0; j < DF_PAIRS; ++j) {
            const int tap = tap0 + 2 * j;
            if (tap < 9) {
#pragma unroll
                for (int cl = 0; cl < DF_CH; ++cl) {
                    const float val = (gw1[j] * pre[j][cl][0] + gw2[j] * pre[j][cl][1] + gw3[j] * pre[j][cl][2] + gw4[j] * pre[j][cl][3]);
                    col[(((cl >> 1) * 9 + tap) * 2 + (cl & 1)) * NPX + pxl] = val * gm[j];
                }
            }
        }
        f32x4* w4 = (f32x4*)(wl0 + buf * DF_ROWS * WN);
#pragma unroll
        for (int j = 0; j < NWR; ++j) {
            const int i = tid + NT * j;
            if (i < DF_ROWS * WN / 4) w4[i] = wreg[j];
        }
    };

    const int nchunks = a.C / DF_CH;
    geometry(0);
    issue(0);
    commit(0);
    __syncthreads();
    int cur = 0;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cnext = (ch + 1) * DF_CH;
        const bool more = ch + 1 < nchunks;
        if (more) {
            if (cnext % cpg == 0) geometry(cnext / cpg);      // next chunk starts a new deformable group
            issue(cnext);
        }
        const float* colb = col0 + cur * DF_ROWS * NPX + half * NPX + wave * 32 + l31;
        const float* wl = wl0 + cur * DF_ROWS * WN + half * WN + l31;
#pragma unroll
        for (int cp = 0; cp < DF_CH / 2; ++cp) {
            float bv[9], av[9][2];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                bv[t] = colb[(cp * 9 + t) * 2 * NPX];
                av[t][0] = wl[(cp * 9 + t) * 2 * WN];
                av[t][1] = wl[(cp * 9 + t) * 2 * WN + 32];
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][0], bv[t], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][1], bv[t], acc[1], 0, 0, 0);
            }
        }
        if (more) commit(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // epilogue (C/D layout: column = pixel lane&31, row = (r&3) + 8*(r>>2) + 4*half)
    const int eox = tx * 32 + l31, eoy = ty * WAVES + wave;
    if (eox >= W || eoy >= H) return;
    float* op = a.out[pz] + ((long)b * a.Cout + (long)cg * WN) * HW + (long)eoy * W + eox;
    const int climit = a.Cout - cg * WN;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = acc[i][r] + bias_s[col];
            if (a.act == MOTIF_ACT_LRELU) v = v > 0.f ? v : 0.1f * v;
            else if (a.act == MOTIF_ACT_RELU) v = v > 0.f ? v : 0.f;
            if (col < climit) op[(long)col * HW] = v;
        }
}

extern "C" int motif_dcn_v2_fused_fwd_multi(int P, const float* const* input, const long* input_bs, const float* const* offset,
                                            const float* const* mask, const float* const* packed3x3, const float* const* bias,
                                            float* const* out, int B, int C, int H, int W, int Cout, int deformable_groups,
                                            long offset_bs, long mask_bs, int act, void* stream) {
    if (P < 1 || P > 4 || !input || !offset || !mask || !packed3x3 || !out || B < 1) return MOTIF_EINVAL;
    if (deformable_groups < 1 || C % deformable_groups || (C / deformable_groups) % DF_CH || (long)H * W >= (1L << 30)) return MOTIF_ELIMIT;
    if (act != MOTIF_ACT_NONE && act != MOTIF_ACT_LRELU && act != MOTIF_ACT_RELU) return MOTIF_ELIMIT;
    const long HW = (long)H * W;
    DcnFusedArgs a;
    for (int i = 0; i < 4; ++i) {
        const int j = i < P ? i : 0;
        if (!input[j] || !offset[j] || !mask[j] || !packed3x3[j] || !out[j]) return MOTIF_EINVAL;
        a.im[i] = input[j]; a.offset[i] = offset[j]; a.mask[i] = mask[j]; a.wp[i] = packed3x3[j];
        a.bias[i] = bias ? bias[j] : nullptr; a.out[i] = out[j];
        a.im_bs[i] = (input_bs && input_bs[j]) ? input_bs[j] : (long)C * HW;
    }
    a.offset_bs = offset_bs ? offset_bs : (long)deformable_groups * 18 * HW;
    a.mask_bs = mask_bs ? mask_bs : (long)deformable_groups * 9 * HW;
    a.B = B; a.C = C; a.H = H; a.W = W; a.Cout = Cout; a.dg = deformable_groups; a.act = act;
    a.ncg = (Cout + 63) / 64;
    a.Kpad = 2 * 9 * ((C + 1) / 2);
    a.tiles_x = (W + 31) / 32;
    // 8-wave blocks (one 92 KB block per CU) by default.  The 4-wave variant (two 55 KB blocks per CU, ~12 % faster on the
    // large maps) is opt-in only: run beside conv_split_kernel<*,4> blocks on the same CU it produced sporadic wrong tiles
    // (tools/dbg/race_dcn4.py; serial runs and every other pairing are bit-reproducible) -- not understood yet, so not used.
    int waves = 8;
    if (const char* ev = getenv("MOTIF_DCN_WAVES")) waves = atoi(ev) == 4 ? 4 : 8;
    a.front_pad = getenv("MOTIF_DCN_FRONT_PAD") ? atoi(getenv("MOTIF_DCN_FRONT_PAD")) : 0;
    const int back_pad = getenv("MOTIF_DCN_BACK_PAD") ? atoi(getenv("MOTIF_DCN_BACK_PAD")) : 0;
    const size_t lds = (size_t)(2 * DF_ROWS * 32 * waves + 2 * DF_ROWS * 64 + 64 + a.front_pad + back_pad) * 4;
    dim3 grid(a.tiles_x * ((H + waves - 1) / waves), a.ncg, P * B);
    hipError_t e;
    if (waves == 8) {
        e = hipFuncSetAttribute((const void*)dcn_fused_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        dcn_fused_kernel<8><<<grid, 512, lds, (hipStream_t)stream>>>(a);
    } else {
        e = hipFuncSetAttribute((const void*)dcn_fused_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        dcn_fused_kernel<4><<<grid, 256, lds, (hipStream_t)stream>>>(a);
    }
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

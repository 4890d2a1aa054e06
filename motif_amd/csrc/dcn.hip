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

__global__ __launch_bounds__(64) MOTIF_SCALAR_F32 void dcn_im2col_kernel(DcnArgs a) {
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
    unsigned* status;        // range status word (include/motif_hip.h): the two-part fp16 form ORs bit 0 into it on a non-finite accumulator
};

typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
#define DF_PAIRS 5          // ceil(9 taps * 256 pixels / 512 threads)
#define DF_CH 4             // channels per chunk
#define DF_ROWS (DF_CH * 9) // K rows per chunk

typedef __attribute__((ext_vector_type(8))) __bf16 dcn_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 dcn_bf16x2;
typedef unsigned dcn_u32x4 __attribute__((ext_vector_type(4)));
__device__ constexpr int DCN_PW[6] = {2, 0, 1, 1, 0, 0};      // (weight part, value part) of the six products, small terms first
__device__ constexpr int DCN_PX[6] = {0, 2, 1, 0, 1, 0};
__device__ __forceinline__ unsigned dcn_pk_bf16(float a, float b) {
    dcn_bf16x2 p = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ void dcn_split8(const float (&v)[8], dcn_u32x4 (&out)[3]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float x0 = v[2 * q], x1 = v[2 * q + 1];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned pk = dcn_pk_bf16(x0, x1);
            out[p][q] = pk;
            if (p < 2) { x0 -= __builtin_bit_cast(float, pk << 16); x1 -= __builtin_bit_cast(float, pk & 0xffff0000u); }
        }
    }
}


// Two-part fp16 form of the window kernel's GEMM (round 4, MotifConvDesc.mma = 7 semantics: conv_wino.hip's header): hi = rne_fp16(x),
// lo = rne_fp16(x - hi), three products instead of six; the weights are packed times 2^8, the epilogue multiplies by 2^-8.  The sampled
// values are bilinear blends of feature values times a sigmoid mask: the range of the features.  Round 5: the low activation part is
// stored times 2^11 and multiplied by 2^-11 x the high weight part (conv_wino.hip, WOrder<2>): a normal fp16 number whenever hi is one.
typedef _Float16 dcn_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 dcn_f16x8 __attribute__((ext_vector_type(8)));
constexpr float kDcnF16Scale = 256.f;
__device__ constexpr int DCN_PW2[3] = {1, 0, 0};
__device__ constexpr int DCN_PX2[3] = {0, 1, 0};
__device__ __forceinline__ unsigned dcn_pk_f16(float a, float b) { const dcn_f16x2 h = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(unsigned, h); }
__device__ __forceinline__ float dcn_sub_lo(float x, unsigned pk) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x)); return r; }
__device__ __forceinline__ float dcn_sub_hi(float x, unsigned pk) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x)); return r; }
constexpr float kDcnLoScale = 2048.f;
__device__ __forceinline__ unsigned dcn_pk_mul(unsigned a, dcn_f16x2 c) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(dcn_f16x2, a) * c); }
__device__ __forceinline__ void dcn_split8_f16(const float (&v)[8], dcn_u32x4 (&out)[2], float s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        out[0][q] = dcn_pk_f16(v[2 * q], v[2 * q + 1]);
        const float r0 = dcn_sub_lo(v[2 * q], out[0][q]), r1 = dcn_sub_hi(v[2 * q + 1], out[0][q]);
        unsigned d;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(d) : "v"(r0), "s"(s));      // rne((x - hi) * 2^11), one rounding
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]\n\ts_nop 0" : "+v"(d) : "v"(r1), "s"(s));      // (one wait state behind a high-half write: siren_split.hip)
        out[1][q] = d;
    }
}

// weight [Cout, C, 3, 3] fp32 -> per cout group of 64 and per 4-channel chunk: A fragments [k-step 3][part 3][tile 2][lane][8]
// bf16 of the chunk's K-rows R = 16s + 8*(lane>>5) + e, R = ((cl>>1)*9 + tap)*2 + (cl&1) (the im2col row order of the fused
// kernel), rows 36..47 zero.
template <int NP>
__global__ void dcn_split_pack_kernel(const float* w, unsigned short* wp, int Cout, int C, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), t = (int)((i >> 9) & 1);
    long r = i >> 10;
    const int part = (int)(r % NP); r /= NP;
    const int ks = (int)(r % 3); r /= 3;
    const int nch = C / DF_CH;
    const int chunk = (int)(r % nch);
    const int cgi = (int)(r / nch);
    const int col = cgi * 64 + t * 32 + (lane & 31);
    const int R = 16 * ks + 8 * (lane >> 5) + e;
    float v = 0.f;
    if (R < DF_ROWS && col < Cout) {
        const int cp = R / 18, tap = (R % 18) >> 1, hf = R & 1;
        const int c = chunk * DF_CH + 2 * cp + hf;
        v = w[((long)col * C + c) * 9 + tap];
    }
    unsigned short out = 0;
    if constexpr (NP == 2) {
        double vd = (double)v * (double)kDcnF16Scale;
        for (int p = 0; p <= part; ++p) {
            const _Float16 h = (_Float16)(float)vd;
            out = __builtin_bit_cast(unsigned short, h);
            vd -= (double)(float)h;
        }
    } else {
        for (int p = 0; p <= part; ++p) {
            const unsigned pk = dcn_pk_bf16(v, 0.f);
            out = (unsigned short)(pk & 0xffffu);
            v -= __builtin_bit_cast(float, pk << 16);
        }
    }
    wp[i] = out;
}

extern "C" long motif_dcn_split_pack(const float* weight, float* packed, int Cout, int C, void* stream) {
    if (Cout < 1 || C < DF_CH || C % DF_CH) return MOTIF_EINVAL;
    const long ncg = (Cout + 63) / 64, nch = C / DF_CH;
    // blob = [three-part bf16 block | two-part fp16 block]: the kernel and arithmetic are chosen per launch (mma = 6 / 7)
    const long floats3 = ncg * nch * 3 * 3 * 2 * 64 * 4, floats2 = ncg * nch * 3 * 2 * 2 * 64 * 4, floats = floats3 + floats2;
    if (!packed) return floats;
    if (!weight) return MOTIF_EINVAL;
    dcn_split_pack_kernel<3><<<cdiv(floats3 * 2, 256), 256, 0, (hipStream_t)stream>>>(weight, (unsigned short*)packed, Cout, C, floats3 * 2);   // 16-bit elements
    dcn_split_pack_kernel<2><<<cdiv(floats2 * 2, 256), 256, 0, (hipStream_t)stream>>>(weight, (unsigned short*)(packed + floats3), Cout, C, floats2 * 2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(long)e - 1000;
    return floats;
}

// WAVES = output rows per block (one wave per row of 32 pixels): 8 -> one 92 KB block per CU; 4 -> two 55 KB blocks per
// CU whose gather latencies and MFMA stretches overlap each other.
// SPLIT: the GEMM runs on the bf16 matrix cores with the 3-way split arithmetic of conv_split.hip (fp32-equivalent): the
// im2col values stay fp32 in LDS, a wave reads the 8 K-rows of its half (16s + 8*half + e) and splits them into three
// packed-bf16 parts on the fly; the weights come split at pack time as MFMA A fragments
// [chunk][k-step 3][part 3][cout tile 2][lane][8 bf16] (the chunk's 36 K-rows padded to 48).  The fp32 MFMA occupies
// the vector ALU that the gather / blend arithmetic needs; the bf16 one does not.
template <int WAVES, bool SPLIT>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void dcn_fused_kernel(DcnFusedArgs a) {
    constexpr int NPX = 32 * WAVES, NT = 64 * WAVES;       // pixels per tile, threads
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int WN = 64;
    constexpr int WCH = SPLIT ? 3 * 3 * 2 * 64 * 4 : DF_ROWS * WN;   // floats of weights per chunk in LDS / in the packed tensor
    float* col0 = smem + a.front_pad;                    // [2][DF_ROWS][NPX]  (front_pad: debugging aid)
    float* wl0 = col0 + 2 * DF_ROWS * NPX;               // [2][WCH]
    float* bias_s = wl0 + 2 * WCH;                       // [WN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
    const int cg = blockIdx.y;
    const int pz = blockIdx.z / a.B, b = blockIdx.z - pz * a.B;
    const int H = a.H, W = a.W;
    const long HW = (long)H * W;
    const float* imb = a.im[pz] + (long)b * a.im_bs[pz];
    const float* offb = a.offset[pz] + (long)b * a.offset_bs;
    const float* mskb = a.mask[pz] + (long)b * a.mask_bs;
    const float* wbase = SPLIT ? a.wp[pz] + (long)cg * (a.C / DF_CH) * WCH : a.wp[pz] + (long)cg * a.Kpad * WN;
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

    // geometry of the current deformable group.  Per (pixel, tap): the two corner ROWS as 8-byte pairs -- offset of the
    // left element (clamped into [0, HW-2], so the unconditional dwordx2 load stays inside the plane) and the bilinear
    // weight of each of the two loaded elements; an invalid corner (dcn_v2_im2col_cuda.cu:37-48) has weight 0, which
    // reproduces its zero contribution without predicating the loads (half the load instructions, no branches).
    int got[DF_PAIRS], gob[DF_PAIRS];
    float gtx[DF_PAIRS], gty[DF_PAIRS], gbx[DF_PAIRS], gby[DF_PAIRS], gm[DF_PAIRS];
    auto geometry = [&](int g) {
#pragma unroll
        for (int j = 0; j < DF_PAIRS; ++j) {
            const int tap = tap0 + 2 * j;
            got[j] = gob[j] = 0; gtx[j] = gty[j] = gbx[j] = gby[j] = 0.f; gm[j] = 0.f;
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
                    const bool vt = h_low >= 0, vb = h_high <= H - 1, vl = w_low >= 0, vr = w_high <= W - 1;
                    float tx_ = (vt && vl) ? hh * hw : 0.f, ty_ = (vt && vr) ? hh * lw : 0.f;
                    float bx_ = (vb && vl) ? lh * hw : 0.f, by_ = (vb && vr) ? lh * lw : 0.f;
                    int ot = vt ? h_low * W + w_low : 0, ob = vb ? h_high * W + w_low : 0;
                    const int last = (int)HW - 1;
                    if (ot < 0) { ot = 0; tx_ = ty_; ty_ = 0.f; } else if (ot >= last) { ot = last - 1; ty_ = tx_; tx_ = 0.f; }
                    if (ob < 0) { ob = 0; bx_ = by_; by_ = 0.f; } else if (ob >= last) { ob = last - 1; by_ = bx_; bx_ = 0.f; }
                    got[j] = ot; gob[j] = ob; gtx[j] = tx_; gty[j] = ty_; gbx[j] = bx_; gby[j] = by_;
                }
            }
        }
    };

    float pre[DF_PAIRS][DF_CH][4];
    constexpr int NWR = (WCH / 4 + NT - 1) / NT;
    static_assert((WCH / 4) % 64 == 0, "a chunk's weights are whole 1 KiB wave pieces");
    // The chunk's weights go global -> LDS directly (global_load_lds_dwordx4: wave-uniform LDS base + lane * 16, the
    // packed chunk is contiguous, so the copy is linear): no staging registers and no ds_write pass.  The staging
    // registers were what pushed the bf16x3 form past 256 VGPRs into scratch (3 / 11 spilled VGPRs for 8 / 4 waves).
    auto issue = [&](int c0, int buf) {                  // corner values of channels c0..c0+3 for my pairs; weight rows
#pragma unroll
        for (int j = 0; j < DF_PAIRS; ++j) {
#pragma unroll
            for (int cl = 0; cl < DF_CH; ++cl) {
                const float* ip = imb + (long)(c0 + cl) * HW;
                const f32x2u t2 = *(const f32x2u*)(ip + got[j]);         // 4-byte aligned 8-byte loads
                const f32x2u b2 = *(const f32x2u*)(ip + gob[j]);
                pre[j][cl][0] = t2[0]; pre[j][cl][1] = t2[1]; pre[j][cl][2] = b2[0]; pre[j][cl][3] = b2[1];
            }
        }
        const f32x4* src = (const f32x4*)(SPLIT ? wbase + (long)(c0 / DF_CH) * WCH : wbase + (long)c0 * 9 * WN);   // fp32: rows c0*9
        f32x4* w4 = (f32x4*)(wl0 + buf * WCH);
#pragma unroll
        for (int j = 0; j < NWR; ++j) {
            const int i = tid + NT * j;                   // wave-uniform condition: WCH/4 is a multiple of 64
            if (i < WCH / 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i),
                                                 (__attribute__((address_space(3))) void*)(w4 + (i - lane)), 16, 0, 0);
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
                    const float val = (gtx[j] * pre[j][cl][0] + gty[j] * pre[j][cl][1] + gbx[j] * pre[j][cl][2] + gby[j] * pre[j][cl][3]);
                    col[(((cl >> 1) * 9 + tap) * 2 + (cl & 1)) * NPX + pxl] = val * gm[j];
                }
            }
        }
    };

    const int nchunks = a.C / DF_CH;
    geometry(0);
    issue(0, 0);
    commit(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cnext = (ch + 1) * DF_CH;
        const bool more = ch + 1 < nchunks;
        if (more) {
            if (cnext % cpg == 0) geometry(cnext / cpg);      // next chunk starts a new deformable group
            issue(cnext, cur ^ 1);                            // buffer cur^1 was last read before the previous barrier
        }
        if constexpr (SPLIT) {
            const float* colp = col0 + cur * DF_ROWS * NPX + wave * 32 + l31;
            const dcn_u32x4* wfr = (const dcn_u32x4*)(wl0 + cur * WCH) + lane;
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = 16 * ks + 8 * half + e;                      // K-row of this lane's half
                    v[e] = (ks < 2 || r < DF_ROWS) ? colp[(r < DF_ROWS ? r : 0) * NPX] : 0.f;
                    if (ks == 2 && r >= DF_ROWS) v[e] = 0.f;
                }
                dcn_u32x4 x[3];
                dcn_split8(v, x);
                dcn_u32x4 w[2][3];
#pragma unroll
                for (int part = 0; part < 3; ++part)
#pragma unroll
                    for (int t = 0; t < 2; ++t) w[t][part] = wfr[((ks * 3 + part) * 2 + t) * 64];
#pragma unroll
                for (int k = 0; k < 6; ++k)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dcn_bf16x8, w[t][DCN_PW[k]]),
                                                                          __builtin_bit_cast(dcn_bf16x8, x[DCN_PX[k]]), acc[t], 0, 0, 0);
            }
        } else {
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
        }
        if (more) commit(cur ^ 1);
        // LDS-DMA is ordered for other waves' ds_reads only by the issuing wave's vmcnt followed by a barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

// ================================================================================================
// Window form of the fused kernel (bf16x3 engine): the bilinear corner values come out of LDS instead of the texture path.
// The gathers above are what bounds dcn_fused_kernel: 2 x 8-byte scattered loads per (pixel, tap, channel) -- 40 wave-level
// gather instructions per thread and chunk against 36 MFMAs.  Deformable offsets are small where the features are aligned
// well, so a block stages, per 4-channel chunk, the input WINDOW of its tile (tile + DW_R pixels on every side, clamped to
// the image; 16-byte coalesced loads, 4 per thread) in LDS and samples the four corners with LDS reads; a sample whose 2x2
// footprint leaves the window falls back to predicated global loads (same corner values, same blend expression -- the
// result does not depend on which path a sample took).  Pipeline per chunk ch, one barrier:
//   request window(ch+2) -> registers | sample chunk ch+1 from LDS window(ch+1), blend -> registers | MFMA(ch) |
//   write col(ch+1), window(ch+2) to LDS | barrier
// Order matters: vmcnt retires in order, so nothing inside the sampling or MFMA stretches may wait on a global load --
// the window / offset requests of later chunks would be dragged into that wait.  Hence the weights travel global -> LDS by
// LDS-DMA (no register results to wait for; drained by the vmcnt(0) in front of the barrier) and the fallback path is the
// only place that waits.  One 8-wave block per CU: col 2 x 36 KB + window 2 x 18 KB + weights 2 x 18 KB = 147 KB.
// ================================================================================================
#define DW_R 8
#ifdef MOTIF_DCN_LOCKSTEP
static constexpr bool motif_dcn_lockstep = true;
#else
static constexpr bool motif_dcn_lockstep = false;
#endif
#ifdef MOTIF_DCN_TRACE
__device__ long long g_dcn_trace[64 * 8];
extern "C" int motif_debug_dcn_trace(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dcn_trace), sizeof(long long) * n); }
#define DT(i) do { const long long now_ = __builtin_amdgcn_s_memtime(); tph[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define DT(i)
#endif
template <int WAVES, int NP>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void dcn_win_kernel(DcnFusedArgs a) {
    constexpr int NPX = 32 * WAVES, NT = 64 * WAVES, TH = WAVES;
    constexpr int WWD = 32 + 2 * DW_R, WHT = TH + 2 * DW_R, WSZ = WHT * WWD;      // window of one channel
    constexpr int NWU = DF_CH * WSZ / 4, NWL = (NWU + NT - 1) / NT;                // 16-byte units per chunk, per thread
    constexpr int WCH = 3 * NP * 2 * 64 * 4;            // [k-step 3][part NP][tile 2][lane 64] x 16 bytes
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* col0 = smem;                                  // [2][DF_ROWS][NPX]
    float* win0 = col0 + 2 * DF_ROWS * NPX;              // [2][DF_CH][WSZ]
    float* wl0 = win0 + 2 * DF_CH * WSZ;                 // [2][WCH] weight fragments of a chunk
    float* bias_s = wl0 + 2 * WCH;                       // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int tile_id = xcd_tile_id();                   // an XCD gets a contiguous run of tiles: neighbouring windows overlap 4.5x
    const int tx = tile_id % a.tiles_x, ty = tile_id / a.tiles_x;
    const int cg = blockIdx.y;
    const int pz = blockIdx.z / a.B, b = blockIdx.z - pz * a.B;
    const int H = a.H, W = a.W;
    const long HW = (long)H * W;
    const float* imb = a.im[pz] + (long)b * a.im_bs[pz];
    const float* offb = a.offset[pz] + (long)b * a.offset_bs;
    const float* mskb = a.mask[pz] + (long)b * a.mask_bs;
    // the blob holds the three-part block of ALL cout groups first, then the two-part block
    const float* wbase = a.wp[pz] + (NP == 2 ? (long)a.ncg * (a.C / DF_CH) * (3 * 3 * 2 * 64 * 4) : 0L) + (long)cg * (a.C / DF_CH) * WCH;
    const int cpg = a.C / a.dg;
    const int wy0 = ty * TH - DW_R, wx0 = tx * 32 - DW_R;
    float lo_scale = kDcnLoScale;                        // scalar register: the mix instructions take no literal
    asm volatile("" : "+s"(lo_scale));
    const dcn_f16x2 ws_c = {(_Float16)(1.f / kDcnLoScale), (_Float16)(1.f / kDcnLoScale)};

    const int pxl = tid % NPX;
    const int oy = ty * TH + (pxl >> 5), ox = tx * 32 + (pxl & 31);
    const bool pix_ok = oy < H && ox < W;
    const long p = (long)oy * W + ox;
    const int tap0 = tid / NPX;

    if (tid < 64) {
        const int col = cg * 64 + tid;
        bias_s[tid] = (a.bias[pz] && col < a.Cout) ? a.bias[pz][col] : 0.f;
    }
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // window staging plan (chunk invariant).  LDS layout of a window: [pixel][4 channels] -- the four channels of a corner are ONE
    // ds_read_b128 (4 reads per sample instead of 8 two-dword reads; sampling is LDS-pipeline bound).  A staging thread owns one
    // (window row, 4-pixel group): it loads that group from the chunk's four planes (16-byte coalesced), transposes 4x4 in
    // registers and stores four pixel quads.  Requests are UNCONDITIONAL (surplus threads re-read unit 0 and skip only the LDS
    // store): a predicated load sits in its own exec-mask region, and at such a region's entry hipcc parks a conservative
    // vmcnt(0) -- which waits for the offset / mask requests issued a moment earlier (2.8 k cycles per chunk in the first timeline).
    constexpr int NWG = WSZ / 4, NWS = (NWG + NT - 1) / NT;      // 4-pixel groups of a window, per thread
    int woff[NWS], wdst[NWS];
    bool wlive[NWS], winside[NWS];
#pragma unroll
    for (int k = 0; k < NWS; ++k) {
        const int u = tid + NT * k;
        wlive[k] = u < NWG;
        const int uc = wlive[k] ? u : 0;
        const int wy = uc / (WWD / 4), wu = uc - wy * (WWD / 4);
        int gy = wy0 + wy, gx = wx0 + 4 * wu;
        winside[k] = gy >= 0 && gy < H && gx >= 0 && gx < W;                     // W % 4 == 0: a group is inside or outside as a whole
        gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);                             // clamped request address; cells outside the image are
        gx = gx < 0 ? 0 : (gx > W - 4 ? W - 4 : gx);                             // stored as 0 (v = 0 of dcn_v2_im2col_cuda.cu:37-48), so
        woff[k] = gy * W + gx;                                                   // 0 * Inf cannot appear
        wdst[k] = (wy * WWD + 4 * wu) * DF_CH;                                   // float index of the group's first pixel quad
    }
    f32x4 wreg[NWS][DF_CH];
    auto win_request = [&](int c0) {
        const float* base = imb + (long)c0 * HW;
#pragma unroll
        for (int k = 0; k < NWS; ++k)
#pragma unroll
            for (int cl = 0; cl < DF_CH; ++cl) wreg[k][cl] = *(const f32x4*)(base + (long)cl * HW + woff[k]);
    };
    auto win_commit = [&](int buf) {
        float* dst = win0 + buf * DF_CH * WSZ;
#pragma unroll
        for (int k = 0; k < NWS; ++k)
            if (wlive[k]) {
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    f32x4 q = {wreg[k][0][px], wreg[k][1][px], wreg[k][2][px], wreg[k][3][px]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) q[e] = winside[k] ? q[e] : 0.f;  // a non-finite edge pixel must not leak through a zero weight
                    *(f32x4*)(dst + wdst[k] + 4 * px) = q;
                }
            }
    };

    // geometry of the current deformable group, per (pixel, tap) pair of this thread: window offset of the top-left corner
    // (or -1: footprint outside the window -> global path), the four corner weights (0 for corners outside the image,
    // dcn_v2_im2col_cuda.cu:37-48), the mask, and for the global path the plane offset + validity bits
    int glt[DF_PAIRS], go1[DF_PAIRS], gfl[DF_PAIRS];
    float gw1[DF_PAIRS], gw2[DF_PAIRS], gw3[DF_PAIRS], gw4[DF_PAIRS];
    float roh[DF_PAIRS], row_[DF_PAIRS], rom[DF_PAIRS];     // offsets / mask of the NEXT group, requested a group ahead
    // Everything below is written branch-free on purpose: per-sample `if`s made the compiler emit one exec-mask region (and
    // one LDS round trip) per sample -- 110 saveexec/branch pairs per chunk; dead pairs (tap 9, pixels outside the image,
    // samples outside the image) simply carry zero weights and read a harmless address.
    const long pc = pix_ok ? p : 0;
    auto geom_request = [&](int g) {
        const float* op = offb + (long)g * 18 * HW + pc;
        const float* mp = mskb + (long)g * 9 * HW + pc;
#pragma unroll
        for (int j = 0; j < DF_PAIRS; ++j) {
            const int tap = tap0 + 2 * j < 9 ? tap0 + 2 * j : 8;
            roh[j] = op[(long)(2 * tap) * HW];
            row_[j] = op[(long)(2 * tap + 1) * HW];
            rom[j] = mp[(long)tap * HW];
        }
    };
    auto geometry = [&]() {                              // from the requested offsets / mask
        // pin the first use of the requested values HERE: left free, the compiler sinks part of this arithmetic into the
        // previous iteration, right behind the requests, and parks a vmcnt(0) there (2.8 k cycles per chunk in the timeline)
#pragma unroll
        for (int j = 0; j < DF_PAIRS; ++j) asm volatile("" : "+v"(roh[j]), "+v"(row_[j]), "+v"(rom[j]));
#pragma unroll
        for (int j = 0; j < DF_PAIRS; ++j) {
            const int tap = tap0 + 2 * j;
            const bool live = tap < 9 && pix_ok;
            const float h_im = (float)(oy - 1 + tap / 3) + roh[j];
            const float w_im = (float)(ox - 1 + tap % 3) + row_[j];
            const bool inside = live && h_im > -1 && w_im > -1 && h_im < H && w_im < W;
            const float hs = inside ? h_im : 0.f, ws = inside ? w_im : 0.f;      // safe coordinates for the dead pairs
            const int h_low = (int)floorf(hs), w_low = (int)floorf(ws);
            const int h_high = h_low + 1, w_high = w_low + 1;
            const float lh = hs - h_low, lw = ws - w_low, hh = 1 - lh, hw = 1 - lw;
            const bool vt = inside && h_low >= 0, vb = inside && h_high <= H - 1, vl = w_low >= 0, vr = w_high <= W - 1;
            // the mask is folded into the four corner weights once per (pixel, tap, group) instead of multiplying every sampled channel by it
            // (round 6: 20 multiplies per chunk and thread less; (sum w_i v_i) m -> sum (w_i m) v_i differs by rounding only, tests at 3e-5)
            const float mk = live ? rom[j] : 0.f;
            gw1[j] = (vt && vl) ? hh * hw * mk : 0.f; gw2[j] = (vt && vr) ? hh * lw * mk : 0.f;
            gw3[j] = (vb && vl) ? lh * hw * mk : 0.f; gw4[j] = (vb && vr) ? lh * lw * mk : 0.f;
            const int ry = h_low - wy0, rx = w_low - wx0;
            const bool inwin = !inside || (ry >= 0 && ry <= WHT - 2 && rx >= 0 && rx <= WWD - 2);
            glt[j] = inwin ? (inside ? ry * WWD + rx : 0) : -1;
            go1[j] = h_low * W + w_low;
            gfl[j] = (vt && vl ? 1 : 0) | (vt && vr ? 2 : 0) | (vb && vl ? 4 : 0) | (vb && vr ? 8 : 0);
        }
    };
    float val[DF_PAIRS][DF_CH];
    auto sample = [&](int c0, int buf) {                 // chunk c0's im2col values of my pairs, from window `buf`
        const float* wb = win0 + buf * DF_CH * WSZ;
        bool far = false;
        // all 40 LDS read pairs of the chunk are issued back to back (scheduling fences: left alone the compiler issues
        // them four at a time with a full LDS round trip each), then blended
        // [pair][corner] x 4 channels, in two batches (3 + 2 pairs) so that a batch's 12 / 8 reads are in flight together
        // without pushing the kernel past 256 VGPRs
#pragma unroll
        for (int j0 = 0; j0 < DF_PAIRS; j0 += 3) {
            f32x4 t[3][4];
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const int j = j0 + jj;
                if (j < DF_PAIRS) {
                    const int lt = glt[j] >= 0 ? glt[j] : 0;
                    far |= glt[j] < 0;
                    const f32x4* q = (const f32x4*)wb + lt;
                    t[jj][0] = q[0]; t[jj][1] = q[1]; t[jj][2] = q[WWD]; t[jj][3] = q[WWD + 1];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const int j = j0 + jj;
                if (j < DF_PAIRS) {
#pragma unroll
                    for (int cl = 0; cl < DF_CH; ++cl)
                        val[j][cl] = gw1[j] * t[jj][0][cl] + gw2[j] * t[jj][1][cl] + gw3[j] * t[jj][2][cl] + gw4[j] * t[jj][3][cl];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (__any(far)) {                                // some sample of this wave left the window: predicated global loads
#pragma unroll
            for (int j = 0; j < DF_PAIRS; ++j) {
                if (glt[j] < 0) {
#pragma unroll
                    for (int cl = 0; cl < DF_CH; ++cl) {
                        const float* ip = imb + (long)(c0 + cl) * HW + go1[j];
                        const int fl = gfl[j];
                        const float v1 = (fl & 1) ? ip[0] : 0.f, v2 = (fl & 2) ? ip[1] : 0.f;
                        const float v3 = (fl & 4) ? ip[W] : 0.f, v4 = (fl & 8) ? ip[W + 1] : 0.f;
                        val[j][cl] = gw1[j] * v1 + gw2[j] * v2 + gw3[j] * v3 + gw4[j] * v4;
                    }
                }
            }
        }
    };
    auto col_commit = [&](int buf) {
        float* col = col0 + buf * DF_ROWS * NPX;
#pragma unroll
        for (int j = 0; j < DF_PAIRS; ++j) {
            const int tap = tap0 + 2 * j;
            if (tap < 9) {
#pragma unroll
                for (int cl = 0; cl < DF_CH; ++cl) col[(((cl >> 1) * 9 + tap) * 2 + (cl & 1)) * NPX + pxl] = val[j][cl];
            }
        }
    };
    constexpr int NWR = (WCH / 4 + NT - 1) / NT;
    static_assert((WCH / 4) % 64 == 0, "a chunk's weights are whole 1 KiB wave pieces");
    auto w_request = [&](int chunk, int buf) {           // LDS-DMA: wave-uniform LDS base + lane * 16, linear copy
        // the chunk = 18 pieces of 1 KiB; every wave copies NWR pieces unconditionally -- where there are more wave slots than
        // pieces, the surplus slots copy one of the last pieces again (same bytes to the same place) instead of branching
        constexpr int NPC = WCH / 4 / 64;
        const f32x4* src = (const f32x4*)(wbase + (long)chunk * WCH);
        f32x4* w4 = (f32x4*)(wl0 + buf * WCH);
#pragma unroll
        for (int k = 0; k < NWR; ++k) {
            int piece = wave + WAVES * k;
            if (piece >= NPC) piece = NPC - 1 - (wave & 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(w4 + piece * 64), 16, 0, 0);
        }
    };

#ifdef MOTIF_DCN_TRACE
    long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = __builtin_amdgcn_s_memtime();
#endif
    const int nchunks = a.C / DF_CH;
    // prologue: window(0) -> LDS, then chunk 0 sampled into col(0); window(1) staged for the first loop iteration
    win_request(0);
    geom_request(0);
    w_request(0, 0);
    geometry();
    win_commit(0);
    if (a.dg > 1) geom_request(1);
    if (nchunks > 1) win_request(DF_CH);
    __syncthreads();
    sample(0, 0);
    col_commit(0);
    if (nchunks > 1) win_commit(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    DT(0);
    int cur = 0;
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks, more2 = ch + 2 < nchunks;
        const int cnext = (ch + 1) * DF_CH;
        const bool newg = more && cnext % cpg == 0;           // next chunk starts a new deformable group: its offsets / mask
        if (newg) geometry();                                 // were requested a whole group ago (and drained at the last barrier)
        DT(5);
        if (more2) win_request((ch + 2) * DF_CH);             // window two chunks ahead: sampling + MFMA stretch cover its latency
        // The two waves of a SIMD (w and w + WAVES/2) take the two stretches of an iteration in OPPOSITE order: one samples chunk
        // ch+1 (LDS reads + blend arithmetic) while the other multiplies chunk ch, then they swap.  Both orders are legal inside
        // one barrier interval (col(ch) and window(ch+1) are complete since the last barrier).  In lock step both waves wanted the
        // matrix pipe at the same time and the older one then waited 1.9 k of the 7.4 k cycles of a chunk at the barrier.
        auto mfma_stretch = [&]() {
        {
                const float* colp = col0 + cur * DF_ROWS * NPX + wave * 32 + l31;
                const dcn_u32x4* wfr = (const dcn_u32x4*)(wl0 + cur * WCH) + lane;
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int r = 16 * ks + 8 * half + e;
                        v[e] = (ks < 2 || r < DF_ROWS) ? colp[(r < DF_ROWS ? r : 0) * NPX] : 0.f;
                        if (ks == 2 && r >= DF_ROWS) v[e] = 0.f;
                    }
                    dcn_u32x4 x[NP];
                    if constexpr (NP == 2) dcn_split8_f16(v, x, lo_scale); else dcn_split8(v, x);
                    dcn_u32x4 w[2][NP];
#pragma unroll
                    for (int part = 0; part < NP; ++part)
#pragma unroll
                        for (int t = 0; t < 2; ++t) w[t][part] = wfr[((ks * NP + part) * 2 + t) * 64];
                    dcn_u32x4 whs[2];                     // two-part form: 2^-11 x the high weight part, the partner of the scaled low activation part
                    if constexpr (NP == 2) {
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int q = 0; q < 4; ++q) whs[t][q] = dcn_pk_mul(w[t][0][q], ws_c);
                    }
#pragma unroll
                    for (int k = 0; k < (NP == 2 ? 3 : 6); ++k)
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            if constexpr (NP == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dcn_f16x8, DCN_PX2[k] == 1 ? whs[t] : w[t][DCN_PW2[k]]),
                                                                                                   __builtin_bit_cast(dcn_f16x8, x[DCN_PX2[k]]), acc[t], 0, 0, 0);
                            else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dcn_bf16x8, w[t][DCN_PW[k]]),
                                                                                   __builtin_bit_cast(dcn_bf16x8, x[DCN_PX[k]]), acc[t], 0, 0, 0);
                        }
                }
            }
        };
        const bool mfma_first = WAVES == 8 && __builtin_amdgcn_readfirstlane(wave) >= WAVES / 2 && !motif_dcn_lockstep;
        if (mfma_first) mfma_stretch();
        if (more) sample(cnext, (ch + 1) & 1);                // LDS only (plus the rare fallback)
        DT(6);
        if (newg && cnext / cpg + 1 < a.dg) geom_request(cnext / cpg + 1);
        if (more) w_request(ch + 1, cur ^ 1);                 // LDS-DMA last (hipcc drains vmcnt at the next ordinary load behind a
                                                              // pending LDS-DMA); weight buffer cur^1 was last read before the previous barrier
        DT(1);
        if (!mfma_first) mfma_stretch();
        DT(2);
        if (more) col_commit(cur ^ 1);
        if (more2) win_commit(ch & 1);                        // window(ch+2) replaces window(ch), last read in the previous iteration
        DT(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // LDS-DMA weights landed before anyone passes the barrier
        __syncthreads();
        DT(4);
        cur ^= 1;
    }
#ifdef MOTIF_DCN_TRACE
    if (lane == 0 && wave == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 64)
        for (int i = 0; i < 8; ++i) g_dcn_trace[blockIdx.x * 8 + i] = tph[i];
#endif

    const int eox = tx * 32 + l31, eoy = ty * TH + wave;
    if (eox >= W || eoy >= H) return;
    // range status word: an operand beyond fp16's range is packed as inf and makes every cout of its pixel non-finite
    if constexpr (NP == 2) { if (a.status && __builtin_amdgcn_class(acc[0][0], 0x207)) atomicOr(a.status, 1u); }
    float* op = a.out[pz] + ((long)b * a.Cout + (long)cg * 64) * HW + (long)eoy * W + eox;
    const int climit = a.Cout - cg * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = (NP == 2 ? acc[i][r] * (1.f / kDcnF16Scale) : acc[i][r]) + bias_s[col];
            if (a.act == MOTIF_ACT_LRELU) v = v > 0.f ? v : 0.1f * v;
            else if (a.act == MOTIF_ACT_RELU) v = v > 0.f ? v : 0.f;
            if (col < climit) op[(long)col * HW] = v;
        }
}

extern "C" int motif_dcn_v2_fused_fwd_multi(int P, const float* const* input, const long* input_bs, const float* const* offset,
                                            const float* const* mask, const float* const* packed3x3, const float* const* bias,
                                            float* const* out, int B, int C, int H, int W, int Cout, int deformable_groups,
                                            long offset_bs, long mask_bs, int act, int mma, uint32_t* status, void* stream) {
    if (P < 1 || P > 4 || !input || !offset || !mask || !packed3x3 || !out || B < 1) return MOTIF_EINVAL;
    if (mma != 0 && mma != 6 && mma != 7) return MOTIF_EINVAL;      // 7: two fp16 parts in the window kernel, three bf16 parts in the fallback form
    if (deformable_groups < 1 || C % deformable_groups || (C / deformable_groups) % DF_CH || (long)H * W >= (1L << 30)) return MOTIF_ELIMIT;
    if (act != MOTIF_ACT_NONE && act != MOTIF_ACT_LRELU && act != MOTIF_ACT_RELU) return MOTIF_ELIMIT;
    if ((long)H * W < 2) return MOTIF_ELIMIT;        // the corner-pair loads of the non-window form need two elements per plane
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
    a.status = status;
    // 4-wave blocks (two 74 KB blocks per CU whose gather latencies and MFMA stretches overlap; 11 % faster than one 8-wave
    // block on the 180x320 maps) where the map is large enough to fill the chip that way, else 8-wave blocks.
    // History: in round 1 the 4-wave form was seen to produce sporadic wrong tile rows beside conv_split blocks of another
    // stream and was made opt-in.  Round 2 could not reproduce that on five fresh MI355X boxes -- neither with the exact
    // round-1 tree (d21b251, 2400 concurrent launches, tools/dbg/race_dcn4.py) nor with this kernel; what changed here: the
    // weights go global -> LDS directly, which removed the kernel's scratch use (3 / 11 spilled VGPRs before), and the LDS-DMA
    // is drained by an explicit vmcnt(0) before each barrier.  tests/test_kernels_gpu.py::test_dcn_concurrent_with_conv_split
    // and the model-level run-to-run test keep watching it.  MOTIF_DCN_WAVES=8 / 4 forces a form.  (Round 6 met what that report may have
    // been: a packed fp32 instruction form that comes out wrong in lanes 48..63 beside fp16 / bf16 MFMA kernels of another stream
    // (common.h: MOTIF_SCALAR_F32) -- the built library is scanned for it, and no kernel of this file holds it.)
    // window form (dcn_win_kernel): bf16x3 engine, rows of whole 16-byte units, 32-bit offsets over 4 planes; one 8-wave block
    // per CU (147 KB of LDS)
    const bool split = mma == 6 || mma == 7;
    const bool win_ok = split && (W & 3) == 0 && W >= 4 && (long)DF_CH * HW < (1L << 30) && !motif_opt(MOTIF_OPT_DCN_NOWIN) &&
                        (((unsigned long long)a.im[0] | (unsigned long long)a.im[1] | (unsigned long long)a.im[2] | (unsigned long long)a.im[3]) & 15) == 0 &&
                        ((a.im_bs[0] | a.im_bs[1] | a.im_bs[2] | a.im_bs[3]) & 3) == 0;
    // Non-window path (fp32 engine, W % 4 != 0, unaligned views): 8-wave blocks by default.  The 4-wave form (11 % faster on the
    // 180x320 maps) stays opt-in (option dcn_waves = 4) until the round-1 co-residency report is closed for good: the guard test
    // now forces that form beside conv kernels of another stream in both engines (tests/test_kernels_gpu.py).
    int waves = 8;
    if (const int wv = motif_opt(MOTIF_OPT_DCN_WAVES)) waves = wv == 4 ? 4 : 8;
    a.front_pad = motif_opt(MOTIF_OPT_DCN_FRONT_PAD);
    const int back_pad = motif_opt(MOTIF_OPT_DCN_BACK_PAD);
    const int wch = split ? 3 * 3 * 2 * 64 * 4 : DF_ROWS * 64;
    const size_t lds = (size_t)(2 * DF_ROWS * 32 * waves + 2 * wch + 64 + a.front_pad + back_pad) * 4;
    dim3 grid(a.tiles_x * ((H + waves - 1) / waves), a.ncg, P * B);
    hipError_t e = hipSuccess;
    if (win_ok) {
        const int np = mma == 7 ? 2 : 3;
        const size_t wlds = (size_t)(2 * DF_ROWS * 32 * waves + 2 * DF_CH * (waves + 2 * DW_R) * (32 + 2 * DW_R) + 2 * 3 * np * 2 * 64 * 4 + 64) * 4;
#define MOTIF_LAUNCH_DCNW(WV, NPV)                                                                                        \
    do {                                                                                                                  \
        e = hipFuncSetAttribute((const void*)dcn_win_kernel<WV, NPV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds); \
        if (e != hipSuccess) return (int)e;                                                                               \
        dcn_win_kernel<WV, NPV><<<grid, 64 * WV, wlds, (hipStream_t)stream>>>(a);                                        \
    } while (0)
        if (waves == 8) { if (np == 2) MOTIF_LAUNCH_DCNW(8, 2); else MOTIF_LAUNCH_DCNW(8, 3); }
        else { if (np == 2) MOTIF_LAUNCH_DCNW(4, 2); else MOTIF_LAUNCH_DCNW(4, 3); }
#undef MOTIF_LAUNCH_DCNW
        MOTIF_LAUNCH_CHECK();
        return MOTIF_OK;
    }
#define MOTIF_LAUNCH_DCN(WV, SP)                                                                                          \
    do {                                                                                                                  \
        e = hipFuncSetAttribute((const void*)dcn_fused_kernel<WV, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                                               \
        dcn_fused_kernel<WV, SP><<<grid, 64 * WV, lds, (hipStream_t)stream>>>(a);                                        \
    } while (0)
    if (waves == 8) { if (split) MOTIF_LAUNCH_DCN(8, true); else MOTIF_LAUNCH_DCN(8, false); }
    else { if (split) MOTIF_LAUNCH_DCN(4, true); else MOTIF_LAUNCH_DCN(4, false); }
#undef MOTIF_LAUNCH_DCN
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// Shared by the convolution engines: launch arguments and the fused epilogue (bias, residual, activation, store).
#pragma once
#include "common.h"
#include <type_traits>

#define MOTIF_MAX_PROBLEMS 4
struct ConvArgs {
    // up to MOTIF_MAX_PROBLEMS independent convolutions of identical shape in one launch (blockIdx.z = p*N + n)
    const float* in0[MOTIF_MAX_PROBLEMS]; const float* in1[MOTIF_MAX_PROBLEMS]; const float* wp[MOTIF_MAX_PROBLEMS];
    const float* bias[MOTIF_MAX_PROBLEMS]; const float* res[MOTIF_MAX_PROBLEMS]; float* out[MOTIF_MAX_PROBLEMS];
    long in0_bs[MOTIF_MAX_PROBLEMS], in1_bs[MOTIF_MAX_PROBLEMS], res_bs[MOTIF_MAX_PROBLEMS], out_bs[MOTIF_MAX_PROBLEMS];
    int N, C0, H, W, Ho, Wo;
    int Cin_g, Cout_g, Cout;
    int KH, KW, stride, pad, dil, pad_mode;
    int act, act2, act_split, res_mode;
    int CK, PH, PW, Kpad, tiles_x, ncg;
    int xoff;  // conv_igemm VEC staging: the LDS patch rows start xoff pixels left of the patch (16-byte aligned loads), PW = their pitch
    int dbg;   // tuning aid: 1 = skip staging, 2 = skip MFMA loop
    unsigned* status;   // MotifConvDesc.status: the fp16-form kernels OR bit 0 into it on a non-finite accumulator (may be null)
};

// conv_wino's CHAIN mode (motif_conv2d_chain_fwd): L dependent same-shape layers in one persistent launch.  `layers` is the caller's DEVICE
// table (MotifChainLayer of the header), buffer id 0 = ConvArgs.in0[0] (chain input), 1 = ConvArgs.out[0] (chain output), >= 2 = scratch.
struct ChainLayerDev { const float* packed; const float* bias; int src, dst, res, act_rm; };     // act_rm = act | res_mode << 8
struct ChainArgs {
    const ChainLayerDev* layers;
    float* work; long buf_floats;      // scratch buffer id i at work + (i - 2) * buf_floats, contiguous NCHW
    unsigned* ws;                      // [0] ticket counter, [1] abort word, [2] tiles published so far (progress: the abort clock restarts when it moves), [64 + (layer * N + image) * tile rows + tile row] completed tiles of that row; zeroed before the launch
    long wp_off;                       // floats from a layer's packed blob to its Winograd block
    int L;
};

// Limits of one reduction chunk (host planner keeps to them): patch elements <= PATCH_MAX, packed weight
// floats <= WCHUNK_MAX, so that a whole chunk can be prefetched into registers while the previous one is
// being multiplied (global -> VGPR issue-early, VGPR -> LDS write-late; two LDS buffers, one barrier per chunk).
#define PATCH_MAX 6144
#define WCHUNK_MAX 8192

template <int ACT>
__device__ __forceinline__ float act_c(float v) {
    if constexpr (ACT == MOTIF_ACT_RELU) return v > 0.f ? v : 0.f;
    else if constexpr (ACT == MOTIF_ACT_LRELU) return v > 0.f ? v : 0.1f * v;
    else if constexpr (ACT == MOTIF_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    else if constexpr (ACT == MOTIF_ACT_TANH) return tanhf(v);
    else return v;
}

// The activation / residual mode is block-uniform: dispatch ONCE to a specialised body `run(actf, res_tag)`;
// actf(v, residual, cout) -> stored value, res_tag = std::true_type when a residual tensor is read.
template <bool SPECIALISE = true, class RUN>
__device__ __forceinline__ void conv_act_dispatch(const ConvArgs& a, RUN&& run) {
    auto run_rm = [&](auto actf) {
        if (a.res_mode) run(actf, std::true_type{}); else run(actf, std::false_type{});
    };
    const int rm = a.res_mode;
    if constexpr (!SPECIALISE) {                        // one generic body (rarely taken paths: keeps code size / compile time down)
        const int act = a.act, act2 = a.act2, asplit = a.act_split;
        run_rm([&](float v, float rv, int co) {
            const int ac = (asplit > 0 && co >= asplit) ? act2 : act;
            if (rm == 1) return act_apply(v + rv, ac);
            float y = act_apply(v, ac);
            if (rm == 2) y += rv; else if (rm == 3) { y += rv; y = y > 0.f ? y : 0.f; } else if (rm == 4) y *= rv;
            return y;
        });
        return;
    }
    if (a.act_split > 0) {
        run_rm([&](float v, float rv, int co) {
            const int act = co >= a.act_split ? a.act2 : a.act;
            if (rm == 1) return act_apply(v + rv, act);
            float y = act_apply(v, act);
            if (rm == 2) y += rv; else if (rm == 3) { y += rv; y = y > 0.f ? y : 0.f; } else if (rm == 4) y *= rv;
            return y;
        });
    } else if (rm == 0) {
        switch (a.act) {
            case MOTIF_ACT_RELU: run([](float v, float, int) { return act_c<MOTIF_ACT_RELU>(v); }, std::false_type{}); break;
            case MOTIF_ACT_LRELU: run([](float v, float, int) { return act_c<MOTIF_ACT_LRELU>(v); }, std::false_type{}); break;
            case MOTIF_ACT_SIGMOID: run([](float v, float, int) { return act_c<MOTIF_ACT_SIGMOID>(v); }, std::false_type{}); break;
            case MOTIF_ACT_TANH: run([](float v, float, int) { return act_c<MOTIF_ACT_TANH>(v); }, std::false_type{}); break;
            default: run([](float v, float, int) { return v; }, std::false_type{}); break;
        }
    } else if (rm == 1 && a.act == MOTIF_ACT_NONE) {
        run([](float v, float rv, int) { return v + rv; }, std::true_type{});
    } else if (rm == 1 && a.act == MOTIF_ACT_LRELU) {
        run([](float v, float rv, int) { return act_c<MOTIF_ACT_LRELU>(v + rv); }, std::true_type{});
    } else {
        const int act = a.act;
        run_rm([&](float v, float rv, int) {
            if (rm == 1) return act_apply(v + rv, act);
            float y = act_apply(v, act);
            if (rm == 2) y += rv; else if (rm == 3) { y += rv; y = y > 0.f ? y : 0.f; } else if (rm == 4) y *= rv;
            return y;
        });
    }
}

// Epilogue of one wave: acc[i][j] = 32 couts (tile i of the block's cout group) x 32 pixels of output row oy0+j,
// column ox.  `a_*` are the per-problem pointers/strides, bias_s the LDS copy of the cout group's bias.
template <int NC, int RPW, bool SPECIALISE = true>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[NC][RPW], const float* bias_s, int n, int g, int cg,
                                              int oy0, int ox, int half, const float* a_res, long a_res_bs, float* a_out,
                                              long a_out_bs) {
    constexpr int WN = 32 * NC;
    // epilogue: C/D layout col = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*half (cout within tile).
    // The activation / residual mode is block-uniform: dispatch once, keep the store loop branch-free.
    if (ox >= a.Wo || (a.dbg & 4)) return;
    const long HWo = (long)a.Ho * a.Wo;
    const int cobase = g * a.Cout_g + cg * WN;
    const int climit = a.Cout_g - cg * WN;                 // valid couts in this group
    // fast path (block-uniform): a full cout group and 32-bit element offsets.  Bias comes from LDS as 8 vector reads,
    // the residual values of a tile are requested together, and every access is uniform-base + one per-lane offset.
    const bool fast = climit >= WN && HWo < (1L << 28);
    auto run = [&](auto actf, auto res_tag) {
        constexpr bool RES = decltype(res_tag)::value;
        if (fast) {
            float bv[NC][16];
#pragma unroll
            for (int i = 0; i < NC; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = *(const f32x4*)(bias_s + i * 32 + 8 * q + 4 * half);
#pragma unroll
                    for (int u = 0; u < 4; ++u) bv[i][4 * q + u] = b4[u];
                }
            float* ob = a_out + (long)n * a_out_bs + (long)cobase * HWo;
            const float* rb = RES ? a_res + (long)n * a_res_bs + (long)cobase * HWo : nullptr;
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const int oy = oy0 + j;
                if (oy >= a.Ho) continue;
                const unsigned lane_off = (unsigned)(oy * a.Wo + ox) + 4u * half * (unsigned)HWo;
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    float rv[16];
                    if constexpr (RES) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) rv[r] = (rb + (long)(i * 32 + (r & 3) + 8 * (r >> 2)) * HWo)[lane_off];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int colu = i * 32 + (r & 3) + 8 * (r >> 2);
                        const float v = actf(acc[i][j][r] + bv[i][r], RES ? rv[r] : 0.f, cobase + colu + 4 * half);
                        (ob + (long)colu * HWo)[lane_off] = v;
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int oy = oy0 + j;
            if (oy >= a.Ho) continue;
            const long pixo = (long)oy * a.Wo + ox;
            float* op = a_out + (long)n * a_out_bs + (long)cobase * HWo + pixo;
            const float* rp = RES ? a_res + (long)n * a_res_bs + (long)cobase * HWo + pixo : nullptr;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                float rv[16];
                if constexpr (RES) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int col = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        rv[r] = (col < climit) ? rp[(long)col * HWo] : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    float v = acc[i][j][r] + bias_s[col];
                    v = actf(v, RES ? rv[r] : 0.f, cobase + col);
                    if (col < climit) op[(long)col * HWo] = v;
                }
            }
        }
    };
    conv_act_dispatch<SPECIALISE>(a, run);
}

// Block-cooperative epilogue through LDS: the accumulators of one 32-cout tile are transposed in `scratch`
// ([32 couts][TH*32 pixels], row stride +8 floats so the two half-waves hit different banks), then every thread
// finishes 8 groups of 4 consecutive pixels of one cout: bias, residual (one 16-byte load), activation, one 16-byte
// store -- 16 store instructions per lane instead of 64.  Needs a full cout group, Wo % 4 == 0 and 16-byte aligned
// tensors (checked by the caller, block-uniform); `scratch` must hold 32 * (TH*32 + 8) floats and be free.
template <int NC, int RPW, int WAVES>
__device__ __forceinline__ void conv_epilogue_lds(const ConvArgs& a, f32x16 (&acc)[NC][RPW], const float* bias_s, float* scratch,
                                                  int n, int g, int cg, int ty, int tx, const float* a_res, long a_res_bs,
                                                  float* a_out, long a_out_bs) {
    constexpr int TH = RPW * WAVES, NT = 64 * WAVES, S = TH * 32 + 8, WN = 32 * NC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const long HWo = (long)a.Ho * a.Wo;
    const int cobase = g * a.Cout_g + cg * WN;
    float* ob = a_out + (long)n * a_out_bs + (long)cobase * HWo;
    const float* rb = a.res_mode ? a_res + (long)n * a_res_bs + (long)cobase * HWo : nullptr;
    conv_act_dispatch(a, [&](auto actf, auto res_tag) {
        constexpr bool RES = decltype(res_tag)::value;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            __syncthreads();                                   // scratch free (main loop / previous tile done)
#pragma unroll
            for (int j = 0; j < RPW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    scratch[((r & 3) + 8 * (r >> 2) + 4 * half) * S + (RPW * wave + j) * 32 + l31] = acc[i][j][r];
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int idx = tid + NT * it;
                const int co = idx / (TH * 8), q = idx - co * (TH * 8);
                const int row = q >> 3, col = (q & 7) * 4;
                const int oy = ty * TH + row, ox = tx * 32 + col;
                if (oy >= a.Ho || ox >= a.Wo) continue;
                f32x4 v = *(const f32x4*)(scratch + co * S + row * 32 + col);
                const float b = bias_s[i * 32 + co];
                const long off = (long)(i * 32 + co) * HWo + (long)oy * a.Wo + ox;
                f32x4 rv = {0.f, 0.f, 0.f, 0.f};
                if constexpr (RES) rv = *(const f32x4*)(rb + off);
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = actf(v[u] + b, rv[u], cobase + i * 32 + co);
                *(f32x4*)(ob + off) = v;
            }
        }
    });
}

// split engine (conv_split.hip)
bool motif_conv_split_eligible(const MotifConvDesc* d);
long motif_conv_split_packed_floats(const MotifConvDesc* d);
int motif_conv_split_pack(const MotifConvDesc* d, const float* weight, float* packed, hipStream_t s);
int motif_conv_split_launch(const MotifConvDesc* d, ConvArgs& a, int P, hipStream_t s);
// conv_direct.hip: narrow layers (<= 32 couts, 1x1 / 3x3) on large maps, vector ALU, reads conv_igemm's packed weights
bool motif_conv_direct_eligible(const MotifConvDesc* d, const ConvArgs& a, int P);
int motif_conv_direct_launch(const MotifConvDesc* d, ConvArgs& a, int P, hipStream_t s);
// round-4 kernel (conv_pw.hip): 1x1 layers on the fp16 matrix cores (two-part split, mma = 7), reading the fp32 engine's packed block
bool motif_conv_pw_eligible(const MotifConvDesc* d, const ConvArgs& a, int P);
int motif_conv_pw_launch(const MotifConvDesc* d, ConvArgs& a, int P, hipStream_t s);
// round-6 kernel (conv_ig16.hip): every other layer shape (stride 2, 7x7, dilated, narrow / wide) on the fp16 matrix cores, mma = 7; its
// fp16 fragment block follows the fp32 block in the packed blob
bool motif_conv_ig16_pack_eligible(const MotifConvDesc* d);
long motif_conv_ig16_packed_floats(const MotifConvDesc* d);
int motif_conv_ig16_pack(const MotifConvDesc* d, const float* weight, float* packed16, hipStream_t s);
int motif_conv_ig16_launch(const MotifConvDesc* d, ConvArgs& a, int P, long fp32_block_floats, hipStream_t s);

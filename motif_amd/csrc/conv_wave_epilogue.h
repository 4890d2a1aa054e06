// Shared by the persistent bf16-matrix-core convolution kernels (conv_split2.hip, conv_wino.hip): the XCD-aware block order
// and the wave-local epilogue (accumulators -> wave-private LDS transpose -> bias / residual / activation -> 16-byte stores).
#pragma once
#include "conv_split_common.h"

namespace {
// XCD-aware block order (1-D grid): workgroups are dealt round-robin over the 8 XCDs, so XCD x gets a contiguous run of b'
// (neighbouring tiles -- shared halo rows, the cout groups of one spatial tile -- meet in one L2).  Bijection for every G.
__device__ __forceinline__ int xcd_block_id(int b, int G) {
    const int q = G >> 3, r = G & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

__device__ __forceinline__ f32x4 act_uniform(f32x4 v, int ac) {     // ac is wave-uniform: scalar branches, one path runs
    if (ac == MOTIF_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    } else if (ac == MOTIF_ACT_LRELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.1f * v[e];
    } else if (ac == MOTIF_ACT_SIGMOID) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + expf(-v[e]));
    } else if (ac == MOTIF_ACT_TANH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
    }
    return v;
}

// Wave-local epilogue: the wave's accumulators (32 couts x RW rows x 32 pixels) go through a wave-private LDS scratch of
// [8 couts][RS rows x 32 pixels] floats in 4 * RW / RS passes and leave as 16-byte row pieces: bias, residual (one 16-byte load,
// requested a pass ahead), activation, one 16-byte store.  Activation and residual mode are wave-uniform run-time switches.
// `cbase` = first cout of this wave's 32 in the tensor, `climit` = valid couts from there (partial last group).  Host guarantees:
// Wo % 4 == 0, 16-byte aligned tensors, 32 * Ho * Wo < 2^31, act_split on an 8-cout boundary.
template <int RW, int RS, bool RES>
__device__ __forceinline__ void conv_epilogue_wave(const ConvArgs& a, f32x16 (&acc)[RW], const float* bias_w, float* sc, int lane,
                                                   int cbase, int climit, int oy0, int ox0, const float* rb, float* ob) {
    constexpr int S = RS * 32, NIT = RS, NRP = RW / RS, NPASS = 4 * NRP;      // 8 couts x RS rows x 8 quads = 64 * RS items per pass
    static_assert(RW % RS == 0, "row passes");
    const int half = lane >> 5, l31 = lane & 31;
    const unsigned HWo = (unsigned)(a.Ho * a.Wo);
    const int rm = a.res_mode;
    // per-lane item geometry, recomputed from the lane id where it is used (a handful of VALU operations; held in registers
    // across the passes it cost 15 VGPRs and pushed the kernel into scratch)
    auto item = [&](int it, int& co, int& row, int& col) {
        const int idx = lane + 64 * it;
        co = idx / (RS * 8); const int q = idx - co * (RS * 8);
        row = q >> 3; col = (q & 7) * 4;
    };
    auto okat = [&](int pass, int it) {                               // inside the image and the cout group
        int co, row, col;
        item(it, co, row, col);
        return oy0 + (pass % NRP) * RS + row < a.Ho && ox0 + col < a.Wo && 8 * (pass / NRP) + co < climit;
    };
    auto offat = [&](int pass, int it) {
        int co, row, col;
        item(it, co, row, col);
        return (unsigned)(8 * (pass / NRP) + co) * HWo + (unsigned)((oy0 + (pass % NRP) * RS + row) * a.Wo + ox0 + col);
    };
    f32x4 rv[2][NIT];
    auto load_res = [&](int pass, f32x4 (&dst)[NIT]) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) dst[it] = *(const f32x4*)(rb + (okat(pass, it) ? offat(pass, it) : 0u));   // masked lanes read element 0
    };
    if constexpr (RES) load_res(0, rv[0]);
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const int cq = pass / NRP, r0 = (pass % NRP) * RS;
        if constexpr (RES) { if (pass + 1 < NPASS) load_res(pass + 1, rv[(pass + 1) & 1]); }
        const int ac = (a.act_split > 0 && cbase + 8 * cq >= a.act_split) ? a.act2 : a.act;     // uniform per pass
#pragma unroll
        for (int j = 0; j < RS; ++j)
#pragma unroll
            for (int r3 = 0; r3 < 4; ++r3) sc[(r3 + 4 * half) * S + j * 32 + l31] = acc[r0 + j][4 * cq + r3];
        f32x4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int co, row, col;
            item(it, co, row, col);
            v[it] = *(const f32x4*)(sc + co * S + row * 32 + col);
            const float b = bias_w[8 * cq + co];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[it][e] += b;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (RES && rm == 1) v[it] += rv[pass & 1][it];
            v[it] = act_uniform(v[it], ac);
            if constexpr (RES) {
                if (rm == 2) v[it] += rv[pass & 1][it];
                else if (rm == 3) {
                    v[it] += rv[pass & 1][it];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[it][e] = v[it][e] > 0.f ? v[it][e] : 0.f;
                } else if (rm == 4) v[it] *= rv[pass & 1][it];
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (okat(pass, it)) *(f32x4*)(ob + offat(pass, it)) = v[it];
    }
}
}  // namespace

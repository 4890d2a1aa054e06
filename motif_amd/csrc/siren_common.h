// Shared by the SIREN kernels (siren.hip: fp32 MFMA; siren_split.hip: 3-way bf16 split on the bf16 matrix cores).
#pragma once
#include "common.h"
#include <stdlib.h>

#define SIREN_THREADS 512
#define SIREN_WAVES (SIREN_THREADS / 64)
#ifndef SIREN_TP_IMNET
#define SIREN_TP_IMNET 1
#endif
#ifndef SIREN_TP_FLOW
#define SIREN_TP_FLOW 2
#endif
#ifndef SIREN_TP_SYNTH
#define SIREN_TP_SYNTH 1
#endif

// ---------------------------------------------------------------- sin(x)
// Default: 2-term FMA Cody-Waite reduction by 2*pi, then the hardware v_sin_f32 on the small remainder.
// Measured on MI355X against fp64 (tools/ubench_sin.hip): max abs error 3.8e-7 for |x| <= 300, independent
// of the range (v_sin_f32 on the unreduced argument: 2.7e-6 at |x|<=30, 2.4e-5 at 300; ocml sinf 7e-8).
// -DMOTIF_SIN_PRECISE selects a pi/2 reduction + cephes polynomials (1.2e-7) at ~4x the VALU cost.
__device__ __forceinline__ float sin_cw(float x) {
#ifndef MOTIF_SIN_PRECISE
    // branch-free for every finite x: the fma keeps j*2pi_hi exact, so the reduction error is ~|j|*1e-14 and the
    // result degrades only with the spacing of x itself; inf/nan give nan.
    const float j = rintf(x * 0.15915494309189535f);
    float r = fmaf(j, -6.2831854820251465f, x);
    r = fmaf(j, 1.7484555e-7f, r);                       // -(2*pi - float(2*pi))
    return __builtin_amdgcn_sinf(r * 0.15915494309189535f);
#else
    float r;
    int q;
    if (__builtin_expect(fabsf(x) <= 3.0e4f, 1)) {
        const float j = rintf(x * 0.636619772367581343f);
        r = fmaf(j, -1.57079637050628662109375f, x);
        r = fmaf(j, 4.37113900018624283e-8f, r);
        q = (int)j;
    } else {
        const double xd = (double)x;
        const double j = rint(xd * 0.63661977236758134308);
        double rd = fma(j, -1.57079632679489655800, xd);
        rd = fma(j, -6.12323399573676603587e-17, rd);
        r = (float)rd;
        q = (int)(j - 4.0 * floor(j * 0.25));
    }
    const float z = r * r;
    const float sp = r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
    const float cp = 1.0f - 0.5f * z + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    float v = (q & 1) ? cp : sp;
    return (q & 2) ? -v : v;
#endif
}

__host__ __device__ constexpr int kmap(int s, int hf) { return 8 * (s >> 2) + 4 * hf + (s & 3); }
__host__ __device__ constexpr int pad8(int k) { return (k + 7) & ~7; }
__host__ __device__ constexpr int pad32(int m) { return (m + 31) & ~31; }

// VALU head partial: M outputs, inputs hc[32] are the k-steps [s0, s0+32) of a K-wide layer
template <int M, int KQ, int TP>
__device__ __forceinline__ void valu_head_partial(const float (&hc)[TP][32], float (&sum)[TP][M], const float* wv, int q0, int hf) {
#pragma unroll
    for (int o = 0; o < M; ++o)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 w = *(const f32x4*)(wv + ((o * KQ + q0 + q) * 2 + hf) * 4);
#pragma unroll
            for (int p = 0; p < TP; ++p) {
                sum[p][o] = fmaf(w[0], hc[p][q * 4 + 0], sum[p][o]);
                sum[p][o] = fmaf(w[1], hc[p][q * 4 + 1], sum[p][o]);
                sum[p][o] = fmaf(w[2], hc[p][q * 4 + 2], sum[p][o]);
                sum[p][o] = fmaf(w[3], hc[p][q * 4 + 3], sum[p][o]);
            }
        }
}

struct SirenArgs {
    const float* packed;
    const float* src_lr;      // LR feature stack [imgs_lr, 64, H, W] to gather from
    const float* acc;         // synth: splat accumulator [B*N,133,Q]
    const int32_t* iy; const int32_t* ix;
    const float* rel_y; const float* rel_x;
    const float* times;       // [B*N]
    float* out;
    int NB, N, B, H, W, HH, WW;   // NB = number of HR images processed
    int stagger;                  // start offset of the second wave per SIMD, in s_sleep(127) units (~8k cycles)
    const float* add_lr;          // imnet, split engine: optional LR tensor [NB,64,H,W] gathered like src_lr and ADDED to the output planes
                                  // (the pre-contracted splat's G term: motif_siren_imnet_add_fwd); last so that older initialisers leave it null
    unsigned* status;             // synth, two-part fp16 form: range status word (include/motif_hip.h), bit 0 ORed in when a frame value is non-finite
};

enum { MODE_IMNET = 0, MODE_FLOW = 1, MODE_SYNTH = 2, MODE_SYNTHC = 3 };   // SYNTHC: first layer pre-contracted into the splat (siren_split.hip)

template <int MODE> struct Net;
template <> struct Net<MODE_IMNET> { static constexpr int K0 = 66, NH = 3, HEAD = 64; };
template <> struct Net<MODE_FLOW>  { static constexpr int K0 = 67, NH = 3, HEAD = 3; };
template <> struct Net<MODE_SYNTH> { static constexpr int K0 = 198, NH = 4, HEAD = 3; };
template <> struct Net<MODE_SYNTHC> { static constexpr int K0 = 198, NH = 4, HEAD = 3; };


// siren_split.hip: same networks on the bf16 matrix cores (blob from motif_siren_pack_split, LR partial required)
int motif_siren_split_launch(int mode, const SirenArgs& a, void* stream, int parts = 3);    // parts 2: blob of two fp16 parts (mode + 8 at pack time)

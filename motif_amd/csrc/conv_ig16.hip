// Generic convolution (any kernel size, stride, dilation, padding mode, group count, two-source concat) as an implicit GEMM on the fp16
// matrix cores with the fp32-equivalent two-part arithmetic of conv_wino.hip / conv_pw.hip (MotifConvDesc.mma = 7).  Round 6: the layers
// conv_wino.hip does not take -- stride 2 (the PCD / RAFT / PWC pyramids), 7x7 stems, dilated 3x3 (PWC-Net's refiner), 17..32-cout and
// narrow 3x3 layers on small maps, wide or narrow 1x1 layers -- ran on conv_igemm.hip's fp32 MFMA (v_mfma_f32_32x32x2_f32: the VECTOR rate,
// 157 TFLOP/s peak, 34 achieved on these launches) while everything else had moved to the 16-bit cores (VERDICT r2..r5 "conv_other").
//
//   D[cout][pixel] = sum_k W[cout][k] * im2col[k][pixel],  k = (channel octet o, tap t, channel e of the octet)
//
// Structure = conv_igemm.hip's (block = 8 waves = 8 output rows x 32 columns x 32 NC couts; the input patch of a channel chunk (+ halo,
// zero / reflect padded) is staged in LDS as fp32, two buffers, one barrier per chunk, the next chunk's global loads in flight under the
// current chunk's matrix instructions); what differs is the product:
//   * K order: per channel OCTET, T = KH KW groups of 8 channels at one tap.  A k-step of v_mfma_f32_32x32x16_f16 takes group 2 ks from the
//     lower half-wave and group 2 ks + 1 from the upper one; an odd T leaves the last upper group of an octet empty (zero weights: 10 % of
//     the matrix work at 3x3, 2 % at 7x7).  The order does not depend on how many octets a launch puts into a chunk, so pack and forward
//     agree from the descriptor alone.
//   * B operand: per lane 8 ds_read_b32 from the fp32 patch (channel stride = patch plane, tap offset from a small LDS table, pixel offset =
//     stride * column), split on the fly into hi = rne16(x), lo_s = rne16((x - hi) 2^11) (the scaled low part of round 5: a normal fp16
//     number whenever hi is one).  Channels past Cin of a ragged octet read the chunk's last real plane (finite) against zero weights.
//   * A operand: the weights times 2^8 split into two fp16 parts at PACK time, laid out as A fragments [cout tile][octet][k-step][part][lane]
//     x 8 halves -- a second block behind the layer's fp32 block in the packed blob (motif_conv2d_packed_size / _pack) -- copied linearly
//     global -> registers -> LDS per chunk like conv_igemm's weight slab; 2^-11 x the high part (the partner of lo_s) is formed in registers.
//   * three products per fp32 MAC, fp32 accumulation, x 2^-8 in front of the shared epilogue (bias, residual modes, activations, activation
//     split: conv_common.h); a non-finite accumulator ORs bit 0 into the range status word (an operand beyond fp16's range).
#include "conv_common.h"

typedef _Float16 ig_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 ig_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned ig_u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr float kIgScale = 256.f, kIgLoScale = 2048.f;
constexpr int IG_NT = 512;                               // 8 waves: one output row of the 8 x 32 tile each
constexpr int IG_PATCH_MAX = 10240;                      // floats of one patch buffer (40 KB): 20 register-prefetched elements per thread
constexpr int IG_NE = IG_PATCH_MAX / IG_NT, IG_NE4 = IG_NE / 4;
constexpr int ig_ksc_max(int nc) { return nc == 2 ? 12 : 26; }          // k-steps of a chunk per cout tile (A fragments: 2 KB per k-step and tile)
constexpr int ig_nw(int nc) { return (ig_ksc_max(nc) * 128 + IG_NT - 1) / IG_NT; }

__device__ __forceinline__ unsigned ig_pk(float a, float b) { const ig_f16x2 h = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(unsigned, h); }
__device__ __forceinline__ float ig_sub_lo(float x, unsigned pk) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x)); return r; }
__device__ __forceinline__ float ig_sub_hi(float x, unsigned pk) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x)); return r; }
// activations: hi = rne(x), lo = rne((x - hi) * 2^11) -- one rounding each (v_fma_mixlo / mixhi_f16)
__device__ __forceinline__ void ig_split8_act(const float (&v)[8], ig_u32x4& hi, ig_u32x4& lo, float s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        hi[q] = ig_pk(v[2 * q], v[2 * q + 1]);
        const float r0 = ig_sub_lo(v[2 * q], hi[q]), r1 = ig_sub_hi(v[2 * q + 1], hi[q]);
        unsigned d;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(d) : "v"(r0), "s"(s));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]\n\ts_nop 0" : "+v"(d) : "v"(r1), "s"(s));      // (one wait state behind a high-half write: hipcc does not look into asm -- siren_split.hip)
        lo[q] = d;
    }
}
__device__ __forceinline__ unsigned ig_pk_mul(unsigned a, ig_f16x2 c) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(ig_f16x2, a) * c); }
}  // namespace

// KSO = k-steps per channel octet = ceil(T / 2); noct_g = octets per group; ntile_pad = 32-cout tiles per group in the packed block (even)
template <int NC, bool VEC>
__global__ __launch_bounds__(IG_NT) void conv_ig16_kernel(ConvArgs a, int KSO, int noct_g, int ntile_pad, long wblock_off) {
    extern __shared__ __attribute__((aligned(16))) float ig_smem[];
    constexpr int NT = IG_NT, NW = ig_nw(NC);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int tile_id = xcd_tile_id();
    const int tx = tile_id % a.tiles_x, ty = tile_id / a.tiles_x;
    const int g = blockIdx.y / a.ncg, cg = blockIdx.y % a.ncg;
    const int pz = blockIdx.z / a.N, n = blockIdx.z - pz * a.N;
    const float* a_in0 = a.in0[pz]; const float* a_in1 = a.in1[pz];
    const float* a_bias = a.bias[pz]; const float* a_res = a.res[pz]; float* a_out = a.out[pz];
    const long a_in0_bs = a.in0_bs[pz], a_in1_bs = a.in1_bs[pz], a_res_bs = a.res_bs[pz], a_out_bs = a.out_bs[pz];
    const int PHW = a.PH * a.PW, T = a.KH * a.KW, CK = a.CK;
    const int creal = CK < a.Cin_g ? CK : a.Cin_g;       // channels a patch buffer really holds
    const int patch_elems = (creal * PHW + 3) & ~3;
    const int KSC = (CK >> 3) * KSO;                     // k-steps of a full chunk
    float* patch0 = ig_smem;
    ig_u32x4* wf0 = (ig_u32x4*)(patch0 + 2 * patch_elems);               // [2 buffers][NC tiles][KSC][2 parts][64 lanes]
    const int wq = NC * KSC * 128;
    int* koff = (int*)(wf0 + 2 * wq);
    float* bias_s = (float*)(koff + ((T + 3) & ~3));

    const long HW = (long)a.H * a.W;
    // chunk-invariant staging plan (conv_igemm.hip): element / quad e = tid + NT j of the [creal][PH][PW] patch
    const int iy0 = ty * 8 * a.stride - a.pad, ix0 = tx * 32 * a.stride - a.pad - (VEC ? a.xoff : 0);
    int eoff[VEC ? IG_NE4 : IG_NE];
    int ech[VEC ? IG_NE4 : IG_NE];
    if constexpr (VEC) {
        const int PWQ = a.PW >> 2, PHQ = a.PH * PWQ, nq = creal * PHQ;
#pragma unroll
        for (int j = 0; j < IG_NE4; ++j) {
            const int e = tid + NT * j;
            eoff[j] = -1;
            ech[j] = 1 << 20;
            if (e < nq) {
                const int c = e / PHQ, p = e - c * PHQ;
                const int py = p / PWQ, xq = p - py * PWQ;
                const int iy = iy0 + py, ix = ix0 + 4 * xq;                    // ix % 4 == 0, W % 4 == 0: a quad is inside or outside as a whole
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) eoff[j] = c * (int)HW + iy * a.W + ix;
                ech[j] = c;
            }
        }
    } else {
        const int CKPHW = creal * PHW;
#pragma unroll
        for (int j = 0; j < IG_NE; ++j) {
            const int e = tid + NT * j;
            eoff[j] = -1;
            ech[j] = 1 << 20;
            if (e < CKPHW) {
                const int c = e / PHW, p = e - c * PHW;
                const int py = p / a.PW, px = p - py * a.PW;
                int iy = iy0 + py, ix = ix0 + px;
                if (a.pad_mode == 1) {
                    if (iy < 0) iy = -iy; else if (iy >= a.H) iy = 2 * (a.H - 1) - iy;
                    if (ix < 0) ix = -ix; else if (ix >= a.W) ix = 2 * (a.W - 1) - ix;
                }
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) eoff[j] = c * (int)HW + iy * a.W + ix;
                ech[j] = c;
            }
        }
    }
    for (int t = tid; t < T; t += NT) {
        const int ky = t / a.KW, kx = t - ky * a.KW;
        koff[t] = ky * a.dil * a.PW + kx * a.dil;
    }
    constexpr int WN = 32 * NC;
    float bias_v = 0.f;
    if (tid < WN && a_bias && cg * WN + tid < a.Cout_g) bias_v = a_bias[g * a.Cout_g + cg * WN + tid];

    f32x16 acc[NC][1];
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

    const float* in0n = a_in0 + (long)n * a_in0_bs;
    const float* in1n = a_in1 ? a_in1 + (long)n * a_in1_bs : nullptr;
    // A fragments of this block's cout tiles: [group][tile][octet][k-step][part][lane] x 16 bytes
    const ig_u32x4* wtile = (const ig_u32x4*)(a.wp[pz] + wblock_off) + ((long)(g * ntile_pad + cg * NC) * noct_g) * KSO * 128;
    const long tile_stride = (long)noct_g * KSO * 128;
    const int pix = wave * a.stride * a.PW + l31 * a.stride + (VEC ? a.xoff : 0);

    float pre[VEC ? 1 : IG_NE];
    f32x4 pre4[VEC ? IG_NE4 : 1];
    ig_u32x4 wreg[NC][NW];
    auto issue = [&](int c0) {            // global -> registers for the chunk starting at channel c0 (a multiple of 8); -> its octets
        const int gch0 = g * a.Cin_g + c0;
        const float* base = (gch0 < a.C0) ? in0n + (long)gch0 * HW : in1n + (long)(gch0 - a.C0) * HW;
        const int cvalid = a.Cin_g - c0;                 // channels of this chunk that exist
        if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < IG_NE4; ++j) {           // branch-free: padding quads read the chunk's first quad and are zeroed
                const bool ok = eoff[j] >= 0 && ech[j] < cvalid;
                f32x4 v = *(const f32x4*)(base + (ok ? eoff[j] : 0));
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = ok ? v[u] : 0.f;
                pre4[j] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < IG_NE; ++j) {
                float v = 0.f;
                if (eoff[j] >= 0 && ech[j] < cvalid) v = base[eoff[j]];
                pre[j] = v;
            }
        }
        const int o0 = c0 >> 3;
        int noct = noct_g - o0; if (noct > (CK >> 3)) noct = CK >> 3;
        const int nq = noct * KSO * 128;                 // 16-byte units per cout tile
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const ig_u32x4* src = wtile + i * tile_stride + (long)o0 * KSO * 128;
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int q = tid + NT * j;
                if (q < nq) wreg[i][j] = src[q];
            }
        }
        return noct;
    };
    auto commit = [&](int buf, int noct) {   // registers -> LDS buffer `buf`
        float* patch = patch0 + buf * patch_elems;
        if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < IG_NE4; ++j) {
                const int e = tid + NT * j;
                if (4 * e < creal * PHW) *(f32x4*)(patch + 4 * e) = pre4[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < IG_NE; ++j) {
                const int e = tid + NT * j;
                if (e < creal * PHW) patch[e] = pre[j];
            }
        }
        const int nq = noct * KSO * 128;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            ig_u32x4* dst = wf0 + buf * wq + i * KSC * 128;
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int q = tid + NT * j;
                if (q < nq) dst[q] = wreg[i][j];
            }
        }
    };

    float lo_scale = kIgLoScale;                         // scalar register: the mix instructions take no literal
    asm volatile("" : "+s"(lo_scale));
    const ig_f16x2 ws_c = {(_Float16)(1.f / kIgLoScale), (_Float16)(1.f / kIgLoScale)};

    int noct_cur = issue(0);
    commit(0, noct_cur);
    if (tid < WN) bias_s[tid] = bias_v;
    __syncthreads();
    int cur = 0;
    for (int c0 = 0; c0 < a.Cin_g; c0 += CK) {
        const bool more = c0 + CK < a.Cin_g;
        int noct_next = 0;
        if (more) noct_next = issue(c0 + CK);            // loads fly while this chunk is multiplied

        const float* patch = patch0 + cur * patch_elems;
        const ig_u32x4* wl = wf0 + cur * wq + lane;
        int cvalid = a.Cin_g - c0; if (cvalid > CK) cvalid = CK;
        const int nks = noct_cur * KSO;
        // one k-step = (octet ol, step ks of the octet): this half-wave's group j = 2 ks + half (tap j; past T: the last tap against zero weights)
        auto fetch = [&](int ol, int ks, float (&v)[8]) __attribute__((always_inline)) {
            const int j = 2 * ks + half;
            const int ko = koff[j < T ? j : T - 1] + pix;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int ce = ol * 8 + e;
                ce = ce < cvalid ? ce : cvalid - 1;      // a channel past Cin: the last real plane (finite values, zero weights)
                v[e] = patch[ce * PHW + ko];
            }
        };
        float vcur[8], vnxt[8];
        int ol = 0, ks = 0;
        if (nks > 0) fetch(0, 0, vcur);
        for (int kk = 0; kk < nks; ++kk) {
            int oln = ol, ksn = ks + 1;
            if (ksn == KSO) { ksn = 0; ++oln; }
            if (kk + 1 < nks) fetch(oln, ksn, vnxt);
            ig_u32x4 hi, lo;
            ig_split8_act(vcur, hi, lo, lo_scale);
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const ig_u32x4 whi = wl[((i * KSC + kk) * 2 + 0) * 64], wlo = wl[((i * KSC + kk) * 2 + 1) * 64];
                ig_u32x4 whs;                            // 2^-11 x the high weight part (exact while normal): the partner of the scaled low activation part
#pragma unroll
                for (int q = 0; q < 4; ++q) whs[q] = ig_pk_mul(whi[q], ws_c);
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ig_f16x8, wlo), __builtin_bit_cast(ig_f16x8, hi), acc[i][0], 0, 0, 0);
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ig_f16x8, whi), __builtin_bit_cast(ig_f16x8, hi), acc[i][0], 0, 0, 0);
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ig_f16x8, whs), __builtin_bit_cast(ig_f16x8, lo), acc[i][0], 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) vcur[e] = vnxt[e];
            ol = oln; ks = ksn;
        }
        if (more) commit(cur ^ 1, noct_next);
        __syncthreads();
        cur ^= 1;
        noct_cur = noct_next;
    }

    // range status word: an operand beyond fp16's range is packed as inf and makes EVERY cout of its pixel non-finite (inf x 0 = NaN):
    // register 0 of tile 0 sees the wave's 32 pixels
    if (a.status && __builtin_amdgcn_class(acc[0][0][0], 0x207)) atomicOr(a.status, 1u);
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] *= 1.f / kIgScale;          // exact
    conv_epilogue<NC, 1>(a, acc, bias_s, n, g, cg, ty * 8 + wave, tx * 32 + l31, half, a_res, a_res_bs, a_out, a_out_bs);
}

// weight [Cout, Cin_g, KH, KW] fp32 -> A fragments [group][cout tile of 32 (ntile_pad)][octet][k-step][part][lane][8] fp16 of 2^8 x w
__global__ void conv_ig16_pack_kernel(const float* w, unsigned short* wp, int Cout_g, int Cin_g, int T, int KSO, int noct_g, int ntile_pad, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), part = (int)((i >> 9) & 1);
    long tt = i >> 10;
    const int ks = (int)(tt % KSO); tt /= KSO;
    const int o = (int)(tt % noct_g); tt /= noct_g;
    const int tile = (int)(tt % ntile_pad);
    const int g = (int)(tt / ntile_pad);
    const int m = tile * 32 + (lane & 31), j = 2 * ks + (lane >> 5), c = o * 8 + e;
    double v = 0.0;
    if (m < Cout_g && c < Cin_g && j < T) v = (double)w[((long)(g * Cout_g + m) * Cin_g + c) * T + j] * (double)kIgScale;
    unsigned short out = 0;
    for (int p = 0; p <= part; ++p) {
        const _Float16 h = (_Float16)(float)v;
        out = __builtin_bit_cast(unsigned short, h);
        v -= (double)(float)h;
    }
    wp[i] = out;
}

// ---- host side -------------------------------------------------------------------------------------------------------------
namespace {
struct Ig16Geom { int Cin_g, Cout_g, T, KSO, noct_g, ntile_pad; };
bool ig16_geom(const MotifConvDesc* d, Ig16Geom* q) {
    if (!d || d->mma != 7 || d->groups < 1 || d->KH < 1 || d->KW < 1 || d->stride < 1 || d->dil < 1) return false;
    const int Cin = d->C0 + d->C1;
    if (Cin <= 0 || d->Cout <= 0 || Cin % d->groups || d->Cout % d->groups) return false;
    q->Cin_g = Cin / d->groups; q->Cout_g = d->Cout / d->groups;
    q->T = d->KH * d->KW;
    q->KSO = (q->T + 1) / 2;
    if (q->KSO > ig_ksc_max(1)) return false;            // one octet's fragments of one cout tile must fit a chunk (<= 7x7)
    q->noct_g = (q->Cin_g + 7) / 8;
    q->ntile_pad = 2 * ((q->Cout_g + 63) / 64);
    if (d->C1 > 0 && (d->groups != 1 || (d->C0 & 7))) return false;     // an octet lies in one source
    return true;
}
struct Ig16Plan { int NC, CK, PH, PW, xoff; size_t lds; };
bool ig16_plan(const MotifConvDesc* d, const Ig16Geom& q, bool vec, Ig16Plan* p) {
    p->PH = 7 * d->stride + (d->KH - 1) * d->dil + 1;
    p->PW = 31 * d->stride + (d->KW - 1) * d->dil + 1;
    p->xoff = 0;
    if (vec) {
        p->xoff = (4 - d->pad % 4) % 4;
        p->PW = 4 * ((p->xoff + p->PW + 3) / 4);
    }
    const long PHW = (long)p->PH * p->PW;
    for (int nc = (q.Cout_g > 32 ? 2 : 1); nc >= 1; --nc) {
        for (int m = q.noct_g < 4 ? q.noct_g : 4; m >= 1; --m) {                // octets per chunk
            const int ck = 8 * m, creal = ck < q.Cin_g ? ck : q.Cin_g;
            if (creal * PHW > IG_PATCH_MAX || m * q.KSO > ig_ksc_max(nc)) continue;
            if (d->C1 > 0 && d->C0 % ck) continue;       // a chunk must not straddle the two sources
            const size_t patch_elems = ((size_t)creal * PHW + 3) & ~(size_t)3;
            const size_t lds = 2 * patch_elems * 4 + 2 * (size_t)nc * m * q.KSO * 128 * 16 + (((size_t)q.T + 3) & ~(size_t)3) * 4 + (size_t)32 * nc * 4;
            if (lds > 156 * 1024) continue;
            p->NC = nc; p->CK = ck; p->lds = lds;
            return true;
        }
    }
    return false;
}
}  // namespace

// Desc-only: does the packed blob of this layer carry the fp16 fragment block (behind the fp32 block of conv_igemm.hip)?
// (no run-time option may enter here: a blob packed under one option value must stay valid under every other)
bool motif_conv_ig16_pack_eligible(const MotifConvDesc* d) {
    Ig16Geom q;
    return ig16_geom(d, &q);
}

long motif_conv_ig16_packed_floats(const MotifConvDesc* d) {
    Ig16Geom q;
    if (!ig16_geom(d, &q)) return 0;
    return (long)d->groups * q.ntile_pad * q.noct_g * q.KSO * 2 * 64 * 4;       // 8 halves = 4 floats per lane
}

int motif_conv_ig16_pack(const MotifConvDesc* d, const float* weight, float* packed16, hipStream_t s) {
    Ig16Geom q;
    if (!ig16_geom(d, &q)) return MOTIF_EINVAL;
    const long total = (long)d->groups * q.ntile_pad * q.noct_g * q.KSO * 2 * 64 * 8;
    conv_ig16_pack_kernel<<<cdiv(total, 256), 256, 0, s>>>(weight, (unsigned short*)packed16, q.Cout_g, q.Cin_g, q.T, q.KSO, q.noct_g, q.ntile_pad, total);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// Per launch: `fp32_block_floats` = size of the layer's fp32 block (the fragment block follows it).  Returns MOTIF_ELIMIT when the shape does
// not fit (the caller then runs the fp32 engine on the fp32 block).
int motif_conv_ig16_launch(const MotifConvDesc* d, ConvArgs& a, int P, long fp32_block_floats, hipStream_t s) {
    Ig16Geom q;
    const int force = motif_opt(MOTIF_OPT_CONV_ENGINE);            // 6: never, 7: wherever it fits (tests), otherwise where it pays
    if (!ig16_geom(d, &q) || force == 6) return MOTIF_ELIMIT;
    // Where it pays (tools/ig16_bench.py, one launch of every such layer of the clip and of PWC-Net against the fp32 engine, profiles/r06_ig16_shapes.txt):
    // a block stages ONE chunk per 8 input channels, so a layer with few channels has nothing to overlap its loads with (3 -> 32 7x7:
    // 112 vs 74 us) and short reductions are launch-bound either way; from 24 channels per group on it wins wherever the fp32 engine's
    // own staging is at its worst -- stride 2 (64 -> 64: 35 vs 61 us), dilation (128 -> 128 d2: 120 vs 193) -- or the layer is wide.
    const bool pays = q.Cin_g >= 24 && ((long)q.Cin_g * q.T >= 200 || q.Cin_g >= 64) && (d->stride > 1 || d->dil > 1 || q.Cin_g >= 64 || q.Cout_g >= 48);
    if (force != 7 && !pays) return MOTIF_ELIMIT;
    if ((long)d->H * d->W >= 0xFFFFFF || (long)q.Cin_g * d->H * d->W >= 0x7fffffffL) return MOTIF_ELIMIT;
    const int Ho = (d->H + 2 * d->pad - (d->dil * (d->KH - 1) + 1)) / d->stride + 1;
    const int Wo = (d->W + 2 * d->pad - (d->dil * (d->KW - 1) + 1)) / d->stride + 1;
    if (Ho <= 0 || Wo <= 0) return MOTIF_ELIMIT;
    bool vec = (d->W & 3) == 0 && d->pad_mode == 0 && !motif_opt(MOTIF_OPT_CONV_NOVEC);
    for (int i = 0; i < P && vec; ++i)
        vec = ((((unsigned long long)a.in0[i] | (unsigned long long)a.in1[i]) & 15) == 0) && (((a.in0_bs[i] | (a.in1[i] ? a.in1_bs[i] : 0)) & 3) == 0);
    Ig16Plan p;
    if (!(vec && ig16_plan(d, q, true, &p))) { vec = false; if (!ig16_plan(d, q, false, &p)) return MOTIF_ELIMIT; }
    a.C0 = d->C0; a.H = d->H; a.W = d->W; a.Ho = Ho; a.Wo = Wo;
    a.Cin_g = q.Cin_g; a.Cout_g = q.Cout_g; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil; a.pad_mode = d->pad_mode;
    a.act = d->act; a.act2 = d->act2; a.act_split = d->act_split; a.res_mode = d->res_mode;
    a.CK = p.CK; a.PH = p.PH; a.PW = p.PW; a.Kpad = 0;
    a.xoff = p.xoff;
    a.tiles_x = (Wo + 31) / 32;
    const int tiles_y = (Ho + 7) / 8;
    a.ncg = (q.Cout_g + 32 * p.NC - 1) / (32 * p.NC);
    dim3 grid(a.tiles_x * tiles_y, d->groups * a.ncg, d->N * P);
#define MOTIF_LAUNCH_IG16(NCV, VECV)                                                                                              \
    do {                                                                                                                           \
        hipError_t e = hipFuncSetAttribute((const void*)conv_ig16_kernel<NCV, VECV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds); \
        if (e != hipSuccess) return (int)e;                                                                                        \
        conv_ig16_kernel<NCV, VECV><<<grid, IG_NT, p.lds, s>>>(a, q.KSO, q.noct_g, q.ntile_pad, fp32_block_floats);              \
    } while (0)
    if (p.NC == 2) { if (vec) MOTIF_LAUNCH_IG16(2, true); else MOTIF_LAUNCH_IG16(2, false); }
    else { if (vec) MOTIF_LAUNCH_IG16(1, true); else MOTIF_LAUNCH_IG16(1, false); }
#undef MOTIF_LAUNCH_IG16
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

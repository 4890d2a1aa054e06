// Dense convolution as an fp32-MFMA implicit GEMM for gfx950.
//
//   D[cout][pixel] = sum_k  Wp[k][cout] * im2col[k][pixel],   k = (c, ky, kx)
//
// v_mfma_f32_32x32x2_f32 computes a 32(cout) x 32(pixel) tile per wave-instruction with K=2: the lower
// half-wave supplies k=2s, the upper half k=2s+1.  NCHW planar fp32 is the natural layout for it: a
// wave's 32 pixels are 32 consecutive x of one output row, so the B operand is one conflict-free
// ds_read_b32 from an LDS input patch, and the A operand one ds_read_b32 from a [k][cout] weight slab.
//
// Block = 4 waves = 8 output rows x 32 columns; each wave owns 2 rows x NC cout-tiles of 32
// (NC*2 accumulators of 16 VGPRs).  The reduction runs over channel chunks: stage the input patch of
// CK channels (+halo, zero/reflect padded) and the matching weight rows in LDS, barrier, MFMA steps.
// The (c,ky,kx)->patch-offset map is a small LDS table so ONE kernel serves every kernel size, stride,
// dilation and group count on the path (3x3 s1/s2, 1x1, 7x7 s2, dilated PWC refiner, grouped).
// Fused: channel concat of two inputs, bias, residual, activation (per channel range).
#include "conv_common.h"
#include <stdlib.h>

// SPEC 1: 3x3, stride 1, dilation 1 (patch pitch 34, or 40 with VEC): tap offsets are immediates, the MFMA loop has no VALU.
// VEC (round 3): the patch is staged in 16-byte row pieces -- 4 pixels of one channel per lane instead of one -- into LDS rows
// that start a.xoff pixels left of the patch (so that every piece is an aligned quad of the image row): a quarter of the
// vector-memory instructions.  The narrow / 1x1 layers this engine serves are bandwidth-bound, and their 4-byte-per-lane staging
// loads cost ~45 cycles each on the CU's texture addresser (what round 3 found to bound conv_split_kernel too).  Needs
// W % 4 == 0, 16-byte aligned inputs and zero padding (host check); everything else takes the scalar plan.
template <int NC, int RPW, int SPEC, bool VEC>
__global__ __launch_bounds__(64 * (8 / RPW)) void conv_igemm_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = 64 * (8 / RPW);       // threads: one wave per RPW output rows of the 8-row tile
    constexpr int WN = 32 * NC;               // weight slab row width
    constexpr int NE_MAX = PATCH_MAX / NT;    // patch elements a thread prefetches per chunk
    constexpr int NE4 = NE_MAX / 4;           // ... as 16-byte pieces (VEC)
    constexpr int PITCH = VEC ? 40 : 34;      // SPEC 1 row pitch
    constexpr int NW_MAX = WCHUNK_MAX / 4 / NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int tile_id = (a.dbg & 64) ? (int)blockIdx.x : xcd_tile_id();
    const int tx = tile_id % a.tiles_x, ty = tile_id / a.tiles_x;
    const int g = blockIdx.y / a.ncg, cg = blockIdx.y % a.ncg;
    const int pz = blockIdx.z / a.N, n = blockIdx.z - pz * a.N;
    const float* a_in0 = a.in0[pz]; const float* a_in1 = a.in1[pz]; const float* a_wp = a.wp[pz];
    const float* a_bias = a.bias[pz]; const float* a_res = a.res[pz]; float* a_out = a.out[pz];
    const long a_in0_bs = a.in0_bs[pz], a_in1_bs = a.in1_bs[pz], a_res_bs = a.res_bs[pz], a_out_bs = a.out_bs[pz];
    const int PHW = a.PH * a.PW;
    const int T = a.KH * a.KW;
    const int KC = a.CK * T;                  // rows per full chunk (host guarantees even)
    const int CKPHW = a.CK * PHW;
    const int patch_elems = (CKPHW + 3) & ~3;
    float* patch0 = smem;
    float* wl0 = patch0 + 2 * patch_elems;
    int* koff = (int*)(wl0 + 2 * KC * WN);
    float* bias_s = (float*)(koff + ((KC + 3) & ~3));      // [WN] bias of this cout group (zeros if none), 16-byte aligned

    const long HW = (long)a.H * a.W;
    // chunk-invariant staging plan, in registers: element e = tid + NT*j of the [CK][PH][PW] patch comes from
    // input offset eoff[j] = c*HW + iy*W + ix relative to the chunk's first channel plane (-1: padding),
    // ech[j] = chunk-local channel (to cut off the tail chunk).  Integer divisions happen once per block.
    const int iy0 = ty * 8 * a.stride - a.pad, ix0 = tx * 32 * a.stride - a.pad - (VEC ? a.xoff : 0);
    int eoff[VEC ? NE4 : NE_MAX];
    int ech[VEC ? NE4 : NE_MAX];
    if constexpr (VEC) {
        const int PWQ = a.PW >> 2, PHQ = a.PH * PWQ, nq = a.CK * PHQ;          // quads per row / channel / chunk
#pragma unroll
        for (int j = 0; j < NE4; ++j) {
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
#pragma unroll
        for (int j = 0; j < NE_MAX; ++j) {
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
    // K order inside a chunk: (channel pair, tap, half) -- an MFMA step takes tap t of channel 2cp from the lower
    // half-wave and of channel 2cp+1 from the upper one, so the patch offset of a step is half*PHW + tapoff[t].
    for (int t = tid; t < T; t += NT) {
        const int ky = t / a.KW, kx = t - ky * a.KW;
        koff[t] = ky * a.dil * a.PW + kx * a.dil;
    }
    float bias_v = 0.f;                       // requested now, parked in LDS before the first barrier
    if (tid < WN && a_bias && cg * WN + tid < a.Cout_g) bias_v = a_bias[g * a.Cout_g + cg * WN + tid];

    f32x16 acc[NC][RPW];
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int j = 0; j < RPW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* in0n = a_in0 + (long)n * a_in0_bs;
    const float* in1n = a_in1 ? a_in1 + (long)n * a_in1_bs : nullptr;
    const float* wbase = a_wp + ((long)(g * a.ncg + cg) * a.Kpad) * WN;
    int pix[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) pix[j] = (RPW * wave + j) * a.stride * a.PW + l31 * a.stride + (VEC ? a.xoff : 0);

    float pre[VEC ? 1 : NE_MAX];
    f32x4 pre4[VEC ? NE4 : 1];
    f32x4 wreg[NW_MAX];
    // the two-source concat never straddles a chunk when C0 % CK == 0 (host falls back to CK | C0 otherwise)
    auto issue = [&](int c0) {            // global -> registers for the chunk starting at channel c0
        const int gch0 = g * a.Cin_g + c0;
        const float* base = (gch0 < a.C0) ? in0n + (long)gch0 * HW : in1n + (long)(gch0 - a.C0) * HW;
        const int cvalid = a.Cin_g - c0;                 // channels of this chunk that exist
        if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < NE4; ++j) {              // branch-free: padding quads read the chunk's first quad and are zeroed
                const bool ok = eoff[j] >= 0 && ech[j] < cvalid;
                f32x4 v = *(const f32x4*)(base + (ok ? eoff[j] : 0));
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = ok ? v[u] : 0.f;
                pre4[j] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NE_MAX; ++j) {
                float v = 0.f;
                if (eoff[j] >= 0 && ech[j] < cvalid) v = base[eoff[j]];
                pre[j] = v;
            }
        }
        const int r0 = c0 * T;
        int rows = a.Kpad - r0; if (rows > KC) rows = KC;
        const int n4 = rows * WN / 4;
        const f32x4* src = (const f32x4*)(wbase + (long)r0 * WN);
#pragma unroll
        for (int j = 0; j < NW_MAX; ++j) {
            const int i = tid + NT * j;
            if (i < n4) wreg[j] = src[i];
        }
        return rows;
    };
    auto commit = [&](int buf, int rows) {   // registers -> LDS buffer `buf`
        float* patch = patch0 + buf * patch_elems;
        f32x4* w4 = (f32x4*)(wl0 + buf * KC * WN);
        if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < NE4; ++j) {
                const int e = tid + NT * j;
                if (4 * e < CKPHW) *(f32x4*)(patch + 4 * e) = pre4[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < NE_MAX; ++j) {
                const int e = tid + NT * j;
                if (e < CKPHW) patch[e] = pre[j];
            }
        }
        const int n4 = rows * WN / 4;
#pragma unroll
        for (int j = 0; j < NW_MAX; ++j) {
            const int i = tid + NT * j;
            if (i < n4) w4[i] = wreg[j];
        }
    };

    int rows_cur = issue(0);
    commit(0, rows_cur);
    if (tid < WN) bias_s[tid] = bias_v;
    __syncthreads();
    int cur = 0;
    for (int c0 = 0; c0 < a.Cin_g; c0 += a.CK) {
        const bool more = c0 + a.CK < a.Cin_g;
        int rows_next = 0;
        if (more && !(a.dbg & 1)) rows_next = issue(c0 + a.CK);          // loads fly while this chunk is multiplied
        else if (more) { rows_next = a.Kpad - (c0 + a.CK) * T; if (rows_next > KC) rows_next = KC; }

        const float* patch = patch0 + cur * patch_elems + half * PHW;
        const float* wl = wl0 + cur * KC * WN + half * WN + l31;
        const int ncp = (a.dbg & 2) ? 0 : rows_cur / (2 * T);          // channel pairs in this chunk
        if constexpr (SPEC == 1) {
            for (int cp = 0; cp < ncp; ++cp) {
                const float* bc = patch + cp * 2 * PHW;
                const float* wc = wl + cp * 18 * WN;
                float bv[9][RPW], av[9][NC];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
#pragma unroll
                    for (int j = 0; j < RPW; ++j) bv[t][j] = bc[(t / 3) * PITCH + (t % 3) + pix[j]];
#pragma unroll
                    for (int i = 0; i < NC; ++i) av[t][i] = wc[t * 2 * WN + i * 32];
                }
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int i = 0; i < NC; ++i)
#pragma unroll
                        for (int j = 0; j < RPW; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][i], bv[t][j], acc[i][j], 0, 0, 0);
            }
        } else {
            for (int cp = 0; cp < ncp; ++cp) {
                const float* bc = patch + cp * 2 * PHW;
                const float* wc = wl + cp * 2 * T * WN;
                int t = 0;
                for (; t + 4 <= T; t += 4) {
                    int ko[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) ko[u] = koff[t + u];
                    float bv[4][RPW], av[4][NC];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
#pragma unroll
                        for (int j = 0; j < RPW; ++j) bv[u][j] = bc[ko[u] + pix[j]];
#pragma unroll
                        for (int i = 0; i < NC; ++i) av[u][i] = wc[(t + u) * 2 * WN + i * 32];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int i = 0; i < NC; ++i)
#pragma unroll
                            for (int j = 0; j < RPW; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u][j], acc[i][j], 0, 0, 0);
                }
                for (; t < T; ++t) {
                    const int ko1 = koff[t];
#pragma unroll
                    for (int i = 0; i < NC; ++i) {
                        const float av1 = wc[t * 2 * WN + i * 32];
#pragma unroll
                        for (int j = 0; j < RPW; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bc[ko1 + pix[j]], acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        if (more && !(a.dbg & 1)) commit(cur ^ 1, rows_next);
        __syncthreads();
        cur ^= 1;
        rows_cur = rows_next;
    }

    conv_epilogue<NC, RPW>(a, acc, bias_s, n, g, cg, ty * 8 + RPW * wave, tx * 32 + l31, half, a_res, a_res_bs, a_out, a_out_bs);
}

// weight [Cout, Cin_g, KH, KW] -> packed [groups][ncg][Kpad][32*NC], zero padded
// weight [Cout, Cin_g, KH, KW] -> packed [groups][ncg][Kpad][32*NC] with row r = ((cp*T + t)*2 + half) holding
// k = (channel 2cp+half, tap t); zero padded (odd channel counts get a zero partner channel)
__global__ void conv_pack_kernel(const float* w, float* wp, int Cout_g, int Cin_g, int T, int Kpad, int ncg, int WN, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int j = (int)(i % WN);
    long tt = i / WN;
    int r = (int)(tt % Kpad);
    tt /= Kpad;
    int cgi = (int)(tt % ncg);
    int g = (int)(tt / ncg);
    int col = cgi * WN + j;
    const int half = r & 1, q = r >> 1, t = q % T, cp = q / T;
    const int c = 2 * cp + half;
    float v = 0.f;
    if (col < Cout_g && c < Cin_g) v = w[((long)(g * Cout_g + col)) * Cin_g * T + (long)c * T + t];
    wp[i] = v;
}

namespace {
struct ConvPlan { int Cin, Cin_g, Cout_g, K, Kpad, NC, WN, ncg, Ho, Wo, PH, PW, CK, T; size_t lds; };

bool plan_conv(const MotifConvDesc* d, ConvPlan* p, bool vec = false) {
    if (!d || d->groups < 1 || d->KH < 1 || d->KW < 1 || d->stride < 1 || d->dil < 1) return false;
    p->Cin = d->C0 + d->C1;
    if (p->Cin % d->groups || d->Cout % d->groups || p->Cin <= 0 || d->Cout <= 0) return false;
    p->Cin_g = p->Cin / d->groups;
    p->Cout_g = d->Cout / d->groups;
    p->T = d->KH * d->KW;
    p->K = p->Cin_g * p->T;
    p->Kpad = 2 * p->T * ((p->Cin_g + 1) / 2);     // channel pairs x taps x 2
    p->NC = p->Cout_g > 32 ? 2 : 1;
    p->WN = 32 * p->NC;
    p->ncg = (p->Cout_g + p->WN - 1) / p->WN;
    p->Ho = (d->H + 2 * d->pad - (d->dil * (d->KH - 1) + 1)) / d->stride + 1;
    p->Wo = (d->W + 2 * d->pad - (d->dil * (d->KW - 1) + 1)) / d->stride + 1;
    if (p->Ho <= 0 || p->Wo <= 0) return false;
    if ((long)d->H * d->W >= 0xFFFFFF) return false;     // plane offsets are packed into 24 bits
    p->PH = 7 * d->stride + (d->KH - 1) * d->dil + 1;
    p->PW = 31 * d->stride + (d->KW - 1) * d->dil + 1;
    if (vec) {                                           // rows of whole aligned quads: xoff pixels of slack on the left (conv_igemm_kernel VEC)
        const int xoff = (4 - d->pad % 4) % 4;
        p->PW = 4 * ((xoff + p->PW + 3) / 4);
        if (d->KH == 3 && d->KW == 3 && d->stride == 1 && d->dil == 1) p->PW = 40;    // the pitch SPEC 1 is compiled for
    }
    // chunk: as many channels as fit the register-prefetch limits and ~29 KB per LDS buffer, CK*T even
    const long PHW = (long)p->PH * p->PW;
    auto ok = [&](int c) {
        return (long)c * PHW <= PATCH_MAX && (long)c * p->T * p->WN <= WCHUNK_MAX && c < 128 &&
               ((long)c * PHW * 4 + (long)c * p->T * p->WN * 4) <= 30 * 1024;
    };
    // with two concatenated sources a chunk must not straddle them: CK | C0 (single group only)
    const bool two = d->C1 > 0;
    if (two && d->groups != 1) return false;
    int ck = 0;
    for (int c = 1; c <= p->Cin_g + 1; ++c) {
        if (!ok(c)) break;
        if ((c & 1) == 0 && (!two || d->C0 % c == 0)) ck = c;
    }
    if (ck == 0) {                       // relax the LDS soft cap, keep the hard register limits
        for (int c = 1; c <= 2; ++c)
            if ((c & 1) == 0 && (!two || d->C0 % c == 0) && (long)c * PHW <= PATCH_MAX && (long)c * p->T * p->WN <= WCHUNK_MAX) { ck = c; break; }
    }
    if (ck == 0) return false;
    if (const int f = motif_opt(MOTIF_OPT_CONV_CK)) {         // tuning aid
        if (f > 0 && f <= ck && (f & 1) == 0 && (!two || d->C0 % f == 0)) ck = f;
    }
    p->CK = ck;
    const size_t patch_elems = ((size_t)ck * PHW + 3) & ~(size_t)3;
    p->lds = (2 * patch_elems + 2 * (size_t)ck * p->T * p->WN + (((size_t)ck * p->T + 3) & ~(size_t)3) + (size_t)p->WN) * 4;
    return p->lds <= 160 * 1024;
}
}  // namespace

extern "C" long motif_conv2d_packed_size(const MotifConvDesc* d) {
    if (motif_conv_split_eligible(d)) return motif_conv_split_packed_floats(d);
    ConvPlan p;
    if (!plan_conv(d, &p)) return MOTIF_EINVAL;
    // mma = 7: the fp16 fragment block of conv_ig16.hip follows the fp32 block (which conv_direct / conv_pw / the fp32 fall-back keep reading)
    return (long)d->groups * p.ncg * p.Kpad * p.WN + (motif_conv_ig16_pack_eligible(d) ? motif_conv_ig16_packed_floats(d) : 0L);
}

extern "C" int motif_conv2d_pack(const MotifConvDesc* d, const float* weight, float* packed, void* stream) {
    if (motif_conv_split_eligible(d)) {
        if (!weight || !packed) return MOTIF_EINVAL;
        return motif_conv_split_pack(d, weight, packed, (hipStream_t)stream);
    }
    ConvPlan p;
    if (!plan_conv(d, &p) || !weight || !packed) return MOTIF_EINVAL;
    long total = (long)d->groups * p.ncg * p.Kpad * p.WN;
    conv_pack_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(weight, packed, p.Cout_g, p.Cin_g, p.T, p.Kpad, p.ncg, p.WN, total);
    MOTIF_LAUNCH_CHECK();
    if (motif_conv_ig16_pack_eligible(d)) return motif_conv_ig16_pack(d, weight, packed + total, (hipStream_t)stream);
    return MOTIF_OK;
}

extern "C" int motif_conv2d_fwd_multi(const MotifConvDesc* d, int P, const float* const* in0, const float* const* in1,
                                      const float* const* packed, const float* const* bias, const float* const* res,
                                      float* const* out, const long* in0_bs, const long* in1_bs, const long* res_bs,
                                      const long* out_bs, void* stream) {
    ConvPlan p;
    const bool split = motif_conv_split_eligible(d);
    if (!split && !plan_conv(d, &p)) return MOTIF_ELIMIT;
    if (split) { p.Ho = d->H + 2 * d->pad - 2; p.Wo = d->W + 2 * d->pad - 2; }
    if (P < 1 || P > MOTIF_MAX_PROBLEMS || !in0 || !packed || !out || d->N < 1) return MOTIF_EINVAL;
    ConvArgs a;
    const long HW = (long)d->H * d->W, HWo = (long)p.Ho * p.Wo;
    for (int i = 0; i < MOTIF_MAX_PROBLEMS; ++i) {
        const int j = i < P ? i : 0;
        if (!in0[j] || !packed[j] || !out[j] || (d->C1 > 0 && (!in1 || !in1[j])) || (d->res_mode && (!res || !res[j]))) return MOTIF_EINVAL;
        a.in0[i] = in0[j]; a.in1[i] = (in1 && d->C1 > 0) ? in1[j] : nullptr; a.wp[i] = packed[j];
        a.bias[i] = bias ? bias[j] : nullptr; a.res[i] = (res && d->res_mode) ? res[j] : nullptr; a.out[i] = out[j];
        a.in0_bs[i] = (in0_bs && in0_bs[j]) ? in0_bs[j] : (long)d->C0 * HW;
        a.in1_bs[i] = (in1_bs && in1_bs[j]) ? in1_bs[j] : (long)d->C1 * HW;
        a.res_bs[i] = (res_bs && res_bs[j]) ? res_bs[j] : (long)d->Cout * HWo;
        a.out_bs[i] = (out_bs && out_bs[j]) ? out_bs[j] : (long)d->Cout * HWo;
    }
    a.N = d->N;
    a.dbg = motif_opt(MOTIF_OPT_CONV_DBG);
    a.status = d->status;
    if (split) {
        a.C0 = d->C0; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
        a.KH = 3; a.KW = 3; a.stride = 1; a.pad = d->pad; a.dil = 1; a.pad_mode = d->pad_mode;
        a.act = d->act; a.act2 = d->act2; a.act_split = d->act_split; a.res_mode = d->res_mode;
        return motif_conv_split_launch(d, a, P, (hipStream_t)stream);
    }
    a.C0 = d->C0; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
    a.act = d->act; a.act2 = d->act2; a.act_split = d->act_split; a.res_mode = d->res_mode;
    if (motif_conv_direct_eligible(d, a, P)) return motif_conv_direct_launch(d, a, P, (hipStream_t)stream);      // narrow layer, large map
    if (motif_conv_pw_eligible(d, a, P)) return motif_conv_pw_launch(d, a, P, (hipStream_t)stream);              // 1x1 layer, mma = 7: conv_pw.hip
    if (motif_conv_ig16_pack_eligible(d)) {                                                                      // everything else, mma = 7: conv_ig16.hip
        const int rc = motif_conv_ig16_launch(d, a, P, (long)d->groups * p.ncg * p.Kpad * p.WN, (hipStream_t)stream);
        if (rc != MOTIF_ELIMIT) return rc;               // (a shape it does not fit runs on the fp32 engine below)
    }
    // 16-byte staging (conv_igemm_kernel VEC) where the layout allows it: rows of whole quads, aligned inputs, zero padding
    bool vec = (d->W & 3) == 0 && d->pad_mode == 0 && !motif_opt(MOTIF_OPT_CONV_NOVEC);
    for (int i = 0; i < P && vec; ++i)
        vec = ((((unsigned long long)a.in0[i] | (unsigned long long)a.in1[i]) & 15) == 0) && (((a.in0_bs[i] | (a.in1[i] ? a.in1_bs[i] : 0)) & 3) == 0);
    if (vec && !plan_conv(d, &p, true)) { vec = false; if (!plan_conv(d, &p)) return MOTIF_ELIMIT; }
    a.C0 = d->C0; a.H = d->H; a.W = d->W; a.Ho = p.Ho; a.Wo = p.Wo;
    a.Cin_g = p.Cin_g; a.Cout_g = p.Cout_g; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil; a.pad_mode = d->pad_mode;
    a.act = d->act; a.act2 = d->act2; a.act_split = d->act_split; a.res_mode = d->res_mode;
    a.CK = p.CK; a.PH = p.PH; a.PW = p.PW; a.Kpad = p.Kpad;
    a.xoff = vec ? (4 - d->pad % 4) % 4 : 0;
    a.tiles_x = (p.Wo + 31) / 32;
    const int tiles_y = (p.Ho + 7) / 8;
    a.ncg = p.ncg;
    dim3 grid(a.tiles_x * tiles_y, d->groups * p.ncg, d->N * P);
    hipStream_t s = (hipStream_t)stream;
    const bool spec = d->KH == 3 && d->KW == 3 && d->stride == 1 && d->dil == 1 && !motif_opt(MOTIF_OPT_CONV_NOSPEC);
#define MOTIF_LAUNCH_CONV(NCV, SPECV, VECV)                                                                                  \
    do {                                                                                                                     \
        if (p.lds > 64 * 1024)                                                                                               \
            (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<NCV, 1, SPECV, VECV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds); \
        conv_igemm_kernel<NCV, 1, SPECV, VECV><<<grid, 512, p.lds, s>>>(a);                                                \
    } while (0)
#define MOTIF_LAUNCH_CONV2(NCV, SPECV) do { if (vec) MOTIF_LAUNCH_CONV(NCV, SPECV, true); else MOTIF_LAUNCH_CONV(NCV, SPECV, false); } while (0)
    if (p.NC == 2) { if (spec) MOTIF_LAUNCH_CONV2(2, 1); else MOTIF_LAUNCH_CONV2(2, 0); }
    else { if (spec) MOTIF_LAUNCH_CONV2(1, 1); else MOTIF_LAUNCH_CONV2(1, 0); }
#undef MOTIF_LAUNCH_CONV2
#undef MOTIF_LAUNCH_CONV
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

extern "C" int motif_conv2d_fwd(const MotifConvDesc* d, const float* in0, const float* in1, const float* packed,
                                const float* bias, const float* res, float* out, void* stream) {
    if (!d) return MOTIF_EINVAL;
    const long bs0 = d->in0_bs, bs1 = d->in1_bs, bsr = d->res_bs, bso = d->out_bs;
    return motif_conv2d_fwd_multi(d, 1, &in0, &in1, &packed, &bias, &res, &out, &bs0, &bs1, &bsr, &bso, stream);
}

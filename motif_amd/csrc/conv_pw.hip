// 1x1 (pointwise) convolutions on the fp16 matrix cores with the two-part split of conv_wino.hip (MotifConvDesc.mma = 7): a
// bandwidth-shaped kernel for the layers the fp32-MFMA engine ran at a fifth of its peak while waiting on memory (round 4; the
// 1x1 layers were 1.06 of conv_other's 2.9 ms per clip: RAFT / PWC feature heads, the ConvLSTM and fusion layers, 8 .. 196 -> 32 .. 96).
//
// out[cout, p] = sum_c W[cout, c] x[c, p] is a GEMM whose B operand needs, per lane, 8 consecutive channels of ONE pixel: with planar
// (NCHW) activations that is 8 coalesced 4-byte loads (32 consecutive pixels of a channel = 128 bytes per half-wave) -- no LDS staging,
// no transposition.  A wave owns 32 consecutive pixels x all couts (<= 4 accumulators), walks the channels 16 at a time with the next
// step's 8 loads in flight, splits the values into two fp16 parts (4 instructions per pair) and issues 3 MFMAs per cout tile and step.
// The weights are read from the layer's ordinary fp32 packed block (motif_conv2d_pack: nothing new in the blob), multiplied by 2^8,
// split and laid out as A fragments in LDS once per workgroup (<= 80 KB); 8 waves per workgroup, two workgroups per CU, so that other
// waves' loads cover a wave's latency.  Arithmetic, range and the 2^-8 in the epilogue: see conv_wino.hip -- including the round-5 form of
// the low ACTIVATION part (stored times 2^11, multiplied by the high weight part times 2^-11, so that it is a normal fp16 number whenever
// the high part is) and the range status word (a non-finite accumulator ORs bit 0 into MotifConvDesc.status).
// Every per-plane byte offset of a buffer access is part of the VECTOR offset: the scalar offset operand is outside the hardware range
// check, and the kernel relies on that check for channels past Cin, couts past Cout and the masked lanes of a ragged last group.
#include "conv_common.h"

typedef _Float16 pw_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 pw_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned pw_u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int PW_WAVES = 8, PW_MAXT = 4;                // waves per workgroup, cout tiles of 32 per wave
constexpr float kPwScale = 256.f, kPwLoScale = 2048.f;

__device__ __forceinline__ unsigned pw_pk(float a, float b) { const pw_f16x2 h = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(unsigned, h); }
__device__ __forceinline__ float pw_sub_lo(float x, unsigned pk) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x)); return r; }
__device__ __forceinline__ float pw_sub_hi(float x, unsigned pk) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x)); return r; }
// activations: hi = rne(x), lo = rne((x - hi) * 2^11) -- one rounding (v_fma_mixlo / mixhi_f16)
__device__ __forceinline__ void pw_split8_act(const float (&v)[8], pw_u32x4& hi, pw_u32x4& lo, float s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        hi[q] = pw_pk(v[2 * q], v[2 * q + 1]);
        const float r0 = pw_sub_lo(v[2 * q], hi[q]), r1 = pw_sub_hi(v[2 * q + 1], hi[q]);
        unsigned d;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(d) : "v"(r0), "s"(s));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]\n\ts_nop 0" : "+v"(d) : "v"(r1), "s"(s));      // (one wait state behind a high-half write: hipcc does not look into asm -- siren_split.hip)
        lo[q] = d;
    }
}
__device__ __forceinline__ unsigned pw_pk_mul(unsigned a, pw_f16x2 c) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(pw_f16x2, a) * c); }
__device__ __forceinline__ void pw_split8(const float (&v)[8], pw_u32x4& hi, pw_u32x4& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        hi[q] = pw_pk(v[2 * q], v[2 * q + 1]);
        lo[q] = pw_pk(pw_sub_lo(v[2 * q], hi[q]), pw_sub_hi(v[2 * q + 1], hi[q]));
    }
}
}  // namespace

// grid = (workgroups per problem, problems); a workgroup takes pixel groups (32 pixels of one image) round-robin.
template <int MT>
__global__ __launch_bounds__(64 * PW_WAVES) void conv_pw_kernel(ConvArgs a, int KS, int wn, int groups_per_img, long ngroups) {
    extern __shared__ __attribute__((aligned(16))) pw_u32x4 pw_lds[];          // [KS][part 2][MT][64 lanes] | bias [32 MT] floats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, l31 = lane & 31;
    const int pz = blockIdx.y;
    const int Cin = a.Cin_g, Cout = a.Cout, C0 = a.C0;
    const long HW = (long)a.H * a.W;
    {   // weights: fp32 packed block [ncg][Kpad][wn] (row = channel) -> x 2^8 -> two fp16 parts -> A fragments
        const float* wp = a.wp[pz];
        const int Kpad = a.Kpad;
        for (int f = tid; f < KS * MT * 64; f += 64 * PW_WAVES) {
            const int ln = f & 63, t = (f >> 6) % MT, ks = f / (64 * MT);
            const int col = 32 * t + (ln & 31), cg = col / wn, j = col - cg * wn;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = 16 * ks + 8 * (ln >> 5) + e;
                v[e] = (col < Cout && c < Cin) ? wp[((long)cg * Kpad + c) * wn + j] * kPwScale : 0.f;
            }
            pw_u32x4 hi, lo;
            pw_split8(v, hi, lo);
            pw_lds[((ks * 2 + 0) * MT + t) * 64 + ln] = hi;
            pw_lds[((ks * 2 + 1) * MT + t) * 64 + ln] = lo;
        }
    }
    float* bias_s = (float*)(pw_lds + KS * 2 * MT * 64);
    if (tid < 32 * MT) bias_s[tid] = (a.bias[pz] && tid < Cout) ? a.bias[pz][tid] : 0.f;
    __syncthreads();
    const float* in0 = a.in0[pz];
    const float* in1 = a.in1[pz];
    const long in0_bs = a.in0_bs[pz], in1_bs = a.in1_bs[pz], out_bs = a.out_bs[pz], res_bs = a.res_bs[pz];
    const float* res = a.res[pz];
    float* out = a.out[pz];
    const pw_u32x4* wl = pw_lds + lane;
    const unsigned HW4 = (unsigned)HW * 4u;
    float lo_scale = kPwLoScale;
    asm volatile("" : "+s"(lo_scale));
    const pw_f16x2 ws_c = {(_Float16)(1.f / kPwLoScale), (_Float16)(1.f / kPwLoScale)};
    for (long g = (long)blockIdx.x * PW_WAVES + wave; g < ngroups; g += (long)gridDim.x * PW_WAVES) {
        const int img = (int)(g / groups_per_img);
        const long p0 = (g - (long)img * groups_per_img) * 32;
        const bool valid = p0 + l31 < HW;
        const long p = valid ? p0 + l31 : HW - 1;                      // masked lanes read the plane's last pixel, store nothing
        // range-checked buffer loads: a channel past the end of its source reads as zero (Cin need not be a multiple of 16); with two
        // concatenated sources a 16-channel step lies in ONE of them (C0 % 16 == 0, host)
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)(in0 + (long)img * in0_bs), 0, (C0 < Cin ? C0 : Cin) * (int)HW4, 0x00020000);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(in1 ? in1 + (long)img * in1_bs : in0), 0, in1 ? (Cin - C0) * (int)HW4 : 0, 0x00020000);
        const unsigned poff = (unsigned)p * 4u + (unsigned)(8 * hf) * HW4;
        auto fetch = [&](int ks, float (&v)[8]) __attribute__((always_inline)) {
            const int c = 16 * ks;                                     // uniform
            const bool first = c < C0;
            const __amdgpu_buffer_rsrc_t rs = first ? r0 : r1;
            const unsigned vo = poff + (unsigned)(first ? c : c - C0) * HW4;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(vo + (unsigned)e * HW4), 0, 0));
        };
        f32x16 acc[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        float cur[8], nxt[8];
        fetch(0, cur);
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) fetch(ks + 1, nxt);
            pw_u32x4 hi, lo;
            pw_split8_act(cur, hi, lo, lo_scale);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const pw_u32x4 whi = wl[((ks * 2 + 0) * MT + t) * 64], wlo = wl[((ks * 2 + 1) * MT + t) * 64];
                pw_u32x4 whs;                            // 2^-11 x the high weight part (exact while normal): the partner of the scaled low activation part
#pragma unroll
                for (int q = 0; q < 4; ++q) whs[q] = pw_pk_mul(whi[q], ws_c);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pw_f16x8, wlo), __builtin_bit_cast(pw_f16x8, hi), acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pw_f16x8, whi), __builtin_bit_cast(pw_f16x8, hi), acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pw_f16x8, whs), __builtin_bit_cast(pw_f16x8, lo), acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) cur[e] = nxt[e];
        }
        // epilogue: x 2^-8 + bias, residual before (mode 1) or after (mode 2) the activation (none / ReLU / leaky ReLU: host); the mode and the
        // activation are uniform.  Range-checked buffer accesses (records = Cout planes): a cout past the end and the masked lanes of a ragged
        // last group (offset 2^31) store nothing and read zero -- one address register for all 16 MT accesses, the plane offset is scalar.
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (long)img * out_bs), 0, Cout * (int)HW4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(res ? res + (long)img * res_bs : out), 0, res ? Cout * (int)HW4 : 0, 0x00020000);
        const unsigned eoff = valid ? (unsigned)p * 4u + (unsigned)(4 * hf) * HW4 : 0x80000000u;
        const int rm = a.res_mode, act = a.act;
        // range status word: an operand beyond fp16's range is packed as inf and makes EVERY cout of its pixel non-finite (inf x 0 = NaN):
        // register 0 of tile 0 (couts 0 / 4 x the group's 32 pixels) sees them all
        if (a.status && __builtin_amdgcn_class(acc[0][0], 0x207)) atomicOr(a.status, 1u);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            __builtin_amdgcn_sched_barrier(0);           // one cout tile at a time (keeps the other tiles' residual values out of the registers)
            float rv[16];
            if (rm) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (int)(eoff + (unsigned)(32 * t + (r & 3) + 8 * (r >> 2)) * HW4), 0, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hf;
                float v = fmaf(acc[t][r], 1.f / kPwScale, bias_s[co]);
                if (rm == 1) v += rv[r];
                if (act == MOTIF_ACT_RELU) v = v > 0.f ? v : 0.f;
                else if (act == MOTIF_ACT_LRELU) v = v > 0.f ? v : 0.1f * v;
                if (rm == 2) v += rv[r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, (int)(eoff + (unsigned)(32 * t + (r & 3) + 8 * (r >> 2)) * HW4), 0, 0);
            }
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
namespace {
int pw_cu_count() {
    static int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus;
}
}  // namespace

// The layer's blob is the fp32 engine's (pack and forward agree from the desc alone: nothing to add); the choice is per launch.
bool motif_conv_pw_eligible(const MotifConvDesc* d, const ConvArgs& a, int P) {
    if (d->mma != 7 || motif_opt(MOTIF_OPT_CONV_ENGINE) == 6) return false;
    if (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0 || d->dil != 1 || d->groups != 1) return false;
    const int Cin = d->C0 + d->C1;
    if (d->Cout > 32 * PW_MAXT || d->Cout < 17 || Cin < 8) return false;      // narrower layers: conv_direct.hip / the fp32 engine
    if (d->act_split > 0 || (d->act != MOTIF_ACT_NONE && d->act != MOTIF_ACT_RELU && d->act != MOTIF_ACT_LRELU) || d->res_mode > 2) return false;
    if (d->C1 > 0 && (d->C0 & 15)) return false;         // a 16-channel step lies in one source
    if ((long)(Cin > d->Cout ? Cin : d->Cout) * d->H * d->W * 4 >= 0x7fffffffL) return false;     // 32-bit byte offsets inside a tensor
    if ((long)d->H * d->W < 32 || (long)d->H * d->W >= (1L << 30)) return false;
    const int KS = (Cin + 15) / 16, MT = (d->Cout + 31) / 32;
    if ((size_t)KS * 2 * MT * 64 * 16 > 80 * 1024) return false;
    (void)a; (void)P;
    return true;
}

int motif_conv_pw_launch(const MotifConvDesc* d, ConvArgs& a, int P, hipStream_t s) {
    const int Cin = d->C0 + d->C1;
    a.Ho = d->H; a.Wo = d->W; a.Cin_g = Cin; a.Cout_g = d->Cout; a.Cout = d->Cout;
    a.Kpad = 2 * ((Cin + 1) / 2);                       // rows of the fp32 packed block (conv_igemm.hip: plan_conv, T = 1)
    const int wn = d->Cout > 32 ? 64 : 32;              // its row width
    const int KS = (Cin + 15) / 16, MT = (d->Cout + 31) / 32;
    const long HW = (long)d->H * d->W;
    const int gpi = (int)((HW + 31) / 32);
    const long ngroups = (long)gpi * d->N;
    const size_t lds = (size_t)KS * 2 * MT * 64 * 16 + (size_t)32 * MT * 4;
    long blocks = (ngroups + PW_WAVES - 1) / PW_WAVES;
    const long cap = 2L * pw_cu_count();                 // two workgroups per CU, each walking its share of the pixel groups
    if (blocks > cap) blocks = cap;
    dim3 grid((unsigned)blocks, (unsigned)P, 1);
#define MOTIF_LAUNCH_PW(MTV)                                                                                              \
    do {                                                                                                                  \
        hipError_t e = hipFuncSetAttribute((const void*)conv_pw_kernel<MTV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                                               \
        conv_pw_kernel<MTV><<<grid, 64 * PW_WAVES, lds, s>>>(a, KS, wn, gpi, ngroups);                                   \
    } while (0)
    if (MT == 1) MOTIF_LAUNCH_PW(1); else if (MT == 2) MOTIF_LAUNCH_PW(2); else if (MT == 3) MOTIF_LAUNCH_PW(3); else MOTIF_LAUNCH_PW(4);
#undef MOTIF_LAUNCH_PW
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

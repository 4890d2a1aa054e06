// The three SIREN MLPs on the 16-bit matrix cores with fp32-equivalent arithmetic, register-chained like siren.hip.  Template
// parameter NP: 3 = 3-way bf16 split, 6 products, fp32 accumulate (see conv_split.hip for the arithmetic; pre = 2); 2 = 2-way fp16 split,
// 3 products (round 4, pre = 3, the default: see SProd below and conv_wino.hip's header -- the operands here are sines, coordinates and
// normalised sums, i.e. made for fp16's range).  The comments below give the three-part figures.
//
// v_mfma_f32_32x32x16_bf16: lane (p = lane&31, hf = lane>>5) supplies 8 consecutive-k bf16 values of pixel p; the
// C/D layout is the fp32 one (row m = 32t + (r&3) + 8(r>>2) + 4hf in register r of output tile t).  Registers
// r = 8u..8u+7 of tile t, split into three packed-bf16 quads, ARE the B fragments of k-step (t,u) of the next layer;
// the packing permutes each layer's K order to match:  k(t,u,hf,e) = 32t + (e&3) + 8(2u + (e>>2)) + 4hf.
// The first layer's LR part comes precomputed (pre=1: a 1x1 conv at LR resolution seeds the accumulator), its
// remaining inputs (coordinates, t, normalised splat accumulator) are split on the fly in natural K order.
// The sine and the splitting run on the VALU while the other wave of the SIMD is in its MFMA stretch (bf16 MFMAs,
// unlike the fp32 ones, do not occupy the vector ALU).  Heads that are 3 wide stay on the VALU in fp32.
// Weights: split at pack time into A fragments [k-step][part][out tile][lane] x 8 bf16, resident in LDS (flow
// 133 KB, imnet 131 KB + head streamed from L2, synth 152 KB + first layer streamed from L2).
#include "siren_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {
// (weight part, activation part) of the products, small terms first.  NP = 3: three bf16 parts per operand, six products.
// NP = 2 (round 4): two fp16 parts per operand (hi = rne(x), lo = rne(x - hi): 22+ significant bits while both parts are normal
// numbers), three products -- half the matrix instructions.  The operands suit fp16: every hidden activation is a sine, the
// first layers' extra inputs are coordinates / t / normalised sums of O(1); the weights (x 30 / 2 pi, magnitude 0.02 .. 0.1) are
// packed times 2^8 so that their low parts are normal fp16 numbers too, every accumulator then holds 2^8 x its sum (exact) and
// the sine (or the store of imnet's linear head) multiplies by 2^-8.
template <int NP> struct SProd;
template <> struct SProd<3> {
    static constexpr int n = 6;
    static constexpr int w[6] = {2, 0, 1, 1, 0, 0};
    static constexpr int x[6] = {0, 2, 1, 0, 1, 0};
};
template <> struct SProd<2> {
    static constexpr int n = 3;
    static constexpr int w[3] = {1, 0, 0};
    static constexpr int x[3] = {0, 1, 0};
};
template <int NP> constexpr bool last_use_w(int k) {      // product k is the last of a k-step to read its weight part
    for (int j = k + 1; j < SProd<NP>::n; ++j) if (SProd<NP>::w[j] == SProd<NP>::w[k]) return false;
    return true;
}
template <int NP> constexpr float kSirenScale = NP == 2 ? 256.f : 1.f;

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    bf16x2 p = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

__device__ __forceinline__ unsigned pk_f16(float a, float b) { const f16x2v h = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(unsigned, h); }
// x - (float)half of a packed pair as ONE mixed-precision FMA (half * -1.0 + x; the product by -1 is exact)
// s_nop 0 behind every inline-assembly mixed-precision instruction (round 6).  hipcc's hazard recognizer inserts the wait states gfx940+
// needs between a VALU producer and its consumer -- among them ONE wait state behind an instruction that writes only the high half of its
// destination (v_fma_mixhi_f16: LLVM's "dst_sel forwarding hazard") -- for its own instructions, but it does not look into an asm
// statement.  Without the wait state the consumer that happens to sit in the next issue slot reads the register's OLD contents, depending on
// how the two waves of the SIMD interleave: the flow kernel was irreproducible from run to run (a third of the pixels by up to 4e-6; found when
// an unrelated change moved the code object and tests/test_model_gpu.py::test_default_precontracted_stage... began to fail; it is also what
// round 5 met as "the sine without its fract is not reproducible").  `-DSIREN_DBG_*` variants: nops in FRONT of the asm statements change nothing,
// one wait state BEHIND them makes 6 runs x 3 stagger settings bit-identical, at no measurable cost (0.825 ms either way).
#define SIREN_ASM_PRE
#define SIREN_ASM_POST "\n\ts_nop 0"
__device__ __forceinline__ float sub_f16_lo(float x, unsigned pk) { float r; asm(SIREN_ASM_PRE "v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" SIREN_ASM_POST : "=v"(r) : "v"(pk), "v"(x)); return r; }
__device__ __forceinline__ float sub_f16_hi(float x, unsigned pk) { float r; asm(SIREN_ASM_PRE "v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" SIREN_ASM_POST : "=v"(r) : "v"(pk), "v"(x)); return r; }

// one value pair -> its NP packed parts
template <int NP>
__device__ __forceinline__ void split_pair(float x0, float x1, u32x4 (&out)[NP], int q) {
    if constexpr (NP == 2) {
        // hi = rne16(x); lo = rne16(x - hi) by ONE mixed-precision FMA per value that rounds straight into its half of the packed register
        // (v_fma_mixlo / mixhi_f16: half * -1.0 + x; x - hi is exact in fp32, so this is the single rounding the two-step form
        // [v_fma_mix_f32, v_cvt_pk_f16_f32] made: the same bits, three instructions per pair instead of four -- round 6)
        const unsigned hi = pk_f16(x0, x1);
        out[0][q] = hi;
#ifdef SIREN_SPLIT4
        out[1][q] = pk_f16(sub_f16_lo(x0, hi), sub_f16_hi(x1, hi));
#else
        unsigned lo;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(x0));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(x1));
        out[1][q] = lo;
#endif
    } else {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const unsigned pk = pk_bf16(x0, x1);
            out[p][q] = pk;
            if (p + 1 < NP) { x0 -= bf_lo(pk); x1 -= bf_hi(pk); }
        }
    }
}

template <int NP>
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&out)[NP]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) split_pair<NP>(v[2 * q], v[2 * q + 1], out, q);
}

template <int NP>
__device__ __forceinline__ f32x16 mfma16(u32x4 w, u32x4 x, f32x16 c) {
    if constexpr (NP == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
}

// The sine layers compute sin(30 * (W h + b)) (SIREN.py:44-45, omega_0 = 30).  Here every layer that feeds a sine is packed with its
// weights and bias multiplied by 30 / 2pi (in fp64, before the 3-way bf16 split), so its accumulator holds the argument IN TURNS and the
// sine is ONE v_sin_f32 (the hardware sine takes turns and reduces the argument itself; until round 6 a v_fract_f32 stood in front of it)
// instead of the six instructions of a scaled Cody-Waite reduction.  Accuracy: the accumulator is an fp32 sum either way -- its rounding (|x| * 2^-24 in radians, x the argument)
// is the error of the argument in both forms; what is dropped is only the reference's extra rounding of 30 * (W h + b).  Measured
// against the oracle in tests/test_kernels_gpu.py (same tolerances as before) and end to end in bench.py's parity block.
#define SIREN_TURNS 4.774648292756860                         // 30 / (2 pi)
template <int NP>
__device__ __forceinline__ float sin_turns(float t) {
    if constexpr (NP == 2) t *= 1.f / kSirenScale<2>;     // the accumulators of the two-part form carry 2^8 x the argument
    // No v_fract in front (round 6): gfx950's v_sin_f32 reduces its argument itself -- against sin(2 pi x) in fp64 over +-1 ... +-3e7 turns the
    // maximum error is 1.1-1.25e-7 with and without it (round 5's measurement; 6 % of the results differ in the last bit) -- and it is one of ~5
    // vector instructions per activation: flow_imnet 0.823 -> 0.787 ms, imnet 0.492 -> 0.479, synth 0.954 -> 0.945 (N = 3, c2 size).
    // Round 5 had built exactly this and dropped it because the two-part flow kernel then differed from run to run; the cause was the missing
    // wait state behind the inline-assembly v_fma_mixhi_f16 (SIREN_ASM_POST above), not the sine: with it, six launches x three stagger settings
    // are bit-identical in both forms (tools/siren_determinism.py flow).  -DSIREN_FRACT restores the old form.
#ifdef SIREN_FRACT
    return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(t));
#else
    return __builtin_amdgcn_sinf(t);
#endif
}

template <int MODE, int NP> struct SLayout {
    static constexpr bool SYN = MODE >= MODE_SYNTH;       // both synth forms: five linear layers, first layer streamed
    static constexpr int KS0 = (MODE == MODE_SYNTH) ? 9 : 1;
    // fragment section, in u32x4 (16-byte) units: layer = [KS][NP parts][MT][64 lanes]
    static constexpr long F_W0 = 0;
    static constexpr long F_W1 = F_W0 + (long)KS0 * NP * 2 * 64;
    static constexpr long F_W1B = F_W1 + 4L * NP * 2 * 64;
    static constexpr long F_W2 = F_W1B + (SYN ? 4L * NP * 2 * 64 : 0);
    static constexpr long F_W3 = F_W2 + 4L * NP * 8 * 64;
    static constexpr long F_END = F_W3 + (MODE == MODE_IMNET ? 16L * NP * 2 * 64 : 0);
    // float section (after the fragments): B1, [B1b], B2, then B3 (imnet) or the fp32 VALU head Wv[3][32][2][4] + bias[4]
    static constexpr int O_B1 = 0;
    static constexpr int O_B1B = 64;
    static constexpr int O_B2 = O_B1B + (SYN ? 64 : 0);
    static constexpr int O_HEAD = O_B2 + 256;
    static constexpr int NFLOATS = O_HEAD + (MODE == MODE_IMNET ? 64 : 3 * 32 * 8 + 4);
    // LDS residency: synth streams its first layer, imnet its head, from L2
    static constexpr long LDS_F0 = SYN ? F_W1 : 0;
    static constexpr long LDS_F1 = (MODE == MODE_IMNET) ? F_W3 : F_END;
    static constexpr long LDS_BYTES = (LDS_F1 - LDS_F0) * 16 + (long)NFLOATS * 4;
    static constexpr long TOTAL_FLOATS = F_END * 4 + NFLOATS;
};

template <int TP, int MT>
__device__ __forceinline__ void init_bias_s(f32x16 (&acc)[TP][MT], const float* bp, int hf) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float b = bp[(t * 16 + r) * 2 + hf];
#pragma unroll
            for (int p = 0; p < TP; ++p) acc[p][t][r] = b;
        }
}

// weight fragments of one k-step for MT output tiles: wk[(part * MTW + t0 + t) * 64]
template <int MT, int MTW, int NP>
__device__ __forceinline__ void load_w(u32x4 (&w)[MT][NP], const u32x4* wk, int t0) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int part = 0; part < NP; ++part) w[t][part] = wk[(part * MTW + t0 + t) * 64];
}

// one k-step.  Products outermost so that consecutive MFMAs go to different accumulators.
template <int MT, int TP, int NP>
__device__ __forceinline__ void mfma_step(const u32x4 (&w)[MT][NP], const u32x4 (&x)[TP][NP], f32x16 (&acc)[TP][MT]) {
#pragma unroll
    for (int k = 0; k < SProd<NP>::n; ++k)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int p = 0; p < TP; ++p)
                acc[p][t] = mfma16<NP>(w[t][SProd<NP>::w[k]], x[p][SProd<NP>::x[k]], acc[p][t]);
}

// First-layer k-steps: their inputs are not sines but network data of any magnitude (coordinates, t, splat statistics, in the literal
// synth form the normalised feature sums), so the two-part form uses the scaled low part of conv_wino.hip (round 5): the activation is
// split by split8_in (lo_s = rne((x - hi) * 2^11): a normal fp16 number whenever hi is one) and lo_s meets 2^-11 x the high weight part,
// formed here by four v_pk_mul_f16 per fragment (exact while normal).  The hidden layers keep the plain split: their inputs are sines,
// |x| <= 1, and the plain low part's absolute error of 2^-25 is then below fp32's own rounding of the operand.
constexpr float kSirenLoScale = 2048.f;
__device__ __forceinline__ unsigned pk_mul_f16(unsigned a, f16x2v c) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(f16x2v, a) * c); }
template <int NP>
__device__ __forceinline__ void split8_in(const float (&v)[8], u32x4 (&out)[NP]) {
    if constexpr (NP == 2) {
        float s = kSirenLoScale;
        asm volatile("" : "+s"(s));                      // scalar register: the mix instructions take no literal
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned hi = pk_f16(v[2 * q], v[2 * q + 1]);
            const float r0 = sub_f16_lo(v[2 * q], hi), r1 = sub_f16_hi(v[2 * q + 1], hi);
            unsigned d;
            asm(SIREN_ASM_PRE "v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" SIREN_ASM_POST : "=v"(d) : "v"(r0), "s"(s));
            asm(SIREN_ASM_PRE "v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" SIREN_ASM_POST : "+v"(d) : "v"(r1), "s"(s));
            out[0][q] = hi;
            out[1][q] = d;
        }
    } else split8<NP>(v, out);
}

template <int MT, int MTW, int TP, int NP>
__device__ __forceinline__ void split_step(const u32x4 (&x)[TP][NP], f32x16 (&acc)[TP][MT], const u32x4* wk, int t0) {
    u32x4 w[MT][NP];
    load_w<MT, MTW, NP>(w, wk, t0);
    if constexpr (NP == 2) {                             // x from split8_in: (w lo, x hi), (w hi, x hi), (2^-11 w hi, 2^11 x lo)
        const f16x2v c = {(_Float16)(1.f / kSirenLoScale), (_Float16)(1.f / kSirenLoScale)};
        u32x4 whs[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) whs[t][q] = pk_mul_f16(w[t][0][q], c);
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int p = 0; p < TP; ++p)
                    acc[p][t] = mfma16<2>(k == 0 ? w[t][1] : k == 1 ? w[t][0] : whs[t], x[p][k == 2 ? 1 : 0], acc[p][t]);
    } else mfma_step<MT, TP, NP>(w, x, acc);
}

// KS k-steps; `wfirst` holds the fragments of k-step 0 (requested by the caller before the preceding sine stretch),
// the fragments of k-step ks+1 are requested before the MFMAs of k-step ks
template <int KS, int MT, int MTW, int TP, int NP>
__device__ __forceinline__ void split_layer(const u32x4 (&h)[TP][KS][NP], f32x16 (&acc)[TP][MT], const u32x4* wp, int t0,
                                            const u32x4 (&wfirst)[MT][NP]) {
    u32x4 w[2][MT][NP];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int part = 0; part < NP; ++part) w[0][t][part] = wfirst[t][part];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) load_w<MT, MTW, NP>(w[(ks + 1) & 1], wp + (long)(ks + 1) * NP * MTW * 64, t0);
        u32x4 hx[TP][NP];
#pragma unroll
        for (int p = 0; p < TP; ++p)
#pragma unroll
            for (int part = 0; part < NP; ++part) hx[p][part] = h[p][ks][part];
        mfma_step<MT, TP, NP>(w[ks & 1], hx, acc);
    }
}

// h[p][2t+u] = split(sin(2 pi * acc[p][t][8u .. 8u+7]))   (accumulators are in turns)
template <int TP, int MT, int NP>
__device__ __forceinline__ void sine_split(const f32x16 (&acc)[TP][MT], u32x4 (&h)[TP][2 * MT][NP]) {
#pragma unroll
    for (int p = 0; p < TP; ++p)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = sin_turns<NP>(acc[p][t][8 * u + e]);
                split8<NP>(v, h[p][2 * t + u]);
                __builtin_amdgcn_sched_barrier(0);
            }
}

#ifdef MOTIF_SIREN_DBG
template <int TP, int MT, int NP>
__device__ __forceinline__ void fake_split(const f32x16 (&acc)[TP][MT], u32x4 (&h)[TP][2 * MT][NP]) {
#pragma unroll
    for (int p = 0; p < TP; ++p)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int part = 0; part < NP; ++part)
#pragma unroll
                    for (int q = 0; q < 4; ++q) h[p][2 * t + u][part][q] = __builtin_bit_cast(unsigned, acc[p][t][8 * u + 2 * q + (part & 1)]);
}
template <int TP, int MT>
__device__ __forceinline__ void fake_f32(const f32x16 (&acc)[TP][MT], float (&h)[TP][MT * 16]) {
#pragma unroll
    for (int p = 0; p < TP; ++p)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[p][t * 16 + r] = acc[p][t][r];
}
#endif

template <int TP, int MT, int NP>
__device__ __forceinline__ void sine_f32(const f32x16 (&acc)[TP][MT], float (&h)[TP][MT * 16]) {
#pragma unroll
    for (int p = 0; p < TP; ++p)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                h[p][t * 16 + r] = sin_turns<NP>(acc[p][t][r]);
            }
}

// One sine/split unit of k-step ks of the layer input `src` (pre-activations): units 0..7 = one sine each,
// units 8..11 = split of one value pair into its NP packed parts.
template <int TP, int NP>
__device__ __forceinline__ void ss_unit(int u, const f32x16 (&src)[TP][2], int ks, float (&tmp)[TP][8], u32x4 (&hx)[TP][NP]) {
    const int t = ks >> 1, uu = ks & 1;
#pragma unroll
    for (int p = 0; p < TP; ++p) {
        if (u < 8) {
            tmp[p][u] = sin_turns<NP>(src[p][t][8 * uu + u]);
        } else {
            const int q = u - 8;
            split_pair<NP>(tmp[p][2 * q], tmp[p][2 * q + 1], hx[p], q);
        }
    }
}

// dst += W . sin(30 src) for a 64-wide input: the sine + split of k-step ks+1 (12 VALU units) is interleaved, unit by
// unit, with the MFMAs of k-step ks (12 with three parts: one unit each; 6 with two: two units each), so only the first k-step's
// vector work and the last k-step's MFMAs are exposed.
// One set of weight fragments, refilled in place after the last use of each part.  KEEP: also store the split input.
template <int MTW, int TP, bool KEEP, int NP>
__device__ __forceinline__ void fused_layer(const f32x16 (&src)[TP][2], f32x16 (&dst)[TP][2], const u32x4* wp, int t0,
                                            u32x4 (&keep)[TP][4][NP]) {
    using SP = SProd<NP>;
    constexpr int UPM = 12 / (2 * SP::n);                  // vector units per MFMA
    u32x4 hx[2][TP][NP];
    float tmp[TP][8];
    u32x4 w[2][NP];
    load_w<2, MTW, NP>(w, wp, t0);
#pragma unroll
    for (int u = 0; u < 12; ++u) ss_unit<TP, NP>(u, src, 0, tmp, hx[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if constexpr (KEEP) {
#pragma unroll
            for (int p = 0; p < TP; ++p)
#pragma unroll
                for (int part = 0; part < NP; ++part) keep[p][ks][part] = hx[ks & 1][p][part];
        }
        const u32x4* wnext = wp + (long)(ks + 1) * NP * MTW * 64;
#pragma unroll
        for (int k = 0; k < SP::n; ++k) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int p = 0; p < TP; ++p)
                    dst[p][t] = mfma16<NP>(w[t][SP::w[k]], hx[ks & 1][p][SP::x[k]], dst[p][t]);
                if (ks < 3) {
#pragma unroll
                    for (int j = 0; j < UPM; ++j) ss_unit<TP, NP>(UPM * (2 * k + t) + j, src, ks + 1, tmp, hx[(ks + 1) & 1]);
                }
                __builtin_amdgcn_sched_barrier(0x180);               // only DS instructions may cross
            }
            if (ks < 3 && last_use_w<NP>(k)) {                        // last use of part w[k] in this k-step
#pragma unroll
                for (int t = 0; t < 2; ++t) w[t][SP::w[k]] = wnext[(SP::w[k] * MTW + t0 + t) * 64];
            }
        }
    }
}
}  // namespace

#ifdef MOTIF_TRACE
__device__ long long g_strace[256 * 8 * 16];
extern "C" int motif_debug_siren_trace(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_strace), sizeof(long long) * n); }
#define PH(i) do { const long long now_ = __builtin_amdgcn_s_memtime(); ph[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define PH(i)
#endif

template <int MODE, int TP, int NP>
__global__ __launch_bounds__(SIREN_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void siren_split_kernel(SirenArgs a) {
#ifdef MOTIF_TRACE
    long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __attribute__((aligned(16))) u32x4 ldsv[];
    using L = SLayout<MODE, NP>;
    using SP = SProd<NP>;
    constexpr float SCL = kSirenScale<NP>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, l31 = lane & 31;
    const u32x4* gfr = (const u32x4*)a.packed;
    float* ldsf0 = (float*)(ldsv + (L::LDS_F1 - L::LDS_F0));
    {
        for (long i = tid; i < L::LDS_F1 - L::LDS_F0; i += SIREN_THREADS) ldsv[i] = gfr[L::LDS_F0 + i];
        const float* gfl = (const float*)(gfr + L::F_END);
        for (int i = tid; i < L::NFLOATS; i += SIREN_THREADS) ldsf0[i] = gfl[i];
    }
    __syncthreads();
    // start the second wave of every SIMD late so that one is in its sine/split (VALU) stretch while the other
    // issues MFMAs
    if (wave >= SIREN_WAVES / 2) {
        const int naps = a.stagger & 255;
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
    }
#ifdef MOTIF_SIREN_DBG
    const int grp = wave >= SIREN_WAVES / 2;
    const bool skipV = (a.stagger >> (8 + grp)) & 1, skipM = (a.stagger >> (10 + grp)) & 1;
#define DBG_V(stmt_real, stmt_skip) do { if (skipV) { stmt_skip; } else { stmt_real; } } while (0)
#define DBG_M(stmt) do { if (!skipM) { stmt; } } while (0)
#else
#define DBG_V(stmt_real, stmt_skip) do { stmt_real; } while (0)
#define DBG_M(stmt) do { stmt; } while (0)
#endif
    const u32x4* lw1_0 = ldsv + (L::F_W1 - L::LDS_F0) + lane;
    const u32x4* lw1b_0 = ldsv + (L::F_W1B - L::LDS_F0) + lane;
    const u32x4* lw2_0 = ldsv + (L::F_W2 - L::LDS_F0) + lane;
    const u32x4* w0 = (L::SYN ? gfr + L::F_W0 : ldsv + (L::F_W0 - L::LDS_F0)) + lane;

    const long Q = (long)a.HH * a.WW;
    const long HWl = (long)a.H * a.W;
    const int tiles_per_img = (int)((Q + 32 * TP - 1) / (32 * TP));
    const long total = (long)a.NB * tiles_per_img;

    // 32-bit tile bookkeeping (Q < 2^31).  The LR partial of the NEXT tile is gathered into acc0 as soon as the current
    // tile has consumed it, so its L2 latency hides behind the rest of the tile.
    struct Tile { int img; int pp[TP], Y[TP], X[TP], lr[TP]; bool valid[TP]; };
    auto locate = [&](unsigned work, Tile& T) {
        const unsigned w = __builtin_amdgcn_readfirstlane(work);
        T.img = (int)(w / (unsigned)tiles_per_img);
        const unsigned tl = w - (unsigned)T.img * (unsigned)tiles_per_img;
#pragma unroll
        for (int p = 0; p < TP; ++p) {
            const unsigned pp = tl * (32 * TP) + 32 * p + l31;
            T.pp[p] = (int)pp;
            T.valid[p] = pp < (unsigned)Q;
            const unsigned pc = T.valid[p] ? pp : (unsigned)Q - 1;
            T.Y[p] = (int)(pc / (unsigned)a.WW);
            T.X[p] = (int)(pc - (unsigned)T.Y[p] * (unsigned)a.WW);
            T.lr[p] = a.iy[T.Y[p]] * a.W + a.ix[T.X[p]];
        }
    };
    auto gather = [&](f32x16 (&acc)[TP][2], const Tile& T) {
        const int ilr = (MODE == MODE_FLOW || L::SYN) ? T.img / a.N : T.img;
#pragma unroll
        for (int p = 0; p < TP; ++p) {
            const float* gp = a.src_lr + (long)ilr * 64 * HWl + T.lr[p] + (long)(4 * hf) * HWl;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][t][r] = gp[(long)(32 * t + (r & 3) + 8 * (r >> 2)) * HWl] * (float)(SIREN_TURNS * SCL);   // LR partial -> turns (x 2^8 in the two-part form)
        }
    };
    const unsigned utotal = (unsigned)total, stride = gridDim.x * SIREN_WAVES;
    unsigned work = blockIdx.x * SIREN_WAVES + wave;
    Tile cur, nxt;
    f32x16 acc0[TP][2];
    if (work < utotal) {
        locate(work, cur);
        if constexpr (!L::SYN) gather(acc0, cur);
    }
    for (; work < utotal; work += stride) {
        // LDS contents never change after the staging barrier, so the compiler would hoist bias / head-weight reads out
        // of this loop and spill them; an opaque zero offset per iteration keeps them where they are used
        int lds_o = 0;
        asm volatile("" : "+s"(lds_o));
        const float* ldsf = ldsf0 + lds_o;
        const u32x4* lw1 = lw1_0 + lds_o;
        const u32x4* lw1b = lw1b_0 + lds_o;
        const u32x4* lw2 = lw2_0 + lds_o;
        const int img = cur.img;
        const int (&Y)[TP] = cur.Y;
        const int (&X)[TP] = cur.X;
        const bool has_next = work + stride < utotal;
        PH(0);                                               // tile bookkeeping
        // ------------------------------------------------ layer 0: the LR partial (gathered ahead) seeds the accumulator
        if constexpr (L::SYN) gather(acc0, cur);    // synth: no spare registers to carry it across the tile
        if constexpr (MODE == MODE_IMNET || MODE == MODE_FLOW) {
            // natural K order: imnet k0 = rel_y, k1 = rel_x; flow k0 = t, k1 = rel_y, k2 = rel_x (lower half-wave)
            u32x4 x[TP][NP];
#pragma unroll
            for (int p = 0; p < TP; ++p) {
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if constexpr (MODE == MODE_FLOW) { v[0] = a.times[img % (a.B * a.N)]; v[1] = a.rel_y[Y[p]]; v[2] = a.rel_x[X[p]]; }
                else { v[0] = a.rel_y[Y[p]]; v[1] = a.rel_x[X[p]]; }
                if (hf) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; }
                split8_in<NP>(v, x[p]);
            }
            split_step<2, 2, TP, NP>(x, acc0, w0, 0);
        } else if constexpr (MODE == MODE_SYNTHC) {
            // pre-contracted first layer (splat.hip, PRE form): the accumulator already holds W0[:, 0:130] . (splat sums), so
            // the pre-activation is  LR partial + sums / warped_z + W0[:, 130:133] . extra + W0[:, 197] t  -- 32 plane loads in
            // C/D order and ONE k-step (natural K order: 0 zmax, 1 cnt/16, 2 wz_/cnt_, 3 t) instead of nine.
            const float tval = a.times[img];
            u32x4 x[TP][NP];
#pragma unroll
            for (int p = 0; p < TP; ++p) {
                const float* A = a.acc + (long)img * 67 * Q + (cur.valid[p] ? cur.pp[p] : (int)Q - 1);
                float wz = A[64 * Q];
                const float zmax = A[65 * Q], cnt = A[66 * Q];
                const float* ap = A + (long)(4 * hf) * Q;
                float sv[2][16];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sv[t][r] = ap[(long)(32 * t + (r & 3) + 8 * (r >> 2)) * Q];
                if (wz == 0.f) wz = 1.0f;                                 // Ours.py:813
                const float cnt_ = (cnt == 0.f) ? 1.0f : cnt;             // Ours.py:828
                const float wz_ = (wz == 1.0f) ? 0.f : wz;                // Ours.py:830
                const float iw = (1.0f / wz) * (float)(SIREN_TURNS * SCL);          // the normalised sums enter the first layer's pre-activation: in turns
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc0[p][t][r] = fmaf(sv[t][r], iw, acc0[p][t][r]);
                float d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (!hf) { d[0] = zmax; d[1] = cnt / 16.0f; d[2] = wz_ / cnt_; d[3] = tval; }
                split8_in<NP>(d, x[p]);
            }
            split_step<2, 2, TP, NP>(x, acc0, w0, 0);
        } else {
            // synth, natural K order: k < 130 sum/wz; 130 zmax; 131 cnt/16; 132 wz_/cnt_; 133 t; 134..143 zero
            const float tval = a.times[img];
            const float* A[TP];
            float wz[TP], cnt[TP], cnt_[TP], wz_[TP];
#pragma unroll
            for (int p = 0; p < TP; ++p) {
                A[p] = a.acc + (long)img * 133 * Q + (cur.valid[p] ? cur.pp[p] : (int)Q - 1);
                wz[p] = A[p][130 * Q];
                cnt[p] = A[p][132 * Q];
                if (wz[p] == 0.f) wz[p] = 1.0f;                           // Ours.py:813
                cnt_[p] = (cnt[p] == 0.f) ? 1.0f : cnt[p];                // Ours.py:828
                wz_[p] = (wz[p] == 1.0f) ? 0.f : wz[p];                   // Ours.py:830
            }
            const float* ap[TP];
            float v[TP][8];
#pragma unroll
            for (int p = 0; p < TP; ++p) {
                ap[p] = A[p] + (long)(8 * hf) * Q;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[p][e] = ap[p][(long)e * Q];
            }
#pragma unroll 1
            for (int ks = 0; ks < 8; ++ks) {         // k = 16ks + 8hf + e < 128: accumulator planes, next step's loads in flight
                float vn[TP][8];
#pragma unroll
                for (int p = 0; p < TP; ++p) {
                    ap[p] += 16 * Q;
                    if (ks < 7) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) vn[p][e] = ap[p][(long)e * Q];
                    }
                }
                u32x4 x[TP][NP];
#pragma unroll
                for (int p = 0; p < TP; ++p) {
                    float d[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) d[e] = v[p][e] / wz[p];
                    split8_in<NP>(d, x[p]);
                }
                split_step<2, 2, TP, NP>(x, acc0, w0 + (long)ks * NP * 2 * 64, 0);
#pragma unroll
                for (int p = 0; p < TP; ++p)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[p][e] = vn[p][e];
            }
            {
                u32x4 x[TP][NP];
#pragma unroll
                for (int p = 0; p < TP; ++p) {
                    float d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (!hf) {
                        d[0] = A[p][128 * Q] / wz[p];
                        d[1] = A[p][129 * Q] / wz[p];
                        d[2] = A[p][131 * Q];
                        d[3] = cnt[p] / 16.0f;
                        d[4] = wz_[p] / cnt_[p];
                        d[5] = tval;
                    }
                    split8_in<NP>(d, x[p]);
                }
                split_step<2, 2, TP, NP>(x, acc0, w0 + 8L * NP * 2 * 64, 0);
            }
        }
        PH(1);                                               // layer 0 (gather + MFMA issue)
        if constexpr (Net<MODE>::HEAD == 3) {
            float sum[TP][3];
#pragma unroll
            for (int p = 0; p < TP; ++p) sum[p][0] = sum[p][1] = sum[p][2] = 0.f;
            // software pipeline over the four 64-wide chunks, interleaved by hand: after every MFMA of chunk c+1 (matrix
            // pipe) comes one unit of chunk c's vector work (a sine, or four head FMAs), so the wave's own VALU
            // instructions fill the issue slots its MFMAs leave free.  The barriers pin MFMA/VALU order only; LDS
            // reads (weight fragments, head weights) may be hoisted across them.
            // 64 -> 64 (x2 for synth) -> first 64-wide chunk of 64 -> 256, each layer fused with the sine of its input
            u32x4 h2[TP][4][NP];
            f32x16 acc1[TP][2];
            init_bias_s(acc1, ldsf + L::O_B1, hf);
            fused_layer<2, TP, false, NP>(acc0, acc1, lw1, 0, h2);
            if constexpr (!L::SYN) {
                if (has_next) { locate(work + stride, nxt); gather(acc0, nxt); }
            }
            PH(3);
            f32x16 acc2[2][TP][2];
            if constexpr (L::SYN) {
                init_bias_s(acc0, ldsf + L::O_B1B, hf);                   // acc0 is free: reuse as the 1b accumulator
                fused_layer<2, TP, false, NP>(acc1, acc0, lw1b, 0, h2);
                init_bias_s(acc2[0], ldsf + L::O_B2, hf);
                fused_layer<8, TP, true, NP>(acc0, acc2[0], lw2, 0, h2);
                if (has_next) locate(work + stride, nxt);
            } else {
                init_bias_s(acc2[0], ldsf + L::O_B2, hf);
                fused_layer<8, TP, true, NP>(acc1, acc2[0], lw2, 0, h2);
            }
            PH(5);                                           // 64->256 chunk 0 MFMAs
            const float* headw = ldsf + L::O_HEAD;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float hc[TP][32];
                auto valu_unit = [&](int u) {                // 32 sines, then 24 x (4 FMAs of one head output)
                    if (u < 32) {
#pragma unroll
                        for (int p = 0; p < TP; ++p) hc[p][u] = sin_turns<NP>(acc2[c & 1][p][u >> 4][u & 15]);
                    } else if (u < 56) {
                        const int q = (u - 32) / 3, o = (u - 32) % 3;
                        const f32x4 w4 = *(const f32x4*)(headw + ((o * 32 + 8 * c + q) * 2 + hf) * 4);
#pragma unroll
                        for (int p = 0; p < TP; ++p) {
                            sum[p][o] = fmaf(w4[0], hc[p][q * 4 + 0], sum[p][o]);
                            sum[p][o] = fmaf(w4[1], hc[p][q * 4 + 1], sum[p][o]);
                            sum[p][o] = fmaf(w4[2], hc[p][q * 4 + 2], sum[p][o]);
                            sum[p][o] = fmaf(w4[3], hc[p][q * 4 + 3], sum[p][o]);
                        }
                    }
                };
                int unit = 0;
                if (c < 3) {
                    f32x16 (&nxt)[TP][2] = acc2[(c + 1) & 1];
                    // one set of weight fragments, refilled in place: part PW[k] of the next k-step is requested right
                    // after its last use in this one (products are ordered so that parts retire 2, 1, 0)
                    u32x4 w[2][NP];
                    load_w<2, 8, NP>(w, lw2, 2 * c + 2);
                    init_bias_s(nxt, ldsf + L::O_B2 + (c + 1) * 64, hf);
                    constexpr int UPM = 12 / (2 * SP::n);    // vector units per MFMA: 48 of the 56 under the chunk's MFMAs either way
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const u32x4* wnext = lw2 + (long)(ks + 1) * NP * 8 * 64;
#pragma unroll
                        for (int k = 0; k < SP::n; ++k) {
#pragma unroll
                            for (int t = 0; t < 2; ++t) {
#pragma unroll
                                for (int p = 0; p < TP; ++p)
                                    nxt[p][t] = mfma16<NP>(w[t][SP::w[k]], h2[p][ks][SP::x[k]], nxt[p][t]);
#pragma unroll
                                for (int j = 0; j < UPM; ++j) valu_unit(unit++);
                                if ((unit & 3) == 0) __builtin_amdgcn_sched_barrier(0);      // bound the LDS-read hoisting
                                else __builtin_amdgcn_sched_barrier(0x180);                   // only DS instructions may cross
                            }
                            if (ks < 3 && last_use_w<NP>(k)) {      // last use of part w[k]
#pragma unroll
                                for (int t = 0; t < 2; ++t) w[t][SP::w[k]] = wnext[(SP::w[k] * 8 + 2 * c + 2 + t) * 64];
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 56; ++u)
                    if (u >= unit) valu_unit(u);
                PH(6);                                       // chunk sine + head (+ next chunk's MFMAs)
            }
            const float* hb = ldsf + L::O_HEAD + 3 * 32 * 8;
#pragma unroll
            for (int p = 0; p < TP; ++p) {
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    sum[p][o] += __shfl_xor(sum[p][o], 32);
                    sum[p][o] += hb[o];
                }
                if (cur.valid[p] && hf == 0) {
                    if constexpr (MODE == MODE_FLOW) {
#pragma unroll
                        for (int o = 0; o < 3; ++o) a.out[((long)img * 3 + o) * Q + cur.pp[p]] = sum[p][o];
                    } else {
                        const int b = img / a.N, n = img % a.N;
                        // range status word: a first-layer input beyond fp16's range (the max plane under a large alpha) is packed as inf
                        // and reaches every output as NaN
                        if constexpr (NP == 2) {
                            if (a.status && (__builtin_amdgcn_class(sum[p][0], 0x207) | __builtin_amdgcn_class(sum[p][1], 0x207) | __builtin_amdgcn_class(sum[p][2], 0x207)))
                                atomicOr(a.status, 1u);
                        }
#pragma unroll
                        for (int o = 0; o < 3; ++o) {
                            float v = sum[p][o];
                            v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
                            a.out[(((long)n * a.B + b) * 3 + o) * Q + cur.pp[p]] = v;
                        }
                    }
                }
            }
        } else {
            u32x4 wn[2][NP];                                  // first fragments of the next layer, in flight during the sine
            load_w<2, 2, NP>(wn, lw1, 0);
            u32x4 h1[TP][4][NP];
            DBG_V(sine_split(acc0, h1), fake_split(acc0, h1));
            if (has_next) { locate(work + stride, nxt); gather(acc0, nxt); }

            PH(2);                                               // sine 1
            // ------------------------------------------------ 64 -> 64 (x2 for synth)
            f32x16 acc1[TP][2];
            init_bias_s(acc1, ldsf + L::O_B1, hf);
            DBG_M((split_layer<4, 2, 2, TP, NP>(h1, acc1, lw1, 0, wn)));
            PH(3);                                               // layer 1 MFMAs
            if constexpr (MODE == MODE_SYNTH) load_w<2, 2, NP>(wn, lw1b, 0); else load_w<2, 8, NP>(wn, lw2, 0);
            u32x4 h2[TP][4][NP];
            DBG_V(sine_split(acc1, h2), fake_split(acc1, h2));
            if constexpr (MODE == MODE_SYNTH) {
                init_bias_s(acc1, ldsf + L::O_B1B, hf);
                DBG_M((split_layer<4, 2, 2, TP, NP>(h2, acc1, lw1b, 0, wn)));
                load_w<2, 8, NP>(wn, lw2, 0);
                DBG_V(sine_split(acc1, h2), fake_split(acc1, h2));
            }

            PH(4);                                               // sine 2 (+ layer 1b)
            // ------------------------------------------------ 64 -> 256 in four 64-wide chunks, each fed to the head
            f32x16 acc3[TP][2];
            const u32x4* w3g = gfr + L::F_W3 + lane;                    // streamed from L2
            init_bias_s(acc3, ldsf + L::O_HEAD, hf);
#pragma unroll 1
            for (int c = 0; c < 4; ++c) {
                f32x16 acc2[TP][2];
                init_bias_s(acc2, ldsf + L::O_B2 + c * 64, hf);
                split_layer<4, 2, 8, TP, NP>(h2, acc2, lw2, 2 * c, wn);
                u32x4 w3[2][NP];
                load_w<2, 2, NP>(w3, w3g + (long)c * 4 * NP * 2 * 64, 0);
                u32x4 hc[TP][4][NP];
                sine_split(acc2, hc);
                if (c < 3) load_w<2, 8, NP>(wn, lw2, 2 * c + 2);
                split_layer<4, 2, 2, TP, NP>(hc, acc3, w3g + (long)c * 4 * NP * 2 * 64, 0, w3);
            }
#pragma unroll
            for (int p = 0; p < TP; ++p) {
                if (!cur.valid[p]) continue;
                if (a.add_lr) {
                    // + the gathered LR term, rounded like the sum the splat kernel used to form (splat.hip, PRE form): u + g in fp32
                    const float* gp = a.add_lr + (long)img * 64 * HWl + cur.lr[p] + (long)(4 * hf) * HWl;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc3[p][t][r] = acc3[p][t][r] * (1.f / SCL) + gp[(long)(32 * t + (r & 3) + 8 * (r >> 2)) * HWl];    // (x 1: exact)
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hf;
                        a.out[((long)img * 64 + m) * Q + cur.pp[p]] = (NP == 2 && !a.add_lr) ? acc3[p][t][r] * (1.f / SCL) : acc3[p][t][r];
                    }
            }
        }
        PH(8);                                               // outputs
        cur = nxt;
    }
#ifdef MOTIF_TRACE
    if (lane == 0 && blockIdx.x < 256)
        for (int i = 0; i < 10; ++i) g_strace[(blockIdx.x * 8 + wave) * 16 + i] = ph[i];
#endif
}

// ---------------------------------------------------------------- packing
// mode 0 imnet (66-64-64-256-64), 1 flow (67-64-64-256-3), 2 synth (198-64-64-64-256-3); w/b: the nn.Linear
// parameters in network order.
struct SplitPackArgs { const float* w[5]; const float* b[5]; int mode; };

template <int MODE, int NP>
__device__ void siren_split_pack_elem(const SplitPackArgs& pa, unsigned short* frags, float* floats, long i) {
    using L = SLayout<MODE, NP>;
    constexpr double SCL = (double)kSirenScale<NP>;            // two-part form: fragments and accumulator biases times 2^8
    constexpr int K0 = Net<MODE>::K0;
    constexpr int NL = L::SYN ? 5 : 4;                              // linear layers incl. the head
    const long nfrag16 = L::F_END * 8;                              // bf16 elements in the fragment section
    if (i < nfrag16) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const long f = i >> 9;                                      // fragment index = ((ks*3 + part)*MT + t) within a layer
        int layer, MT, K, M;
        long fb;
        if (f < L::F_W1 / 64) { layer = 0; fb = 0; MT = 2; K = K0; M = 64; }
        else if (f < L::F_W1B / 64) { layer = 1; fb = L::F_W1 / 64; MT = 2; K = 64; M = 64; }
        else if (f < L::F_W2 / 64) { layer = 2; fb = L::F_W1B / 64; MT = 2; K = 64; M = 64; }
        else if (f < L::F_W3 / 64) { layer = NL - 2; fb = L::F_W2 / 64; MT = 8; K = 64; M = 256; }
        else { layer = NL - 1; fb = L::F_W3 / 64; MT = 2; K = 256; M = 64; }
        const long fl = f - fb;
        const int t = (int)(fl % MT), part = (int)((fl / MT) % NP), ks = (int)(fl / (NP * MT));
        const int hfl = lane >> 5, m = 32 * t + (lane & 31);
        int k;
        if (layer == 0) {
            const int kn = 16 * ks + 8 * hfl + e;                   // natural order of the non-LR inputs
            if (MODE == MODE_IMNET) k = kn < 2 ? 64 + kn : -1;
            else if (MODE == MODE_FLOW) k = kn < 3 ? 64 + kn : -1;
            else if (MODE == MODE_SYNTH) k = kn < 133 ? kn : (kn == 133 ? 197 : -1);
            else k = kn < 3 ? 130 + kn : (kn == 3 ? 197 : -1);           // SYNTHC: extra(3) | t
        } else {
            const int tt = ks >> 1, u = ks & 1;                     // chained order
            k = 32 * tt + (e & 3) + 8 * (2 * u + (e >> 2)) + 4 * hfl;
        }
        float v = (k >= 0 && k < K && m < M) ? pa.w[layer][(long)m * K + k] : 0.f;
        double vd = (double)v * SCL;
        if (!(MODE == MODE_IMNET && layer == NL - 1)) vd *= SIREN_TURNS;     // every fragment layer but imnet's linear head feeds a sine
        unsigned short out = 0;
        if constexpr (NP == 2) {
            for (int p = 0; p <= part; ++p) {
                const _Float16 h = (_Float16)(float)vd;
                out = __builtin_bit_cast(unsigned short, h);
                vd -= (double)(float)h;
            }
        } else {
            v = (float)vd;
            for (int p = 0; p <= part; ++p) {
                const unsigned pk = pk_bf16(v, 0.f);
                out = (unsigned short)(pk & 0xffffu);
                v -= bf_lo(pk);
            }
        }
        frags[i] = out;
        return;
    }
    const int j = (int)(i - nfrag16);
    if (j >= L::NFLOATS) return;
    auto bias_cd = [&](const float* b, int jj, int M) {              // [t][r][hf] order of the C/D layout
        const int hfl = jj & 1, r = (jj >> 1) & 15, t = jj >> 5;
        const int m = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hfl;
        return m < M ? b[m] : 0.f;
    };
    float v;
    if (j < L::O_B1B) v = (float)((double)bias_cd(pa.b[1], j - L::O_B1, 64) * SIREN_TURNS * SCL);
    else if (j < L::O_B2) v = (float)((double)bias_cd(pa.b[2], j - L::O_B1B, 64) * SIREN_TURNS * SCL);
    else if (j < L::O_HEAD) v = (float)((double)bias_cd(pa.b[NL - 2], j - L::O_B2, 256) * SIREN_TURNS * SCL);
    else if (MODE == MODE_IMNET) v = (float)((double)bias_cd(pa.b[NL - 1], j - L::O_HEAD, 64) * SCL);
    else {
        const int jj = j - L::O_HEAD;
        if (jj < 3 * 32 * 8) {                                      // Wv[o][q][hf][r]: k = 8q + 4hf + r
            const int r = jj & 3, hfl = (jj >> 2) & 1, q = (jj >> 3) & 31, o = jj >> 8;
            v = pa.w[NL - 1][(long)o * 256 + 8 * q + 4 * hfl + r];
        } else {
            const int o = jj - 3 * 32 * 8;
            v = o < 3 ? pa.b[NL - 1][o] : 0.f;
        }
    }
    floats[j] = v;
}

template <int NP>
__global__ void siren_split_pack_kernel(SplitPackArgs pa, float* out, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (pa.mode == MODE_IMNET) siren_split_pack_elem<MODE_IMNET, NP>(pa, (unsigned short*)out, out + SLayout<MODE_IMNET, NP>::F_END * 4, i);
    else if (pa.mode == MODE_FLOW) siren_split_pack_elem<MODE_FLOW, NP>(pa, (unsigned short*)out, out + SLayout<MODE_FLOW, NP>::F_END * 4, i);
    else if (pa.mode == MODE_SYNTH) siren_split_pack_elem<MODE_SYNTH, NP>(pa, (unsigned short*)out, out + SLayout<MODE_SYNTH, NP>::F_END * 4, i);
    else siren_split_pack_elem<MODE_SYNTHC, NP>(pa, (unsigned short*)out, out + SLayout<MODE_SYNTHC, NP>::F_END * 4, i);
}

template <int NP>
static long siren_pack_split_np(int mode, const float* const* w, const float* const* b, float* packed, void* stream) {
    const long floats = mode == MODE_IMNET ? SLayout<MODE_IMNET, NP>::TOTAL_FLOATS
                      : mode == MODE_FLOW ? SLayout<MODE_FLOW, NP>::TOTAL_FLOATS
                      : mode == MODE_SYNTH ? SLayout<MODE_SYNTH, NP>::TOTAL_FLOATS : SLayout<MODE_SYNTHC, NP>::TOTAL_FLOATS;
    if (!packed) return floats;
    if (!w || !b) return MOTIF_EINVAL;
    const int nl = mode >= MODE_SYNTH ? 5 : 4;
    SplitPackArgs pa;
    for (int l = 0; l < 5; ++l) { pa.w[l] = l < nl ? w[l] : nullptr; pa.b[l] = l < nl ? b[l] : nullptr; }
    pa.mode = mode;
    const long fend = mode == MODE_IMNET ? SLayout<MODE_IMNET, NP>::F_END : mode == MODE_FLOW ? SLayout<MODE_FLOW, NP>::F_END
                    : mode == MODE_SYNTH ? SLayout<MODE_SYNTH, NP>::F_END : SLayout<MODE_SYNTHC, NP>::F_END;
    const long nfl = floats - fend * 4;
    const long total = fend * 8 + nfl;                              // one thread per 16-bit element, then per float
    siren_split_pack_kernel<NP><<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(pa, packed, total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(long)e - 1000;
    return floats;
}

// mode: network kind 0 .. 3, + 8 for the two-part fp16 form (forward calls: pre = 3 instead of 2)
extern "C" long motif_siren_pack_split(int mode, const float* const* w, const float* const* b, float* packed, void* stream) {
    const int kind = mode & 7, two = mode >> 3;
    if (mode < 0 || kind > 3 || two > 1) return MOTIF_EINVAL;
    return two ? siren_pack_split_np<2>(kind, w, b, packed, stream) : siren_pack_split_np<3>(kind, w, b, packed, stream);
}

template <int MODE, int TP, int NP>
static int launch_siren_split(const SirenArgs& a_in, void* stream) {
    using L = SLayout<MODE, NP>;
    static_assert(L::LDS_BYTES <= 160 * 1024, "resident part of the packed network must fit the 160 KB LDS");
    SirenArgs a = a_in;
    a.stagger = 4;
    if (const int sv = motif_opt(MOTIF_OPT_SIREN_STAGGER)) a.stagger = sv < 0 ? 0 : sv;
    hipError_t e = hipFuncSetAttribute((const void*)siren_split_kernel<MODE, TP, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long Q = (long)a.HH * a.WW;
    const long tiles = (long)a.NB * ((Q + 32 * TP - 1) / (32 * TP));
    long blocks = (tiles + SIREN_WAVES - 1) / SIREN_WAVES;
    if (blocks > cus) blocks = cus;
    siren_split_kernel<MODE, TP, NP><<<(int)blocks, SIREN_THREADS, (size_t)L::LDS_BYTES, (hipStream_t)stream>>>(a);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// parts: 3 = blob of three bf16 parts (pre = 2), 2 = blob of two fp16 parts (pre = 3)
int motif_siren_split_launch(int mode, const SirenArgs& a, void* stream, int parts) {
    if (parts == 2) {
        if (mode == MODE_IMNET) return launch_siren_split<MODE_IMNET, 1, 2>(a, stream);
        if (mode == MODE_FLOW) return launch_siren_split<MODE_FLOW, 1, 2>(a, stream);
        if (mode == MODE_SYNTHC) return launch_siren_split<MODE_SYNTHC, 1, 2>(a, stream);
        return launch_siren_split<MODE_SYNTH, 1, 2>(a, stream);
    }
    if (mode == MODE_IMNET) return launch_siren_split<MODE_IMNET, 1, 3>(a, stream);
    if (mode == MODE_FLOW) return launch_siren_split<MODE_FLOW, 1, 3>(a, stream);
    if (mode == MODE_SYNTHC) return launch_siren_split<MODE_SYNTHC, 1, 3>(a, stream);
    return launch_siren_split<MODE_SYNTH, 1, 3>(a, stream);
}

// 3x3 / stride 1 convolution on the bf16 matrix cores with fp32-equivalent arithmetic ("split" engine).
//
// gfx950 runs v_mfma_f32_32x32x2_f32 at the fp32 VECTOR rate (157 TFLOP/s); v_mfma_f32_32x32x16_bf16 is 16x
// faster.  An fp32 number is the exact sum of three bf16 numbers (8+8+8 mantissa bits):
//        x = xh + xm + xl,   xh = bf16(x), xm = bf16(x - xh), xl = bf16(x - xh - xm)      (all three exact)
// so   x*w = xh*wh + xh*wm + xm*wh + xm*wm + xh*wl + xl*wh  + O(2^-25 |x w|),
// six bf16 products accumulated in fp32 by the MFMA: 6/16 of the fp32-MFMA time for the same MACs.  The dropped
// terms are below the rounding error of an fp32 fma chain (tests/test_kernels_gpu.py measures both against fp64).
// NP selects the number of parts: 3 -> 6 products (fp32-equivalent), 2 -> 3 products (16-bit mantissa),
// 1 -> plain bf16.
//
// Data flow per block (WAVES waves, tile = 2*WAVES output rows x 32 columns x 64 couts):
//   * activations stay planar fp32 NCHW in HBM; a 16-channel chunk of the (rows+2) x 34 input patch is loaded
//     into registers (issue-early), split on the VALU (hidden under the MFMAs of the other resident wave) and
//     written to LDS as [part][channel octet][py][px] x 8 bf16, so one ds_read_b128 is a B fragment;
//   * weights are split once at pack time into MFMA A fragments [k-step][part][cout tile][lane] x 8 bf16 and
//     read straight from global memory (1 KB contiguous per fragment, L1/L2 resident: every wave of every block
//     walks the same 6 KB per k-step), one k-step ahead in registers;
//   * k-step = (16-channel group, tap); lower half-wave = channels 0..7 of the group, upper = 8..15.
// The epilogue (bias, residual, activation, planar stores) is the one of the fp32 engine: same C/D layout.
#include "conv_split_common.h"
#include <stdlib.h>
#include <type_traits>

#ifdef MOTIF_TRACE
__device__ long long g_trace[4096 * 16];
#define TRACE(slot) do { if (lane == 0 && blockIdx.y == 0) { const int b_ = blockIdx.x + gridDim.x * blockIdx.z; if (b_ < 1024) g_trace[(b_ * 4 + wave) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } } while (0)
extern "C" int motif_debug_trace(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), sizeof(long long) * n); }
#else
#define TRACE(slot)
#endif

// Tried and rejected: a thread-level-parallel variant (three blocks per CU at <= 168 VGPRs, one LDS buffer, weight
// fragments just in time) -- 15-20 % slower than hiding the latencies inside the wave as below.
template <int NP, int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_split_kernel(ConvArgs a) {
    constexpr int RP = 2, NC = 2, NT = 64 * WAVES, TH = RP * WAVES, PH = TH + 2, PW = 34, PHW = PH * PW;
    constexpr int NI = (2 * PHW + NT - 1) / NT;          // (pixel, channel octet) items a thread stages per chunk
    using PR = SplitProducts<NP>;
    constexpr int SLOTS = 2 * PHW + 4;                   // per (buffer, part): [2 octets][PHW] 16-byte slots + dummy
    extern __shared__ __attribute__((aligned(16))) u32x4 lds_raw[];
    float* bias_s = (float*)lds_raw;                     // [64] bias of this cout group
    u32x4* lds = lds_raw + 16;                           // [2 buffers][NP][SLOTS]; reused by the epilogue transpose
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    // XCD-aware tile order: workgroups go round-robin over the 8 XCDs (each with its own L2) in linear-id order, so
    // the blocks that land on one XCD get a contiguous run of tiles -- vertically adjacent tiles share their halo rows
    // in that XCD's L2 instead of fetching them once per XCD.
    const int tile_id = (a.dbg & 64) ? (int)blockIdx.x : xcd_tile_id();
    const int tx = tile_id % a.tiles_x, ty = tile_id / a.tiles_x;
    const int g = blockIdx.y / a.ncg, cg = blockIdx.y % a.ncg;
    const int pz = blockIdx.z / a.N, n = blockIdx.z - pz * a.N;
    const float* a_in0 = a.in0[pz]; const float* a_in1 = a.in1[pz];
    const float* a_bias = a.bias[pz]; const float* a_res = a.res[pz]; float* a_out = a.out[pz];
    const long a_res_bs = a.res_bs[pz], a_out_bs = a.out_bs[pz];
    const int HW = a.H * a.W;

    TRACE(0);
    // chunk-invariant staging plan
    const int iy0 = ty * TH - a.pad, ix0 = tx * 32 - a.pad;
    int eoff[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int e = tid + NT * j;
        eoff[j] = -1;
        if (e < 2 * PHW) {
            const int p = e >= PHW ? e - PHW : e;
            const int py = p / PW, px = p - py * PW;
            int iy = iy0 + py, ix = ix0 + px;
            if (a.pad_mode == 1) {
                if (iy < 0) iy = -iy; else if (iy >= a.H) iy = 2 * (a.H - 1) - iy;
                if (ix < 0) ix = -ix; else if (ix >= a.W) ix = 2 * (a.W - 1) - ix;
            }
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) eoff[j] = iy * a.W + ix + (e >= PHW ? 8 * HW : 0);
        }
    }
    TRACE(12);
    float bias_v = 0.f;                                      // requested now, parked in LDS before the first barrier
    if (tid < 64 && a_bias && cg * 64 + tid < a.Cout_g) bias_v = a_bias[g * a.Cout_g + cg * 64 + tid];

    f32x16 acc[NC][RP];
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int j = 0; j < RP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* in0n = a_in0 + (long)n * a.in0_bs[pz];
    const float* in1n = a_in1 ? a_in1 + (long)n * a.in1_bs[pz] : nullptr;
    const int nks = a.Kpad;                                  // k-steps = 9 * ceil(Cin_g / 16)
    const u32x4* wbase = (const u32x4*)a.wp[pz] + (long)(g * a.ncg + cg) * nks * (NP * NC * 64) + lane;

    float pre[NI][8];
    // item j of a thread = 8 channels of one patch pixel: 8 coalesced dword loads, later 3 x 16 B to LDS
    auto issue_item = [&](int j, int c0) {                   // global -> registers
        const int gch0 = g * a.Cin_g + c0;
        const float* base = (gch0 < a.C0) ? in0n + (long)gch0 * HW : in1n + (long)(gch0 - a.C0) * HW;
        const int cvalid = a.Cin_g - c0 - ((tid + NT * j >= PHW) ? 8 : 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool ok = eoff[j] >= 0 && q < cvalid;               // branch-free: padding reads base[0] and is
            const unsigned idx = ok ? (unsigned)(eoff[j] + q * HW) : 0u;   // zeroed at commit time; uniform base + 32-bit
            pre[j][q] = base[idx];                                      // per-lane offset (one address VGPR per load)
        }
    };
    auto commit_item = [&](int j, int buf, int c0) {         // mask, split, registers -> LDS (branch-free)
        const int e = tid + NT * j;
        const int cvalid = a.Cin_g - c0 - (e >= PHW ? 8 : 0);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (!(eoff[j] >= 0 && q < cvalid)) pre[j][q] = 0.f;
        u32x4 parts[NP];
        split8<NP>(pre[j], parts);
        const int ew = e < 2 * PHW ? e : 2 * PHW;             // surplus threads write the dummy slot
#pragma unroll
        for (int p = 0; p < NP; ++p) lds[(buf * NP + p) * SLOTS + ew] = parts[p];
    };
    u32x4 wf[2][NP][NC];
    auto loadw = [&](int ks, u32x4 (&dst)[NP][NC]) {
        const u32x4* src = wbase + (long)ks * (NP * NC * 64);
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < NC; ++i) dst[p][i] = src[(p * NC + i) * 64];
    };

    loadw(0, wf[0]);
    TRACE(13);
#pragma unroll
    for (int j = 0; j < NI; ++j) issue_item(j, 0);
    TRACE(1);
#pragma unroll
    for (int j = 0; j < NI; ++j) commit_item(j, 0, 0);
    if (tid < 64) bias_s[tid] = bias_v;
    TRACE(2);
    __syncthreads();
    TRACE(3);
    const int nch = nks / 9;
    int cur = 0;
    // One 16-channel chunk = 9 taps.  The staging of the NEXT chunk is spread over the taps (loads of item t at tap
    // t, split + LDS write of item j two taps apart at the end) so that a wave's instruction stream is a uniform
    // mix of MFMA, VALU and memory operations: the two waves sharing a SIMD never fall into lock-step phases.
    auto chunk_body = [&](int chunk, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        const u32x4* pb = lds + cur * NP * SLOTS + half * PHW + (RP * wave) * PW + l31;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t < 8 || MORE) loadw(chunk * 9 + t + 1, wf[(t + 1) & 1]);
            if (MORE && t < NI) issue_item(t, (chunk + 1) * 16);
            __builtin_amdgcn_sched_barrier(0);
            {
                u32x4 bf[NP][RP];
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int j = 0; j < RP; ++j) bf[p][j] = pb[p * SLOTS + (j + t / 3) * PW + (t % 3)];
#pragma unroll
                for (int k = 0; k < PR::n; ++k)
#pragma unroll
                    for (int i = 0; i < NC; ++i)
#pragma unroll
                        for (int j = 0; j < RP; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, wf[t & 1][PR::w[k]][i]), __builtin_bit_cast(bf16x8, bf[PR::x[k]][j]),
                                acc[i][j], 0, 0, 0);
            }
            if (MORE) {
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    if (t == 8 - 2 * (NI - 1 - j)) commit_item(j, cur ^ 1, (chunk + 1) * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        TRACE(4 + 2 * chunk);
        __syncthreads();
        TRACE(5 + 2 * chunk);
        cur ^= 1;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < NC; ++i) wf[0][p][i] = wf[1][p][i];
    };
    for (int chunk = 0; chunk + 1 < nch; ++chunk) chunk_body(chunk, std::true_type{});
    chunk_body(nch - 1, std::false_type{});
    // 16-byte stores through an LDS transpose when the layout allows it (block-uniform test), else per-lane stores
    const bool vec_ok = !(a.dbg & 4) && a.Cout_g - cg * 64 >= 64 && (a.Wo & 3) == 0 && (a_out_bs & 3) == 0 &&
                        (((unsigned long long)a_out) & 15) == 0 &&
                        (!a.res_mode || ((a_res_bs & 3) == 0 && (((unsigned long long)a_res) & 15) == 0)) && !(a.dbg & 32);
    if (vec_ok)
        conv_epilogue_lds<NC, RP, WAVES>(a, acc, bias_s, (float*)lds, n, g, cg, ty, tx, a_res, a_res_bs, a_out, a_out_bs);
    else
        conv_epilogue<NC, RP, false>(a, acc, bias_s, n, g, cg, ty * TH + RP * wave, tx * 32 + l31, half, a_res, a_res_bs, a_out, a_out_bs);
    TRACE(14);
    __builtin_amdgcn_s_waitcnt(0);
    TRACE(15);
}

// weight [Cout, Cin_g, 3, 3] fp32 -> A fragments [group][cout group of 64][k-step][part][cout tile][lane][8] bf16,
// k-step = (channel group of 16, tap); lane = (cout & 31) + 32 * (channel octet); zero padded
__global__ void conv_split_pack_kernel(const float* w, unsigned short* wp, int Cout_g, int Cin_g, int nks, int ncg, int NP, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), ct = (int)((i >> 9) & 1);
    long tt = i >> 10;
    const int part = (int)(tt % NP); tt /= NP;
    const int ks = (int)(tt % nks); tt /= nks;
    const int cgi = (int)(tt % ncg);
    const int g = (int)(tt / ncg);
    const int col = cgi * 64 + ct * 32 + (lane & 31);
    const int c = (ks / 9) * 16 + 8 * (lane >> 5) + e, t = ks % 9;
    float v = 0.f;
    if (col < Cout_g && c < Cin_g) v = w[((long)(g * Cout_g + col) * Cin_g + c) * 9 + t];
    unsigned short out = 0;
    for (int p = 0; p <= part; ++p) {
        const unsigned pk = pk_bf16(v, 0.f);
        out = (unsigned short)(pk & 0xffffu);
        v -= bf_lo(pk);
    }
    wp[i] = out;
}

// ---- host side (called from conv_igemm.hip's entry points) -------------------------------------------------------

// The engine takes a layer when it is 3x3, stride 1, dilation 1 with > 32 couts and >= 16 input channels per group
// (depends on the weight shape and hyper-parameters only, so pack and forward always agree).
bool motif_conv_split_eligible(const MotifConvDesc* d) {
    if (!d || split_parts(d->mma) == 0) return false;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->dil != 1 || d->groups < 1) return false;
    const int Cin = d->C0 + d->C1;
    if (Cin % d->groups || d->Cout % d->groups) return false;
    // > 32 couts per group fill both cout tiles of a workgroup; 17 .. 32 fill one (half the matrix work is spent on zero weights), which
    // still beats the fp32 engine when the reduction is long (PWC-Net's 497 .. 629 -> 32 layers: 190-240 us there)
    const int Cin_g = Cin / d->groups, Cout_g = d->Cout / d->groups;
    return Cin_g >= 16 && (Cout_g > 32 || (Cout_g >= 17 && Cin_g >= 128));
}

long motif_conv_split_packed_floats_direct(const MotifConvDesc* d) {
    const int Cin_g = (d->C0 + d->C1) / d->groups, Cout_g = d->Cout / d->groups;
    const long nks = 9L * ((Cin_g + 15) / 16), ncg = (Cout_g + 63) / 64;
    return (long)d->groups * ncg * nks * split_parts(d->mma) * 2 * 64 * 4;     // 8 bf16 = 4 floats per lane
}

// blob = [direct fragments | Winograd F(2,3) fragments (mma = 6 only, conv_wino.hip)]: the kernel is chosen per launch
long motif_conv_split_packed_floats(const MotifConvDesc* d) {
    return motif_conv_split_packed_floats_direct(d) + motif_conv_wino_packed_floats(d);
}

int motif_conv_split_pack(const MotifConvDesc* d, const float* weight, float* packed, hipStream_t s) {
    const int Cin_g = (d->C0 + d->C1) / d->groups, Cout_g = d->Cout / d->groups;
    const int nks = 9 * ((Cin_g + 15) / 16), ncg = (Cout_g + 63) / 64, NP = split_parts(d->mma);
    const long total = (long)d->groups * ncg * nks * NP * 2 * 64 * 8;
    conv_split_pack_kernel<<<cdiv(total, 256), 256, 0, s>>>(weight, (unsigned short*)packed, Cout_g, Cin_g, nks, ncg, NP, total);
    MOTIF_LAUNCH_CHECK();
    if (motif_conv_wino_packed_floats(d) > 0) return motif_conv_wino_pack(d, weight, packed + motif_conv_split_packed_floats_direct(d), s);
    return MOTIF_OK;
}

int motif_conv_split_launch(const MotifConvDesc* d, ConvArgs& a, int P, hipStream_t s) {
    {                                                    // round 4: conv_wino.hip (conv_engine 5 = wherever it applies, 6 = never)
        const int force = motif_opt(MOTIF_OPT_CONV_ENGINE);
        // by measurement inside the clip (gpurun_out/r4/shapes_*.txt): the three-part form is faster on every 3x3 layer except those
        // with a transcendental epilogue (the offset | sigmoid(mask) layers of the DCNs: a lone wave hides none of the exp / rcp chains
        // of its exposed epilogue); the two-part form (mma = 7) is faster there too (507 vs 723 us on 8 x 64 -> 216 x 180 x 320)
        const bool plain_act = d->mma == 7 || (d->act_split <= 0 && (d->act == MOTIF_ACT_NONE || d->act == MOTIF_ACT_RELU || d->act == MOTIF_ACT_LRELU));
        if ((force == 5 || (force == 0 && plain_act)) && motif_conv_wino_eligible(d, a, P)) return motif_conv_wino_launch(d, a, P, s);
    }
    if (motif_opt(MOTIF_OPT_CONV_ENGINE) != 1 && motif_conv_split2_eligible(d, a, P)) return motif_conv_split2_launch(d, a, P, s);   // round 3: conv_split2.hip
    const int Cin_g = (d->C0 + d->C1) / d->groups, Cout_g = d->Cout / d->groups;
    if (d->C1 > 0 && (d->groups != 1 || d->C0 % 16)) return MOTIF_ELIMIT;   // a 16-channel chunk must not straddle the sources
    if ((long)d->H * d->W * 16 >= 0x7fffffffL) return MOTIF_ELIMIT;
    const int Ho = d->H + 2 * d->pad - 2, Wo = d->W + 2 * d->pad - 2;
    if (Ho <= 0 || Wo <= 0) return MOTIF_EINVAL;
    const int NP = split_parts(d->mma);
    a.Ho = Ho; a.Wo = Wo; a.Cin_g = Cin_g; a.Cout_g = Cout_g;
    a.Kpad = 9 * ((Cin_g + 15) / 16);
    a.ncg = (Cout_g + 63) / 64;
    a.tiles_x = (Wo + 31) / 32;
    const int waves = 4;           // 2-wave (4 blocks per CU) and 8-wave (one block per CU) tiles were measured slower
    const int TH = 2 * waves, tiles_y = (Ho + TH - 1) / TH;
    size_t ldsb = (size_t)2 * NP * (2 * (TH + 2) * 34 + 4) * 16;
    const size_t scratch = (size_t)32 * (TH * 32 + 8) * 4;                      // epilogue transpose of one 32-cout tile
    ldsb = (ldsb > scratch ? ldsb : scratch) + 64 * 4;
    if (const int pad = motif_opt(MOTIF_OPT_LDS_PAD); pad > 0) ldsb = (ldsb + pad - 1) / pad * pad;
    dim3 grid(a.tiles_x * tiles_y, d->groups * a.ncg, d->N * P);
#define MOTIF_LAUNCH_SPLIT(NPV, WV)                                                                                     \
    do {                                                                                                                \
        if (ldsb > 64 * 1024)                                                                                           \
            (void)hipFuncSetAttribute((const void*)conv_split_kernel<NPV, WV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb); \
        conv_split_kernel<NPV, WV><<<grid, 64 * WV, ldsb, s>>>(a);                                                     \
    } while (0)
    if (NP == 3) MOTIF_LAUNCH_SPLIT(3, 4); else if (NP == 2) MOTIF_LAUNCH_SPLIT(2, 4); else MOTIF_LAUNCH_SPLIT(1, 4);
#undef MOTIF_LAUNCH_SPLIT
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

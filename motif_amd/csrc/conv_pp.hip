// 3x3 / stride 1 convolution on the bf16 matrix cores, fp32-equivalent 3-way split arithmetic (see conv_split.hip), as a
// persistent PING-PONG kernel: round-3 replacement of conv_split_kernel on the launches that carry the time.
//
// Why.  conv_split_kernel runs two independent 4-wave blocks per CU.  Its per-wave timeline showed that the
// two blocks of a CU start together and stay in lock step: both stage, both multiply, both store at the same time, so a
// block's prologue / staging / epilogue (32 k of its 80 k cycles) are almost never covered by the other block's MFMAs.
// Here the complementarity is built in.  One 8-wave workgroup per CU = two HALVES of four waves (waves w and w+4 share a
// SIMD).  A half alternates between two kinds of phases, and the halves are offset by one phase with workgroup barriers:
//      phase p   : half A  COMPUTE chunk k   (pure ds_read / weight loads / MFMA: the matrix pipe is its alone)
//                  half B  OTHER             (epilogue of a finished tile, global loads + 3-way split + LDS write of its
//                                             next 16-channel chunk, first weight fragments of its next compute phase)
//      phase p+1 : roles swapped.
// Each half owns ITS OWN sequence of output tiles (4*RP rows x 32 columns x 64 couts, a wave = RP rows), its own single
// staging buffer (stage(k) -> barrier -> compute(k) -> barrier -> stage(k+1): no double buffer needed) and wave-private
// epilogue scratch, so the halves share nothing but the barriers and the SIMDs.  The kernel is persistent over tiles
// (tile i of half h of block b = i*2G + h*G + b'), so prologue and epilogue of every tile but the first / last of a block are
// hidden as well.  Activations reach LDS by LDS-DMA (16 bytes per lane into a wave-private fp32 landing area, split from there):
// the dword staging loads of conv_split_kernel were 3.8 k of a phase's 8 k cycles on the CU's vector-memory path.
// Weights: the packed A fragments of conv_split.hip, unchanged ([k-step][part][cout tile][lane] x 8 bf16).
#include "conv_split_common.h"

#ifdef MOTIF_TRACE
__device__ long long g_pp_trace[1024 * 8 * 32];
#define PPTRACE_AT(slot) do { if (lane == 0 && blockIdx.x < 1024) g_pp_trace[(blockIdx.x * 8 + wave) * 32 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define PPTRACE(slot) do { if ((slot) < 21) PPTRACE_AT(slot); } while (0)
extern "C" int motif_debug_pp_trace(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pp_trace), sizeof(long long) * n); }
#else
#define PPTRACE(slot)
#define PPTRACE_AT(slot)
#endif

namespace {
// XCD-aware block order (1-D grid): workgroups are dealt round-robin over the 8 XCDs, so XCD x gets a contiguous run of b'
// (neighbouring tiles -- shared halo rows, the cout groups of one spatial tile -- meet in one L2).  Bijection for every G.
__device__ __forceinline__ int xcd_block_id(int b, int G) {
    const int q = G >> 3, r = G & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// Products ordered by ACTIVATION part, smallest part first (w = weight part, x = activation part): when the last product of
// a part has been issued, its B-fragment registers take the same part of the next tap.
template <int NP> struct PPOrder;
template <> struct PPOrder<2> { static constexpr int n = 3; static constexpr int w[3] = {0, 1, 0}; static constexpr int x[3] = {1, 0, 0}; };
template <> struct PPOrder<3> {
    static constexpr int n = 6;
    static constexpr int w[6] = {0, 1, 0, 2, 1, 0};
    static constexpr int x[6] = {2, 1, 1, 0, 0, 0};
};

// Static schedule of one tap of the compute phase: after MFMA m (m = product * RW + row) at most ONE operand request is
// issued, so that the wave -- alone on the matrix pipe of its SIMD -- never spends more than one memory-instruction issue
// between two MFMAs.
//   bpart/brow/bnext[m]: B fragment (activation part, row) requested after MFMA m, for the next tap (bnext) or for this one;
//   widx[m]: weight fragment (part) of a later tap requested after MFMA m.
template <int NP, int RW>
struct PPSched {
    static constexpr int M = PPOrder<NP>::n * RW;
    int bpart[M], brow[M], bnext[M], widx[M];
    constexpr PPSched() : bpart(), brow(), bnext(), widx() {
        for (int m = 0; m < M; ++m) { bpart[m] = -1; brow[m] = 0; bnext[m] = 0; widx[m] = -1; }
        for (int xp = NP - 1; xp >= 0; --xp) {
            int last = 0;
            for (int k = 0; k < PPOrder<NP>::n; ++k) if (PPOrder<NP>::x[k] == xp) last = k;
            const int gend = (last + 1) * RW - 1;              // last MFMA that reads part xp
            for (int j = 0; j < RW; ++j) {
                int mm = gend + j, nx = 1;
                if (mm >= M) { mm -= M; nx = 0; }              // part 0 wraps into the first MFMAs of the tap it is for
                bpart[mm] = xp; brow[mm] = j; bnext[mm] = nx;
            }
        }
        int wi = 0;
        for (int m = 0; m < M && wi < NP; ++m) if (bpart[m] < 0) widx[m] = wi++;
    }
};

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
// [8 couts][RW*32 pixels] floats in four passes and leave as 16-byte row pieces: bias, residual (one 16-byte load, requested a
// pass ahead), activation, one 16-byte store.  No workgroup barrier (the other half is in its MFMA phase meanwhile).
// Activation and residual mode are wave-uniform run-time switches.  `cbase` = first cout of this wave's 32 in the tensor,
// `climit` = valid couts from there (partial last group).  Host guarantees: Wo % 4 == 0, 16-byte aligned tensors,
// 32 * Ho * Wo < 2^31, act_split on an 8-cout boundary.
template <int RW, bool RES>
__device__ __forceinline__ void conv_epilogue_wave(const ConvArgs& a, f32x16 (&acc)[RW], const float* bias_w, float* sc, int lane,
                                                   int cbase, int climit, int oy0, int ox0, const float* rb, float* ob) {
    constexpr int S = RW * 32, NIT = RW;                              // 8 couts x RW rows x 8 quads = 64 * RW items per pass
    const int half = lane >> 5, l31 = lane & 31;
    const unsigned HWo = (unsigned)(a.Ho * a.Wo);
    const int rm = a.res_mode;
    unsigned loff[NIT]; int scoff[NIT], coi[NIT]; bool ok[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = lane + 64 * it;
        const int co = idx / (RW * 8), q = idx - co * (RW * 8);
        const int row = q >> 3, col = (q & 7) * 4;
        const int oy = oy0 + row, ox = ox0 + col;
        ok[it] = oy < a.Ho && ox < a.Wo;
        loff[it] = ok[it] ? (unsigned)co * HWo + (unsigned)(oy * a.Wo + ox) : 0u;    // masked lanes read element 0, store nothing
        scoff[it] = co * S + row * 32 + col;
        coi[it] = co;
    }
    f32x4 rv[2][NIT];
    auto load_res = [&](int pass, f32x4 (&dst)[NIT]) {
        const float* base = rb + (long)(8 * pass) * HWo;
#pragma unroll
        for (int it = 0; it < NIT; ++it) dst[it] = *(const f32x4*)(base + (8 * pass + coi[it] < climit ? loff[it] : 0u));
    };
    if constexpr (RES) load_res(0, rv[0]);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        if constexpr (RES) { if (pass + 1 < 4) load_res(pass + 1, rv[(pass + 1) & 1]); }
        const int ac = (a.act_split > 0 && cbase + 8 * pass >= a.act_split) ? a.act2 : a.act;     // uniform per pass
#pragma unroll
        for (int j = 0; j < RW; ++j)
#pragma unroll
            for (int r3 = 0; r3 < 4; ++r3) sc[(r3 + 4 * half) * S + j * 32 + l31] = acc[j][4 * pass + r3];
        f32x4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            v[it] = *(const f32x4*)(sc + scoff[it]);
            const float b = bias_w[8 * pass + coi[it]];
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
        float* obp = ob + (long)(8 * pass) * HWo;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (ok[it] && 8 * pass + coi[it] < climit) *(f32x4*)(obp + loff[it]) = v[it];
    }
}
}  // namespace

// One workgroup = 8 waves = two halves; a half's tile = 8 rows x 32 columns x 64 couts, wave (ct, rg) of a half = cout tile ct
// (32 couts) x rows 4*rg .. 4*rg+3: the four waves of a half read only TWO distinct sets of weight fragments (half the L1
// traffic of a row split -- the vector-memory path is the resource the two halves compete for).
template <int NP>
__global__ __launch_bounds__(512) void conv_pp_kernel(ConvArgs a, int ntiles, int tiles_y) {
    constexpr int RW = 4, TH = 8, PH = TH + 2, PW = 34, PHW = PH * PW;
    constexpr int SLOTS = 2 * PHW + 4;                   // per part: [2 octets][PHW] 16-byte slots
    constexpr int STG = NP * SLOTS;                      // bf16 staging buffer of one half (u32x4)
    constexpr int SCR = 8 * RW * 32 / 4;                 // epilogue scratch of one wave (u32x4)
    constexpr int QW = 10, UW = 2 * PH / 4;              // staged row = 10 aligned pixel quads; (octet, row) units per wave
    constexpr int NQ = UW * 8 * QW, NDMA = (NQ + 63) / 64, F32W = NDMA * 64;   // fp32 landing area of one wave (u32x4)
    constexpr int NITEM = UW * PW, NSPL = (NITEM + 63) / 64;                   // (unit, pixel) items a wave splits per chunk
    extern __shared__ __attribute__((aligned(16))) u32x4 lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = wave >> 2, w4 = wave & 3, ct = w4 & 1, rg = w4 >> 1;
    float* bias_w = (float*)lds_raw + wave * 64;                       // [8 waves][64] (32 used)
    float* scr = (float*)(lds_raw + 8 * 16 + wave * SCR);             // [8 waves][8][RW*32]
    u32x4* stg = lds_raw + 8 * 16 + 8 * SCR + h * STG;                // [2 halves][NP][SLOTS]
    u32x4* f32w = lds_raw + 8 * 16 + 8 * SCR + 2 * STG + wave * F32W; // [8 waves][F32W]: fp32 landing area (LDS-DMA)

    const int G = gridDim.x, bq = xcd_block_id(blockIdx.x, G);
    const int t_first = h * G + bq, t_step = 2 * G;
    const int nt_h = t_first < ntiles ? (ntiles - t_first - 1) / t_step + 1 : 0;
    const int nt_0 = bq < ntiles ? (ntiles - bq - 1) / t_step + 1 : 0;  // half 0 never has fewer tiles than half 1
    const int nch = a.Kpad / 9;
    const int Kh = nt_h * nch, Kmax = nt_0 * nch;
    const int ncgG = a.CK;                                              // groups * ncg (CK is otherwise unused by this kernel)
    const int HW = a.H * a.W;

    // ---- per-tile state ---------------------------------------------------------------------------------------------
    int st_tile = 0, done_tile = 0;                      // tile being staged / multiplied, tile whose epilogue is due
    int doff[NDMA];                                      // per-lane source offset of each DMA request (-1: outside the image)
    unsigned ivalid = 0;                                 // bit it: split item it of this lane lies inside the image
    const float* in0n = nullptr; const float* in1n = nullptr;
    const u32x4* wbase = nullptr;
    int st_g = 0;
    float bias_v = 0.f;
    auto decode = [&](int tv, int& n, int& pz, int& g, int& cg, int& ty, int& tx) {
        const int t = __builtin_amdgcn_readfirstlane(tv);              // wave-uniform: the per-problem pointers come by scalar loads
        const int cgg = t % ncgG; int s = t / ncgG;
        tx = s % a.tiles_x; s /= a.tiles_x;
        ty = s % tiles_y; const int z = s / tiles_y;
        g = cgg / a.ncg; cg = cgg - g * a.ncg;
        pz = z / a.N; n = z - pz * a.N;
    };
    auto setup_tile = [&](int t) {                       // staging plan of tile t (chunk-invariant)
        int n, pz, g, cg, ty, tx;
        decode(t, n, pz, g, cg, ty, tx);
        st_g = g;
        in0n = a.in0[pz] + (long)n * a.in0_bs[pz];
        in1n = a.in1[pz] ? a.in1[pz] + (long)n * a.in1_bs[pz] : nullptr;
        wbase = (const u32x4*)a.wp[pz] + (long)(g * a.ncg + cg) * a.Kpad * (NP * 2 * 64) + ct * 64 + lane;
        const float* bp = a.bias[pz];
        bias_v = (bp && lane < 32 && cg * 64 + ct * 32 + lane < a.Cout_g) ? bp[g * a.Cout_g + cg * 64 + ct * 32 + lane] : 0.f;
        const int iy0 = ty * TH - 1, x0 = tx * 32 - 4;   // pad = 1 (host); the staged rows start 4 pixels left of the tile: aligned quads
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int qd = i * 64 + lane;
            const int ul = qd / (8 * QW), r = qd - ul * (8 * QW), ch = r / QW, xq = r - ch * QW;
            const int U = w4 * UW + ul, o = U / PH, py = U - o * PH;
            const int iy = iy0 + py, x = x0 + 4 * xq;
            doff[i] = (qd < NQ && iy >= 0 && iy < a.H && x >= 0 && x < a.W) ? iy * a.W + x + (8 * o + ch) * HW : -1;
        }
        ivalid = 0;
#pragma unroll
        for (int it = 0; it < NSPL; ++it) {
            const int id = lane + 64 * it;
            const int ul = id / PW, p = id - ul * PW;
            const int U = w4 * UW + ul, py = U % PH;
            const int iy = iy0 + py, ix = tx * 32 - 1 + p;
            if (id < NITEM && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ivalid |= 1u << it;
        }
    };
    // global -> LDS, 16 bytes per lane, no registers: the wave's (octet, row) units of the 16-channel chunk at c0
    auto issue_dma = [&](int c0) {
        const int gch0 = st_g * a.Cin_g + c0;
        const float* base = (gch0 < a.C0) ? in0n + (long)gch0 * HW : in1n + (long)(gch0 - a.C0) * HW;
        const int crem = a.Cin_g - c0;                   // valid channels from c0 on
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int qd = i * 64 + lane;
            const int ul = qd / (8 * QW), r = qd - ul * (8 * QW), ch = r / QW;
            const int o = (w4 * UW + ul) / PH;
            const bool okc = doff[i] >= 0 && 8 * o + ch < crem;       // padding lanes fetch element 0 of the chunk (masked at the split)
            const float* src = base + (okc ? doff[i] : 0);
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) unsigned*>(reinterpret_cast<uintptr_t>(src)),
                                             reinterpret_cast<__attribute__((address_space(3))) unsigned*>(reinterpret_cast<uintptr_t>(f32w + i * 64)), 16, 0, 0);
        }
    };
    // landed fp32 -> 3 bf16 parts -> staging buffer ([part][octet][py][px] x 8 channels)
    auto split_items = [&](int c0) {
        const int crem = a.Cin_g - c0;
        const float* fw = (const float*)f32w;
#pragma unroll
        for (int it = 0; it < NSPL; ++it) {
            const int id = lane + 64 * it;
            if (NSPL * 64 > NITEM && id >= NITEM) continue;
            const int ul = id / PW, p = id - ul * PW;
            const int U = w4 * UW + ul, o = U / PH, py = U - o * PH;
            const bool inside = (ivalid >> it) & 1;
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float x = fw[(ul * 8 + q) * (4 * QW) + p + 3];
                v[q] = (inside && 8 * o + q < crem) ? x : 0.f;
            }
            u32x4 parts[NP];
            split8<NP>(v, parts);
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) stg[pp * SLOTS + o * PHW + py * PW + p] = parts[pp];
        }
    };

    f32x16 acc[RW];
    constexpr int WB = 3;                                // weight fragments two taps ahead
    u32x4 wf[WB][NP];
    auto loadw = [&](int ks, u32x4 (&dst)[NP]) {
        const u32x4* src = wbase + (long)ks * (NP * 2 * 64);
#pragma unroll
        for (int p = 0; p < NP; ++p) dst[p] = src[p * 2 * 64];
    };
    auto compute = [&](int c) {                          // 9 taps of chunk c: MFMAs and their operand requests, nothing else
        using PO = PPOrder<NP>;
        constexpr PPSched<NP, RW> SCH{};
        constexpr int M = PPSched<NP, RW>::M;
        const u32x4* pb = stg + half * PHW + (RW * rg) * PW + l31;
        u32x4 bfr[NP][RW];
        auto loadb = [&](int t, int p, int j) { bfr[p][j] = pb[p * SLOTS + (j + t / 3) * PW + (t % 3)]; };
#ifndef PP_PRIO
#define PP_PRIO 0
#endif
        __builtin_amdgcn_s_setprio(PP_PRIO);
#pragma unroll
        for (int p = NP - 1; p >= 0; --p)
#pragma unroll
            for (int j = 0; j < RW; ++j) loadb(0, p, j);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const int k = m / RW, j = m % RW;
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[t % WB][PO::w[k]]),
                                                                 __builtin_bit_cast(bf16x8, bfr[PO::x[k]][j]), acc[j], 0, 0, 0);
#ifndef PP_ABL_NOB
                if (SCH.bpart[m] >= 0) {
                    if (SCH.bnext[m]) { if (t < 8) loadb(t + 1, SCH.bpart[m], SCH.brow[m]); }
                    else if (t > 0) loadb(t, SCH.bpart[m], SCH.brow[m]);
                }
#endif
#ifndef PP_ABL_NOW
                if (SCH.widx[m] >= 0 && t + WB - 1 <= 8)
                    wf[(t + WB - 1) % WB][SCH.widx[m]] = (wbase + (long)(c * 9 + t + WB - 1) * (NP * 2 * 64))[SCH.widx[m] * 2 * 64];
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto epilogue = [&](int t) {
        int n, pz, g, cg, ty, tx;
        decode(t, n, pz, g, cg, ty, tx);
        const int cbase = g * a.Cout_g + cg * 64 + ct * 32, climit = a.Cout_g - cg * 64 - ct * 32;
        if (climit <= 0) return;                         // the upper cout tile of a partial group has nothing to store
        const long HWo = (long)a.Ho * a.Wo;
        float* ob = a.out[pz] + (long)n * a.out_bs[pz] + (long)cbase * HWo;
        const float* rb = a.res_mode ? a.res[pz] + (long)n * a.res_bs[pz] + (long)cbase * HWo : nullptr;
        // The epilogue sits inside the phase loop: an opaque copy of the lane id keeps hipcc from hoisting its per-lane address
        // arithmetic out of the loop (where it would be spilled).
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        if (a.res_mode) conv_epilogue_wave<RW, true>(a, acc, bias_w, scr, lane_e, cbase, climit, ty * TH + RW * rg, tx * 32, rb, ob);
        else conv_epilogue_wave<RW, false>(a, acc, bias_w, scr, lane_e, cbase, climit, ty * TH + RW * rg, tx * 32, rb, ob);
    };

    // ---- the phase loop ---------------------------------------------------------------------------------------------
    int c = 0, ti = 0, slot = 0;
    PPTRACE(slot); ++slot;
    if (h == 1) __syncthreads();                         // half 1 runs one phase behind half 0
    for (int k = 0; k <= Kmax; ++k) {
        if (k <= Kh) {                                   // OTHER phase (the other half multiplies meanwhile)
#ifndef PP_PRIO_OTHER
#define PP_PRIO_OTHER 3
#endif
            // A wave whose next instruction is an MFMA waiting for the matrix pipe keeps winning the VALU arbitration of its SIMD:
            // at equal priority this phase got ~one issue slot per MFMA of the other half (measured: 30 cycles per instruction).
            // The MFMA stream needs one slot in eight, so THIS phase takes the priority.
            __builtin_amdgcn_s_setprio(PP_PRIO_OTHER);
            if (k == nch) PPTRACE_AT(26);                // first tile boundary
            if (c == 0) {
                // hipcc waits vmcnt(0) for every ordinary load while an LDS-DMA request is outstanding, so the epilogue (with its
                // residual loads a pass ahead) runs BEFORE the next tile's first DMA requests, not under them
                if (k < Kh) { st_tile = t_first + ti * t_step; ++ti; setup_tile(st_tile); }   // plan + bias request of the next tile
                if (k > 0) epilogue(done_tile);
                if (k == nch) PPTRACE_AT(27);
#pragma unroll
                for (int j = 0; j < RW; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
                if (k < Kh) {
                    if (lane < 32) bias_w[lane] = bias_v;
                    done_tile = st_tile;
                }
                if (k == nch) PPTRACE_AT(28);
            }
            if (k < Kh) {
                loadw(c * 9, wf[0]);                     // first weight fragments of the compute phase (arrive under the DMA)
                loadw(c * 9 + 1, wf[1]);
#ifndef PP_ABL_NOSTAGE
                issue_dma(c * 16);
#endif
                if (k == 1) PPTRACE_AT(22);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (k == 1) PPTRACE_AT(23);
#ifndef PP_ABL_NOCOMMIT
                split_items(c * 16);
#endif
                if (k == 1) PPTRACE_AT(24);
            }
            if (k == nch) PPTRACE_AT(29);
            __builtin_amdgcn_s_setprio(0);
        }
        PPTRACE(slot); ++slot;
        __syncthreads();
        PPTRACE(slot); ++slot;
        if (k < Kh) {                                    // COMPUTE phase
            compute(c);
            c = (c + 1 == nch) ? 0 : c + 1;
        }
        PPTRACE(slot); ++slot;
        __syncthreads();
        PPTRACE(slot); ++slot;
    }
    if (h == 0) __syncthreads();
    PPTRACE_AT(31);
}

// ---- host side ------------------------------------------------------------------------------------------------------
namespace {
int pp_cu_count() {                                      // init-once device probe (the only cached state)
    static int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus;
}
}  // namespace

// What the ping-pong kernel takes (everything else stays on conv_split_kernel): fp32-equivalent or 2-part arithmetic, zero
// padding 1, rows of whole 16-byte units, 16-byte aligned tensors (LDS-DMA staging and 16-byte stores), activation split on an
// 8-cout boundary.
bool motif_conv_pp_eligible(const MotifConvDesc* d, const ConvArgs& a, int P) {
    if (split_parts(d->mma) < 2 || d->pad != 1 || d->pad_mode != 0 || (d->W & 3)) return false;
    const long HW = (long)d->H * d->W;
    if (HW * 64 >= 0x7fffffffL) return false;
    const int Cout_g = d->Cout / d->groups;
    if (d->act_split > 0 && ((d->act_split & 7) || (d->groups > 1 && (Cout_g & 7)))) return false;
    if (d->C1 > 0 && (d->groups != 1 || d->C0 % 16)) return false;
    for (int i = 0; i < P; ++i) {
        unsigned long long bits = (unsigned long long)a.in0[i] | (unsigned long long)a.out[i] | (unsigned long long)a.in1[i] | (unsigned long long)a.res[i];
        if (bits & 15) return false;
        if ((a.in0_bs[i] | a.out_bs[i] | (a.in1[i] ? a.in1_bs[i] : 0) | (a.res[i] ? a.res_bs[i] : 0)) & 3) return false;
    }
    return true;
}

// One persistent workgroup per CU (or per tile when there are fewer tiles than CUs); tile t of half h of workgroup b = i*2G + h*G + b'.
int motif_conv_pp_launch(const MotifConvDesc* d, ConvArgs& a, int P, hipStream_t s) {
    const int Cin_g = (d->C0 + d->C1) / d->groups, Cout_g = d->Cout / d->groups;
    const int Ho = d->H, Wo = d->W;                      // pad 1
    const int NP = split_parts(d->mma);
    a.Ho = Ho; a.Wo = Wo; a.Cin_g = Cin_g; a.Cout_g = Cout_g;
    a.Kpad = 9 * ((Cin_g + 15) / 16);
    a.ncg = (Cout_g + 63) / 64;
    a.tiles_x = (Wo + 31) / 32;
    const int ncgG = d->groups * a.ncg, cus = pp_cu_count();
    const int tiles_y = (Ho + 7) / 8;
    const long T = (long)a.tiles_x * tiles_y * ncgG * d->N * P;
    if (T >= 0x7fffffffL) return MOTIF_ELIMIT;
    const int G = (int)(T < cus ? T : cus);
    const size_t ldsb = ((size_t)8 * 16 + (size_t)8 * (8 * 4 * 32 / 4) + (size_t)2 * NP * (2 * 340 + 4) + (size_t)8 * 448) * 16;
#define MOTIF_LAUNCH_PP(NPV)                                                                                            \
    do {                                                                                                                \
        static bool attr_set = false;                                                                                   \
        if (!attr_set) {                                                                                                \
            (void)hipFuncSetAttribute((const void*)conv_pp_kernel<NPV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        conv_pp_kernel<NPV><<<dim3(G, 1, 1), 512, ldsb, s>>>(a, (int)T, tiles_y);                                      \
    } while (0)
    a.Cout = d->Cout;
    a.CK = ncgG;                                         // unused by this kernel otherwise: carries groups * ncg
    if (NP == 3) MOTIF_LAUNCH_PP(3); else MOTIF_LAUNCH_PP(2);
#undef MOTIF_LAUNCH_PP
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

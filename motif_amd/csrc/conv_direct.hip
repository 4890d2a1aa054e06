// Bandwidth-shaped convolution for the NARROW layers on large maps (RAFT-small's bottleneck blocks at half resolution: 32 -> 8,
// 8 -> 8 (3x3), 32 -> 16 channels at 360x640): a few hundred MACs per pixel against 150-300 bytes of traffic per pixel.
// In the MFMA engine (conv_igemm.hip) such a layer fills 8 of the 32 rows of a matrix tile and spends its time in the
// load -> LDS -> barrier -> MFMA -> store latency of one block after the other (37-55 us per launch against 6-20 us of HBM time).
// Here a thread owns FOUR consecutive output pixels and up to 16 couts: per input channel it loads the pixels it needs with
// 16-byte accesses (the 3x3 halo columns with two dword loads), the weights of that channel are wave-uniform (scalar loads from
// the packed block conv_igemm.hip already uses: [k = (channel, tap)][32 couts], zero padded), and the whole layer is
// COUT * Cin * K * K fused multiply-adds per pixel on the vector ALU with nothing but occupancy between a load and its use.
// fp32 multiply-add in (channel, tap) order: the same arithmetic class as the fp32 MFMA engine (not the same summation order).
#include "conv_common.h"
#include <type_traits>

namespace {
template <int ACT> __device__ __forceinline__ float actd(float v) { return act_c<ACT>(v); }

__device__ __forceinline__ float act_any(float v, int ac) {          // ac is wave-uniform
    if (ac == MOTIF_ACT_RELU) return v > 0.f ? v : 0.f;
    if (ac == MOTIF_ACT_LRELU) return v > 0.f ? v : 0.1f * v;
    if (ac == MOTIF_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    if (ac == MOTIF_ACT_TANH) return tanhf(v);
    return v;
}
}  // namespace

// NCO = couts per thread (8 or 16), K = 1 or 3 (pad K/2, stride 1).  grid = (ceil(W/4 * H / 256), cout slices of NCO, N * P)
template <int NCO, int K>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs a) {
    const int pz = blockIdx.z / a.N, n = blockIdx.z - pz * a.N;
    const int W4 = a.W >> 2;
    const int q = blockIdx.x * 256 + threadIdx.x;            // quad index in the image
    if (q >= W4 * a.H) return;
    const int y = q / W4, x = (q - y * W4) * 4;
    const int co0 = blockIdx.y * NCO;
    const long HW = (long)a.H * a.W;
    const float* in0n = a.in0[pz] + (long)n * a.in0_bs[pz];
    const float* in1n = a.in1[pz] ? a.in1[pz] + (long)n * a.in1_bs[pz] : nullptr;
    // [Kpad][32]: row ((c >> 1) * T + t) * 2 + (c & 1), 32 couts (Cout <= 32: one group).  Read through the constant address space:
    // the indices are wave-uniform, and only a pointer the compiler knows to be read-only becomes scalar loads (s_load_dwordx8).
    typedef const __attribute__((address_space(4))) float* cptr;
    cptr wp = (cptr)a.wp[pz];
    constexpr int T = K * K;

    f32x4 acc[NCO];
#pragma unroll
    for (int o = 0; o < NCO; ++o) acc[o] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int Cin = a.Cin_g;
    auto plane_of = [&](int c) { return c < a.C0 ? in0n + (long)c * HW : in1n + (long)(c - a.C0) * HW; };
    auto wrow_of = [&](int c) { return wp + (long)(((c >> 1) * T) * 2 + (c & 1)) * 32 + co0; };       // tap t: + t * 64
    // channels in groups of CG: all of a group's loads are requested before the first multiply (a loop over single channels keeps
    // ONE load in flight per thread -- the trip count is a run-time value, the compiler will not pipeline it)
    constexpr int CG = K == 1 ? 8 : 2;
    auto group = [&](int c0, auto n_tag) {
        constexpr int NG = decltype(n_tag)::value;
        if constexpr (K == 1) {
            f32x4 v[NG];
#pragma unroll
            for (int i = 0; i < NG; ++i) v[i] = *(const f32x4*)(plane_of(c0 + i) + (long)y * a.W + x);
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                cptr wrow = wrow_of(c0 + i);
#pragma unroll
                for (int o = 0; o < NCO; ++o) {
                    const float w = wrow[o];
                    acc[o] += v[i] * w;
                }
            }
        } else {
            // rows y-1 .. y+1, columns x-1 .. x+4 (zero outside the image)
            f32x4 m[NG][3];
            float l[NG][3], rr[NG][3];
#pragma unroll
            for (int i = 0; i < NG; ++i)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int yy = y + dy - 1;
                    const bool rowok = yy >= 0 && yy < a.H;
                    const float* rp = plane_of(c0 + i) + (long)(rowok ? yy : y) * a.W + x;
                    m[i][dy] = *(const f32x4*)rp;
                    l[i][dy] = rp[x > 0 ? -1 : 0];
                    rr[i][dy] = rp[x + 4 < a.W ? 4 : 3];
                }
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                cptr wrow = wrow_of(c0 + i);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int yy = y + dy - 1;
                    const bool rowok = yy >= 0 && yy < a.H;
                    float r[6];
                    r[0] = (rowok && x > 0) ? l[i][dy] : 0.f; r[5] = (rowok && x + 4 < a.W) ? rr[i][dy] : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[1 + e] = rowok ? m[i][dy][e] : 0.f;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        cptr wt = wrow + (dy * 3 + dx) * 64;
#pragma unroll
                        for (int o = 0; o < NCO; ++o) {
                            const float w = wt[o];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[o][e] = fmaf(r[dx + e], w, acc[o][e]);
                        }
                    }
                }
            }
        }
    };
    int c = 0;
    for (; c + CG <= Cin; c += CG) group(c, std::integral_constant<int, CG>{});
    for (; c < Cin; ++c) group(c, std::integral_constant<int, 1>{});

    const float* bias = a.bias[pz];
    const float* resn = a.res_mode ? a.res[pz] + (long)n * a.res_bs[pz] : nullptr;
    float* outn = a.out[pz] + (long)n * a.out_bs[pz];
    const int rm = a.res_mode;
    const long pix = (long)y * a.W + x;
    // bias and residual values of all couts first: a load inside the store loop waits for every store issued before it (loads and
    // stores share the vmcnt counter, so the compiler can only wait for zero)
    float bvv[NCO];
    f32x4 rvv[NCO];
#pragma unroll
    for (int o = 0; o < NCO; ++o) {
        const int co = min(co0 + o, a.Cout - 1);
        bvv[o] = bias ? bias[co] : 0.f;
        rvv[o] = rm ? *(const f32x4*)(resn + (long)co * HW + pix) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int o = 0; o < NCO; ++o) {
        const int co = co0 + o;
        if (co >= a.Cout) break;
        const int ac = (a.act_split > 0 && co >= a.act_split) ? a.act2 : a.act;
        f32x4 v = acc[o] + bvv[o];
        const f32x4 rv = rvv[o];
        if (rm == 1) v += rv;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_any(v[e], ac);
        if (rm == 2) v += rv;
        else if (rm == 3) {
            v += rv;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        } else if (rm == 4) v *= rv;
        *(f32x4*)(outn + (long)co * HW + pix) = v;
    }
}

// Deep-K, tiny-Cout form (RAFT's flow head: 128 -> 2 channels, 3x3): the matrix engine spends a 32-row tile on 2 couts and is bound
// by its k-loop (52 us for 0.27 GFLOP).  Here the input channels are cut into NS slices, one per wave of the workgroup: a thread owns
// four pixels x NCO couts over its slice's channels, the NS partial sums meet in LDS and wave 0 adds them in slice order (fixed order:
// deterministic), then bias / activation / residual / store as above.
// 3x3 form (round 5): a slice's weights -- 9 x NCO floats per channel, 2 .. 4 of every 32-float row of the packed block -- are gathered
// into LDS once (64 loads in flight per instruction) and read from there.  Scalar loads of them were one row = one scalar-cache miss each,
// ~2 us of L2 latency per channel pair: every 529 .. 661 -> 2 flow head of PWC-Net took 80 us whatever the map size, RAFT's 128 -> 2 head
// 26 us.  The workgroup's dynamic LDS is NS regions of `rs` 16-byte units: a wave's weights first, its partial sums afterwards (the
// same wave, in program order: no barrier between the two uses).
// ROWS (round 6, K == 3): a workgroup owns WHOLE rows (64 / W4 of them; W <= 256), so the pixel left of a quad and the pixel right of it are held
// by the neighbouring lanes and come by a lane shuffle instead of two more loads per (channel, row): 12 registers per channel in flight instead
// of 18, and SIX channels' loads in flight where the 128-register cap of the 16-slice form allowed three -- PWC-Net's 529 .. 661 -> 2 flow heads
// (42 channels per slice: 14 load latencies, 75 us on every level) walk 7.  Same multiply-adds in the same order: bit-identical to the quad
// form (option conv_direct_quads = 1 keeps that one; tests/test_kernels_gpu.py compares them).
template <int NCO, int K, int NS, bool ROWS = false>
__global__ __launch_bounds__(64 * NS) void conv_direct_deep_kernel(ConvArgs a, int rs) {
    extern __shared__ __attribute__((aligned(16))) f32x4 dsm[];        // [NS][rs]
    const int pz = blockIdx.z / a.N, n = blockIdx.z - pz * a.N;
    const int W4 = a.W >> 2, nquads = W4 * a.H;
    const int lane = threadIdx.x & 63, slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bool live;
    int y, x;
    if constexpr (ROWS) {
        const int rpb = 64 / W4;                                     // rows per workgroup
        const int r = lane / W4, xq = lane - r * W4;
        const int yy = blockIdx.x * rpb + r;
        live = r < rpb && yy < a.H;
        y = min(yy, a.H - 1); x = xq * 4;
    } else {
        const int q = blockIdx.x * 64 + lane;
        live = q < nquads;
        const int qc = live ? q : nquads - 1;
        y = qc / W4; x = (qc - y * W4) * 4;
    }
    const long HW = (long)a.H * a.W;
    const float* in0n = a.in0[pz] + (long)n * a.in0_bs[pz];
    const float* in1n = a.in1[pz] ? a.in1[pz] + (long)n * a.in1_bs[pz] : nullptr;
    typedef const __attribute__((address_space(4))) float* cptr;
    cptr wp = (cptr)a.wp[pz];
    constexpr int T = K * K;
    const int Cin = a.Cin_g;
    const int per = ((Cin + NS - 1) / NS + 1) & ~1;                  // channels per slice, even
    const int cbeg = slice * per, cend = min(Cin, cbeg + per);
    float* wl = (float*)(dsm + slice * rs);                          // this wave's weights [channel][tap][cout] (K == 3)
    if constexpr (K == 3) {
        const int nw = (cend - cbeg) * T * NCO;
        for (int i0 = 0; i0 < nw; i0 += 64 * 8) {          // eight gathers in flight per lane, then their LDS stores
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 64 * u + lane;
                const int ic = i < nw ? i : 0;
                const int cl = ic / (T * NCO), r = ic - cl * (T * NCO), tap = r / NCO, o = r - tap * NCO;
                const int c = cbeg + cl;
                wv[u] = wp[(long)(((c >> 1) * T) * 2 + (c & 1)) * 32 + tap * 64 + o];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 64 * u + lane;
                if (i < nw) wl[i] = wv[u];
            }
        }
    }

    f32x4 acc[NCO];
#pragma unroll
    for (int o = 0; o < NCO; ++o) acc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto plane_of = [&](int c) { return c < a.C0 ? in0n + (long)c * HW : in1n + (long)(c - a.C0) * HW; };
    auto wrow_of = [&](int c) { return wp + (long)(((c >> 1) * T) * 2 + (c & 1)) * 32; };
    constexpr int CG = K == 1 ? 8 : ROWS ? 6 : (NS >= 16 ? 3 : 2);    // channels whose loads are in flight together (the 16-slice form serves small maps: latency; four spill at its 128-register cap)
    auto group = [&](int c0, auto n_tag) {
        constexpr int NG = decltype(n_tag)::value;
        if constexpr (K == 1) {
            f32x4 v[NG];
#pragma unroll
            for (int i = 0; i < NG; ++i) v[i] = *(const f32x4*)(plane_of(c0 + i) + (long)y * a.W + x);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                cptr wrow = wrow_of(c0 + i);
#pragma unroll
                for (int o = 0; o < NCO; ++o) acc[o] += v[i] * wrow[o];
            }
        } else {
            f32x4 m[NG][3];
            float l[NG][3], rr[NG][3];
#pragma unroll
            for (int i = 0; i < NG; ++i)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int yy = y + dy - 1;
                    const bool rowok = yy >= 0 && yy < a.H;
                    const float* rp = plane_of(c0 + i) + (long)(rowok ? yy : y) * a.W + x;
                    m[i][dy] = *(const f32x4*)rp;
                    if constexpr (!ROWS) {
                        l[i][dy] = rp[x > 0 ? -1 : 0];
                        rr[i][dy] = rp[x + 4 < a.W ? 4 : 3];
                    }
                }
            // every load of the group is REQUESTED before the first multiply: left alone, the scheduler sinks each channel's loads next to
            // their uses to save registers and the slice becomes one load latency per channel (72 us for 42 channels, round 5 ISA)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ROWS) {                       // the neighbours' edge pixels (lanes of other rows are never used: x > 0 / x + 4 < W below)
#pragma unroll
                for (int i = 0; i < NG; ++i)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        l[i][dy] = __shfl_up(m[i][dy][3], 1);
                        rr[i][dy] = __shfl_down(m[i][dy][0], 1);
                    }
            }
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const float* wrow = wl + (c0 + i - cbeg) * (T * NCO);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int yy = y + dy - 1;
                    const bool rowok = yy >= 0 && yy < a.H;
                    float r[6];
                    r[0] = (rowok && x > 0) ? l[i][dy] : 0.f; r[5] = (rowok && x + 4 < a.W) ? rr[i][dy] : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[1 + e] = rowok ? m[i][dy][e] : 0.f;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float* wt = wrow + (dy * 3 + dx) * NCO;
#pragma unroll
                        for (int o = 0; o < NCO; ++o) {
                            const float w = wt[o];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[o][e] = fmaf(r[dx + e], w, acc[o][e]);
                        }
                    }
                }
            }
        }
    };
    int c = cbeg;
    for (; c + CG <= cend; c += CG) group(c, std::integral_constant<int, CG>{});
    for (; c < cend; ++c) group(c, std::integral_constant<int, 1>{});
#pragma unroll
    for (int o = 0; o < NCO; ++o) dsm[slice * rs + o * 64 + lane] = acc[o];
    __syncthreads();
    if (slice != 0 || !live) return;
    const float* bias = a.bias[pz];
    const float* resn = a.res_mode ? a.res[pz] + (long)n * a.res_bs[pz] : nullptr;
    float* outn = a.out[pz] + (long)n * a.out_bs[pz];
    const int rm = a.res_mode;
    const long pix = (long)y * a.W + x;
    float bvv[NCO];
    f32x4 rvv[NCO];
#pragma unroll
    for (int o = 0; o < NCO; ++o) {
        const int co = min(o, a.Cout - 1);
        bvv[o] = bias ? bias[co] : 0.f;
        rvv[o] = rm ? *(const f32x4*)(resn + (long)co * HW + pix) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int o = 0; o < NCO; ++o) {
        if (o >= a.Cout) break;
        f32x4 v = dsm[o * 64 + lane];
#pragma unroll
        for (int sidx = 1; sidx < NS; ++sidx) v += dsm[sidx * rs + o * 64 + lane];
        const int ac = (a.act_split > 0 && o >= a.act_split) ? a.act2 : a.act;
        v = v + bvv[o];
        const f32x4 rv = rvv[o];
        if (rm == 1) v += rv;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_any(v[e], ac);
        if (rm == 2) v += rv;
        else if (rm == 3) {
            v += rv;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        } else if (rm == 4) v *= rv;
        *(f32x4*)(outn + (long)o * HW + pix) = v;
    }
}

// ---- host side (called from conv_igemm.hip's forward entry; the packed weights are conv_igemm's) -------------------------
// Decided per launch (map size and alignment matter), never at pack time: both kernels read the same packed block.
// tiny-Cout / deep-K form (flow heads): any map size that fills the chip.  ONE predicate for eligibility and launch.
static bool direct_is_deep(const MotifConvDesc* d, int P) {
    const int Cin = d->C0 + d->C1;
    // small maps too when the reduction is long (PWC-Net's flow heads: 529 .. 661 -> 2 on 12x20 .. 96x160 maps took 200-250 us on the
    // MFMA engine -- one or a few workgroups walking 5 000 K-rows for 2 of a tile's 32 couts; here 8 waves share the channels: ~10 us)
    return d->Cout <= 4 && Cin >= 32 && (long)Cin * d->Cout * d->KH * d->KW <= 16384 && ((long)d->N * P * d->H * d->W >= 16384 || Cin >= 128) &&
           (d->C1 == 0 || (d->C0 & 1) == 0);
}

bool motif_conv_direct_eligible(const MotifConvDesc* d, const ConvArgs& a, int P) {
    if (motif_opt(MOTIF_OPT_CONV_NODIRECT)) return false;
    if (d->groups != 1 || d->stride != 1 || d->dil != 1 || d->KH != d->KW || (d->KH != 1 && d->KH != 3) || d->pad != d->KH / 2) return false;
    if (d->KH == 3 && d->pad_mode != 0) return false;
    const int Cin = d->C0 + d->C1;
    const bool deep = direct_is_deep(d, P);
    if (!deep && (d->Cout > 16 || Cin > 64)) return false;
    if (d->W & 3) return false;               // measured on the 360x640 layers: 32->8 38 -> 22 us, 8->8 3x3 33 -> 24,
                                                                             // 32->16 40 -> 29; 8->32 (store bound, two cout slices) 27 -> 31: not taken
    if (!deep && (long)Cin * d->Cout * d->KH * d->KW > 4096) return false;   // MACs per pixel: beyond this the matrix engine wins
    if (!deep && (long)d->N * P * d->H * d->W < 131072) return false;        // small maps: the MFMA engine's blocks fill the chip anyway
    for (int i = 0; i < P; ++i) {
        unsigned long long bits = (unsigned long long)a.in0[i] | (unsigned long long)a.out[i] | (unsigned long long)a.in1[i] | (unsigned long long)a.res[i];
        if (bits & 15) return false;
        if ((a.in0_bs[i] | a.out_bs[i] | (a.in1[i] ? a.in1_bs[i] : 0) | (a.res[i] ? a.res_bs[i] : 0)) & 3) return false;
    }
    return true;
}

int motif_conv_direct_launch(const MotifConvDesc* d, ConvArgs& a, int P, hipStream_t s) {
    a.Ho = d->H; a.Wo = d->W; a.Cin_g = d->C0 + d->C1; a.Cout_g = d->Cout;
    const long quads = (long)(d->W >> 2) * d->H;
    if (direct_is_deep(d, P)) {                          // deep form: 8 channel slices (waves) per 64 pixel quads
        dim3 grid((unsigned)((quads + 63) / 64), 1, d->N * P);
        // few workgroups and a long reduction (PWC-Net's flow heads on the coarse levels): 16 slices, four channels' loads in flight
        const int Cin = d->C0 + d->C1;
        // (decided from per-IMAGE quantities only: the slice count is the number of terms of the fixed-order partial-sum reduction, and an image
        // of a batch must come out with the bits it has when run alone)
        const bool tiny = d->KH == 3 && d->Cout <= 2 && (long)grid.x < 512 && Cin >= 256;
        auto region = [&](int ns, int nco) {               // 16-byte units per slice: its weights (3x3) or its partial sums, whichever is larger
            const int per = ((Cin + ns - 1) / ns + 1) & ~1;
            const int wq = d->KH == 3 ? (per * 9 * nco + 3) / 4 : 0;
            return wq > nco * 64 ? wq : nco * 64;
        };
#define MOTIF_LAUNCH_DEEP(NCOV, KV, NSV)                                                                                    \
    do {                                                                                                                     \
        const int rs_ = region(NSV, NCOV);                                                                                   \
        const size_t lds_ = (size_t)NSV * rs_ * 16;                                                                          \
        if (lds_ > 64 * 1024) (void)hipFuncSetAttribute((const void*)conv_direct_deep_kernel<NCOV, KV, NSV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); \
        conv_direct_deep_kernel<NCOV, KV, NSV><<<grid, 64 * NSV, lds_, s>>>(a, rs_);                                        \
    } while (0)
        // whole rows per workgroup where the row fits a wave (per-image quantities only, like `tiny`)
        const bool rows = d->KH == 3 && d->Cout <= 2 && (d->W >> 2) <= 64 && !motif_opt(MOTIF_OPT_CONV_DIRECT_QUADS);
#define MOTIF_LAUNCH_DEEP_ROWS(NSV)                                                                                        \
    do {                                                                                                                     \
        const int rpb = 64 / (d->W >> 2);                                                                                    \
        grid.x = (unsigned)((d->H + rpb - 1) / rpb);                                                                         \
        const int rs_ = region(NSV, 2);                                                                                      \
        const size_t lds_ = (size_t)NSV * rs_ * 16;                                                                          \
        if (lds_ > 64 * 1024) (void)hipFuncSetAttribute((const void*)conv_direct_deep_kernel<2, 3, NSV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); \
        conv_direct_deep_kernel<2, 3, NSV, true><<<grid, 64 * NSV, lds_, s>>>(a, rs_);                                      \
    } while (0)
        if (rows && tiny) MOTIF_LAUNCH_DEEP_ROWS(16);
        else if (rows) MOTIF_LAUNCH_DEEP_ROWS(8);                     // (RAFT's 128 -> 2 flow head: 8 slices of 16 channels, two groups instead of eight)
        else if (tiny) MOTIF_LAUNCH_DEEP(2, 3, 16);
        else if (d->KH == 1) MOTIF_LAUNCH_DEEP(4, 1, 8);
        else if (d->Cout <= 2) MOTIF_LAUNCH_DEEP(2, 3, 8);
        else MOTIF_LAUNCH_DEEP(4, 3, 8);
#undef MOTIF_LAUNCH_DEEP
#undef MOTIF_LAUNCH_DEEP_ROWS
        MOTIF_LAUNCH_CHECK();
        return MOTIF_OK;
    }
    constexpr int nco = 8;                               // eligibility caps Cout at 16: cout slices of 8 (a 16-wide slice was never reachable)
    dim3 grid((unsigned)((quads + 255) / 256), (d->Cout + nco - 1) / nco, d->N * P);
    if (d->KH == 1) conv_direct_kernel<8, 1><<<grid, 256, 0, s>>>(a); else conv_direct_kernel<8, 3><<<grid, 256, 0, s>>>(a);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

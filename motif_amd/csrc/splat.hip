// Fused reliability-aware soft-splat (forward warping) for gfx950.
//
// Two forms.  The OPERATOR form (motif_splat_fwd, the reference's module surface) and the fall-back for far sources: one thread
// per SOURCE pixel, the target corner indices and bilinear weights computed once (the reference recomputes flow+floor for every one
// of its 131 channels), then a channel loop of global_atomic_add_f32 (built with -munsafe-fp-atomics).  The fused MoTIF form
// (motif_splat_motif_*_fwd, the product path) is OWNER-COMPUTES: a workgroup owns a 16 x 64 tile of the accumulator, lists the
// sources of both directions that touch it, buckets the (source, corner) pairs by cell and GATHERS them, one thread per cell, into
// exact integer sums in registers -- no global atomics, no zero fill, bit-reproducible (see the block comment further down).
//
// Edges of the fused form (tests/test_kernels_gpu.py::test_splat_motif_value_clamp_and_non_finite_inputs): plane values are clamped
// to +-2^17 when they are staged (so +-Inf contribute as +-2^17 and a NaN leaves the accumulator finite -- the reference's float
// atomicAdd would propagate both); a source whose flow is not finite touches no cell, the hit count included (the reference
// asserts finite flows up front, softsplat_cp.py:25-26).  Far sources (fall-back kernel) are not clamped.
//
// Bit-level contract with the kernel text (softsplat_cp.py:27-38 etc.): floor -> int corners, weights
// as (SE - o) products, bounds test >=0 & <size, addend = (value*e^z) rounded, then * weight rounded.
#include "common.h"
#include <type_traits>

// The reference forms its flow tensor with separately rounded torch multiplies (Ours.py:794) and its kernel then ADDS the pixel
// index (softsplat_cp.py:27-28).  hipcc contracts a*b+c into one FMA by default, which rounds once and can move a coordinate
// across an integer -- a different floor(), i.e. other target cells and another count plane (caught by
// tests/test_kernels_gpu.py::test_splat_motif_rounds_the_flow_before_adding_the_pixel_index at scale ratio 3; exact products
// at power-of-two ratios hid it).  Every FMA in this file is an explicit fmaf.
#pragma clang fp contract(off)

struct Corners {
    int x0, y0;
    float wnw, wne, wsw, wse;
};

__device__ __forceinline__ Corners corners_of(int X, int Y, float fx, float fy) {
    Corners c;
    const float ox = (float)X + fx, oy = (float)Y + fy;
    c.x0 = (int)floorf(ox);
    c.y0 = (int)floorf(oy);
    const float xe = (float)(c.x0 + 1), ye = (float)(c.y0 + 1), xw = (float)c.x0, yn = (float)c.y0;
    c.wnw = (xe - ox) * (ye - oy);
    c.wne = (ox - xw) * (ye - oy);
    c.wsw = (xe - ox) * (oy - yn);
    c.wse = (ox - xw) * (oy - yn);
    return c;
}

__device__ __forceinline__ void atomic_max_float(float* addr, float value) {
    if (value >= 0) atomicMax((int*)addr, __float_as_int(value));
    else atomicMin((unsigned int*)addr, __float_as_uint(value));
}

struct Scatter {
    long onw, one, osw, ose;   // plane offsets of the four targets
    bool vnw, vne, vsw, vse;
    float wnw, wne, wsw, wse;
    __device__ __forceinline__ void init(const Corners& c, int H, int W) {
        const bool xl = c.x0 >= 0 && c.x0 < W, xr = c.x0 + 1 >= 0 && c.x0 + 1 < W;
        const bool yt = c.y0 >= 0 && c.y0 < H, yb = c.y0 + 1 >= 0 && c.y0 + 1 < H;
        vnw = xl && yt; vne = xr && yt; vsw = xl && yb; vse = xr && yb;
        onw = (long)c.y0 * W + c.x0; one = onw + 1; osw = onw + W; ose = osw + 1;
        wnw = c.wnw; wne = c.wne; wsw = c.wsw; wse = c.wse;
    }
    __device__ __forceinline__ void add(float* plane, float v) const {
        if (vnw) atomicAdd(plane + onw, v * wnw);
        if (vne) atomicAdd(plane + one, v * wne);
        if (vsw) atomicAdd(plane + osw, v * wsw);
        if (vse) atomicAdd(plane + ose, v * wse);
    }
    __device__ __forceinline__ void max(float* plane, float v) const {
        if (vnw) atomic_max_float(plane + onw, v * wnw);
        if (vne) atomic_max_float(plane + one, v * wne);
        if (vsw) atomic_max_float(plane + osw, v * wsw);
        if (vse) atomic_max_float(plane + ose, v * wse);
    }
    __device__ __forceinline__ void count(float* plane, float v) const {
        if (vnw) atomicAdd(plane + onw, v);
        if (vne) atomicAdd(plane + one, v);
        if (vsw) atomicAdd(plane + osw, v);
        if (vse) atomicAdd(plane + ose, v);
    }
};

// ---------------------------------------------------------------- operator form (module surface)
__global__ __launch_bounds__(256) void splat_plain_kernel(const float* src, const float* flow, const float* z,
                                                         float* out_sum, float* out_norm, float* out_max, float* out_cnt,
                                                         int C, int H, int W, int tiles_x) {
    const int n = blockIdx.z;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int X = tx * 64 + (threadIdx.x & 63), Y = ty * 4 + (threadIdx.x >> 6);
    if (X >= W || Y >= H) return;
    const long Q = (long)H * W, p = (long)Y * W + X;
    const float fx = flow[((long)n * 2 + 0) * Q + p], fy = flow[((long)n * 2 + 1) * Q + p];
    Scatter s;
    s.init(corners_of(X, Y, fx, fy), H, W);
    float e = 1.f;
    if (z) e = expf(z[(long)n * Q + p]);
    if (out_sum)
        for (int c = 0; c < C; ++c) {
            float v = src[((long)n * C + c) * Q + p];
            if (z) v = v * e;
            s.add(out_sum + ((long)n * C + c) * Q, v);
        }
    if (out_norm) s.add(out_norm + (long)n * Q, e);
    if (out_max) s.max(out_max + (long)n * Q, z ? e : src[(long)n * C * Q + p]);
    if (out_cnt) s.count(out_cnt + (long)n * Q, 1.0f);
}

extern "C" int motif_splat_fwd(const float* src, const float* flow, const float* z,
                               float* out_sum, float* out_norm, float* out_max, float* out_cnt,
                               int N, int C, int H, int W, void* stream) {
    if (!flow || N < 1 || C < 1 || H < 1 || W < 1) return MOTIF_EINVAL;
    if ((out_sum || (!z && out_max)) && !src) return MOTIF_EINVAL;
    const int tiles_x = (W + 63) / 64, tiles_y = (H + 3) / 4;
    dim3 grid(tiles_x * tiles_y, 1, N);
    splat_plain_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, flow, z, out_sum, out_norm, out_max, out_cnt, C, H, W, tiles_x);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// ---------------------------------------------------------------- fused MoTIF form: owner-computes
// Forward splatting is a scatter, but the predicted HR flow is locally bounded, so it can be turned
// inside out: a workgroup OWNS a 16x64 tile of the accumulator, scans the source pixels of BOTH
// directions within +-R of the tile and keeps those whose 2x2 footprint touches the tile (compacted into an
// LDS list in scan order: ballot counts per 64-source segment + one prefix sum).
//
// Round 3: the accumulation itself is a GATHER, one thread per accumulator cell.  (Rounds 1-2 scattered with
// ds_add_u64 into an LDS tile: 4 atomics per source and plane.  rocprofv3 on that kernel: LDS pipe busy 70 % of the
// workgroup's life at 16 cycles per atomic instruction -- half of them bank / same-address conflict cycles, a
// conflict-free ds_add_u64 costs 6.9 (tools/ubench_atomics.hip) -- vector ALU 28 %.)  Per tile:
//   1. scale pass: per cell, the maximum of e^z*weight over its sources (= the reference's max plane,
//      softsplat_max_cp.py:12-58, before its init-1 clamp), its binary exponent E, and the hit count (LDS integer atomics,
//      once per tile, not per plane);
//   2. the (source, corner) pairs are bucketed by cell (prefix sum of the counts, one returning atomic per pair);
//   3. per chunk of OT_CC = 4 planes: every listed source stages value*e^z (fp32, rounded as softsplat_cp.py:35-40 rounds it) in LDS,
//      then every cell walks its bucket: addend = staged value * corner weight (fp32, the reference's addend bit for bit),
//      scaled by 2^(32-E) and rounded to an integer in double precision (one fma against 1.5*2^52); the RAW BITS of those doubles are
//      summed as 64-bit integers in registers (see FIX_MAGIC below) -- integer sums are exact, so the result does not depend on the order of the
//      bucket (which the atomics of step 2 do not fix) and is bit-identical run to run and across tilings;
//   4. the sum is scaled back by 2^(E-32) and rounded to fp32 ONCE, and stored with plain coalesced stores.
// The per-cell scale E exists because the reliability weight e^z = exp(-20 relu(p2)) (Ours.py:794) spans the whole fp32
// range: a fixed 2^-32 grid would flush exactly the occluded sources soft-splatting exists for.  With it every addend is
// exact to 2^-32 of the cell's largest weight, so a normalised output sum/warped_z (Ours.py:811-814) is off by at most
// (hits per cell) * 2^-32 * max|value| -- below one fp32 ulp of the reference's own atomic sums, which vary run to run by more.
// No global atomics, no accumulator tile in LDS, no zero-fill pass, and the two directions are summed in registers.  A
// source whose footprint leaves its own +-R neighbourhood ("far") is skipped here and scattered by splat_far_kernel
// afterwards with global atomics -- both kernels evaluate the same predicate on the same inputs, so every source is
// accounted exactly once.  Tiles that collect more than OT_EB sources (sinks) are processed in batches of OT_EB with the
// buckets rebuilt per batch; the double accumulators run across the batches, so the result is the same exact sum.
#define OT_H 16
#define OT_W 64
#define OT_CC 4                         // planes per chunk
#define OT_TPH (OT_H + 2)
#define OT_TPW (OT_W + 2)
#define OT_TP (OT_TPH * OT_TPW)
#define OT_THREADS 1024                 // = OT_H * OT_W: thread t owns cell (t >> 6, t & 63) in the gather
#define OT_EB 2560                      // sources staged at a time (a tile of smooth flow lists ~2 * 17 * 65 = 2210; flow_imnet's flows with
                                        // the synthetic weights: p99 2313, max 2346 -- tools/splat_model_flow.py)
#define OT_NCACHE 3                     // list entries per thread whose geometry stays in registers (>= OT_EB / OT_THREADS)

struct MotifSplatArgs {
    const float* imnet_out; const float* pred; const float* feat_lr;
    const float* ab;      // PRE form: [2][64] = columns 64 and 65 of synth_net's first layer (the raw-flow channels)
    const int32_t* iy; const int32_t* ix; const float* alpha;
    float s20, sr; float* acc;
    int B, N, H, W, HH, WW, R;
    int row0;      // image row of local row 0 (row-band rendering): float coordinates are formed with GLOBAL rows so that
                   // floor() and the bilinear weights are bit-identical to the untiled render
    int accumulate;   // 0: acc is written; 1: this call's sums / count are added to acc and the max plane is max-ed into it
                      // (a further pair of source directions of the same frames: Ours_44.py:713-719 sums four)
};

struct SrcGeom {
    float p0, p1, e;
    float ox, oy;     // target position (global rows)
    int x0, y0;
    float wnw, wne, wsw, wse;
    bool near_;
};

// bilinear weight of corner q (bit 0: east, bit 1: south) of a source landing at (ox, oy): softsplat_cp.py:30-38
__device__ __forceinline__ float corner_weight(float ox, float oy, int q) {
    const float flx = floorf(ox), fly = floorf(oy);
    const float xe = flx + 1.f, ye = fly + 1.f;
    const float wx = (q & 1) ? ox - flx : xe - ox;
    const float wy = (q & 2) ? oy - fly : ye - oy;
    return wx * wy;
}

// geometry of one source pixel; `near_` = footprint within +-R of the source (float test: safe for huge flows)
__device__ __forceinline__ SrcGeom src_geom(const MotifSplatArgs& a, int img, int x, int y, bool need_z) {
    SrcGeom g;
    const long Q = (long)a.HH * a.WW, p = (long)y * a.WW + x;
    g.p0 = a.pred[((long)img * 3 + 0) * Q + p];
    g.p1 = a.pred[((long)img * 3 + 1) * Q + p];
    const float fx = (g.p0 * a.s20) * a.sr, fy = (g.p1 * a.s20) * a.sr;          // Ours.py:794
    const float yg = (float)(y + a.row0);
    const float ox = (float)x + fx, oy = yg + fy;
    const float flx = floorf(ox), fly = floorf(oy);
    const float R = (float)a.R;
    g.near_ = (flx >= (float)x - R) && (flx + 1.f <= (float)x + R) && (fly >= yg - R) && (fly + 1.f <= yg + R);
    g.x0 = g.near_ ? (int)flx : 0;
    g.y0 = g.near_ ? (int)fly - a.row0 : 0;
    g.ox = ox; g.oy = oy;
    const float xe = flx + 1.f, ye = fly + 1.f;
    g.wnw = (xe - ox) * (ye - oy);
    g.wne = (ox - flx) * (ye - oy);
    g.wsw = (xe - ox) * (oy - fly);
    g.wse = (ox - flx) * (oy - fly);
    g.e = 1.f;
    if (need_z) {
        const float p2 = a.pred[((long)img * 3 + 2) * Q + p];
        g.e = expf((p2 > 0.f ? p2 : 0.f) * a.alpha[0]);
    }
    return g;
}

// binary exponent E of a cell's largest e^z*weight m (float bits; m >= 0): m * 2^-E in [1,2); E = 0 for an empty cell
__device__ __forceinline__ int cell_exponent(unsigned bits) {
    return bits ? (int)(bits >> 23) - 127 : 0;
}

// Workgroup barrier that orders LDS accesses only.  __syncthreads() also waits for every global load and store in flight
// (vmcnt(0)): in the plane loop that would end the prefetch of the next chunk's plane values at the first barrier and make each
// barrier wait for the acknowledgement of the previous chunk's stores.  No thread reads global memory another thread of the
// workgroup wrote, so LDS ordering is all the loop needs.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Fixed point without conversion instructions: for |y| < 2^51 the significand field of the double (y + 1.5 * 2^52) is
// 2^51 + round(y) (RNE), so the RAW BITS of such doubles can be summed as 64-bit integers (one v_lshl_add_u64 each) and
// n * bits(1.5 * 2^52) subtracted at the end: what is left is the exact integer sum of the round(y).  y = addend * 2^(32-E)
// with |addend| < |value| * 2^(E+1), so plane values are clamped to +-2^17 when they are staged.
#define FIX_MAGIC 6755399441055744.0                        // 1.5 * 2^52
#define FIX_MAGIC_BITS 0x4338000000000000ull                // its bit pattern
#define FIX_VMAX 131072.0f                                  // 2^17

#ifdef MOTIF_TRACE
// phase clocks of the owner kernel (s_memtime), thread 0 of the first 2048 workgroups:
// 0 scan | 1 list | 2 scale + buckets | 3 barrier after staging | 4 barrier after gather | 5 write | 6 (list length) | 7 staging | 8 gather loop
__device__ long long g_sptrace[2048 * 12];
__device__ long long g_sptrace2[64 * 16 * 12];     // absolute clocks of chunk 3, every wave of the first 64 workgroups
extern "C" int motif_debug_splat_trace2(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sptrace2), sizeof(long long) * n); }
extern "C" int motif_debug_splat_trace(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sptrace), sizeof(long long) * n); }
#define SPH(i) do { const long long now_ = __builtin_amdgcn_s_memtime(); ph[i] += now_ - tlast; tlast = now_; if (trace_k == 3) ts[i] = now_; } while (0)

#else
#define SPH(i)
#endif

// PRE = false: the 130 source planes of Ours.py:786-791 -> acc [.,133,Q].
// PRE = true : the splat is linear in its sources and synth_net's first layer is linear in the normalised splat
//   (Ours.py:811-814, 839-856), so the 130 planes are contracted with W0[:, 0:130] BEFORE the splat: a source carries
//   the 64 values  U[c] + G[c] + A[c]*p0 + B[c]*p1  with  U = (W0[:, 0:64] . imnet head) (HR, folded into the imnet
//   kernel's head weights),  G = W0[:, 66:130] . feat_low (a 1x1 convolution at LR, gathered here),  A, B = W0[:, 64],
//   W0[:, 65].  Half the planes to accumulate, write and re-read: acc [.,67,Q] = 64 sums | norm | max | count.
//   GLR = false (PRE only): the caller's U already holds U + G (motif_siren_imnet_add_fwd adds the gathered LR term when it
//   stores U, same rounding as the sum formed here), so a source costs ONE load per plane and a.feat_lr is not read.
template <bool PRE, bool GLR>
__global__ __launch_bounds__(OT_THREADS) void splat_owner_kernel(MotifSplatArgs a, int cap) {
    static_assert(OT_CC == 4, "a staged source is one 16-byte vector");
    static_assert(PRE || GLR, "the literal form always gathers the LR features");
#ifdef MOTIF_TRACE
    long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = __builtin_amdgcn_s_memtime();
    const long long tstart = tlast;
    long long ts[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int trace_k = -1;
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* stage = lds;                                              // [2][OT_EB][OT_CC] value * e^z of the staged sources, two chunk buffers, 16 bytes
                                                                     // per source (consecutive sources = consecutive banks)
    float* oxy = stage + 2 * OT_EB * OT_CC;                          // [OT_EB][2] their target positions (batched form)
    unsigned short* list = (unsigned short*)(oxy + OT_EB * 2);       // [cap] (direction << 15) | (region row << 8) | region column
    float* bucket_w = oxy;                                           // [4 * OT_EB] single-batch form: corner weight of each bucket entry, over
                                                                     // oxy + list (the list is dead once the entries are cached in registers)
    unsigned* tmaxb = (unsigned*)(bucket_w + 4 * OT_EB);             // [OT_TP] float bits of max e^z*w (>= 0: unsigned order)
    unsigned* tcnt = tmaxb + OT_TP;                                  // [OT_TP] hit count
    unsigned* cfill = tcnt + OT_TP;                                  // [OT_TP] bucket counts / fill cursors of the current batch
    unsigned* segbase = cfill + OT_TP;                               // [192] hits per 64-source segment, then their prefix sums
    unsigned short* cstart = (unsigned short*)(segbase + 192);       // [OT_TP + 2] bucket starts of the current batch (< 4 * OT_EB)
    unsigned short* bucket = cstart + OT_TP + 2;                     // [4 * OT_EB] (staged source << 2) | corner
    float* abl = (float*)(bucket + 4 * OT_EB);                       // [128] PRE: a.ab (read through LDS: the compiler will not use scalar loads
                                                                     // for a pointer it cannot prove unaliased with the stores, and every
                                                                     // vector load of a uniform value waits for all loads in flight)
    signed char* texp = (signed char*)(abl + 128);                   // [OT_TP] per-cell scale exponent E
    __shared__ unsigned count;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid over (tile, frame) with the FRAME fastest, in XCD-aware order: an XCD gets a contiguous run of ids, i.e. all B*N frames
    // of a tile and then the next tiles (several tile rows).  The frames of a tile read the same source values U (only their flows
    // differ), so the second and third frame find them in that XCD's L2 instead of streaming the 472 MB tensor from HBM once per
    // frame; the +-16-pixel scan regions of neighbouring tiles share the L2 as before.
    const int BN = a.B * a.N, gid = xcd_tile_id();
    const int bn = gid % BN, b = bn / a.N, n = bn % a.N;
    const int tiles_x = (a.WW + OT_W - 1) / OT_W, tile_id = gid / BN;
    const int tx0 = (tile_id % tiles_x) * OT_W, ty0 = (tile_id / tiles_x) * OT_H;
    const long Q = (long)a.HH * a.WW, HWl = (long)a.H * a.W;
    if (tid == 0) count = 0;
    for (int i = tid; i < OT_TP; i += OT_THREADS) { tmaxb[i] = 0u; tcnt[i] = 0u; }
    if (PRE && tid < 128) abl[tid] = a.ab[tid];
    __syncthreads();

    // ---- pass 1: compact the contributing sources of both directions
    const int ry0 = max(ty0 - a.R, 0), ry1 = min(ty0 + OT_H - 1 + a.R, a.HH - 1);
    const int rx0 = max(tx0 - a.R, 0), rx1 = min(tx0 + OT_W - 1 + a.R, a.WW - 1);
    const int RH = ry1 - ry0 + 1, RW = rx1 - rx0 + 1;
    const int xiters = (RW + 63) >> 6;
    // The list is written in SCAN ORDER (direction, row, column), not in the order the waves happen to finish: ballot counts per
    // 64-source segment, one prefix sum, then every hit goes to its rank.  Consecutive entries are then consecutive sources of
    // one row, so the plane loads of the staging step are coalesced.  Slot j = (row step, direction, segment) is a compile-time
    // loop with unconditional, clamped loads.
    constexpr int NWV = OT_THREADS / 64, RSTEPS = (OT_H + 32 + NWV - 1) / NWV, XIT = (OT_W + 32 + 63) / 64, NJ = RSTEPS * 2 * XIT;
    static_assert(NJ <= 32 && 2 * (OT_H + 32) * XIT <= 192, "hit bits / segment scan sizes");
    static_assert(OT_H + 32 <= 128 && OT_W + 32 <= 128, "list entry fields");
    static_assert(OT_EB * 4 <= 65536 && OT_EB % 4 == 0 && OT_TP % 2 == 0, "bucket entries / starts are 16 bits: (staged source << 2) | corner");
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned hitbits = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = j / (2 * XIT), d = (j / XIT) & 1, it = j % XIT;
        const int row = wave + NWV * r, xc = it * 64 + lane;
        const bool valid = row < RH && xc < RW;
        const SrcGeom g = src_geom(a, (d * a.B + b) * a.N + n, rx0 + min(xc, RW - 1), ry0 + min(row, RH - 1), false);
        const bool hit = valid && g.near_ && g.x0 >= tx0 - 1 && g.x0 <= tx0 + OT_W - 1 && g.y0 >= ty0 - 1 && g.y0 <= ty0 + OT_H - 1;
        const unsigned long long m = __ballot(hit);
        if (lane == 0 && row < RH && it < xiters) segbase[(d * RH + row) * xiters + it] = (unsigned)__popcll(m);
        if (hit) hitbits |= 1u << j;
    }
    __syncthreads();
    SPH(0);
    if (wave == 0) {
        const int nseg = 2 * RH * xiters;
        unsigned v[3], sum = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) { const int sid = lane * 3 + i; v[i] = sid < nseg ? segbase[sid] : 0u; sum += v[i]; }
        unsigned incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        unsigned run = incl - sum;
#pragma unroll
        for (int i = 0; i < 3; ++i) { const int sid = lane * 3 + i; if (sid < nseg) segbase[sid] = run; run += v[i]; }
        if (lane == 63) count = incl;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = j / (2 * XIT), d = (j / XIT) & 1, it = j % XIT;
        const int row = wave + NWV * r;
        const bool hit = (hitbits >> j) & 1u;
        const unsigned long long m = __ballot(hit);
        if (hit) list[segbase[(d * RH + row) * xiters + it] + (unsigned)__popcll(m & lt)] = (unsigned short)((d << 15) | (row << 8) | (it * 64 + lane));
    }
    __syncthreads();
    SPH(1);
    const int cnt = (int)count;

    // ---- scale pass: per cell max of e^z*weight (= the max plane before its init-1 clamp) and the hit count.
    // The geometry of a thread's first OT_NCACHE list entries (entry = tid + i * OT_THREADS) stays in registers for everything
    // below; longer lists (sinks) recompute theirs from the list.
    // c_up / c_fp: the entry's element of plane 0 of the HR / LR source tensors -- plane c is a UNIFORM offset away, so a plane
    // load costs one 64-bit add (per-lane 64-bit index products cost six vector instructions per load, two of them quarter rate)
    struct Ent { int off; float ox, oy, e, p0, p1; const float* up; const float* fp; };
    auto make_ent = [&](int e) -> Ent {
        const unsigned ent = list[e];
        const int d = ent >> 15, y = ry0 + ((ent >> 8) & 127), x = rx0 + (ent & 255);
        const SrcGeom g = src_geom(a, (d * a.B + b) * a.N + n, x, y, true);
        Ent t;
        t.off = (g.y0 - (ty0 - 1)) * OT_TPW + (g.x0 - (tx0 - 1));
        t.ox = g.ox; t.oy = g.oy; t.e = g.e; t.p0 = g.p0; t.p1 = g.p1;
        t.up = a.imnet_out + (long)(d * a.B + b) * 64 * Q + ((long)y * a.WW + x);
        t.fp = GLR ? a.feat_lr + (long)(d * a.B + b) * 64 * HWl + ((long)a.iy[y] * a.W + a.ix[x]) : nullptr;
        return t;
    };
    // Two forms of everything below.  SINGLE (cnt <= OT_EB, every tile of a smooth flow): one batch, buckets built once, all entries
    // cached.  Otherwise (sinks): batches of OT_EB, buckets rebuilt per batch and chunk, entries recomputed from the list.
    Ent ce[OT_NCACHE];
    const int niter = (cnt + OT_THREADS - 1) / OT_THREADS;
    const bool single = cnt <= OT_EB;
    if (single) {
#pragma unroll
        for (int i = 0; i < OT_NCACHE; ++i) {
            const int e = tid + i * OT_THREADS;
            ce[i].off = 0; ce[i].ox = 0.f; ce[i].oy = 0.f; ce[i].e = 0.f; ce[i].p0 = 0.f; ce[i].p1 = 0.f; ce[i].up = a.imnet_out; ce[i].fp = GLR ? a.feat_lr : nullptr;
            if (e < cnt) ce[i] = make_ent(e);
        }
    }
    // f(entry, list index) for this thread's entries in [b0, b1)
    auto for_entries = [&](auto single_tag, int b0, int b1, auto&& f) {
        if constexpr (decltype(single_tag)::value) {
#pragma unroll
            for (int i = 0; i < OT_NCACHE; ++i) {
                const int e = tid + i * OT_THREADS;
                if (e < b1) f(ce[i], e);
            }
        } else {
            for (int i = 0; i < niter; ++i) {
                const int e = tid + i * OT_THREADS;
                if (e >= b0 && e < b1) f(make_ent(e), e);
            }
        }
    };
    auto scale_entry = [&](const Ent& t, int) {
        unsigned* tm = tmaxb + t.off;
        atomicMax(tm, __float_as_uint(t.e * corner_weight(t.ox, t.oy, 0)));
        atomicMax(tm + 1, __float_as_uint(t.e * corner_weight(t.ox, t.oy, 1)));
        atomicMax(tm + OT_TPW, __float_as_uint(t.e * corner_weight(t.ox, t.oy, 2)));
        atomicMax(tm + OT_TPW + 1, __float_as_uint(t.e * corner_weight(t.ox, t.oy, 3)));
        unsigned* tn = tcnt + t.off;
        atomicAdd(tn, 1u);
        atomicAdd(tn + 1, 1u);
        atomicAdd(tn + OT_TPW, 1u);
        atomicAdd(tn + OT_TPW + 1, 1u);
    };
    if (single) for_entries(std::true_type{}, 0, cnt, scale_entry);
    else for_entries(std::false_type{}, 0, cnt, scale_entry);
    __syncthreads();
    for (int i = tid; i < OT_TP; i += OT_THREADS) texp[i] = (signed char)min(cell_exponent(tmaxb[i]), 127);
    __syncthreads();

    constexpr int NPL = PRE ? 64 : 130, NCH = NPL / OT_CC, NREM = NPL - NCH * OT_CC;    // OT_CC = 4 planes per chunk: 130 = 32*4 + 2, 64 = 16*4 + 0
    float* abase = a.acc + (long)bn * (NPL + 3) * Q;
    // c is uniform (chunk index * 8 + unrolled plane)
    auto plane_value = [&](int c, const Ent& t) -> float {
        if constexpr (PRE) {
            float u = t.up[(long)c * Q];
            if constexpr (GLR) u = u + t.fp[(long)c * HWl];
            return fmaf(abl[64 + c], t.p1, fmaf(abl[c], t.p0, u));
        } else {
            if (c < 64) return t.up[(long)c * Q];
            if (c == 64) return t.p0;
            if (c == 65) return t.p1;
            return t.fp[(long)(c - 66) * HWl];
        }
    };
    const int ly = tid >> 6, lx = tid & 63, mycell = (ly + 1) * OT_TPW + lx + 1;
    const int Y = ty0 + ly, X = tx0 + lx;
    const int myE = (int)texp[mycell];

    auto planes = [&](auto single_tag) {
        constexpr bool SINGLE = decltype(single_tag)::value;
        // ---- buckets of one batch of list entries [b0, b1): (staged index, corner) pairs grouped by cell
        auto build_buckets = [&](int b0, int b1) {
            for (int i = tid; i < OT_TP; i += OT_THREADS) cfill[i] = 0u;
            __syncthreads();
            for_entries(single_tag, b0, b1, [&](const Ent& t, int) {
                atomicAdd(cfill + t.off, 1u); atomicAdd(cfill + t.off + 1, 1u); atomicAdd(cfill + t.off + OT_TPW, 1u); atomicAdd(cfill + t.off + OT_TPW + 1, 1u);
            });
            __syncthreads();
            if (wave == 0) {                                              // exclusive prefix sum over the OT_TP cells, 19 per lane
                constexpr int PER = (OT_TP + 63) / 64;
                unsigned sum = 0;
#pragma unroll 1
                for (int i = 0; i < PER; ++i) { const int c = lane * PER + i; sum += c < OT_TP ? cfill[c] : 0u; }
                unsigned incl = sum;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
                unsigned run = incl - sum;
#pragma unroll 1
                for (int i = 0; i < PER; ++i) {
                    const int c = lane * PER + i;
                    if (c < OT_TP) { const unsigned v = cfill[c]; cstart[c] = (unsigned short)run; cfill[c] = 0u; run += v; }
                }
                if (lane == 63) cstart[OT_TP] = (unsigned short)incl;
            }
            __syncthreads();
            for_entries(single_tag, b0, b1, [&](const Ent& t, int e) {
                const unsigned le = (unsigned)(e - b0) << 2;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cell = t.off + (q & 1) + (q >> 1) * OT_TPW;
                    const unsigned slot = cstart[cell] + atomicAdd(cfill + cell, 1u);
                    bucket[slot] = (unsigned short)(le | (unsigned)q);
                    if constexpr (SINGLE) bucket_w[slot] = corner_weight(t.ox, t.oy, q);
                }
                if constexpr (!SINGLE) { oxy[2 * (e - b0)] = t.ox; oxy[2 * (e - b0) + 1] = t.oy; }
            });
            __syncthreads();
        };
        if constexpr (SINGLE) build_buckets(0, cnt);
        SPH(2);

        // ---- the feature planes in chunks of OT_CC, then [remaining features, norm | max | count]
        auto plane_store = [&](int c, float v) {                     // accumulator plane c of this thread's cell
            float* o = abase + (long)c * Q + (long)Y * a.WW + X;
            if (a.accumulate) v = *o + v;
            // The store is hidden from the compiler's wait-count bookkeeping on purpose: gfx9 counts loads and stores in ONE counter
            // (vmcnt) and they may complete out of order with respect to each other, so with a store pending every wait for a load
            // becomes vmcnt(0) -- which also waits for the plane loads just issued for the chunk after next.  Uncounted stores can
            // only make a wait for the k-th oldest load longer, never shorter (loads complete in order among themselves).
            asm volatile("global_store_dword %0, %1, off" :: "v"(o), "v"(v) : "memory");
        };
        // gather: this thread's cell walks its bucket over the staged values in `buf`; raw bits of (addend * 2^(32-E) + 1.5 * 2^52)
        // are summed as integers
        auto gather = [&](const f32x4* buf, unsigned long long (&acci)[OT_CC], auto nst_tag) {
            constexpr int nst = decltype(nst_tag)::value;
            const double up_scale = __longlong_as_double((long long)(1023 + 32 - myE) << 52);      // 2^(32-E)
            const int j0 = cstart[mycell], j1 = cstart[mycell + 1];
            if (j0 >= j1) return;
            // (a two-deep software pipeline of the LDS reads of this loop measured 7 % SLOWER: the four waves of a SIMD already cover them)
            auto weight = [&](int j, unsigned be) -> float {
                if constexpr (SINGLE) return bucket_w[j];
                else return corner_weight(oxy[2 * (be >> 2)], oxy[2 * (be >> 2) + 1], be & 3);
            };
            for (int j = j0; j < j1; ++j) {
                const unsigned be = bucket[j];
                const float w = weight(j, be);
                const f32x4 sv = buf[be >> 2];
#pragma unroll
                for (int cc = 0; cc < nst; ++cc) {
                    const float x = sv[cc] * w;
                    acci[cc] += (unsigned long long)__double_as_longlong(fma((double)x, up_scale, FIX_MAGIC));
                }
            }
        };
        const unsigned long long nadd = tcnt[mycell];                                              // addends per plane of this cell
        auto finish = [&](unsigned long long v) -> float {                                         // one rounding: exact integer sum * 2^(E-32)
            const double down_scale = __longlong_as_double((long long)(1023 - 32 + myE) << 52);
            return (float)((double)(long long)(v - nadd * FIX_MAGIC_BITS) * down_scale);
        };
        const bool inside = Y < a.HH && X < a.WW;
        auto write_last = [&](const unsigned long long (&acci)[OT_CC]) {
            if (!inside) return;
#pragma unroll
            for (int cc = 0; cc < NREM + 1; ++cc) plane_store(NCH * OT_CC + cc, finish(acci[cc]));
            float* om = abase + (long)(NCH * OT_CC + NREM + 1) * Q + (long)Y * a.WW + X;
            float vm = fmaxf(1.0f, __uint_as_float(tmaxb[mycell]));                                 // max-splat output starts at ones (softsplat_max_cp.py:254)
            if (a.accumulate) vm = fmaxf(*om, vm);
            *om = vm;
            plane_store(NCH * OT_CC + NREM + 2, (float)tcnt[mycell]);
        };
        f32x4* const sbuf = (f32x4*)stage;                                                         // two buffers of [OT_EB] x 4 planes

        if constexpr (SINGLE) {
            // Software pipeline over the chunks: the plane values of chunk k+2 are in flight (registers pvA / pvB alternate) while chunk
            // k is gathered from one stage buffer and chunk k+1 is staged into the other; one barrier per chunk.  A dependent batch of
            // these loads has a tail latency of several microseconds on the loaded chip (one DRAM page per plane and row), which is what
            // the two-chunk distance is for.
            constexpr int NLD = (PRE && GLR) ? 2 * OT_CC : OT_CC;
            float pvA[OT_NCACHE][NLD], pvB[OT_NCACHE][NLD];
            auto issue_loads = [&](int k, float (&pv)[OT_NCACHE][NLD]) {     // invalid entries read plane 0 of the tensors
#pragma unroll
                for (int i = 0; i < OT_NCACHE; ++i)
#pragma unroll
                    for (int cc = 0; cc < OT_CC; ++cc) {
                        const int c = k * OT_CC + cc;                       // uniform
                        if constexpr (PRE) {
                            pv[i][cc] = ce[i].up[(long)c * Q];
                            if constexpr (GLR) pv[i][OT_CC + cc] = ce[i].fp[(long)c * HWl];
                        } else {
                            pv[i][cc] = c < 64 ? ce[i].up[(long)c * Q] : c >= 66 ? ce[i].fp[(long)(c - 66) * HWl] : 0.f;
                        }
                    }
            };
            auto stage_chunk = [&](int k, const float (&pv)[OT_NCACHE][NLD], f32x4* buf) {
#pragma unroll
                for (int i = 0; i < OT_NCACHE; ++i) {
                    const int e = tid + i * OT_THREADS;
                    f32x4 v;
#pragma unroll
                    for (int cc = 0; cc < OT_CC; ++cc) {
                        const int c = k * OT_CC + cc;
                        float x;
                        if constexpr (PRE) x = fmaf(abl[64 + c], ce[i].p1, fmaf(abl[c], ce[i].p0, GLR ? pv[i][cc] + pv[i][NLD - OT_CC + cc] : pv[i][cc]));
                        else x = c == 64 ? ce[i].p0 : c == 65 ? ce[i].p1 : pv[i][cc];
                        v[cc] = __builtin_amdgcn_fmed3f(x, -FIX_VMAX, FIX_VMAX) * ce[i].e;
                    }
                    if (e < cnt) buf[e] = v;
                }
            };
            auto stage_last = [&](f32x4* buf) {                            // remaining feature planes (literal form: 128, 129) and the norm plane
#pragma unroll
                for (int i = 0; i < OT_NCACHE; ++i) {
                    const int e = tid + i * OT_THREADS;
                    if (e < cnt) {
                        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int cc = 0; cc < NREM + 1; ++cc) {
                            float x = 1.0f;
                            if (cc < NREM) x = __builtin_amdgcn_fmed3f(ce[i].fp[(long)(62 + cc) * HWl], -FIX_VMAX, FIX_VMAX);
                            v[cc] = x * ce[i].e;
                        }
                        buf[e] = v;
                    }
                }
            };
            static_assert(NCH % 2 == 0 && NCH >= 4, "the pipeline is unrolled by two chunks");
            issue_loads(0, pvA);
            issue_loads(1, pvB);
            stage_chunk(0, pvA, sbuf);
            issue_loads(2, pvA);
            lds_barrier();
            SPH(3);
            // one step: gather chunk k from buffer (k & 1), stage chunk k + 1 into the other, request chunk k + 3, store chunk k
            auto step = [&](int k, float (&pvn)[OT_NCACHE][NLD]) {
                unsigned long long acci[OT_CC];
#pragma unroll
                for (int cc = 0; cc < OT_CC; ++cc) acci[cc] = 0ull;
                gather(sbuf + (k & 1) * OT_EB, acci, std::integral_constant<int, OT_CC>{});
                SPH(8);
                if (k + 1 < NCH) {
                    stage_chunk(k + 1, pvn, sbuf + ((k + 1) & 1) * OT_EB);
                    if (k + 3 < NCH) issue_loads(k + 3, pvn);
                } else {
                    stage_last(sbuf + ((k + 1) & 1) * OT_EB);
                }
                SPH(7);
                if (inside) {
#pragma unroll
                    for (int cc = 0; cc < OT_CC; ++cc) plane_store(k * OT_CC + cc, finish(acci[cc]));
                }
                SPH(5);
                lds_barrier();
                SPH(4);
            };
            for (int k = 0; k < NCH; k += 2) {
                step(k, pvB);
                step(k + 1, pvA);
            }
            unsigned long long accl[OT_CC];
#pragma unroll
            for (int cc = 0; cc < OT_CC; ++cc) accl[cc] = 0ull;
            gather(sbuf + (NCH & 1) * OT_EB, accl, std::integral_constant<int, NREM + 1>{});
            SPH(9);
            write_last(accl);
            SPH(11);
        } else {
            // sinks: batches of OT_EB, buckets rebuilt per batch and chunk, plane values loaded where they are staged
            const int nbatch = (cnt + OT_EB - 1) / OT_EB;
            for (int k = 0; k <= NCH; ++k) {
                unsigned long long acci[OT_CC];
#pragma unroll
                for (int cc = 0; cc < OT_CC; ++cc) acci[cc] = 0ull;
                for (int bb = 0; bb < nbatch; ++bb) {
                    const int b0 = bb * OT_EB, b1 = min(cnt, b0 + OT_EB);
                    build_buckets(b0, b1);
                    for_entries(single_tag, b0, b1, [&](const Ent& t, int e) {
                        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int cc = 0; cc < OT_CC; ++cc) {
                            float x = 1.0f;
                            if (k < NCH) x = plane_value(k * OT_CC + cc, t);
                            else if (cc < NREM) x = t.fp[(long)(62 + cc) * HWl];
                            v[cc] = __builtin_amdgcn_fmed3f(x, -FIX_VMAX, FIX_VMAX) * t.e;
                        }
                        sbuf[e - b0] = v;
                    });
                    __syncthreads();
                    if (k < NCH) gather(sbuf, acci, std::integral_constant<int, OT_CC>{});
                    else gather(sbuf, acci, std::integral_constant<int, NREM + 1>{});
                    __syncthreads();
                }
                if (k < NCH) {
                    if (inside) {
#pragma unroll
                        for (int cc = 0; cc < OT_CC; ++cc) plane_store(k * OT_CC + cc, finish(acci[cc]));
                    }
                } else {
                    write_last(acci);
                }
            }
        }
    };
    if (single) planes(std::true_type{});
    else planes(std::false_type{});
#ifdef MOTIF_TRACE
    const int blk = blockIdx.x;
    if (tid == 0 && blk < 2048) {
        ph[6] = cnt;
        ph[10] = __builtin_amdgcn_s_memtime() - tstart;
        for (int i = 0; i < 12; ++i) g_sptrace[blk * 12 + i] = ph[i];
    }
    if (lane == 0 && blk < 64)
        for (int i = 0; i < 12; ++i) g_sptrace2[(blk * 16 + wave) * 12 + i] = ts[i];
#endif
}

// far sources (footprint outside their own +-R neighbourhood): rare; global atomics, after the owner pass
template <bool PRE, bool GLR>
__global__ __launch_bounds__(256) void splat_far_kernel(MotifSplatArgs a, int tiles_x) {
    const int img = blockIdx.z;
    const int n = img % a.N, db = img / a.N, b = db % a.B;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int X = tx * 64 + (threadIdx.x & 63), Y = ty * 4 + (threadIdx.x >> 6);
    if (X >= a.WW || Y >= a.HH) return;
    const SrcGeom g0 = src_geom(a, img, X, Y, false);
    if (g0.near_) return;
    const long Q = (long)a.HH * a.WW, p = (long)Y * a.WW + X, HWl = (long)a.H * a.W;
    const float p0 = g0.p0, p1 = g0.p1;
    const float p2 = a.pred[((long)img * 3 + 2) * Q + p];
    const float e = expf((p2 > 0.f ? p2 : 0.f) * a.alpha[0]);
    const float fx = (p0 * a.s20) * a.sr, fy = (p1 * a.s20) * a.sr;
    Scatter s;
    const float ox = (float)X + fx, oy = (float)(Y + a.row0) + fy;
    // far targets may be anywhere (or nowhere): clamp before the int conversion, bounds tests do the rest
    if (!(ox > -4.f && ox < (float)a.WW + 4.f && oy > (float)a.row0 - 4.f && oy < (float)(a.row0 + a.HH) + 4.f)) return;
    Corners cn = corners_of(X, Y + a.row0, fx, fy);
    cn.y0 -= a.row0;
    s.init(cn, a.HH, a.WW);
    constexpr int NPL = PRE ? 64 : 130;
    float* abase = a.acc + (long)(b * a.N + n) * (NPL + 3) * Q;
    const long lr = (long)a.iy[Y] * a.W + a.ix[X];
    for (int c = 0; c < NPL; ++c) {
        float v;
        if constexpr (PRE) {
            float u = a.imnet_out[((long)db * 64 + c) * Q + p];
            if constexpr (GLR) u = u + a.feat_lr[((long)db * 64 + c) * HWl + lr];
            v = fmaf(a.ab[64 + c], p1, fmaf(a.ab[c], p0, u));
        } else {
            if (c < 64) v = a.imnet_out[((long)db * 64 + c) * Q + p];
            else if (c == 64) v = p0;
            else if (c == 65) v = p1;
            else v = a.feat_lr[((long)db * 64 + (c - 66)) * HWl + lr];
        }
        s.add(abase + (long)c * Q, v * e);
    }
    s.add(abase + (long)NPL * Q, e);
    s.max(abase + (long)(NPL + 1) * Q, e);
    s.count(abase + (long)(NPL + 2) * Q, 1.0f);
}

template <bool PRE, bool GLR>
static int launch_motif_splat(MotifSplatArgs a, void* stream) {
    const int cap = 2 * (OT_H + 2 * a.R) * (OT_W + 2 * a.R);
    if ((size_t)OT_EB * 8 + (size_t)cap * 2 > (size_t)OT_EB * 16) return MOTIF_EINVAL;      // oxy + list lie inside the bucket_w region
    const size_t lds = (size_t)OT_EB * 2 * OT_CC * 4 + (size_t)OT_EB * 16 + (size_t)(3 * OT_TP + 192) * 4 + (size_t)(OT_TP + 2) * 2 + (size_t)4 * OT_EB * 2
                       + 128 * 4 + ((OT_TP + 3) & ~3);
    hipError_t e = hipFuncSetAttribute((const void*)splat_owner_kernel<PRE, GLR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    dim3 grid(((a.WW + OT_W - 1) / OT_W) * ((a.HH + OT_H - 1) / OT_H) * a.B * a.N, 1, 1);
    splat_owner_kernel<PRE, GLR><<<grid, OT_THREADS, lds, (hipStream_t)stream>>>(a, cap);
    MOTIF_LAUNCH_CHECK();
    const int tiles_x = (a.WW + 63) / 64, tiles_y = (a.HH + 3) / 4;
    dim3 grid2(tiles_x * tiles_y, 1, 2 * a.B * a.N);
    splat_far_kernel<PRE, GLR><<<grid2, 256, 0, (hipStream_t)stream>>>(a, tiles_x);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

extern "C" int motif_splat_motif_acc_fwd(const float* imnet_out, const float* pred, const float* feat_lr,
                                         const int32_t* iy, const int32_t* ix, const float* alpha, float flow_scale,
                                         float* acc, int B, int N, int H, int W, int HH, int WW, int row0, int accumulate, void* stream) {
    if (!imnet_out || !pred || !feat_lr || !iy || !ix || !alpha || !acc) return MOTIF_EINVAL;
    if (B < 1 || N < 1 || H < 1 || W < 1 || HH < 1 || WW < 1 || row0 < 0) return MOTIF_EINVAL;
    MotifSplatArgs a{imnet_out, pred, feat_lr, nullptr, iy, ix, alpha, 20.0f, flow_scale, acc, B, N, H, W, HH, WW, 16, row0, accumulate ? 1 : 0};
    return launch_motif_splat<false, true>(a, stream);
}

extern "C" int motif_splat_motif_pre_fwd(const float* u_hr, const float* pred, const float* g_lr, const float* ab,
                                         const int32_t* iy, const int32_t* ix, const float* alpha, float flow_scale,
                                         float* acc, int B, int N, int H, int W, int HH, int WW, int row0, int accumulate, void* stream) {
    if (!u_hr || !pred || !ab || !iy || !ix || !alpha || !acc) return MOTIF_EINVAL;      // g_lr may be NULL: u_hr already holds U + G
    if (B < 1 || N < 1 || H < 1 || W < 1 || HH < 1 || WW < 1 || row0 < 0) return MOTIF_EINVAL;
    MotifSplatArgs a{u_hr, pred, g_lr, ab, iy, ix, alpha, 20.0f, flow_scale, acc, B, N, H, W, HH, WW, 16, row0, accumulate ? 1 : 0};
    return g_lr ? launch_motif_splat<true, true>(a, stream) : launch_motif_splat<true, false>(a, stream);
}

extern "C" int motif_splat_motif_fwd(const float* imnet_out, const float* pred, const float* feat_lr,
                                     const int32_t* iy, const int32_t* ix, const float* alpha, float flow_scale,
                                     float* acc, int B, int N, int H, int W, int HH, int WW, int row0, void* stream) {
    return motif_splat_motif_acc_fwd(imnet_out, pred, feat_lr, iy, ix, alpha, flow_scale, acc, B, N, H, W, HH, WW, row0, 0, stream);
}

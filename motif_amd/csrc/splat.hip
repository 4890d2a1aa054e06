// Fused reliability-aware soft-splat (forward warping) for gfx950.
//
// One thread per SOURCE pixel: the target corner indices and bilinear weights are computed once
// (the reference recomputes flow+floor for every one of its 131 channels), then the channel loop
// scatters with global_atomic_add_f32 (built with -munsafe-fp-atomics).  A wave's 64 lanes are 64
// consecutive x of one source row, so for smooth flow every per-channel atomic instruction lands on
// 1-2 contiguous cache lines of the accumulator plane.  Sum, max and count share the index math.
//
// Bit-level contract with the kernel text (softsplat_cp.py:27-38 etc.): floor -> int corners, weights
// as (SE - o) products, bounds test >=0 & <size, addend = (value*e^z) rounded, then * weight rounded.
#include "common.h"

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
// directions within +-R of the tile, keeps those whose 2x2 footprint touches the tile (compacted into an
// LDS list in scan order: ballot counts per 64-source segment + one prefix sum), accumulates 8 accumulator planes at a time in LDS (the padded
// 18x66 tile absorbs footprint cells that spill over the border, so the inner loop has no bounds tests),
// and writes each finished plane tile with plain coalesced stores.
// LDS accumulation is 32.32 FIXED POINT with ds_add_u64, on a PER-CELL binary scale: measured on MI355X
// (tools/ubench_atomics.hip) ds_add_f32 retires one wave-instruction per ~194 cycles per CU, integer LDS
// atomics one per ~5 -- float LDS atomics are 40x slower than integer ones and slower than
// global_atomic_add_f32.  The reliability weight e^z = exp(-20 relu(p2)) (Ours.py:794) spans the whole fp32
// range, so a fixed 2^-32 grid would flush exactly the occluded sources soft-splatting exists for.  A scale
// pass therefore first takes, per accumulator cell, the maximum of e^z*weight over the contributing sources
// (the reference's max plane, softsplat_max_cp.py:12-58, before its init-1 clamp) and its binary exponent E.
// Each addend (value*e^z, then *weight, both rounded to fp32 exactly as softsplat_cp.py:35-40 does) is
// multiplied by 2^-E -- exact, folded into the four corner weights -- so that the largest e^z*weight of a cell
// lands in [1,2), then converted exactly up to 2^-32 (floor / fract / two cvt), summed as integers -- order
// independent, deterministic -- and rounded to fp32 ONCE, together with the 2^E back-scale, at write-out.
// Error bound: every addend is exact to 2^-32 of the cell's largest weight, so a normalised output
// sum/warped_z (Ours.py:811-814) is off by at most (hits per cell) * 2^-32 * max|value| -- below one fp32 ulp
// of the reference's own atomic sums, which vary run to run by more.
// No global atomics, no zero-fill pass, and the two directions are summed in LDS.  A source whose
// footprint leaves its own +-R neighbourhood ("far") is skipped here and scattered by
// splat_far_kernel afterwards with global atomics -- both kernels evaluate the same predicate on the
// same inputs, so every source is accounted exactly once.
#define OT_H 16
#define OT_W 64
#define OT_CC 8
#define OT_TPH (OT_H + 2)
#define OT_TPW (OT_W + 2)
#define OT_TP (OT_TPH * OT_TPW)
#define OT_THREADS 1024

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
    int x0, y0;
    float wnw, wne, wsw, wse;
    bool near_;
};

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

__device__ __forceinline__ void add_fix(unsigned long long* cell, float x) {
    const float fl = floorf(x);
    const unsigned lo = (unsigned)((x - fl) * 4294967296.0f);        // fract < 1, exact; cvt saturates
    const int hi = (int)fl;
    atomicAdd(cell, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)lo);
}

__device__ __forceinline__ float fix_to_float(unsigned long long v, int E) {
    return (float)ldexp((double)(long long)v, E - 32);        // one rounding: exact integer sum * 2^(E-32)
}

// binary exponent E of a cell's largest e^z*weight m (float bits; m >= 0): m * 2^-E in [1,2); E = 0 for an empty cell
__device__ __forceinline__ int cell_exponent(unsigned bits) {
    return bits ? (int)(bits >> 23) - 127 : 0;
}

// PRE = false: the 130 source planes of Ours.py:786-791 -> acc [.,133,Q].
// PRE = true : the splat is linear in its sources and synth_net's first layer is linear in the normalised splat
//   (Ours.py:811-814, 839-856), so the 130 planes are contracted with W0[:, 0:130] BEFORE the splat: a source carries
//   the 64 values  U[c] + G[c] + A[c]*p0 + B[c]*p1  with  U = (W0[:, 0:64] . imnet head) (HR, folded into the imnet
//   kernel's head weights),  G = W0[:, 66:130] . feat_low (a 1x1 convolution at LR, gathered here),  A, B = W0[:, 64],
//   W0[:, 65].  Half the planes to accumulate, write and re-read: acc [.,67,Q] = 64 sums | norm | max | count.
template <bool PRE>
__global__ __launch_bounds__(OT_THREADS) void splat_owner_kernel(MotifSplatArgs a, int cap) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned long long* tile = (unsigned long long*)lds;            // [OT_CC][OT_TP] 32.32 fixed point
    unsigned* tmaxb = (unsigned*)(tile + OT_CC * OT_TP);             // [OT_TP] float bits of max e^z*w (>= 0: unsigned order)
    unsigned* tcnt = tmaxb + OT_TP;                                  // [OT_TP] hit count
    int* texp = (int*)(tcnt + OT_TP);                                // [OT_TP] per-cell scale exponent E
    unsigned* list = (unsigned*)(texp + OT_TP);                      // [cap]
    unsigned* segbase = list + cap;                                  // [192] hits per 64-source segment, then their prefix sums
    __shared__ unsigned count;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bn = blockIdx.z, b = bn / a.N, n = bn % a.N;
    // 1-D tile grid in XCD-aware order: an XCD gets a contiguous run of tiles (several tile rows), so the +-16-pixel scan regions
    // of neighbouring tiles share its L2
    const int tiles_x = (a.WW + OT_W - 1) / OT_W, tile_id = xcd_tile_id();
    const int tx0 = (tile_id % tiles_x) * OT_W, ty0 = (tile_id / tiles_x) * OT_H;
    const long Q = (long)a.HH * a.WW, HWl = (long)a.H * a.W;
    if (tid == 0) count = 0;
    for (int i = tid; i < OT_TP; i += OT_THREADS) { tmaxb[i] = 0u; tcnt[i] = 0u; }
    __syncthreads();

    // ---- pass 1: compact the contributing sources of both directions
    const int ry0 = max(ty0 - a.R, 0), ry1 = min(ty0 + OT_H - 1 + a.R, a.HH - 1);
    const int rx0 = max(tx0 - a.R, 0), rx1 = min(tx0 + OT_W - 1 + a.R, a.WW - 1);
    const int RH = ry1 - ry0 + 1, RW = rx1 - rx0 + 1;
    const int xiters = (RW + 63) >> 6;
    // The list is written in SCAN ORDER (direction, row, column), not in the order the waves happen to finish: ballot counts per
    // 64-source segment, one prefix sum, then every hit goes to its rank.  Consecutive entries are then consecutive sources of
    // one row -- whose targets are consecutive accumulator cells for any smooth flow -- so the 64 lanes of an LDS-atomic instruction
    // in pass 2 hit 64 consecutive 8-byte cells (bank-conflict free) instead of pieces of several rows (rocprofv3: 56 % of the LDS
    // cycles of this kernel were bank conflicts with the append-as-you-go list).  Slot j = (row step, direction, segment) is a
    // compile-time loop with unconditional, clamped loads.
    constexpr int NWV = OT_THREADS / 64, RSTEPS = (OT_H + 32 + NWV - 1) / NWV, XIT = (OT_W + 32 + 63) / 64, NJ = RSTEPS * 2 * XIT;
    static_assert(NJ <= 32 && 2 * (OT_H + 32) * XIT <= 192, "hit bits / segment scan sizes");
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
        if (hit) list[segbase[(d * RH + row) * xiters + it] + (unsigned)__popcll(m & lt)] = ((unsigned)d << 16) | ((unsigned)row << 8) | (unsigned)(it * 64 + lane);
    }
    __syncthreads();
    const int cnt = (int)count;

    // ---- scale pass: per cell max of e^z*weight (= the max plane before its init-1 clamp) and the hit count
    for (int e = tid; e < cnt; e += OT_THREADS) {
        const unsigned ent = list[e];
        const int d = ent >> 16, y = ry0 + ((ent >> 8) & 255), x = rx0 + (ent & 255);
        const SrcGeom g = src_geom(a, (d * a.B + b) * a.N + n, x, y, true);
        const int off = (g.y0 - (ty0 - 1)) * OT_TPW + (g.x0 - (tx0 - 1));
        unsigned* tm = tmaxb + off;
        atomicMax(tm, __float_as_uint(g.e * g.wnw));
        atomicMax(tm + 1, __float_as_uint(g.e * g.wne));
        atomicMax(tm + OT_TPW, __float_as_uint(g.e * g.wsw));
        atomicMax(tm + OT_TPW + 1, __float_as_uint(g.e * g.wse));
        unsigned* tn = tcnt + off;
        atomicAdd(tn, 1u);
        atomicAdd(tn + 1, 1u);
        atomicAdd(tn + OT_TPW, 1u);
        atomicAdd(tn + OT_TPW + 1, 1u);
    }
    __syncthreads();
    for (int i = tid; i < OT_TP; i += OT_THREADS) texp[i] = cell_exponent(tmaxb[i]);

    // ---- pass 2: the feature planes in chunks of 8, then [remaining features, norm | max | count]
    constexpr int NPL = PRE ? 64 : 130, NCH = NPL / OT_CC, NREM = NPL - NCH * OT_CC;    // 130 = 16*8 + 2, 64 = 8*8 + 0
    float* abase = a.acc + (long)bn * (NPL + 3) * Q;
    for (int k = 0; k <= NCH; ++k) {
        const bool last = (k == NCH);
        const int n64 = (last ? NREM + 1 : OT_CC) * OT_TP;
        for (int i = tid; i < n64; i += OT_THREADS) tile[i] = 0ull;
        __syncthreads();
        for (int e = tid; e < cnt; e += OT_THREADS) {
            const unsigned ent = list[e];
            const int d = ent >> 16, y = ry0 + ((ent >> 8) & 255), x = rx0 + (ent & 255);
            const int db = d * a.B + b, img = db * a.N + n;
            const SrcGeom g = src_geom(a, img, x, y, true);
            const int off = (g.y0 - (ty0 - 1)) * OT_TPW + (g.x0 - (tx0 - 1));
            const int* te = texp + off;
            // weight * 2^-E(cell): exact, so (v*e)*w' == ((v*e)*w) * 2^-E bit for bit
            const float wnw = ldexpf(g.wnw, -te[0]), wne = ldexpf(g.wne, -te[1]);
            const float wsw = ldexpf(g.wsw, -te[OT_TPW]), wse = ldexpf(g.wse, -te[OT_TPW + 1]);
            const long p = (long)y * a.WW + x;
            const long lr = (long)a.iy[y] * a.W + a.ix[x];
            if (!last) {
                float v[OT_CC];
#pragma unroll
                for (int cc = 0; cc < OT_CC; ++cc) {
                    const int c = k * OT_CC + cc;
                    if constexpr (PRE) {
                        const float u = a.imnet_out[((long)db * 64 + c) * Q + p] + a.feat_lr[((long)db * 64 + c) * HWl + lr];
                        v[cc] = fmaf(a.ab[64 + c], g.p1, fmaf(a.ab[c], g.p0, u));
                    } else {
                        if (c < 64) v[cc] = a.imnet_out[((long)db * 64 + c) * Q + p];
                        else if (c == 64) v[cc] = g.p0;
                        else if (c == 65) v[cc] = g.p1;
                        else v[cc] = a.feat_lr[((long)db * 64 + (c - 66)) * HWl + lr];
                    }
                }
#pragma unroll
                for (int cc = 0; cc < OT_CC; ++cc) {
                    const float ve = v[cc] * g.e;
                    unsigned long long* tc = tile + cc * OT_TP + off;
                    add_fix(tc, ve * wnw);
                    add_fix(tc + 1, ve * wne);
                    add_fix(tc + OT_TPW, ve * wsw);
                    add_fix(tc + OT_TPW + 1, ve * wse);
                }
            } else {
#pragma unroll
                for (int cc = 0; cc < NREM + 1; ++cc) {
                    const float ve = (cc < NREM) ? a.feat_lr[((long)db * 64 + 62 + cc) * HWl + lr] * g.e : g.e;
                    unsigned long long* tc = tile + cc * OT_TP + off;
                    add_fix(tc, ve * wnw);
                    add_fix(tc + 1, ve * wne);
                    add_fix(tc + OT_TPW, ve * wsw);
                    add_fix(tc + OT_TPW + 1, ve * wse);
                }
            }
        }
        __syncthreads();
        const int nplanes = last ? NREM + 3 : OT_CC;
        for (int i = tid; i < nplanes * OT_H * OT_W; i += OT_THREADS) {
            const int cc = i >> 10, rem = i & 1023, ly = rem >> 6, lx = rem & 63;
            const int Y = ty0 + ly, X = tx0 + lx;
            if (Y >= a.HH || X >= a.WW) continue;
            const int cell = (ly + 1) * OT_TPW + lx + 1;
            float v;
            if (!last || cc < NREM + 1) v = fix_to_float(tile[cc * OT_TP + cell], texp[cell]);
            else if (cc == NREM + 1) v = fmaxf(1.0f, __uint_as_float(tmaxb[cell]));     // max-splat output starts at ones (softsplat_max_cp.py:254)
            else v = (float)tcnt[cell];
            float* o = abase + (long)(k * OT_CC + cc) * Q + (long)Y * a.WW + X;
            if (a.accumulate) v = (last && cc == NREM + 1) ? fmaxf(*o, v) : *o + v;
            *o = v;
        }
        __syncthreads();
    }
}

// far sources (footprint outside their own +-R neighbourhood): rare; global atomics, after the owner pass
template <bool PRE>
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
            const float u = a.imnet_out[((long)db * 64 + c) * Q + p] + a.feat_lr[((long)db * 64 + c) * HWl + lr];
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

template <bool PRE>
static int launch_motif_splat(MotifSplatArgs a, void* stream) {
    const int cap = 2 * (OT_H + 2 * a.R) * (OT_W + 2 * a.R);
    const size_t lds = (size_t)OT_CC * OT_TP * 8 + (size_t)3 * OT_TP * 4 + (size_t)cap * 4 + 192 * 4;
    hipError_t e = hipFuncSetAttribute((const void*)splat_owner_kernel<PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    dim3 grid(((a.WW + OT_W - 1) / OT_W) * ((a.HH + OT_H - 1) / OT_H), 1, a.B * a.N);
    splat_owner_kernel<PRE><<<grid, OT_THREADS, lds, (hipStream_t)stream>>>(a, cap);
    MOTIF_LAUNCH_CHECK();
    const int tiles_x = (a.WW + 63) / 64, tiles_y = (a.HH + 3) / 4;
    dim3 grid2(tiles_x * tiles_y, 1, 2 * a.B * a.N);
    splat_far_kernel<PRE><<<grid2, 256, 0, (hipStream_t)stream>>>(a, tiles_x);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

extern "C" int motif_splat_motif_acc_fwd(const float* imnet_out, const float* pred, const float* feat_lr,
                                         const int32_t* iy, const int32_t* ix, const float* alpha, float flow_scale,
                                         float* acc, int B, int N, int H, int W, int HH, int WW, int row0, int accumulate, void* stream) {
    if (!imnet_out || !pred || !feat_lr || !iy || !ix || !alpha || !acc) return MOTIF_EINVAL;
    if (B < 1 || N < 1 || H < 1 || W < 1 || HH < 1 || WW < 1 || row0 < 0) return MOTIF_EINVAL;
    MotifSplatArgs a{imnet_out, pred, feat_lr, nullptr, iy, ix, alpha, 20.0f, flow_scale, acc, B, N, H, W, HH, WW, 16, row0, accumulate ? 1 : 0};
    return launch_motif_splat<false>(a, stream);
}

extern "C" int motif_splat_motif_pre_fwd(const float* u_hr, const float* pred, const float* g_lr, const float* ab,
                                         const int32_t* iy, const int32_t* ix, const float* alpha, float flow_scale,
                                         float* acc, int B, int N, int H, int W, int HH, int WW, int row0, int accumulate, void* stream) {
    if (!u_hr || !pred || !g_lr || !ab || !iy || !ix || !alpha || !acc) return MOTIF_EINVAL;
    if (B < 1 || N < 1 || H < 1 || W < 1 || HH < 1 || WW < 1 || row0 < 0) return MOTIF_EINVAL;
    MotifSplatArgs a{u_hr, pred, g_lr, ab, iy, ix, alpha, 20.0f, flow_scale, acc, B, N, H, W, HH, WW, 16, row0, accumulate ? 1 : 0};
    return launch_motif_splat<true>(a, stream);
}

extern "C" int motif_splat_motif_fwd(const float* imnet_out, const float* pred, const float* feat_lr,
                                     const int32_t* iy, const int32_t* ix, const float* alpha, float flow_scale,
                                     float* acc, int B, int N, int H, int W, int HH, int WW, int row0, void* stream) {
    return motif_splat_motif_acc_fwd(imnet_out, pred, feat_lr, iy, ix, alpha, flow_scale, acc, B, N, H, W, HH, WW, row0, 0, stream);
}

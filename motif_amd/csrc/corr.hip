// Correlation kernels of the flow extractors (gfx950).
#include "common.h"
#include <stdlib.h>

// ------------------------------------------------------------------ RAFT windowed bilinear lookup
// One wave per query pixel: the 64 lanes are the (2r+2)^2 = 8x8 integer neighbourhood of floor(coords)
// (r = 3, RAFT-small), each lane owns one 128-channel dot product against the channels-last f2 (float4
// loads, f1 broadcast).  The (2r+1)^2 = 49 window values are the bilinear blends of neighbouring lanes
// (three ds_bpermute shuffles), staged per block in LDS as [49][64 queries] and written with coalesced
// 256-byte rows into the [B,196,H,W] correlation tensor the update block consumes.
struct LookupArgs {
    const float* fmap1; const float* fmap2[4]; const float* coords;
    float coord_scale[4]; float* out;
    int H1, W1, H2[4], W2[4], C, out_C, ch_off[4];
    float div;
    int idx1[16], idx2[16];   // pair b reads fmap1[idx1[b]] and fmap2[.][idx2[b]] (identity unless the caller pairs maps of a shared stack)
};

// grid: x = 64-query tiles, y = batch, z = pyramid level (all levels of AlternateCorrBlock in one launch)
// Dot products: 16 lanes share one neighbour (8 channels = 32 bytes each, so a load instruction reads four half rows of
// 256 contiguous bytes instead of 64 scattered 16-byte pieces), partial sums are reduced inside the 16-lane DPP row and
// parked in LDS so that lane n ends up with the dot of neighbour n.  C = 128 only (RAFT-small fnet); other widths use
// the one-lane-per-neighbour form.
__device__ __forceinline__ float row16_sum(float v) {         // sum over the 16 lanes of a DPP row, valid in lane 15
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));   // row_shr:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));   // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));   // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));   // row_shr:1
    return v;
}

__global__ __launch_bounds__(256) void raft_lookup_kernel(LookupArgs a) {
    __shared__ float tile[49][65];
    __shared__ float dots[4][64];
    const int b = blockIdx.y, lv = blockIdx.z;
    const int b1 = a.idx1[b & 15], b2 = a.idx2[b & 15];
    const float* __restrict__ fmap2 = a.fmap2[lv];
    const int H2 = a.H2[lv], W2 = a.W2[lv], C = a.C;
    const float cs = a.coord_scale[lv];
    const long HW1 = (long)a.H1 * a.W1;
    // XCD-aware block order: the 64-pixel blocks that land on one XCD cover a contiguous band of rows, so that XCD's L2 holds the
    // matching band of fmap2 (+-3 rows and the flow) instead of the whole map (measured: 555 MB of HBM-side fetches per call before)
    const long q0 = (long)xcd_tile_id() * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gy = lane >> 3, gx = lane & 7;
    const int sub = lane & 15, grp = lane >> 4;
    for (int i = 0; i < 16; ++i) {
        // the four waves work on four ADJACENT pixels at a time: their 8x8 windows overlap, so the lines one wave pulls into the L1 serve
        // the others (16 pixels apart they shared nothing and each window is the size of the L1)
        const int ql = i * 4 + wave;
        const long q = q0 + ql;
        if (q >= HW1) break;                                    // wave-uniform
        const float x = a.coords[((long)b * 2) * HW1 + q] * cs;
        const float y = a.coords[((long)b * 2 + 1) * HW1 + q] * cs;
        const float fx = floorf(x), fy = floorf(y);
        const float dx = x - fx, dy = y - fy;
        float s = 0.f;
        if (C == 128) {
            const f32x4* f1 = (const f32x4*)(a.fmap1 + ((long)b1 * HW1 + q) * 128) + sub * 2;
            const f32x4 u0 = f1[0], u1 = f1[1];
            // the 16 neighbour rows of this 16-lane group in two batches of 8: UNCONDITIONAL loads from the clamped position
            // (out-of-range neighbours are zeroed afterwards), all of a batch in flight together -- predicated loads put each
            // one in its own exec-mask region with a full memory round trip per neighbour
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                f32x4 v0[8], v1[8];
                bool okn[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int nb = (hb * 8 + i) * 4 + grp;                  // neighbour handled by this 16-lane row
                    const int h2 = (int)fy - 3 + (nb >> 3), w2 = (int)fx - 3 + (nb & 7);
                    okn[i] = h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2;
                    const int hc = h2 < 0 ? 0 : (h2 > H2 - 1 ? H2 - 1 : h2), wc = w2 < 0 ? 0 : (w2 > W2 - 1 ? W2 - 1 : w2);
                    const f32x4* f2 = (const f32x4*)(fmap2 + (((long)b2 * H2 + hc) * W2 + wc) * 128) + sub * 2;
                    v0[i] = f2[0]; v1[i] = f2[1];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int nb = (hb * 8 + i) * 4 + grp;
                    float s0 = u0[0] * v0[i][0], s1 = u0[1] * v0[i][1], s2 = u0[2] * v0[i][2], s3 = u0[3] * v0[i][3];
                    s0 = fmaf(u1[0], v1[i][0], s0); s1 = fmaf(u1[1], v1[i][1], s1); s2 = fmaf(u1[2], v1[i][2], s2); s3 = fmaf(u1[3], v1[i][3], s3);
                    float part = okn[i] ? (s0 + s1) + (s2 + s3) : 0.f;
                    part = row16_sum(part);
                    if (sub == 15) dots[wave][nb] = part;
                }
            }
            s = dots[wave][lane];                                       // same wave wrote it: LDS ops of a wave are ordered
        } else {
            const int h2 = (int)fy - 3 + gy, w2 = (int)fx - 3 + gx;
            if (h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2) {
                const f32x4* f1 = (const f32x4*)(a.fmap1 + ((long)b1 * HW1 + q) * C);
                const f32x4* f2 = (const f32x4*)(fmap2 + (((long)b2 * H2 + h2) * W2 + w2) * C);
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
                for (int c = 0; c < C / 4; ++c) {
                    const f32x4 u = f1[c], v = f2[c];
                    s0 = fmaf(u[0], v[0], s0); s1 = fmaf(u[1], v[1], s1); s2 = fmaf(u[2], v[2], s2); s3 = fmaf(u[3], v[3], s3);
                }
                s = (s0 + s1) + (s2 + s3);
            }
        }
        const float s_e = __shfl(s, (lane + 1) & 63), s_s = __shfl(s, (lane + 8) & 63), s_se = __shfl(s, (lane + 9) & 63);
        if (gy < 7 && gx < 7) {
            float v = s * (1.f - dy) * (1.f - dx);
            v += s_e * (1.f - dy) * dx;
            v += s_s * dy * (1.f - dx);
            v += s_se * dy * dx;
            tile[gx * 7 + gy][ql] = v / a.div;                  // channel = ix*(2r+1) + iy
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 49 * 64; idx += 256) {
        const int ch = idx >> 6, ql = idx & 63;
        const long q = q0 + ql;
        if (q < HW1) a.out[((long)b * a.out_C + a.ch_off[lv] + ch) * HW1 + q] = tile[ch][ql];
    }
}

extern "C" int motif_raft_corr_lookup_pyramid(const float* fmap1, const float* const* fmap2, const int* H2, const int* W2, int levels,
                                              const float* coords, float* out, int B, int H1, int W1, int C, int r,
                                              int out_C, float div, const int32_t* index1_host, const int32_t* index2_host, void* stream) {
    if (!fmap1 || !fmap2 || !coords || !out || !H2 || !W2 || B < 1 || levels < 1 || levels > 4) return MOTIF_EINVAL;
    if (r != 3 || (C & 3)) return MOTIF_ELIMIT;
    // the kernel's batch index maps hold 16 pairs: larger batches go in slices (pair b of a slice = pair b0 + b of the call; the map
    // entries are absolute indices into the feature stacks)
    for (int b0 = 0; b0 < B; b0 += 16) {
        const int nb = B - b0 < 16 ? B - b0 : 16;
        LookupArgs a;
        for (int i = 0; i < 16; ++i) {
            const int bi = b0 + (i < nb ? i : 0);
            a.idx1[i] = index1_host ? index1_host[bi] : bi;
            a.idx2[i] = index2_host ? index2_host[bi] : bi;
            if (a.idx1[i] < 0 || a.idx2[i] < 0) return MOTIF_EINVAL;
        }
        a.fmap1 = fmap1; a.coords = coords + (long)b0 * 2 * H1 * W1; a.out = out + (long)b0 * out_C * H1 * W1;
        a.H1 = H1; a.W1 = W1; a.C = C; a.out_C = out_C; a.div = div;
        for (int i = 0; i < 4; ++i) {
            const int j = i < levels ? i : 0;
            if (!fmap2[j]) return MOTIF_EINVAL;
            a.fmap2[i] = fmap2[j]; a.H2[i] = H2[j]; a.W2[i] = W2[j];
            a.coord_scale[i] = 1.0f / (float)(1 << j);              // coords / 2**i (corr.py:81)
            a.ch_off[i] = j * (2 * r + 1) * (2 * r + 1);
        }
        dim3 grid(cdiv((long)H1 * W1, 64), nb, levels);
        raft_lookup_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
        MOTIF_LAUNCH_CHECK();
    }
    return MOTIF_OK;
}

extern "C" int motif_raft_corr_lookup(const float* fmap1, const float* fmap2, const float* coords, float coord_scale,
                                      float* out, int B, int H1, int W1, int H2, int W2, int C, int r,
                                      int out_C, int ch_off, float div, void* stream) {
    if (!fmap1 || !fmap2 || !coords || !out || B < 1) return MOTIF_EINVAL;
    if (r != 3 || (C & 3)) return MOTIF_ELIMIT;
    if (B > 16) {                                        // the kernel's batch index maps hold 16 entries: larger batches in slices
        for (int b0 = 0; b0 < B; b0 += 16) {
            const int nb = B - b0 < 16 ? B - b0 : 16;
            const int rc = motif_raft_corr_lookup(fmap1 + (long)b0 * H1 * W1 * C, fmap2 + (long)b0 * H2 * W2 * C, coords + (long)b0 * 2 * H1 * W1, coord_scale,
                                                  out + (long)b0 * out_C * H1 * W1, nb, H1, W1, H2, W2, C, r, out_C, ch_off, div, stream);
            if (rc != MOTIF_OK) return rc;
        }
        return MOTIF_OK;
    }
    LookupArgs a;
    for (int i = 0; i < 16; ++i) a.idx1[i] = a.idx2[i] = i;
    a.fmap1 = fmap1; a.coords = coords; a.out = out; a.H1 = H1; a.W1 = W1; a.C = C; a.out_C = out_C; a.div = div;
    for (int i = 0; i < 4; ++i) { a.fmap2[i] = fmap2; a.H2[i] = H2; a.W2[i] = W2; a.coord_scale[i] = coord_scale; a.ch_off[i] = ch_off; }
    dim3 grid(cdiv((long)H1 * W1, 64), B, 1);
    raft_lookup_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// ------------------------------------------------------------------ PWC-Net 9x9 cost volume
// Two kernels (correlation.py:44-112 runs one 32-thread block per PIXEL with a serial reduce):
//  * corr81_tiled_kernel, maps of >= 64x96 pixels: LDS-tiled and register-blocked.  A block owns 8 rows x 32 columns of
//    output pixels.  Channels go by in chunks of 8: the chunk's f2 window (tile + 4 px halo = 16 x 40, zero outside the
//    image = the reference's zero padding, correlation.py:17-42) is staged in LDS with coalesced row loads.  A thread owns
//    4 horizontally adjacent pixels and 3 of the 9 vertical displacements (wave g = rows dy 3g..3g+2): per (channel, dy)
//    it reads the 12 window values its 4 pixels x 9 horizontal displacements need as three aligned ds_read_b128 and does
//    36 FMAs into 108 register accumulators -- 1 LDS instruction per 12 FMAs; f1 is read from HBM once (as float4s), f2
//    1.56x (halo) instead of 81 times each through L1/L2.
//  * corr81_small_kernel, the coarse pyramid levels (a few hundred pixels, up to 196 channels): one thread per
//    (displacement, pixel) -- there the 81-fold parallelism matters more than the reuse, everything is L2 resident.
// Channel order of the sum: ascending in both.
#define C81_TH 8
#define C81_TW 32
#define C81_WS (C81_TW + 8)
#define C81_WH (C81_TH + 8)
#define C81_CK 8
// DYW = vertical displacements per wave: 3 (three waves per block: the large maps, whose tiles fill the chip) or 1 (nine waves per block:
// the middle pyramid levels, 60-240 tiles -- three times the waves for the same window, a third of the FMA chain per wave).
template <int DYW>
__global__ __launch_bounds__(64 * (9 / DYW)) void corr81_tiled_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                                       float* __restrict__ out, int C, int H, int W, int act) {
    constexpr int C81_NT = 64 * (9 / DYW);
    __shared__ __attribute__((aligned(16))) float win[C81_CK][C81_WH * C81_WS];
    const int tid = threadIdx.x, g = tid >> 6, q = tid & 63, qy = q >> 3, qx = q & 7;
    const int x0 = blockIdx.x * C81_TW, y0 = blockIdx.y * C81_TH, b = blockIdx.z;
    const int x = x0 + 4 * qx, y = y0 + qy;                                   // first of this thread's 4 pixels
    const long HW = (long)H * W;
    const bool row_ok = y < H;
    const bool vec = (W & 3) == 0;                                             // block-uniform: float4 global access
    const float* f1b = f1 + (long)b * C * HW;
    const float* f2b = f2 + (long)b * C * HW;
    float acc[DYW][9][4];
#pragma unroll
    for (int i = 0; i < DYW; ++i)
#pragma unroll
        for (int d = 0; d < 9; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][d][j] = 0.f;
    // software pipeline: the f1 values and the f2 window of chunk c0 + 8 are requested (into registers) before chunk c0 is
    // multiplied, so HBM / L2 latency hides behind the FMAs even with one block per CU
    constexpr int NST = (C81_CK * C81_WH * C81_WS + C81_NT - 1) / C81_NT;
    float st[NST], an[C81_CK][4];
    auto request = [&](int c0) {
#pragma unroll
        for (int c = 0; c < C81_CK; ++c) {
            const bool cok = row_ok && c0 + c < C;
            const float* p = f1b + (long)(cok ? c0 + c : 0) * HW + (long)(row_ok ? y : 0) * W;
            if (vec && cok && x + 3 < W) {
                const f32x4 v = *(const f32x4*)(p + x);
                an[c][0] = v[0]; an[c][1] = v[1]; an[c][2] = v[2]; an[c][3] = v[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) an[c][j] = (cok && x + j < W) ? p[x + j] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = tid + C81_NT * k;
            const int c = e / (C81_WH * C81_WS), r = e - c * (C81_WH * C81_WS);
            const int wy = r / C81_WS, wx = r - wy * C81_WS;
            const int gy = y0 - 4 + wy, gx = x0 - 4 + wx;
            const bool okv = e < C81_CK * C81_WH * C81_WS && c0 + c < C && gy >= 0 && gy < H && gx >= 0 && gx < W;
            st[k] = okv ? f2b[(long)(c0 + c) * HW + (long)gy * W + gx] : 0.f;
        }
    };
    request(0);
    for (int c0 = 0; c0 < C; c0 += C81_CK) {
        float a[C81_CK][4];
        __syncthreads();                                   // previous chunk's window fully consumed
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = tid + C81_NT * k;
            if (e < C81_CK * C81_WH * C81_WS) (&win[0][0])[e] = st[k];
        }
#pragma unroll
        for (int c = 0; c < C81_CK; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[c][j] = an[c][j];
        __syncthreads();
        if (c0 + C81_CK < C) request(c0 + C81_CK);
        const float* wp = &win[0][(qy + DYW * g) * C81_WS + 4 * qx];
#pragma unroll
        for (int c = 0; c < C81_CK; ++c)
#pragma unroll
            for (int i = 0; i < DYW; ++i) {
                float w[12];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const f32x4 v = *(const f32x4*)(wp + c * (C81_WH * C81_WS) + i * C81_WS + 4 * k);
                    w[4 * k] = v[0]; w[4 * k + 1] = v[1]; w[4 * k + 2] = v[2]; w[4 * k + 3] = v[3];
                }
#pragma unroll
                for (int d = 0; d < 9; ++d)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][d][j] = fmaf(a[c][j], w[j + d], acc[i][d][j]);
            }
    }
    if (!row_ok || x >= W) return;
    const float inv = (float)C;
    float* ob = out + (long)b * 81 * HW + (long)y * W + x;
#pragma unroll
    for (int i = 0; i < DYW; ++i)
#pragma unroll
        for (int d = 0; d < 9; ++d) {
            float* op = ob + (long)((DYW * g + i) * 9 + d) * HW;
            if (vec && x + 3 < W) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = act_apply(acc[i][d][j] / inv, act);
                *(f32x4*)op = v;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (x + j < W) op[j] = act_apply(acc[i][d][j] / inv, act);
            }
        }
}

__global__ void corr81_small_kernel(const float* __restrict__ f1, const float* __restrict__ f2, float* __restrict__ out,
                                    int C, int H, int W, int act) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    const int b = blockIdx.z / 81, d = blockIdx.z % 81;
    if (x >= W) return;
    const int dy = d / 9 - 4, dx = d % 9 - 4;
    const int y2 = y + dy, x2 = x + dx;
    const long HW = (long)H * W;
    float s = 0.f;
    if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) {
        const float* a = f1 + (long)b * C * HW + (long)y * W + x;
        const float* v = f2 + (long)b * C * HW + (long)y2 * W + x2;
        // eight channels' loads in flight, then the same ascending-order FMA chain (one load pair per FMA was 57 us of pure latency for
        // the 196 channels of the coarsest level)
        int c = 0;
        for (; c + 8 <= C; c += 8) {
            float av[8], vv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { av[e] = a[(long)(c + e) * HW]; vv[e] = v[(long)(c + e) * HW]; }
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(av[e], vv[e], s);
        }
        for (; c < C; ++c) s = fmaf(a[(long)c * HW], v[(long)c * HW], s);
    }
    out[((long)b * 81 + d) * HW + (long)y * W + x] = act_apply(s / (float)C, act);
}

extern "C" int motif_corr81_fwd(const float* first, const float* second, float* out, int B, int C, int H, int W,
                                int act, void* stream) {
    if (!first || !second || !out || B < 1 || C < 1) return MOTIF_EINVAL;
    const int force = motif_opt(MOTIF_OPT_CORR81);           // 1 = tiled, 2 = small: tests and tools/pwc_bench.py
    const bool tiled = force ? force != 2 : (long)H * W >= 64L * 96;
    if (tiled) {
        dim3 grid(cdiv(W, C81_TW), cdiv(H, C81_TH), B);
        // (DYW = 1, nine waves per block, for the 60-240-tile levels: measured SLOWER, 79 -> 102 and 52 -> 74 us -- the window staging and
        // its two barriers per chunk do not shrink with the FMA chain; option corr81 = 3 keeps it reachable for tests)
        if (force == 3) corr81_tiled_kernel<1><<<grid, 576, 0, (hipStream_t)stream>>>(first, second, out, C, H, W, act);
        else corr81_tiled_kernel<3><<<grid, 192, 0, (hipStream_t)stream>>>(first, second, out, C, H, W, act);
    } else {
        dim3 grid(cdiv(W, 64), H, B * 81);
        corr81_small_kernel<<<grid, 64, 0, (hipStream_t)stream>>>(first, second, out, C, H, W, act);
    }
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}


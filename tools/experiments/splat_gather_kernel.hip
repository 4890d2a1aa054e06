// EXPERIMENT (round 2, not built): a gather formulation of the fused soft-splat.  Every accumulator cell belongs to one
// thread that sums its (source, corner) pairs in fp32 registers in a fixed order -- no LDS atomics, no fixed point.
// Correct (it passed every splat test of tests/test_kernels_gpu.py, incl. sinks, overfull tiles, row bands) but NOT
// faster than the fixed-point scatter tiles of motif_amd/csrc/splat.hip on MI355X:
//   bench.py c2, splat stage per clip: fixed-point tiles 2.84 ms; gather 16x64 tiles / 1024 threads 3.06 ms (128-VGPR cap:
//   the pipelined version spills, 5.7 ms); gather 16x32 tiles / 512 threads / 1 block per CU 3.17 ms.
//   s_memtime phase trace of a 16x32 tile (68 us): compaction 11 %, entries 3 %, counting sort 3 %, pair sort + weights
//   11 %, plane passes 73 % = waiting for the staged global loads 33 % + issuing them 21 % + barrier 21 % + gather 16 % +
//   stores 8 %.  The instruction count is ~3x below the fixed-point kernel's, but with 8 waves per CU in lock step nothing
//   hides the load latency; the fixed-point kernel hides the same loads behind its VALU work with 16 waves.
// What would be needed: <= 128 VGPRs at 1024 threads (packed pair indices, no second operand stream -- fold the LR term G
// into the HR plane U in the imnet kernel's epilogue, which alone is worth -4 % on the fixed-point tiles and -10 % here).
// Kept for the next attempt; splice between splat_far_kernel and launch_motif_splat to build it.
// ---------------------------------------------------------------- fused MoTIF form: owner-computes, GATHER per cell
// Same ownership and the same compaction of the contributing sources as splat_owner_kernel, but the accumulation is
// turned inside out once more: every accumulator cell of the tile belongs to ONE thread, which walks the (source,
// corner) pairs that land on it and sums them in registers in plain fp32 -- no LDS atomics, no fixed point (the
// fixed-point path spends 7 VALU operations and one ds_add_u64 per addend; here an addend is one multiply and one add).
//   1. compaction in SCAN ORDER (ballot counts per 64-source segment, one prefix sum): entry index == rank of the
//      source in (direction, row, column) order, run to run and for every tile / row-band decomposition;
//   2. counting sort of the (entry, corner) pairs by target cell (LDS counters, block prefix sum, fill), then each
//      cell sorts its own short segment by entry index, so the fp32 summation order of a cell is fixed: results are
//      bit-reproducible and independent of the tiling (cells hit by more than 32 pairs -- a sink in the flow field --
//      are rank-sorted by the whole block);
//   3. per pass of GT_C planes: the entry threads stage value * e^z of their sources in LDS (coalesced global reads,
//      one 16-byte LDS write), barrier, the cell threads gather (one ds_read_b128 per pair; entry indices and
//      bilinear weights of the first GT_K pairs stay in registers), write the finished cells with coalesced stores.
// Each addend is (value * e^z) rounded, times the weight, rounded, then added -- softsplat_cp.py:35-50 literally.
// A tile holds 3 sources per cell (the smooth-flow load is 2.2); the sources beyond that are scattered with global
// atomics by the same block after its stores (order-dependent like the reference, never seen on real flows).
#define GT_H 16
#define GT_K 12
#ifndef GT_C
#define GT_C 8
#endif
typedef float svec __attribute__((ext_vector_type(GT_C)));

#ifdef MOTIF_SPLAT_TRACE
__device__ long long g_splat_trace[2048 * 8];
#define STRACE_T() __builtin_amdgcn_s_memtime()
#define STRACE(slot) do { if (threadIdx.x == 0) { const int b_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); if (b_ < 1024) g_splat_trace[b_ * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } } while (0)
extern "C" int motif_debug_splat_trace(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_splat_trace), sizeof(long long) * n); }
#else
#define STRACE(slot)
#define STRACE_T() 0ll
#endif

__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// weight of corner (bit 0: east, bit 1: south) of a source that lands at (ox, oy): the expressions of softsplat_cp.py:35-38
__device__ __forceinline__ float corner_weight(float ox, float oy, int corner) {
    const float flx = floorf(ox), fly = floorf(oy);
    const float wx = (corner & 1) ? (ox - flx) : ((flx + 1.f) - ox);
    const float wy = (corner & 2) ? (oy - fly) : ((fly + 1.f) - oy);
    return wx * wy;
}

template <bool PRE, int TW>
__global__ __launch_bounds__(GT_H * TW) void splat_gather_kernel(MotifSplatArgs a) {
    constexpr int NT = GT_H * TW, NW = NT / 64, EPT = 3, CAPE = EPT * NT;
    constexpr int SEGMAX = 2 * (GT_H + 32) * ((TW + 32 + 63) / 64);
    constexpr int NPL = PRE ? 64 : 130, NSUM = NPL + 1, NCHUNK = (NSUM + GT_C - 1) / GT_C;
    static_assert(SEGMAX <= 192, "segment scan is one wave x 3");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    svec* S = (svec*)lds;                                          // [2][CAPE + 1] staged value * e^z of GT_C planes (double buffer), slot CAPE = 0
    unsigned* list = (unsigned*)(S + CAPE + 1);                    // [CAPE] source keys d<<16 | row<<8 | column: dead before S[1] is used
    float* eox = (float*)(S + 2 * (CAPE + 1));                     // [CAPE] target position of the entry
    float* eoy = eox + CAPE;
    unsigned* cstart = (unsigned*)(eoy + CAPE);                    // [NT] first pair of the cell
    unsigned* ccur = cstart + NT;                                  // [NT] pair counter / fill cursor
    unsigned* segbase = ccur + NT;                                 // [SEGMAX + 1]
    unsigned* wsum = segbase + SEGMAX + 1;                         // [NW]
    unsigned* misc = wsum + NW;                                    // [2] total entries, big cells
    unsigned short* contrib = (unsigned short*)(misc + 2);         // [4 * CAPE] entry << 2 | corner, grouped by cell
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bn = blockIdx.z, b = bn / a.N, n = bn % a.N;
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * GT_H;
    const long Q = (long)a.HH * a.WW, HWl = (long)a.H * a.W;
    ccur[tid] = 0u;
    if (tid < 2) { misc[tid] = 0u; S[tid * (CAPE + 1) + CAPE] = svec(0.f); }
    STRACE(0);

    // ---- compaction in scan order
    const int ry0 = max(ty0 - a.R, 0), ry1 = min(ty0 + GT_H - 1 + a.R, a.HH - 1);
    const int rx0 = max(tx0 - a.R, 0), rx1 = min(tx0 + TW - 1 + a.R, a.WW - 1);
    const int RH = ry1 - ry0 + 1, RW = rx1 - rx0 + 1;
    const int xiters = (RW + 63) >> 6, nseg = 2 * RH * xiters;
    const unsigned long long lt = (1ull << lane) - 1ull;
    constexpr int RSTEPS = (GT_H + 32 + NW - 1) / NW, XIT = (TW + 32 + 63) / 64, NJ = RSTEPS * 2 * XIT;
    unsigned hitbits = 0;
    // slot j = (row step, direction, 64-column segment): compile-time loop, loads unconditional from clamped
    // coordinates (a predicated load is its own exec-mask region: one full memory round trip per slot)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = j / (2 * XIT), d = (j / XIT) & 1, it = j % XIT;
        const int row = wave + NW * r, xc = it * 64 + lane;
        const bool valid = row < RH && xc < RW;
        const SrcGeom g = src_geom(a, (d * a.B + b) * a.N + n, rx0 + min(xc, RW - 1), ry0 + min(row, RH - 1), false);
        const bool hit = valid && g.near_ && g.x0 >= tx0 - 1 && g.x0 <= tx0 + TW - 1 && g.y0 >= ty0 - 1 && g.y0 <= ty0 + GT_H - 1;
        const unsigned long long m = __ballot(hit);
        if (lane == 0 && row < RH && it < xiters) segbase[(d * RH + row) * xiters + it] = (unsigned)__popcll(m);
        if (hit) hitbits |= 1u << j;
    }
    __syncthreads();
    if (wave == 0) {
        unsigned v[3], sum = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) { const int sid = lane * 3 + i; v[i] = sid < nseg ? segbase[sid] : 0u; sum += v[i]; }
        const unsigned incl = wave_incl_scan(sum, lane);
        unsigned run = incl - sum;
#pragma unroll
        for (int i = 0; i < 3; ++i) { const int sid = lane * 3 + i; if (sid < nseg) segbase[sid] = run; run += v[i]; }
        if (lane == 63) misc[0] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = j / (2 * XIT), d = (j / XIT) & 1, it = j % XIT;
        const int row = wave + NW * r;
        const bool hit = (hitbits >> j) & 1u;
        const unsigned long long m = __ballot(hit);
        if (hit) {
            const unsigned idx = segbase[(d * RH + row) * xiters + it] + (unsigned)__popcll(m & lt);
            if (idx < (unsigned)CAPE) list[idx] = ((unsigned)d << 16) | ((unsigned)row << 8) | (unsigned)(it * 64 + lane);
        }
    }
    __syncthreads();
    const unsigned total = misc[0];
    const int cnt = total < (unsigned)CAPE ? (int)total : CAPE;
    STRACE(1);

    // ---- the entries of this thread: geometry once, kept in registers (loads unconditional: an absent entry reads
    // source (rx0, ry0) of direction 0 and is masked afterwards)
    bool ev[EPT];
    unsigned uo[EPT], go[EPT];                                     // element offsets into imnet_out / feat_lr (plane 0 of the entry's direction)
    float ee[EPT], ep0[EPT], ep1[EPT];
    int ecx[EPT], ecy[EPT];
    {
        unsigned ent[EPT];
#pragma unroll
        for (int j = 0; j < EPT; ++j) { ev[j] = tid + NT * j < cnt; ent[j] = list[tid + NT * j]; }
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            if (!ev[j]) ent[j] = 0u;
            const int d = ent[j] >> 16, y = ry0 + ((ent[j] >> 8) & 255), x = rx0 + (ent[j] & 255);
            const int db = d * a.B + b;
            const SrcGeom g = src_geom(a, db * a.N + n, x, y, true);
            ee[j] = g.e; ep0[j] = g.p0; ep1[j] = g.p1;
            uo[j] = (unsigned)((long)db * 64 * Q + ((long)y * a.WW + x));
            go[j] = (unsigned)((long)db * 64 * HWl + ((long)a.iy[y] * a.W + a.ix[x]));
            ecx[j] = g.x0 - tx0; ecy[j] = g.y0 - ty0;
            if (ev[j]) {
                eox[tid + NT * j] = g.ox; eoy[tid + NT * j] = g.oy;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ux = ecx[j] + (c & 1), uy = ecy[j] + (c >> 1);
                    if (ux >= 0 && ux < TW && uy >= 0 && uy < GT_H) atomicAdd(&ccur[uy * TW + ux], 1u);
                }
            }
        }
    }
    __syncthreads();
    STRACE(2);
    // ---- counting sort of the pairs by cell
    const int npair = (int)ccur[tid];
    {
        const unsigned incl = wave_incl_scan((unsigned)npair, lane);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned wp = 0;
        for (int i = 0; i < wave; ++i) wp += wsum[i];
        cstart[tid] = wp + incl - (unsigned)npair;
        ccur[tid] = 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        unsigned cs[4];
        bool ok[4];
        int cell[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ux = ecx[j] + (c & 1), uy = ecy[j] + (c >> 1);
            ok[c] = ev[j] && ux >= 0 && ux < TW && uy >= 0 && uy < GT_H;
            cell[c] = ok[c] ? uy * TW + ux : 0;
            cs[c] = cstart[cell[c]];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (ok[c]) contrib[cs[c] + atomicAdd(&ccur[cell[c]], 1u)] = (unsigned short)(((tid + NT * j) << 2) | c);
    }
    __syncthreads();
    STRACE(3);
    // ---- fixed summation order: each cell sorts its pairs by (entry, corner).  Up to GT_K pairs: in registers, by an
    // odd-even transposition network (branch-free); up to 32: insertion sort in LDS; more: rank sort by the whole block.
    unsigned short* seg = contrib + cstart[tid];
    unsigned key[GT_K];
#pragma unroll
    for (int k = 0; k < GT_K; ++k) key[k] = seg[k];                  // reads past the segment stay inside the (padded) array
    if (npair > 32) {
        ((unsigned*)S)[atomicAdd(&misc[1], 1u)] = (unsigned)tid;
    } else if (npair > GT_K) {
        for (int i = 1; i < npair; ++i) {
            const unsigned short v = seg[i];
            int k = i - 1;
            while (k >= 0 && seg[k] > v) { seg[k + 1] = seg[k]; --k; }
            seg[k + 1] = v;
        }
    }
    __syncthreads();
    {
        const int nbig = (int)misc[1];
        unsigned short* tmp = (unsigned short*)((unsigned*)S + NT);
        for (int q = 0; q < nbig; ++q) {                               // block-uniform; rank sort (all keys distinct)
            const int cell = (int)((unsigned*)S)[q];
            const int m = (int)ccur[cell];
            const unsigned short* src = contrib + cstart[cell];
            for (int i = tid; i < m; i += NT) {
                const unsigned short v = src[i];
                int r = 0;
                for (int k = 0; k < m; ++k) r += src[k] < v ? 1 : 0;
                tmp[r] = v;
            }
            __syncthreads();
            for (int i = tid; i < m; i += NT) contrib[cstart[cell] + i] = tmp[i];
            __syncthreads();
        }
    }
    if (npair > GT_K) {
#pragma unroll
        for (int k = 0; k < GT_K; ++k) key[k] = seg[k];
    } else {
#pragma unroll
        for (int k = 0; k < GT_K; ++k) if (k >= npair) key[k] = 0xffffu;
#pragma unroll
        for (int round = 0; round < GT_K; ++round)
#pragma unroll
            for (int k = round & 1; k + 1 < GT_K; k += 2) {
                const unsigned lo = min(key[k], key[k + 1]), hi = max(key[k], key[k + 1]);
                key[k] = lo; key[k + 1] = hi;
            }
    }
    int pidx[GT_K];                                                    // absent pairs: the zero slot S[CAPE], weight 0
    float pw[GT_K];
    {
        float kx[GT_K], ky[GT_K];
#pragma unroll
        for (int k = 0; k < GT_K; ++k) {
            const int e = (k < npair) ? (int)(key[k] >> 2) : 0;
            kx[k] = eox[e]; ky[k] = eoy[e];
        }
#pragma unroll
        for (int k = 0; k < GT_K; ++k) {
            const bool ok = k < npair;
            pidx[k] = ok ? (int)(key[k] >> 2) : CAPE;
            pw[k] = ok ? corner_weight(kx[k], ky[k], (int)(key[k] & 3u)) : 0.f;
        }
    }
    __syncthreads();                                                   // S held the big-cell scratch
    STRACE(4);

    // ---- the planes, GT_C at a time.  Software pipeline: the global reads of pass k+2 are in flight and pass k+1 is
    // staged into the other S buffer while pass k is gathered, so a pass costs one barrier and no exposed memory latency.
    float* abase = a.acc + (long)bn * (NPL + 3) * Q;
    const int lx = tid % TW, ly = tid / TW, X = tx0 + lx, Y = ty0 + ly;
    const bool inimg = X < a.WW && Y < a.HH;
    const unsigned ocell = (unsigned)(Y * a.WW + X);
    float mx = 0.f;
    float u[EPT][GT_C], g[EPT][GT_C];
    auto request = [&](int k) {                                        // global -> registers (unconditional: plane / entry 0 when unused)
        const int c0 = k * GT_C;
#pragma unroll
        for (int cc = 0; cc < GT_C; ++cc) {
            const int c = c0 + cc;                                     // uniform plane base + 32-bit per-lane offset
            if constexpr (PRE) {
                const int cl = c < NPL ? c : 0;
                const float* pu = a.imnet_out + (long)cl * Q;
#pragma unroll
                for (int j = 0; j < EPT; ++j) u[j][cc] = pu[uo[j]];
                if (a.feat_lr) {                                       // uniform; null: G is already folded into the HR plane
                    const float* pg = a.feat_lr + (long)cl * HWl;
#pragma unroll
                    for (int j = 0; j < EPT; ++j) g[j][cc] = pg[go[j]];
                } else {
#pragma unroll
                    for (int j = 0; j < EPT; ++j) g[j][cc] = 0.f;
                }
            } else {
                const bool fromu = c < 64, fromg = c >= 66 && c < NPL;
                const float* pp = fromu ? a.imnet_out + (long)c * Q : a.feat_lr + (long)(fromg ? c - 66 : 0) * HWl;
#pragma unroll
                for (int j = 0; j < EPT; ++j) u[j][cc] = pp[fromu ? uo[j] : go[j]];
            }
        }
    };
    auto stage = [&](int k) {                                          // registers -> S[k & 1]
        const int c0 = k * GT_C;
        svec* Sb = S + (k & 1) * (CAPE + 1);
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            svec sv;
#pragma unroll
            for (int cc = 0; cc < GT_C; ++cc) {
                const int c = c0 + cc;
                float v = 0.f;
                if constexpr (PRE) {
                    if (c < NPL) v = fmaf(a.ab[64 + c], ep1[j], fmaf(a.ab[c], ep0[j], u[j][cc] + g[j][cc])) * ee[j];
                    else if (c == NPL) v = ee[j];
                } else {
                    if (c == 64) v = ep0[j] * ee[j];
                    else if (c == 65) v = ep1[j] * ee[j];
                    else if (c < NPL) v = u[j][cc] * ee[j];
                    else if (c == NPL) v = ee[j];
                }
                sv[cc] = v;
            }
            if (ev[j]) Sb[tid + NT * j] = sv;
        }
    };
    request(0);
    stage(0);
    if (NCHUNK > 1) request(1);
    __syncthreads();
    long long tr[5] = {0, 0, 0, 0, 0};
    for (int k = 0; k < NCHUNK; ++k) {
        const int c0 = k * GT_C;
        long long t0 = STRACE_T();
        if (k + 1 < NCHUNK) stage(k + 1);
        long long t1 = STRACE_T(); tr[0] += t1 - t0;
        if (k + 2 < NCHUNK) request(k + 2);
        t0 = STRACE_T(); tr[1] += t0 - t1;
        const svec* Sb = S + (k & 1) * (CAPE + 1);
        float acc[GT_C];
#pragma unroll
        for (int cc = 0; cc < GT_C; ++cc) acc[cc] = 0.f;
        const bool last = (k == NCHUNK - 1);
#pragma unroll
        for (int qb = 0; qb < GT_K; qb += 4) {                       // unconditional reads, four pairs in flight: absent pairs read the zero slot
            svec sv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) sv[q] = Sb[pidx[qb + q]];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int cc = 0; cc < GT_C; ++cc) {
                    const float t = sv[q][cc] * pw[qb + q];
                    acc[cc] = acc[cc] + t;
                    if (cc == (NPL % GT_C) && last) mx = fmaxf(mx, t);
                }
            }
        }
        for (int q = GT_K; q < npair; ++q) {
            const int it = seg[q];
            const float w = corner_weight(eox[it >> 2], eoy[it >> 2], it & 3);
            const svec sv = Sb[it >> 2];
#pragma unroll
            for (int cc = 0; cc < GT_C; ++cc) {
                const float t = sv[cc] * w;
                acc[cc] = acc[cc] + t;
                if (cc == (NPL % GT_C) && last) mx = fmaxf(mx, t);
            }
        }
        t1 = STRACE_T(); tr[2] += t1 - t0;
        if (inimg) {
#pragma unroll
            for (int cc = 0; cc < GT_C; ++cc) {
                if (c0 + cc < NSUM) {
                    float* o = abase + (long)(c0 + cc) * Q + ocell;
                    *o = a.accumulate ? *o + acc[cc] : acc[cc];
                }
            }
        }
        t0 = STRACE_T(); tr[3] += t0 - t1;
        __syncthreads();
        tr[4] += STRACE_T() - t0;
    }
#ifdef MOTIF_SPLAT_TRACE
    if (threadIdx.x == 0) { const int b_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); if (b_ < 1024) for (int i = 0; i < 5; ++i) g_splat_trace[(1024 + b_) * 8 + i] = tr[i]; }
#endif
    STRACE(5);
    if (inimg) {
        float* o = abase + (long)(NPL + 1) * Q + ocell;
        const float vmax = fmaxf(1.0f, mx);                          // max-splat output starts at ones (softsplat_max_cp.py:254)
        *o = a.accumulate ? fmaxf(*o, vmax) : vmax;
        o += Q;
        *o = a.accumulate ? *o + (float)npair : (float)npair;
    }
    // ---- sources the tile could not hold: global atomics on top of the finished tile
    if (total > (unsigned)CAPE) {
        __threadfence();
        __syncthreads();
        for (int j = 0; j < NJ; ++j) {
            const int r = j / (2 * XIT), d = (j / XIT) & 1, it = j % XIT;
            const int row = wave + NW * r;
            const bool hit = (hitbits >> j) & 1u;
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const unsigned idx = segbase[(d * RH + row) * xiters + it] + (unsigned)__popcll(m & lt);
                if (idx >= (unsigned)CAPE) {
                    const int img = (d * a.B + b) * a.N + n, x = rx0 + it * 64 + lane, y = ry0 + row;
                    const long p = (long)y * a.WW + x;
                    scatter_source_global<PRE>(a, img, x, y, a.pred[((long)img * 3) * Q + p], a.pred[((long)img * 3 + 1) * Q + p],
                                               tx0, tx0 + TW - 1, ty0, ty0 + GT_H - 1);
                }
            }
        }
    }
}

template <bool PRE, int TW>
static int launch_gather(const MotifSplatArgs& a, hipStream_t stream) {
    constexpr int NT = GT_H * TW, CAPE = 3 * NT, SEGMAX = 2 * (GT_H + 32) * ((TW + 32 + 63) / 64);
    const size_t lds = (size_t)(CAPE + 1) * 2 * GT_C * 4 + (size_t)CAPE * 8 + (size_t)NT * 8 + (size_t)(SEGMAX + 1 + NT / 64 + 2) * 4 + (size_t)CAPE * 4 * 2 + 64;
    hipError_t e = hipFuncSetAttribute((const void*)splat_gather_kernel<PRE, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    dim3 grid((a.WW + TW - 1) / TW, (a.HH + GT_H - 1) / GT_H, a.B * a.N);
    splat_gather_kernel<PRE, TW><<<grid, NT, lds, stream>>>(a);
    return MOTIF_OK;
}


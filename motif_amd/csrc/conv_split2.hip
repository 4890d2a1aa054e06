// 3x3 / stride 1 convolution on the bf16 matrix cores, fp32-equivalent 3-way split arithmetic (see conv_split.hip): the
// round-3 kernel for the launches that carry the clip's time.  Same packed weights, same ConvArgs, same results policy.
//
// What round 3 measured about conv_split_kernel (tools/trace_pp.py, ablation builds) and what this kernel does
// about it:
//   * The CU's vector-memory path was the contended resource, not the matrix pipe: per 16-channel chunk a 4-wave block issued
//     96 dword staging loads (~45 cycles each through the texture addresser) and 216 16-byte weight-fragment loads.
//       -> activations are fetched as 16-byte row pieces (4 pixels of one channel per lane, 5 loads per lane and chunk), parked in
//          a wave-private LDS landing area and split from there; a wave owns 3 rows x 32 couts instead of 2 rows x 64 couts, so the
//          eight waves of a CU read two distinct weight-fragment streams instead of eight copies of one twice as long.
//   * Work placed in ANOTHER wave of the SIMD does not hide under an MFMA stream (a wave whose next instruction is an MFMA
//     waiting for the pipe keeps the issue port: ~1 slot per MFMA for the neighbour, measured with a two-phase "ping-pong" variant
//     of this kernel); work placed in the SAME wave between its MFMAs does.
//       -> one uniform 8-wave workgroup per CU; every wave runs the same statically scheduled stream: after each MFMA at most one
//          operand request and one small staging step (load / park / read / split half-step / store) of the NEXT chunk.
//   * Tile quantisation: 690 tiles of 8 rows on 512 block slots cost the trunk launches a third of their time.
//       -> 12-row tiles (450 tiles for the trunk: 24 rows on the busiest CU instead of 32), persistent workgroups (tile = b' + i*G,
//          XCD-aware b'), the next tile's first chunk staged under the current tile's last one.
// Layout in LDS: bf16 staging [2 buffers][part][octet][py][px] x 8 channels (one ds_read_b128 = one B fragment, as before),
// landing area [8 waves][5 x 64 lanes x 16 bytes] (reused as the wave's epilogue scratch), bias [8 waves][32].
#include "conv_wave_epilogue.h"

#ifdef MOTIF_TRACE
__device__ long long g_s2_trace[1024 * 8 * 32];
#define S2TRACE(slot) do { if (lane == 0 && blockIdx.x < 1024) g_s2_trace[(blockIdx.x * 8 + wave) * 32 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int motif_debug_s2_trace(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_s2_trace), sizeof(long long) * n); }
#else
#define S2TRACE(slot)
#endif

namespace {

// Products ordered by ACTIVATION part, smallest part first (w = weight part, x = activation part): when the last product of
// a part has been issued, its B-fragment registers take the same part of the next tap (one live copy of the B fragments).
struct S2Order {
    static constexpr int n = 6;
    static constexpr int w[6] = {0, 1, 0, 2, 1, 0};
    static constexpr int x[6] = {2, 1, 1, 0, 0, 0};
};

// Static schedule of a chunk: slot s = tap * M + m follows MFMA m (= product * RW + row) of that tap.
//   bpart/brow/bnext[m]: B fragment (activation part, row) requested after MFMA m, for the next tap (bnext) or for this one;
//   widx[m]: weight fragment (part) of the tap after next requested after MFMA m;
//   ext[s]: staging step of the NEXT chunk done after slot s: kind << 8 | index.  The wave's share of the chunk is staged in
//           ROUNDS rounds through one landing area; per round r: 1 = global load (r*8 + i), 2 = mask + park it (r*8 + i),
//           3 = read four channels of item set it (r*8 + it*2 + half), 4 = split half-step (r*32 + it*8 + 2*pair + half),
//           5 = store the three parts of item set it (r*4 + it), 6 = both half-steps of a pair (as 4), 7 = all eight channels (as 3) -- MERGE.
template <int RW, int NLD, int NSPL, int ROUNDS>
struct S2Sched {
    static constexpr bool MERGE = RW < 3;                      // few free slots per tap: a channel pair's two split half-steps share a slot
    static constexpr int NP = 3, M = S2Order::n * RW, S = 9 * M;
    int bpart[M], brow[M], bnext[M], widx[M], ext[S], used, nfree;
    constexpr S2Sched() : bpart(), brow(), bnext(), widx(), ext(), used(0), nfree(0) {
        for (int m = 0; m < M; ++m) { bpart[m] = -1; brow[m] = 0; bnext[m] = 0; widx[m] = -1; }
        for (int xp = NP - 1; xp >= 0; --xp) {
            int last = 0;
            for (int k = 0; k < S2Order::n; ++k) if (S2Order::x[k] == xp) last = k;
            const int gend = (last + 1) * RW - 1;              // last MFMA that reads part xp
            for (int j = 0; j < RW; ++j) {
                int mm = gend + j, nx = 1;
                if (mm >= M) { mm -= M; nx = 0; }              // part 0 wraps into the first MFMAs of the tap it is for
                bpart[mm] = xp; brow[mm] = j; bnext[mm] = nx;
            }
        }
        int wi = 0;
        for (int m = 0; m < M && wi < NP; ++m) if (bpart[m] < 0) widx[m] = wi++;
        // staging steps go to the slots that carry no operand request; a round's values fly >= two taps before they are touched,
        // the next round's loads are issued as soon as the registers are free (behind the parks of the round before)
        for (int s = 0; s < S; ++s) ext[s] = 0;
        int fr[S] = {}, nf = 0;
        for (int s = 0; s < S; ++s) { const int m = s % M; if (bpart[m] < 0 && widx[m] < 0) fr[nf++] = s; }
        const int per_tap = nf / 9;
        int f = 0, ready = 0;                                  // ready: first free-slot index at which the loads in flight may be parked
        ready = f + 2 * per_tap;                               // (two taps after the round's FIRST load was issued)
        for (int i = 0; i < NLD; ++i) ext[fr[f++]] = (1 << 8) | i;
        for (int r = 0; r < ROUNDS; ++r) {
            if (f < ready) f = ready;
            for (int i = 0; i < NLD; ++i) ext[fr[f++]] = (2 << 8) | (r * 8 + i);
            if (r + 1 < ROUNDS) {
                ready = f + 2 * per_tap;
                for (int i = 0; i < NLD; ++i) ext[fr[f++]] = (1 << 8) | ((r + 1) * 8 + i);
            }
            for (int it = 0; it < NSPL; ++it) {
                for (int h = 0; h < 2; h += MERGE ? 2 : 1) ext[fr[f++]] = ((MERGE ? 7 : 3) << 8) | (r * 8 + it * 2 + h);
                for (int hs = 0; hs < 8; hs += MERGE ? 2 : 1) ext[fr[f++]] = ((MERGE ? 6 : 4) << 8) | (r * 32 + it * 8 + hs);
                ext[fr[f++]] = (5 << 8) | (r * 4 + it);
            }
        }
        used = f; nfree = nf;                                  // static_assert at the use: the steps must fit the free slots
    }
};

}  // namespace

// Workgroup = WAVES waves, tile = (WAVES/2 * RW) rows x 32 columns x 64 couts; wave (ct, rg) = cout tile ct (32 couts) x rows
// RW*rg .. RW*rg + RW-1.  Two shapes are built, both one 8-wave workgroup per CU: <8, 3, 1> -- 12-row tiles -- and <8, 2, 1> --
// 8-row tiles, for the launches whose 12-row tiles would fill the rounds of the 256 CUs badly.  (A <4, 4, 2> shape -- two 4-wave
// workgroups per CU, the wave's share staged in two rounds -- compiles to 250 spilled VGPRs and was not pursued.)
template <int WAVES, int RW, int ROUNDS>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_split2_kernel(ConvArgs a, int ntiles, int tiles_y) {
    constexpr int NP = 3, TH = (WAVES / 2) * RW, PH = TH + 2, PW = 34, PHW = PH * PW;
    constexpr int SLOTS = 2 * PHW + 4;                   // per part: [2 octets][PHW] 16-byte slots + dummy
    constexpr int STG = NP * SLOTS;                      // one bf16 staging buffer (u32x4)
    constexpr int UW = 2 * PH * 2 / WAVES, UWR = UW / ROUNDS;   // (octet, row, half row) units per wave / per round: 8 channels x 5 quads
    constexpr int NQ = UWR * 40, NLD = (NQ + 63) / 64;   // 16-byte pieces of a round, loads per lane
    constexpr int RS = (WAVES == 8 || RW % 2) ? RW : 2;  // rows per epilogue pass (scratch = 8 couts x RS rows)
    constexpr int LW = NQ > 64 * RS ? NQ : 64 * RS;      // landing area of one wave (u32x4), reused as its epilogue scratch
    constexpr int NITEM = UWR * 17, NSPL = (NITEM + 63) / 64;            // (unit, pixel) items a wave splits per round
    static_assert(2 * PH * 2 % WAVES == 0 && UW % ROUNDS == 0 && NSPL <= 4 && NLD <= 8 && ROUNDS <= 2, "unit split");
    using SCHT = S2Sched<RW, NLD, NSPL, ROUNDS>;
    constexpr int M = SCHT::M;
    extern __shared__ __attribute__((aligned(16))) u32x4 lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), ct = wave & 1, rg = wave >> 1;
    float* bias_w = (float*)lds_raw + wave * 64;                       // [WAVES][64] (32 used)
    u32x4* stg0 = lds_raw + WAVES * 16;                               // [2 buffers][NP][SLOTS]
    u32x4* land = lds_raw + WAVES * 16 + 2 * STG + wave * LW;         // [WAVES][LW]: fp32 landing area / epilogue scratch

    const int G = gridDim.x, bq = xcd_block_id(blockIdx.x, G);
    if (bq >= ntiles) return;
    const int nch = a.Kpad / 9;
    const int ncgG = a.CK;                                              // groups * ncg (CK is otherwise unused by this kernel)
    const int HW = a.H * a.W;

    // Tile coordinates as mixed-radix digits (cout group | column | row | image): a workgroup steps by G tiles, so the digits are
    // advanced by G's digits with carries -- no integer division per tile (the SALU has none; 18 of them cost ~1.7 k cycles).
    struct TileC { int cgg, tx, ty, z; };
    auto coords_of = [&](int t) {
        TileC c;
        c.cgg = t % ncgG; int s = t / ncgG;
        c.tx = s % a.tiles_x; s /= a.tiles_x;
        c.ty = s % tiles_y; c.z = s / tiles_y;
        return c;
    };
    const TileC stepc = coords_of(__builtin_amdgcn_readfirstlane(G));
    auto advance = [&](TileC c) {
        c.cgg += stepc.cgg; int cy = c.cgg >= ncgG; c.cgg -= cy ? ncgG : 0;
        c.tx += stepc.tx + cy; cy = c.tx >= a.tiles_x; c.tx -= cy ? a.tiles_x : 0;
        c.ty += stepc.ty + cy; cy = c.ty >= tiles_y; c.ty -= cy ? tiles_y : 0;
        c.z += stepc.z + cy;
        return c;
    };
    auto decode = [&](const TileC& c, int& n, int& pz, int& g, int& cg, int& ty, int& tx) {
        tx = c.tx; ty = c.ty;
        g = a.ncg == ncgG ? 0 : c.cgg / a.ncg; cg = c.cgg - g * a.ncg;        // one group: no division
        pz = (c.z >= a.N) + (c.z >= 2 * a.N) + (c.z >= 3 * a.N); n = c.z - pz * a.N;   // <= MOTIF_MAX_PROBLEMS problems
    };

    // ---- staging plan of a tile (chunk-invariant) --------------------------------------------------------------------
    int doff[ROUNDS][NLD];                               // per-lane source offset of each staging load (-1: outside the image)
    const float* in0n = nullptr; const float* in1n = nullptr;
    int st_g = 0;
    float bias_v = 0.f;
    // weight fragments: buffer loads, descriptor = the (group, cout group)'s packed block, lane offset fixed, fragment offset scalar
    auto wptr = [&](const TileC& t) {
        int n, pz, g, cg, ty, tx;
        decode(t, n, pz, g, cg, ty, tx);
        const u32x4* base = (const u32x4*)a.wp[pz] + (long)(g * a.ncg + cg) * a.Kpad * (NP * 2 * 64);
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    };
    const int wvoff = (ct * 64 + lane) * 16;
    auto wfrag = [&](__amdgpu_buffer_rsrc_t wb, int ks, int p) {
        return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wb, wvoff, (ks * NP + p) * (2 * 64 * 16), 0));
    };
    auto unit_of = [&](int r, int ul, int& o, int& py, int& hx) {        // this wave's unit ul of round r -> (octet, patch row, half row)
        const int U = wave * UW + r * UWR + ul;
        o = U / (2 * PH); const int rem = U - o * (2 * PH); py = rem >> 1; hx = rem & 1;
    };
    auto setup_stage = [&](const TileC& t) {
        int n, pz, g, cg, ty, tx;
        decode(t, n, pz, g, cg, ty, tx);
        st_g = g;
        in0n = a.in0[pz] + (long)n * a.in0_bs[pz];
        in1n = a.in1[pz] ? a.in1[pz] + (long)n * a.in1_bs[pz] : nullptr;
        const float* bp = a.bias[pz];
        bias_v = (bp && lane < 32 && cg * 64 + ct * 32 + lane < a.Cout_g) ? bp[g * a.Cout_g + cg * 64 + ct * 32 + lane] : 0.f;
        const int iy0 = ty * TH - 1, x0 = tx * 32 - 4;   // pad 1 (host); staged rows start 4 pixels left of the tile: aligned quads
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int qd = i * 64 + lane;
                const int ul = qd / 40, rr = qd - ul * 40, ch = rr / 5, xq = rr - ch * 5;
                int o, py, hx;
                unit_of(r, ul, o, py, hx);
                const int iy = iy0 + py, x = x0 + 20 * hx + 4 * xq;
                // BYTE offset from the chunk's first plane; 2^31 = "outside the image": beyond any descriptor's range, so the load returns 0
                doff[r][i] = (qd < NQ && iy >= 0 && iy < a.H && x >= 0 && x < a.W) ? (iy * a.W + x + (8 * o + ch) * HW) * 4 : (int)0x80000000;
            }
    };
    // the steps of staging one 16-channel chunk (this wave's units, ROUNDS rounds), callable one small step at a time
    f32x4 gq[NLD];
    // Staging loads are BUFFER loads with a descriptor that ends where the chunk's channels end (the last channel of the group /
    // of the source tensor): the hardware range check returns 0 for the channels beyond Cin of a ragged last chunk and for the
    // quads outside the image (offset 2^31) -- no per-lane masking, and nothing is read past the end of the tensor.
    auto st_load = [&](int r, int i, int c0) {           // global -> registers: 4 pixels of one channel
        const int gch0 = st_g * a.Cin_g + c0;
        const bool first = gch0 < a.C0;
        const float* base = first ? in0n + (long)gch0 * HW : in1n + (long)(gch0 - a.C0) * HW;
        const int nch = (first && in1n) ? a.C0 - c0 : a.Cin_g - c0;           // channels from c0 to the end of the group / of the first source
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nch * HW * 4, 0x00020000);
        gq[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, doff[r][i], 0, 0));
    };
    auto st_park = [&](int r, int i, int c0) {           // park in LDS (zeros already in place)
        const int qd = i * 64 + lane;
        if (NLD * 64 == NQ || qd < NQ) *(f32x4*)(land + qd) = gq[i];
    };
    float sv[8];
    u32x4 sparts[NP];
    auto st_read = [&](int r, int it, int h4) {          // landing area -> four channels of this lane's item
        const int id = lane + 64 * it;
        const int ul = id / 17, pi = id - ul * 17;
        int o, py, hx;
        unit_of(r, ul, o, py, hx);
        const float* fw = (const float*)land + (ul < UWR ? ul : 0) * 160 + pi + 3 - 3 * hx;
#pragma unroll
        for (int q = 0; q < 4; ++q) sv[4 * h4 + q] = fw[(4 * h4 + q) * 20];
    };
    auto st_split = [&](int q, int hs) {                 // 3-way split of channel pair q: leading part | the two lower parts
        if (hs == 0) {
            const unsigned pk = pk_bf16(sv[2 * q], sv[2 * q + 1]);
            sparts[0][q] = pk;
            sv[2 * q] -= bf_lo(pk); sv[2 * q + 1] -= bf_hi(pk);
        } else {
            const unsigned pk = pk_bf16(sv[2 * q], sv[2 * q + 1]);
            sparts[1][q] = pk;
            sparts[2][q] = pk_bf16(sv[2 * q] - bf_lo(pk), sv[2 * q + 1] - bf_hi(pk));
        }
    };
    auto st_store = [&](int r, int it, u32x4* dstbuf) {  // the item's three parts -> staging buffer
        const int id = lane + 64 * it;
        const int ul = id / 17, pi = id - ul * 17;
        int o, py, hx;
        unit_of(r, ul, o, py, hx);
        const int slot = id < NITEM ? o * PHW + py * PW + 17 * hx + pi : 2 * PHW;      // surplus lanes write the dummy slot
#pragma unroll
        for (int p = 0; p < NP; ++p) dstbuf[p * SLOTS + slot] = sparts[p];
    };

    f32x16 acc[RW];
    constexpr int WB = 3;                                // weight fragments two taps ahead
    u32x4 wf[WB][NP];
    __amdgpu_buffer_rsrc_t wbase, wnext;
    auto loadw = [&](__amdgpu_buffer_rsrc_t wb, int ks, u32x4 (&dst)[NP]) {
#pragma unroll
        for (int p = 0; p < NP; ++p) dst[p] = wfrag(wb, ks, p);
    };

    // 9 taps of chunk c on staging buffer `buf`; STAGE: the staging steps of chunk `sc` (source plan as set up) go to the other
    // buffer, and the first two taps' weight fragments of that chunk (from `wn`) are requested during taps 7 and 8.
    auto chunk_body = [&](int c, int buf, auto stage_tag, int sc, __amdgpu_buffer_rsrc_t wn) {
        constexpr bool STAGE = decltype(stage_tag)::value;
        constexpr SCHT SCH{};
        static_assert(SCH.used <= SCH.nfree, "staging steps do not fit the free slots of a chunk");
        const u32x4* pb = stg0 + buf * STG + half * PHW + (RW * rg) * PW + l31;
        u32x4* dstbuf = stg0 + (buf ^ 1) * STG;
        u32x4 bfr[NP][RW];
        auto loadb = [&](int t, int p, int j) { bfr[p][j] = pb[p * SLOTS + (j + t / 3) * PW + (t % 3)]; };
#pragma unroll
        for (int p = NP - 1; p >= 0; --p)
#pragma unroll
            for (int j = 0; j < RW; ++j) loadb(0, p, j);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const int k = m / RW, j = m % RW;
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[t % WB][S2Order::w[k]]),
                                                                 __builtin_bit_cast(bf16x8, bfr[S2Order::x[k]][j]), acc[j], 0, 0, 0);
                if (SCH.bpart[m] >= 0) {
                    if (SCH.bnext[m]) { if (t < 8) loadb(t + 1, SCH.bpart[m], SCH.brow[m]); }
                    else if (t > 0) loadb(t, SCH.bpart[m], SCH.brow[m]);
                }
                if (SCH.widx[m] >= 0) {
                    const int p = SCH.widx[m];
                    if (t + 2 <= 8) wf[(t + 2) % WB][p] = wfrag(wbase, c * 9 + t + 2, p);
                    else if (STAGE) wf[(t + 2) % WB][p] = wfrag(wn, sc * 9 + t + 2 - 9, p);
                }
                if constexpr (STAGE) {
                    const int e = SCH.ext[t * M + m], kind = e >> 8, idx = e & 255;
                    if (kind == 1) st_load(idx >> 3, idx & 7, sc * 16);
                    else if (kind == 2) st_park(idx >> 3, idx & 7, sc * 16);
                    else if (kind == 3) st_read(idx >> 3, (idx >> 1) & 3, idx & 1);
                    else if (kind == 4) st_split((idx & 7) >> 1, idx & 1);
                    else if (kind == 6) { st_split((idx & 7) >> 1, 0); st_split((idx & 7) >> 1, 1); }
                    else if (kind == 7) { st_read(idx >> 3, (idx >> 1) & 3, 0); st_read(idx >> 3, (idx >> 1) & 3, 1); }
                    else if (kind == 5) st_store(idx >> 2, idx & 3, dstbuf);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto epilogue = [&](const TileC& t) {
        int n, pz, g, cg, ty, tx;
        decode(t, n, pz, g, cg, ty, tx);
        const int cbase = g * a.Cout_g + cg * 64 + ct * 32, climit = a.Cout_g - cg * 64 - ct * 32;
        if (climit <= 0) return;                         // the upper cout tile of a partial group has nothing to store
        const long HWo = (long)a.Ho * a.Wo;
        float* ob = a.out[pz] + (long)n * a.out_bs[pz] + (long)cbase * HWo;
        const float* rb = a.res_mode ? a.res[pz] + (long)n * a.res_bs[pz] + (long)cbase * HWo : nullptr;
        int lane_e = lane;                               // opaque copy: keeps the per-lane address arithmetic inside the tile loop
        asm volatile("" : "+v"(lane_e));
        if (a.res_mode) conv_epilogue_wave<RW, RS, true>(a, acc, bias_w, (float*)land, lane_e, cbase, climit, ty * TH + RW * rg, tx * 32, rb, ob);
        else conv_epilogue_wave<RW, RS, false>(a, acc, bias_w, (float*)land, lane_e, cbase, climit, ty * TH + RW * rg, tx * 32, rb, ob);
    };

    // ---- prologue: first chunk of the first tile (nothing to hide it under) --------------------------------------------
    int t = bq;
    TileC tc = coords_of(__builtin_amdgcn_readfirstlane(t)), tn = tc;
    S2TRACE(0);
    setup_stage(tc);
    wbase = wnext = wptr(tc);
    loadw(wbase, 0, wf[0]);
    loadw(wbase, 1, wf[1]);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) st_load(r, i, 0);
        if (r == 0 && lane < 32) bias_w[lane] = bias_v;
#pragma unroll
        for (int i = 0; i < NLD; ++i) st_park(r, i, 0);
#pragma unroll
        for (int it = 0; it < NSPL; ++it) {
            st_read(r, it, 0); st_read(r, it, 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) { st_split(q, 0); st_split(q, 1); }
            st_store(r, it, stg0);
        }
    }
#pragma unroll
    for (int j = 0; j < RW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    __syncthreads();
    S2TRACE(1);

    // ---- persistent tile loop ------------------------------------------------------------------------------------------
    int buf = 0, slot = 2;
    for (;;) {
        const int t_next = t + G;
        const bool has_next = t_next < ntiles;
        for (int c = 0; c < nch; ++c) {
            const bool last = c + 1 == nch;
            if (last && has_next) { tn = advance(tc); setup_stage(tn); wnext = wptr(tn); }   // from here on the staging steps belong to the next tile
            if (!last) chunk_body(c, buf, std::true_type{}, c + 1, wbase);
            else if (has_next) chunk_body(c, buf, std::true_type{}, 0, wnext);
            else chunk_body(c, buf, std::false_type{}, 0, wbase);
            __syncthreads();
            buf ^= 1;
            if (slot < 30) { S2TRACE(slot); ++slot; }
        }
        epilogue(tc);
        if (slot < 30) { S2TRACE(slot); ++slot; }
        if (!has_next) break;
        t = t_next; tc = tn; wbase = wnext;
        if (lane < 32) bias_w[lane] = bias_v;
#pragma unroll
        for (int j = 0; j < RW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    }
    S2TRACE(31);
}

// ---- host side ------------------------------------------------------------------------------------------------------
namespace {
int s2_cu_count() {                                      // init-once device probe (the only cached state)
    static int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus;
}
}  // namespace

// What these kernels take (everything else stays on conv_split_kernel): fp32-equivalent arithmetic, zero padding 1, rows of whole
// 16-byte units, 16-byte aligned tensors (16-byte staging loads and stores), activation split on an 8-cout boundary.
bool motif_conv_split2_eligible(const MotifConvDesc* d, const ConvArgs& a, int P) {
    if (split_parts(d->mma) != 3 || d->pad != 1 || d->pad_mode != 0 || (d->W & 3)) return false;
    const long HW = (long)d->H * d->W;
    if (HW * 64 >= 0x7fffffffL) return false;
    const int Cout_g = d->Cout / d->groups;
    if ((long)((d->C0 + d->C1) / d->groups) * HW * 4 >= 0x7fffffffL) return false;       // byte offsets of the staging buffer loads
    if (d->act_split > 0 && ((d->act_split & 7) || (d->groups > 1 && (Cout_g & 7)))) return false;
    if (d->C1 > 0 && (d->groups != 1 || d->C0 % 16)) return false;
    for (int i = 0; i < P; ++i) {
        unsigned long long bits = (unsigned long long)a.in0[i] | (unsigned long long)a.out[i] | (unsigned long long)a.in1[i] | (unsigned long long)a.res[i];
        if (bits & 15) return false;
        if ((a.in0_bs[i] | a.out_bs[i] | (a.in1[i] ? a.in1_bs[i] : 0) | (a.res[i] ? a.res_bs[i] : 0)) & 3) return false;
    }
    const int force = motif_opt(MOTIF_OPT_CONV_ENGINE);
    if (force == 2 || force == 3 || force == 4) return true;           // forced (tests, tools)
    // Measured crossover on the clip's launches (tools/r3_conv.sh: per-shape A/B inside the model): the 12-row shape wins where
    // its tiles fill >= 0.9 of the rounds of the CUs (+3 ... +13 %), the 8-row shape where every CU gets at most one tile (+4 ... +20 %:
    // the deep-K layers of RAFT's update block); in between the two-block kernel (8-row tiles, 512 block slots) is as good or better.
    const long per_row_tile = (long)((d->W + 31) / 32) * d->groups * ((Cout_g + 63) / 64) * d->N * P, cus = s2_cu_count();
    const long T12 = per_row_tile * ((d->H + 11) / 12), T8 = per_row_tile * ((d->H + 7) / 8);
    return (double)T12 / (double)(((T12 + cus - 1) / cus) * cus) >= 0.9 || T8 <= cus;
}

// Persistent workgroups, tile i of workgroup b = b' + i * G.  Shape by tile count: 12-row tiles with one 8-wave workgroup per CU
// where they fill >= 0.9 of the rounds of the CUs, else 8-row tiles (option conv_engine: 2 forces the first, 3 the second).
int motif_conv_split2_launch(const MotifConvDesc* d, ConvArgs& a, int P, hipStream_t s) {
    const int Cin_g = (d->C0 + d->C1) / d->groups, Cout_g = d->Cout / d->groups;
    const int Ho = d->H, Wo = d->W;                      // pad 1
    a.Ho = Ho; a.Wo = Wo; a.Cin_g = Cin_g; a.Cout_g = Cout_g;
    a.Kpad = 9 * ((Cin_g + 15) / 16);
    a.ncg = (Cout_g + 63) / 64;
    a.tiles_x = (Wo + 31) / 32;
    const int ncgG = d->groups * a.ncg, cus = s2_cu_count();
    a.Cout = d->Cout;
    a.CK = ncgG;                                         // unused by these kernels otherwise: carries groups * ncg
    const long per_row_tile = (long)a.tiles_x * ncgG * d->N * P;
    const long T12 = per_row_tile * ((Ho + 11) / 12), T8 = per_row_tile * ((Ho + 7) / 8);
    if (T8 >= 0x7fffffffL) return MOTIF_ELIMIT;
    const long rounds12 = (T12 + cus - 1) / cus;
    const int force = motif_opt(MOTIF_OPT_CONV_ENGINE);
    const bool big = force == 2 || (force != 3 && (double)T12 / (double)(rounds12 * cus) >= 0.9);      // else: 8-row tiles (T8 <= CUs, or forced)
    // per launch, like the other kernels: a process-wide "done once" flag is neither per device nor thread-safe
    {
        const void* kfn = (force == 4) ? (const void*)conv_split2_kernel<4, 3, 1> : big ? (const void*)conv_split2_kernel<8, 3, 1> : (const void*)conv_split2_kernel<8, 2, 1>;
        const hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, force == 4 ? 80 * 1024 : 160 * 1024);
        if (e != hipSuccess) return (int)e;
    }
    if (force == 4) {                                    // 6-row tiles, two 4-wave workgroups per CU: best on the smallest maps in isolation
                                                         // (+10 ... +35 % on 45x80 / 90x160 single launches), a draw or worse inside the model; opt-in
        const long T6 = per_row_tile * ((Ho + 5) / 6);
        const int G = (int)(T6 < 2 * cus ? T6 : 2 * cus);
        const size_t ldsb = ((size_t)4 * 16 + (size_t)2 * 3 * (2 * 8 * 34 + 4) + (size_t)4 * 320) * 16;
        conv_split2_kernel<4, 3, 1><<<dim3(G, 1, 1), 256, ldsb, s>>>(a, (int)T6, (Ho + 5) / 6);
    } else if (big) {
        const int G = (int)(T12 < cus ? T12 : cus);
        const size_t ldsb = ((size_t)8 * 16 + (size_t)2 * 3 * (2 * 14 * 34 + 4) + (size_t)8 * 280) * 16;
        conv_split2_kernel<8, 3, 1><<<dim3(G, 1, 1), 512, ldsb, s>>>(a, (int)T12, (Ho + 11) / 12);
    } else {
        const int G = (int)(T8 < cus ? T8 : cus);
        const size_t ldsb = ((size_t)8 * 16 + (size_t)2 * 3 * (2 * 10 * 34 + 4) + (size_t)8 * 200) * 16;
        conv_split2_kernel<8, 2, 1><<<dim3(G, 1, 1), 512, ldsb, s>>>(a, (int)T8, (Ho + 7) / 8);
    }
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

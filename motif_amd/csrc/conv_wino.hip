// 3x3 / stride 1 convolution with 1.5x fewer matrix instructions: Winograd F(2,3) along the image ROWS, direct along the
// columns, fp32-equivalent split arithmetic on the 16-bit matrix cores.  Round-4 kernel, in two forms (template parameter NP):
//   NP = 3  three bf16 parts per operand, six products (the arithmetic of conv_split.hip; MotifConvDesc.mma = 6);
//   NP = 2  two fp16 parts per operand (hi = rne(x), lo = rne(x - hi): 22+ bits), THREE products (mma = 7, the default): half the matrix
//           instructions again.  Why it pays twice: under a chip-wide MFMA load on real data the shader clock is 1.5-1.8 GHz, not 2.4
//           (tools/ubench_clock.hip: the matrix cores are power-bound), so an MFMA not issued returns its cycles AND clock.  Range = fp16's:
//           |V| (= up to 2 |activation|) or 2^8 |U| beyond 65504 give inf / NaN, never a clamped value; below 2^-14 a part is subnormal, i.e.
//           carries an absolute error <= 2^-25.  The weights are packed times 2^8 (so that the low parts of everyday weights are normal
//           numbers), the accumulators start at 2^8 bias, the epilogue multiplies by 2^-8 -- all exact.  Measured against fp64 the form's
//           error is at or below the three-part one's (tools/wino_check.py, tests/test_kernels_gpu.py, tests/test_split_arith.py).
//
// For an output row pair (2T, 2T+1), input rows d0..d3 = 2T-1 .. 2T+2 and the kernel rows g0, g1, g2 (per column tap kx):
//     V0 = d0 - d2     U0 = g0                    M_p = sum over (cin, kx) of U_p[kx] * V_p[x + kx]
//     V1 = d1 + d2     U1 = (g0 + g1 + g2) / 2    out(2T)   = M0 + M1 + M2
//     V2 = d2 - d1     U2 = (g0 - g1 + g2) / 2    out(2T+1) = M1 - M2 - M3
//     V3 = d1 - d3     U3 = g2
// i.e. 4 x 3 = 12 k-steps per 16-channel chunk and row PAIR instead of 2 x 9 = 18.  The transform is along y only: along x the
// kernel stays a direct convolution, so a B fragment is still one ds_read_b128 at an immediate column offset, the C/D layout of an
// output row is the one of the direct kernels, and the 2-D form's operand traffic (no fragment reuse at all: one 1 KB fragment per
// MFMA) and 256-register accumulator set are avoided.  U is formed in fp64 from the fp32 weights and split
// into three bf16 parts at pack time; V is one fp32 addition per value, then the same exact 3-way split as the direct kernels.
// Rounding differs from the direct form by that one addition per operand and by the three-term output sums.
//
// Shape: workgroup = 4 waves, ONE WAVE PER SIMD (512 registers per lane), tile = 8 rows x 32 columns x 64 couts.  MFMA role of wave
// (ct, tp): cout tile ct x row pairs 2tp, 2tp+1 (8 accumulators = [pair][position]).  Staging role of wave w: row pair T = w -- it
// fetches the pair's 4 input rows of a 16-channel chunk as 16-byte row pieces (range-checked buffer loads: zeros outside the image
// and past Cin), parks them in its private landing area, reads them back pixel-major, forms V (one lane = one pixel x 8 channels x
// 4 positions), splits and writes the staging buffer [part][octet][pair][position][34 px] x 8 bf16; the two halo columns of the pair
// are done channel-parallel, one (position, channel pair) per lane.
//
// A lone wave hides nothing behind another wave, so EVERYTHING is a statically scheduled filler of the MFMA stream (what the
// stream tolerates was measured with tools/ubench_lone*.hip: <= 4 independent vector instructions per MFMA are free, a dependent
// one costs ~8 cycles, v_pk_add_f32 ~12 that do NOT overlap, a 1 KB load 64 cycles of the CU's L1 path (16 per wave), LDS stores 13):
//   * the (tile, chunk) pairs of a persistent workgroup form one flat stream of steps; during step s the wave multiplies step s,
//     parks / transforms / splits step s+1 (loaded during step s-1) and requests the raw rows of step s+2;
//   * a chunk is 6 super-steps (position pair, kx) of 24 MFMAs (NP = 2: 12); B fragments have ONE register copy (a part's registers take the
//     next super-step's fragment as soon as its last product is issued), weight fragments are two super-steps ahead, vector-memory
//     requests are never in adjacent slots;
//   * the epilogue of a tile is exposed (carrying it under the next tile's chunks needs 64 holding registers on top of the 128
//     accumulators and ~300 operand / staging registers: hipcc spilled 400 of them) but short: inverse transform + bias in the C/D
//     layout, four 8-cout passes through the wave's landing area (two 4 KB halves: pass p+1 is written before pass p is read back),
//     residual quads requested two passes ahead, 16-byte stores as inline assembly the wait-count pass does not see (a pending
//     store would turn every counted wait for a load -- the next step's row pieces and weights are in flight -- into vmcnt(0)).
// The two-part form runs at the balance point of three CU resources (per chunk: matrix pipe 2.3 k cycles, LDS 304 KB = 2.4 k, L1 path
// 136 KB = 2.1 k): its row-piece requests are therefore spread two per super-step instead of sitting behind the parks.
#include "conv_wave_epilogue.h"
#include <utility>

#ifdef MOTIF_TRACE
__device__ long long g_wn_trace[1024 * 4 * 32];
#define WNTRACE(slot) do { if (lane == 0 && blockIdx.x < 1024) g_wn_trace[(blockIdx.x * 4 + wave) * 32 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define WNTRACE_RT(slot) do { if (lane == 0 && blockIdx.x < 1024) g_wn_trace[(blockIdx.x * 4 + wave) * 32 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)   // the constant 100 MHz counter: shader clock = cycles / time
extern "C" int motif_debug_wino_trace(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wn_trace), sizeof(long long) * n); }
// per-super-step stamps of the first 8 chunks of a workgroup: [block][wave][chunk][7]
__device__ long long g_wn_trace2[256 * 4 * 8 * 8];
extern "C" int motif_debug_wino_trace2(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wn_trace2), sizeof(long long) * n); }
// chain mode, wave 0, end of the first 8 tiles of a workgroup: [block][tile][8] stamps (see CHSTAMP)
__device__ long long g_ch_trace[256 * 8 * 8];
extern "C" int motif_debug_chain_trace(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ch_trace), sizeof(long long) * n); }
#define CHSTAMP(i) do { if (CHAIN && wave == 0 && lane == 0 && blockIdx.x < 256 && ti < 8) g_ch_trace[(blockIdx.x * 8 + ti) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WNTRACE(slot)
#define WNTRACE_RT(slot)
#define CHSTAMP(i)
#endif

#ifndef WINO_ABL
#define WINO_ABL 0        // ablation builds (tools/wino_ablate.sh): bit 0 no loads, 1 no parks, 2 no read-back, 3 no transform/split,
#endif                    // 4 no staging stores, 5 no halo item, 6 no B fragments, 7 no weight fragments, 8 no epilogue pieces,
                          // 9 every second product only (the MFMA count of a 3-product split), 10 no MFMAs at all

#ifndef CHAIN_DEFER
#define CHAIN_DEFER 0     // 0: a tile is published at its own end, behind s_waitcnt vmcnt(0); 1: under chunk 1 of the next tile (see pub_pending:
                          // measured SLOWER on the same box, 3.11 against 2.89 ms for the 40-block trunk)
#endif
#ifndef CHAIN_ABL
#define CHAIN_ABL 0       // chain-mode ablation builds (timing only, results invalid): bit 0 no store wait before a tile is published, 1 plain
#endif                    // (not device-coherent) activation loads / stores, 2 no flag polls (every tile counts as ready)

namespace {
// Products ordered by ACTIVATION part, smallest part first (w = weight part, x = activation part).
// NP = 3: three bf16 parts per operand, the six products whose dropped remainder is <= 2^-25 relative (conv_split.hip).
// NP = 2: two fp16 parts per operand (hi = rne(x), lo = rne(x - hi): 22+ significant bits, |x - hi - lo| <= 2^-23 |x| while the
//         parts stay normal numbers), three products -- half the matrix instructions.  See the header of the file for the range.
template <int NP> struct WOrder;
template <> struct WOrder<3> {
    static constexpr int n = 6;
    static constexpr int w[6] = {0, 1, 0, 2, 1, 0};
    static constexpr int x[6] = {2, 1, 1, 0, 0, 0};
};
//         Round 5: the low activation part is stored times 2^11 (lo_s = rne((x - hi) * 2^11): a number of hi's own magnitude, so it is a
//         NORMAL fp16 number whenever hi is -- unscaled it was a subnormal, an absolute 2^-25, for every |x| < 0.25) and multiplied by the
//         high weight part times 2^-11 (weight "part" 2: formed in registers by four v_pk_mul_f16 per fragment, exact while normal).
template <> struct WOrder<2> {
    static constexpr int n = 3;
    static constexpr int w[3] = {1, 0, 2};
    static constexpr int x[3] = {0, 0, 1};
};

enum { WP_PARK = 2, WP_READ = 3, WP_X = 4, WP_W = 5, WP_HREAD = 6, WP_HX = 7, WP_HW = 8, WP_READ2 = 9 };

// Static schedule of a chunk: 6 super-steps (position pair, kx) of M = 4 * products MFMAs = product x (position of the pair, row
// pair), four accumulators in rotation.  Slot s = ss * M + m follows MFMA m of super-step ss and holds at most one request of each
// kind and one staging piece.  Every operand register has ONE copy per prefetch depth and is re-requested right after the last
// product that reads it (NP = 3 figures; NP = 2 follows the same rules on its 12-slot super-steps):
//   rb[s]:  B fragment (part * 4 + j) of the NEXT super-step (part 2 after MFMAs 0..3, part 1 after 8..11, part 0 after 20..23);
//   ra[s]:  weight fragment (position of the pair * NP + part) of the super-step THREE ahead, into the set in use (three sets): part 2
//           after MFMAs 13 / 15, part 1 after 17 / 19, part 0 after 21 / 23 -- ~2.2 super-steps (~2 k cycles) before its first use.
//           vmcnt retires in order, so a weight fragment can only be consumed once every OLDER load has returned, raw row pieces
//           included (first touches of another XCD's output: ~2 k cycles); hence
//   rl[s]:  raw row piece i of the step after next is requested as soon as its landing registers are free (right after its
//           park): it then has >= four super-steps before the next chunk parks it, and the weight fragments requested behind it are
//           not needed for two.  Parks and requests are spread over super-steps 0 and 1: 14 one-KB requests in one super-step are
//           900 cycles of the CU's L1 path, 40 ds_write_b128 of four waves 500 of the LDS store path.
//   ext[s]: staging piece of the NEXT step: parks (NP = 3: every fourth slot of super-steps 0 and 1; NP = 2: every second), read-back,
//           halo reads, per position XS transform / split stages of <= 4 INDEPENDENT instructions + NP stores, the halo item last.
template <int NP>
struct WSched {
    static constexpr int SS = 6, M = 4 * WOrder<NP>::n, S = SS * M, NLD = 10;
    static constexpr int XS = NP == 3 ? 13 : 7;          // transform / split stages per position (st_x)
    static constexpr int HXS = NP == 3 ? 4 : 3;          // ... of the halo item (st_hx)
    int rb[S], ra[S], rl[S], ext[S], used, clash;
    constexpr WSched() : rb(), ra(), rl(), ext(), used(0), clash(0) {
        using O = WOrder<NP>;
        for (int s = 0; s < S; ++s) { rb[s] = -1; ra[s] = -1; rl[s] = -1; ext[s] = 0; }
        int lastx[NP] = {}, lastw[NP] = {};                                   // last product reading activation / weight part p
        for (int k = 0; k < O::n; ++k) { lastx[O::x[k]] = k; if (O::w[k] < NP) lastw[O::w[k]] = k; }
        for (int ss = 0; ss < SS; ++ss) {
            if (ss + 1 < SS)
                for (int p = 0; p < NP; ++p)
                    for (int j = 0; j < 4; ++j) rb[ss * M + lastx[p] * 4 + j] = p * 4 + j;
            for (int p = 0; p < NP; ++p)
                for (int pi = 0; pi < 2; ++pi) ra[ss * M + lastw[p] * 4 + 2 * pi + 1] = pi * NP + p;
        }
        int f = 0;
        if (NP == 3) {
            int pslot[NLD] = {};
            for (int i = 0; i < NLD; ++i) { ext[f] = (WP_PARK << 8) | i; pslot[i] = f; f += 4; }
            for (int i = 0; i < NLD; ++i) {
                const int ls = pslot[i] + 2;                                     // a row piece is requested right after its park
                if (rl[ls] >= 0 || ra[ls] >= 0) clash = 1;
                rl[ls] = i;
            }
            for (int r = 0; r < 16; ++r) ext[f++] = (WP_READ << 8) | r;
        } else {
            // parks in the odd slots 1 .. 19; the read-back of row r (row piece quads 160 r .. 160 r + 159: pieces 0-2 | 2-4 | 5-7 | 7-9)
            // in the even slots behind its last park.  The row-piece requests are NOT right behind the parks here but two per
            // super-step in even slots of super-steps 1 .. 5 (the weight requests sit in odd slots): a super-step is 384 MFMA cycles =
            // 24 KB of the CU's 64 B/clk L1 path, and the four waves' weight fragments alone are 16 KB of it -- with five of the ten
            // requests in super-step 0 and five in 1 those took 700 / 1050 cycles, spread they all take 540-650 (tools/trace_wino.py SS=1).
            for (int i = 0; i < NLD; ++i) {
                ext[2 * i + 1] = (WP_PARK << 8) | i;
                const int ls = 14 + 6 * i;               // (measured beside: right behind the park, 14 + 4 i, 22 + 4 i)
                if (rl[ls] >= 0 || ra[ls] >= 0) clash = 1;
                rl[ls] = i;
            }
            constexpr int rslot[8] = {6, 8, 10, 12, 16, 18, 20, 21};
            for (int r = 0; r < 8; ++r) { if (ext[rslot[r]]) clash = 1; ext[rslot[r]] = (WP_READ2 << 8) | r; }
            f = 22;
        }
        for (int r = 0; r < 2; ++r) ext[f++] = (WP_HREAD << 8) | r;
        for (int pos = 0; pos < 4; ++pos)
            for (int i = 0; i < XS + NP; ++i) ext[f++] = (i < XS ? (WP_X << 8) | (pos * 16 + i) : (WP_W << 8) | (pos * NP + i - XS));
        for (int h = 0; h < HXS; ++h) ext[f++] = (WP_HX << 8) | h;
        for (int p = 0; p < NP; ++p) ext[f++] = (WP_HW << 8) | p;
        used = f;
    }
};

// Scalar fp32 add / subtract the compiler cannot pair into v_pk_add_f32: a packed add costs a lone wave ~12 cycles that do NOT
// overlap its MFMA stream (tools/ubench_lone.hip: 2 per MFMA -> 55 cycles per MFMA), two plain ones 8 that do.
__device__ __forceinline__ float fadd1(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float fsub1(float a, float b) { float r; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// A load through a pointer rebuilt from integers (the tile table) is a FLAT load to hipcc: it could be LDS, returns out of order with
// the buffer loads, and ONE such request possibly pending anywhere in the tile loop turns every counted wait of the loop into
// s_waitcnt vmcnt(0) (found as an unconditional vmcnt(0) in front of every chunk's first B fragments).  Hence: explicitly global.
template <class T>
__device__ __forceinline__ T gload(const T* p) { return *(const __attribute__((address_space(1))) T*)p; }

template <int NP> constexpr WSched<NP> kWSchedOf{};
static_assert(kWSchedOf<3>.used <= WSched<3>::S && kWSchedOf<3>.clash == 0, "pieces do not fit the slots of a chunk");
static_assert(kWSchedOf<2>.used <= WSched<2>::S && kWSchedOf<2>.clash == 0, "pieces do not fit the slots of a chunk (two-part form)");

// fp16 pair helpers of the two-part form: round-to-nearest pack (v_cvt_pk_f16_f32), and x - (float)half of a packed pair as ONE
// mixed-precision FMA (v_fma_mix_f32: half * -1.0 + x; the product by -1 is exact)
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pk_f16(float a, float b) { const f16x2v h = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(unsigned, h); }
__device__ __forceinline__ float sub_f16_lo(float x, unsigned pk) { float r; asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x)); return r; }
__device__ __forceinline__ float sub_f16_hi(float x, unsigned pk) { float r; asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(x)); return r; }
// the weights of the two-part form are packed times 2^8 (a weight of everyday magnitude 1e-3 .. 1 then has BOTH its parts in fp16's
// normal range; 2^8 |U| must stay below 65504, U = the transformed kernel rows of the header), the epilogue multiplies by 2^-8 (exact)
constexpr float kWinoF16Scale = 256.f;
// (a, b) * s -> packed fp16 pair, ONE rounding each (v_fma_mixlo / mixhi_f16: fp32 a * fp32 s + 0 -> fp16): the low activation part times 2^11
__device__ __forceinline__ void pk_f16_scaled_lo(unsigned& d, float a, float s) { asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(d) : "v"(a), "s"(s)); }
__device__ __forceinline__ void pk_f16_scaled_hi(unsigned& d, float b, float s) { asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(d) : "v"(b), "s"(s)); }
constexpr float kWinoLoScale = 2048.f;                  // 2^11: |x - hi| <= 2^-11 |x|, so the scaled low part never exceeds |x|
__device__ __forceinline__ unsigned pk_mul_f16(unsigned a, f16x2v c) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(f16x2v, a) * c); }

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>) -- every slot of the schedule is its own
// instantiation (the loop unroller's size estimate, taken before the dispatch on the slot's piece is folded, refuses 144 slots)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
}  // namespace

// TR (round 5): the two MFMA operands are SWAPPED (activations as A, weights as B -- both operands have the same per-lane structure, a lane
// = one row / column index and 8 consecutive k, so the staging and the weight fragments are untouched) and the accumulators come out
// TRANSPOSED: lane = cout, registers = pixels, four consecutive registers = four consecutive pixels of a row.  A lane then holds 16-byte
// pieces of NCHW rows, and the epilogue stores them straight from registers: no transposition through LDS (16 ds_write + 4 ds_read_b128 per
// pass and their round trip), the bias is one value per lane, the residual is loaded in the same pieces.  ~400 instead of ~750 instructions
// in the exposed epilogue of every tile.  Per-cout activation switches (act_split: the DCNs' offset | sigmoid(mask) layers) would diverge
// across lanes there: those launches keep the row-major form.  RESULT: bit-identical, and no faster (see motif_conv_wino_launch): opt-in.
// MULTI: more than one problem in the launch (the per-problem argument selects are ~100 scalar instructions per tile: a lone wave
// hides none of them).
// CHAIN (round 5): ch.L dependent same-shape layers (a residual trunk: module_util.py:34-52 x 40, Ours.py:349) in ONE persistent launch.  The
// tiles of all layers form one ticket sequence in layer-major order (atomic counter: a workgroup that is not resident holds no ticket, so the
// smallest unfinished tile is always owned by a running workgroup -- no deadlock whatever the residency, two clips in flight included); tile
// (l, y, x) needs the <= 9 tiles (l-1, y+-1, x+-1): one completion flag per tile, set after the tile's stores have been acknowledged, polled
// WITHOUT blocking while the previous tile is still being multiplied (the next tile's row pieces are requested two chunks ahead) and waited
// for only between tiles, when the workgroup has nothing in flight.  The XCDs' L2s are not coherent inside a kernel: activations and
// residuals are loaded and stored with sc1 (device-coherent), weights and biases (read-only) stay cached; tools/ubench_chain.hip is the
// protocol on its own (exact over 80 layers x 690 tiles; the plain data path fails there).  No launch boundaries, ONE prologue per workgroup
// instead of one per layer, and the round fill of a layer (690 tiles on 256 CUs: 0.90) becomes that of the whole chain (~1.0).
template <int NP, bool MULTI, bool TR, bool CHAIN>
__device__ __forceinline__ void conv_wino_body(const ConvArgs& a, const int ntiles, const int tiles_y, const ChainArgs& ch) {
    static_assert(!CHAIN || (NP == 2 && !MULTI && !TR), "chain mode: two-part row-major form, one problem");
    using WS = WSched<NP>;
    using WO = WOrder<NP>;
    constexpr int WAVES = 4, TH = 8, PW = 34, OCT = 16 * PW;   // 16 (pair, position) planes of 34 pixels per octet
    constexpr int SLOTS = 2 * OCT + 4;                   // per part: [2 octets][16 planes][34] 16-byte slots (+ pad)
    constexpr int STG = NP * SLOTS;                      // one bf16 staging buffer (u32x4)
    constexpr int NLD = WS::NLD, LW = NLD * 64;      // a wave's landing area: [4 rows][2 octets][8 channels][40 px] floats
    constexpr int M = WS::M, SS = WS::SS;
    extern __shared__ __attribute__((aligned(16))) u32x4 lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), ct = wave & 1, tp = wave >> 1;
    float* bias_w = (float*)lds_raw + wave * 64;                       // [WAVES][64] (32 used)
    u32x4* stg0 = lds_raw + WAVES * 16;                               // [2 buffers][NP][SLOTS]
    u32x4* land = lds_raw + WAVES * 16 + 2 * STG + wave * LW;         // [WAVES][LW]: fp32 landing area / epilogue scratch
    const float* landf = (const float*)land;
    u32x4 zq = {0u, 0u, 0u, 0u};                         // see init_acc
    asm volatile("" : "+v"(zq));
    constexpr int TT = 128;                              // tile-parameter table: [TT][4] x 16 bytes, entry = ordinal of the tile in this workgroup
    u32x4* ttab = lds_raw + WAVES * 16 + 2 * STG + WAVES * LW;

    const int G = gridDim.x, bq = CHAIN ? 0 : xcd_block_id(blockIdx.x, G);
    if (!CHAIN && bq >= ntiles) return;
    const int nch = a.Kpad / 12;                                        // >= 2 (host)
    const int ncgG = a.CK;                                              // groups * ncg
    const int HW = a.H * a.W;
    const unsigned HWo = (unsigned)(a.Ho * a.Wo);

    // Per-problem launch arguments by CONSTANT index + scalar selects: a dynamic index into the kernel-argument arrays becomes a
    // VECTOR load whose result every later use waits for with vmcnt(0) -- i.e. for all the row pieces and weight fragments in flight.
    auto pick = [](const auto (&arr)[MOTIF_MAX_PROBLEMS], int pz) __attribute__((always_inline)) {
        if constexpr (!MULTI) return arr[0];
        else return pz == 0 ? arr[0] : pz == 1 ? arr[1] : pz == 2 ? arr[2] : arr[3];
    };

    // ---- tile parameters.  Tile e of this workgroup = bq + e * G; its coordinates (three integer divisions), source / weight / output /
    // residual / bias addresses (64-bit multiply-adds, per-problem selects) are computed by ONE LANE EACH of wave 0 -- 64 tiles at a
    // time on the vector ALU -- and parked in LDS; a tile change then costs four broadcast ds_read_b128 and sixteen readfirstlanes
    // instead of ~150 dependent scalar instructions (a lone wave hides none of them: they were 1-1.5 k of a tile's 34 k cycles).
    struct TileP { const float* in0n; const float* in1n; const u32x4* wb; float* ob; const float* rb; const float* bp; int ty, tx, g, clg, cbg, act, rm, tk; };
    auto fill_table = [&](int e0) __attribute__((always_inline)) {       // entries e0 .. e0 + 63, by the 64 lanes of the calling wave
        const int e = e0 + lane;
        const long tl = (long)bq + (long)e * G;
        if (tl < ntiles) {
            const int tt = (int)tl;
            const int cgg = tt % ncgG; int s2 = tt / ncgG;
            const int tx = s2 % a.tiles_x; s2 /= a.tiles_x;
            const int ty = s2 % tiles_y, z = s2 / tiles_y;
            const int g = a.ncg == ncgG ? 0 : cgg / a.ncg, cg = cgg - g * a.ncg;
            int pz = 0, n = z;
            if constexpr (MULTI) { pz = (z >= a.N) + (z >= 2 * a.N) + (z >= 3 * a.N); n = z - pz * a.N; }
            const long HWol = (long)a.Ho * a.Wo;
            const int cbg = g * a.Cout_g + cg * 64;
            const unsigned long long i0 = (unsigned long long)(pick(a.in0, pz) + (long)n * pick(a.in0_bs, pz));
            const float* i1p = pick(a.in1, pz);
            const unsigned long long i1 = i1p ? (unsigned long long)(i1p + (long)n * pick(a.in1_bs, pz)) : 0ull;
            const unsigned long long wb = (unsigned long long)((const u32x4*)pick(a.wp, pz) + (long)(g * a.ncg + cg) * a.Kpad * (NP * 2 * 64));
            const unsigned long long ob = (unsigned long long)(pick(a.out, pz) + (long)n * pick(a.out_bs, pz) + (long)cbg * HWol);
            const float* rp = pick(a.res, pz);
            const unsigned long long rb = (a.res_mode && rp) ? (unsigned long long)(rp + (long)n * pick(a.res_bs, pz) + (long)cbg * HWol) : 0ull;
            const float* bpp = pick(a.bias, pz);
            const unsigned long long bp = bpp ? (unsigned long long)(bpp + cbg) : 0ull;
            u32x4* r = ttab + (e & (TT - 1)) * 4;
            r[0] = u32x4{(unsigned)i0, (unsigned)(i0 >> 32), (unsigned)i1, (unsigned)(i1 >> 32)};
            r[1] = u32x4{(unsigned)wb, (unsigned)(wb >> 32), (unsigned)ob, (unsigned)(ob >> 32)};
            r[2] = u32x4{(unsigned)rb, (unsigned)(rb >> 32), (unsigned)bp, (unsigned)(bp >> 32)};
            r[3] = u32x4{(unsigned)ty | ((unsigned)tx << 16), (unsigned)g, (unsigned)(a.Cout_g - cg * 64), (unsigned)cbg};
        }
    };
    auto load_tile = [&](int e) __attribute__((always_inline)) {         // uniform: every lane reads the same entry
        const u32x4* r = ttab + (e & (TT - 1)) * 4;
        const u32x4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
        auto sg = [](unsigned v) __attribute__((always_inline)) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
        auto p64 = [&](unsigned lo, unsigned hi) __attribute__((always_inline)) { return ((unsigned long long)sg(hi) << 32) | (unsigned long long)sg(lo); };
        TileP t;
        t.in0n = (const float*)p64(r0[0], r0[1]); t.in1n = (const float*)p64(r0[2], r0[3]);
        t.wb = (const u32x4*)p64(r1[0], r1[1]); t.ob = (float*)p64(r1[2], r1[3]);
        t.rb = (const float*)p64(r2[0], r2[1]); t.bp = (const float*)p64(r2[2], r2[3]);
        const unsigned yx = sg(r3[0]);
        t.ty = (int)(yx & 0xffffu); t.tx = (int)(yx >> 16); t.clg = (int)sg(r3[2]);
        if constexpr (CHAIN) {                           // one group, one cout group: the two words carry the layer's epilogue and the tile's ticket
            const unsigned ar = sg(r3[1]);
            t.g = 0; t.cbg = 0; t.act = (int)(ar & 255u); t.rm = (int)((ar >> 8) & 255u); t.tk = (int)sg(r3[3]);
        } else { t.g = (int)sg(r3[1]); t.cbg = (int)sg(r3[3]); t.act = a.act; t.rm = a.res_mode; t.tk = 0; }
        return t;
    };

    // ---- CHAIN mode: tickets, table entries, completion counters.  Wave 0 does all of it with SCALAR instructions; the other waves learn the
    // outcome from LDS behind a barrier.  Nothing here may touch the vector-memory queue: vmcnt retires in order, so a value loaded there is
    // only readable once every row piece and weight fragment requested before it has returned -- the first version (vector atomic for the
    // ticket, nine flag loads) stalled wave 0 for 2.9 k cycles per tile at its two consumption points.  s_atomic_add ... glc returns to an
    // SGPR and is counted by lgkmcnt (tools/ubench_chain.hip: one counter for all 8 XCDs, coherent read-back by adding 0).  Issue and use
    // are a chunk apart, so the result lands in a MAILBOX register the compiler never allocates: s100 / s101 (hipcc's allocatable range on
    // gfx950 ends at s99; tests/test_isa_hygiene.py holds it to that -- a compiler-visible destination could be copied or spilled before the
    // value has arrived).
    // Completion is counted per (layer, image, tile row) in ONE word per row that also carries its two neighbours: a finished tile of row y
    // adds 1 << 10 to word y, 1 to word y-1 and 1 << 20 to word y+1, so tile (l, n, y, x) may start when word y of (l-1, n) reads
    // tiles_x in every field whose row exists -- one scalar read per poll.
    int* chs = (int*)((float*)lds_raw + 32);             // spare words of wave 0's bias area: [0] pending ticket, [1] its state (0 none left, 1 ready, 2 not ready yet, 3 aborted)
    u32x4* ltab = ttab + TT * 4;                         // [L][4]: the layers resolved (see chain_entry)
    const int chain_total = CHAIN ? ch.L * ntiles : 0;   // < 2^24 (host)
    const unsigned long long ws64 = (unsigned long long)ch.ws;
    // The lane index, recomputed where it is used (two v_mbcnt): OPAQUE, so a lane test is never merged with another test of the lane index
    // across a barrier (tools/ubench_chain.hip: hipcc threaded two `tid == 0` blocks round a loop and split wave 0 at the barriers), and
    // never a long-lived register -- a copy kept for the whole kernel was spilled, and its reload is a scratch load + s_waitcnt vmcnt(0):
    // a wait for every store of the tile just finished (2 k cycles at each tile end, tools/trace_wino.py CHAIN=..)
    auto lane_now = []() __attribute__((always_inline)) {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
#define CHAIN_SADD(REG, BYTEOFF, VAL) asm volatile("s_mov_b32 " REG ", %2\n\ts_atomic_add " REG ", %0, %1 glc" :: "s"(ws64), "s"(BYTEOFF), "s"(VAL) : "memory", REG)
#define CHAIN_STAKE(REG, OUT) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, " REG : "=s"(OUT) :: "memory")
#define CHAIN_SSET(REG, VAL) asm volatile("s_mov_b32 " REG ", %0" :: "s"(VAL) : REG)
    auto chain_ticket_issue = [&]() __attribute__((always_inline)) { CHAIN_SADD("s100", 0, 1); };
    auto chain_ticket_take = [&]() __attribute__((always_inline)) { int tk; CHAIN_STAKE("s100", tk); return tk; };
    int pr_row = -1, pr_ty = 0;                          // of the tile last decoded: its own row word in the layer before (-1: layer 0), its tile row
    int e_row = 0;                                       // ... its own row word (what it adds to when it is finished)
    int dec_tk = -1, dec_l = 0, dec_r = 0;               // the last ticket decoded: tickets only grow, so (layer, tile) advance by the difference -- no division
    // LDS tables built at kernel start: ltab[l] = the layer RESOLVED, 4 quads: {src base lo, hi, batch stride in bytes, act | res_mode << 8}
    // {dst base, stride, 0} {res base (0: none), stride, 0} {weights lo, hi, bias lo, hi};  dtab[tile of a layer] = ty | tx << 10 | image << 20
    const unsigned* dtab = (const unsigned*)(ltab + 4 * (CHAIN ? ch.L : 0));
    // One table entry = five 8-byte fields and one quad, each formed by ITS OWN LANE (base + image * stride is one v_mad_u64_u32): a lone
    // wave issues a dependent instruction every ~8 cycles, and the scalar form of this -- three 64-bit multiply-adds, the buffer selects,
    // sixteen moves -- was a chain of 1.9 k cycles at the end of every tile (tools/trace_wino.py CHAIN=..).
    auto chain_entry = [&](int e, int tk) __attribute__((always_inline)) {      // tk uniform
        int l, r;
        if (dec_tk < 0) { l = tk / ntiles; r = tk - l * ntiles; }
        else { l = dec_l; r = dec_r + (tk - dec_tk); while (r >= ntiles) { r -= ntiles; ++l; } }
        dec_tk = tk; dec_l = l; dec_r = r;
        const int jl = lane_now();
        const unsigned dwv = dtab[r];                     // the two LDS reads are independent: one round trip
        const u32x4 q = ltab[4 * l + (jl < 3 ? jl : jl < 5 ? 3 : 0)];
        const unsigned dw = (unsigned)__builtin_amdgcn_readfirstlane((int)dwv);
        const int ty = (int)(dw & 1023u), tx = (int)((dw >> 10) & 1023u), n = (int)(dw >> 20);
        e_row = (l * a.N + n) * tiles_y + ty;
        pr_row = l > 0 ? e_row - a.N * tiles_y : -1;
        pr_ty = ty;
        const unsigned long long base = ((unsigned long long)q[1] << 32) | q[0];
        unsigned long long v64 = base + (unsigned long long)(unsigned)n * (unsigned long long)q[2];
        if (jl == 2 && base == 0ull) v64 = 0ull;
        if (jl == 3) v64 = base;
        if (jl == 4) v64 = ((unsigned long long)q[3] << 32) | q[2];
        if (jl == 6) v64 = 0ull;                          // (no second source in a chain)
        char* ent = (char*)(ttab + (e & (TT - 1)) * 4);
        const int wr = jl == 0 ? 0 : jl == 1 ? 24 : jl == 2 ? 32 : jl == 3 ? 16 : jl == 4 ? 40 : 8;
        if (jl < 5 || jl == 6) *(unsigned long long*)(ent + wr) = v64;
        if (jl == 5) ((u32x4*)ent)[3] = u32x4{(unsigned)ty | ((unsigned)tx << 16), q[3], (unsigned)a.Cout_g, (unsigned)e_row};      // [3]: the tile's own completion word
    };
    // the pending tile's row word of the layer before -> mailbox s101 (layer 0: the expected value at once)
    auto chain_rows_want = [&]() __attribute__((always_inline)) {
        return (pr_ty + 1 < tiles_y ? a.tiles_x : 0) | (a.tiles_x << 10) | (pr_ty > 0 ? a.tiles_x << 20 : 0);
    };
    auto chain_rows_issue = [&]() __attribute__((always_inline)) {
        if (pr_row >= 0) CHAIN_SADD("s101", (64 + pr_row) * 4, 0); else CHAIN_SSET("s101", chain_rows_want());
    };
    auto chain_rows_take = [&]() __attribute__((always_inline)) {
        int c1;
        CHAIN_STAKE("s101", c1);
        return c1 == chain_rows_want();
    };
    // between tiles only (nothing of this workgroup in flight, so waiting here can block nobody this workgroup could unblock); bounded: a chain
    // in which NO tile of ANY workgroup has been published for a second (word 2 of the workspace counts published tiles: the clock restarts
    // whenever it has moved, so a workgroup that merely waits long for its turn -- pre-emption, a debugger, another stream's kernels on the
    // CUs -- does not give up while the chain advances) sets the abort word and status bit 1 instead of hanging the device.  Without a status
    // word the launch traps: an abandoned chain must never look like a finished one.
    auto chain_wait = [&]() __attribute__((always_inline)) {
        unsigned t0 = (unsigned)__builtin_amdgcn_s_memrealtime();     // 100 MHz; the low word (differences are taken modulo 2^32 = 43 s)
        int prog = -1;
        auto give_up = [&]() __attribute__((always_inline)) {
            if (lane_now() == 0) {
                __hip_atomic_store(ch.ws + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.status) atomicOr(a.status, 2u); else asm volatile("s_trap 2");      // (not __builtin_trap: a no-return call in a divergent branch makes every value live across it a vector value)
            }
        };
        if (a.dbg & 128) { give_up(); return false; }    // option conv_dbg bit 7: every workgroup gives up at its first tile -- the test of this path
        for (;;) {
            chain_rows_issue();
            if (chain_rows_take()) return true;
            int ab, pg;
            CHAIN_SADD("s101", 4, 0);                    // (s100 may hold a ticket in flight)
            CHAIN_STAKE("s101", ab);
            if (ab) return false;
            CHAIN_SADD("s101", 8, 0);
            CHAIN_STAKE("s101", pg);
            const unsigned now = (unsigned)__builtin_amdgcn_s_memrealtime();
            if (pg != prog) { prog = pg; t0 = now; }
            if (now - t0 > 100000000u) { give_up(); return false; }
            __builtin_amdgcn_s_sleep(4);
        }
    };
    int cur_row = 0, cur_ty = 0, pend_row = 0, pend_ty = 0, fut_row = 0, fut_ty = 0;     // completion word / tile row of the current tile, the next, the one after
    auto chain_publish = [&](int row, int ty) __attribute__((always_inline)) {      // lanes 0 .. 2: words row - 1, row, row + 1; lane 3: the progress count (one atomic instruction)
        const int lc = lane_now();
        const bool on = lc < 4 && (lc != 0 || ty > 0) && (lc != 2 || ty + 1 < tiles_y);
        const int word = lc == 3 ? 2 : 64 + row + lc - 1;                   // (an index on the uniform base: nothing 64-bit per lane to keep alive over the tile loop)
        if (on) __hip_atomic_fetch_add(ch.ws + word, lc == 0 ? 1u : lc == 1 ? 1u << 10 : lc == 2 ? 1u << 20 : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // activations / residuals of a chain come from other XCDs inside this launch: device-coherent (sc1) accesses
    auto rload = [&](const float* base, unsigned off_elems) __attribute__((always_inline)) {
        if constexpr (CHAIN) {
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)(off_elems * 4u), 0, (CHAIN_ABL & 2) ? 0 : 16));
        } else return gload((const f32x4*)(base + off_elems));
    };

    // ---- per-lane constants of the staging role (row pair T = wave) ------------------------------------------------------
    const int mrd = half * 8 * 40 + 4 + l31;                           // main item: octet `half`, pixel 1 + l31 (landing column 4 + l31)
    const int mslot = half * OCT + (wave * 4) * PW + 1 + l31;          // its staging slot at position 0
    const int h_side = (lane >> 4) & 1, h_oct = lane >> 5, h_pos = (lane >> 2) & 3, h_pair = lane & 3;   // halo item: one (position, channel pair)
    const int h_ra = h_pos == 0 ? 0 : h_pos == 2 ? 2 : 1, h_rb = h_pos == 0 ? 2 : h_pos == 2 ? 1 : h_pos == 1 ? 2 : 3;
    const float h_sgn = h_pos == 1 ? 1.f : -1.f;
    const int h_rda = ((h_ra * 2 + h_oct) * 8 + 2 * h_pair) * 40 + (h_side ? 36 : 3);
    const int h_rdb = ((h_rb * 2 + h_oct) * 8 + 2 * h_pair) * 40 + (h_side ? 36 : 3);
    const int h_wr = (h_oct * OCT + (wave * 4 + h_pos) * PW + (h_side ? 33 : 0)) * 4 + h_pair;   // dword index within a part
    // row piece i of a step: quad qd = i * 64 + lane = ((row * 2 + octet) * 8 + channel) * 10 + column quad.  pl = its byte offset from
    // the tile's first staged pixel of the chunk's first plane, rx = (row | quad << 2) of the ten pieces packed six bits each.
    int pl[NLD];
    unsigned rx0 = 0, rx1 = 0;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int qd = i * 64 + lane;
        const int xq = qd % 10, r = qd / 10, ch = r & 7, oct = (r >> 3) & 1, row = r >> 4;
        pl[i] = ((8 * oct + ch) * HW + row * a.W + 4 * xq) * 4;
        const unsigned f6 = (unsigned)(row | (xq << 2));
        if (i < 5) rx0 |= f6 << (6 * i); else rx1 |= f6 << (6 * (i - 5));
    }

    // ---- load plan of the tile the row-piece requests currently target -----------------------------------------------------
    constexpr bool POFF = NP == 2;                       // the ten request offsets of the target tile held in registers (the three-part form has none to spare)
    int po[POFF ? NLD : 1] = {};
    unsigned vmask = 0;                                  // bit i: row piece i lies inside the image
    int origin = 0;                                      // byte offset of the tile's first staged pixel within a plane (uniform)
    const float* in0n = nullptr; const float* in1n = nullptr;
    int st_g = 0;
    auto wptr = [&](const TileP& t) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)t.wb, 0, 0x7fffffff, 0x00020000);
    };
    const int wvoff = (ct * 64 + lane) * 16;
    auto wfrag = [&](__amdgpu_buffer_rsrc_t wb, int ks, int p) __attribute__((always_inline)) {
        return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wb, wvoff, (ks * NP + p) * (2 * 64 * 16), 0));
    };
    auto bias_of = [&](const TileP& t) __attribute__((always_inline)) {
        if constexpr (TR) return (t.bp && ct * 32 + l31 < t.clg) ? gload(t.bp + ct * 32 + l31) : 0.f;      // lane = cout in both half-waves
        else return (t.bp && lane < 32 && ct * 32 + lane < t.clg) ? gload(t.bp + ct * 32 + lane) : 0.f;
    };
    // the accumulators of the two-part form carry 2^8 x the sums (exact).  Scaled where the value is STORED for init_acc, not where it is
    // requested: arithmetic on it at the request would wait for every vector-memory request in flight (vmcnt is one in-order queue)
    auto bias_scaled = [](float b) __attribute__((always_inline)) { return NP == 2 ? b * kWinoF16Scale : b; };
    auto setup_loads = [&](const TileP& t, bool valid) __attribute__((always_inline)) {
        st_g = t.g;
        in0n = t.in0n;
        in1n = t.in1n;
        const int iy0 = t.ty * TH - 1 + 2 * wave, x0 = t.tx * 32 - 4;  // pad 1; rows start 4 pixels left of the tile: aligned quads
        const int rlo = iy0 < 0 ? -iy0 : 0, rhi = valid ? (a.H - iy0 < 4 ? a.H - iy0 : 4) : 0;
        const int qlo = x0 < 0 ? 1 : 0, qhi = (a.W - x0) >> 2;         // W % 4 == 0, x0 % 4 == 0
        origin = (iy0 * a.W + x0) * 4;
        if (valid && iy0 >= 0 && iy0 + 4 <= a.H && x0 >= 0 && x0 + 40 <= a.W) vmask = 0x3ffu;   // interior tile: every piece inside
        else {
            vmask = 0;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const unsigned f6 = ((i < 5 ? rx0 >> (6 * i) : rx1 >> (6 * (i - 5)))) & 63u;
                const int row = (int)(f6 & 3u), xq = (int)(f6 >> 2);
                vmask |= (row >= rlo && row < rhi && xq >= qlo && xq < qhi) ? 1u << i : 0u;
            }
        }
        if constexpr (POFF) {                            // once per tile instead of four vector instructions per request and chunk
#pragma unroll
            for (int i = 0; i < NLD; ++i) po[i] = (vmask >> i) & 1u ? pl[i] + origin : (int)0x80000000;
        }
    };
    // ---- the requests and pieces of staging one 16-channel chunk -----------------------------------------------------------
    f32x4 gq[NLD] = {};
    auto st_load = [&](int i, int c0) __attribute__((always_inline)) {                  // global -> registers: 4 pixels of one channel of one row
        const int gch0 = st_g * a.Cin_g + c0;
        const bool first = gch0 < a.C0;
        const float* base = first ? in0n + (long)gch0 * HW : in1n + (long)(gch0 - a.C0) * HW;
        const int nchv = (first && in1n) ? a.C0 - c0 : a.Cin_g - c0;          // channels from c0 to the end of the group / of the first source
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nchv * HW * 4, 0x00020000);
        // BYTE offset from the chunk's first plane; 2^31 = "outside the image": beyond any descriptor's range, the load returns 0
        const int off = POFF ? po[POFF ? i : 0] : ((vmask >> i) & 1u ? pl[i] + origin : (int)0x80000000);
        gq[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, (CHAIN && !(CHAIN_ABL & 2)) ? 16 : 0));
    };
    auto st_park = [&](int i) __attribute__((always_inline)) { *(f32x4*)(land + i * 64 + lane) = gq[i]; };
    float sv[4][8] = {};                                   // [input row of the pair][channel of the octet]
    auto st_read = [&](int r) __attribute__((always_inline)) {                          // piece r: row r/4, channels 2(r%4), 2(r%4)+1
        const int row = r >> 2, q = (r & 3) * 2;
        sv[row][q] = landf[mrd + ((row * 2) * 8 + q) * 40];
        sv[row][q + 1] = landf[mrd + ((row * 2) * 8 + q + 1) * 40];
    };
    float lo_scale = kWinoLoScale;                      // in a scalar register (the mix instructions take no literal)
    asm volatile("" : "+s"(lo_scale));
    float xv[8] = {}, xe[8] = {};
    unsigned xpk[4] = {};
    u32x4 sparts[NP] = {};
    auto st_x = [&](int idx) __attribute__((always_inline)) {     // transform + split of one position, stage idx & 15 (four independent instructions each)
        const int pos = idx >> 4, st = idx & 15;
        auto expand = [&](int h) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 2 * h; q < 2 * h + 2; ++q) { xe[2 * q] = bf_lo(xpk[q]); xe[2 * q + 1] = bf_hi(xpk[q]); }
        };
        auto sub = [&](int h) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 4 * h; e < 4 * h + 4; ++e) xv[e] = fsub1(xv[e], xe[e]);
        };
        auto pack = [&](int p) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { xpk[q] = NP == 2 ? pk_f16(xv[2 * q], xv[2 * q + 1]) : pk_bf16(xv[2 * q], xv[2 * q + 1]); sparts[p][q] = xpk[q]; }
        };
        if (st < 2) {
            const int ra = pos == 0 ? 0 : pos == 2 ? 2 : 1, rb = pos == 0 ? 2 : pos == 2 ? 1 : pos == 1 ? 2 : 3;
#pragma unroll
            for (int e = 4 * st; e < 4 * st + 4; ++e) xv[e] = pos == 1 ? fadd1(sv[ra][e], sv[rb][e]) : fsub1(sv[ra][e], sv[rb][e]);
        } else if (st == 2) pack(0);
        else if constexpr (NP == 2) {
            if (st == 3 || st == 4) {                     // remainder after the first part: one mixed-precision FMA per value
#pragma unroll
                for (int q = 2 * (st - 3); q < 2 * (st - 3) + 2; ++q) { xv[2 * q] = sub_f16_lo(xv[2 * q], xpk[q]); xv[2 * q + 1] = sub_f16_hi(xv[2 * q + 1], xpk[q]); }
            } else if (st == 5) {                         // low part times 2^11, one rounding (see WOrder<2>)
#pragma unroll
                for (int q = 0; q < 4; ++q) pk_f16_scaled_lo(xpk[q], xv[2 * q], lo_scale);
            } else if (st == 6) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { pk_f16_scaled_hi(xpk[q], xv[2 * q + 1], lo_scale); sparts[1][q] = xpk[q]; }
            }
        } else {
            if (st == 3 || st == 4) expand(st - 3);
            else if (st == 5 || st == 6) sub(st - 5);
            else if (st == 7) pack(1);
            else if (st == 8 || st == 9) expand(st - 8);
            else if (st == 10 || st == 11) sub(st - 10);
            else if (st == 12) pack(2);
        }
    };
    auto st_w = [&](int idx, u32x4* dstbuf) __attribute__((always_inline)) {            // one part of position idx / 3 -> staging buffer
        const int pos = idx / NP, p = idx - pos * NP;
        dstbuf[p * SLOTS + mslot + pos * PW] = sparts[p];
    };
    float ha0 = 0.f, ha1 = 0.f, hb0 = 0.f, hb1 = 0.f, hv0 = 0.f, hv1 = 0.f, he0 = 0.f, he1 = 0.f;
    unsigned hpk = 0, hparts[NP] = {};
    auto st_hread = [&](int r) __attribute__((always_inline)) {
        if (r == 0) { ha0 = landf[h_rda]; ha1 = landf[h_rda + 40]; }
        else { hb0 = landf[h_rdb]; hb1 = landf[h_rdb + 40]; }
    };
    auto st_hx = [&](int h) __attribute__((always_inline)) {
        if constexpr (NP == 2) {
            if (h == 0) {
                hv0 = __builtin_fmaf(h_sgn, hb0, ha0); hv1 = __builtin_fmaf(h_sgn, hb1, ha1);   // a +- b: the product by +-1 is exact
                hpk = pk_f16(hv0, hv1);
                hparts[0] = hpk;
            } else if (h == 1) {
                hv0 = sub_f16_lo(hv0, hpk); hv1 = sub_f16_hi(hv1, hpk);
            } else { pk_f16_scaled_lo(hpk, hv0, lo_scale); pk_f16_scaled_hi(hpk, hv1, lo_scale); hparts[1] = hpk; }
        } else if (h == 0) {
            hv0 = __builtin_fmaf(h_sgn, hb0, ha0); hv1 = __builtin_fmaf(h_sgn, hb1, ha1);   // a +- b: the product by +-1 is exact
            hpk = pk_bf16(hv0, hv1);
            hparts[0] = hpk;
        } else if (h == 1) {
            hv0 = fsub1(hv0, bf_lo(hpk)); hv1 = fsub1(hv1, bf_hi(hpk));
        } else if (h == 2) {
            hpk = pk_bf16(hv0, hv1);
            hparts[1] = hpk;
            he0 = bf_lo(hpk); he1 = bf_hi(hpk);
        } else {
            hparts[NP - 1] = pk_bf16(fsub1(hv0, he0), fsub1(hv1, he1));
        }
    };
    auto st_hw = [&](int p, u32x4* dstbuf) __attribute__((always_inline)) { ((unsigned*)(dstbuf + p * SLOTS))[h_wr] = hparts[p]; };
    auto piece = [&](auto ec, u32x4* dstbuf) __attribute__((always_inline)) {
        constexpr int e = decltype(ec)::value, kind = e >> 8, idx = e & 255;
        if constexpr (kind == WP_PARK) { if constexpr (!(WINO_ABL & 2)) st_park(idx); }
        else if constexpr (kind == WP_READ) { if constexpr (!(WINO_ABL & 4)) st_read(idx); }
        else if constexpr (kind == WP_READ2) { if constexpr (!(WINO_ABL & 4)) { st_read(2 * idx); st_read(2 * idx + 1); } }
        else if constexpr (kind == WP_X) { if constexpr (!(WINO_ABL & 8)) st_x(idx); }
        else if constexpr (kind == WP_W) { if constexpr (!(WINO_ABL & 16)) st_w(idx, dstbuf); }
        else if constexpr (kind == WP_HREAD) { if constexpr (!(WINO_ABL & 32)) st_hread(idx); }
        else if constexpr (kind == WP_HX) { if constexpr (!(WINO_ABL & 32)) st_hx(idx); }
        else if constexpr (kind == WP_HW) { if constexpr (!(WINO_ABL & 32)) st_hw(idx, dstbuf); }
    };

    // Residual quads of the tile's first two epilogue passes, REQUESTED UNDER ITS LAST CHUNK (round 5, row-major form).  tools/trace_wino.py:
    // requested at the start of the epilogue, the 16 quads first stalled 1.8 k cycles at issue behind the ~20 KB of next-tile row pieces and
    // weight fragments in flight, then pass 0 waited another 1.8 k for them (in-order return): 9.5 k instead of 5.8 k cycles of epilogue.
    // Eight quads fit the registers the main loop leaves free; they go into four request-free even slots of super-step 3, where every later
    // request of the chunk belongs to the NEXT tile (its weight fragments and row pieces are consumed after this tile's epilogue), so
    // nothing the MFMA stream is about to need queues behind a cold residual line.  Passes 2 / 3 are requested in the epilogue as before.
    f32x4 rpre[2][4] = {};
    bool rpre_on = false;                                // uniform: the current chunk is the last of a tile that has a residual (and couts)
    const float* rpre_base = nullptr;                    // uniform
    unsigned rpre_lb = 0;                                // per lane: offset of pass 0 / item 0 (see finish_tile)
    int rpre_cl = 0;                                     // uniform: valid couts of this wave's cout tile
    bool rpre_okl = false, rpre_full = false;            // per lane / uniform
    f32x16 acc[2][4];                                    // [row pair of this wave][position]
    constexpr int WB = 3;                                // weight fragment sets: a set is refilled for three super-steps ahead while in use
    u32x4 wf[WB][2][NP];                                 // [set][position of the pair][weight part]
    u32x4 bfr[NP][4];                                    // [activation part][position of the pair * 2 + row pair]: ONE copy
    u32x4 ws[2] = {};                                    // two-part form: 2^-11 x the high weight fragment of the current super-step, per position of the pair
    f16x2v ws_c = {(_Float16)(1.f / kWinoLoScale), (_Float16)(1.f / kWinoLoScale)};
    asm volatile("" : "+v"(ws_c));
    __amdgpu_buffer_rsrc_t wbase, wnext;
    auto wks = [](int ss, int pi) __attribute__((always_inline)) { return (2 * (ss / 3) + pi) * 3 + ss % 3; };   // packed k-step (position, kx) within a chunk
    auto loadw = [&](__amdgpu_buffer_rsrc_t wb, int ss, u32x4 (&dst)[2][NP]) __attribute__((always_inline)) {
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
            for (int p = 0; p < NP; ++p) dst[pi][p] = wfrag(wb, wks(ss, pi), p);
    };
#ifdef MOTIF_TRACE_SS
    int trace_chunk = 0;
#endif

    // Chunk c of the current tile on staging buffer `buf` (ONE instantiation: a second body -- say with the constant 0 as C of each
    // accumulator's first product -- gets its own accumulator registers, and 128 of them are copied at every transition).
    // sc / wn: chunk index and weights of the NEXT step (staged into the other buffer; its first three super-steps' weight fragments
    // are requested during super-steps 3 .. 5); lc0: first channel of the step after next (row pieces, per the load plan).
    auto chunk_body = [&](int c, int buf, int sc, __amdgpu_buffer_rsrc_t wn, int lc0) __attribute__((always_inline)) {
        const u32x4* pb = stg0 + buf * STG + half * OCT + (tp * 8) * PW + l31;
        u32x4* dstbuf = stg0 + (buf ^ 1) * STG;
        auto loadb = [&](int ss, int p, int j) __attribute__((always_inline)) {
            bfr[p][j] = pb[p * SLOTS + ((j & 1) * 4 + 2 * (ss / 3) + (j >> 1)) * PW + (ss % 3)];
        };
        if constexpr (!(WINO_ABL & 64)) {
#pragma unroll
            for (int p = NP - 1; p >= 0; --p)
#pragma unroll
                for (int j = 0; j < 4; ++j) loadb(0, p, j);
        }
#ifdef MOTIF_TRACE_SS
        long long ts[7];
#endif
        static_for<SS * M>([&](auto ic) __attribute__((always_inline)) {
            constexpr int s = decltype(ic)::value, ss = s / M, m = s % M, k = m >> 2, j = m & 3, pi = j >> 1, tl = j & 1, pos = 2 * (ss / 3) + pi;
#ifdef MOTIF_TRACE_SS
            if constexpr (m == 0) ts[ss] = __builtin_amdgcn_s_memtime();
#endif
            if constexpr (!(WINO_ABL & 1024) && (!(WINO_ABL & 512) || (k & 1)))
            {
                if constexpr (NP == 2 && TR) acc[tl][pos] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bfr[WO::x[k]][j]),
                                                                                                  __builtin_bit_cast(f16x8, WO::w[k] == 2 ? ws[pi] : wf[ss % WB][pi][WO::w[k] & 1]), acc[tl][pos], 0, 0, 0);
                else if constexpr (NP == 2) acc[tl][pos] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, WO::w[k] == 2 ? ws[pi] : wf[ss % WB][pi][WO::w[k] & 1]),
                                                                                            __builtin_bit_cast(f16x8, bfr[WO::x[k]][j]), acc[tl][pos], 0, 0, 0);
                else acc[tl][pos] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[ss % WB][pi][WO::w[k]]),
                                                                            __builtin_bit_cast(bf16x8, bfr[WO::x[k]][j]), acc[tl][pos], 0, 0, 0);
            }
            // two-part form: quad m & 3 of the scaled high weight fragment of position m >> 2, one v_pk_mul_f16 per slot, in front of the
            // request that refills the fragment's registers (slots 5 / 7) and of the products that read it (MFMAs 8 .. 11)
            if constexpr (NP == 2 && m < 8) ws[m >> 2][m & 3] = pk_mul_f16(wf[ss % WB][m >> 2][0][m & 3], ws_c);
            if constexpr (kWSchedOf<NP>.rb[s] >= 0 && !(WINO_ABL & 64)) loadb(ss + 1, kWSchedOf<NP>.rb[s] >> 2, kWSchedOf<NP>.rb[s] & 3);
            if constexpr (kWSchedOf<NP>.ra[s] >= 0 && !(WINO_ABL & 128)) {
                constexpr int qi = kWSchedOf<NP>.ra[s] / NP, p = kWSchedOf<NP>.ra[s] % NP;
                if constexpr (ss + 3 < SS) wf[ss % WB][qi][p] = wfrag(wbase, c * 12 + wks(ss + 3, qi), p);
                else wf[ss % WB][qi][p] = wfrag(wn, sc * 12 + wks(ss + 3 - SS, qi), p);
            }
            if constexpr (kWSchedOf<NP>.rl[s] >= 0 && !(WINO_ABL & 1)) st_load(kWSchedOf<NP>.rl[s], lc0);
            if constexpr (NP == 2 && !TR && (s == 36 || s == 40 || s == 42 || s == 46)) {
                if (rpre_on) {                           // uniform branch; two residual quads per slot
                    constexpr int b0 = s == 36 ? 0 : s == 40 ? 2 : s == 42 ? 4 : 6;
#pragma unroll
                    for (int e = b0; e < b0 + 2; ++e) {
                        const int pass = e >> 2, it = e & 3;
                        const bool okq = (bool)((int)rpre_full | (int)(rpre_okl && 8 * pass + 2 * it + half < rpre_cl));    // bitwise: see the note at the transposed form's loads
                        rpre[pass][it] = rload(rpre_base, okq ? rpre_lb + (unsigned)(8 * pass + 2 * it) * HWo : 0u);
                    }
                }
            }
            piece(std::integral_constant<int, kWSchedOf<NP>.ext[s]>{}, dstbuf);
            __builtin_amdgcn_sched_barrier(0);
        });
#ifdef MOTIF_TRACE_SS
        ts[6] = __builtin_amdgcn_s_memtime();
        if (lane == 0 && blockIdx.x < 256 && trace_chunk < 8) {
#pragma unroll
            for (int i = 0; i < 7; ++i) g_wn_trace2[((blockIdx.x * 4 + wave) * 8 + trace_chunk) * 8 + i] = ts[i];
        }
        ++trace_chunk;
#endif
    };
    int ti = 0;                                          // ordinal of the current tile in this workgroup
    // Accumulators of a new tile: zero, except position 1, which starts from the bias (M1 enters both output rows of a pair with +1:
    // out(2T) = M0 + M1 + M2, out(2T+1) = M1 - M2 - M3) -- the zeros written by SIX matrix instructions with zero operands and C = 0
    // instead of 96 v_accvgpr_write (one issue slot each, beside the epilogue's vector work), the bias by 32 moves instead of 64
    // additions in the epilogue.  The wave's bias lies in LDS (bias_w, written by its lanes 0..31 just before: same wave, in order).
    float bias_lane = 0.f;                               // TR: this lane's (cout's) bias of the tile about to start, already scaled
    auto init_bias = [&]() __attribute__((always_inline)) {
        f32x16 bvec;
        if constexpr (TR) {
#pragma unroll
            for (int r = 0; r < 16; ++r) bvec[r] = bias_lane;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *(const f32x4*)(bias_w + 8 * q + 4 * half);
#pragma unroll
                for (int u = 0; u < 4; ++u) bvec[4 * q + u] = b4[u];
            }
        }
        return bvec;
    };
    auto init_acc_from = [&](const f32x16& bvec) __attribute__((always_inline)) {
        // zq: a register quad of zeros written once at kernel start (opaque to the compiler: it cannot re-materialise it right in front
        // of an instruction whose operand hazards it does not know); s_nop: wait states of a just-written operand, whatever wrote it
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (p == 1) acc[tl][p] = bvec;           // (C and D of an MFMA share a register class: the bias vector goes in by plain moves)
                else if constexpr (NP == 2) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %1, 0" : "=a"(acc[tl][p]) : "v"(zq));
                else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, 0" : "=a"(acc[tl][p]) : "v"(zq));
            }
    };
    auto init_acc = [&]() __attribute__((always_inline)) { init_acc_from(init_bias()); };
    // After a tile's last chunk: inverse transform + bias in the C/D layout, then four passes of 8 couts x 4 rows x 32 pixels through
    // the landing area (free between the last read-back and the next chunk's parks): lane item it of a pass = cout half + 2 it, row
    // l31 / 8, columns 4 (l31 % 8) .. + 3 -- residual, activation, one 16-byte store.
    auto finish_tile = [&](const TileP& t) __attribute__((always_inline)) {
        const int ty = t.ty, tx = t.tx;
        const int cb = t.cbg + ct * 32, cl = t.clg - ct * 32;
        if (cl <= 0 || (WINO_ABL & 256)) return;        // the upper cout tile of a partial group has nothing to store (the caller re-initialises the accumulators)
#ifdef MOTIF_TRACE
        long long es[8];
        int esn = 0;
        es[esn++] = __builtin_amdgcn_s_memtime();
#define EPSTAMP() do { if (esn < 8) es[esn++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define EPSTAMP()
#endif
        const unsigned long long obp = (unsigned long long)(t.ob + (long)(ct * 32) * (long)HWo);            // uniform: the stores take it in SGPRs
        const float* rb = t.rb ? t.rb + (long)(ct * 32) * (long)HWo : nullptr;
        const int rm = CHAIN ? t.rm : a.res_mode;
        if constexpr (TR) {
            // Transposed accumulators: lane = cout l5 of this wave's cout tile (both half-waves), register r = pixel (r & 3) + 8 (r >> 2) + 4 hf of
            // the 32-pixel row: registers 4q .. 4q+3 are the four consecutive pixels 8q + 4hf .. + 3, i.e. ONE 16-byte piece of an NCHW row.
            // Per lane 2 row pairs x 2 rows x 4 pieces = 16 pieces: residual pieces requested first, inverse transform / scale / residual /
            // activation in registers, 16-byte stores as inline assembly (see the row-major form below for the wait-count reasons).
            int lane_t = lane;                           // opaque copy: keeps the per-lane geometry inside the tile loop
            asm volatile("" : "+v"(lane_t));
            const int hf_t = lane_t >> 5, co = lane_t & 31;
            const bool cok = co < cl;
            const int oyb = ty * TH + 4 * tp, oxb = tx * 32 + 4 * hf_t;
            const unsigned lbase = (unsigned)co * HWo + (unsigned)(oyb * a.Wo + oxb);
            const bool fullt = cl >= 32 && oyb + 4 <= a.Ho && tx * 32 + 32 <= a.Wo;                 // uniform: every lane stores every piece
            auto okp = [&](int row, int q) __attribute__((always_inline)) { return cok && oyb + row < a.Ho && oxb + 8 * q < a.Wo; };      // Wo % 4 == 0: a piece is inside or outside as a whole
            auto offp = [&](int row, int q) __attribute__((always_inline)) { return lbase + (unsigned)(row * a.Wo + 8 * q); };
            f32x4 rv[4][4];
            if (rm) {
#pragma unroll
                for (int row = 0; row < 4; ++row)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // branch-free on purpose (bitwise |): with the short-circuit form hipcc 7.2 dropped the per-lane select of the FIRST piece
                        // on the not-full path (an empty exec region in the ISA: every lane read element 0 there)
                        const bool okl = (bool)((int)fullt | (int)okp(row, q));
                        rv[row][q] = gload((const f32x4*)(rb + (okl ? offp(row, q) : 0u)));       // masked pieces read element 0
                    }
            }
            const int act = a.act;
            // the scalar operands of the inline-assembly stores, pinned to scalar registers (the compiler is free to keep a uniform 64-bit
            // value in vector registers, which the `s` constraint then receives as they are)
            auto sgpr64 = [](unsigned long long v) __attribute__((always_inline)) {
                return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
            };
            const unsigned long long obq = sgpr64(obp), stp = sgpr64((unsigned long long)a.status);
            float chk = 0.f;                             // range status: 0 * value accumulates NaN for any non-finite output of this lane
            // AC / RM >= 0: activation / residual mode known at compile time (the common layers: one straight-line body -- 16 pieces x a chain of
            // wave-uniform tests is what a lone wave cannot hide); -1: run-time switches
            auto pieces = [&](auto ac_tag, auto rm_tag) __attribute__((always_inline)) {
            constexpr int AC = decltype(ac_tag)::value, RM = decltype(rm_tag)::value;
            const int rmv = RM >= 0 ? RM : rm, acv = AC >= 0 ? AC : act;
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = 2 * tl + j;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int r = 4 * q + u;
                            v[u] = j == 0 ? (acc[tl][0][r] + acc[tl][1][r]) + acc[tl][2][r]         // the bias is inside M1 (init_acc)
                                          : (acc[tl][1][r] - acc[tl][2][r]) - acc[tl][3][r];
                        }
                        if constexpr (NP == 2) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) chk = __builtin_fmaf(v[u], 0.f, chk);
                            v *= 1.f / kWinoF16Scale;
                        }
                        if (rmv == 1) v += rv[row][q];
                        if (acv == MOTIF_ACT_RELU) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) v[u] = v[u] > 0.f ? v[u] : 0.f;
                        } else if (acv == MOTIF_ACT_LRELU) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) v[u] = v[u] > 0.f ? v[u] : 0.1f * v[u];
                        } else if (acv != MOTIF_ACT_NONE) v = act_uniform(v, acv);
                        if (rmv == 2) v += rv[row][q];
                        else if (rmv == 3) {
                            v += rv[row][q];
#pragma unroll
                            for (int u = 0; u < 4; ++u) v[u] = v[u] > 0.f ? v[u] : 0.f;
                        } else if (rmv == 4) v *= rv[row][q];
                        const unsigned bo = offp(row, q) * 4u;
                        if (fullt) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(bo), "v"(v), "s"(obq) : "memory");
                        else if (okp(row, q)) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(bo), "v"(v), "s"(obq) : "memory");
                    }
                }
            };
            using I0 = std::integral_constant<int, 0>;
            if (rm == 0 && act == MOTIF_ACT_NONE) pieces(std::integral_constant<int, MOTIF_ACT_NONE>{}, I0{});
            else if (rm == 0 && act == MOTIF_ACT_RELU) pieces(std::integral_constant<int, MOTIF_ACT_RELU>{}, I0{});
            else if (rm == 0 && act == MOTIF_ACT_LRELU) pieces(std::integral_constant<int, MOTIF_ACT_LRELU>{}, I0{});
            else if (rm == 1 && act == MOTIF_ACT_NONE) pieces(std::integral_constant<int, MOTIF_ACT_NONE>{}, std::integral_constant<int, 1>{});
            else if (rm == 1 && act == MOTIF_ACT_LRELU) pieces(std::integral_constant<int, MOTIF_ACT_LRELU>{}, std::integral_constant<int, 1>{});
            else pieces(std::integral_constant<int, -1>{}, std::integral_constant<int, -1>{});
            if constexpr (NP == 2) {
                // an operand beyond fp16's range makes every cout of its pixel non-finite: any lane's check sees it
                if (stp && __builtin_amdgcn_class(chk, 0x207)) asm volatile("s_nop 5\n\tglobal_atomic_or %0, %1, %2" :: "v"(0u), "v"(1u), "s"(stp) : "memory");
            }
            return;
        }
        int lane_e = lane;                               // opaque copy: keeps the per-lane geometry inside the tile loop
        asm volatile("" : "+v"(lane_e));
        const int hf = lane_e >> 5, l5 = lane_e & 31;
        const int oy = ty * TH + 4 * tp + (l5 >> 3), ox = tx * 32 + (l5 & 7) * 4;
        const bool okl = oy < a.Ho && ox < a.Wo;
        const unsigned lb = (unsigned)hf * HWo + (unsigned)(oy * a.Wo + ox);
        const bool full = cl >= 32 && ty * TH + 4 * tp + 4 <= a.Ho && tx * 32 + 32 <= a.Wo;     // uniform: every lane stores every item
        auto okv = [&](int pass, int it) __attribute__((always_inline)) { return okl && 8 * pass + 2 * it + hf < cl; };
        auto offv = [&](int pass, int it) __attribute__((always_inline)) { return lb + (unsigned)(8 * pass + 2 * it) * HWo; };
        // all 16 residual quads are requested before the inverse transform (first touches of another XCD's output: ~2 k cycles; the
        // operand registers of the main loop are free here)
        f32x4 rv[4][4];
        auto load_res = [&](int pass) __attribute__((always_inline)) {
#pragma unroll
            for (int it = 0; it < 4; ++it) rv[pass][it] = rload(rb, (full || okv(pass, it)) ? offv(pass, it) : 0u);   // masked lanes read element 0
        };
        if (rm) {
            if (rpre_on) {                               // passes 0 / 1 were requested under the last chunk
#pragma unroll
                for (int it = 0; it < 4; ++it) { rv[0][it] = rpre[0][it]; rv[1][it] = rpre[1][it]; }
            } else { load_res(0); load_res(1); }
            load_res(2); load_res(3);
        }
        EPSTAMP();
        float* scr = (float*)land;                       // two halves of [8 couts][4 rows x 32 px]
        const int ew = hf * 512 + l5, er = hf * 128 + (l5 >> 3) * 32 + (l5 & 7) * 4;
        // pass p = couts 8p .. 8p+7 = registers 4p .. 4p+3 of every accumulator: inverse transform of those 16 values, transpose through
        // the scratch half p & 1 -- written per pass so that the compiler overlaps pass p+1's register work with pass p's LDS round trip
        auto write_pass = [&](int pass) __attribute__((always_inline)) {
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int r3 = 0; r3 < 4; ++r3) {
                    const int r = 4 * pass + r3;
                    const float o0 = (acc[tl][0][r] + acc[tl][1][r]) + acc[tl][2][r];        // the bias is inside M1 (init_acc)
                    const float o1 = (acc[tl][1][r] - acc[tl][2][r]) - acc[tl][3][r];
                    scr[(pass & 1) * 1024 + ew + r3 * 128 + (2 * tl) * 32] = o0;
                    scr[(pass & 1) * 1024 + ew + r3 * 128 + (2 * tl + 1) * 32] = o1;
                }
        };
        // AC / RM >= 0 (AC = -2: no activation below act_split, sigmoid from there on): activation / residual mode known at compile time (the common layers: one straight-line body, no per-item
        // scalar branches -- 16 items x the generic chain of wave-uniform tests cost a lone wave ~2 k cycles); -1: run-time switches.
        auto passes = [&](auto ac_tag, auto rm_tag) __attribute__((always_inline)) {
            constexpr int AC = decltype(ac_tag)::value, RM = decltype(rm_tag)::value;
            const int rmv = RM >= 0 ? RM : rm;
            const unsigned long long obq = obp;
            const unsigned long long stp = (unsigned long long)a.status;
            write_pass(0);
            EPSTAMP();
            // every residual quad is waited for BEFORE the first (uncounted) store: behind a store the counted wait of a later quad
            // could only end when that store has completed too
            if (rmv) asm volatile("" :: "v"(rv[3][0]), "v"(rv[3][1]), "v"(rv[3][2]), "v"(rv[3][3]));
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                if (pass + 1 < 4) write_pass(pass + 1);
                const int ac = AC >= 0 ? AC : ((a.act_split > 0 && cb + 8 * pass >= a.act_split) ? a.act2 : (CHAIN ? t.act : a.act));      // uniform per pass
                f32x4 v[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) v[it] = *(const f32x4*)(scr + (pass & 1) * 1024 + er + it * 256);
                if constexpr (NP == 2) {
                    // Range status word (MotifConvDesc.status).  An operand beyond fp16's range is packed as inf, and inf times ANY weight
                    // (0 included) is non-finite: every cout of the pixel comes out inf / NaN, so item 0 of pass 0 (couts 0 / 1 x the wave's
                    // 4 rows x 32 pixels) sees every pixel of the tile.  Fire-and-forget atomic as inline assembly (invisible to the
                    // wait-count pass, like the stores below); the branch is not taken on in-range data.
                    if (pass == 0 && stp) {
                        const bool bad = __builtin_amdgcn_class(v[0][0], 0x207) | __builtin_amdgcn_class(v[0][1], 0x207) |
                                         __builtin_amdgcn_class(v[0][2], 0x207) | __builtin_amdgcn_class(v[0][3], 0x207);
                        // s_nop: the pointer may have just been written to its scalar registers by a vector instruction (v_readlane of a spilled
                        // value): five wait states before a memory instruction may read it -- hipcc counts them for its own instructions only
                        if (bad) asm volatile("s_nop 5\n\tglobal_atomic_or %0, %1, %2" :: "v"(0u), "v"(1u), "s"(stp) : "memory");
                    }
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    if constexpr (NP == 2) v[it] *= 1.f / kWinoF16Scale;
                    if (rmv == 1) v[it] += rv[pass][it];
                    if constexpr (AC == MOTIF_ACT_RELU) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[it][q] = v[it][q] > 0.f ? v[it][q] : 0.f;
                    } else if constexpr (AC == MOTIF_ACT_LRELU) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[it][q] = v[it][q] > 0.f ? v[it][q] : 0.1f * v[it][q];
                    } else if constexpr (AC == -2) {         // offset | sigmoid(mask): one uniform test per pass, 1 / (1 + e^-x) with the hardware reciprocal (1 ulp)
                        if (cb + 8 * pass >= a.act_split) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[it][q] = __builtin_amdgcn_rcpf(1.f + expf(-v[it][q]));
                        }
                    } else if constexpr (AC < 0) v[it] = act_uniform(v[it], ac);
                    if (rmv == 2) v[it] += rv[pass][it];
                    else if (rmv == 3) {
                        v[it] += rv[pass][it];
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[it][q] = v[it][q] > 0.f ? v[it][q] : 0.f;
                    } else if (rmv == 4) v[it] *= rv[pass][it];
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const unsigned bo = offv(pass, it) * 4u;
                    const f32x4 val = v[it];
                    // s_nop: the wait state between a store of more than 8 bytes and the next vector write of its data registers -- hipcc
                    // inserts it for its own stores, not around inline assembly
                    if constexpr (CHAIN && !(CHAIN_ABL & 2)) {
                        if (full) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" :: "v"(bo), "v"(val), "s"(obq) : "memory");
                        else if (okv(pass, it)) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" :: "v"(bo), "v"(val), "s"(obq) : "memory");
                    } else {
                        if (full) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(bo), "v"(val), "s"(obq) : "memory");
                        else if (okv(pass, it)) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(bo), "v"(val), "s"(obq) : "memory");
                    }
                }
                EPSTAMP();
            }
        };
        using I = std::integral_constant<int, 0>;
        const int act = CHAIN ? t.act : a.act;
        if (a.act_split > 0 && rm == 0 && act == MOTIF_ACT_NONE && a.act2 == MOTIF_ACT_SIGMOID) passes(std::integral_constant<int, -2>{}, I{});    // the DCNs' offset | mask layer
        else if (a.act_split > 0) passes(std::integral_constant<int, -1>{}, std::integral_constant<int, -1>{});
        else if (rm == 0 && act == MOTIF_ACT_NONE) passes(std::integral_constant<int, MOTIF_ACT_NONE>{}, I{});
        else if (rm == 0 && act == MOTIF_ACT_RELU) passes(std::integral_constant<int, MOTIF_ACT_RELU>{}, I{});
        else if (rm == 0 && act == MOTIF_ACT_LRELU) passes(std::integral_constant<int, MOTIF_ACT_LRELU>{}, I{});
        else if (rm == 1 && act == MOTIF_ACT_NONE) passes(std::integral_constant<int, MOTIF_ACT_NONE>{}, std::integral_constant<int, 1>{});
        else if (rm == 1 && act == MOTIF_ACT_LRELU) passes(std::integral_constant<int, MOTIF_ACT_LRELU>{}, std::integral_constant<int, 1>{});
        else if (rm == 1 && act == MOTIF_ACT_RELU) passes(std::integral_constant<int, MOTIF_ACT_RELU>{}, std::integral_constant<int, 1>{});
        else if (rm == 2 && act == MOTIF_ACT_RELU) passes(std::integral_constant<int, MOTIF_ACT_RELU>{}, std::integral_constant<int, 2>{});
        else passes(std::integral_constant<int, -1>{}, std::integral_constant<int, -1>{});
#ifdef MOTIF_TRACE
        if (lane == 0 && blockIdx.x < 256 && ti < 8) {
#pragma unroll
            for (int i = 0; i < 7; ++i) g_wn_trace2[((blockIdx.x * 4 + wave) * 8 + ti) * 8 + i] = i < esn ? es[i] : 0;
        }
#endif
#undef EPSTAMP
    };

    // ---- prologue: step 0 staged in full, the row pieces of step 1 requested (nothing to hide them under) --------------------
    int t = bq;
    // CHAIN: the scalar atomics are ISSUED only where a tile ends (behind init_acc, in front of the store wait and the barrier): an
    // outstanding scalar request holds every `s_waitcnt lgkmcnt` of the LDS traffic -- the B fragments of a chunk body -- until it has
    // returned (~700 cycles; issued inside a chunk they cost wave 0, and through the barrier the workgroup, 0.8 + 2.4 k cycles per tile).
    // At the end of tile i: the ticket requested one tile earlier is taken (tile i + 2), its table entry written, its row word requested,
    // the next ticket requested; under chunk 1 of tile i + 1 the row word is taken: the state of tile i + 2, used from chunk nch - 2 on.
    // A finished tile is published at its own end, once every wave has waited for its stores (s_waitcnt vmcnt(0): the acknowledgements take
    // 2-3 k cycles from the last store, partly under init_acc and the bookkeeping above).  CHAIN_DEFER = 1 is the alternative that was built and
    // measured: publish under chunk 1 of the NEXT tile without any explicit wait -- vmcnt retires loads and stores in one order on gfx950
    // (tools/ubench_chain.hip: [store sc1; load; s_waitcnt vmcnt(1)] takes the store's latency), and every wave parks, in chunk 1, row pieces
    // it requested in chunk 0, i.e. behind the last store of the tile before.  Exact too, and slower: the acknowledgements then hold up the
    // in-order queue of the next tile's first chunk.
    bool tick_inflight = false, rows_inflight = false, pub_pending = false;
    int pub_row = 0, pub_ty = 0;
    bool first_run = true;
    if constexpr (CHAIN) {
        const long plane_b = (long)a.Cout * a.Ho * a.Wo * 4;
        for (int l = tid; l < ch.L; l += 256) {
            const u32x4 q0 = gload((const u32x4*)ch.layers + 2 * l), q1 = gload((const u32x4*)ch.layers + 2 * l + 1);
            auto bufq = [&](int id) __attribute__((always_inline)) {      // buffer id -> {base lo, hi, batch stride in bytes (< 2^32: host), 0}
                const unsigned long long b = id < 0 ? 0ull : id == 0 ? (unsigned long long)a.in0[0] : id == 1 ? (unsigned long long)a.out[0]
                                                   : (unsigned long long)(ch.work + (long)(id - 2) * ch.buf_floats);
                const long bs = id == 0 ? a.in0_bs[0] * 4 : id == 1 ? a.out_bs[0] * 4 : plane_b;
                return u32x4{(unsigned)b, (unsigned)(b >> 32), (unsigned)bs, 0u};
            };
            u32x4 s0 = bufq((int)q1[0]);
            s0[3] = q1[3];
            const unsigned long long wb = (((unsigned long long)q0[1] << 32) | q0[0]) + (unsigned long long)ch.wp_off * 4ull;
            ltab[4 * l] = s0;
            ltab[4 * l + 1] = bufq((int)q1[1]);
            ltab[4 * l + 2] = bufq((int)q1[2]);
            ltab[4 * l + 3] = u32x4{(unsigned)wb, (unsigned)(wb >> 32), q0[2], q0[3]};
        }
        for (int r = tid; r < ntiles; r += 256) {
            const int tx = r % a.tiles_x, s2 = r / a.tiles_x;
            ((unsigned*)(ltab + 4 * ch.L))[r] = (unsigned)(s2 % tiles_y) | ((unsigned)tx << 10) | ((unsigned)(s2 / tiles_y) << 20);
        }
        __syncthreads();
    }
    // CHAIN: one iteration per RUN = prologue + tiles that follow each other through the step pipeline.  A run ends when the tickets are used
    // up or when the next tile's producers had not all finished at the time it was polled; the workgroup then finishes its tile, waits with
    // nothing in flight, and starts the next run.  Otherwise: one iteration.
    for (;;) {
    WNTRACE(0);
    WNTRACE_RT(30);
#ifdef MOTIF_TRACE
    long long pst[6];
    pst[0] = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (CHAIN) {
        if (wave == 0) {                                 // a safe point: nothing of this workgroup in flight, blocking is allowed
            int st_cur = 1;
            if (first_run) {
                chain_ticket_issue();
                const int tk0 = chain_ticket_take();
                if (tk0 < chain_total) { chain_entry(ti, tk0); chain_ticket_issue(); tick_inflight = true; } else st_cur = 0;
            } else if (__builtin_amdgcn_readfirstlane(chs[1]) != 2) st_cur = 0;         // 2: the tile the last run polled too early (its entry is slot ti now)
            if (st_cur) {
                const TileP tc = load_tile(ti);
                cur_row = tc.tk; cur_ty = tc.ty;           // .tk = the tile's own row word; its producers' = one layer back
                pr_row = tc.tk >= a.N * tiles_y ? tc.tk - a.N * tiles_y : -1;
                pr_ty = tc.ty;
                if (!chain_wait()) st_cur = 3;
            }
            int stp = 0;
            if (st_cur == 1 && tick_inflight) {          // the tile after it: entry, one poll
                const int tkp = chain_ticket_take();
                tick_inflight = false;
                if (tkp < chain_total) {
                    chain_entry(ti + 1, tkp);
                    pend_row = e_row; pend_ty = pr_ty;
                    chain_rows_issue();
                    stp = chain_rows_take() ? 1 : 2;
                    chain_ticket_issue();
                    tick_inflight = true;
                }
            }
            rows_inflight = false;
            if (lane_now() == 0) { chs[0] = st_cur; chs[1] = stp; }
        }
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(chs[0]) != 1) return;
        first_run = false;
    } else {
        if (wave == 0) {
            fill_table(0);
            if ((long)bq + 64L * G < ntiles) fill_table(64);
        }
        __syncthreads();
    }
#ifdef MOTIF_TRACE
    pst[1] = __builtin_amdgcn_s_memtime();
#endif
    {
        const TileP t0 = load_tile(ti);
        setup_loads(t0, true);
        wbase = wnext = wptr(t0);
    }
    loadw(wbase, 0, wf[0]);
    loadw(wbase, 1, wf[1]);
    loadw(wbase, 2, wf[2]);
    float bias_v = bias_of(load_tile(ti));
    const float* bp_next = nullptr;                      // bias pointer / valid couts of the next tile (uniform), set at the tile change
    int clg_next = 0;
#pragma unroll
    for (int i = 0; i < NLD; ++i) st_load(i, 0);
#ifdef MOTIF_TRACE
    pst[2] = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pst[3] = __builtin_amdgcn_s_memtime();
#endif
    static_for<WS::S>([&](auto ic) __attribute__((always_inline)) { piece(std::integral_constant<int, kWSchedOf<NP>.ext[decltype(ic)::value]>{}, stg0); });
#ifdef MOTIF_TRACE
    pst[4] = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (TR) bias_lane = bias_scaled(bias_v); else { if (lane < 32) bias_w[lane] = bias_scaled(bias_v); }
    init_acc();
#pragma unroll
    for (int i = 0; i < NLD; ++i) st_load(i, 16);        // nch >= 2: step 1 is chunk 1 of this tile
    __syncthreads();
#ifdef MOTIF_TRACE
    pst[5] = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x < 256) { for (int i = 0; i < 6; ++i) g_wn_trace2[((blockIdx.x * 4 + wave) * 8 + 7) * 8 + i] = pst[i]; }
#endif
    WNTRACE(1);

    // ---- persistent tile loop ------------------------------------------------------------------------------------------------
    int buf = 0, slot = 2;
    bool has_next = false;
    for (;;) {
        const int t_next = t + G;
        if constexpr (!CHAIN) has_next = t_next < ntiles;
        for (int c = 0; c < nch; ++c) {
            if constexpr (CHAIN) {
                if (wave == 0 && c == 1 && rows_inflight) {         // requested at the end of the tile before: the state of the tile after this one
                    const int st = (chain_rows_take() || (CHAIN_ABL & 4)) ? 1 : 2;
                    rows_inflight = false;
                    if (lane_now() == 0) chs[1] = st;
                }
                if (c == nch - 2) has_next = __builtin_amdgcn_readfirstlane(chs[1]) == 1;
            }
            if (c == nch - 2) {                          // from here on the row-piece requests belong to the next tile
                const TileP tq = load_tile(has_next ? ti + 1 : ti);
                if (has_next) { wnext = wptr(tq); bp_next = tq.bp; clg_next = tq.clg; } else wnext = wbase;      // (the next tile's bias is requested under the epilogue)
                setup_loads(tq, has_next);
            }
            const bool last = c + 1 == nch;
            if constexpr (NP == 2 && !TR) {
                const bool res_tile = last && (CHAIN ? load_tile(ti).rm != 0 : a.res_mode != 0);
                if (res_tile) {                          // geometry of this tile's epilogue (finish_tile computes the same values)
                    const TileP tc = load_tile(ti);
                    rpre_cl = tc.clg - ct * 32;
                    rpre_on = rpre_cl > 0 && tc.rb != nullptr && !(WINO_ABL & 256) && !(a.dbg & 64);
                    rpre_base = tc.rb ? tc.rb + (long)(ct * 32) * (long)HWo : nullptr;
                    int lane_q = lane;
                    asm volatile("" : "+v"(lane_q));
                    const int l5q = lane_q & 31, oyq = tc.ty * TH + 4 * tp + (l5q >> 3), oxq = tc.tx * 32 + (l5q & 7) * 4;
                    rpre_okl = oyq < a.Ho && oxq < a.Wo;
                    rpre_lb = (unsigned)(lane_q >> 5) * HWo + (unsigned)(oyq * a.Wo + oxq);
                    rpre_full = rpre_cl >= 32 && tc.ty * TH + 4 * tp + 4 <= a.Ho && tc.tx * 32 + 32 <= a.Wo;
                }
            }
            const int sc = last ? 0 : c + 1, lc0 = (c + 2 < nch ? c + 2 : c + 2 - nch) * 16;
            const __amdgpu_buffer_rsrc_t wn = last ? wnext : wbase;
            chunk_body(c, buf, sc, wn, lc0);
            if (slot < 29) WNTRACE(slot);                // trace: body end | epilogue end | barrier passed, for the first 9 chunks
            if (last) {
                if constexpr (TR) bias_v = (has_next && bp_next && ct * 32 + l31 < clg_next) ? gload(bp_next + ct * 32 + l31) : 0.f;
                else bias_v = (has_next && bp_next && lane < 32 && ct * 32 + lane < clg_next) ? gload(bp_next + ct * 32 + lane) : 0.f;
                finish_tile(load_tile(ti));
                rpre_on = false;
                if constexpr (TR) bias_lane = bias_scaled(bias_v); else { if (lane < 32) bias_w[lane] = bias_scaled(bias_v); }    // the next tile's bias,
                CHSTAMP(0);
                const f32x16 bvec = init_bias();         // into its accumulators (below: CHAIN requests its scalar atomics in between, behind the last LDS read)
                CHSTAMP(1);
                if constexpr (CHAIN) {
                    if (wave == 0 && has_next) {         // (a run that ends here leaves the ticket in flight: the next run takes it)
                        bool none = true;
                        if (tick_inflight) {
                            const int tkn = chain_ticket_take();
                            CHSTAMP(2);
                            tick_inflight = false;
                            if (tkn < chain_total) {
                                chain_entry(ti + 2, tkn);
                                fut_row = e_row; fut_ty = pr_ty;
                                CHSTAMP(3);
                                chain_rows_issue();
                                rows_inflight = true;
                                chain_ticket_issue();
                                tick_inflight = true;
                                none = false;
                            }
                        }
                        if (none && lane_now() == 0) chs[1] = 0;            // the tickets are used up: the tile about to start is this workgroup's last
                    }
                }
                init_acc_from(bvec);
                // CHAIN: the tile is published once every wave's stores have been acknowledged (a store is counted until it is written)
                CHSTAMP(4);
                if constexpr (CHAIN && !(CHAIN_ABL & 1)) { if (!has_next || !CHAIN_DEFER) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                CHSTAMP(5);
            }
            if (slot < 29) WNTRACE(slot + 1);
            __syncthreads();
            if constexpr (CHAIN) {
                if (last) CHSTAMP(6);
                if (last) {
                    if (has_next && CHAIN_DEFER) { pub_pending = true; pub_row = cur_row; pub_ty = cur_ty; }
                    else if (wave == 0) chain_publish(cur_row, cur_ty);   // (from scalar registers: an LDS read here would wait for the scalar atomics just requested)
                }
                if (c == 1 && pub_pending) {             // the tile before: see pub_pending
                    if (wave == 0) chain_publish(pub_row, pub_ty);
                    pub_pending = false;
                }
                if (last) CHSTAMP(7);
            }
            buf ^= 1;
            if (slot < 29) { WNTRACE(slot + 2); slot += 3; }
        }
        if (!has_next) break;
        t = t_next; ++ti; wbase = wnext;
        if constexpr (CHAIN) { cur_row = pend_row; cur_ty = pend_ty; pend_row = fut_row; pend_ty = fut_ty; }
        // the table is a ring of 128 entries: when the first half of an epoch of 64 tiles begins, the entries of the epoch after next go
        // where the epoch before lay (every wave has passed the last barrier of tile ti - 1, the last reader of those)
        if constexpr (!CHAIN) { if ((ti & 63) == 0 && wave == 0) fill_table(ti + 64); }
    }
    if constexpr (!CHAIN) break;
    ++ti;                                                // CHAIN: the pending tile's entry (if any) becomes the current one
    }
    WNTRACE(31);
    WNTRACE_RT(29);
#undef CHAIN_SADD
#undef CHAIN_STAKE
#undef CHAIN_SSET
}

template <int NP, bool MULTI, bool TR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_wino_kernel(ConvArgs a, int ntiles, int tiles_y) {
    conv_wino_body<NP, MULTI, TR, false>(a, ntiles, tiles_y, ChainArgs{});
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_wino_chain_kernel(ConvArgs a, int ntiles, int tiles_y, ChainArgs ch) {
    conv_wino_body<2, false, false, true>(a, ntiles, tiles_y, ch);
}

// weight [Cout, Cin_g, 3, 3] fp32 -> A fragments [group][cout group of 64][k-step][part][cout tile][lane][8] bf16 (NP = 3) / fp16 of 2^8 x (NP = 2),
// k-step = (channel group of 16, position, kx); value = U_position[kx] of the file header, formed in fp64 and split from there
template <int NP>
__global__ void conv_wino_pack_kernel(const float* w, unsigned short* wp, int Cout_g, int Cin_g, int nks, int ncg, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), ct = (int)((i >> 9) & 1);
    long tt = i >> 10;
    const int part = (int)(tt % NP); tt /= NP;
    const int ks = (int)(tt % nks); tt /= nks;
    const int cgi = (int)(tt % ncg);
    const int g = (int)(tt / ncg);
    const int col = cgi * 64 + ct * 32 + (lane & 31);
    const int c = (ks / 12) * 16 + 8 * (lane >> 5) + e, s = ks % 12, pos = s / 3, kx = s % 3;
    double v = 0.0;
    if (col < Cout_g && c < Cin_g) {
        const float* wk = w + ((long)(g * Cout_g + col) * Cin_g + c) * 9 + kx;
        const double g0 = wk[0], g1 = wk[3], g2 = wk[6];
        v = pos == 0 ? g0 : pos == 1 ? 0.5 * (g0 + g1 + g2) : pos == 2 ? 0.5 * (g0 - g1 + g2) : g2;
    }
    unsigned short out = 0;
    if constexpr (NP == 2) {
        v *= (double)kWinoF16Scale;
        for (int p = 0; p <= part; ++p) {
            const _Float16 h = (_Float16)(float)v;
            out = __builtin_bit_cast(unsigned short, h);
            v -= (double)(float)h;
        }
    } else {
        for (int p = 0; p <= part; ++p) {
            const unsigned pk = pk_bf16((float)v, 0.f);
            out = (unsigned short)(pk & 0xffffu);
            v -= (double)bf_lo(pk);
        }
    }
    wp[i] = out;
}

// ---- host side -----------------------------------------------------------------------------------------------------------
namespace {
int wn_cu_count() {
    static int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus;
}
}  // namespace

// The Winograd block follows the direct block in the packed blob of every split-eligible layer with mma = 6 (three bf16 parts) or
// mma = 7 (two fp16 parts; the direct block of such a layer is the three-part one: the other kernels know no other); pack and forward
// agree from the desc alone.  WHICH kernel runs is decided per launch.
static inline int wino_parts(int mma) { return mma == 7 ? 2 : mma == 6 ? 3 : 0; }

long motif_conv_wino_packed_floats(const MotifConvDesc* d) {
    const int NP = wino_parts(d->mma);
    if (NP == 0) return 0;
    const int Cin_g = (d->C0 + d->C1) / d->groups, Cout_g = d->Cout / d->groups;
    const long nks = 12L * ((Cin_g + 15) / 16), ncg = (Cout_g + 63) / 64;
    return (long)d->groups * ncg * nks * NP * 2 * 64 * 4;
}

int motif_conv_wino_pack(const MotifConvDesc* d, const float* weight, float* packed, hipStream_t s) {
    const int Cin_g = (d->C0 + d->C1) / d->groups, Cout_g = d->Cout / d->groups, NP = wino_parts(d->mma);
    const int nks = 12 * ((Cin_g + 15) / 16), ncg = (Cout_g + 63) / 64;
    const long total = (long)d->groups * ncg * nks * NP * 2 * 64 * 8;
    if (NP == 2) conv_wino_pack_kernel<2><<<cdiv(total, 256), 256, 0, s>>>(weight, (unsigned short*)packed, Cout_g, Cin_g, nks, ncg, total);
    else conv_wino_pack_kernel<3><<<cdiv(total, 256), 256, 0, s>>>(weight, (unsigned short*)packed, Cout_g, Cin_g, nks, ncg, total);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// Same layout conditions as conv_split2 (zero padding 1, rows of whole 16-byte units, aligned tensors, activation split on an
// 8-cout boundary); split arithmetic only.
bool motif_conv_wino_eligible(const MotifConvDesc* d, const ConvArgs& a, int P) {
    if (wino_parts(d->mma) == 0 || d->pad != 1 || d->pad_mode != 0 || (d->W & 3)) return false;
    const long HW = (long)d->H * d->W;
    if (HW * 64 >= 0x7fffffffL) return false;
    const int Cout_g = d->Cout / d->groups;
    if ((long)((d->C0 + d->C1) / d->groups) * HW * 4 >= 0x7fffffffL) return false;
    if (d->act_split > 0 && ((d->act_split & 7) || (d->groups > 1 && (Cout_g & 7)))) return false;
    if (d->C1 > 0 && (d->groups != 1 || d->C0 % 16)) return false;
    if ((d->C0 + d->C1) / d->groups <= 16) return false;               // the kernel's step pipeline looks two chunks ahead within a tile
    for (int i = 0; i < P; ++i) {
        unsigned long long bits = (unsigned long long)a.in0[i] | (unsigned long long)a.out[i] | (unsigned long long)a.in1[i] | (unsigned long long)a.res[i];
        if (bits & 15) return false;
        if ((a.in0_bs[i] | a.out_bs[i] | (a.in1[i] ? a.in1_bs[i] : 0) | (a.res[i] ? a.res_bs[i] : 0)) & 3) return false;
    }
    return true;
}

int motif_conv_wino_launch(const MotifConvDesc* d, ConvArgs& a, int P, hipStream_t s) {
    const int Cin_g = (d->C0 + d->C1) / d->groups, Cout_g = d->Cout / d->groups, NP = wino_parts(d->mma);
    const int Ho = d->H, Wo = d->W;                      // pad 1
    a.Ho = Ho; a.Wo = Wo; a.Cin_g = Cin_g; a.Cout_g = Cout_g;
    a.Kpad = 12 * ((Cin_g + 15) / 16);
    a.ncg = (Cout_g + 63) / 64;
    a.tiles_x = (Wo + 31) / 32;
    const int ncgG = d->groups * a.ncg, cus = wn_cu_count();
    a.Cout = d->Cout;
    a.CK = ncgG;
    const long direct = motif_conv_split_packed_floats_direct(d);      // the Winograd block follows the direct one
    for (int i = 0; i < MOTIF_MAX_PROBLEMS; ++i) a.wp[i] = a.wp[i] + direct;
    const int tiles_y = (Ho + 7) / 8;
    const long T = (long)a.tiles_x * ncgG * d->N * P * tiles_y;
    if (T >= 0x7fffffffL) return MOTIF_ELIMIT;
    const int G = (int)(T < cus ? T : cus);
    const size_t ldsb = ((size_t)4 * 16 + (size_t)2 * NP * (2 * 16 * 34 + 4) + (size_t)4 * 640 + (size_t)128 * 4) * 16;      // bias | staging | landing | tile table
    // transposed accumulators (register-only epilogue): OPT-IN (option conv_wino_tr = 1), for launches whose activation is uniform over the
    // couts.  Measured (tools/conv_bench.py, same box, 3 x 64 -> 64 x 180 x 320): 49.7 vs 49.0 us without a residual -- the 350 instructions
    // and the LDS round trip it removes are not what the exposed epilogue waits for -- and 61.7 vs 53.7 us WITH one (a residual piece per lane
    // = 32 cache lines per load instruction instead of 8).  The default stays the row-major form.
    const bool tr = NP == 2 && d->act_split <= 0 && motif_opt(MOTIF_OPT_CONV_WINO_TR) == 1;
    if (motif_opt(MOTIF_OPT_CONV_WINO_RPRE) == 1) a.dbg |= 64;          // A/B switch: residual quads requested in the epilogue only
    const void* fn = NP == 2 ? (tr ? (P > 1 ? (const void*)conv_wino_kernel<2, true, true> : (const void*)conv_wino_kernel<2, false, true>)
                                    : (P > 1 ? (const void*)conv_wino_kernel<2, true, false> : (const void*)conv_wino_kernel<2, false, false>))
                             : (P > 1 ? (const void*)conv_wino_kernel<3, true, false> : (const void*)conv_wino_kernel<3, false, false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    if (NP == 2 && tr) {
        if (P > 1) conv_wino_kernel<2, true, true><<<dim3(G, 1, 1), 256, ldsb, s>>>(a, (int)T, tiles_y);
        else conv_wino_kernel<2, false, true><<<dim3(G, 1, 1), 256, ldsb, s>>>(a, (int)T, tiles_y);
    } else if (NP == 2) {
        if (P > 1) conv_wino_kernel<2, true, false><<<dim3(G, 1, 1), 256, ldsb, s>>>(a, (int)T, tiles_y);
        else conv_wino_kernel<2, false, false><<<dim3(G, 1, 1), 256, ldsb, s>>>(a, (int)T, tiles_y);
    } else {
        if (P > 1) conv_wino_kernel<3, true, false><<<dim3(G, 1, 1), 256, ldsb, s>>>(a, (int)T, tiles_y);
        else conv_wino_kernel<3, false, false><<<dim3(G, 1, 1), 256, ldsb, s>>>(a, (int)T, tiles_y);
    }
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// ---- chain mode (motif_conv2d_chain_fwd) -----------------------------------------------------------------------------------------
// One block in front of the chain launch: zeroes the workspace (ticket, abort word, progress count, completion counters) and checks the
// caller's layer table -- a DEVICE array the host entry cannot read -- against the scratch buffers the caller's workspace really holds
// (ids 2 .. 1 + nbuf), that no layer writes x and that every layer has a source.  A bad table poisons the ticket counter: every workgroup
// of the chain launch draws a ticket beyond the last tile and leaves before anything is read or written; status bit 2 reports it (without
// a status word the launch traps -- it must not look like a finished one).
__global__ __launch_bounds__(256) void conv_wino_chain_prepare_kernel(unsigned* ws, int words, const ChainLayerDev* layers, int L, int nbuf, unsigned* status) {
    for (int i = threadIdx.x; i < words; i += 256) ws[i] = 0u;
    int bad = 0;
    for (int l = threadIdx.x; l < L; l += 256) {
        const u32x4 q1 = *((const u32x4*)layers + 2 * l + 1);
        const int ids = (int)q1[0], idd = (int)q1[1], idr = (int)q1[2];
        bad |= (int)(ids < 0 || ids >= 2 + nbuf) | (int)(idd < 1 || idd >= 2 + nbuf) | (int)(idr >= 2 + nbuf);
    }
    if (__syncthreads_or(bad) && threadIdx.x == 0) {
        ws[0] = 0x40000000u;
        ws[1] = 1u;
        if (status) atomicOr(status, 4u); else asm volatile("s_trap 2");
    }
}

static_assert(sizeof(ChainLayerDev) == sizeof(MotifChainLayer) && sizeof(MotifChainLayer) == 32, "the device table is read as two 16-byte quads");

static bool wino_chain_shape_ok(const MotifConvDesc* d, int L, long* tiles) {
    if (!d || L < 1 || wino_parts(d->mma) != 2 || d->groups != 1 || d->C1 != 0 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->dil != 1) return false;
    if (d->pad != 1 || d->pad_mode != 0 || (d->W & 3) || d->N < 1 || d->H < 1 || d->W < 1) return false;
    if (d->C0 != d->Cout || d->Cout > 64 || d->C0 < 49) return false;          // one cout group; >= 4 chunks: the ticket and the flags are a chunk each ahead of the tile change
    const long HW = (long)d->H * d->W;
    if (HW * 64 * 4 >= 0x7fffffffL) return false;
    const long T = (long)((d->W + 31) / 32) * ((d->H + 7) / 8) * d->N;
    if (T * L >= (1L << 24) || L > 256 || T > 6144) return false;      // the layer table (<= 16 KB) and the tile decode table (<= 24 KB) live in LDS
    if (d->N >= 4096 || (d->W + 31) / 32 >= 1024 || (d->H + 7) / 8 >= 1024) return false;
    if (tiles) *tiles = T;
    return true;
}

extern "C" long motif_conv2d_chain_ws_words(const MotifConvDesc* d, int L) {
    long T = 0;
    if (!wino_chain_shape_ok(d, L, &T)) return 0;
    return 64 + (long)L * d->N * ((d->H + 7) / 8);       // ticket, abort word, one completion counter per (layer, image, tile row)
}

extern "C" int motif_conv2d_chain_fwd(const MotifConvDesc* d, int L, const MotifChainLayer* layers, const float* x, float* out, float* work,
                                      long work_floats, uint32_t* ws, void* stream) {
    long T = 0;
    if (!wino_chain_shape_ok(d, L, &T)) return MOTIF_ELIMIT;
    if (!layers || !x || !out || !ws) return MOTIF_EINVAL;
    const long HW = (long)d->H * d->W, plane = (long)d->Cout * HW;
    const long bsx = d->in0_bs ? d->in0_bs : plane, bso = d->out_bs ? d->out_bs : plane;
    if ((((unsigned long long)x | (unsigned long long)out | (unsigned long long)work | (unsigned long long)layers) & 15) || ((bsx | bso) & 3)) return MOTIF_EINVAL;
    if (bsx < 0 || bso < 0 || bsx * 4 >= (1L << 32) || bso * 4 >= (1L << 32) || (long)d->N * plane * 4 >= (1L << 40)) return MOTIF_ELIMIT;      // batch strides in bytes are 32-bit table fields
    if (work_floats < 0 || (work_floats > 0 && !work)) return MOTIF_EINVAL;
    // scratch buffers the caller's workspace holds: ids 2 .. 1 + nbuf are valid in the (device-resident) layer table; the kernel checks every id
    // against it when it resolves the table and reports a bad one through status bit 2 (all workgroups leave before anything is written)
    const long nbuf_l = work ? work_floats / ((long)d->N * plane) : 0;
    const int nbuf = (int)(nbuf_l > 64 ? 64 : nbuf_l);
    hipStream_t s = (hipStream_t)stream;
    ConvArgs a = {};
    a.in0[0] = x; a.out[0] = out; a.in0_bs[0] = bsx; a.out_bs[0] = bso;
    a.N = d->N; a.C0 = d->C0; a.H = d->H; a.W = d->W; a.Ho = d->H; a.Wo = d->W; a.Cout = d->Cout;
    a.Cin_g = d->C0; a.Cout_g = d->Cout;
    a.KH = 3; a.KW = 3; a.stride = 1; a.pad = 1; a.dil = 1; a.pad_mode = 0;
    a.Kpad = 12 * ((d->C0 + 15) / 16);
    a.ncg = 1; a.CK = 1;
    a.tiles_x = (d->W + 31) / 32;
    a.dbg = motif_opt(MOTIF_OPT_CONV_DBG);
    if (motif_opt(MOTIF_OPT_CONV_WINO_RPRE) == 1) a.dbg |= 64;
    a.status = d->status;
    ChainArgs ch = {};
    ch.layers = (const ChainLayerDev*)layers;
    ch.work = work; ch.buf_floats = (long)d->N * plane;
    ch.ws = ws;
    ch.wp_off = motif_conv_split_packed_floats_direct(d);               // the Winograd block follows the direct one in every layer's blob
    ch.L = L;
    const int tiles_y = (d->H + 7) / 8, cus = wn_cu_count();
    const long total = T * L;
    int G = (int)(total < cus ? total : cus);
    if (const int w = motif_opt(MOTIF_OPT_CONV_CHAIN_WGS); w > 0 && w < G) G = w;        // fewer workgroups leave CUs to the launches of other streams
    const size_t ldsb = ((size_t)4 * 16 + (size_t)2 * 2 * (2 * 16 * 34 + 4) + (size_t)4 * 640 + (size_t)128 * 4) * 16;      // as motif_conv_wino_launch, NP = 2
    const size_t ldsc = ldsb + (size_t)L * 64 + (size_t)T * 4;          // + the resolved layer table and the tile decode table
    hipError_t e = hipFuncSetAttribute((const void*)conv_wino_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    conv_wino_chain_prepare_kernel<<<1, 256, 0, s>>>(ws, (int)(64 + (long)L * d->N * tiles_y), (const ChainLayerDev*)layers, L, nbuf, d->status);
    conv_wino_chain_kernel<<<dim3(G, 1, 1), 256, ldsc, s>>>(a, (int)T, tiles_y, ch);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

// Space-time local implicit MLPs (SIREN, omega0 = 30) as register-chained fp32-MFMA kernels, gfx950.
//
// Orientation: out[m][pixel] = sum_k W[m][k] * x[k][pixel]; a wave owns 32 pixels (MFMA columns).
// The C/D layout of v_mfma_f32_32x32x2_f32 gives lane (hf = lane>>5, p = lane&31) accumulator reg r of
// output tile t the row m = 32t + (r&3) + 8(r>>2) + 4hf.  Read the same register file as the NEXT
// layer's B operand: register s = 16t + r of lane (hf, p) is input k = kmap(s, hf) = 8(s>>2) + 4hf + (s&3)
// of pixel p -- exactly "lower half-wave supplies one k, upper half the other" that an MFMA step
// needs.  So activations never leave registers between layers: each layer is
//     for s: for t: acc[t] = mfma(Wp[s][t][lane], h[s], acc[t])
// with the weights pre-permuted on the host side of the ABI (motif_siren_pack) so that the A operand
// of step (s, t) is one contiguous 256-byte, conflict-free LDS read.  The whole packed network lives
// in LDS (<= 156 KB; one persistent 512-thread block per CU), the first layer's inputs are fetched
// straight into B-operand form (nearest gather of LR features, coordinate tables, splat accumulator
// with the post-splat normalisation), and narrow heads (256 -> 3) run on the VALU, which has the same
// fp32 rate as the f32 MFMA and no 32-row padding.
#include "siren_common.h"

// packed blob layout per MFMA layer: Wp[KS][MT][64] then Bp[MT][16][2]; per VALU head: Wv[M][KQ][2][4], bias[M] (padded to 4)
__host__ __device__ constexpr long mfma_layer_floats(int K, int M) { return (long)(pad8(K) / 2) * (pad32(M) / 32) * 64 + (long)(pad32(M) / 32) * 32; }
__host__ __device__ constexpr long valu_head_floats(int K, int M) { return (long)M * (pad8(K) / 8) * 8 + 4; }

// ---------------------------------------------------------------- layer primitives
// A wave processes TP = 2 pixel tiles (64 pixels) in lockstep: every A operand read from LDS feeds two
// MFMAs, and while one tile's accumulators go through the sine (VALU) the other tile's MFMAs execute.

template <int TP, int MT>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[TP][MT], const float* bp, int hf) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float b = bp[(t * 16 + r) * 2 + hf];
#pragma unroll
            for (int p = 0; p < TP; ++p) acc[p][t][r] = b;
        }
}

// full layer: KS steps, MT output tiles, inputs h[TP][KS] in registers
template <int KS, int MT, int MTW, int TP>
__device__ __forceinline__ void mfma_layer(const float (&h)[TP][KS], f32x16 (&acc)[TP][MT], const float* wp, int t0, int lane) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const float w = wp[(s * MTW + t0 + t) * 64 + lane];
#pragma unroll
            for (int p = 0; p < TP; ++p) acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, h[p][s], acc[p][t], 0, 0, 0);
        }
        if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // bound the scheduler's LDS-read hoisting (VGPR pressure)
    }
}

template <int TP, int MT>
__device__ __forceinline__ void sine(const f32x16 (&acc)[TP][MT], float (&h)[TP][MT * 16]) {
#pragma unroll
    for (int p = 0; p < TP; ++p)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                h[p][t * 16 + r] = sin_cw(30.0f * acc[p][t][r]);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep at most 4 sin pipelines live
            }
}

// offsets (floats) of each layer inside the packed blob
template <int MODE> struct Layout {
    static constexpr int K0 = Net<MODE>::K0;
    static constexpr long W0 = 0;
    static constexpr long B0 = W0 + (long)(pad8(K0) / 2) * 2 * 64;
    static constexpr long W1 = B0 + 64;                       // 64 -> 64
    static constexpr long B1 = W1 + 32L * 2 * 64;
    // synth has one more 64 -> 64 layer
    static constexpr long W1b = B1 + 64;
    static constexpr long B1b = W1b + (MODE == MODE_SYNTH ? 32L * 2 * 64 : 0);
    static constexpr long W2 = (MODE == MODE_SYNTH ? B1b + 64 : B1 + 64);   // 64 -> 256
    static constexpr long B2 = W2 + 32L * 8 * 64;
    static constexpr long W3 = B2 + 256;                      // head
    static constexpr long TOTAL = W3 + (Net<MODE>::HEAD == 64 ? mfma_layer_floats(256, 64) : valu_head_floats(256, 3));
    // how much of the blob is staged in LDS (imnet's 256->64 head is streamed from L2 instead)
    static constexpr long LDS_FLOATS = (MODE == MODE_IMNET) ? W3 : TOTAL;
};

// PRE: the LR-resolution part of layer 0 (W0[:, gathered channels] . feature + b0) was precomputed at LR
// resolution by a 1x1 convolution (it does not depend on the HR pixel or on t); `src_lr` then holds that partial
// pre-activation and seeds the accumulator instead of the bias, and only the remaining input channels go
// through MFMA steps.
template <int MODE, int TP, bool PRE>
__global__ __launch_bounds__(SIREN_THREADS) MOTIF_SCALAR_F32 void siren_kernel(SirenArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using L = Layout<MODE>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, l31 = lane & 31;
    {
        const f32x4* src = (const f32x4*)a.packed;
        f32x4* dst = (f32x4*)lds;
        for (long i = tid; i < L::LDS_FLOATS / 4; i += SIREN_THREADS) dst[i] = src[i];
    }
    __syncthreads();

    // The two waves that share a SIMD run the same code from the same start and stay phase-locked: both in
    // their MFMA stretch, then both in their sine (VALU) stretch, so the matrix and vector pipes never overlap
    // (measured: MFMA busy 57 % + VALU busy 36 %).  Start the second wave of each SIMD half a tile late.
    if (wave >= SIREN_WAVES / 2) {
        const int naps = a.stagger;
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
    }
    const long Q = (long)a.HH * a.WW;
    const long HWl = (long)a.H * a.W;
    const int pairs_per_img = (int)((Q + 32 * TP - 1) / (32 * TP));
    const long total = (long)a.NB * pairs_per_img;

    for (long work = (long)blockIdx.x * SIREN_WAVES + wave; work < total; work += (long)gridDim.x * SIREN_WAVES) {
        const int img = (int)(work / pairs_per_img);
        const long pbase = (long)(work % pairs_per_img) * (32 * TP) + l31;
        long pp[TP], pc[TP], lr[TP];
        int Y[TP], X[TP];
        bool valid[TP];
#pragma unroll
        for (int p = 0; p < TP; ++p) {
            pp[p] = pbase + 32 * p;
            valid[p] = pp[p] < Q;
            pc[p] = valid[p] ? pp[p] : Q - 1;
            Y[p] = (int)(pc[p] / a.WW);
            X[p] = (int)(pc[p] - (long)Y[p] * a.WW);
            lr[p] = (long)a.iy[Y[p]] * a.W + a.ix[X[p]];
        }

        // ------------------------------------------------ layer 0: inputs fetched in B-operand form
        f32x16 acc0[TP][2];
        if constexpr (PRE) {
            const int ilr = (MODE == MODE_FLOW || MODE == MODE_SYNTH) ? img / a.N : img;
#pragma unroll
            for (int p = 0; p < TP; ++p) {
                const float* gp = a.src_lr + (long)ilr * 64 * HWl + lr[p] + (long)(4 * hf) * HWl;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc0[p][t][r] = gp[(long)(32 * t + (r & 3) + 8 * (r >> 2)) * HWl];
            }
        } else {
            init_bias(acc0, lds + L::B0, hf);
        }
        const float* w0 = lds + L::W0;
        auto step0 = [&](int s, const float (&v)[TP]) {      // one k-step of layer 0 for both tiles
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float w = w0[(s * 2 + t) * 64 + lane];
#pragma unroll
                for (int p = 0; p < TP; ++p) acc0[p][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, v[p], acc0[p][t], 0, 0, 0);
            }
        };
        if constexpr (MODE == MODE_IMNET || MODE == MODE_FLOW) {
            // img = b2 (imnet) or b2*N + n (flow); source LR image = b2
            const int b2 = (MODE == MODE_FLOW) ? img / a.N : img;
            if constexpr (!PRE) {
                const float* f[TP];
    #pragma unroll
                for (int p = 0; p < TP; ++p) f[p] = a.src_lr + (long)b2 * 64 * HWl + lr[p] + (long)(4 * hf) * HWl;
                float v[4][TP];
    #pragma unroll
                for (int r = 0; r < 4; ++r)
    #pragma unroll
                    for (int p = 0; p < TP; ++p) v[r][p] = f[p][(long)r * HWl];
    #pragma unroll 1
                for (int q = 0; q < 8; ++q) {      // software pipelined: the gather of q+1 flies under the MFMAs of q
                    float vn[4][TP];
    #pragma unroll
                    for (int p = 0; p < TP; ++p) f[p] += 8 * HWl;
                    if (q < 7) {
    #pragma unroll
                        for (int r = 0; r < 4; ++r)
    #pragma unroll
                            for (int p = 0; p < TP; ++p) vn[r][p] = f[p][(long)r * HWl];
                    }
    #pragma unroll
                    for (int r = 0; r < 4; ++r) step0(q * 4 + r, v[r]);
    #pragma unroll
                    for (int r = 0; r < 4; ++r)
    #pragma unroll
                        for (int p = 0; p < TP; ++p) v[r][p] = vn[r][p];
                }
            }
            float e[4][TP];
#pragma unroll
            for (int p = 0; p < TP; ++p) {
                if constexpr (MODE == MODE_FLOW) {
                    e[0][p] = a.times[img % (a.B * a.N)]; e[1][p] = a.rel_y[Y[p]]; e[2][p] = a.rel_x[X[p]]; e[3][p] = 0.f;
                } else {
                    e[0][p] = a.rel_y[Y[p]]; e[1][p] = a.rel_x[X[p]]; e[2][p] = 0.f; e[3][p] = 0.f;
                }
            }
#pragma unroll
            for (int s = 32; s < 36; ++s) {
                float v[TP];
#pragma unroll
                for (int p = 0; p < TP; ++p) v[p] = hf ? 0.f : e[s - 32][p];
                step0(s, v);
            }
        } else {
            // synth: img = b*N + n.  k<130: sum/wz ; 130: zmax ; 131: cnt/16 ; 132: wz_/cnt_ ;
            // 133..196: residual (gathered LR F01 of batch b) ; 197: t ; 198,199: zero pad
            const int b = img / a.N;
            const float tval = a.times[img];
            const float* A[TP];
            const float* R[TP];
            float wz[TP], cnt[TP], cnt_[TP], wz_[TP];
#pragma unroll
            for (int p = 0; p < TP; ++p) {
                A[p] = a.acc + (long)img * 133 * Q + pc[p];
                R[p] = a.src_lr + (long)b * 64 * HWl + lr[p];
                wz[p] = A[p][130 * Q];
                cnt[p] = A[p][132 * Q];
                if (wz[p] == 0.f) wz[p] = 1.0f;                           // Ours.py:813
                cnt_[p] = (cnt[p] == 0.f) ? 1.0f : cnt[p];                // Ours.py:828
                wz_[p] = (wz[p] == 1.0f) ? 0.f : wz[p];                   // Ours.py:830
            }
            // q = 0..15: k = 8q + 4hf + r < 128, all accumulator planes
            {
                const float* ap[TP];
#pragma unroll
                for (int p = 0; p < TP; ++p) ap[p] = A[p] + (long)(4 * hf) * Q;
                float v[4][TP];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int p = 0; p < TP; ++p) v[r][p] = ap[p][(long)r * Q];
#pragma unroll 1
                for (int q = 0; q < 16; ++q) {
                    float vn[4][TP];
#pragma unroll
                    for (int p = 0; p < TP; ++p) ap[p] += 8 * Q;
                    if (q < 15) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int p = 0; p < TP; ++p) vn[r][p] = ap[p][(long)r * Q];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float d[TP];
#pragma unroll
                        for (int p = 0; p < TP; ++p) d[p] = v[r][p] / wz[p];
                        step0(q * 4 + r, d);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int p = 0; p < TP; ++p) v[r][p] = vn[r][p];
                }
            }
            // q = 16 (k = 128..135) and q = 24 (k = 192..199) straddle input classes
            auto general = [&](int k, int p) -> float {
                if (k < 130) return A[p][(long)k * Q] / wz[p];
                if (k == 130) return A[p][131 * Q];
                if (k == 131) return cnt[p] / 16.0f;
                if (k == 132) return wz_[p] / cnt_[p];
                if (k <= 196) return PRE ? 0.f : R[p][(long)(k - 133) * HWl];
                if (k == 197) return tval;
                return 0.f;
            };
#pragma unroll
            for (int s = 64; s < 68; ++s) {
                float v[TP];
#pragma unroll
                for (int p = 0; p < TP; ++p) v[p] = general(kmap(s, hf), p);
                step0(s, v);
            }
            if constexpr (!PRE) {
                // q = 17..23: k = 136..191, all residual channels k-133
                {
                    const float* rp[TP];
    #pragma unroll
                    for (int p = 0; p < TP; ++p) rp[p] = R[p] + (long)(136 - 133 + 4 * hf) * HWl;
                    float v[4][TP];
    #pragma unroll
                    for (int r = 0; r < 4; ++r)
    #pragma unroll
                        for (int p = 0; p < TP; ++p) v[r][p] = rp[p][(long)r * HWl];
    #pragma unroll 1
                    for (int q = 17; q < 24; ++q) {
                        float vn[4][TP];
    #pragma unroll
                        for (int p = 0; p < TP; ++p) rp[p] += 8 * HWl;
                        if (q < 23) {
    #pragma unroll
                            for (int r = 0; r < 4; ++r)
    #pragma unroll
                                for (int p = 0; p < TP; ++p) vn[r][p] = rp[p][(long)r * HWl];
                        }
    #pragma unroll
                        for (int r = 0; r < 4; ++r) step0(q * 4 + r, v[r]);
    #pragma unroll
                        for (int r = 0; r < 4; ++r)
    #pragma unroll
                            for (int p = 0; p < TP; ++p) v[r][p] = vn[r][p];
                    }
                }
            }
#pragma unroll
            for (int s = 96; s < 100; ++s) {
                float v[TP];
#pragma unroll
                for (int p = 0; p < TP; ++p) v[p] = general(kmap(s, hf), p);
                step0(s, v);
            }
        }
        float h1[TP][32];
        sine(acc0, h1);

        // ------------------------------------------------ 64 -> 64 (x2 for synth)
        f32x16 acc1[TP][2];
        init_bias(acc1, lds + L::B1, hf);
        mfma_layer<32, 2, 2>(h1, acc1, lds + L::W1, 0, lane);
        float h2[TP][32];
        sine(acc1, h2);
        if constexpr (MODE == MODE_SYNTH) {
            init_bias(acc1, lds + L::B1b, hf);
            mfma_layer<32, 2, 2>(h2, acc1, lds + L::W1b, 0, lane);
            sine(acc1, h2);
        }

        // ------------------------------------------------ 64 -> 256 in four 64-wide chunks, each fed to the head
        if constexpr (Net<MODE>::HEAD == 3) {
            float sum[TP][3];
#pragma unroll
            for (int p = 0; p < TP; ++p) sum[p][0] = sum[p][1] = sum[p][2] = 0.f;
#pragma unroll 1
            for (int c = 0; c < 4; ++c) {
                f32x16 acc2[TP][2];
                init_bias(acc2, lds + L::B2 + c * 64, hf);
                mfma_layer<32, 2, 8>(h2, acc2, lds + L::W2, 2 * c, lane);
                float hc[TP][32];
                sine(acc2, hc);
                valu_head_partial<3, 32>(hc, sum, lds + L::W3, 8 * c, hf);
            }
            const float* hb = lds + L::W3 + 3 * 32 * 8;
#pragma unroll
            for (int p = 0; p < TP; ++p) {
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    sum[p][o] += __shfl_xor(sum[p][o], 32);
                    sum[p][o] += hb[o];
                }
                if (valid[p] && hf == 0) {
                    if constexpr (MODE == MODE_FLOW) {
#pragma unroll
                        for (int o = 0; o < 3; ++o) a.out[((long)img * 3 + o) * Q + pp[p]] = sum[p][o];
                    } else {
                        const int b = img / a.N, n = img % a.N;
#pragma unroll
                        for (int o = 0; o < 3; ++o) {
                            float v = sum[p][o];
                            v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
                            a.out[(((long)n * a.B + b) * 3 + o) * Q + pp[p]] = v;
                        }
                    }
                }
            }
        } else {
            f32x16 acc3[TP][2];
            const float* w3g = a.packed + L::W3;                       // streamed from L2
            const float* b3g = w3g + 128L * 2 * 64;
            init_bias(acc3, b3g, hf);
#pragma unroll 1
            for (int c = 0; c < 4; ++c) {
                f32x16 acc2[TP][2];
                init_bias(acc2, lds + L::B2 + c * 64, hf);
                mfma_layer<32, 2, 8>(h2, acc2, lds + L::W2, 2 * c, lane);
                float hc[TP][32];
                sine(acc2, hc);
                mfma_layer<32, 2, 2>(hc, acc3, w3g + (long)c * 32 * 2 * 64, 0, lane);
            }
#pragma unroll
            for (int p = 0; p < TP; ++p) {
                if (!valid[p]) continue;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hf;
                        a.out[((long)img * 64 + m) * Q + pp[p]] = acc3[p][t][r];
                    }
            }
        }
    }
}

// ---------------------------------------------------------------- packing (device side, weights are device tensors)
struct PackArgs { const float* w[6]; const float* b[6]; int dims[7]; int n_layers; };

__global__ void siren_pack_kernel(PackArgs pa, float* out, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    long off = 0;
    for (int l = 0; l < pa.n_layers; ++l) {
        const int K = pa.dims[l], M = pa.dims[l + 1];
        const bool head = (l == pa.n_layers - 1) && M <= 4;
        const long wsz = head ? (long)M * (pad8(K) / 8) * 8 : (long)(pad8(K) / 2) * (pad32(M) / 32) * 64;
        const long bsz = head ? 4 : (long)(pad32(M) / 32) * 32;
        if (i < off + wsz) {
            const long j = i - off;
            int m, k;
            if (head) {
                const int r = (int)(j & 3), hf = (int)((j >> 2) & 1);
                const long t = j >> 3;
                const int KQ = pad8(K) / 8;
                const int q = (int)(t % KQ);
                m = (int)(t / KQ);
                k = 8 * q + 4 * hf + r;
            } else {
                const int lane = (int)(j & 63);
                const long t2 = j >> 6;
                const int MT = pad32(M) / 32;
                const int t = (int)(t2 % MT), s = (int)(t2 / MT);
                m = 32 * t + (lane & 31);
                k = kmap(s, lane >> 5);
            }
            out[i] = (m < M && k < K) ? pa.w[l][(long)m * K + k] : 0.f;
            return;
        }
        off += wsz;
        if (i < off + bsz) {
            const long j = i - off;
            int m;
            if (head) m = (int)j;
            else { const int hf = (int)(j & 1); const int r = (int)((j >> 1) & 15); const int t = (int)(j >> 5); m = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hf; }
            out[i] = (m < M) ? pa.b[l][m] : 0.f;
            return;
        }
        off += bsz;
    }
}

extern "C" long motif_siren_pack(const float* const* w, const float* const* b, const int* dims, int n_layers,
                                 float* packed, void* stream) {
    if (!dims || n_layers < 1 || n_layers > 6) return MOTIF_EINVAL;
    long total = 0;
    for (int l = 0; l < n_layers; ++l) {
        const int K = dims[l], M = dims[l + 1];
        const bool head = (l == n_layers - 1) && M <= 4;
        total += head ? valu_head_floats(K, M) : mfma_layer_floats(K, M);
    }
    if (!packed) return total;
    if (!w || !b) return MOTIF_EINVAL;
    PackArgs pa;
    for (int l = 0; l < n_layers; ++l) { pa.w[l] = w[l]; pa.b[l] = b[l]; }
    for (int l = 0; l <= n_layers; ++l) pa.dims[l] = dims[l];
    pa.n_layers = n_layers;
    siren_pack_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(pa, packed, total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(long)e - 1000;
    return total;
}

template <int MODE, int TP, bool PRE>
static int launch_siren(const SirenArgs& a_in, void* stream) {
    using L = Layout<MODE>;
    const size_t lds = (size_t)L::LDS_FLOATS * 4;
    static_assert(L::LDS_FLOATS * 4 <= 160 * 1024, "packed network must fit the 160 KB LDS");
    static_assert(L::LDS_FLOATS % 4 == 0, "blob prefix must be float4 sized");
    SirenArgs a = a_in;
    a.stagger = 2;
    if (const int sv = motif_opt(MOTIF_OPT_SIREN_STAGGER)) a.stagger = sv < 0 ? 0 : sv;
    hipError_t e = hipFuncSetAttribute((const void*)siren_kernel<MODE, TP, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long Q = (long)a.HH * a.WW;
    const long tiles = (long)a.NB * ((Q + 32 * TP - 1) / (32 * TP));
    long blocks = (tiles + SIREN_WAVES - 1) / SIREN_WAVES;
    if (blocks > cus) blocks = cus;
    siren_kernel<MODE, TP, PRE><<<(int)blocks, SIREN_THREADS, lds, (hipStream_t)stream>>>(a);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

extern "C" int motif_siren_imnet_fwd(const float* packed, const float* feat_lr, const int32_t* iy, const int32_t* ix,
                                     const float* rel_y, const float* rel_x, float* out,
                                     int B2, int H, int W, int HH, int WW, int pre, void* stream) {
    if (!packed || !feat_lr || !iy || !ix || !rel_y || !rel_x || !out || B2 < 1) return MOTIF_EINVAL;
    SirenArgs a{packed, feat_lr, nullptr, iy, ix, rel_y, rel_x, nullptr, out, B2, 1, B2, H, W, HH, WW};
    if (pre == 2 || pre == 3) return motif_siren_split_launch(MODE_IMNET, a, stream, pre == 3 ? 2 : 3);
    return pre ? launch_siren<MODE_IMNET, SIREN_TP_IMNET, true>(a, stream) : launch_siren<MODE_IMNET, SIREN_TP_IMNET, false>(a, stream);
}

extern "C" int motif_siren_imnet_add_fwd(const float* packed, const float* feat_lr, const float* add_lr, const int32_t* iy, const int32_t* ix,
                                         const float* rel_y, const float* rel_x, float* out,
                                         int B2, int H, int W, int HH, int WW, int pre, void* stream) {
    if (!add_lr) return motif_siren_imnet_fwd(packed, feat_lr, iy, ix, rel_y, rel_x, out, B2, H, W, HH, WW, pre, stream);
    if (!packed || !feat_lr || !iy || !ix || !rel_y || !rel_x || !out || B2 < 1) return MOTIF_EINVAL;
    if (pre != 2 && pre != 3) return MOTIF_EINVAL;    // the added LR term exists for the split (16-bit matrix core) engines only
    SirenArgs a{packed, feat_lr, nullptr, iy, ix, rel_y, rel_x, nullptr, out, B2, 1, B2, H, W, HH, WW};
    a.add_lr = add_lr;
    return motif_siren_split_launch(MODE_IMNET, a, stream, pre == 3 ? 2 : 3);
}

extern "C" int motif_siren_flow_fwd(const float* packed, const float* flowfeat_lr, const int32_t* iy, const int32_t* ix,
                                    const float* rel_y, const float* rel_x, const float* times, float* pred,
                                    int B2, int N, int H, int W, int HH, int WW, int pre, void* stream) {
    if (!packed || !flowfeat_lr || !iy || !ix || !rel_y || !rel_x || !times || !pred || B2 < 2 || (B2 & 1) || N < 1) return MOTIF_EINVAL;
    SirenArgs a{packed, flowfeat_lr, nullptr, iy, ix, rel_y, rel_x, times, pred, B2 * N, N, B2 / 2, H, W, HH, WW};
    if (pre == 2 || pre == 3) return motif_siren_split_launch(MODE_FLOW, a, stream, pre == 3 ? 2 : 3);
    return pre ? launch_siren<MODE_FLOW, SIREN_TP_FLOW, true>(a, stream) : launch_siren<MODE_FLOW, SIREN_TP_FLOW, false>(a, stream);
}

extern "C" int motif_siren_synth_fwd(const float* packed, const float* acc, const float* residual_lr,
                                     const int32_t* iy, const int32_t* ix, const float* times, float* frames,
                                     int B, int N, int H, int W, int HH, int WW, int pre, uint32_t* status, void* stream) {
    if (!packed || !acc || !residual_lr || !iy || !ix || !times || !frames || B < 1 || N < 1) return MOTIF_EINVAL;
    SirenArgs a{packed, residual_lr, acc, iy, ix, nullptr, nullptr, times, frames, B * N, N, B, H, W, HH, WW};
    a.status = status;
    if (pre == 2 || pre == 3) return motif_siren_split_launch(MODE_SYNTH, a, stream, pre == 3 ? 2 : 3);
    return pre ? launch_siren<MODE_SYNTH, SIREN_TP_SYNTH, true>(a, stream) : launch_siren<MODE_SYNTH, SIREN_TP_SYNTH, false>(a, stream);
}

extern "C" int motif_siren_synth_pre_fwd(const float* packed, const float* acc, const float* residual_l0,
                                         const int32_t* iy, const int32_t* ix, const float* times, float* frames,
                                         int B, int N, int H, int W, int HH, int WW, int pre, uint32_t* status, void* stream) {
    if (!packed || !acc || !residual_l0 || !iy || !ix || !times || !frames || B < 1 || N < 1 || (pre != 2 && pre != 3)) return MOTIF_EINVAL;
    SirenArgs a{packed, residual_l0, acc, iy, ix, nullptr, nullptr, times, frames, B * N, N, B, H, W, HH, WW};
    a.status = status;
    return motif_siren_split_launch(MODE_SYNTHC, a, stream, pre == 3 ? 2 : 3);
}

// parity aid: the 198-channel synth input, materialised (never used on the product path)
__global__ void synth_input_kernel(const float* acc, const float* res_lr, const int32_t* iy, const int32_t* ix,
                                   const float* times, float* out, int B, int N, int H, int W, int HH, int WW) {
    const long Q = (long)HH * WW;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int img = blockIdx.y;
    if (p >= Q) return;
    const int Y = (int)(p / WW), X = (int)(p % WW);
    const long lr = (long)iy[Y] * W + ix[X];
    const long HWl = (long)H * W;
    const float* A = acc + (long)img * 133 * Q + p;
    float wz = A[130 * Q];
    const float cnt = A[132 * Q];
    if (wz == 0.f) wz = 1.0f;
    const float cnt_ = (cnt == 0.f) ? 1.0f : cnt;
    const float wz_ = (wz == 1.0f) ? 0.f : wz;
    float* O = out + (long)img * 198 * Q + p;
    for (int k = 0; k < 130; ++k) O[(long)k * Q] = A[(long)k * Q] / wz;
    O[130 * Q] = A[131 * Q];
    O[131 * Q] = cnt / 16.0f;
    O[132 * Q] = wz_ / cnt_;
    const float* R = res_lr + (long)(img / N) * 64 * HWl + lr;
    for (int k = 0; k < 64; ++k) O[(long)(133 + k) * Q] = R[(long)k * HWl];
    O[197 * Q] = times[img];
}

extern "C" int motif_synth_input_fwd(const float* acc, const float* residual_lr, const int32_t* iy, const int32_t* ix,
                                     const float* times, float* out, int B, int N, int H, int W, int HH, int WW, void* stream) {
    if (!acc || !residual_lr || !iy || !ix || !times || !out) return MOTIF_EINVAL;
    const long Q = (long)HH * WW;
    dim3 grid(cdiv(Q, 256), B * N);
    synth_input_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(acc, residual_lr, iy, ix, times, out, B, N, H, W, HH, WW);
    MOTIF_LAUNCH_CHECK();
    return MOTIF_OK;
}

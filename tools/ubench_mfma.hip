// Micro-benchmark: what v_mfma_f32_32x32x16_bf16 sustains on an MI355X under load (clock included), as a ceiling for
// conv_split's roofline fraction.  Variants: waves per SIMD (1 / 2), accumulators per wave (4 / 8), and the operand
// traffic of conv_split's tap beside the MFMAs (6 or 12 ds_read_b128 + 6 global b128 loads per 24 / 48 MFMAs).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mfma tools/ubench_mfma.hip && tools/ubench_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool TRAFFIC, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void mfma_kernel(const u32x4* w, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[2048];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 a[6], b[NACC == 4 ? 6 : 12];
#pragma unroll
    for (int i = 0; i < 6; ++i) a[i] = w[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < (NACC == 4 ? 6 : 12); ++i) b[i] = lds[i * 64 + lane];
    for (int it = 0; it < iters; ++it) {
        if (TRAFFIC) {
#pragma unroll
            for (int i = 0; i < 6; ++i) a[i] = w[((it & 7) * 6 + i) * 64 + lane];
#pragma unroll
            for (int i = 0; i < (NACC == 4 ? 6 : 12); ++i) b[i] = lds[((it & 1) * 12 + i) * 64 + lane];
        }
        // 6 products x NACC accumulators, like one tap of conv_split
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(k + (i & 1) * 3) % 6]),
                                                                 __builtin_bit_cast(bf16x8, b[(k % 3) * (NACC / 2) + (i >> 1)]), acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool TRAFFIC, int WPE>
void run(const char* name, const u32x4* w, float* out, int blocks) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_kernel<NACC, TRAFFIC, WPE><<<blocks, 256>>>(w, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_kernel<NACC, TRAFFIC, WPE><<<blocks, 256>>>(w, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 6 * NACC * 2.0 * 32 * 32 * 16;
    printf("%-58s %8.3f ms  %7.1f TFLOP/s bf16  (= %6.1f fp32-equivalent at 6 products)  %5.1f cycles/MFMA/SIMD at 2.4 GHz\n", name, ms,
           flop / ms / 1e9, flop / ms / 1e9 / 6, ms * 1e-3 * 2.4e9 / ((double)iters * 6 * NACC * (blocks / 256.0)));
}


// conv_split's chunk structure around the same MFMA stream: STAGE 1 = + 24 coalesced global dword loads per 9 taps (8 at taps 0..2),
// 2 = + the 3-way bf16 split of those 24 values and 9 ds_write_b128 (taps 4, 6, 8), 3 = + one __syncthreads per 9 taps.
__device__ __forceinline__ unsigned pk2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 p = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, p);
}
template <int STAGE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void chunk_kernel(const u32x4* w, const float* x, float* out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];          // 2 x 3 x 684 slots like conv_split
    const int lane = threadIdx.x & 63, tid = threadIdx.x;
    for (int i = tid; i < 2 * 3 * 684; i += 256) lds[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float pre[3][8];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) pre[j][q] = 1.0f;
    int cur = 0;
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            u32x4 a[6], b[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) a[i] = w[(((c * 9 + t) & 7) * 6 + i) * 64 + lane];
            if (STAGE >= 1 && t < 3) {
#pragma unroll
                for (int q = 0; q < 8; ++q) pre[t][q] = x[(long)((c & 3) * 8 + q) * 57600 + blockIdx.x * 64 + (tid + 256 * t) % 340];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 6; ++i) b[i] = lds[(cur * 3 + i / 2) * 684 + (i & 1) * 34 + (t / 3) * 34 + (t % 3) + lane % 32 + (lane / 32) * 340];
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(k + (i & 1) * 3) % 6]),
                                                                     __builtin_bit_cast(bf16x8, b[(k % 3) * 2 + (i >> 1)]), acc[i], 0, 0, 0);
            if (STAGE >= 2 && (t == 4 || t == 6 || t == 8)) {
                const int j = (t - 4) / 2;
                u32x4 parts[3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x0 = pre[j][2 * q], x1 = pre[j][2 * q + 1];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const unsigned pk = pk2(x0, x1);
                        parts[p][q] = pk;
                        if (p < 2) { x0 -= __builtin_bit_cast(float, pk << 16); x1 -= __builtin_bit_cast(float, pk & 0xffff0000u); }
                    }
                }
#pragma unroll
                for (int p = 0; p < 3; ++p) lds[((cur ^ 1) * 3 + p) * 684 + (tid + 256 * j) % 680] = parts[p];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (STAGE >= 3) __syncthreads();
        cur ^= 1;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s + pre[0][0] + pre[1][1] + pre[2][2];
}

template <int STAGE>
void run_chunk(const char* name, const u32x4* w, const float* x, float* out) {
    const int chunks = 400, blocks = 512;
    const size_t ldsb = 2 * 3 * 684 * 16;
    hipFuncSetAttribute((const void*)chunk_kernel<STAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    chunk_kernel<STAGE><<<blocks, 256, ldsb>>>(w, x, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chunk_kernel<STAGE><<<blocks, 256, ldsb>>>(w, x, out, chunks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * chunks * 9 * 24 * 2.0 * 32 * 32 * 16;
    printf("%-58s %8.3f ms  %7.1f TFLOP/s bf16  (= %6.1f fp32-equivalent)  %5.2f kcyc per chunk at 2.4 GHz\n", name, ms, flop / ms / 1e9,
           flop / ms / 1e9 / 6, ms * 1e-3 * 2.4e9 / chunks / 1e3);
}

int main() {
    u32x4* w; float* out;
    hipMalloc(&w, 8 * 6 * 64 * 16);
    hipMemset(w, 0x3f, 8 * 6 * 64 * 16);
    hipMalloc(&out, 1024 * 256 * 4);
    run<4, false, 1>("1 wave/SIMD, 4 acc, MFMA only", w, out, 256);
    run<4, false, 2>("2 waves/SIMD, 4 acc, MFMA only", w, out, 512);
    run<8, false, 1>("1 wave/SIMD, 8 acc, MFMA only", w, out, 256);
    run<4, true, 2>("2 waves/SIMD, 4 acc, +6 ds_read_b128 +6 global b128 per 24", w, out, 512);
    run<8, true, 1>("1 wave/SIMD, 8 acc, +12 ds_read_b128 +6 global b128 per 48", w, out, 256);
    run<4, true, 1>("1 wave/SIMD, 4 acc, +6 ds_read_b128 +6 global b128 per 24", w, out, 256);
    float* x;
    hipMalloc(&x, 32L * 57600 * 4 + (1 << 20));
    hipMemset(x, 0, 32L * 57600 * 4 + (1 << 20));
    run_chunk<0>("chunk loop (2 blocks/CU): MFMA + operand reads", w, x, out);
    run_chunk<1>("  + 24 staged global loads per chunk", w, x, out);
    run_chunk<2>("  + 3-way split and 9 ds_write_b128 per chunk", w, x, out);
    run_chunk<3>("  + one barrier per chunk", w, x, out);
    return 0;
}

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
    return 0;
}

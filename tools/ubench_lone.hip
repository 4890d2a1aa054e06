// Micro-benchmark for the one-wave-per-SIMD kernels (conv_wino.hip): what a LONE wave sustains when v_mfma_f32_32x32x16_bf16 is
// interleaved with filler instructions -- cycles per MFMA (s_memtime, shader clock) and wall time, for
//   accumulators in AGPRs / VGPRs, 2 / 4 / 8 accumulators in rotation, F independent or dependent vector instructions per MFMA,
//   conversions, packed adds, LDS reads / writes, buffer loads.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_lone tools/ubench_lone.hip && tools/ubench_lone
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { F_NONE, F_VALU_IND, F_VALU_DEP, F_CVT, F_PKADD, F_DSREAD, F_DSWRITE, F_VMEM, F_SPLIT, F_DSREAD_W, F_WINO };

template <bool AGPR>
__device__ __forceinline__ void mfma(f32x16& acc, const u32x4& a, const u32x4& b) {
    if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

template <int NACC, bool AGPR, int KIND, int F, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void lone_kernel(const u32x4* w, float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 64 * WAVES) lds[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 a = w[lane], b = w[64 + lane];
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (float)(lane + i);
    float y = 1.0001f;
    unsigned pk[4] = {0, 0, 0, 0};
    u32x4 ld[4] = {a, a, a, a};
    f32x4 gl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gl[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4* lp = lds + lane + (threadIdx.x >> 6) * 512;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            mfma<AGPR>(acc[m % NACC], a, b);
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const int u = (m * F + f);
                if constexpr (KIND == F_VALU_IND) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[u % 8]) : "v"(y));
                else if constexpr (KIND == F_VALU_DEP) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(y));
                else if constexpr (KIND == F_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[u % 4]) : "v"(x[u % 8]), "v"(x[(u + 1) % 8]));
                else if constexpr (KIND == F_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&x[2 * (u % 4)]) : "v"(*(double*)&x[2 * ((u + 1) % 4)]));
                else if constexpr (KIND == F_DSREAD) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[u % 4]) : "v"((unsigned)(size_t)lp), "n"(0));
                else if constexpr (KIND == F_DSWRITE) asm volatile("ds_write_b128 %0, %1 offset:16384" : : "v"((unsigned)(size_t)lp), "v"(ld[u % 4]) : "memory");
                else if constexpr (KIND == F_VMEM) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gl[u % 4]) : "v"(w + lane + 64 * (u % 8)));
                else if constexpr (KIND == F_WINO) {   // F = 1: operands only; 2: + 4 v_add per MFMA; 3: + those + a ds_write_b128 every 6 MFMAs
                    if (f == 0) {
                        if (m < 12) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[m % 4]) : "v"((unsigned)(size_t)lp), "n"(0));
                        else if (m < 18) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gl[m % 4]) : "v"(w + lane + 64 * (m % 8)));
                        if (m == 11) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
                        if (m == 23) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    } else {
#pragma unroll
                        for (int v = 0; v < 4; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(4 * m + v) % 8]) : "v"(y));
                        if (F == 3 && f == 2 && m % 6 == 3) asm volatile("ds_write_b128 %0, %1 offset:16384" : : "v"((unsigned)(size_t)lp), "v"(ld[0]) : "memory");
                    }
                }
                else if constexpr (KIND == F_SPLIT) {   // one step of the 3-way split chain on pair u % 4: cvt, expand, expand, sub, sub (dependent within the pair)
                    const int q = u % 4, st = (u / 4) % 4;
                    if (st == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[q]) : "v"(x[2 * q]), "v"(x[2 * q + 1]));
                    else if (st == 1) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(ld[q][0]) : "v"(pk[q]));
                    else if (st == 2) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(ld[q][1]) : "v"(pk[q]));
                    else asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[2 * q]) : "v"(ld[q][0]));
                }
            }
            // wait for the group BEFORE the one just issued (software-pipelined: no latency exposed, only issue / throughput cost)
            if constexpr (KIND == F_VMEM) { if (m % 12 == 11) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(12 * F) : "memory"); }
            if constexpr (KIND == F_DSREAD || KIND == F_DSWRITE) { if (m % 6 == 5) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(6 * F) : "memory"); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = y;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (float)pk[i] + (float)ld[i][0] + (float)ld[i][1] + gl[i][0];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, bool AGPR, int KIND, int F, int WAVES = 4>
void run(const char* name, const u32x4* w, float* out, long long* cyc) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    lone_kernel<NACC, AGPR, KIND, F, WAVES><<<blocks, 64 * WAVES>>>(w, out, cyc, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    lone_kernel<NACC, AGPR, KIND, F, WAVES><<<blocks, 64 * WAVES>>>(w, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < blocks; ++i) c += (double)h[i];
    c /= blocks;
    const double nm = (double)iters * 24 * (WAVES / 4);
    printf("%-64s %7.3f ms  %6.1f cyc/MFMA/SIMD  clock %.2f GHz  %6.0f TFLOP/s bf16\n", name, ms, c / nm, c / (ms * 1e6),
           (double)blocks * WAVES * iters * 24 * 2.0 * 32 * 32 * 16 / ms / 1e9);
}

int main() {
    u32x4* w; float* out; long long* cyc;
    hipMalloc(&w, 1 << 20); hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 4096);
    hipMemset(w, 0x3f, 1 << 20);
    run<2, true, F_NONE, 0>("MFMA only, 2 acc (AGPR)", w, out, cyc);
    run<4, true, F_NONE, 0>("MFMA only, 4 acc (AGPR)", w, out, cyc);
    run<8, true, F_NONE, 0>("MFMA only, 8 acc (AGPR)", w, out, cyc);
    run<2, false, F_NONE, 0>("MFMA only, 2 acc (VGPR)", w, out, cyc);
    run<4, false, F_NONE, 0>("MFMA only, 4 acc (VGPR)", w, out, cyc);
    run<8, false, F_NONE, 0>("MFMA only, 8 acc (VGPR)", w, out, cyc);
    run<4, true, F_NONE, 0, 8>("MFMA only, 4 acc (AGPR), TWO waves per SIMD", w, out, cyc);
    run<4, true, F_VALU_IND, 1>("4 acc + 1 independent v_add_f32 per MFMA", w, out, cyc);
    run<4, true, F_VALU_IND, 2>("4 acc + 2 independent v_add_f32 per MFMA", w, out, cyc);
    run<4, true, F_VALU_IND, 3>("4 acc + 3 independent v_add_f32 per MFMA", w, out, cyc);
    run<4, true, F_VALU_IND, 4>("4 acc + 4 independent v_add_f32 per MFMA", w, out, cyc);
    run<4, true, F_VALU_IND, 6>("4 acc + 6 independent v_add_f32 per MFMA", w, out, cyc);
    run<4, true, F_VALU_IND, 8>("4 acc + 8 independent v_add_f32 per MFMA", w, out, cyc);
    run<4, true, F_VALU_DEP, 2>("4 acc + 2 DEPENDENT v_add_f32 per MFMA", w, out, cyc);
    run<4, true, F_VALU_DEP, 4>("4 acc + 4 DEPENDENT v_add_f32 per MFMA", w, out, cyc);
    run<4, true, F_CVT, 2>("4 acc + 2 v_cvt_pk_bf16_f32 per MFMA", w, out, cyc);
    run<4, true, F_CVT, 4>("4 acc + 4 v_cvt_pk_bf16_f32 per MFMA", w, out, cyc);
    run<4, true, F_PKADD, 2>("4 acc + 2 v_pk_add_f32 per MFMA", w, out, cyc);
    run<4, true, F_PKADD, 4>("4 acc + 4 v_pk_add_f32 per MFMA", w, out, cyc);
    run<4, true, F_SPLIT, 2>("4 acc + 2 split-chain steps per MFMA", w, out, cyc);
    run<4, true, F_SPLIT, 4>("4 acc + 4 split-chain steps per MFMA", w, out, cyc);
    run<4, true, F_DSREAD, 1>("4 acc + 1 ds_read_b128 per MFMA", w, out, cyc);
    run<4, true, F_DSREAD, 2>("4 acc + 2 ds_read_b128 per MFMA", w, out, cyc);
    run<4, true, F_DSWRITE, 1>("4 acc + 1 ds_write_b128 per MFMA", w, out, cyc);
    run<4, true, F_DSWRITE, 2>("4 acc + 2 ds_write_b128 per MFMA", w, out, cyc);
    run<4, true, F_VMEM, 1>("4 acc + 1 global_load_dwordx4 (L1/L2 hit) per MFMA", w, out, cyc);
    run<4, true, F_VMEM, 2>("4 acc + 2 global_load_dwordx4 (L1/L2 hit) per MFMA", w, out, cyc);
    run<4, true, F_VMEM, 3>("4 acc + 3 global_load_dwordx4 (L1/L2 hit) per MFMA", w, out, cyc);
    run<4, true, F_WINO, 1>("4 acc + operand pattern of a conv_wino super-step (12 ds_read_b128 + 6 loads per 24 MFMAs)", w, out, cyc);
    run<4, true, F_WINO, 2>("  the same + 4 independent v_add_f32 per MFMA", w, out, cyc);
    run<4, true, F_WINO, 3>("  the same + 1 ds_write_b128 per 6 MFMAs", w, out, cyc);
    run<4, true, F_VALU_IND, 4, 8>("4 acc + 4 independent v_add_f32, TWO waves per SIMD", w, out, cyc);
    run<4, false, F_VALU_IND, 4>("4 acc (VGPR) + 4 independent v_add_f32 per MFMA", w, out, cyc);
    return 0;
}

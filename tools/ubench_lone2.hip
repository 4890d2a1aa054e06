// Follow-up to ubench_lone.hip: which part of conv_wino's filler mix stalls a lone wave's MFMA stream.  Per 24 MFMAs: RD ds_read_b128,
// LD global_load_dwordx4, WR LDS writes (kind WK: 0 b128, 1 two b64, 2 four b32) at spacing / bunching WS, VA v_add_f32 per MFMA,
// DMA global_load_lds_dwordx4; DEPHASE staggers the four waves of a workgroup.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_lone2 tools/ubench_lone2.hip && tools/ubench_lone2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int RD, int LD, int WR, int WK, int BUNCH, int VA, int DMA, int DEPHASE>
__global__ __launch_bounds__(256) void k(const u32x4* w, float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[8192];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 a = w[lane], b = w[64 + lane];
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (float)(lane + i);
    float y = 1.0001f;
    u32x4 ld[4] = {a, a, a, a};
    f32x4 gl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gl[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned lp = (unsigned)(size_t)(lds + lane + wave * 512);
    if (DEPHASE) for (int i = 0; i < wave * DEPHASE; ++i) __builtin_amdgcn_s_sleep(1);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m % 4]) : "v"(a), "v"(b));
            if (m < RD) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[m % 4]) : "v"(lp));
            else if (m < RD + LD) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gl[m % 4]) : "v"(w + lane + 64 * (m % 8)));
            else if (m < RD + LD + DMA) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + lane + 64 * (m % 8)),
                                                 (__attribute__((address_space(3))) void*)(lds + 4096 + wave * 1024 + (m % 4) * 64), 16, 0, 0);
            }
            bool wr = false;
            if constexpr (WR > 0) { constexpr int PER = 24 / (WR > 0 ? WR : 1); if (BUNCH) wr = m >= 24 - WR; else wr = (m % PER) == PER / 2; }
            if (wr) {
                if (WK == 0) asm volatile("ds_write_b128 %0, %1 offset:32768" :: "v"(lp), "v"(ld[0]) : "memory");
                else if (WK == 1) { typedef double f64x2 __attribute__((ext_vector_type(2))); const f64x2 dd = __builtin_bit_cast(f64x2, ld[0]); asm volatile("ds_write_b64 %0, %1 offset:32768" :: "v"(lp), "v"(dd[0]) : "memory"); asm volatile("ds_write_b64 %0, %1 offset:32776" :: "v"(lp), "v"(dd[1]) : "memory"); }
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(lp), "v"(ld[0][e]), "n"(32768) : "memory");
                }
            }
#pragma unroll
            for (int v = 0; v < VA; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(4 * m + v) % 8]) : "v"(y));
            if (m == RD - 1 && RD > 0) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(RD > 15 ? 15 : RD) : "memory");
            if (m == 23 && LD + DMA > 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LD + DMA) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = y;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (float)ld[i][0] + (float)ld[i][1] + gl[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)lds[4096 + threadIdx.x][0];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int RD, int LD, int WR, int WK, int BUNCH, int VA, int DMA, int DEPHASE>
void run(const char* name, const u32x4* w, float* out, long long* cyc) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<RD, LD, WR, WK, BUNCH, VA, DMA, DEPHASE><<<blocks, 256>>>(w, out, cyc, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<RD, LD, WR, WK, BUNCH, VA, DMA, DEPHASE><<<blocks, 256>>>(w, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < blocks; ++i) c += (double)h[i];
    c /= blocks;
    printf("%-72s %7.3f ms  %6.1f cyc/MFMA  clock %.2f GHz\n", name, ms, c / ((double)iters * 24), c / (ms * 1e6));
}

int main() {
    u32x4* w; float* out; long long* cyc;
    hipMalloc(&w, 1 << 20); hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 4096);
    hipMemset(w, 0x3f, 1 << 20);
    //   RD LD WR WK BUNCH VA DMA DEPHASE
    run<12, 6, 0, 0, 0, 0, 0, 0>("12 rd + 6 ld", w, out, cyc);
    run<12, 6, 0, 0, 0, 3, 0, 0>("12 rd + 6 ld + 3 valu", w, out, cyc);
    run<12, 6, 4, 0, 0, 3, 0, 0>("12 rd + 6 ld + 3 valu + 4 wr b128 spread", w, out, cyc);
    run<12, 6, 4, 0, 1, 3, 0, 0>("12 rd + 6 ld + 3 valu + 4 wr b128 bunched", w, out, cyc);
    run<12, 6, 4, 1, 0, 3, 0, 0>("12 rd + 6 ld + 3 valu + 4 x 2 wr b64 spread", w, out, cyc);
    run<12, 6, 4, 2, 0, 3, 0, 0>("12 rd + 6 ld + 3 valu + 4 x 4 wr b32 spread", w, out, cyc);
    run<12, 6, 2, 0, 0, 3, 0, 0>("12 rd + 6 ld + 3 valu + 2 wr b128 spread", w, out, cyc);
    run<12, 6, 1, 0, 0, 3, 0, 0>("12 rd + 6 ld + 3 valu + 1 wr b128", w, out, cyc);
    run<12, 0, 4, 0, 0, 3, 0, 0>("12 rd + 0 ld + 3 valu + 4 wr b128 spread", w, out, cyc);
    run<0, 6, 4, 0, 0, 3, 0, 0>("0 rd + 6 ld + 3 valu + 4 wr b128 spread", w, out, cyc);
    run<0, 0, 4, 0, 0, 3, 0, 0>("0 rd + 0 ld + 3 valu + 4 wr b128 spread", w, out, cyc);
    run<0, 0, 4, 0, 0, 0, 0, 0>("0 rd + 0 ld + 0 valu + 4 wr b128 spread", w, out, cyc);
    run<12, 6, 4, 0, 0, 3, 0, 3>("12 rd + 6 ld + 3 valu + 4 wr b128 spread, waves de-phased", w, out, cyc);
    run<12, 6, 0, 0, 0, 3, 2, 0>("12 rd + 6 ld + 3 valu + 2 LDS-DMA b128", w, out, cyc);
    run<12, 6, 2, 0, 0, 3, 2, 0>("12 rd + 6 ld + 3 valu + 2 wr b128 + 2 LDS-DMA b128", w, out, cyc);
    run<12, 4, 2, 0, 0, 3, 0, 0>("12 rd + 4 ld + 3 valu + 2 wr b128", w, out, cyc);
    run<12, 8, 0, 0, 0, 3, 0, 0>("12 rd + 8 ld + 3 valu", w, out, cyc);
    run<12, 10, 0, 0, 0, 3, 0, 0>("12 rd + 10 ld + 3 valu", w, out, cyc);
    return 0;
}

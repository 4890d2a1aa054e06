// What does the shader clock do under a chip-wide MFMA load?  One wave per SIMD on every CU issues back-to-back
// v_mfma_f32_32x32x16_bf16 for a few milliseconds; each workgroup records the shader-clock counter (s_memtime) and the constant 100 MHz
// counter (s_memrealtime) at both ends.  MHz = cycles / realtime.  DUTY: MFMAs issued per 8 slots (the rest s_nop of the same length), to
// see the clock against matrix-core duty; DATA: 0 = operands of zeros, 1 = random bit patterns (toggle rate drives the power).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_clock tools/ubench_clock.hip && tools/ubench_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DUTY>
__global__ __launch_bounds__(256) void k(const u32x4* w, float* out, long long* stamps, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const u32x4 a = w[lane], b = w[64 + lane];
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            if ((m & 7) < DUTY) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m % 4]) : "v"(a), "v"(b));
            else asm volatile("s_nop 15\n\ts_nop 15");
        }
    }
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = c1 - c0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int DUTY>
void run(const u32x4* w, float* out, long long* st, int blocks, int iters, const char* what) {
    std::vector<long long> h(blocks * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        k<DUTY><<<blocks, 256>>>(w, out, st, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0;
        for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
        cyc /= blocks; rt /= blocks;
        const double mfma = (double)iters * 32 * DUTY / 8;
        printf("%-18s duty %d/8 blocks %4d: %.3f ms  shader clock %.0f MHz  cycles per MFMA slot %.1f  chip %.0f TFLOP/s bf16\n", what, DUTY, blocks, ms,
               cyc / (rt / 100.0), cyc / (iters * 32.0), mfma * 32768.0 * 4 * blocks / (rt / 100.0 * 1e-6) / 1e12);
    }
}

int main() {
    u32x4 *w; float* out; long long* st;
    hipMalloc(&w, 128 * 16); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&st, 1024 * 16);
    std::vector<unsigned> h(512);
    for (int data = 0; data < 2; ++data) {
        for (auto& v : h) {
            // bf16 pairs with a sane exponent: random mantissas and signs (1), or zeros (0)
            const unsigned lo = 0x3f80u | (rand() & 0x807f), hi = 0x3f80u | (rand() & 0x807f);
            v = data ? (hi << 16 | lo) : 0u;
        }
        hipMemcpy(w, h.data(), 2048, hipMemcpyHostToDevice);
        const char* what = data ? "random operands" : "zero operands";
        run<8>(w, out, st, 256, 4000, what);
        run<6>(w, out, st, 256, 4000, what);
        run<4>(w, out, st, 256, 4000, what);
        run<2>(w, out, st, 256, 4000, what);
        run<8>(w, out, st, 32, 4000, what);
    }
    return 0;
}

// Micro-benchmark: LDS / global atomic throughput on gfx950 (design input for the splat kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(1024) void lds_kernel(float* out, int iters, int stride) {
    __shared__ float tile[16384];
    __shared__ unsigned cnt[4096];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) tile[i] = 0.f;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const int base = (threadIdx.x * stride) & 16383;
    float v = 1.0f + threadIdx.x * 1e-6f;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        const int a = (base + it * 67) & 16383;
        if (MODE == 0) atomicAdd(&tile[a], v);                       // ds_add_f32
        else if (MODE == 1) acc += atomicAdd(&cnt[a & 4095], 1u);    // ds_add_rtn_u32
        else if (MODE == 2) atomicAdd(&cnt[a & 4095], 1u);           // ds_add_u32
        else if (MODE == 3) tile[a] += v;                            // plain RMW (racy, timing only)
        else if (MODE == 4) atomicMax((int*)&tile[a], __float_as_int(v));
    }
    __syncthreads();
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = tile[threadIdx.x] + cnt[threadIdx.x] + acc;
}

// 64-bit vs 32-bit integer LDS atomics on consecutive cells (lane l -> cell base + l), the splat accumulate pattern
template <int MODE>
__global__ __launch_bounds__(1024) void lds64_kernel(float* out, int iters) {
    __shared__ unsigned long long t64[8192];
    unsigned* t32 = (unsigned*)t64;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) t64[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long v = 0x100000001ull * (threadIdx.x + 1);
    for (int it = 0; it < iters; ++it) {
        const int a = (wave * 67 + it * 131 + lane) & 4095;
        if (MODE == 0) atomicAdd(&t64[a], v);                                            // ds_add_u64, consecutive 8-byte cells
        else if (MODE == 1) atomicAdd(&t32[a], (unsigned)v);                             // ds_add_u32, consecutive 4-byte cells
        else if (MODE == 2) { atomicAdd(&t32[a], (unsigned)v); atomicAdd(&t32[4096 + a], (unsigned)(v >> 32)); }   // two planes of u32
        else if (MODE == 3) atomicAdd(&t32[2 * a], (unsigned)v);                         // u32 at 8-byte stride
        else if (MODE == 4) atomicAdd(&t64[(a & ~63) + ((lane * 2) & 63) + (lane >> 5)], v);   // u64, even cells then odd cells
        else if (MODE == 5) t64[a] += v;                                                 // plain 64-bit read + write (racy, timing only)
    }
    __syncthreads();
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = (float)t64[threadIdx.x];
}

__global__ void glob_kernel(float* buf, long n, int iters, int spread) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const long a = (t + (long)it * spread) % n;
        atomicAdd(&buf[a], 1.0f);
    }
}

template <typename F>
float time_ms(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    float* out; hipMalloc(&out, 1 << 20);
    const int iters = 4096, blocks = 256;
    const char* names[] = {"ds_add_f32", "ds_add_rtn_u32", "ds_add_u32", "plain lds rmw", "ds_max_i32"};
    for (int stride = 1; stride <= 33; stride += 32) {
        printf("stride %d\n", stride);
        float ms;
        ms = time_ms([&] { lds_kernel<0><<<blocks, 1024>>>(out, iters, stride); });
        printf("  %-16s %8.3f ms  %.1f G lane-ops/s  (%.2f cycles per wave-instr per CU @2.4GHz)\n", names[0], ms, blocks * 1024.0 * iters / ms / 1e6, ms * 1e-3 * 2.4e9 / (16.0 * iters));
        ms = time_ms([&] { lds_kernel<1><<<blocks, 1024>>>(out, iters, stride); });
        printf("  %-16s %8.3f ms  %.1f G lane-ops/s  (%.2f)\n", names[1], ms, blocks * 1024.0 * iters / ms / 1e6, ms * 1e-3 * 2.4e9 / (16.0 * iters));
        ms = time_ms([&] { lds_kernel<2><<<blocks, 1024>>>(out, iters, stride); });
        printf("  %-16s %8.3f ms  %.1f G lane-ops/s  (%.2f)\n", names[2], ms, blocks * 1024.0 * iters / ms / 1e6, ms * 1e-3 * 2.4e9 / (16.0 * iters));
        ms = time_ms([&] { lds_kernel<3><<<blocks, 1024>>>(out, iters, stride); });
        printf("  %-16s %8.3f ms  %.1f G lane-ops/s  (%.2f)\n", names[3], ms, blocks * 1024.0 * iters / ms / 1e6, ms * 1e-3 * 2.4e9 / (16.0 * iters));
        ms = time_ms([&] { lds_kernel<4><<<blocks, 1024>>>(out, iters, stride); });
        printf("  %-16s %8.3f ms  %.1f G lane-ops/s  (%.2f)\n", names[4], ms, blocks * 1024.0 * iters / ms / 1e6, ms * 1e-3 * 2.4e9 / (16.0 * iters));
    }
    {
        const char* n64[] = {"ds_add_u64 consecutive", "ds_add_u32 consecutive", "2 x ds_add_u32 (two planes)", "ds_add_u32 stride 8 B", "ds_add_u64 even/odd", "plain 64-bit rmw"};
        float ms;
#define RUN64(M) ms = time_ms([&] { lds64_kernel<M><<<blocks, 1024>>>(out, iters); }); \
        printf("  %-28s %8.3f ms  (%.2f cycles per wave-iteration per CU @2.4GHz)\n", n64[M], ms, ms * 1e-3 * 2.4e9 / (16.0 * iters));
        RUN64(0) RUN64(1) RUN64(2) RUN64(3) RUN64(4) RUN64(5)
    }
    float* buf; const long n = 1L << 28; hipMalloc(&buf, n * 4); hipMemset(buf, 0, n * 4);
    for (int spread : {1, 4099, 1 << 20}) {
        const int gi = 64;
        float ms = time_ms([&] { glob_kernel<<<4096, 256>>>(buf, n, gi, spread * 256 * 16); });
        printf("global_atomic_add_f32 spread %8d: %8.3f ms  %.1f G lane-ops/s\n", spread, ms, 4096.0 * 256 * gi / ms / 1e6);
    }
    return 0;
}
